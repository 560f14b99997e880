# LDS / issue counters of the halo conv kernel on one layer (run ON the GPU box: gpurun -- 'bash tools/pmc_lds.sh "G.b5" fwd')
L=${1:-G.b5}; OP=${2:-fwd}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
W=gpurun_out/pmc_lds_work; rm -rf $W; mkdir -p $W
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $W/p$i -- python3 tools/conv_bench.py --layers "$L" --ops $OP --nscale 4 --iters 3 > $W/log$i.txt 2>&1 || { echo "pass $i failed"; tail -5 $W/log$i.txt; }
done
python3 tools/pmc_any.py $W/p1 $W/p2 $W/p3 $W/p4 $W/p5 | head -12
rm -rf $W
