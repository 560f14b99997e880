"""MFMA-pipe utilisation and effective clock per kernel from one rocprofv3 PMC pass (profiles/*_mfma_util.json).

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -- python3 bench.py ...
    python tools/mfma_util.py gpurun_out/pmc_mfma profiles/r02_mfma_util_cfg2_bf16.json

Per MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over all SIMDs (4 x 256 on the chip);
GRBM_GUI_ACTIVE is the busy-cycle count summed over the 8 XCDs, so the kernel's duration in shader cycles is GRBM_GUI_ACTIVE / 8 and
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8),    clock = GRBM_GUI_ACTIVE / 8 / duration.
The kernel-trace of the same pass gives the durations.  Profiled passes run ~2-3 % slower clocks than un-profiled ones.
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def main():
    d, out = sys.argv[1:3]
    files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    assert files, d
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for fn in files:
        with open(fn) as f:
            for row in csv.DictReader(f):
                k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    cnt[k] += 1
                    if "Start_Timestamp" in row and "End_Timestamp" in row:
                        acc[k]["_ns"] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    res = {}
    for k, v in acc.items():
        gui, busy = v.get("GRBM_GUI_ACTIVE", 0.0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if gui <= 0 or busy <= 0:
            continue
        cyc = gui / 8.0
        res[k] = {"launches": cnt[k], "mfma_util": busy / (1024.0 * cyc), "kernel_cycles_per_launch": cyc / cnt[k]}
        if v.get("_ns"):
            res[k]["effective_clock_ghz"] = cyc / v["_ns"]
            res[k]["ms_per_launch"] = v["_ns"] / cnt[k] * 1e-6
    res = dict(sorted(res.items(), key=lambda kv: -kv[1]["kernel_cycles_per_launch"] * kv[1]["launches"]))
    json.dump({"formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs); clock = GRBM_GUI_ACTIVE / 8 / duration",
               "kernels": res}, open(out, "w"), indent=1)
    for k, v in list(res.items())[:10]:
        print(f"{k[:64]:64s} n={v['launches']:5d} mfma_util={v['mfma_util']:.3f} clock={v.get('effective_clock_ghz', 0):.2f} GHz")


if __name__ == "__main__":
    main()
