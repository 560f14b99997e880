"""bench.py --per-layer JSON line -> the text table kept under profiles/ (rXX_per_layer_<workload>.txt).

    python bench.py --no-cpu-baseline --per-layer > b.json;  python tools/per_layer_table.py b.json > profiles/r02_per_layer_cfg2.txt
"""
import json
import sys


def main():
    d = json.load(open(sys.argv[1]))
    steps = d["steps"]
    print(f"# python bench.py --no-cpu-baseline --per-layer --dtype {d['dtype']}  (1x MI355X, {d['config']['workload'].split(':')[0]}; "
          f"live HIP events per launch, {steps} timed steps)")
    print(f"# {d['value']:.3f} samples/s, {d['ms_per_step']:.2f} ms/step; tag = op:Cin>Cout@H[s2|ps][:C = critic]:k<kernel mask 1 generic 2 fast "
          f"8 halo 16 im2col 32 fp8 halo>; flops = algorithmic (real channels)")
    print(f"{'op:layer':46s} {'launches':>8s} {'ms/step':>8s} {'TFLOP/s':>8s} {'alg TB/s':>8s}")
    for t, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["seconds"]):
        tb = v.get("alg_tbps")
        print(f"{t:46s} {v['launches']:8d} {v['seconds'] / steps * 1e3:8.2f} {v['tflops'] if v['tflops'] is not None else 0:8.1f} "
              f"{tb if tb is not None else 0:8.2f}")


if __name__ == "__main__":
    main()
