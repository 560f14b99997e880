"""Summarise rocprofv3 --pmc passes into per-kernel HBM traffic (profiles/*_pmc_traffic.json).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/pmc_traffic_cfg2_bf16.json

The output records the sha256 of the libdowngan_hip.so the passes ran with (run this ON THE BOX, right after the passes, or
anywhere with the same in-tree build): bench.py reports `roofline.traffic` only when that hash equals the library it loads.

FETCH_SIZE and WRITE_SIZE are in KiB and need separate passes (TCC slots); on gfx950 FETCH_SIZE counts
128-B read requests as 64 B, so reads are doubled (MI355X_MICROARCH.md, HBM section):
    traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch.
"""
from __future__ import annotations

import csv
import glob
import json
import re
import sys
from collections import defaultdict


def per_kernel(d, counter):
    files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    assert files, f"no counter_collection.csv under {d}"
    tot = defaultdict(float)
    cnt = defaultdict(int)
    for fn in files:
        with open(fn) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
                tot[k] += float(row["Counter_Value"])
                cnt[k] += 1
    return tot, cnt


def main():
    dfetch, dwrite, out = sys.argv[1:4]
    ft, fc = per_kernel(dfetch, "FETCH_SIZE")
    wt, wc = per_kernel(dwrite, "WRITE_SIZE")
    res = {}
    for k in sorted(ft, key=lambda k: -(2 * ft[k] + wt.get(k, 0))):
        n = fc[k]
        rd = 2 * ft[k] * 1024 / n
        wr = wt.get(k, 0.0) * 1024 / max(wc.get(k, 0), 1)
        res[k] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                  "traffic_bytes_per_launch": rd + wr}
    import hashlib
    import os
    import subprocess
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "downgan_amd", "csrc", "libdowngan_hip.so")
    sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()
    try:
        commit = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(lib)).stdout.strip() or None
    except Exception:
        commit = None
    json.dump({"formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch (gfx950 read correction)", "lib_sha256": sha,
               "commit_at_summary": commit, "kernels": res}, open(out, "w"), indent=1)
    for k, v in list(res.items())[:12]:
        print(f"{k[:70]:70s} n={v['launches']:5d} rd={v['read_bytes_per_launch'] / 1e6:9.1f} MB wr={v['write_bytes_per_launch'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main()
