"""Summarise rocprofv3 --pmc passes into per-kernel HBM traffic (profiles/*_pmc_traffic.json).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01f_pmc_traffic.json

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
    json.dump({"formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch (gfx950 read correction)", "kernels": res},
              open(out, "w"), indent=1)
    for k, v in list(res.items())[:12]:
        print(f"{k[:70]:70s} n={v['launches']:5d} rd={v['read_bytes_per_launch'] / 1e6:9.1f} MB wr={v['write_bytes_per_launch'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main()
