"""Per-launch MFMA utilisation / effective clock of one kernel from a PMC pass, bucketed by launch duration (the GRBM_GUI_ACTIVE
quotient reads high on short dispatches: MI355X_MICROARCH.md, DVFS give-back)."""
import csv, glob, re, sys
from collections import defaultdict
d, pat = sys.argv[1], sys.argv[2]
rows = defaultdict(dict)
for fn in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if pat not in r["Kernel_Name"]:
            continue
        k = r["Dispatch_Id"]
        rows[k][r["Counter_Name"]] = float(r["Counter_Value"])
        rows[k]["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        rows[k]["grid"] = int(r["Grid_Size"])
b = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for k, v in rows.items():
    if "GRBM_GUI_ACTIVE" not in v or "SQ_VALU_MFMA_BUSY_CYCLES" not in v:
        continue
    ms = v["ns"] * 1e-6
    key = "<0.3ms" if ms < 0.3 else "0.3-1ms" if ms < 1 else "1-3ms" if ms < 3 else "3-6ms" if ms < 6 else ">6ms"
    a = b[key]
    a[0] += 1; a[1] += v["GRBM_GUI_ACTIVE"] / 8; a[2] += v["SQ_VALU_MFMA_BUSY_CYCLES"]; a[3] += v["ns"]
for key in ("<0.3ms", "0.3-1ms", "1-3ms", "3-6ms", ">6ms"):
    if key in b:
        n, cyc, busy, ns = b[key]
        print(f"{key:8s} n={n:5d} util={busy / (1024 * cyc):.3f} clock={cyc / ns:.3f} GHz  util*clock={busy / 1024 / ns:.3f}  avg {ns / n * 1e-6:.3f} ms")
