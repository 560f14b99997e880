"""Per-kernel means of every counter found in the rocprofv3 --pmc output directories given:  python tools/pmc_any.py <dir> [<dir> ...]
(one line per kernel: launches, then counter = value per launch; GRBM_GUI_ACTIVE / 8 = kernel cycles)."""
import csv, glob, re, sys
from collections import defaultdict
acc, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
for d in sys.argv[1:]:
    for fn in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in sorted(acc, key=lambda k: -acc[k].get("GRBM_GUI_ACTIVE", 0.0)):
    print(k[:110])
    print("   " + "  ".join(f"{c}={acc[k][c] / cnt[k][c]:.4g} (n={cnt[k][c]})" for c in sorted(acc[k])))
