"""Per-kernel FETCH_SIZE / WRITE_SIZE of one rocprofv3 --pmc pass: python tools/pmc_one.py <dir> <counter>  (KiB summed / launches)."""
import csv, glob, re, sys
from collections import defaultdict
d, counter = sys.argv[1:3]
tot, cnt = defaultdict(float), defaultdict(int)
for fn in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        if row["Counter_Name"] != counter:
            continue
        k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
        tot[k] += float(row["Counter_Value"]); cnt[k] += 1
for k in sorted(tot, key=lambda k: -tot[k])[:6]:
    mult = 2 if counter == "FETCH_SIZE" else 1
    print(f"{k[:80]:80s} n={cnt[k]:4d} {counter} {mult * tot[k] * 1024 / cnt[k] / 1e9:8.3f} GB/launch")
