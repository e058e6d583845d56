"""Summarise the counter passes of tools/pmc_developed.sh: mean per dispatch of the pair kernel
over the first 40 dispatches (the lattice at rest) and over the last 40 (developed flow)."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(list)
for f in sorted(glob.glob(root + "/*/*/*_counter_collection.csv")):
    with open(f) as fh:
        rows = [r for r in csv.DictReader(fh) if "k_pair" in r["Kernel_Name"]]
    by = defaultdict(list)
    for r in rows:
        by[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, v in by.items():
        v.sort()
        acc[c] = [x for _, x in v]
print("%-34s %16s %16s   (dispatches)" % ("counter", "at rest", "developed"))
for c in sorted(acc):
    v = acc[c]
    a, b = v[8:48], v[-40:]
    print("%-34s %16.6g %16.6g   (%d)" % (c, sum(a) / max(len(a), 1), sum(b) / max(len(b), 1), len(v)))
