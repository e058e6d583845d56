"""Summarise rocprofv3 --pmc CSVs: mean counter value per dispatch of each kernel."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/*/*/*_counter_collection.csv"):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-34s mean %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
