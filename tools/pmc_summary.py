"""Per-kernel means of the counters that tools/pmc.sh collected: python tools/pmc_summary.py <outdir> [name-regex]"""
import csv, glob, re, sys, collections

out = sys.argv[1]
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if pat.search(name):
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, ctrs in sorted(acc.items()):
    print(name)
    for c, v in sorted(ctrs.items()):
        print("  %-32s n=%-5d mean %.6g  min %.6g  max %.6g" % (c, len(v), sum(v) / len(v), min(v), max(v)))
