"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel (all dispatches); --sum: totals."""
import csv, glob, re, sys, collections
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*", "", n).replace("void ", "")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
SUM = '--sum' in sys.argv
for path in [a for a in sys.argv[1:] if a != '--sum']:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in agg.values() for c in k})
for k, cs in sorted(agg.items()):
    print(k[:70])
    print("    " + "  ".join(f"{c.replace('SQ_', '')}={sum(cs[c]) / (1 if SUM else len(cs[c])):.4g}" for c in names if c in cs))
