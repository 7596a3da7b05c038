"""Sum rocprofv3 --pmc counter_collection.csv per kernel (short names)."""
import csv, sys, collections, glob, re
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(attn_\w+?)(<[^>]*>)?\(", r["Kernel_Name"])
        k = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:30]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    if "attn" in k:
        print(k, {n: f"{x:.3g}" for n, x in sorted(v.items())})
