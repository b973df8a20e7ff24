"""Summarise rocprofv3 --pmc CSVs: mean counter value per dispatch of the timed k_render<.., false> kernel."""
import collections, csv, glob, sys
root = sys.argv[1]
agg = collections.defaultdict(list)
for f in sorted(glob.glob(root + "/*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "k_render" in r["Kernel_Name"] and "false" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    print(f"{k:32s} n={len(v):3d} mean={sum(v) / len(v):18.1f}")
