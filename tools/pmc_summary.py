"""Summarise rocprofv3 --pmc CSVs: mean counter value per dispatch of the timed render kernel, i.e. of the
k_render<.., false, ..> launches with the largest grid (the F-frame launches of bench.py; single-frame and tail
launches of the same kernel have smaller grids and are listed separately)."""
import collections, csv, glob, sys
root = sys.argv[1]
rows = collections.defaultdict(list)          # counter -> [(grid, value)]
for f in sorted(glob.glob(root + "/*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "k_render" in r["Kernel_Name"] and "false" in r["Kernel_Name"]:
            rows[r["Counter_Name"]].append((int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"])))
for k in sorted(rows):
    big = max(g for g, _ in rows[k])
    v = [x for g, x in rows[k] if g == big]
    print(f"{k:32s} grid={big:10d} n={len(v):4d} mean={sum(v) / len(v):18.1f}")
