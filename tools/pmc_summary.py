"""Summarise rocprofv3 --pmc CSVs: mean counter value per dispatch, per (kernel, grid size).  By default only the
uncounted render kernels (k_render<.., false, ..> and the k_stream_* and k_sort_* kernels of the wavefront form) are listed; of
k_render only the largest grid (the F-frame launches of bench.py) unless --all-grids.
usage: pmc_summary.py <dir with one sub-directory per pass> [--all-grids]"""
import collections, csv, glob, re, sys
root = sys.argv[1]
all_grids = "--all-grids" in sys.argv
rows = collections.defaultdict(list)          # (kernel, counter) -> [(grid, value)]


def counted(name):
    """is this the collect_stats instantiation?  k_render<MODE, STATS, WPB, DEEP>, k_stream_primary<STATS>, k_stream_level<MODE, STATS>"""
    m = re.search(r"(k_\w+)<([^>]*)>", name.split("(")[0])
    if not m:
        return False
    args = [a.strip() for a in m.group(2).split(",")]
    pos = {"k_render": 1, "k_stream_primary": 0, "k_stream_level": 1}.get(m.group(1))
    return pos is not None and pos < len(args) and args[pos] == "true"

for f in sorted(glob.glob(root + "/**/*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if not (("k_render" in name or "k_stream" in name or "k_sort" in name) and not counted(name)):
            continue
        short = re.sub(r"^void ", "", name.split("(")[0])
        rows[(short, r["Counter_Name"])].append((int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"])))
for (k, ctr) in sorted(rows):
    grids = sorted(set(g for g, _ in rows[(k, ctr)]))
    if not all_grids:
        grids = grids[-1:]
    for g in grids:
        v = [x for gg, x in rows[(k, ctr)] if gg == g]
        print(f"{k:34s} {ctr:30s} grid={g:10d} n={len(v):4d} mean={sum(v) / len(v):18.1f}")
