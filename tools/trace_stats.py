"""Per-kernel statistics from a rocprofv3 --kernel-trace CSV, split by grid size: bench.py launches the same
k_render instantiation with F frames (the timed launches), with the K mod F tail and with a single frame, and
`--stats` lumps them together.  usage: trace_stats.py <dir with *_kernel_trace.csv> > kernel_stats_by_grid.csv"""
import collections, csv, glob, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        grid = r.get("Grid_Size") or str(int(r.get("Grid_Size_X", 0) or 0) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1))
        wg = r.get("Workgroup_Size") or str(int(r.get("Workgroup_Size_X", 0) or 0) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1))
        agg[(name, int(grid), int(wg), r.get("VGPR_Count", ""), r.get("SGPR_Count", ""),
             r.get("Scratch_Size", r.get("Private_Segment_Size", "")))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "grid_threads", "workgroup", "vgpr", "sgpr", "scratch", "calls", "avg_ns", "min_ns", "max_ns", "total_ns"])
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    w.writerow(list(k) + [len(v), round(sum(v) / len(v), 1), min(v), max(v), sum(v)])
