#!/bin/bash
# VALU issue ceiling of the chip, with the clock it holds under that load: kernel time from --kernel-trace,
# GRBM_GUI_ACTIVE (summed over the 8 XCDs) from a separate --pmc pass.  usage: run_valu_rate.sh  -> gpurun_out/valu_rate/
OUT=/root/repo/gpurun_out/valu_rate; mkdir -p $OUT
cd /root/repo/tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize valu_rate.hip -o valu_rate 2>/dev/null || exit 1
export TMPDIR=/tmp; cd /tmp
/root/repo/tools/ubench/valu_rate > $OUT/plain.txt 2>&1
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- /root/repo/tools/ubench/valu_rate > $OUT/trace.log 2>&1
timeout -k 10 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU --output-format csv -d $OUT/pmc -- /root/repo/tools/ubench/valu_rate > $OUT/pmc.log 2>&1
python3 - <<'PY'
import csv, glob
out = "/root/repo/gpurun_out/valu_rate"
tr = [r for f in glob.glob(out + "/trace/**/*_kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
pm = [r for f in glob.glob(out + "/pmc/**/*_counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
tr = sorted(tr, key=lambda r: int(r["Start_Timestamp"]))
by = {}
for r in pm:
    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
print("dispatch grid_threads kernel_ms GRBM_GUI_ACTIVE/8 clock_GHz SQ_INSTS_VALU G_instr_per_s cycles_per_instr_per_SIMD")
for i, r in enumerate(tr):
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    c = by.get(sorted(by)[i], {}) if i < len(by) else {}
    g = c.get("GRBM_GUI_ACTIVE", 0) / 8
    v = c.get("SQ_INSTS_VALU", 0)
    grid = r.get("Grid_Size") or r.get("Grid_Size_X")
    if ms > 0.5 and v:
        print(i, grid, round(ms, 3), int(g), round(g / (ms * 1e6), 3), int(v), round(v / ms / 1e6, 1), round(g * 1024 / v, 3))
PY
