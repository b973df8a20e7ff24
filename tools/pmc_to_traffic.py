"""profiles/traffic_latest.json from a pmc_summary.txt of tools/profile_bench.sh: per-FRAME counter figures of the
timed F-frame launches (instruction counts are deterministic per frame; the mean over the F cameras of bench.py's
orbit is what is stored), which bench.py scales by its frames per launch.
usage: pmc_to_traffic.py <pmc_summary.txt> <workload> <frames per launch> <source text> > profiles/traffic_latest.json"""
import json, sys
vals = {}
import re
for line in open(sys.argv[1]):
    m = re.match(r"^(k_render.*?)\s{2,}(\S+)\s+grid=.*mean=\s*([0-9.eE+-]+)", line)   # rows: kernel  counter  grid=.. n=.. mean=..
    if m:
        vals[m.group(2)] = float(m.group(3))
workload, F, source = sys.argv[2], int(sys.argv[3]), sys.argv[4]
fetch_kb, write_kb = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
out = {
    "workload": workload, "frames_per_launch_profiled": F, "source": source,
    "valu_insts_per_frame": vals["SQ_INSTS_VALU"] / F,
    "salu_insts_per_frame": vals["SQ_INSTS_SALU"] / F,
    "smem_insts_per_frame": vals["SQ_INSTS_SMEM"] / F,
    "branch_insts_per_frame": vals.get("SQ_INSTS_BRANCH", 0.0) / F,
    "FETCH_SIZE_KB_raw_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
    "correction": "gfx950: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM section: the counter tallies 128-B requests at 64 B; "
                  "calibrated there for 16 B/lane streams, uncalibrated for this kernel's scalar record fetches, so the "
                  "doubled figure is an upper bound); WRITE_SIZE taken as is",
    "hbm_bytes_per_frame": (2.0 * fetch_kb + write_kb) * 1024.0 / F,
    "valu_issue_peak_measured_ginst_s": 1000.0,
    "valu_issue_peak_measured_source": "tools/ubench/valu_rate.hip on MI355X (round 1): 982-1015 G wave64 f32 instr/s chip-wide at "
                                       ">= 2 waves/SIMD (2.45 clocks per instruction per SIMD)",
}
json.dump(out, sys.stdout, indent=1)
print()
