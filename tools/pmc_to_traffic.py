"""Per-FRAME counter figures of a workload from a pmc_summary.txt of tools/profile_bench.sh, for bench.py's roofline:
instruction counts are deterministic per frame (the mean over the F cameras of bench.py's orbit is what is stored),
so bench.py scales them by its frames per launch.
  packet / lane / adaptive kernels: one k_render launch holds F frames      -> per frame = mean per launch / F
  wavefront form: per frame one k_stream_primary, `depth` x (k_sort_scan, k_sort_scatter, k_stream_level) and one
  k_stream_resolve launch                       -> per frame = primary + depth x (scan + scatter + level) + resolve
usage: pmc_to_traffic.py <pmc_summary.txt> <workload> <frames per launch> <source text> [stream depth] > profiles/traffic_<workload>.json"""
import collections, json, os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "yahr_amd"))
from srchash import kernel_source_hash        # (the module alone: no GPU library is loaded for this)
vals = collections.defaultdict(dict)          # kernel -> counter -> mean per dispatch (largest grid)
for line in open(sys.argv[1]):
    m = re.match(r"^(k_\S.*?)\s{2,}(\S+)\s+grid=\s*(\d+)\s+n=.*mean=\s*([0-9.eE+-]+)", line)
    if m:
        vals[m.group(1).split("<")[0]][m.group(2)] = float(m.group(4))
workload, F, source = sys.argv[2], int(sys.argv[3]), sys.argv[4]
depth = int(sys.argv[5]) if len(sys.argv) > 5 else 0


def per_frame(counter):
    if depth:                                   # wavefront form
        return (vals["k_stream_primary"].get(counter, 0.0) + vals["k_stream_resolve"].get(counter, 0.0)
                + vals["k_stream_resolve_lanes"].get(counter, 0.0)        # (one of the two resolve kernels runs, by spp)
                + depth * (vals["k_stream_level"].get(counter, 0.0) + vals["k_sort_scan"].get(counter, 0.0)
                           + vals["k_sort_scatter"].get(counter, 0.0)))
    return vals["k_render"].get(counter, 0.0) / F


fetch_kb, write_kb = per_frame("FETCH_SIZE"), per_frame("WRITE_SIZE")
out = {
    "workload": workload, "frames_per_launch_profiled": F, "source": source,
    # fingerprint of the kernel sources + compiler flags the profile was taken from: bench.py quotes a roofline fraction
    # only while it still matches (the profile must be converted in the same tree state it was measured in)
    "kernel_source_sha256": kernel_source_hash(),
    "kernels": ("k_stream_primary + %d x (k_sort_scan + k_sort_scatter + k_stream_level) + k_stream_resolve per frame" % depth) if depth else "k_render, %d frames per launch" % F,
    "valu_insts_per_frame": per_frame("SQ_INSTS_VALU"),
    "salu_insts_per_frame": per_frame("SQ_INSTS_SALU"),
    "smem_insts_per_frame": per_frame("SQ_INSTS_SMEM"),
    "vmem_rd_insts_per_frame": per_frame("SQ_INSTS_VMEM_RD"),
    "FETCH_SIZE_KB_raw_per_frame": fetch_kb, "WRITE_SIZE_KB_per_frame": write_kb,
    "correction": "gfx950: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM section: the counter tallies 128-B requests at 64 B; "
                  "calibrated there for 16 B/lane streams, uncalibrated for scalar record fetches and 32-byte gathers, so the "
                  "doubled figure is an upper bound); WRITE_SIZE taken as is",
    "hbm_bytes_per_frame": (2.0 * fetch_kb + write_kb) * 1024.0,
    "valu_issue_peak_measured_ginst_s": 1000.0,
    "valu_issue_peak_measured_source": "tools/ubench/run_valu_rate.sh on MI355X (profiles/r02_valu_rate.txt): 984-1019 G wave64 f32 instr/s "
                                       "chip-wide at 2-8 waves/SIMD; the chip holds 2.13-2.16 GHz under that load, i.e. 2.18-2.23 cycles per instruction per SIMD",
}
json.dump(out, sys.stdout, indent=1)
print()
