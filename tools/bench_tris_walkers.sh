#!/bin/bash
# tris1m bench line per walker of the stream form's secondary rays -> gpurun_out/tris_walker_<walker>.json
cd /root/repo
for w in ${*:-stream stack hybrid_stack ordered}; do
  timeout -k 10 300 python3 bench.py --workload tris1m --traverse $w --steps 8 --warmup 2 --min-seconds 1 --no-cpu-baseline > gpurun_out/tris_walker_$w.json 2> gpurun_out/tris_walker_$w.err || echo FAIL $w
  python3 - <<PY
import json
try:
    j = json.load(open("gpurun_out/tris_walker_$w.json"))
    print("$w", j["value"], "Mrays/s", j["ms_per_step"], "ms/frame", j["image_checksum"])
except Exception as e:
    print("$w", "no line", e)
PY
done
