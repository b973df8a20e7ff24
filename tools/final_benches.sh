#!/bin/bash
# the five bench lines of a round with the committed profiles in place (no re-profiling).  usage (GPU box): tools/final_benches.sh <round tag>
R=${1:-r03}
export TMPDIR=/tmp; cd /tmp
timeout -k 10 400 python3 /root/repo/bench.py > /root/repo/gpurun_out/${R}_final_bench.log 2>&1 && grep '^{"metric"' /root/repo/gpurun_out/${R}_final_bench.log | tail -1 > /root/repo/gpurun_out/${R}_final_bench.json
timeout -k 10 400 python3 /root/repo/bench.py --steps 20 --warmup 5 > /root/repo/gpurun_out/${R}_final_bench_driver.log 2>&1 && grep '^{"metric"' /root/repo/gpurun_out/${R}_final_bench_driver.log | tail -1 > /root/repo/gpurun_out/${R}_final_bench_driver_args.json
for w in spheres bunny4k tris1m; do
  timeout -k 10 400 python3 /root/repo/bench.py --workload $w --steps 8 --warmup 8 > /root/repo/gpurun_out/${R}_final_bench_$w.log 2>&1 && grep '^{"metric"' /root/repo/gpurun_out/${R}_final_bench_$w.log | tail -1 > /root/repo/gpurun_out/${R}_final_bench_$w.json
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob('/root/repo/gpurun_out/${R}_final_bench*.json')):
    d = json.load(open(f)); r = d['roofline']
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('single_frame_ms'), r['frac'], r.get('profile_matches_source'), (r.get('hbm') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'), (d.get('parity') or {}).get('pixels_outside_tolerance'))
PY
