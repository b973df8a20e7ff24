#!/bin/bash
# the four bench lines kept under profiles/ (run on the GPU box): default arguments, the driver's arguments, other workloads
export TMPDIR=/tmp; cd /tmp
OUT=/root/repo/gpurun_out/final; mkdir -p $OUT
timeout -k 10 400 python /root/repo/bench.py > $OUT/bench.log 2>&1 && grep '^{"metric"' $OUT/bench.log | tail -1 > $OUT/r02_final_bench.json
timeout -k 10 400 python /root/repo/bench.py --steps 20 --warmup 5 > $OUT/bench_driver.log 2>&1 && grep '^{"metric"' $OUT/bench_driver.log | tail -1 > $OUT/r02_final_bench_driver_args.json
for w in spheres bunny4k tris1m; do
  timeout -k 10 400 python /root/repo/bench.py --workload $w --steps 8 --warmup 8 > $OUT/bench_$w.log 2>&1 && grep '^{"metric"' $OUT/bench_$w.log | tail -1 > $OUT/r02_final_bench_$w.json
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('/root/repo/gpurun_out/final/r02_final_bench*.json')):
    d = json.load(open(f)); r = d['roofline']
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('single_frame_ms'), r['frac'], r.get('frac_of_measured_issue_peak'), (r.get('hbm') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'))
PY
