#!/bin/bash
# bench line of every workload -> gpurun_out/bench_<workload>.json (args: workloads)
cd /root/repo
for w in ${*:-bunny spheres bunny4k tris1m}; do
  steps=40; [ $w = bunny4k ] && steps=4; [ $w = tris1m ] && steps=8
  timeout -k 10 400 python3 bench.py --workload $w --steps $steps --warmup 4 --min-seconds 1 --no-cpu-baseline > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err || echo FAIL $w
  python3 -c "
import json
j = json.load(open('gpurun_out/bench_$w.json'))
print('$w', j['value'], 'Mrays/s', j['ms_per_step'], 'ms/frame', 'single', j['single_frame_ms'])"
done
