#!/bin/bash
# round-end evidence on the GPU box: PMC + kernel-trace profiles of the four workloads (-> gpurun_out/prof_<tag>/), then the
# traffic_<workload>.json files bench.py reads (with the kernel source hash), then the bench lines.  usage: final_profiles.sh <round tag>
R=${1:-r04}
cd /root/repo
tools/profile_bench.sh ${R}_bunny --steps 200 --warmup 24 > gpurun_out/prof_${R}_bunny.log 2>&1
tools/profile_bench.sh ${R}_spheres --workload spheres --steps 200 --warmup 24 > gpurun_out/prof_${R}_spheres.log 2>&1
# (config 4 is a deep scene: its level spill would cut an 8-frame batch into several launches; 2 frames per launch in ONE launch)
YAHR_SPILL_BUDGET_KB=120000000 tools/profile_bench.sh ${R}_bunny4k --workload bunny4k --steps 2 --warmup 2 --frames-per-launch 2 > gpurun_out/prof_${R}_bunny4k.log 2>&1
tools/profile_bench.sh ${R}_tris1m --workload tris1m --steps 8 --warmup 8 > gpurun_out/prof_${R}_tris1m.log 2>&1
SRC="rocprofv3 --pmc passes (one counter group per pass, tools/profile_bench.sh) of bench.py --workload %s --no-cpu-baseline --repeats 2 on one MI355X (profiles/${R}_%s_pmc_summary.txt)"
python3 tools/pmc_to_traffic.py gpurun_out/prof_${R}_bunny/pmc_summary.txt bunny 20 "$(printf "$SRC" bunny bunny)" > gpurun_out/traffic_latest.json
python3 tools/pmc_to_traffic.py gpurun_out/prof_${R}_spheres/pmc_summary.txt spheres 20 "$(printf "$SRC" spheres spheres)" > gpurun_out/traffic_spheres.json
python3 tools/pmc_to_traffic.py gpurun_out/prof_${R}_bunny4k/pmc_summary.txt bunny4k 2 "$(printf "$SRC" bunny4k bunny4k)" > gpurun_out/traffic_bunny4k.json
python3 tools/pmc_to_traffic.py gpurun_out/prof_${R}_tris1m/pmc_summary.txt tris1m 8 "$(printf "$SRC" tris1m tris1m)" 2 > gpurun_out/traffic_tris1m.json
for w in bunny spheres bunny4k tris1m; do
  cp gpurun_out/prof_${R}_$w/pmc_summary.txt gpurun_out/${R}_${w}_pmc_summary.txt
  cp gpurun_out/prof_${R}_$w/kernel_stats_by_grid.csv gpurun_out/${R}_${w}_kernel_stats_by_grid.csv
  cp gpurun_out/prof_${R}_$w/bench_line_under_rocprof.json gpurun_out/${R}_${w}_bench_under_rocprof.json
done
# bench lines with the fresh profiles in place (bench.py reads profiles/traffic_*.json)
cp gpurun_out/traffic_latest.json gpurun_out/traffic_spheres.json gpurun_out/traffic_bunny4k.json gpurun_out/traffic_tris1m.json profiles/
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
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('single_frame_ms'), r['frac'], r.get('profile_matches_source'), (r.get('hbm') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'))
PY
