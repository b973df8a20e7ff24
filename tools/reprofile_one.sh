#!/bin/bash
# re-profile ONE workload and refresh its traffic json + bench line (see final_profiles.sh).  usage: reprofile_one.sh <round> <workload>
R=$1; W=$2; cd /root/repo
case $W in
  bunny) tools/profile_bench.sh ${R}_bunny --steps 200 --warmup 24 > gpurun_out/prof_${R}_bunny.log 2>&1; F=20; D=; OUT=traffic_latest.json;;
  spheres) tools/profile_bench.sh ${R}_spheres --workload spheres --steps 200 --warmup 24 > gpurun_out/prof_${R}_spheres.log 2>&1; F=20; D=; OUT=traffic_spheres.json;;
  bunny4k) YAHR_SPILL_BUDGET_KB=120000000 tools/profile_bench.sh ${R}_bunny4k --workload bunny4k --steps 2 --warmup 2 --frames-per-launch 2 > gpurun_out/prof_${R}_bunny4k.log 2>&1; F=2; D=; OUT=traffic_bunny4k.json;;
  tris1m) tools/profile_bench.sh ${R}_tris1m --workload tris1m --steps 8 --warmup 8 > gpurun_out/prof_${R}_tris1m.log 2>&1; F=8; D=2; OUT=traffic_tris1m.json;;
esac
SRC="rocprofv3 --pmc passes (one counter group per pass, tools/profile_bench.sh) of bench.py --workload $W --no-cpu-baseline --repeats 2 on one MI355X (profiles/${R}_${W}_pmc_summary.txt)"
python3 tools/pmc_to_traffic.py gpurun_out/prof_${R}_$W/pmc_summary.txt $W $F "$SRC" $D > gpurun_out/$OUT
cp gpurun_out/prof_${R}_$W/pmc_summary.txt gpurun_out/${R}_${W}_pmc_summary.txt
cp gpurun_out/prof_${R}_$W/kernel_stats_by_grid.csv gpurun_out/${R}_${W}_kernel_stats_by_grid.csv
cp gpurun_out/prof_${R}_$W/bench_line_under_rocprof.json gpurun_out/${R}_${W}_bench_under_rocprof.json
cp gpurun_out/$OUT profiles/
export TMPDIR=/tmp; cd /tmp
timeout -k 10 400 python3 /root/repo/bench.py --workload $W --steps 8 --warmup 8 > /root/repo/gpurun_out/${R}_final_bench_$W.log 2>&1 && grep '^{"metric"' /root/repo/gpurun_out/${R}_final_bench_$W.log | tail -1 > /root/repo/gpurun_out/${R}_final_bench_$W.json
python3 -c "
import json; d=json.load(open('/root/repo/gpurun_out/${R}_final_bench_$W.json')); r=d['roofline']; print('$W', d['value'], d['ms_per_step'], r['frac'], r.get('profile_matches_source'), (r.get('hbm') or {}).get('frac'))"
