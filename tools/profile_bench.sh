#!/bin/bash
# rocprofv3 evidence for a bench line: kernel-trace statistics, then PMC passes (separate runs, counters only),
# all of the SAME command.  usage: tools/profile_bench.sh <tag> [bench.py arguments]   -> gpurun_out/prof_<tag>/
TAG=${1:-run}; shift
ARGS=${*:---steps 200 --warmup 24}
OUT=/root/repo/gpurun_out/prof_${TAG}
mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
CMD="python3 /root/repo/bench.py $ARGS --no-cpu-baseline --repeats 2"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || echo FAIL trace
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python3 /root/repo/tools/trace_stats.py $OUT/trace > $OUT/kernel_stats_by_grid.csv
grep '^{"metric"' $OUT/trace.log | tail -1 > $OUT/bench_line_under_rocprof.json
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
            "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  n=$(echo $pass | cut -d" " -f1)
  timeout -k 10 400 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc/$n -- $CMD > $OUT/pmc_$n.log 2>&1 || echo FAIL $n
done
python3 /root/repo/tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt
rm -rf $OUT/pmc $OUT/trace                     # keep the summaries, not the raw per-dispatch CSVs / databases
cat $OUT/pmc_summary.txt; head -5 $OUT/kernel_stats_by_grid.csv
