#!/bin/bash
# kernel trace + PMC passes of one tools/perf_ab.py command.  usage: pmc_ab.sh <tag> <perf_ab arguments...>
TAG=$1; shift
OUT=/root/repo/gpurun_out/pmcab_${TAG}
mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
CMD="python3 /root/repo/tools/perf_ab.py --rounds 2 $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || echo FAIL trace
python3 /root/repo/tools/trace_stats.py $OUT/trace > $OUT/kernel_stats_by_grid.csv
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
            "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  n=$(echo $pass | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc/$n -- $CMD > $OUT/pmc_$n.log 2>&1 || echo FAIL $n
done
python3 /root/repo/tools/pmc_summary.py $OUT/pmc --all-grids > $OUT/pmc_summary.txt
rm -rf $OUT/pmc $OUT/trace
cat $OUT/kernel_stats_by_grid.csv | cut -c1-200; cat $OUT/pmc_summary.txt
