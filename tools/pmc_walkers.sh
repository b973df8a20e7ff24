#!/bin/bash
# PMC comparison of the stream form's level launches per secondary walker (1M triangles, 16 spp) -> gpurun_out/pmc_walker_<walker>.txt
export TMPDIR=/tmp; cd /tmp
for w in $*; do
  OUT=/root/repo/gpurun_out/pmc_walker_$w; mkdir -p $OUT
  CMD="python3 /root/repo/bench.py --workload tris1m --traverse $w --steps 2 --warmup 1 --repeats 1 --frames-per-launch 1 --no-cpu-baseline"
  for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
              "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $pass | cut -d" " -f1)
    timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc/$n -- $CMD > $OUT/pmc_$n.log 2>&1 || echo FAIL $w $n
  done
  python3 /root/repo/tools/pmc_summary.py $OUT/pmc > /root/repo/gpurun_out/pmc_walker_$w.txt
  rm -rf $OUT
  grep "k_stream_level" /root/repo/gpurun_out/pmc_walker_$w.txt
done
