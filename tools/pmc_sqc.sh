#!/bin/bash
# scalar-cache / instruction-cache PMC passes.  usage: pmc_sqc.sh <tag> <variant>
TAG=$1; VAR=$2
export TMPDIR=/tmp; cd /tmp
for pass in "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_DCACHE_BUSY_CYCLES SQC_DCACHE_INPUT_VALID_READYB SQC_TC_DATA_READ_REQ SQC_TC_STALL" \
            "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_TC_INST_REQ SQC_TC_REQ SQC_ICACHE_MISSES_DUPLICATE" \
            "SQ_INST_LEVEL_SMEM SQ_IFETCH_LEVEL SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM" \
            "SQC_DCACHE_REQ_READ_1 SQC_DCACHE_REQ_READ_2 SQC_DCACHE_REQ_READ_4 SQC_DCACHE_REQ_READ_8 SQC_DCACHE_REQ_READ_16 SQ_BUSY_CYCLES SQ_CYCLES SQ_WAVES"; do
  n=$(echo $pass | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d /root/repo/gpurun_out/pmc_${TAG}/$n -- python3 /root/repo/tools/perf_ab.py --rounds 3 $VAR > /root/repo/gpurun_out/pmc_${TAG}_$n.log 2>&1 || echo FAIL $n
done
python3 /root/repo/tools/pmc_summary.py /root/repo/gpurun_out/pmc_${TAG} > /root/repo/gpurun_out/pmc_${TAG}_summary.txt
cat /root/repo/gpurun_out/pmc_${TAG}_summary.txt
