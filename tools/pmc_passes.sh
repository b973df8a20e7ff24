#!/bin/bash
# rocprofv3 PMC passes for one perf_ab variant (separate passes, no trace domains).  usage: pmc_passes.sh <tag> <variant> [scene]
TAG=$1; VAR=$2; SCENE=${3:-bunny}
export TMPDIR=/tmp; cd /tmp
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
            "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SMEM" \
            "SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAVE32_INSTS SQ_INSTS_VALU_MFMA_I8 SQ_LEVEL_WAVES SQ_CYCLES" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  n=$(echo $pass | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d /root/repo/gpurun_out/pmc_${TAG}/$n -- python3 /root/repo/tools/perf_ab.py --scene $SCENE --rounds 3 $VAR > /root/repo/gpurun_out/pmc_${TAG}_$n.log 2>&1 || echo FAIL $n
done
python3 /root/repo/tools/pmc_summary.py /root/repo/gpurun_out/pmc_${TAG} > /root/repo/gpurun_out/pmc_${TAG}_summary.txt
cat /root/repo/gpurun_out/pmc_${TAG}_summary.txt
