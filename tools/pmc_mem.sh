#!/bin/bash
# vector-memory-path PMC passes (texture addresser, L1, address translation).  usage: pmc_mem.sh <tag> <variant> [scene]
TAG=$1; VAR=$2; SCENE=${3:-tris}
export TMPDIR=/tmp; cd /tmp
for pass in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
            "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_SERIALIZATION_STALL_sum" \
            "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TA_FLAT_READ_WAVEFRONTS_sum" \
            "GRBM_GUI_ACTIVE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
  n=$(echo $pass | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d /root/repo/gpurun_out/pmc_${TAG}/$n -- python3 /root/repo/tools/perf_ab.py --scene $SCENE --rounds 3 $VAR > /root/repo/gpurun_out/pmc_${TAG}_$n.log 2>&1 || echo FAIL $n
done
python3 /root/repo/tools/pmc_summary.py /root/repo/gpurun_out/pmc_${TAG} > /root/repo/gpurun_out/pmc_${TAG}_summary.txt
cat /root/repo/gpurun_out/pmc_${TAG}_summary.txt
