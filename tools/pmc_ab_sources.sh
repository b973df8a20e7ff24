#!/bin/bash
# instruction counters of one perf_ab command under two versions of render_kernel.hip (first two PMC passes only)
# usage: pmc_ab_sources.sh fileA fileB <perf_ab arguments...>
A=$1; B=$2; shift 2
cd /root/repo/yahr_amd/csrc
cp render_kernel.hip /tmp/rk_orig.hip
export TMPDIR=/tmp
for f in $A $B; do
  cp $f render_kernel.hip; rm -f build/render_kernel.o; make >/dev/null 2>&1 || { echo BUILD FAIL $f; continue; }
  echo "== $f"
  OUT=/tmp/pmcab_$$; rm -rf $OUT; mkdir -p $OUT
  for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
              "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU"; do
    n=$(echo $pass | cut -d" " -f1)
    (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc/$n -- python3 /root/repo/tools/perf_ab.py --rounds 2 "$@" > $OUT/pmc_$n.log 2>&1) || echo FAIL $n
  done
  python3 /root/repo/tools/pmc_summary.py $OUT/pmc --all-grids
  rm -rf $OUT
done
cp /tmp/rk_orig.hip render_kernel.hip; rm -f build/render_kernel.o; make >/dev/null 2>&1
