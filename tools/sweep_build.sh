#!/bin/bash
# rebuild render_kernel.o with -D flags per variant on the GPU box and time it (packet + lane walkers)
cd /root/repo/yahr_amd/csrc
for v in "$@"; do
  flags=$(echo $v | tr ',' ' ')
  rm -f build/render_kernel.o
  make EXTRA="$flags" >/dev/null 2>&1 || { echo "BUILD FAIL $v"; continue; }
  echo "== $v"
  (cd /root/repo && timeout -k 10 120 python tools/perf_ab.py --rounds 9 ${SWEEP_VARIANTS:-packet:packet:0} 2>&1 | grep -E "min|WARNING")
done
rm -f build/render_kernel.o; make >/dev/null 2>&1
