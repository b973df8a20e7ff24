#!/bin/bash
# sweep lane-walker build parameters: rebuild per variant, time the incoherent scenes (lane + adaptive walkers)
cd /root/repo/yahr_amd/csrc
for v in "$@"; do
  flags=$(echo $v | tr ',' ' ')
  rm -f build/render_kernel.o
  make EXTRA="$flags" >/dev/null 2>&1 || { echo "BUILD FAIL $v"; continue; }
  echo "== $v"
  for sc in ${SWEEP_SCENES:-tris tris200k}; do
    (timeout -k 10 120 python /root/repo/tools/perf_ab.py --scene $sc --rounds 4 lane:lane:0 adaptive:hybrid:0 2>&1 | grep -E "min|WARNING" | sed "s/^/  $sc /")
  done
done
rm -f build/render_kernel.o; make >/dev/null 2>&1
