#!/bin/bash
# A/B two versions of render_kernel.hip on the same box: usage ab_sources.sh fileA fileB [perf_ab args...]
A=$1; B=$2; shift 2
cd /root/repo/yahr_amd/csrc
cp render_kernel.hip /tmp/rk_orig.hip
for round in 1 2; do
  for f in $A $B; do
    cp $f render_kernel.hip; rm -f build/render_kernel.o; make >/dev/null 2>&1 || { echo BUILD FAIL $f; continue; }
    echo "== $f (round $round)"
    (timeout -k 10 150 python /root/repo/tools/perf_ab.py "$@" 2>&1 | grep -E "min|WARNING")
  done
done
cp /tmp/rk_orig.hip render_kernel.hip; rm -f build/render_kernel.o; make >/dev/null 2>&1
