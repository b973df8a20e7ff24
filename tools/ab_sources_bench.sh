#!/bin/bash
# A/B two versions of render_kernel.hip with bench.py on the same box: usage ab_sources_bench.sh "<bench args>" fileA fileB ...
ARGS=$1; shift
cd /root/repo/yahr_amd/csrc
cp render_kernel.hip /tmp/rk_orig.hip
for round in 1 2 3; do
  for f in "$@"; do
    cp $f render_kernel.hip; rm -f build/render_kernel.o; make >/dev/null 2>&1 || { echo BUILD FAIL $f; continue; }
    (timeout -k 10 300 python3 /root/repo/bench.py $ARGS --no-cpu-baseline --min-seconds 2 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$f round $round:', j['value'], 'Mrays/s', j['ms_per_step'], 'ms/frame single', j['single_frame_ms'], 'kernel', j['roofline']['kernel_ms_avg'])")
  done
done
cp /tmp/rk_orig.hip render_kernel.hip; rm -f build/render_kernel.o; make >/dev/null 2>&1
