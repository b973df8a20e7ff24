#!/bin/bash
# A/B compile-time variants on a bench.py workload.  usage: ab_flags_bench.sh "<bench args>" "<EXTRA 1>" "<EXTRA 2>" ...
ARGS=$1; shift
cd /root/repo/yahr_amd/csrc
for round in 1 2; do
  for extra in "$@"; do
    rm -f build/render_kernel.o; make EXTRA="$extra" >/dev/null 2>&1 || { echo BUILD FAIL "$extra"; continue; }
    (timeout -k 10 300 python3 /root/repo/bench.py $ARGS --no-cpu-baseline --min-seconds 1 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('EXTRA=[$extra] round $round:', j['value'], 'Mrays/s', j['ms_per_step'], 'ms/frame', j['image_checksum'])")
  done
done
rm -f build/render_kernel.o; make >/dev/null 2>&1
