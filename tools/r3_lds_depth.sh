#!/bin/bash
# bunny4k (depth 4): all four levels of the radiance stack in LDS (6 KiB per wavefront) against three + one spilled
cd /root/repo/yahr_amd/csrc
for round in 1 2; do
for d in 4 3; do
  rm -f build/render_kernel.o build/abi_device.o; make EXTRA="-DYD_LDS_DEPTH_OVERRIDE=$d" >/dev/null 2>&1 || echo BUILD FAIL
  timeout -k 10 300 python3 /root/repo/bench.py --workload bunny4k --steps 3 --warmup 1 --min-seconds 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('all-LDS up to depth $d:', j['value'], 'Mrays/s', j['ms_per_step'], 'ms/frame', j['image_checksum'])"
done
done
rm -f build/render_kernel.o build/abi_device.o; make >/dev/null 2>&1
