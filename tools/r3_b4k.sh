#!/bin/bash
cd /root/repo
for w in auto stream hybrid_stack; do
  timeout -k 10 300 python3 bench.py --workload bunny4k --traverse $w --steps 3 --warmup 1 --min-seconds 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', j['value'], 'Mrays/s', j['ms_per_step'], 'ms/frame')"
done
