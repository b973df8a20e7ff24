#!/bin/bash
# rebuild per variant (comma-separated -D flags) and run the perf_ab command lines given in SWEEP_CMDS (one per line)
cd /root/repo/yahr_amd/csrc
for v in "$@"; do
  flags=$(echo $v | tr ',' ' ')
  rm -f build/render_kernel.o
  make EXTRA="$flags" >/dev/null 2>&1 || { echo "BUILD FAIL $v"; continue; }
  echo "== $v"
  while IFS= read -r line; do
    [ -z "$line" ] && continue
    (timeout -k 10 200 python /root/repo/tools/perf_ab.py $line 2>&1 | grep -E "min|WARNING" | sed "s|^|  [$line] |")
  done <<< "$SWEEP_CMDS"
done
rm -f build/render_kernel.o; make >/dev/null 2>&1
