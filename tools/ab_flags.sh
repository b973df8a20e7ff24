#!/bin/bash
# A/B compile-time variants of render_kernel.hip on the same box.
# usage: ab_flags.sh "<perf_ab arguments>" "<EXTRA flags of variant 1>" "<EXTRA flags of variant 2>" ...   ("" = baseline)
ARGS=$1; shift
cd /root/repo/yahr_amd/csrc
for round in 1 2; do
  for extra in "$@"; do
    rm -f build/render_kernel.o; make EXTRA="$extra" >/dev/null 2>&1 || { echo BUILD FAIL "$extra"; continue; }
    echo "== EXTRA='$extra' (round $round)"
    (timeout -k 10 200 python /root/repo/tools/perf_ab.py $ARGS 2>&1 | grep -E "min|WARNING")
  done
done
rm -f build/render_kernel.o; make >/dev/null 2>&1
