#!/bin/bash
# A/B of compile-time variants of render_kernel.hip on the same box, interleaved over two rounds:
#   usage: ab_flags.sh "<command>" "<EXTRA flags A>" "<EXTRA flags B>" ...      ("" = the build as committed)
# The library is rebuilt (render_kernel.o only) for every variant and restored to the committed build at the end.
CMD=$1; shift
cd /root/repo/yahr_amd/csrc
for round in 1 2; do
  for f in "$@"; do
    rm -f build/render_kernel.o; make EXTRA="$f" >/dev/null 2>&1 || { echo "BUILD FAIL [$f]"; continue; }
    echo "== [$f] (round $round)"
    (cd /root/repo && timeout -k 10 300 bash -c "$CMD" 2>&1 | grep -v amdgpu.ids)
  done
done
rm -f build/render_kernel.o; make >/dev/null 2>&1
