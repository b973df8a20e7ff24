#!/bin/bash
cd /root/repo
for lds in 0 4096 5120 5632 8192; do
  echo "== stream (old walker) with $lds B of LDS per wave"
  YAHR_DEBUG_LDS=$lds tools/r3_tris.sh stream
done
