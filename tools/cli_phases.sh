#!/bin/bash
# phases of one command-line render of the 1 M-triangle scene (YAHR_TIMING=1).  usage (GPU box): tools/cli_phases.sh
cd "$(dirname "$0")/.."
python3 - <<PY
import sys; sys.path[:0]=["."]
from yahr_amd import scenes
open("/tmp/tris1m.yahrr","w").write(scenes.random_tris(1_000_000, 1920, 1080, depth=2, bvh_depth=20).to_text())
open("/tmp/bunny.yahrr","w").write(scenes.pseudo_bunny().to_text())
PY
for f in tris1m bunny; do for i in 1 2 3; do echo "--- $f run $i"; YAHR_TIMING=1 bin/yahr /tmp/$f.yahrr /tmp/o.ppm -p hip 2>&1 | grep timing; done; done
