#!/bin/bash
# AddressSanitizer + UBSan, then ThreadSanitizer, over the HOST side of the library (threaded .yahrr parser, threaded mesh
# expansion and BVH build, PNG writer in bands) -- g++ only, no GPU code involved.  usage: tools/sanitize/run.sh
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd); SRC=$ROOT/yahr_amd/csrc; W=$(mktemp -d)
python3 - "$W" <<PY
import sys; sys.path[:0] = ["$ROOT", "$ROOT/tests"]
import numpy as np
from yahr_amd import scenes
import test_loader as T
w = sys.argv[1]
open(w + "/mesh.yahrr", "w").write(scenes.random_tris(60000, 64, 36, depth=2, bvh_depth=20).to_text())
txt, _ = T._tricky_scene_text(np.random.default_rng(3), 4000)
open(w + "/tricky.yahrr", "w").write(txt)
open(w + "/bad.yahrr", "w").write(txt[: len(txt) // 2] + ") " + txt[len(txt) // 2:])
PY
export YAHR_PARSE_PARALLEL_MIN=1 YAHR_PARSE_THREADS=6 YAHR_PNG_THREADS=5 YAHR_BUILD_CHUNK_MIN=64
for san in address,undefined thread; do
  g++ -std=c++17 -O1 -g -fsanitize=$san -I$ROOT/include -I$SRC $ROOT/tools/sanitize/host_harness.cpp $SRC/scene_parse.cpp $SRC/host_scene.cpp $SRC/abi_host.cpp -lz -lpthread -o $W/h
  echo "== -fsanitize=$san"; (cd $W && ./h mesh.yahrr tricky.yahrr bad.yahrr)
done
rm -rf "$W"
