#!/bin/bash
# GPU box: the library rebuilt with AddressSanitizer + UBSan on the HOST code only (device code untouched), and
# tools/sanitize/device_harness.cpp run over a small bunny and a triangle soup.  usage: tools/sanitize/run_device.sh
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd); SRC=$ROOT/yahr_amd/csrc; W=$(mktemp -d)
python3 - "$W" <<PY
import sys; sys.path[:0] = ["$ROOT"]
from yahr_amd import scenes
w = sys.argv[1]
open(w + "/bunny.yahrr", "w").write(scenes.pseudo_bunny(203, 117, n_lon=48, n_lat=25, depth=3).to_text())
open(w + "/soup.yahrr", "w").write(scenes.random_tris(160000, 160, 90, depth=6, bvh_depth=18).to_text())   # depth 6: level spill; 160 000: device BVH build, coherence probe of yahr_ctx_create
PY
SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer"
HF="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize $SAN -I$ROOT/include -I$SRC"
cd $W
/opt/rocm/bin/hipcc $HF -c $SRC/render_kernel.hip -o rk.o
/opt/rocm/bin/hipcc $HF -c $SRC/bvh_build.hip -o bb.o
for f in abi_device abi_multi; do /opt/rocm/bin/hipcc $HF -x hip -c $SRC/$f.cpp -o $f.o; done
for f in abi_host host_scene scene_parse; do /opt/rocm/bin/hipcc $HF -c $SRC/$f.cpp -o $f.o; done
/opt/rocm/bin/hipcc $HF -c $ROOT/tools/sanitize/device_harness.cpp -o harness.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 $SAN harness.o rk.o bb.o abi_device.o abi_multi.o abi_host.o host_scene.o scene_parse.o -lz -ldl -lpthread -o h
ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 ./h bunny.yahrr soup.yahrr
rm -rf "$W"
