#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...]  ->  gpurun_abl/NAME/libsurfacenet_hip.so
# A/B builds of the HIP library with extra compile-time switches (SN_TIMING, SN_MX_S_ACT, ...). gpurun_abl/ is git-ignored but travels to the
# GPU box; select a variant at run time with SURFACENET_HIP_LIB=gpurun_abl/NAME/libsurfacenet_hip.so (surfacenet_amd/_lib.py).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; shift
OUT="$ROOT/gpurun_abl/$NAME"
mkdir -p "$OUT"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -DSN_DEBUG_HOOKS -Wall -Wno-unused-function $*"
cd "$ROOT/surfacenet_amd/csrc"
pids=()
for f in sn_api sn_post sn_simil; do
  /opt/rocm/bin/hipcc $FLAGS -c -o "$OUT/$f.o" $f.hip &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -Wl,--version-script=exports.map -o "$OUT/libsurfacenet_hip.so" "$OUT"/sn_api.o "$OUT"/sn_post.o "$OUT"/sn_simil.o
rm -f "$OUT"/*.o
echo "$OUT/libsurfacenet_hip.so  [$*]"
