#!/bin/bash
# Build the PROFILING library of another commit (default HEAD) into sxxcvr_amd/lib/prev/libsxfir_prof.so, for
# before / after timings of a kernel change on the same GPU box (CPU only, hipcc cross-compiles):
#     bash tools/prev_lib.sh [commit]
#     gpurun -- 'bash tools/gpu_steps.sh <tag> kbprev:8'     # kbench alternating between the two libraries
# tools/kbench.py picks it up through SXFIR_PROF_LIB (honoured by the profiling loader only).
set -eu
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/sxprev.XXXXXX)
git -C "$ROOT" archive "$REV" sxxcvr_amd/csrc include | tar -x -C "$TMP"
mkdir -p "$ROOT/sxxcvr_amd/lib/prev"
/opt/rocm/bin/hipcc -std=c++17 -fPIC -shared -Wall -Wno-unused-function -DSXFIR_PROFILING -I"$TMP/include" --offload-arch=gfx950 -O3 \
    "$TMP/sxxcvr_amd/csrc/sxfir.hip" -o "$ROOT/sxxcvr_amd/lib/prev/libsxfir_prof.so" -ldl
git -C "$ROOT" rev-parse --short "$REV" > "$ROOT/sxxcvr_amd/lib/prev/REV"
rm -rf "$TMP"
echo "built sxxcvr_amd/lib/prev/libsxfir_prof.so from $(cat "$ROOT/sxxcvr_amd/lib/prev/REV")"
