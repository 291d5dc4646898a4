#!/bin/bash
# Variant builds of the library for profiles/r6_two_streams.md: only preprocess_raw.hip (the victim kernel) differs.
#   noslab : -DTRASE_RAW_NO_SLAB        the victim uses no LDS at all (f_rest rows read per lane from memory)
#   noslp  : -fno-slp-vectorize         the victim's SH / EWA arithmetic without v_pk_*_f32 (scalar fp32 VALU only)
#   prio   : -DTRASE_RAW_SETPRIO        s_setprio 3 at the top of the victim
# Output: trase_amd/lib/variants/libtrase_rast_<name>.so (git-ignored; travels with gpurun).  Run from the repo root after `make`.
set -e
cd "$(dirname "$0")/../../trase_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function -Wno-unused-value -Wno-unused-result"
mkdir -p build/variants ../lib/variants
OTHERS=$(ls build/*.o | grep -v preprocess_raw.o)
build() { $HIPCC $FLAGS $2 -c preprocess_raw.hip -o build/variants/preprocess_raw_$1.o && $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libtrase_rast_$1.so $OTHERS build/variants/preprocess_raw_$1.o; }
build noslab "-DTRASE_RAW_NO_SLAB" &
build noslp "-fno-slp-vectorize" &
build prio "-DTRASE_RAW_SETPRIO" &
wait
ls -la ../lib/variants
# allnoslp: EVERY translation unit without SLP vectorisation (no compiler-generated v_pk_*_f32 anywhere in the library)
mkdir -p build/allnoslp
for f in api preprocess preprocess_raw binning render render_fwd_mf render_bwd_gs render_bwd_hw selftest knn mlp smooth loss optim contrastive densify pairhead nnfm; do
  x=""; [ $f = pairhead ] && x="-ffp-contract=off"
  $HIPCC $FLAGS -fno-slp-vectorize $x -c $f.hip -o build/allnoslp/$f.o &
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libtrase_rast_allnoslp.so build/allnoslp/*.o
ls -la ../lib/variants
