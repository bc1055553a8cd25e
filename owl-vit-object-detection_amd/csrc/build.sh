#!/bin/bash
# Build libowlhip.so (gfx950 only) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
ARCH=${OWL_ARCH:-gfx950}
# -fvisibility=hidden: only the OWL_API entry points (common.h) reach the dynamic symbol table
FLAGS="--offload-arch=${ARCH} -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-value"
# OWL_TUNING=1: also export the process-global tuning switches of include/owl_hip_tuning.h (tools/ only; never the shipped build)
# Objects of a tuning build and of the shipped build live in different directories: staleness is judged by file times alone, so one shared
# directory would let a default build re-use -DOWL_TUNING objects (and ship the process-global setters), or the reverse.
BUILD=build
if [ "${OWL_TUNING:-0}" = "1" ]; then FLAGS="$FLAGS -DOWL_TUNING"; BUILD=build_tuning; fi
mkdir -p $BUILD
objs=""
pids=""
# csrc/ holds the shipped kernels only; the whole-file experiments (free-running / four-phase / four-wave GEMMs, one-wave-per-SIMD attention forward)
# live in tools/experiments/csrc/ and are compiled into a TUNING build alone (one command: tools/experiments/build.sh)
SRCS=$(ls *.hip)
if [ "${OWL_TUNING:-0}" = "1" ]; then SRCS="$SRCS $(ls ../../tools/experiments/csrc/*.hip)"; FLAGS="$FLAGS -I$(pwd)"; fi
for f in $SRCS; do
  b=$(basename "$f")
  o=$BUILD/${b%.hip}.o
  stale=0
  for h in *.h; do [ "$h" -nt "$o" ] && stale=1; done
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ $stale = 1 ]; then
    extra=""
    case "$b" in
      loss.hip|postprocess.hip) extra="-ffp-contract=off";;
      # packed f32 VALU (v_pk_mul/add_f32) beside MFMAs costs more than the two scalar instructions it replaces
      # (MI355X_MICROARCH.md; measured -1.7 % on the backward pair): no SLP packing in the attention kernels
      attention_bwd.hip|attention_fwd.hip) extra="-fno-slp-vectorize";;
    esac
    rm -f "$o"                                   # a failed compile must not leave the previous object for the link step
    hipcc $FLAGS $extra -c "$f" -o "$o" &
    pids="$pids $!"
  fi
  objs="$objs $o"
done
fail=0
for pid in $pids; do wait $pid || fail=1; done
if [ $fail = 1 ]; then echo "build.sh: a HIP source failed to compile" >&2; exit 1; fi
g++ -O2 -fPIC -fvisibility=hidden -std=c++17 -ffp-contract=off -c runtime.cpp -o $BUILD/runtime.o
hipcc --offload-arch=${ARCH} -shared -fPIC -Wl,--version-script=libowlhip.map -o ../libowlhip.so $objs $BUILD/runtime.o -ldl
echo "built $(cd .. && pwd)/libowlhip.so"
