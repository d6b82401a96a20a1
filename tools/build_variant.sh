#!/bin/bash
# build_variant.sh NAME [extra hipcc flags...]  -> build_ab/lib_NAME.so
# (FILE=pg_eval16.hip by default: that source recompiled with the extra flags, other objects
#  from the regular build)
set -e
cd "$(dirname "$0")/../posegen_amd/csrc"
name=$1; shift
src=${FILE:-pg_eval16.hip}
base=${src%.hip}
mkdir -p ../../build_ab
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $src -o ../../build_ab/${base}_$name.o
objs=""
for o in pg_api pg_repack pg_eval16 pg_eval16r pg_rayrec pg_eval32 pg_evalc pg_evalc2 pg_kernels pg_train pg_pack; do
  if [ "$o" = "$base" ]; then objs="$objs ../../build_ab/${base}_$name.o"; else objs="$objs ../_lib/obj/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_ab/lib_$name.so $objs
echo built build_ab/lib_$name.so
