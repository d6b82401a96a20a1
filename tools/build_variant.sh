#!/bin/bash
# build_variant.sh NAME [extra hipcc flags...]  -> build_ab/lib_NAME.so
# (pg_eval16.hip recompiled with the extra flags, other objects from the regular build)
set -e
cd "$(dirname "$0")/../posegen_amd/csrc"
name=$1; shift
mkdir -p ../../build_ab
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FLAGS "$@" -c pg_eval16.hip -o ../../build_ab/eval16_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_ab/lib_$name.so ../../build_ab/eval16_$name.o \
   ../_lib/obj/pg_api.o ../_lib/obj/pg_eval32.o ../_lib/obj/pg_kernels.o ../_lib/obj/pg_pack.o
echo built build_ab/lib_$name.so
