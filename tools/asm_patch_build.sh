#!/bin/bash
# asm_patch_build.sh NAME SRC.hip PATCH.py [extra hipcc flags...] -> build_ab/lib_NAME.so
# Compiles SRC's device code to assembly, pipes it through PATCH.py (stdin -> stdout), assembles the
# result and links it with the regular objects: for hardware experiments that need an instruction
# stream hipcc would not emit (inserted s_nop, moved instructions).
set -e
cd "$(dirname "$0")/../posegen_amd/csrc"
name=$1; src=$2; patch=$3; shift 3
base=${src%.hip}
LLVM=/opt/rocm/lib/llvm/bin
out=../../build_ab; mkdir -p $out
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FLAGS "$@" -S --cuda-device-only -o $out/${base}_$name.orig.s $src
python3 $patch < $out/${base}_$name.orig.s > $out/${base}_$name.s
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $out/${base}_$name.s -o $out/${base}_$name.dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $out/${base}_$name.hsaco $out/${base}_$name.dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
    -input=/dev/null -input=$out/${base}_$name.hsaco -output=$out/${base}_$name.hipfb
/opt/rocm/bin/hipcc $FLAGS "$@" --cuda-host-only -c $src -Xclang -fcuda-include-gpubinary -Xclang $out/${base}_$name.hipfb -o $out/${base}_$name.o
objs=""
for o in pg_api pg_eval16 pg_eval16r pg_rayrec pg_eval32 pg_evalc pg_kernels pg_pack; do
  if [ "$o" = "$base" ]; then objs="$objs $out/${base}_$name.o"; else objs="$objs ../_lib/obj/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/lib_$name.so $objs
echo built build_ab/lib_$name.so
