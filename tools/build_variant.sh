#!/bin/bash
# Tuning aid: csrc/variants/<name>.so = the library with ONE source recompiled under extra flags.
#   tools/build_variant.sh <name> <source.hip> [flags...]      (run `make -C ebnerd-benchmark_amd/csrc` first)
set -e
cd "$(dirname "$0")/../ebnerd-benchmark_amd/csrc"
name=$1; src=$2; shift 2
mkdir -p variants
extra=""
[ "$src" = ebn_attention_mfma.hip ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -munsafe-fp-atomics -Wno-unused-function $extra "$@" -c $src -o variants/$name.o
objs=""
for o in *.o; do [ "$o" = "${src%.hip}.o" ] && objs="$objs variants/$name.o" || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o variants/$name.so
rm -f variants/$name.o
ls -la variants/$name.so
