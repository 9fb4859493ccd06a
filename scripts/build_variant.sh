#!/bin/bash
# A variant build of libaltro_hip.so for A/B runs (ALTRO_HIP_LIB): scripts/build_variant.sh <name> <extra compiler flags...>
#   -> altro-cpp_amd/csrc/_x/libaltro_<name>.so (objects in _x/<name>/)
cd "$(dirname "$0")/../altro-cpp_amd/csrc" || exit 1
name=$1; shift
FLAGS=$(grep '^CXXFLAGS' Makefile | sed 's/^CXXFLAGS := //; s/\$(ARCH)/gfx950/')
mkdir -p _x/$name
pids=()
for f in inst_unicycle_f64 inst_unicycle_r32 inst_tripleint_f64 inst_tripleint_r32 inst_quad12_f64 inst_quad12_r32; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $f.hip -o _x/$name/$f.o & pids+=($!)
done
/opt/rocm/bin/hipcc $FLAGS "$@" -x hip -c altro_capi.cpp -o _x/$name/altro_capi.o & pids+=($!)
for p in "${pids[@]}"; do wait $p || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _x/libaltro_$name.so _x/$name/*.o && echo "built _x/libaltro_$name.so"
