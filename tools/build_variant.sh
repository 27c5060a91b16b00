#!/bin/bash
# Builds a tuning variant of the HIP library: tools/build_variant.sh NAME [-DSKD_...=...]
# -> sasktran2_amd/_lib/variants/NAME.so (select it at run time with SASKTRAN2_AMD_LIB=<path>)
set -e
name=$1; shift
cd "$(dirname "$0")/../sasktran2_amd/csrc"
out=../_lib/variants; mkdir -p $out/$name
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-variable -Wno-unused-but-set-variable"
/opt/rocm/bin/hipcc $F "$@" -c hip/disco_engine.hip -o $out/$name/disco_engine.o &
/opt/rocm/bin/hipcc $F "$@" -c hip/capi.hip -o $out/$name/capi.o &
wait
# the other engines are not touched by the SKD_ switches: their objects come from the regular build (make in sasktran2_amd/csrc)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/$name.so $out/$name/disco_engine.o $out/$name/capi.o ../_lib/so_engine.o ../_lib/ts_engine.o -ldl -Wl,-rpath,/opt/rocm/lib
rm -rf $out/$name
echo built $out/$name.so
