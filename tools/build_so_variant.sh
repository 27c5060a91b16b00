#!/bin/bash
# tuning variant of the successive-orders engine: tools/build_so_variant.sh NAME [-DSKSO_...=...] -> sasktran2_amd/_lib/variants/NAME.so
set -e
name=$1; shift
cd "$(dirname "$0")/../sasktran2_amd/csrc"
out=../_lib/variants; mkdir -p $out/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-variable -Wno-unused-but-set-variable "$@" -c hip/so_engine.hip -o $out/$name/so_engine.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/$name.so $out/$name/so_engine.o ../_lib/disco_engine.o ../_lib/capi.o ../_lib/ts_engine.o
rm -rf $out/$name
echo built $out/$name.so
