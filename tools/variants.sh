#!/bin/bash
# build compile-time variants of the library next to the product one: tools/variants.sh name "-DX=1 -DY=2" [name2 "flags" ...]
# -> pecaller_amd/libpemap_hip.<name>.so ; run with PEMAP_LIB=$PWD/pecaller_amd/libpemap_hip.<name>.so
cd $(dirname $0)/..
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -mno-amdgpu-ieee -fPIC -shared -Wno-unused-function -Wno-unused-value -Wno-uninitialized $f \
    -o pecaller_amd/libpemap_hip.$n.so pecaller_amd/csrc/pemap_capi.hip pecaller_amd/csrc/pecall_capi.hip &
done
wait
ls -la pecaller_amd/libpemap_hip.*.so
