#!/bin/bash
# usage: tools/build_variant.sh /tmp/libaps_x.so -DAPS_TS_WH=4 ...   (a tuning build of the library next to the shipped one)
out=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
pkg="$root/hydrodynamic-limits-of-active-particle-systems-with-mean-field-interactions_amd"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared "$@" -I "$root/include" -o "$out" "$pkg/csrc/aps_hip.hip" "$pkg/csrc/pde_hip.hip" "$pkg/csrc/gillespie_hip.hip" "$pkg/csrc/gillespie_big_hip.hip"
