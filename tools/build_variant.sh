#!/bin/bash
# development helper: build an experimental variant of the library next to the product one
# usage: tools/build_variant.sh <name> [-DFLAG ...]   ->  build/variants/<name>.so  (use with BALATRO_MI355X_LIB)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/variants
hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value "$@" \
  -o build/variants/$name.so balatro_gym_amd/csrc/bg_lib.hip
echo build/variants/$name.so
