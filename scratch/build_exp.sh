#!/bin/bash
# build an experiment library: scratch/build_exp.sh <name> <source.hip> <extra flags...>  ->  scratch/exp/lib_<name>.so
set -e
cd "$(dirname "$0")/../gbx-lm_amd/csrc"
name=$1; src=$2; shift 2
make -s
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-value -ffp-contract=on "$@" -c $src -o ../../scratch/exp/${name}.o
objs=$(ls _obj/*.o | grep -v "_obj/x_" | grep -v "_obj/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/exp/lib_${name}.so $objs ../../scratch/exp/${name}.o
rm ../../scratch/exp/${name}.o
