#!/bin/bash
# experiment library with scratch/negative/mfma_batch2.hip linked in: scratch/build_v2.sh <name> <flags for mfma_batch2.hip...>  ->  scratch/exp/lib_<name>.so
set -e
cd "$(dirname "$0")/../gbx-lm_amd/csrc"
name=$1; shift
make -s
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-value -ffp-contract=on"
/opt/rocm/bin/hipcc $F -DBX_WITH_V2 -c mfma_batch.hip -o ../../scratch/exp/${name}_b1.o
/opt/rocm/bin/hipcc $F -DBX2_ONE "$@" -c ../../scratch/negative/mfma_batch2.hip -o ../../scratch/exp/${name}_b2.o
objs=$(ls _obj/*.o | grep -v "_obj/x_" | grep -v "_obj/mfma_batch.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/exp/lib_${name}.so $objs ../../scratch/exp/${name}_b1.o ../../scratch/exp/${name}_b2.o
rm ../../scratch/exp/${name}_b1.o ../../scratch/exp/${name}_b2.o
