#!/bin/bash
# diagnostic build of the PRODUCT library with in-kernel s_memrealtime stamps (-DGBX_STAMPS) -> scratch/stamps/libgbx_stamps.so
set -e
cd "$(dirname "$0")/../gbx-lm_amd/csrc"
mkdir -p ../../scratch/stamps/_obj
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-value -ffp-contract=on -DGBX_STAMPS $EXTRA_FLAGS"
SRCS=$(make -s -p -n 2>/dev/null | grep '^SRCS :=' | head -1 | sed 's/^SRCS := //')
objs=""
for f in $SRCS; do
  o=../../scratch/stamps/_obj/${f%.hip}.o
  objs="$objs $o"
  if [ ! -f $o ] || [ $f -nt $o ] || [ -n "$(find . -name '*.h' -newer $o)" ]; then
    echo "/opt/rocm/bin/hipcc $FLAGS -c $f -o $o"
  fi
done | xargs -P 8 -I{} bash -c "{}"
objs=$(for f in $SRCS; do echo ../../scratch/stamps/_obj/${f%.hip}.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/stamps/libgbx_stamps.so $objs
