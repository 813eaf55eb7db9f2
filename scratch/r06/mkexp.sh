# usage: mkexp.sh NAME "-DFLAG ..."   -> scratch/exp/lib_NAME.so (dblock.hip rebuilt with the flags, other objects from the product build)
set -e
cd /root/repo/gbx-lm_amd/csrc
make -j8 libgbx_qlinear.so >/dev/null
mkdir -p ../../scratch/exp
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-value -ffp-contract=on $2 -c dblock.hip -o /tmp/dblock_$1.o
OBJS=$(ls _obj/*.o | grep -v '/x_' | grep -v dblock.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/exp/lib_$1.so $OBJS /tmp/dblock_$1.o
echo built lib_$1.so
