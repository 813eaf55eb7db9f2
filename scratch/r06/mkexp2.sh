# usage: mkexp2.sh NAME "-DFLAG ..." file1.hip [file2.hip ...]  -> scratch/exp/lib_NAME.so (those files rebuilt with the flags, other objects from the product build)
set -e
cd /root/repo/gbx-lm_amd/csrc
make -j8 libgbx_qlinear.so >/dev/null
mkdir -p ../../scratch/exp
name=$1; flags=$2; shift 2
OBJS=$(ls _obj/*.o | grep -v '/x_')
extra=""
for f in "$@"; do
  b=${f%.hip}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-value -ffp-contract=on $flags -c $f -o /tmp/${b}_$name.o &
  OBJS=$(echo "$OBJS" | grep -v "_obj/$b.o")
  extra="$extra /tmp/${b}_$name.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/exp/lib_$name.so $OBJS $extra
echo built lib_$name.so
