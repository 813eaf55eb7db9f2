cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
O=gpurun_out/r06
tl() { env "$@" CONTEXT=128 CAPACITY=768 GBX_QLINEAR_LIB=scratch/stamps/libgbx_stamps.so timeout 300 python scratch/layer_timeline2.py 2>&1 | grep -v amdgpu.ids | head -34; }
echo "=== baseline (no probe stream)" > $O/tl_probe_b.txt; tl X=0 >> $O/tl_probe_b.txt
for kb in 112 88 72 56; do
echo "=== mode 17: producers $kb KB behind a workgroup barrier behind their granule publish" >> $O/tl_probe_b.txt; tl GBX_PROBE_MODE=17 GBX_PROBE_KB_P=$kb >> $O/tl_probe_b.txt
done
echo "=== mode 21: producers 72 KB (barrier) + attention workgroups 64 KB behind their output" >> $O/tl_probe_b.txt; tl GBX_PROBE_MODE=21 GBX_PROBE_KB_P=72 GBX_PROBE_KB_A=64 >> $O/tl_probe_b.txt
echo "=== mode 29: + attention workgroups 32 KB early (behind K/V prefetch and o words)" >> $O/tl_probe_b.txt; tl GBX_PROBE_MODE=29 GBX_PROBE_KB_P=72 GBX_PROBE_KB_A=32 GBX_PROBE_KB_A0=32 >> $O/tl_probe_b.txt
echo "=== baseline again" >> $O/tl_probe_b.txt; tl X=0 >> $O/tl_probe_b.txt
grep -E "^===|span" $O/tl_probe_b.txt
