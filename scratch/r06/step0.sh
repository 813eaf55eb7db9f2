cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 300 scratch/probe7/bin/probe_block > $O/probe_block.txt 2>&1
tail -30 $O/probe_block.txt
tl() { env "$@" CONTEXT=128 CAPACITY=768 GBX_QLINEAR_LIB=scratch/stamps/libgbx_stamps.so timeout 300 python scratch/layer_timeline2.py 2>&1 | grep -v amdgpu.ids | head -34; }
echo "=== baseline (no probe stream)" > $O/tl_probe.txt; tl X=0 >> $O/tl_probe.txt
echo "=== mode 1: producers 112 KB behind their granule publish" >> $O/tl_probe.txt; tl GBX_PROBE_MODE=1 GBX_PROBE_KB_P=112 >> $O/tl_probe.txt
echo "=== mode 3: the same, the two H2-polling waves issue none" >> $O/tl_probe.txt; tl GBX_PROBE_MODE=3 GBX_PROBE_KB_P=112 >> $O/tl_probe.txt
echo "=== mode 1: producers 72 KB" >> $O/tl_probe.txt; tl GBX_PROBE_MODE=1 GBX_PROBE_KB_P=72 >> $O/tl_probe.txt
echo "=== mode 5: producers 112 KB + attention workgroups 90 KB behind their output" >> $O/tl_probe.txt; tl GBX_PROBE_MODE=5 GBX_PROBE_KB_P=112 GBX_PROBE_KB_A=90 >> $O/tl_probe.txt
echo "=== mode 13: + attention workgroups 32 KB early (behind K/V prefetch), 58 KB behind their output" >> $O/tl_probe.txt; tl GBX_PROBE_MODE=13 GBX_PROBE_KB_P=112 GBX_PROBE_KB_A=58 GBX_PROBE_KB_A0=32 >> $O/tl_probe.txt
echo "=== baseline again" >> $O/tl_probe.txt; tl X=0 >> $O/tl_probe.txt
grep -E "^===|span|H2 received|s6 |s4 |s9 |s5 " $O/tl_probe.txt
