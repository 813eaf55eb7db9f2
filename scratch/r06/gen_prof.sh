cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
python scratch/gen_rate2.py 2>&1 | grep -v amdgpu.ids | head -8
cd /tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/gen1 -- python3 $R/scratch/gen_rate.py S22 300 > $O/gen1.log 2>&1
tail -3 $O/gen1.log
ls $O/gen1/*/ | head
python3 $R/profiles/summarize.py $(ls $O/gen1/*/*_kernel_stats.csv | head -1) 30 2>&1 | grep -v "dblock\|dgemv\|dattn\|dg_pack\|pack_sz" | cut -c1-170 | head -30
for f in $O/gen1/*/*memory_copy_stats.csv; do echo $f; head -8 $f; done
