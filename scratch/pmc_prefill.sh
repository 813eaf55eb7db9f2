set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmcp/p1 -- python3 $R/scratch/one_prefill.py 16384 4096 4096 4 > $R/gpurun_out/pmcp/p1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/pmcp/p2 -- python3 $R/scratch/one_prefill.py 16384 4096 4096 4 > $R/gpurun_out/pmcp/p2.log 2>&1
ls -R $R/gpurun_out/pmcp | head -30
