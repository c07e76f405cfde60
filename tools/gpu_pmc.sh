cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc1 $GRAFT_REPO_ROOT/gpurun_out/pmc2 $GRAFT_REPO_ROOT/gpurun_out/pmc3
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc1 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --stage prefilter --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --stage prefilter --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc3 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --stage prefilter --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/pmc1 $GRAFT_REPO_ROOT/gpurun_out/pmc2 $GRAFT_REPO_ROOT/gpurun_out/pmc3
