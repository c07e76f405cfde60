# GPU box: per-pipe activity of the bench workload's kernels (SQ_ACTIVE_INST_VALU / _SCA, scalar cycles, LDS issue waits).  NOTE: the TA_* / TCP_* counters
# (TA_TA_BUSY, TCP_PENDING_STALL_CYCLES, TCP_TCC_READ_REQ ...) never complete on this pool -- each such pass sat until its timeout with "1 incomplete
# dispatch" -- and rocprofv3's PC sampling reports "configuration is not supported on any of the agents"; neither is attempted here.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_l1d
timeout 400 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/pmc_l1d -o p -- python3 $R/tools/pmc_workload.py full 2>&1 | tail -2
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_l1.json $(find gpurun_out/pmc_l1d -name '*counter_collection.csv') > /dev/null
