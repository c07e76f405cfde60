# counter passes (each in its own run, --kernel-trace only) + the kernel-time profile of bench.py; summaries -> gpurun_out/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/pmc_sq $R/gpurun_out/pmc_lds $R/gpurun_out/prof_full
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o p -- python3 $R/tools/pmc_workload.py full 2>&1 | tail -3
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o p -- python3 $R/tools/pmc_workload.py full > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_sq -o p -- python3 $R/tools/pmc_workload.py full > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_lds -o p -- python3 $R/tools/pmc_workload.py full > /dev/null 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_traffic.json $(find gpurun_out/pmc_fetch gpurun_out/pmc_write -name '*counter_collection.csv')
python3 tools/pmc_summary.py gpurun_out/pmc_sq.json $(find gpurun_out/pmc_sq gpurun_out/pmc_lds -name '*counter_collection.csv') | cut -c1-400
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_full -o full -- python3 $R/bench.py --steps 100 --warmup 3 --regions 2 --cpu-sample 0 --no-module-e2e --no-dense --no-probe-proxy 2>&1 | grep "^{" | tail -1 > $R/gpurun_out/bench_under_rocprof.json
cat $R/gpurun_out/prof_full/full_kernel_stats.csv | cut -c1-200
cd $R && timeout 900 python3 bench.py > gpurun_out/bench_default.json 2>gpurun_out/bench_default.err; cut -c1-600 gpurun_out/bench_default.json
