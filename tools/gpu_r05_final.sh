#!/bin/bash
# round-5 profile set at HEAD: everything profiles/r05_* quotes (run on the GPU box; results under gpurun_out/r05)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05
O=gpurun_out/r05
{ echo "tools/kmatch_time.py: HIP-event time of the matching kernel, one context alone; workloads a (10 000 spacers vs 1e7), b (100 000 vs 1e7), c (10 000 vs 2e8: the dense index, k_match<512,false,false>)";
  timeout 900 python tools/kmatch_time.py a b c 10 2>&1 | grep workload; 
  echo "per-wave busy time on workload c (SP_DEBUG):"; SP_DEBUG=1 timeout 300 python tools/kmatch_time.py c 4 2>&1 | grep "k_match waves\|busiest" | tail -2;
  echo "blocks per CU on workload c (SP_MATCH_PERCU = 1, 2, 3):"; for b in 1 2 3; do SP_MATCH_PERCU=$b timeout 300 python tools/kmatch_time.py c 6 2>&1 | grep workload; done; } > $O/kmatch_workloads.txt 2>&1
bash tools/gpu_pmc_dense.sh > $O/pmc_dense.txt 2>&1
rm -f gpurun_out/pmc_quick.txt
PASSES="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD;SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR;TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum;TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" bash tools/gpu_pmc_quick.sh > /dev/null 2>&1
cp gpurun_out/pmc_quick.txt $O/pmc_dense_sq_mem.txt
bash tools/gpu_pmc.sh > $O/pmc_log.txt 2>&1
cp gpurun_out/pmc_traffic.json gpurun_out/pmc_sq.json gpurun_out/bench_under_rocprof.json gpurun_out/bench_default.json $O/ 2>/dev/null
cp gpurun_out/prof_full/full_kernel_stats.csv $O/bench_full_kernel_stats.csv 2>/dev/null
# the driver's command line
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmdline.json 2> $O/bench_driver_cmdline.err
MODES="--pipeline=2,--host-driven --pipeline=2 --pipeline=3,--host-driven" bash tools/gpu_dense_step.sh > /dev/null 2>&1; cp gpurun_out/dense_step.txt $O/dense_step_modes.txt
bash tools/gpu_headline_sweep.sh > /dev/null 2>&1; cp gpurun_out/headline_sweep.txt $O/headline_sweep.txt
SPACERS=100000 bash tools/e2e_served_laps.sh > $O/served_laps.txt 2>&1
TARGETS=1000000 bash tools/e2e_cold_laps.sh > $O/cold_laps_dense.txt 2>&1
for l in libspacepharer_amd.so; do SP_LIB_NAME=$l python tools/index_time.py 50000 1000000; done > $O/index_time.txt 2>&1
ls -la $O
