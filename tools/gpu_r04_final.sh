#!/bin/bash
# round-4 profile set at HEAD (after the Z-curve slot numbering and the dense-index changes): everything profiles/r04_* quotes
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r04
O=gpurun_out/r04
{ echo "tools/kmatch_time.py: HIP-event time of the matching kernel, one context alone, 12 passes; SP_MATCH_GROUP=1 (k_match_grp, default) / 0 (k_match); heavy queries by exact counts (the default while one context searches) and by the estimate";
  for g in 1 0; do for e in 1 0; do echo "== SP_MATCH_GROUP=$g SP_MATCH_WEIGH_EXACT=$e"; SP_MATCH_GROUP=$g SP_MATCH_WEIGH_EXACT=$e timeout 600 python tools/kmatch_time.py a b 12 2>&1 | grep workload; done; done;
  echo "== workload c (dense index: k_match<512,false,false>, the large footprint)"; timeout 600 python tools/kmatch_time.py c 8 2>&1 | grep workload; } > $O/kmatch_workloads.txt 2>&1
{ echo "tools/gpu_ablate.sh on k_match_grp (developer builds -DSP_ABLATE=n: 1 no double-hit tail, 2 also no slot fetch, 3 also no probe / compaction, 4 also no enumeration of the tile)"; SP_MATCH_GROUP=1 bash tools/gpu_ablate.sh;
  echo "k_match<256,false,true> for comparison (the same cut points)"; SP_MATCH_GROUP=0 LIBS="libspacepharer_amd.so libspacepharer_amd_ab1.so libspacepharer_amd_ab2.so libspacepharer_amd_ab3.so" bash tools/gpu_ablate.sh; } > $O/kmatch_ablation.txt 2>&1
bash tools/gpu_grp_pmc.sh "1 0" > /dev/null 2>&1; cp gpurun_out/grp_pmc.txt $O/kmatch_sq_counters.txt
bash tools/gpu_grp_traffic.sh "1 0" > /dev/null 2>&1; cp gpurun_out/grp_traffic.txt $O/kmatch_traffic.txt
{ echo "dense index (workload c: 10 000 spacers vs 2e8 residues), k_match<512,false,false>: time with parts left out (make ablate_dense; wrong results, timing only):";
  echo "ab3 = enumeration + mapping only; abd2 + bucket loads; abd3 + emission of the bucket entries (no list tails), no tail of the query; abd5 all but the tail of the query; abd7 = abd5 without the stores of the LDS window; abd6 = abd5 without the seen / again marks; abd8 = everything, the tail without its reads of the scratch";
  LIBS="libspacepharer_amd.so libspacepharer_amd_ab3.so libspacepharer_amd_abd2.so libspacepharer_amd_abd3.so libspacepharer_amd_abd5.so libspacepharer_amd_abd7.so libspacepharer_amd_abd6.so libspacepharer_amd_abd8.so" bash tools/gpu_dense_ablate.sh;
  echo "phase shares of a wave's time (make phase): p0 query set-up, p1 window start, p2 enumeration, p3 mapping, p4 pre-filter rounds (scratch read back), p5 bucket phase, p6 sort, p7 records";
  SP_LIB_NAME=libspacepharer_amd_phase.so timeout 300 python tools/kmatch_time.py c 5 2>&1 | tail -2;
  echo "bucket phase split (make phase_split): p0 += wait for the buckets, p1 += scans and window moves, p2 += emission, p3 += list tails; p5 = issue of the loads";
  SP_LIB_NAME=libspacepharer_amd_phase2.so timeout 300 python tools/kmatch_time.py c 5 2>&1 | tail -2;
  echo "blocks per CU (SP_MATCH_PERCU = 1, 2, 3):"; for b in 1 2 3; do SP_MATCH_PERCU=$b timeout 300 python tools/kmatch_time.py c 6 2>&1 | grep workload; done; } > $O/dense_ablation.txt 2>&1
bash tools/gpu_pmc_dense.sh > $O/pmc_dense.txt 2>&1
bash tools/gpu_pmc.sh > $O/pmc_log.txt 2>&1
cp gpurun_out/pmc_traffic.json gpurun_out/pmc_sq.json gpurun_out/bench_under_rocprof.json gpurun_out/bench_default.json $O/ 2>/dev/null
cp gpurun_out/prof_full/full_kernel_stats.csv $O/bench_full_kernel_stats.csv 2>/dev/null
bash tools/e2e_served_laps.sh > $O/served_laps.txt 2>&1
timeout 1500 python tools/predictmatch_full.py --reference > $O/predictmatch_full.json 2> $O/predictmatch_full.err
ls -la $O
