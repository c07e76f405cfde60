#!/bin/bash
# round-4 profile set, part 1: matching kernels on the three proxy workloads, ablation counters, occupancy, wave balance, locality experiments
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r04
O=gpurun_out/r04
{ echo "tools/kmatch_time.py: HIP-event time of the matching kernel, one context alone, 12 passes; SP_MATCH_GROUP=1 (k_match_grp, default) / 0 (k_match); heavy queries by exact counts (the default while one context searches) and by the estimate";
  for g in 1 0; do for e in 1 0; do echo "== SP_MATCH_GROUP=$g SP_MATCH_WEIGH_EXACT=$e"; SP_MATCH_GROUP=$g SP_MATCH_WEIGH_EXACT=$e timeout 600 python tools/kmatch_time.py a b 12 2>&1 | grep workload; done; done;
  echo "== workload c (dense index: k_match<512,false,false>, the large footprint)"; timeout 600 python tools/kmatch_time.py c 8 2>&1 | grep workload; } > $O/kmatch_workloads.txt 2>&1
{ echo "tools/gpu_ablate.sh on k_match_grp (developer builds -DSP_ABLATE=n: 1 no double-hit tail, 2 also no slot fetch, 3 also no probe / compaction, 4 also no enumeration of the tile)"; SP_MATCH_GROUP=1 bash tools/gpu_ablate.sh;
  echo "k_match<256,false,true> for comparison (the same cut points)"; SP_MATCH_GROUP=0 LIBS="libspacepharer_amd.so libspacepharer_amd_ab1.so libspacepharer_amd_ab2.so libspacepharer_amd_ab3.so" bash tools/gpu_ablate.sh; } > $O/kmatch_ablation.txt 2>&1
{ echo "blocks per CU (SP_MATCH_PERCU) x kernel: exclusive time on workload a, estimated weights"; EXACT=0 bash tools/gpu_grp_occ.sh 2>&1 | grep -v "^\[gpurun"; } > $O/kmatch_occupancy.txt 2>&1
{ echo "per-wave busy time (SP_DEBUG): how evenly the 4096 persistent waves finish"; for g in 1 0; do for e in 1 0; do echo "== group=$g exact=$e"; SP_DEBUG=1 SP_MATCH_GROUP=$g SP_MATCH_WEIGH_EXACT=$e python tools/kmatch_time.py a 6 2>&1 | grep "k_match waves\|busiest" | tail -2; done; done; } > $O/kmatch_wave_balance.txt 2>&1
{ echo "tools/exp_query_order.py: the order / placement of the queries (L2 locality of the occupancy rows)"; for g in 1 0; do for m in none g34 r34 ia; do SP_MATCH_GROUP=$g python tools/exp_query_order.py $m 2>&1 | tail -1; done; done;
  echo "XCD affinity (query placed on the XCD owning the row class of its first window), with L2 counters:"; bash tools/gpu_xcd_pmc.sh 2>&1 | grep "all static\|original order"; } > $O/kmatch_locality.txt 2>&1
bash tools/gpu_grp_pmc.sh "1 0" > /dev/null 2>&1; cp gpurun_out/grp_pmc.txt $O/kmatch_sq_counters.txt
bash tools/gpu_grp_traffic.sh "1 0" > /dev/null 2>&1; cp gpurun_out/grp_traffic.txt $O/kmatch_traffic.txt
ls -la $O
