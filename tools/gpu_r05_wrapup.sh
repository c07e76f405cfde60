#!/bin/bash
# round-5 wrap-up on the GPU box: the dense ablation ladder from HEAD (make variant NAME=abdN DEFS=-DSP_ABLATE_DENSE=N, N = 2 3 5 6 7 8), the round's profile set, the probe-list cut, configs[4]'s chain
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05
L="libspacepharer_amd.so libspacepharer_amd_abd2.so libspacepharer_amd_abd3.so libspacepharer_amd_abd5.so libspacepharer_amd_abd6.so libspacepharer_amd_abd7.so libspacepharer_amd_abd8.so"
LIBS="$L" WL=c REPS=8 bash tools/gpu_ab_libs.sh r05/dense_ablation_percu3 > /dev/null 2>&1
SP_MATCH_PERCU=2 LIBS="$L" WL=c REPS=8 bash tools/gpu_ab_libs.sh r05/dense_ablation_percu2 > /dev/null 2>&1
bash tools/gpu_r05_final.sh > gpurun_out/r05/final.log 2>&1
bash tools/gpu_probe_cut.sh > /dev/null 2>&1
bash tools/gpu_pm_full.sh > gpurun_out/r05/pm.log 2>&1
timeout 300 python3 tools/dense_select_probe.py > gpurun_out/r05/dense_select.txt 2>&1
ls gpurun_out/r05
