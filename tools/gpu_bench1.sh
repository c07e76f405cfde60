cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout 600 python bench.py --stage prefilter --steps 5 --warmup 1 --cpu-sample 1500 2>&1 | tail -5 | tee gpurun_out/bench_prefilter_r01.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_prefilter -o pf -- python3 $GRAFT_REPO_ROOT/bench.py --stage prefilter --steps 5 --warmup 1 --cpu-sample 0 2>&1 | tail -3
find $GRAFT_REPO_ROOT/gpurun_out/prof_prefilter -name '*stats*' | head; 
