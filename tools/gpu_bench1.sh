cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout 600 python bench.py --steps 10 --warmup 2 --cpu-sample 1500 2>&1 | tail -3 | tee gpurun_out/bench_full_r01.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_full -o full -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-sample 0 2>&1 | tail -2
find $GRAFT_REPO_ROOT/gpurun_out/prof_full -type f | head
