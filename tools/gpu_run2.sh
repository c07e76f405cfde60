cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout 300 python -m pytest tests/test_gpu_prefilter.py -x -q -m gpu --timeout 120 2>&1 | tail -5
timeout 300 python bench.py --stage prefilter --steps 10 --warmup 2 --cpu-sample 0 2>&1 | tail -3 | tee gpurun_out/bench_prefilter_r01.json
