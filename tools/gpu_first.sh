set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -m2 gfx
nproc; lscpu | grep -m1 'Model name'; free -g | head -2; getconf LEVEL2_CACHE_SIZE
make -s -C oracle liboracle.so
timeout 900 python -m pytest tests/test_gpu_prefilter.py -x -q -m gpu 2>&1 | tail -40
