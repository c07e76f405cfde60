#!/bin/bash
# Z-curve / tiled slot numbering on the GPU box: golden parity, kernel times + crc, L2 miss counters, the default bench line
mkdir -p gpurun_out
bash tools/gpu_grp_ab.sh > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_search_step.py tests/test_gpu_dense.py tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/zorder_tests.txt
bash tools/gpu_grp_traffic.sh > /dev/null 2>&1
timeout 900 python bench.py --no-dense > gpurun_out/zorder_bench.json 2> gpurun_out/zorder_bench.err
cat gpurun_out/grp_ab.txt gpurun_out/zorder_tests.txt; tail -3 gpurun_out/zorder_bench.json | cut -c1-1500
