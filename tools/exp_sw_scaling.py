#!/usr/bin/env python3
"""Experiment (not a bench): forward-SW time against the number of pairs -- shows the per-wave latency (few pairs: one wave
per SIMD) and the saturated rate.  usage: python tools/exp_sw_scaling.py"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import spacepharer_amd as sa

tres, toff, qres, qoff = bench.make_workload(10000, 50000, 200, 12345)
ctx = sa.SearchContext()
ctx.set_targets_raw(tres.tobytes(), bench.u64p(toff), len(toff) - 1)
nq = len(qoff) - 1
ctx.set_queries_raw(qres.tobytes(), bench.u64p(qoff), nq)
ctx.set_timing(2)
rng = np.random.default_rng(1)
for n in (256, 2048, 8192, 16384, 32768, 57344, 65536, 73465, 114688, 131072, 262144):
    qi = rng.integers(0, nq, size=n).astype(np.uint32)
    ti = rng.integers(0, len(toff) - 1, size=n).astype(np.uint32)
    ms = []
    for it in range(4):
        ctx.align_raw(qi, ti)
        if it:
            ms.append(ctx.stat("ms_sw_forward"))
    print("pairs %7d  waves(8/wave) %6d  per SIMD %.2f   sw_forward %.3f ms   %.2f ns/pair" % (n, (n + 7) // 8, (n + 7) // 8 / 1024, min(ms), min(ms) * 1e6 / n), flush=True)
