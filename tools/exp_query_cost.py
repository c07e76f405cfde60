#!/usr/bin/env python3
"""Experiment: distribution of the cost of ONE query in k_match (similar k-mers generated, kernel time when it is alone)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import spacepharer_amd as sa
tres, toff, qres, qoff = bench.make_workload(10000, 50000, 200, 12345)
ctx = sa.SearchContext()
ctx.set_targets_raw(tres.tobytes(), bench.u64p(toff), len(toff) - 1)
nq = len(qoff) - 1
q = qres.reshape(nq, 10)
one_off = np.array([0, 10], dtype=np.uint64)
K, T, Mh = [], [], []
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for i in range(N):
    ctx.set_queries_raw(q[i].tobytes(), bench.u64p(one_off), 1)
    ctx.prefilter()
    K.append(ctx.stat("kmers")); T.append(ctx.stat("ms_match")); Mh.append(ctx.stat("db_matches"))
K, T, Mh = np.array(K), np.array(T), np.array(Mh)
for p in (0, 10, 50, 90, 99, 99.9, 100):
    print("percentile %5.1f: similar k-mers %8.0f   alone in k_match %.3f ms   index hits %6.0f" % (p, np.percentile(K, p), np.percentile(T, p), np.percentile(Mh, p)))
o = np.argsort(K)[-5:]
print("heaviest:", [(int(K[i]), round(float(T[i]), 3), q[i].tobytes().decode()) for i in o])
print("corr time~K: %.3f ; fit ms = %.4f + %.3f us per 1000 k-mers" % (np.corrcoef(K, T)[0, 1], *np.polyfit(K, T, 1)[::-1] * np.array([1, 1e6])))
