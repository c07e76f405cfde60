#!/usr/bin/env python3
"""Experiment (not a bench): how much of k_match is HBM line traffic?  Same 60 000-peptide batch size, but the batch is made of
`distinct` different peptides repeated -- everything a repeated peptide touches is in L2 after its first copy.
usage: python tools/exp_locality.py [distinct ...]"""
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
q = qres.reshape(nq, 10)
for distinct in [int(x) for x in sys.argv[1:]] or [nq, 6000, 600, 60]:
    qq = np.ascontiguousarray(q[np.arange(nq) % distinct]).reshape(-1)
    ctx.set_queries_raw(qq.tobytes(), bench.u64p(qoff), nq)
    ms = []
    for it in range(6):
        ctx.prefilter()
        ctx.hits_compact()
        if it >= 2:
            ms.append(ctx.stat("ms_match"))
    print("distinct peptides %6d: k_match %.3f ms   similar k-mers %.3g  index entries %.3g" % (distinct, sum(ms) / len(ms), ctx.stat("kmers"), ctx.stat("db_matches")), flush=True)
