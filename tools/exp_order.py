"""experiment: does processing queries in an order that groups similar k-mer neighbourhoods speed up k_match (L2 locality)?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, spacepharer_amd as sa
tres, toff, qres, qoff = bench.make_workload(10000, 50000, 200, 12345)
nq = len(qoff) - 1
q = qres.reshape(nq, 10)
lut = np.zeros(256, dtype=np.int64); lut[list(b"ACDEFGHIKLMNPQRSTVWY")] = np.arange(20)
n = lut[q]
ib0 = n[:, 4] + 20 * n[:, 5] + 400 * n[:, 7]
ia0 = n[:, 0] + 20 * n[:, 1] + 400 * n[:, 3]
ctx = sa.SearchContext()
ctx.set_targets_raw(tres.tobytes(), bench.u64p(toff), len(toff) - 1)
def run(order, tag):
    qq = np.ascontiguousarray(q[order]).reshape(-1)
    ctx.set_queries_raw(qq.tobytes(), bench.u64p(qoff), nq)
    ts = []
    for _ in range(6):
        ctx.prefilter(); ts.append(ctx.stat("ms_match"))
    print(tag, "k_match ms", round(min(ts), 4), "hits", int(ctx.stat("hits")))
run(np.arange(nq), "original order      ")
run(np.argsort(ib0, kind="stable"), "sorted by ib(win 0)  ")
run(np.argsort(n[:, 7] * 400 + n[:, 5] * 20 + n[:, 4], kind="stable"), "sorted by ib (same)  ")
run(np.lexsort((ia0, ib0)), "sorted by (ib, ia)   ")
run(np.argsort(n[:, 7], kind="stable"), "sorted by residue 7  ")
