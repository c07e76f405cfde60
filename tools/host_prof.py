"""host-side timing of the calls of one bench step (wall clock per call against the kernel times the library reports)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, spacepharer_amd as sa
tres, toff, qres, qoff = bench.make_workload(10000, 50000, 200, 12345)
ctx = sa.SearchContext()
ctx.set_targets_raw(tres.tobytes(), bench.u64p(toff), len(toff) - 1)
nq = len(qoff) - 1
ctx.set_queries_raw(qres.tobytes(), bench.u64p(qoff), nq)
ctx.set_timing(2)
acc = [0.0] * 4
N = 30
for it in range(N + 5):
    t0 = time.perf_counter(); ctx.prefilter(); t1 = time.perf_counter(); n = ctx.num_hits(); ctx.align_hits_begin(); t2 = time.perf_counter(); ctx.hits_view(); t3 = time.perf_counter()
    if it >= 5:
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2
        pre = sum(ctx.stat(k) for k in ("ms_prep", "ms_match", "ms_score", "ms_select"))
        aln = sum(ctx.stat(k) for k in ("ms_sw_forward", "ms_sw_reverse", "ms_traceback"))
        acc[3] += 0
ctx.align_hits_end()
print("per step: prefilter() %.3f ms (kernels %.3f)   align_hits_begin() %.3f ms (kernels %.3f)   hits_view() %.3f ms" % (acc[0] / N * 1e3, pre, acc[1] / N * 1e3, aln, acc[2] / N * 1e3))
