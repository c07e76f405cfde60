"""dense workload (10 000 spacers vs 2e8 residues): distribution of the double-hit diagonals per query (what k_select's paths see) and k_select's exclusive time"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spacepharer_amd as sa  # noqa: E402
import numpy as np  # noqa: E402

tres, toff, qres, qoff = bench.make_workload(10000, int(os.environ.get("TARGETS", "1000000")), 200, 12345)
ctx = sa.SearchContext()
ctx.set_targets_raw(tres.tobytes(), bench.u64p(toff), len(toff) - 1)
ctx.set_queries_raw(qres.tobytes(), bench.u64p(qoff), len(qoff) - 1)
ctx.set_timing(2)
for _ in range(5):
    ctx.prefilter()
print("ms: match %.3f score %.3f select %.3f" % (ctx.stat("ms_match"), ctx.stat("ms_score"), ctx.stat("ms_select")))
hits, offs, total = ctx.hits_compact()
cnt = np.diff(np.asarray(offs[:len(qoff)], dtype=np.int64))
print("hits per query: mean %.1f  p50 %d p90 %d p99 %d max %d;  > 64: %d queries, > 300: %d; diagonals %d" % (cnt.mean(), np.percentile(cnt, 50), np.percentile(cnt, 90), np.percentile(cnt, 99), cnt.max(), (cnt > 64).sum(), (cnt >= 300).sum(), int(ctx.stat("diagonals"))))
