"""the matching kernel of the target-SHARDED search (k-mer ordinals recorded) alone on the GPU: 100 000 spacers against one shard (1 / S of 1e7 residues);
SP_MATCH_GROUP=1 (k_match_grp<true>, default) / 0 (k_match<256, true, true>).  usage: python tools/shard_kmatch_time.py [S]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spacepharer_amd as sa  # noqa: E402
from spacepharer_amd.dist import decompose_by_residues  # noqa: E402
import numpy as np  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
tres, toff, qres, qoff = bench.make_workload(100000, 50000, 200, 12345)
lens = np.diff(toff).astype(np.int64) + 2
start, count = decompose_by_residues(lens, 0, S)
ctx = sa.SearchContext()
ctx.set_targets_shard_raw(tres.tobytes(), bench.u64p(toff), 50000, start, start + count)
ctx.set_queries_raw(qres.tobytes(), bench.u64p(qoff), len(qoff) - 1)
ctx.set_timing(2)
ms = []
for _ in range(8):
    ctx.prefilter_candidates()
    ms.append(ctx.stat("ms_match"))
c = ctx.candidates()
print("SP_MATCH_GROUP=%s shard 1/%d: k_match min %.3f median %.3f ms, %d candidates, digest %08x" % (os.environ.get("SP_MATCH_GROUP", "1"), S, min(ms[2:]), sorted(ms[2:])[3], len(c), __import__("zlib").crc32(np.ascontiguousarray(c).tobytes())))
