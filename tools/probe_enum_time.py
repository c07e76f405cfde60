"""the enumerating side of the probe-list cut alone: one shard context of S, maps of all S shards (built on the fly from S short-lived contexts), slice 1 / S of the queries;
HIP-event time of k_match_grp<true, 1> (ms_match).  usage: python tools/probe_enum_time.py [S] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spacepharer_amd as sa  # noqa: E402
from spacepharer_amd.dist import decompose_by_residues  # noqa: E402
import numpy as np  # noqa: E402
import ctypes as C  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
NQ = int(os.environ.get("SPACERS", "100000"))
tres, toff, qres, qoff = bench.make_workload(NQ, 50000, 200, 12345)
lens = np.diff(toff).astype(np.int64) + 2
maps = []
c0 = None
for r in range(S):
    start, count = decompose_by_residues(lens, r, S)
    c = sa.SearchContext()
    c.set_targets_shard_raw(tres.tobytes(), bench.u64p(toff), 50000, start, start + count)
    maps.append(c.shard_occupancy())
    if r == 0:
        c0 = c
    else:
        c.close()
c0.set_queries_raw(qres.tobytes(), bench.u64p(qoff), len(qoff) - 1)
c0.set_timing(2)
c0.probe_set_shards(maps)
qlens = np.diff(qoff).astype(np.int64) + 2
a, n = decompose_by_residues(qlens, 0, S)
counts = (C.c_uint64 * 8)()
ms = []
for _ in range(REPS):
    sa._lib._check(c0.L.sp_ctx_probe_enumerate(c0.ctx, a, a + n, counts))
    ms.append(c0.stat("ms_match"))
print("%s: S=%d slice %d queries -> %d records: k_match_grp<true, 1> min %.3f median %.3f ms" % (os.environ.get("SP_LIB_NAME", "libspacepharer_amd.so"), S, n, sum(counts), min(ms[1:]), sorted(ms[1:])[len(ms[1:]) // 2]))
