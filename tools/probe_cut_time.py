"""the probe-list cut against the plain target-sharded pass, one GPU playing shard 0 of S (100 000 spacers against 1 / S of 1e7 residues):
  plain     every shard enumerates all queries itself                      (k_match_grp<true>: sp_ctx_prefilter_candidates)
  cut       enumerate the slice 1 / S of the queries against the UNION map (k_match_grp<true, 1>: sp_ctx_probe_enumerate, wall clock incl. k_prep8 and the counts' copy)
            + fetch the probes the S enumerating ranks address to this shard (k_match_grp<true, 2>: sp_ctx_probe_fetch; placing + kernel; ms_match = the kernel alone)
All S shard contexts live on this GPU (their maps are needed for the union; their enumerations produce the probes shard 0 receives).
usage: python tools/probe_cut_time.py [S]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spacepharer_amd as sa  # noqa: E402
from spacepharer_amd.dist import decompose_by_residues  # noqa: E402
import numpy as np  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NQ = int(os.environ.get("SPACERS", "100000"))
tres, toff, qres, qoff = bench.make_workload(NQ, 50000, 200, 12345)
lens = np.diff(toff).astype(np.int64) + 2
ctxs = []
for r in range(S):
    start, count = decompose_by_residues(lens, r, S)
    c = sa.SearchContext()
    c.set_targets_shard_raw(tres.tobytes(), bench.u64p(toff), 50000, start, start + count)
    c.set_queries_raw(qres.tobytes(), bench.u64p(qoff), len(qoff) - 1)
    c.set_timing(2)
    c.set_tuning("match_blocks_per_cu", int(os.environ.get("BLOCKS_PER_CU", "-1")))      # -1: all that fit, as for the one context a rank of the sharded job has (0: the library's choice -- two while it sees several contexts searching on the device)
    ctxs.append(c)
c0 = ctxs[0]
ms = []
for _ in range(8):
    c0.prefilter_candidates()
    ms.append(c0.stat("ms_match"))
plain = np.ascontiguousarray(c0.candidates())
print("plain: shard 1/%d, %d queries: k_match_grp<true> min %.3f median %.3f ms (prep %.3f), %d candidates" % (S, NQ, min(ms[2:]), sorted(ms[2:])[3], c0.stat("ms_prep"), len(plain)))
maps = [c.shard_occupancy() for c in ctxs]
print("occupancy of the slot table: shards %s, union %.4f" % (" ".join("%.4f" % (np.unpackbits(m).sum() / 64e6) for m in maps), np.unpackbits(np.bitwise_or.reduce(np.stack(maps))).sum() / 64e6))
for c in ctxs:
    c.probe_set_shards(maps)
qlens = np.diff(qoff).astype(np.int64) + 2
sl = [decompose_by_residues(qlens, r, S) for r in range(S)]
recs, en = [], []
per_query = np.zeros(len(qoff) - 1, dtype=np.uint32)
for r, c in enumerate(ctxs):
    a, n = sl[r]
    best = 1e9
    for _ in range(4 if r == 0 else 1):
        c.sync()
        t0 = time.perf_counter()
        out = c.probe_enumerate(a, a + n)
        best = min(best, time.perf_counter() - t0)
        if r == 0:
            en_kernel = c.stat("ms_match")
    en.append(best * 1e3)
    recs.append(out[0][0]); per_query[a:a + n] = out[0][1]
    if r == 0:
        print("cut, enumerate: slice of %d queries -> %s records per destination (16 bytes each: %.1f MB sent in total), k_match_grp<true, 1> %.3f ms (wall %.3f ms incl. k_prep8 of all queries + copy of the lists to the host)"
              % (n, [len(o[0]) for o in out], sum(len(o[0]) for o in out) * 16 / 1e6, en_kernel, best * 1e3))
R = np.concatenate(recs)
ms, wall = [], []
for _ in range(8):
    c0.sync()
    t0 = time.perf_counter()
    n = c0.probe_fetch(R, per_query)
    wall.append((time.perf_counter() - t0) * 1e3)
    ms.append(c0.stat("ms_match"))
cut = np.ascontiguousarray(c0.candidates())
same = sorted(bytes(r) for r in cut) == sorted(bytes(r) for r in plain)
print("cut, fetch: %d probes received (%.1f MB): k_match_grp<true, 2> min %.3f median %.3f ms; wall incl. upload + placing + score + candidates %.3f ms; candidates equal the plain pass': %s"
      % (len(R), len(R) * 16 / 1e6, min(ms[2:]), sorted(ms[2:])[3], min(wall[2:]), same))
assert same
