"""Experiment: does the ORDER of the queries matter to k_match?  The occupancy rows a window probes are those of the similar first 3-mers of its k-mer; with the
residues renumbered C | GASTP | NDEQ | KRH | YWF | MLIV the row index is r0 + 20 r1 + 400 r3, so windows whose residue at offset 3 falls into the same exchange group probe
the same sixth of the bit table.  Sorting a batch of 10-residue peptides by (group of residue 3, group of residue 4) -- the two scanned windows -- keeps the working set
of every moment at 2/6 of the table, inside one XCD's 4 MB L2.  usage: python tools/exp_query_order.py [none|g34|r34|ia]"""
import os, sys, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import spacepharer_amd as sa
mode = sys.argv[1] if len(sys.argv) > 1 else "g34"
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
tres, toff, qres, qoff = bench.make_workload(ns, 50000, 200, 12345)
nq = len(qoff) - 1
q = qres.reshape(nq, 10)
order = "CGASTPNDEQKRHYWFMLIV"
rank = np.zeros(256, dtype=np.int64); group = np.zeros(256, dtype=np.int64)
for i, c in enumerate(order): rank[ord(c)] = i
for g, s in enumerate(["C", "GASTP", "NDEQ", "KRH", "YWF", "MLIV"]):
    for c in s: group[ord(c)] = g
if mode == "g34": key = group[q[:, 3]] * 6 + group[q[:, 4]]
elif mode == "r34": key = rank[q[:, 3]] * 20 + rank[q[:, 4]]
elif mode == "ia": key = ((rank[q[:, 3]] * 20 + rank[q[:, 1]]) * 20 + rank[q[:, 0]]) * 20 + rank[q[:, 4]]
else: key = np.arange(nq)
perm = np.argsort(key, kind="stable")
q2 = np.ascontiguousarray(q[perm])
ctx = sa.SearchContext()
ctx.set_targets_raw(tres.tobytes(), bench.u64p(toff), len(toff) - 1)
ctx.set_queries_raw(q2.tobytes(), bench.u64p(qoff), nq)
ctx.set_timing(2)
ms = []
for _ in range(12):
    ctx.prefilter(); ms.append(ctx.stat("ms_match"))
ms = sorted(ms[2:])
print("order %-5s grouped=%s: k_match min %.4f median %.4f ms   K=%d M=%d D=%d" % (mode, os.environ.get("SP_MATCH_GROUP", "1"), ms[0], ms[len(ms) // 2], ctx.stat("kmers"), ctx.stat("db_matches"), ctx.stat("diagonals")))
# ---- XCD affinity (SP_EXP_XCD=1): all queries dealt statically (match_static 8): chunk c of four queries goes to wave c % nWaves, block wave / 4, XCD block % 8;
# queries whose residue 3 falls into the x-th eighth of the renumbered alphabet are placed into chunks that land on XCD x
if os.environ.get("SP_EXP_XCD"):
    perCU = int(os.environ.get("SP_MATCH_PERCU", "4"))
    nWaves = 256 * perCU * 4
    nChunks = nq // 4
    xcd_of_chunk = ((np.arange(nChunks) % nWaves) // 4) % 8
    for label, cls in (("random", np.random.default_rng(5).integers(0, 8, nq)), ("residue 3", (rank[q[:, 3]] * 8) // 20)):
        slots = [list(np.nonzero(xcd_of_chunk == x)[0]) for x in range(8)]
        members = [list(np.nonzero(cls == x)[0]) for x in range(8)]
        perm2 = np.full(nq, -1, dtype=np.int64); spill = []
        for x in range(8):
            cap = len(slots[x]) * 4
            take, rest = members[x][:cap], members[x][cap:]
            spill += rest
            for k, qi in enumerate(take): perm2[slots[x][k // 4] * 4 + k % 4] = qi
        free = np.nonzero(perm2 < 0)[0]
        perm2[free] = np.array(spill + [i for i in range(nq) if False], dtype=np.int64)[:len(free)] if len(spill) >= len(free) else np.concatenate([np.array(spill, dtype=np.int64), np.setdiff1d(np.arange(nq), np.concatenate([perm2[perm2 >= 0], np.array(spill, dtype=np.int64)]))])[:len(free)]
        assert len(np.unique(perm2)) == nq
        q3 = np.ascontiguousarray(q[perm2])
        ctx.set_queries_raw(q3.tobytes(), bench.u64p(qoff), nq)
        ctx.set_tuning("match_static", 8)
        ms = []
        for _ in range(12):
            ctx.prefilter(); ms.append(ctx.stat("ms_match"))
        ms = sorted(ms[2:])
        print("all static, class by %-9s (%.0f %% of the queries on their XCD): k_match min %.4f median %.4f ms" % (label, 100.0 * (1 - len(spill) / nq), ms[0], ms[len(ms) // 2]))
