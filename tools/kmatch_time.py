"""k_match A/B timing on the GPU box: HIP-event time of the matching kernel (one context, GPU to itself) on the three workloads
DESIGN.md quotes -- 10 000 spacers / 100 000 spacers vs 1e7 residues, 10 000 spacers vs 2e8 residues -- plus a crc of the delivered
hit list, so that two builds can be compared for speed AND bytes.  usage: python tools/kmatch_time.py [a] [b] [c] [reps]"""
import ctypes as C
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spacepharer_amd as sa  # noqa: E402

which = [a for a in sys.argv[1:] if a in ("a", "b", "c")] or ["a", "b", "c"]
reps = next((int(a) for a in sys.argv[1:] if a.isdigit()), 12)
W = {"a": (10000, 50000, 200), "b": (100000, 50000, 200), "c": (10000, 1000000, 200)}
for w in which:
    ns, nt, tl = W[w]
    tres, toff, qres, qoff = bench.make_workload(ns, nt, tl, 12345)
    ctx = sa.SearchContext()
    ctx.set_targets_raw(tres.tobytes(), bench.u64p(toff), nt)
    ctx.set_queries_raw(qres.tobytes(), bench.u64p(qoff), len(qoff) - 1)
    ctx.set_timing(2)
    ms, sel, prep = [], [], []
    for _ in range(reps):
        ctx.prefilter()
        ms.append(ctx.stat("ms_match")); sel.append(ctx.stat("ms_select")); prep.append(ctx.stat("ms_prep"))
    hp, op, n = ctx.hits_view()
    crc = zlib.crc32(C.string_at(hp, 8 * n), zlib.crc32(C.string_at(op, 4 * len(qoff))))
    K, M, D = (ctx.stat(k) for k in ("kmers", "db_matches", "diagonals"))
    algo = 22.0 * K + 6.0 * M + 7.0 * D
    ms = sorted(ms[2:])
    med = ms[len(ms) // 2]
    print("workload %s (%d spacers vs %.0e residues): k_match min %.4f median %.4f ms  (%.1f %% of 8 TB/s)  prep %.4f select %.4f  K=%d M=%d D=%d hits=%d crc=%08x"
          % (w, ns, nt * tl, ms[0], med, 100 * algo / (med * 1e-3) / 8e12, sorted(prep)[len(prep) // 2], sorted(sel)[len(sel) // 2], K, M, D, n, crc), flush=True)
    ctx.close()
