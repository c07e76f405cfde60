"""experiment: one served `search` call (100 000 spacers vs 1e7 residues) as ONE context taking the whole batch against TWO contexts (shared target set, a host thread each)
taking the batch in quarters alternately -- set_queries + prefilter + align-and-format-on-the-device into page-locked buffers; wall clock per variant (the DB write is not in it)"""
import ctypes as C
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spacepharer_amd as sa  # noqa: E402
import numpy as np  # noqa: E402

n = int(os.environ.get("SPACERS", "100000"))
tres, toff, qres, qoff = bench.make_workload(n, 50000, 200, 12345)
nq = len(qoff) - 1
L = sa.load_library()
a = sa.SearchContext(); a.set_targets_raw(tres.tobytes(), bench.u64p(toff), len(toff) - 1)
b = sa.SearchContext(); b.share_targets(a)
qb = qres.tobytes()
L.sp_host_alloc.restype = C.c_void_p


def pinned(nbytes):
    return L.sp_host_alloc(C.c_uint64(nbytes))


def piece(ctx, q0, q1, buf, cap, lens):
    off = (qoff[q0:q1 + 1] - qoff[q0]).astype(np.uint64)
    ctx.set_queries_raw(qb[int(qoff[q0]):int(qoff[q1])], bench.u64p(off), q1 - q0)
    ctx.prefilter()
    used = C.c_uint64()
    rc = L.sp_ctx_align_hits_text(ctx.ctx, C.c_double(0.0), 0, C.c_void_p(buf), C.c_uint64(cap), C.byref(used), C.c_void_p(lens))
    assert rc == 0, (rc, L.sp_last_error())
    return used.value


L.sp_ctx_align_hits_text.argtypes = [C.c_void_p, C.c_double, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p]
cap = 64 << 20
bufs = [pinned(cap) for _ in range(4)]
lens = [pinned(4 * (nq + 16)) for _ in range(4)]
for rep in range(3):
    t0 = time.perf_counter()
    u = piece(a, 0, nq, bufs[0], cap, lens[0])
    t1 = time.perf_counter()
    whole = C.string_at(bufs[0], u)
    cuts = [nq * k // 4 for k in range(5)]
    used = [0] * 4

    def lane(ctx, ks):
        for k in ks:
            used[k] = piece(ctx, cuts[k], cuts[k + 1], bufs[k], cap, lens[k])
    t2 = time.perf_counter()
    th = [threading.Thread(target=lane, args=(a, (0, 2))), threading.Thread(target=lane, args=(b, (1, 3)))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    t3 = time.perf_counter()
    parts = b"".join(C.string_at(bufs[k], used[k]) for k in range(4))
    # one lane, quarters (what the sub-batching alone costs)
    t4 = time.perf_counter()
    lane(a, (0, 1, 2, 3))
    t5 = time.perf_counter()
    print("rep %d: one context, whole batch %.2f ms | two contexts, quarters alternately %.2f ms | one context, quarters %.2f ms | text equal: %s (%d bytes)"
          % (rep, (t1 - t0) * 1e3, (t3 - t2) * 1e3, (t5 - t4) * 1e3, parts == whole, len(whole)))
