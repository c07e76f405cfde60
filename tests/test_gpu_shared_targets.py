"""GPU: several search contexts on ONE staged target set (sp_ctx_share_targets): the tables and the set are shared, not copied -- same bytes as a context that stages
its own, from concurrent host threads too, HBM of one copy, and a context that stages a new set detaches without disturbing the others."""
import ctypes as C
import threading
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free(L):
    f, t = C.c_uint64(), C.c_uint64()
    assert L.sp_device_memory(C.byref(f), C.byref(t)) == 0
    return int(f.value)


def _crc(ctx, nq):
    hp, op, n = ctx.hits_view()
    return n, zlib.crc32(C.string_at(hp, 8 * n), zlib.crc32(C.string_at(op, 4 * (nq + 1))))


def test_contexts_share_one_staged_target_set():
    import bench
    import spacepharer_amd as sa
    L = sa.load_library()
    tres, toff, qres, qoff = bench.make_workload(2000, 30000, 200, 4242)
    nq = len(qoff) - 1
    # four different query batches (rotations of one)
    q = qres.reshape(nq, 10)
    batches = [np.ascontiguousarray(np.roll(q, k * 1000, axis=0)) for k in range(4)]
    warm = sa.SearchContext(); warm.set_targets_raw(tres.tobytes(), bench.u64p(toff), len(toff) - 1); warm.set_queries_raw(batches[0].tobytes(), bench.u64p(qoff), nq); warm.prefilter(); warm.align_hits_raw(); warm.close()
    f0 = _free(L)
    own = sa.SearchContext()
    own.set_targets_raw(tres.tobytes(), bench.u64p(toff), len(toff) - 1)
    f1 = _free(L)
    staged = f0 - f1                                  # tables + target set of one context
    assert staged > 250 << 20                         # (256 MB of similar 3-mers alone)
    ref = []
    for b in batches:
        own.set_queries_raw(b.tobytes(), bench.u64p(qoff), nq)
        own.prefilter()
        ref.append(_crc(own, nq))
    assert ref[0][0] > 1000 and len({r[1] for r in ref}) == 4
    f2 = _free(L)
    sharers = [sa.SearchContext() for _ in range(3)]
    for c in sharers:
        c.share_targets(own)
        assert c.stat("index_entries") == own.stat("index_entries")
    f3 = _free(L)
    assert f2 - f3 < staged // 8, (f2 - f3, staged)   # no second copy of anything staged
    ctxs = [own] + sharers
    got = [None] * 4

    def work(i):
        c = ctxs[i]
        c.set_queries_raw(batches[i].tobytes(), bench.u64p(qoff), nq)
        for _ in range(5):
            c.prefilter()
        got[i] = _crc(c, nq)

    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in th: t.start()
    for t in th: t.join()
    assert got == ref
    # the large footprint derives its bucket table once for all of them (whoever gets there first), concurrently
    def dense(i):
        ctxs[i].set_tuning("match_small", 0)
        ctxs[i].prefilter()
        got[i] = _crc(ctxs[i], nq)
        ctxs[i].set_tuning("match_small", -1)
    th = [threading.Thread(target=dense, args=(i,)) for i in range(4)]
    for t in th: t.start()
    for t in th: t.join()
    assert got == ref
    # a sharer that stages another set detaches; the others keep theirs -- also when the context that staged the set goes away first
    tres2, toff2, _, _ = bench.make_workload(10, 5000, 150, 99)
    sharers[0].set_targets_raw(tres2.tobytes(), bench.u64p(toff2), len(toff2) - 1)
    sharers[0].set_queries_raw(batches[0].tobytes(), bench.u64p(qoff), nq)
    sharers[0].prefilter()
    assert _crc(sharers[0], nq) != ref[0]
    own.close()
    for i in (1, 2):
        sharers[i].prefilter()
        assert _crc(sharers[i], nq) == ref[i + 1]
        aln, cig, na = sharers[i].align_hits_raw()
        assert na == ref[i + 1][0]
    with pytest.raises(sa.SpError):
        other = sa.SearchContext(sensitivity=7.5)
        try:
            other.share_targets(sharers[1])           # another sensitivity = another k-mer threshold: refused
        finally:
            other.close()
    for c in sharers:
        c.close()
    assert abs(_free(L) - f0) < 64 << 20             # everything went back
