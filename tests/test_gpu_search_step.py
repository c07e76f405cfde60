"""GPU: the device-driven search step (sp_ctx_search_begin / _end: prefilter + align of every hit + delivery queued without a host round trip,
two steps in flight) must deliver the bytes of the synchronous path (sp_ctx_prefilter + sp_ctx_align_hits) -- hit list, offsets,
sp_alignment records, backtrace pool -- also when a step meets a capacity and is repeated inside _end."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


class Bufs:
    def __init__(self, L, nq, cap):
        self.L, self.nq, self.cap, self.cig_cap = L, nq, cap, 64 * cap + 4096
        self.hits, self.off = L.sp_host_alloc(8 * cap), L.sp_host_alloc(4 * (nq + 1))
        self.rec, self.cig = L.sp_host_alloc(64 * cap), L.sp_host_alloc(self.cig_cap)

    def begin(self, ctx):
        ctx.search_begin(self.hits, self.off, self.cap, self.rec, self.cig, self.cig_cap)

    def result(self, n, used):
        import bench
        return (C.string_at(self.off, 4 * (self.nq + 1)), C.string_at(self.hits, 8 * n), bench.record_fields(self.rec, n) if n else b"", C.string_at(self.cig, used))

    def free(self):
        for p in (self.hits, self.off, self.rec, self.cig):
            self.L.sp_host_free(p)


def sync_result(ctx, nq):
    """the same four byte strings through sp_ctx_prefilter + sp_ctx_align_hits"""
    import bench
    ctx.prefilter()
    hp, op, n = ctx.hits_view()
    off = C.string_at(op, 4 * (nq + 1))
    hits = C.string_at(hp, 8 * n)
    aln, cig, na = ctx.align_hits_raw()
    assert na == n
    r = np.frombuffer(C.string_at(C.addressof(aln), 64 * n), dtype=np.uint32).reshape(n, 16) if n else np.zeros((0, 16), np.uint32)
    used = int((r[:, 13].astype(np.int64) + r[:, 12] + 1)[r[:, 12] > 0].max()) if n and (r[:, 12] > 0).any() else 0      # cigar_offset + aln_len + 1 (words 13, 12 of sp_alignment)
    return (off, hits, bench.record_fields(C.addressof(aln), n) if n else b"", C.string_at(C.addressof(cig), used)), n


@pytest.mark.parametrize("case", ["spacer", "small", "family", "longq"])
def test_steps_equal_the_synchronous_path(case):
    import spacepharer_amd as sa
    L = sa.load_library()
    g = load_golden(case)
    T, Q = g["targets"], g["queries"]
    kw = {"max_seqs": 20} if case == "family" else {}
    ctx = sa.SearchContext(**kw)
    ctx.set_targets(T)
    ctx.set_queries(Q)
    ref, n = sync_result(ctx, len(Q))
    assert n > 0
    bufs = [Bufs(L, len(Q), n + 64) for _ in range(3)]
    # one step at a time
    bufs[0].begin(ctx)
    assert bufs[0].result(*ctx.search_end()) == ref
    # two in flight, several rounds, every step into its own buffers
    order = []
    bufs[0].begin(ctx)
    for k in range(1, 7):
        bufs[k % 3].begin(ctx)
        order.append(bufs[(k - 1) % 3].result(*ctx.search_end()))
    order.append(bufs[6 % 3].result(*ctx.search_end()))
    assert all(x == ref for x in order)
    repeated = ctx.stat("steps_repeated")
    if case == "longq":      # long queries: some tracebacks need the host-driven path -> every step is repeated inside search_end
        assert repeated == 8
    else:
        assert repeated == 0
    # a step sized too small is clipped on the device, flagged, repeated -- and the context remembers the size
    ctx.set_tuning("step_capacity", max(1, n // 3))
    bufs[0].begin(ctx)
    bufs[1].begin(ctx)
    assert bufs[0].result(*ctx.search_end()) == ref
    assert bufs[1].result(*ctx.search_end()) == ref
    assert ctx.stat("steps_repeated") >= repeated + 1
    before = ctx.stat("steps_repeated")
    bufs[2].begin(ctx)
    assert bufs[2].result(*ctx.search_end()) == ref
    if case != "longq":
        assert ctx.stat("steps_repeated") == before
    # the synchronous entry points refuse to run underneath a step in flight, and work again afterwards
    bufs[0].begin(ctx)
    with pytest.raises(sa.SpError):
        ctx.prefilter()
    ctx.search_end()
    again, n2 = sync_result(ctx, len(Q))
    assert again == ref and n2 == n
    # buffers smaller than the pass: a clear error from search_end, nothing written past them
    small = Bufs(L, len(Q), max(1, n // 2))
    small.begin(ctx)
    with pytest.raises(sa.SpError):
        ctx.search_end()
    small.free()
    for b in bufs:
        b.free()
    ctx.close()


def test_a_small_batch_with_a_generous_caller_buffer():
    """a few queries, a small --max-seqs and room for many more hits than the selection can ever deliver: the step's capacity is bounded by queries x max-seqs (what the
    device hit buffer holds), not by the caller's buffers -- the copies stay inside the device allocation and the results are those of the synchronous path; the other
    synchronous entry points refuse to run underneath a step as well"""
    import spacepharer_amd as sa
    L = sa.load_library()
    g = load_golden("family")
    T, Q = g["targets"], g["queries"][:8]
    ctx = sa.SearchContext(max_seqs=3)
    ctx.set_targets(T)
    ctx.set_queries(Q)
    ref, n = sync_result(ctx, len(Q))
    assert 0 < n <= 24
    big = [Bufs(L, len(Q), 100000) for _ in range(2)]
    for rounds in range(3):
        big[0].begin(ctx)
        big[1].begin(ctx)
        assert big[0].result(*ctx.search_end()) == ref
        assert big[1].result(*ctx.search_end()) == ref
    assert ctx.stat("steps_repeated") == 0
    big[0].begin(ctx)
    with pytest.raises(sa.SpError):
        ctx.rescore([(0, 0, 0)], 1)
    with pytest.raises(sa.SpError):
        ctx.prefilter_candidates()
    assert big[0].result(*ctx.search_end()) == ref
    for b in big:
        b.free()
    ctx.close()
