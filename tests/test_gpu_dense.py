"""GPU, the dense index: 10 000 spacers against 2 x 10^8 target residues (10^6 ORFs; the survey's estimate of the real
GenBank_phage_2018_09) -- three index entries per k-mer slot, 2 200 index hits per query on average and up to tens of thousands, so
k_match runs its other regime: hits spill from LDS to the per-wave HBM scratch, the exact pre-filter iterates with rotating hashes,
long index lists are walked on the spot.  Parity at that density:
  * the unsplit search must equal, byte for byte, the union of 8 target shards (each 8x sparser: mostly the LDS-only path) merged
    by the selection on the union -- two regimes of the same kernels, one answer;
  * at 5 x 10^7 residues (550 hits per query: the spill path with an index the CPU oracle can still build) a random sample of the
    queries against the oracle, hits and alignments."""
import ctypes as C
import gzip
import json
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hits_np(ctx):
    hits, offs, n = ctx.hits_compact()
    h = np.ctypeslib.as_array(C.cast(hits, C.POINTER(C.c_int32)), shape=(max(1, n), 3))[:n].copy()
    return h, np.ctypeslib.as_array(offs).astype(np.int64).copy(), n


def test_dense_unsplit_equals_union_of_eight_sparse_shards():
    import bench
    import spacepharer_amd as sa
    from spacepharer_amd.dist import decompose_by_residues
    n_t, tl = 1000000, 200
    tres, toff, qres, qoff = bench.make_workload(10000, n_t, tl, 12345)
    nq = len(qoff) - 1
    ctx = sa.SearchContext()
    ctx.set_targets_raw(tres.tobytes(), bench.u64p(toff), n_t)
    ctx.set_queries_raw(qres.tobytes(), bench.u64p(qoff), nq)
    ctx.prefilter()
    h, off, n = _hits_np(ctx)
    assert ctx.stat("db_matches") > 1.2e8 and ctx.stat("db_matches") / nq > 2000       # the dense regime
    assert n > 1.0e6 and np.all(np.diff(off) <= 300)
    ctx.prefilter()                                                                     # idempotent
    h2, off2, n2 = _hits_np(ctx)
    assert n2 == n and np.array_equal(h2, h) and np.array_equal(off2, off)
    ctx.set_tuning("match_small", 1)                                                    # the footprint meant for sparse indices (256 hits in LDS,
    try:                                                                                # 8 192-bucket pre-filter): slower here, same hits
        ctx.prefilter()
    finally:
        ctx.set_tuning("match_small", -1)
    h2, off2, n2 = _hits_np(ctx)
    assert n2 == n and np.array_equal(h2, h) and np.array_equal(off2, off)
    ctx.set_tuning("match_dense2", 1)                                                   # round 6's kernel for this regime (64-byte double buckets, a 5-byte
    try:                                                                                # hit stream read back by LDS-DMA; opt-in: it is slower): same hits
        ctx.prefilter()
        assert ctx.stat("db_matches") > 1.2e8
    finally:
        ctx.set_tuning("match_dense2", 0)
    h2, off2, n2 = _hits_np(ctx)
    assert n2 == n and np.array_equal(h2, h) and np.array_equal(off2, off)
    aln, cig, na = ctx.align_hits_raw()
    rec = bench.record_fields(C.addressof(aln), n)
    ctx.close()
    lens = np.diff(toff).astype(np.int64) + 2
    parts, keep = [], None
    for r in range(8):
        start, count = decompose_by_residues(lens, r, 8)
        c = sa.SearchContext()
        c.set_targets_shard_raw(tres.tobytes(), bench.u64p(toff), n_t, start, start + count)
        c.set_queries_raw(qres.tobytes(), bench.u64p(qoff), nq)
        c.prefilter_candidates()
        parts.append(c.candidates().copy())
        if r == 7:
            keep = c
        else:
            c.close()
    keep.merge_candidates(np.concatenate(parts, axis=0))
    hs, offs, ns = _hits_np(keep)
    assert ns == n and np.array_equal(offs, off) and np.array_equal(hs, h)
    aln2, cig2, _ = keep.align_hits_raw()
    assert bench.record_fields(C.addressof(aln2), n) == rec
    keep.close()


def test_spill_regime_sample_against_oracle():
    import bench
    import spacepharer_amd as sa
    from oracle_lib import Alp, Oracle
    n_t, tl = 250000, 200
    tres, toff, qres, qoff = bench.make_workload(4000, n_t, tl, 4242)
    nq = len(qoff) - 1
    ctx = sa.SearchContext()
    ctx.set_targets_raw(tres.tobytes(), bench.u64p(toff), n_t)
    ctx.set_queries_raw(qres.tobytes(), bench.u64p(qoff), nq)
    ctx.prefilter()
    h, off, n = _hits_np(ctx)
    assert ctx.stat("db_matches") / nq > 512                      # on average more hits per query than the LDS buffer holds
    aln, cig, na = ctx.align_hits_raw()
    o = Oracle()
    o.set_targets([tres[i * tl:(i + 1) * tl].tobytes() for i in range(n_t)])
    g = json.load(gzip.open(os.path.join(ROOT, "tests/golden/spacer.json.gz")))["ref"]
    o.set_alp(Alp.from_ref(g["alpGapped"]), Alp.from_ref(g["alpGapless"]))
    rnd = random.Random(3)
    checked = 0
    for q in rnd.sample(range(nq), 150):
        qs = qres[q * 10:(q + 1) * 10].tobytes().decode()
        exp = o.prefilter_list(qs)
        got = [[int(x[0]), int(x[1]), ((int(x[2]) & 0xFFFF) ^ 0x8000) - 0x8000] for x in h[off[q]:off[q + 1]]]
        assert got == exp, q
        for k, e in enumerate(exp):
            r = aln[off[q] + k]
            a = o.align(qs, e[0])
            assert (r.raw_score, r.bit_score, r.q_end, r.db_end, r.passed) == (a.rawScore, a.score, a.qEndPos, a.dbEndPos, a.passed)
            if a.passed:
                assert (r.q_start, r.db_start, r.evalue) == (a.qStartPos, a.dbStartPos, a.eval)
                assert C.string_at(C.addressof(cig) + r.cigar_offset).decode() == a.backtrace.decode()
            checked += 1
    assert checked > 100
    o.close()
    ctx.close()


def test_real_density_sample_against_oracle():
    """2 x 10^8 residues -- the density three of BASELINE.json's five configs name -- against the CPU oracle itself (not against the GPU's own shards):
    600 sampled peptides of the bench's dense workload, hit lists and alignments field by field.  (The oracle indexes 10^6 targets in about a minute on
    one core; bench.py's dense leg makes the same comparison against the REAL reference classes, `dense.cpu_baseline.results_equal_gpu`.)"""
    import bench
    import spacepharer_amd as sa
    from oracle_lib import Alp, Oracle
    n_t, tl = 1000000, 200
    tres, toff, qres, qoff = bench.make_workload(10000, n_t, tl, 12345)
    nq = len(qoff) - 1
    ctx = sa.SearchContext()
    ctx.set_targets_raw(tres.tobytes(), bench.u64p(toff), n_t)
    ctx.set_queries_raw(qres.tobytes(), bench.u64p(qoff), nq)
    ctx.prefilter()
    h, off, n = _hits_np(ctx)
    assert ctx.stat("db_matches") / nq > 2000
    aln, cig, na = ctx.align_hits_raw()
    o = Oracle()
    o.set_targets([tres[i * tl:(i + 1) * tl].tobytes() for i in range(n_t)])
    g = json.load(gzip.open(os.path.join(ROOT, "tests/golden/spacer.json.gz")))["ref"]
    o.set_alp(Alp.from_ref(g["alpGapped"]), Alp.from_ref(g["alpGapless"]))
    rnd = random.Random(5)
    checked = accepted = 0
    for q in rnd.sample(range(nq), 600):
        qs = qres[q * 10:(q + 1) * 10].tobytes().decode()
        exp = o.prefilter_list(qs)
        got = [[int(x[0]), int(x[1]), ((int(x[2]) & 0xFFFF) ^ 0x8000) - 0x8000] for x in h[off[q]:off[q + 1]]]
        assert got == exp, q
        for k, e in enumerate(exp):
            r = aln[off[q] + k]
            a = o.align(qs, e[0])
            assert (r.raw_score, r.bit_score, r.q_end, r.db_end, r.passed) == (a.rawScore, a.score, a.qEndPos, a.dbEndPos, a.passed)
            if a.passed:
                assert (r.q_start, r.db_start, r.evalue) == (a.qStartPos, a.dbStartPos, a.eval)
                assert C.string_at(C.addressof(cig) + r.cigar_offset).decode() == a.backtrace.decode()
                accepted += 1
            checked += 1
    assert checked > 5000 and accepted > 500, (checked, accepted)
    o.close()
    ctx.close()
