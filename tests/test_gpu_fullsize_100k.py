"""GPU, BASELINE.json configs[3] at its full size: 100 000 spacers = ONE batch of 600 000 peptides against 1e7 target residues
(heavy-query list off, different work distribution than the 10 000-spacer batch).  (1) the unsplit search against the oracle on
a random sample of the queries, prefilter + alignments; (2) the target-sharded search with 8 emulated shards (8 contexts on the
one GPU, candidate lists concatenated as the all-gather delivers them) must reproduce the unsplit bytes -- hit list, offsets,
alignment records and backtraces."""
import ctypes as C
import gzip
import json
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SPACERS, N_TARGETS, TLEN = 100000, 50000, 200
REC = np.dtype([("db_key", "<u4"), ("bit_score", "<i4"), ("raw_score", "<i4"), ("seq_id", "<f4"), ("evalue", "<f8"), ("q_start", "<i4"), ("q_end", "<i4"),
                ("q_len", "<u4"), ("db_start", "<i4"), ("db_end", "<i4"), ("db_len", "<u4"), ("aln_len", "<u4"), ("cigar_offset", "<u4"), ("passed", "<i4"), ("pad", "<u4")])      # sp_alignment (64 bytes with its tail padding)
assert REC.itemsize == 64


def _hits_np(ctx):
    hits, offs, n = ctx.hits_compact()
    h = np.ctypeslib.as_array(C.cast(hits, C.POINTER(C.c_int32)), shape=(max(1, n), 3))[:n].copy()
    return h, np.ctypeslib.as_array(offs).astype(np.int64).copy(), n


@pytest.fixture(scope="module")
def big():
    import bench
    import spacepharer_amd as sa
    tres, toff, qres, qoff = bench.make_workload(N_SPACERS, N_TARGETS, TLEN, 12345)
    nq = len(qoff) - 1
    ctx = sa.SearchContext()
    ctx.set_targets_raw(tres.tobytes(), bench.u64p(toff), N_TARGETS)
    ctx.set_queries_raw(qres.tobytes(), bench.u64p(qoff), nq)
    ctx.prefilter()
    h, off, n = _hits_np(ctx)
    aln, cig, na = ctx.align_hits_raw()
    assert na == n
    rec = bench.record_fields(C.addressof(aln), n)
    r = np.frombuffer(C.string_at(C.addressof(aln), 64 * n), dtype=REC)
    used = int((r["cigar_offset"].astype(np.int64) + r["aln_len"] + 1)[r["aln_len"] > 0].max())      # bytes of the backtrace pool in use
    pool = C.string_at(C.addressof(cig), used)
    yield dict(ctx=ctx, tres=tres, toff=toff, qres=qres, qoff=qoff, nq=nq, h=h, off=off, n=n, aln=aln, cig=cig, rec=rec, pool=pool)
    ctx.close()


def test_one_batch_of_600k_peptides_shapes(big):
    assert big["nq"] == 600000 and big["n"] > 600000
    off = big["off"]
    assert off[0] == 0 and off[-1] == big["n"] and np.all(np.diff(off) >= 0) and np.all(np.diff(off) <= 300)


def test_sample_against_oracle(big):
    from oracle_lib import Alp, Oracle
    tres, qres, h, off, nq, aln, cig = (big[k] for k in ("tres", "qres", "h", "off", "nq", "aln", "cig"))
    o = Oracle()
    o.set_targets([tres[i * TLEN:(i + 1) * TLEN].tobytes() for i in range(N_TARGETS)])
    g = json.load(gzip.open(os.path.join(ROOT, "tests/golden/spacer.json.gz")))["ref"]
    o.set_alp(Alp.from_ref(g["alpGapped"]), Alp.from_ref(g["alpGapless"]))
    rnd = random.Random(11)
    checked = 0
    for q in rnd.sample(range(nq), 400):
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


def test_eight_emulated_shards_reproduce_the_unsplit_bytes(big):
    import bench
    import spacepharer_amd as sa
    from spacepharer_amd.dist import decompose_by_residues
    tres, toff, qres, qoff, nq = (big[k] for k in ("tres", "toff", "qres", "qoff", "nq"))
    lens = np.diff(toff).astype(np.int64) + 2
    parts, keep = [], None
    for r in range(8):
        start, count = decompose_by_residues(lens, r, 8)
        c = sa.SearchContext()
        c.set_targets_shard_raw(tres.tobytes(), bench.u64p(toff), N_TARGETS, start, start + count)
        c.set_queries_raw(qres.tobytes(), bench.u64p(qoff), nq)
        n = c.prefilter_candidates()
        cd = c.candidates()
        assert cd.shape == (n, 16) and n > 0
        sid = cd.view(np.uint32).reshape(-1, 4)[:, 1]
        assert sid.min() >= start and sid.max() < start + count
        parts.append(cd.copy())
        if r == 7:
            keep = c
        else:
            c.close()
    union = np.concatenate(parts, axis=0)
    union = union[np.random.default_rng(3).permutation(len(union))]
    keep.merge_candidates(union)
    h, off, n = _hits_np(keep)
    assert n == big["n"] and np.array_equal(off, big["off"]) and np.array_equal(h, big["h"])
    aln, cig, na = keep.align_hits_raw()
    assert na == n and bench.record_fields(C.addressof(aln), n) == big["rec"]
    assert C.string_at(C.addressof(cig), len(big["pool"])) == big["pool"]
    keep.close()
