"""GPU parity of the Smith-Waterman (align) and diagonal-rescoring kernels against the reference-generated golden
vectors: integer scores, end/start positions, E-values (bit-identical doubles), acceptance and backtrace strings."""
import os

import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

CASES = ["small", "spacer", "family", "longq", "family16", "longt", "sens75", "sens1"]


@pytest.mark.parametrize("name", CASES)
def test_align_matches_reference_golden(name):
    import spacepharer_amd as sa
    g = load_golden(name)
    r = g["ref"]
    ctx = sa.SearchContext(sensitivity=r["sens"], max_seqs=r["maxSeqs"])
    ctx.set_targets(g["targets"])
    ctx.set_queries(g["queries"])
    pairs, exp = [], []
    for a in r["aln"]:
        for raw in a["raw"]:
            pairs.append((a["id"], raw[0]))
            exp.append(raw)
    assert pairs
    got = ctx.align(pairs)
    if name == "longq":      # a long gapped alignment outgrows its share of the traceback pool: the host-driven retry ran
        assert ctx.stat("align_retries") >= 1
    gapped = word = 0
    for (res, cig), raw in zip(got, exp):
        dbKey, bits, qs, qe, ts, te, alnLen, ev, ok, bt = raw
        assert res.db_key == dbKey
        assert res.bit_score == bits, raw
        assert (res.q_end, res.db_end) == (qe, te), raw
        assert res.passed == ok, raw
        if ok:
            assert (res.q_start, res.db_start, res.aln_len) == (qs, ts, alnLen), raw
            assert res.evalue == float.fromhex(ev)
            assert cig == bt, raw
            gapped += ("I" in bt) or ("D" in bt)
        word += res.raw_score >= 255
    if name in ("family", "longq"):
        assert gapped > 0
    if name == "longq":
        assert word > 0          # exercises the 16-bit pass
    ctx.close()


@pytest.mark.parametrize("name", CASES)
def test_rescorediagonal_matches_reference_golden(name):
    import spacepharer_amd as sa
    g = load_golden(name)
    r = g["ref"]
    ctx = sa.SearchContext(sensitivity=r["sens"], max_seqs=r["maxSeqs"])
    ctx.set_targets(g["targets"])
    ctx.set_queries(g["queries"])
    triples, exp = [], []
    for a, qh in zip(r["aln"], r["queries"]):
        for raw, hit in zip(a["rescore"], qh["hits"]):
            triples.append((a["id"], raw[0], hit[2]))
            exp.append(raw)
    m1 = ctx.rescore(triples, 1)
    m2 = ctx.rescore(triples, 2)
    for a, b, raw in zip(m1, m2, exp):
        dbKey, s1, s2, st, en, dlen, dist, diag, ev, bits = raw
        assert a.score == s1
        assert (b.score, b.start_pos, b.end_pos, b.diagonal_len, b.dist_to_diagonal, b.diagonal) == (s2, st, en, dlen, dist, diag)
        assert b.evalue == float.fromhex(ev)
        assert b.bit_score == bits
    # --rescore-mode 0 (the module's default: equal letters), 3 (global) and 4 (window quality): letters compared as they are, '*' at the ends skipped
    exp3 = [three for a in r["aln"] for three in a["rescoreModes"]]
    assert len(exp3) == len(triples)
    for k, mode in enumerate((0, 3, 4)):
        got = ctx.rescore(triples, mode)
        for x, three in zip(got, exp3):
            assert [x.score, x.start_pos, x.end_pos, x.diagonal_len, x.dist_to_diagonal, x.diagonal] == three[k], (mode, three)
    ctx.close()


def test_align_matches_oracle_on_fresh_inputs():
    from golden.make_golden import make_case
    from oracle_lib import Alp, Oracle
    import spacepharer_amd as sa
    g = load_golden("spacer")["ref"]
    T, Q = make_case(seed=99, n_targets=1500, n_queries=600, tmin=30, tmax=350, qmin=9, qmax=31, families=2)
    o = Oracle()
    o.set_targets(T)
    o.set_alp(Alp.from_ref(g["alpGapped"]), Alp.from_ref(g["alpGapless"]))
    ctx = sa.SearchContext()
    ctx.set_targets(T)
    ctx.set_queries(Q)
    ctx.prefilter()
    hits = ctx.hits()
    pairs = [(qi, h[0]) for qi, hs in enumerate(hits) for h in hs]
    assert len(pairs) > 200
    got = ctx.align(pairs)
    for (qi, ti), (res, cig) in zip(pairs, got):
        e = o.align(Q[qi], ti)
        assert (res.raw_score, res.bit_score, res.q_end, res.db_end, res.passed) == (e.rawScore, e.score, e.qEndPos, e.dbEndPos, e.passed)
        if e.passed:
            assert (res.q_start, res.db_start, res.aln_len, res.evalue, cig) == (e.qStartPos, e.dbStartPos, e.alnLength, e.eval, e.backtrace.decode())
            assert res.seq_id == e.seqId
    ctx.close()
    o.close()


@pytest.mark.parametrize("gap_open,gap_extend", [(11, 1), (5, 1), (8, 3), (16, 2)])
def test_align_with_other_gap_penalties_against_oracle(gap_open, gap_extend, monkeypatch):
    """cheap gaps make the lazy-F step of the forward pass (its one-row shortcut, its closed-form scan) and the banded traceback do real work; query lengths cover
    every width of the packed kernels (10 / 12 / 16 / 32 rows) and the striped ones beyond.  The Gumbel parameters are the tabulated 16/2 set on both sides
    (SP_ALP_PARAMS here, set_alp there): scores, ends, starts and backtraces are what is compared"""
    import random
    from golden.make_golden import make_case, mutate, rand_seq
    from oracle_lib import Alp, Oracle
    import spacepharer_amd as sa
    g = load_golden("spacer")["ref"]
    names = ["lambda", "K", "a_I", "b_I", "alpha_I", "beta_I", "a_J", "b_J", "alpha_J", "beta_J", "sigma", "tau"]
    monkeypatch.setenv("SP_ALP_PARAMS", ",".join(repr(float.fromhex(g["alpGapped"][n])) for n in names))
    rng = random.Random(1000 * gap_open + gap_extend)
    T, _ = make_case(seed=rng.randrange(1 << 30), n_targets=300, n_queries=1, tmin=60, tmax=400, qmin=9, qmax=12, families=1)
    o = Oracle()
    o.set_targets(T)
    o.set_alp(Alp.from_ref(g["alpGapped"]), Alp.from_ref(g["alpGapless"]))
    for qmax in (10, 12, 16, 32, 60):
        Q, pairs = [], []
        for _ in range(160):
            ti = rng.randrange(len(T))
            L = rng.randint(max(6, qmax - 4), qmax)
            p = rng.randrange(0, len(T[ti]) - L + 1)
            q = mutate(rng, T[ti][p:p + L], rng.choice([0, 1, 2]))
            if L > 8 and rng.random() < 0.6:      # one or two gaps against the target
                for _g in range(rng.choice([1, 1, 2])):
                    c = rng.randrange(2, len(q) - 2)
                    q = q[:c] + rand_seq(rng, rng.randint(1, 3)) + q[c:] if rng.random() < 0.5 else q[:c] + q[c + rng.randint(1, 2):]
            q = q[:qmax]
            if len(q) < 6:
                continue
            Q.append(q)
            pairs += [(len(Q) - 1, ti), (len(Q) - 1, rng.randrange(len(T)))]
        ctx = sa.SearchContext(gap_open=gap_open, gap_extend=gap_extend)
        ctx.set_targets(T)
        ctx.set_queries(Q)
        got = ctx.align(pairs)
        gapped = 0
        for (qi, ti), (res, cig) in zip(pairs, got):
            e = o.align(Q[qi], ti, gap_open=gap_open, gap_extend=gap_extend)
            assert (res.raw_score, res.q_end, res.db_end, res.passed) == (e.rawScore, e.qEndPos, e.dbEndPos, e.passed), (qmax, qi, ti, Q[qi])
            if e.passed:
                assert (res.q_start, res.db_start, res.aln_len, cig) == (e.qStartPos, e.dbStartPos, e.alnLength, e.backtrace.decode()), (qmax, qi, ti, Q[qi])
                gapped += ("I" in cig) or ("D" in cig)
        assert gapped >= (3 if gap_open <= 11 else 0), (qmax, gapped)
        ctx.close()
    o.close()


def test_align_with_tabulated_gumbel_parameters_of_another_gap_pair():
    """gap open 11 / extend 1 without SP_ALP_PARAMS: the E-values come from csrc/sp_alp_table.h (the reference's own fit, tools/gen_alp_table.py); the oracle gets the same
    twelve numbers from the fixture the reference's EvalueComputation wrote (tests/golden/alp_probes.json) -- E-values, bit scores and the pass decision must agree"""
    import json
    from golden.make_golden import make_case
    from oracle_lib import Alp, Oracle
    import spacepharer_amd as sa
    assert "SP_ALP_PARAMS" not in os.environ
    g = load_golden("spacer")["ref"]
    probes = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "alp_probes.json")))["cases"]
    T, Q = make_case(seed=4711, n_targets=400, n_queries=160, tmin=40, tmax=300, qmin=9, qmax=30, families=1)
    o = Oracle()
    o.set_targets(T)
    o.set_alp(Alp(*[float.fromhex(x) for x in probes["VTML40.out 11 1"]["alp"]]), Alp.from_ref(g["alpGapless"]))
    ctx = sa.SearchContext(gap_open=11, gap_extend=1)
    ctx.set_targets(T)
    ctx.set_queries(Q)
    ctx.prefilter()
    pairs = [(qi, h[0]) for qi, hs in enumerate(ctx.hits()) for h in hs]
    assert len(pairs) > 100
    n_pass = 0
    for (qi, ti), (res, cig) in zip(pairs, ctx.align(pairs)):
        e = o.align(Q[qi], ti, gap_open=11, gap_extend=1)
        assert (res.raw_score, res.bit_score, res.q_end, res.db_end, res.passed) == (e.rawScore, e.score, e.qEndPos, e.dbEndPos, e.passed), (qi, ti)
        if e.passed:
            assert (res.q_start, res.db_start, res.aln_len, res.evalue, cig) == (e.qStartPos, e.dbStartPos, e.alnLength, e.eval, e.backtrace.decode())
            n_pass += 1
    assert n_pass > 50
    ctx.close()
    o.close()


def test_align_against_a_target_longer_than_65535_residues():
    """the forward kernel for short queries packs end columns in 16 bits; a target set with a longer target takes the general kernel: ends beyond column 65 535 vs the oracle"""
    import random
    from golden.make_golden import make_case, rand_seq
    from oracle_lib import Alp, Oracle
    import spacepharer_amd as sa
    g = load_golden("spacer")["ref"]
    T, _ = make_case(seed=5, n_targets=40, n_queries=1, tmin=30, tmax=300, qmin=9, qmax=12, families=0)
    rng = random.Random(17)
    T.append(rand_seq(rng, 70000))
    Q, pairs = [], []
    for pos in (0, 1000, 32760, 65530, 65536, 66000, 69990):      # windows of the long target (one substitution each), and the same queries against a short target
        q = list(T[-1][pos:pos + 10])
        q[4] = "W" if q[4] != "W" else "Y"
        Q.append("".join(q))
        pairs += [(len(Q) - 1, len(T) - 1), (len(Q) - 1, 3)]
    o = Oracle()
    o.set_targets(T)
    o.set_alp(Alp.from_ref(g["alpGapped"]), Alp.from_ref(g["alpGapless"]))
    ctx = sa.SearchContext()
    ctx.set_targets(T)
    ctx.set_queries(Q)
    got = ctx.align(pairs)
    far = 0
    for (qi, ti), (res, cig) in zip(pairs, got):
        e = o.align(Q[qi], ti)
        assert (res.raw_score, res.bit_score, res.q_end, res.db_end, res.passed) == (e.rawScore, e.score, e.qEndPos, e.dbEndPos, e.passed), (qi, ti)
        if e.passed:
            assert (res.q_start, res.db_start, res.aln_len, res.evalue, cig) == (e.qStartPos, e.dbStartPos, e.alnLength, e.eval, e.backtrace.decode())
        far += e.dbEndPos > 65535
    assert far >= 3
    ctx.close()
    o.close()


def test_align_hits_equals_align_pairs():
    """the fused prefilter -> align path (pairs built on the device) returns exactly what aligning the downloaded hits does"""
    import spacepharer_amd as sa
    g = load_golden("family")
    r = g["ref"]
    ctx = sa.SearchContext(sensitivity=r["sens"], max_seqs=r["maxSeqs"])
    ctx.set_targets(g["targets"])
    ctx.set_queries(g["queries"])
    ctx.prefilter()
    hits = ctx.hits()
    pairs = [(qi, h[0]) for qi, hs in enumerate(hits) for h in hs]
    a = ctx.align(pairs)
    aln, cig, n = ctx.align_hits_raw()
    assert n == len(pairs)
    import ctypes as C
    for i, (res, c1) in enumerate(a):
        b = aln[i]
        assert (b.db_key, b.bit_score, b.raw_score, b.q_start, b.q_end, b.db_start, b.db_end, b.aln_len, b.passed, b.evalue) == \
               (res.db_key, res.bit_score, res.raw_score, res.q_start, res.q_end, res.db_start, res.db_end, res.aln_len, res.passed, res.evalue)
        if b.aln_len:
            assert C.string_at(C.addressof(cig) + b.cigar_offset).decode() == c1
    ctx.close()


@pytest.mark.parametrize("heavy_lengths,short_range,min_word", [((20, 24, 28, 32), (9, 12), 8), ((13, 14, 15, 16), (9, 12), 0),
                                                              ((11, 12, 12, 12), (9, 11), 0), ((9, 10, 10, 10), (9, 10), 0)])
def test_packed_sw_kernel_byte_overflow_and_ragged_groups(heavy_lengths, short_range, min_word):
    """the packed kernels (4 alignments per wave for queries up to 32 aa, 8 / 10 / 12 per wave when no query is longer than
    16 / 12 / 10: groups of 10 and 12 lanes straddle the DPP rows) on
    awkward material: W/C-rich queries whose perfect matches overflow the 8-bit pass when they are long enough
    (score + bias >= 255 -> 16-bit redo, mixed byte/word lists in the reverse pass), queries of very different lengths
    side by side in one wave, targets shorter than a column chunk, and a pair count that is not a multiple of 8"""
    import random
    from oracle_lib import Alp, Oracle
    import spacepharer_amd as sa
    g = load_golden("spacer")["ref"]
    rnd = random.Random(4242)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    T, Q = [], []
    for i in range(60):                       # heavy queries + targets that contain them (exact, or with an insertion / deletion)
        L = rnd.choice(heavy_lengths)
        q = "".join(rnd.choice("WWWCCYFHM" if rnd.random() < 0.8 else aa) for _ in range(L))
        Q.append(q)
        left, right = ("".join(rnd.choice(aa) for _ in range(rnd.randint(0, 40))) for _ in range(2))
        T.append(left + q + right)
        T.append(left + q[:L // 2] + rnd.choice(aa) + q[L // 2:] + right)
        T.append(q[2:L - 3])                  # shorter than the query
    for i in range(45):                       # short ordinary queries in between
        q = "".join(rnd.choice(aa) for _ in range(rnd.randint(*short_range)))
        Q.insert(rnd.randrange(len(Q)), q)
        T.append("".join(rnd.choice(aa) for _ in range(rnd.randint(5, 31))) + q + "".join(rnd.choice(aa) for _ in range(rnd.randint(0, 9))))
    o = Oracle()
    o.set_targets(T)
    o.set_alp(Alp.from_ref(g["alpGapped"]), Alp.from_ref(g["alpGapless"]))
    ctx = sa.SearchContext()
    ctx.set_targets(T)
    ctx.set_queries(Q)
    pairs = [(qi, ti) for qi in range(len(Q)) for ti in range(len(T)) if rnd.random() < 0.08]
    pairs += [(qi, ti) for qi, q in enumerate(Q) for ti, t in enumerate(T) if q in t]
    while len(pairs) % 8 != 5:
        pairs.append((0, 0))
    got = ctx.align(pairs)
    word = 0
    for (qi, ti), (res, cig) in zip(pairs, got):
        e = o.align(Q[qi], ti)
        assert (res.raw_score, res.bit_score, res.q_end, res.db_end, res.passed) == (e.rawScore, e.score, e.qEndPos, e.dbEndPos, e.passed), (qi, ti)
        if e.passed:
            assert (res.q_start, res.db_start, res.aln_len, res.evalue, cig) == (e.qStartPos, e.dbStartPos, e.alnLength, e.eval, e.backtrace.decode()), (qi, ti)
        word += e.rawScore >= 240
    assert word >= min_word      # the 16-bit redo path really ran (long queries)
    ctx.close()
    o.close()


def test_more_sixteen_bit_pairs_than_the_word_kernels_grid_holds():
    """k_sw_word is a small grid (two blocks per CU, 16 pairs per block and trip) that WALKS the redo list: with more 16-bit pairs than one trip of
    the grid covers (2 x 256 CUs x 16 = 8 192) every block takes several; the same pairs through one call each way must agree, and a sample against the oracle"""
    import random
    from oracle_lib import Alp, Oracle
    import spacepharer_amd as sa
    g = load_golden("spacer")["ref"]
    rnd = random.Random(99)
    Q = ["".join(rnd.choice("WCHMYPNDFQRE") for _ in range(32)) for _ in range(48)]      # perfect matches score 250-290 (diverse: little composition bias): most leave the 8-bit pass
    T = []
    for q in Q:
        for _ in range(5):
            T.append("".join(rnd.choice("ACDEFGHIKLMNPQRSTVWY") for _ in range(rnd.randint(0, 20))) + q + "".join(rnd.choice("ACDEFGHIKLMNPQRSTVWY") for _ in range(rnd.randint(0, 20))))
    ctx = sa.SearchContext()
    ctx.set_targets(T)
    ctx.set_queries(Q)
    own = [(qi, 5 * qi + k) for qi in range(len(Q)) for k in range(5)]      # 240 pairs that overflow the 8-bit pass
    pairs = (own * 60)[:13000]                                                # ... 13 000 of them in one call, plus ordinary ones in between
    pairs += [(rnd.randrange(len(Q)), rnd.randrange(len(T))) for _ in range(1500)]
    rnd.shuffle(pairs)
    got = ctx.align(pairs)
    first = {}
    n_word = 0
    for pr, (res, cig) in zip(pairs, got):
        key = (res.raw_score, res.bit_score, res.q_start, res.q_end, res.db_start, res.db_end, res.aln_len, res.passed, cig)
        assert first.setdefault(pr, key) == key, pr                          # the same pair: the same record wherever it stood in the list
        n_word += res.raw_score >= 255
    assert n_word > 8192 + 256, n_word
    o = Oracle()
    o.set_targets(T)
    o.set_alp(Alp.from_ref(g["alpGapped"]), Alp.from_ref(g["alpGapless"]))
    for (qi, ti) in own[:40] + pairs[:60]:
        e = o.align(Q[qi], ti)
        k = first[(qi, ti)]
        assert (k[0], k[1], k[3], k[5], k[7]) == (e.rawScore, e.score, e.qEndPos, e.dbEndPos, e.passed), (qi, ti)
        if e.passed:
            assert (k[2], k[4], k[6], k[8]) == (e.qStartPos, e.dbStartPos, e.alnLength, e.backtrace.decode()), (qi, ti)
    ctx.close()
    o.close()


def test_align_hits_begin_end_equals_blocking_call():
    """the split call (results still travelling when _begin returns; the next prefilter may be queued meanwhile) delivers
    the same bytes as the blocking one"""
    import ctypes as C
    import spacepharer_amd as sa
    g = load_golden("spacer")
    ctx = sa.SearchContext(sensitivity=g["ref"]["sens"], max_seqs=g["ref"]["maxSeqs"])
    ctx.set_targets(g["targets"])
    ctx.set_queries(g["queries"])
    ctx.prefilter()
    aln, cig, n = ctx.align_hits_raw()
    ref_rec = C.string_at(C.addressof(aln), n * C.sizeof(sa._lib.Alignment))
    ref_cig = bytes(cig[:4096])
    C.memset(C.addressof(aln), 0xEE, n * C.sizeof(sa._lib.Alignment))
    assert ctx.align_hits_begin() == n
    ctx.prefilter()                                   # queued while the records are on their way
    aln2, cig2 = ctx.align_hits_end()
    assert C.string_at(C.addressof(aln2), n * C.sizeof(sa._lib.Alignment)) == ref_rec
    assert bytes(cig2[:4096]) == ref_cig
    assert n > 50
    ctx.close()
