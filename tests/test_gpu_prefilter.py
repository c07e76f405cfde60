"""GPU parity: the HIP prefilter (through the C ABI) against the reference-generated golden vectors and against the
CPU oracle on fresh seeded inputs.  Bit-exact: hit sets, scores, diagonals, order."""
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

CASES = ["small", "spacer", "family", "longq", "family16", "longt", "sens75", "sens1"]


def _ctx(r):
    import spacepharer_amd as sa
    return sa.SearchContext(sensitivity=r["sens"], max_seqs=r["maxSeqs"], bin_size=r["binSize"])


@pytest.mark.parametrize("name", CASES)
def test_prefilter_matches_reference_golden(name):
    g = load_golden(name)
    r = g["ref"]
    ctx = _ctx(r)
    ctx.set_targets(g["targets"])
    assert int(ctx.stat("index_entries")) == r["index"]["entries"]
    assert int(ctx.stat("masked_residues")) == r["maskedResidues"]
    assert int(ctx.stat("kmer_thr")) == r["kmerThr"]
    # the target store is built on the device (tantan masking, index construction): masked sequences and the digest over
    # the (k-mer, seqId, pos) triples in table order must be the reference's (tantan::maskSequences, IndexTable)
    assert ctx.masked_targets() == "".join(r["masked"])
    n_entries, digest = ctx.index_digest()
    assert n_entries == r["index"]["entries"] and "%016x" % digest == r["index"]["digest"]
    assert int(ctx.stat("index_slots")) == r["index"]["nonEmpty"]
    if name == "small":      # the similar-3-mer table is sorted on the device: same bytes as ExtendedSubstitutionMatrix::calcScoreMatrix
        assert "%016x" % ctx.ext3_digest() == r["ext3"]["digest"]
    ctx.set_queries(g["queries"])
    ctx.prefilter()
    got = ctx.hits()
    for q in r["queries"]:
        assert [list(h) for h in got[q["id"]]] == q["hits"], (name, q["id"])
    # double-hit diagonals found over all queries (QueryMatcher::match output sizes)
    assert ctx.stat("diagonals") == sum(len(q["diag"]) for q in r["queries"])
    # k_match has two footprints and picks by the index' density: the other one must find the same
    # ... and for short queries against a sparse index there are two kernels: k_match_grp (four queries per pass of a wave, the default where its
    # preconditions hold) and k_match itself (match_group 0)
    try:
        for small, group in ((0, 1), (1, 1), (1, 0)):
            ctx.set_tuning("match_small", small)
            ctx.set_tuning("match_group", group)
            ctx.prefilter()
            got = ctx.hits()
            for q in r["queries"]:
                assert [list(h) for h in got[q["id"]]] == q["hits"], (name, q["id"], small, group)
            assert ctx.stat("diagonals") == sum(len(q["diag"]) for q in r["queries"])
    finally:
        ctx.set_tuning("match_small", -1)
        ctx.set_tuning("match_group", 1)
    ctx.close()


def test_prefilter_matches_oracle_on_fresh_inputs():
    from golden.make_golden import make_case
    from oracle_lib import Oracle
    import spacepharer_amd as sa
    T, Q = make_case(seed=424242, n_targets=3000, n_queries=1500, tmin=30, tmax=350, qmin=8, qmax=14, families=3)
    Q += ["", "A", "ACDEFGH", "XXXXXXXXXX", "ACDEFGHIKLMNPQRSTVWY" * 3]     # empty / too short / all-X / long
    o = Oracle(sens=5.7, max_seqs=300)
    o.set_targets(T)
    ctx = sa.SearchContext()
    ctx.set_targets(T)
    ctx.set_queries(Q)
    ctx.prefilter()
    got = ctx.hits()
    nhits = 0
    for i, q in enumerate(Q):
        exp = o.prefilter_list(q)
        assert [list(h) for h in got[i]] == exp, i
        nhits += len(exp)
    assert nhits > 100
    ctx.close()
    o.close()


def test_selection_paths_by_number_of_diagonals_against_oracle():
    """k_select takes a query by its number of double-hit diagonals: <= 4 inside a lane, <= 64 one record per lane, <= 256 four per lane + one LDS sort
    (selectMedium, round 5), the general path beyond or at --max-seqs.  Families of near-identical targets (and exact copies: several diagonals and equal scores
    per target) put queries into every class; each one must equal the oracle's list -- same hits, same scores, same order."""
    import random
    from golden.make_golden import make_case, mutate, rand_seq
    from oracle_lib import Oracle
    import spacepharer_amd as sa
    T, Q = make_case(seed=777001, n_targets=1200, n_queries=60, tmin=40, tmax=200, qmin=9, qmax=30, families=0)
    rng = random.Random(5150)
    planted = []
    for size in (3, 30, 70, 130, 200, 222, 232, 250, 290, 340):              # family sizes -> diagonals per query of that family
        base = rand_seq(rng, 90)
        members = [mutate(rng, base, rng.randint(0, 2)) + rand_seq(rng, rng.randint(0, 15)) for _ in range(size)]
        members += [members[i % len(members)] for i in range(size // 10)]          # exact copies
        members += [m[:50] + m[20:] for m in members[:size // 8]]                  # a repeated stretch: two diagonals on one target
        T += members
        for _ in range(4):
            p = rng.randrange(0, 60)
            planted.append(mutate(rng, base[p:p + rng.randint(10, 24)], rng.choice([0, 0, 1])))
    Q = planted + Q
    for max_seqs in (300, 100):
        o = Oracle(sens=5.7, max_seqs=max_seqs)
        o.set_targets(T)
        ctx = sa.SearchContext(max_seqs=max_seqs)
        ctx.set_targets(T)
        ctx.set_queries(Q)
        ctx.prefilter()
        got = ctx.hits()
        sizes = []
        for i, q in enumerate(Q):
            exp = o.prefilter_list(q)
            assert [list(h) for h in got[i]] == exp, (max_seqs, i, len(exp))
            sizes.append(len(exp))
        if max_seqs == 300:
            assert any(s <= 4 for s in sizes) and any(4 < s <= 64 for s in sizes) and any(64 < s <= 256 for s in sizes) and any(256 < s < 300 for s in sizes) and any(s == 300 for s in sizes), sorted(sizes)
        ctx.close()
        o.close()


def test_degenerate_targets_and_queries_against_oracle():
    """the device-side target preparation and the search on empty / shorter-than-a-window / all-X / repeat-rich targets"""
    from golden.make_golden import make_case
    from oracle_lib import Oracle
    import spacepharer_amd as sa
    T, Q = make_case(seed=808, n_targets=400, n_queries=200, tmin=20, tmax=120, qmin=8, qmax=12, families=2)
    odd = ["", "A", "ACDEFGH", "ACDEFGHI", "XXXXXXXXXXXXXXXX", "ACACACACACACACACACACACACACACACACACACAC", "KKKKKKKKKKKKKKKKKKKKKKKKKKKK",
           "MKVLAAGIXXXXXXXXGGSTRRW", "W" * 70, "ACDEFGHIKLMNPQRSTVWY" * 40]
    T = odd[:5] + T[:200] + odd[5:] + T[200:] + [""]
    Q = Q + ["ACDEFGHIK", "ACACACACAC", "KKKKKKKKKK", "WWWWWWWWWW", "ACDEFGHIKLMNPQRSTVWY"]
    o = Oracle(sens=5.7, max_seqs=300)
    o.set_targets(T)
    ctx = sa.SearchContext()
    ctx.set_targets(T)
    from oracle_lib import lib
    assert int(ctx.stat("index_entries")) == lib().spo_index_entries(o.ctx)
    assert ctx.masked_targets() == "".join(o.masked())           # tantan on the device == the checker's, residue by residue
    ctx.set_queries(Q)
    ctx.prefilter()
    got = ctx.hits()
    total = 0
    for i, q in enumerate(Q):
        exp = o.prefilter_list(q)
        assert [list(h) for h in got[i]] == exp, (i, q)
        total += len(exp)
    assert total > 50
    ctx.close()
    o.close()


def test_empty_batches_and_no_hit_batches():
    """zero queries, queries that hit nothing, an align call without hits: nothing to do must not be an error"""
    import spacepharer_amd as sa
    from golden.make_golden import make_case
    T, Q = make_case(seed=5150, n_targets=100, n_queries=10, tmin=30, tmax=80, qmin=9, qmax=12)
    ctx = sa.SearchContext()
    ctx.set_targets(T)
    ctx.set_queries([])
    ctx.prefilter()
    assert ctx.hits() == [] and ctx.num_hits() == 0
    ctx.set_queries(["WWWWWWWWWW", "CCCCCCCCCC", "", "AC"])
    ctx.prefilter()
    assert ctx.hits() == [[], [], [], []]
    aln, cig, n = ctx.align_hits_raw()
    assert n == 0
    assert ctx.align([]) == []
    ctx.set_queries(Q)              # and the context still works afterwards
    ctx.prefilter()
    assert ctx.num_hits() > 0
    aln, cig, n = ctx.align_hits_raw()
    assert n == ctx.num_hits()
    ctx.close()


def test_the_order_the_queries_are_matched_in_never_changes_a_result():
    """k_match starts with the queries k_prep8 weighs as heavy -- by the calibrated estimate (default), by exact counts (match_weigh_exact) or not at
    all (match_heavy_threshold 0) -- and deals the rest in chunks: the hit list must be the same bytes whichever way"""
    import ctypes as C
    import zlib
    import numpy as np
    import bench
    import spacepharer_amd as sa
    tres, toff, qres, qoff = bench.make_workload(1500, 6000, 200, 777)
    crcs = {}
    for name, knobs in (("estimate", {}), ("exact", {"match_weigh_exact": 1}), ("none", {"match_heavy_threshold": 0}), ("all heavy", {"match_heavy_threshold": 1}),
                        ("chunks of 1", {"match_chunk": 1}), ("chunks of 7, all dynamic", {"match_chunk": 7, "match_static": 0}),
                        ("one query at a time", {"match_group": 0}), ("one query at a time, exact weights", {"match_group": 0, "match_weigh_exact": 1}),
                        ("one query at a time, chunks of 5", {"match_group": 0, "match_chunk": 5, "match_static": 2})):
        ctx = sa.SearchContext()
        ctx.set_targets_raw(tres.tobytes(), bench.u64p(toff), len(toff) - 1)
        ctx.set_queries_raw(qres.tobytes(), bench.u64p(qoff), len(qoff) - 1)
        for k, v in knobs.items():
            ctx.set_tuning(k, v)
        ctx.prefilter()
        hp, op, n = ctx.hits_view()
        crcs[name] = (n, zlib.crc32(C.string_at(hp, 8 * n), zlib.crc32(C.string_at(op, 4 * len(qoff)))))
        ctx.close()
    assert crcs["estimate"][0] > 1000
    assert len(set(crcs.values())) == 1, crcs


@pytest.mark.perf
def test_the_matching_kernel_times_itself():
    """sp_ctx_stat("ms_match_active"): k_match from its first wave's start to its last wave's end by the device wall clock -- inside the HIP-event time of the
    same launch (the events also hold launch latency), and not much shorter when the context has the GPU to itself"""
    import bench
    import spacepharer_amd as sa
    tres, toff, qres, qoff = bench.make_workload(3000, 20000, 200, 31)
    ctx = sa.SearchContext()
    ctx.set_targets_raw(tres.tobytes(), bench.u64p(toff), len(toff) - 1)
    ctx.set_queries_raw(qres.tobytes(), bench.u64p(qoff), len(qoff) - 1)
    ctx.set_timing(1)
    for _ in range(3):
        ctx.prefilter()
    ev, act = ctx.stat("ms_match"), ctx.stat("ms_match_active")
    assert 0.02 < act <= ev * 1.02 and act > 0.6 * ev, (ev, act)
    ctx.close()


def test_a_destroyed_context_gives_its_device_memory_back():
    """contexts come and go in a long-lived process (the resident server, this test session): HBM in use must return to where it was"""
    import ctypes as C
    import spacepharer_amd as sa
    L = sa.load_library()
    g = load_golden("small")

    def free_bytes():
        f, t = C.c_uint64(), C.c_uint64()
        assert L.sp_device_memory(C.byref(f), C.byref(t)) == 0
        return int(f.value)

    def cycle():
        ctx = sa.SearchContext()
        ctx.set_targets(g["targets"])
        ctx.set_queries(g["queries"])
        ctx.prefilter()
        ctx.align_hits_raw()
        ctx.set_tuning("match_small", 0)      # the large footprint and its 2 GB direct table
        ctx.prefilter()
        ctx.close()

    cycle()                                  # (the first context also brings up what stays for the process: code objects, the runtime's pools)
    before = free_bytes()
    for _ in range(5):
        cycle()
    assert before - free_bytes() < (1 << 30), (before, free_bytes())
