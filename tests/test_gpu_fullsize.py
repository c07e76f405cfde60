"""GPU, BASELINE.json's full single-GPU size (10 000 spacers = 60 000 peptides vs 1e7 target residues): size-independent
properties of the whole path + bit-exact comparison with the oracle on a random sample of the queries."""
import ctypes as C
import json
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def full():
    import bench
    import spacepharer_amd as sa
    tres, toff, qres, qoff = bench.make_workload(10000, 50000, 200, 12345)
    ctx = sa.SearchContext()
    ctx.set_targets_raw(tres.tobytes(), bench.u64p(toff), len(toff) - 1)
    nq = len(qoff) - 1
    ctx.set_queries_raw(qres.tobytes(), bench.u64p(qoff), nq)
    ctx.prefilter()
    hits, offs, n = ctx.hits_compact()
    h = np.ctypeslib.as_array(C.cast(hits, C.POINTER(C.c_int32)), shape=(max(1, n), 3))[:n].copy()
    off = np.ctypeslib.as_array(offs).astype(np.int64).copy()
    yield dict(ctx=ctx, tres=tres, qres=qres, nq=nq, h=h, off=off, n=n)
    ctx.close()


def test_hits_view_is_the_compact_list_without_a_copy(full):
    """sp_ctx_hits_view(): the page-locked delivery buffers hold exactly what sp_ctx_get_hits_compact() copies out"""
    ctx, h, off, n = full["ctx"], full["h"], full["off"], full["n"]
    hp, op, cnt = ctx.hits_view()
    assert cnt == n
    v = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_uint16)), shape=(max(1, n), 4))[:n]
    seq = v[:, 0].astype(np.int64) | (v[:, 1].astype(np.int64) << 16)
    assert np.array_equal(seq, h[:, 0]) and np.array_equal(v[:, 2].astype(np.int64), h[:, 1])
    assert np.array_equal(v[:, 3].astype(np.int16), ((h[:, 2] << 16) >> 16).astype(np.int16))
    assert np.array_equal(np.ctypeslib.as_array(op, shape=(len(off),)).astype(np.int64), off)


def test_counters_and_shapes(full):
    ctx, h, off, nq, n = full["ctx"], full["h"], full["off"], full["nq"], full["n"]
    assert off[0] == 0 and off[-1] == n and np.all(np.diff(off) >= 0) and np.all(np.diff(off) <= 300)
    assert int(ctx.stat("hits")) == n
    assert ctx.stat("index_entries") > 9e6          # ~9.6e6 at 1e7 residues
    assert ctx.stat("db_matches") > ctx.stat("diagonals") >= n * 0.9
    seq = h[:, 0]; score = h[:, 1]; diag = (h[:, 2] << 16) >> 16
    assert seq.min() >= 0 and seq.max() < 50000
    assert score.min() >= 15                         # --min-ungapped-score
    assert np.all(np.abs(diag) < 200 + 10)


def test_order_and_uniqueness_per_query(full):
    h, off, nq = full["h"], full["off"], full["nq"]
    for q in range(0, nq, 7):
        s = h[off[q]:off[q + 1]]
        if len(s) < 2:
            continue
        key = list(zip((-s[:, 1]).tolist(), s[:, 0].tolist()))
        assert key == sorted(key)                    # (score desc, seqId asc)
        assert len(set(s[:, 0].tolist())) == len(s)  # one diagonal per target


def test_planted_windows_are_found(full):
    """even-indexed peptides are windows of a target with one substitution: whenever the reference algorithm can see them
    (two scanned windows with a double k-mer hit) the source target must be in the list with the planted diagonal"""
    import bench
    rng = np.random.default_rng(12345)
    # regenerate the planting decisions exactly as make_workload does
    n_targets, tl, nq, ql = 50000, 200, 60000, 10
    rng.choice(20, size=n_targets * tl, p=bench.BG)
    rng.choice(20, size=nq * ql, p=bench.BG)
    tid = rng.integers(0, n_targets, size=nq)
    pos = rng.integers(0, tl - ql + 1, size=nq)
    h, off = full["h"], full["off"]
    found = 0
    for q in range(0, nq, 2):
        s = h[off[q]:off[q + 1]]
        m = s[s[:, 0] == tid[q]]
        if len(m) and (((int(m[0, 2]) << 16) >> 16) & 0xFFFF) == ((-int(pos[q])) & 0xFFFF):
            found += 1
    assert found > 0.5 * (nq // 2)                   # most planted windows survive prefilter (the rest lose their k-mer pair to the substitution)


def test_idempotent(full):
    ctx, h, off = full["ctx"], full["h"], full["off"]
    ctx.prefilter()
    hits, offs, n = ctx.hits_compact()
    h2 = np.ctypeslib.as_array(C.cast(hits, C.POINTER(C.c_int32)), shape=(max(1, n), 3))[:n]
    assert n == full["n"] and np.array_equal(h2, h) and np.array_equal(np.ctypeslib.as_array(offs).astype(np.int64), off)


def test_work_order_and_timing_level_do_not_change_results(full):
    """the scheduler's knobs only reorder the work: heavy-query list off / with another threshold, every HIP-event timing level
    (stages that are not timed report NaN) -- the hit list stays bit-identical"""
    ctx, h, off = full["ctx"], full["h"], full["off"]
    def same():
        ctx.prefilter()
        hits, offs, n = ctx.hits_compact()
        h2 = np.ctypeslib.as_array(C.cast(hits, C.POINTER(C.c_int32)), shape=(max(1, n), 3))[:n]
        return n == full["n"] and np.array_equal(h2, h) and np.array_equal(np.ctypeslib.as_array(offs).astype(np.int64), off)
    try:
        for thr in (0, 1500, 20000):
            ctx.set_tuning("match_heavy_threshold", thr)
            assert same(), thr
    finally:
        ctx.set_tuning("match_heavy_threshold", -1)
    # the list schedule of the grouped kernel (batch sorted by weight, tickets) against heavy-first + dealt shares, estimated against exact weights, the ungrouped kernel
    try:
        for knob, value in (("match_order", 0), ("match_weigh_exact", 1), ("match_order", 1), ("match_group", 0), ("match_weigh_exact", 0)):
            ctx.set_tuning(knob, value)
            assert same(), (knob, value)
    finally:
        ctx.set_tuning("match_order", 1); ctx.set_tuning("match_weigh_exact", -1); ctx.set_tuning("match_group", 1)
    try:
        ctx.set_timing(0); assert same(); assert ctx.stat("ms_match") != ctx.stat("ms_match")          # NaN
        ctx.set_timing(2); assert same(); assert ctx.stat("ms_match") > 0 and ctx.stat("ms_select") > 0
    finally:
        ctx.set_timing(1)
    assert same() and ctx.stat("ms_match") > 0 and ctx.stat("ms_select") != ctx.stat("ms_select")
    import spacepharer_amd as sa
    try:
        ctx.set_tuning("match_blocks_per_cu", 2); assert same()
        ctx.set_tuning("match_blocks_per_cu", 1); assert same()
    finally:
        ctx.set_tuning("match_blocks_per_cu", 0)
    with pytest.raises(sa.SpError):
        ctx.set_tuning("no_such_knob", 1)


def test_sample_against_oracle_at_full_size(full):
    import gzip
    from oracle_lib import Alp, Oracle
    tres, qres, h, off, nq, ctx = full["tres"], full["qres"], full["h"], full["off"], full["nq"], full["ctx"]
    o = Oracle()
    o.set_targets([tres[i * 200:(i + 1) * 200].tobytes() for i in range(50000)])
    g = json.load(gzip.open(os.path.join(ROOT, "tests/golden/spacer.json.gz")))["ref"]
    o.set_alp(Alp.from_ref(g["alpGapped"]), Alp.from_ref(g["alpGapless"]))
    assert int(ctx.stat("index_entries")) == lib_entries(o)
    rnd = random.Random(7)
    sample = rnd.sample(range(nq), 400)
    aln, cig, n = ctx.align_hits_raw()
    checked = 0
    for q in sample:
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


def lib_entries(o):
    from oracle_lib import lib
    return lib().spo_index_entries(o.ctx)


def bench_mod():
    import bench
    return bench


def _deliver(ctx, slot):
    """one step exactly as bench.py's timed loop runs it; returns the delivered bytes (offsets, hits, records, backtrace pool)"""
    ctx.set_hit_sink(slot.hits, slot.off, slot.cap)
    ctx.prefilter()
    n = slot.n = ctx.num_hits()
    ctx.align_hits_begin(into=slot.into)
    hp, _, cnt = ctx.hits_view()
    assert C.cast(hp, C.c_void_p).value == slot.hits and cnt == n
    ctx.align_hits_end()
    return (C.string_at(slot.off, 4 * (slot.nq + 1)), C.string_at(slot.hits, 8 * n), bench_mod().record_fields(slot.rec, n), C.string_at(slot.cig, int(slot.used.value)))


def test_contexts_in_flight_deliver_the_single_context_bytes():
    """the configurations bench.py times (`step_modes`): search contexts on one GPU, each on its own host thread and HIP streams with its own query batch, ONE staged
    target set shared by all of them (sp_ctx_share_targets), the bench's match_blocks_per_cu -- FOUR contexts with the host-driven step (what `value` comes from:
    step_modes.value_from) and TWO with every step queued device-driven (sp_ctx_search_begin / _end, two steps in flight per context), into their own page-locked
    buffers.  Every step of every context must deliver exactly the bytes (hit list, offsets, sp_alignment records, backtrace pool) the same batch gives through the
    synchronous path when its context has the GPU to itself with default tuning."""
    import argparse
    import threading
    import bench
    import spacepharer_amd as sa
    L = sa.load_library()
    match_blocks = 0      # bench.py --match-blocks default: the library's own choice (two blocks of k_match_grp per CU while several contexts search)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'ap.add_argument("--match-blocks", type=int, default=%d' % match_blocks in src
    src_modes = src[src.index("dual = world == 1"):]
    assert "pipeline=4, host_driven=True" in src_modes      # (the host-driven configuration of the default run: four contexts)
    batches = [bench.make_workload(10000, 50000, 200, 12345, qs) for qs in [None] + [12345 + 104729 * b for b in range(1, 4)]]      # bench.py: batch b of context b
    ctxs, slots, ref = [], [], []
    rounds = 12
    for tres, toff, qres, qoff in batches:
        c = sa.SearchContext()
        if ctxs:
            c.share_targets(ctxs[0])
        else:
            c.set_targets_raw(tres.tobytes(), bench.u64p(toff), len(toff) - 1)
        c.set_queries_raw(qres.tobytes(), bench.u64p(qoff), len(qoff) - 1)
        c.prefilter()
        ss = [bench.Slot(L, len(qoff) - 1, int(c.num_hits() * 1.05) + 1024, True) for _ in range(3)]
        ref.append(_deliver(c, ss[0]))                # alone on the GPU, synchronous path
        ctxs.append(c); slots.append(ss)
    assert len({r[1] for r in ref}) == 4 and len(ref[0][1]) > 8 * 60000 and len(ref[1][2]) > 60 * 60000
    for c in ctxs:
        c.set_tuning("match_blocks_per_cu", match_blocks)

    def got(s, n, used):
        return (C.string_at(s.off, 4 * (s.nq + 1)), C.string_at(s.hits, 8 * n), bench_mod().record_fields(s.rec, n), C.string_at(s.cig, used))

    for device_driven in (True, False):
        bad, errors = [], []
        n_ctx = 2 if device_driven else 4
        gate = threading.Barrier(n_ctx)

        def run(i):
            try:
                c, ss = ctxs[i], slots[i]
                gate.wait()
                if not device_driven:
                    for r in range(rounds):
                        for part, (g, e) in enumerate(zip(_deliver(c, ss[0]), ref[i])):
                            if g != e:
                                bad.append((i, r, part))
                    return
                prev = None
                for r in range(rounds + 1):
                    if r < rounds:
                        s = ss[r % 3]
                        c.search_begin(s.hits, s.off, s.cap, s.rec, s.cig, s.cig_cap)
                    if prev is not None:
                        n, used = c.search_end()
                        for part, (g, e) in enumerate(zip(got(prev, n, used), ref[i])):
                            if g != e:
                                bad.append((i, r - 1, part))
                    prev = ss[r % 3] if r < rounds else None
            except Exception as e:      # noqa: BLE001
                errors.append(repr(e))

        th = [threading.Thread(target=run, args=(i,)) for i in range(n_ctx)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, (device_driven, errors)
        assert not bad, (device_driven, bad[:10])
    assert all(c.stat("steps_repeated") == 0 for c in ctxs[:2])      # (the two that ran device-driven steps)
    for ss in slots:
        for s in ss:
            s.free()
    for c in ctxs:
        c.close()


def test_bench_checks_every_delivered_result(tmp_path):
    """bench.py digests the result of every timed step of every context after the clock stops and fails on a difference; the
    line says how many it checked"""
    import subprocess
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "1", "--regions", "2", "--spacers", "1500",
                        "--targets", "8000", "--cpu-sample", "0"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["results_checked"]["steps"] == 12 and out["results_checked"]["equal_to_single_context_result"] is True
    assert out["regions"]["n"] == 2
    assert out["regions"]["min"] <= out["value"] <= out["regions"]["max"]
    # no mode asked for: both step modes are measured, ONE fixed configuration carries the value (host-driven steps in four contexts), the other stays beside it
    modes = out["step_modes"]
    assert set(modes) == {"device-driven x4", "host-driven x4", "value_from"}
    assert modes["value_from"].startswith("host-driven x4")
    assert out["config"]["distinct_query_batches"] == 4
    assert abs(out["value"] - modes["host-driven x4"]["spacer_queries_per_s"]) < 1e-6 * out["value"]
    assert modes["device-driven x4"]["spacer_queries_per_s"] > 0
