"""BASELINE.json configs[4] AT ITS STATED SIZE on one MI355X (tools/predictmatch_full.py; data/predictmatch.sh:27-125): 100 000 spacers in
5 000 arrays -- ONE batch of ~4 x 10^5 query peptides -- against 9 000 genomes (900 unique x 10 key-distinct replicas, ~10^8 translated ORF
residues, a dense index) and against the codon-reversed control, both searches, the set-level chain in memory, FDR.

Checked: (1) the shape of the run and that what was planted is what is found, in every replica; (2) a random sample of query peptides
against the C oracle at that size -- prefilter hits and alignment records; (3) the combined-score entries of sampled arrays recomputed FROM
SCRATCH from nothing but the peptides (C oracle: prefilter + align over all 10^8 residues; Python oracle: proteinaln2nucl ... combinescore),
for the target AND the control search; (4) the match list from the two score DBs; (5) 8 emulated target shards (the 8-GPU form of the config)
reproduce the unsplit bytes: hit list, offsets, alignment records, backtraces."""
import ctypes as C
import gzip
import json
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu

REC = np.dtype([("db_key", "<u4"), ("bit_score", "<i4"), ("raw_score", "<i4"), ("seq_id", "<f4"), ("evalue", "<f8"), ("q_start", "<i4"), ("q_end", "<i4"),
                ("q_len", "<u4"), ("db_start", "<i4"), ("db_end", "<i4"), ("db_len", "<u4"), ("aln_len", "<u4"), ("cigar_offset", "<u4"), ("passed", "<i4"), ("pad", "<u4")])


def _hits_np(ctx):
    hits, offs, n = ctx.hits_compact()
    h = np.ctypeslib.as_array(C.cast(hits, C.POINTER(C.c_int32)), shape=(max(1, n), 3))[:n].copy()
    return h, np.ctypeslib.as_array(offs).astype(np.int64).copy(), n


@pytest.fixture(scope="module")
def full():
    import bench
    import predictmatch_full as pf
    info, st = pf.run()
    # the target search once more on the same context, its bytes kept for the comparisons below (run() went on to the control search)
    ctx, q, t = st["ctx"], st["dbs"]["q"], st["dbs"]["t"]
    ctx.set_targets_raw(t["aa"].ptr(), t["aa"].offs(), t["aa"].n)
    ctx.set_queries_raw(q["aa"].ptr(), q["aa"].offs(), q["aa"].n)
    ctx.prefilter()
    h, off, n = _hits_np(ctx)
    aln, cig, na = ctx.align_hits_raw()
    assert na == n == info["alignments_target"]
    r = np.frombuffer(C.string_at(C.addressof(aln), 64 * n), dtype=REC)
    used = int((r["cigar_offset"].astype(np.int64) + r["aln_len"] + 1)[r["aln_len"] > 0].max())
    st.update(h=h, off=off, n=n, aln=aln, cig=cig, rec=bench.record_fields(C.addressof(aln), n), pool=C.string_at(C.addressof(cig), used))
    yield info, st
    ctx.close()


_oracles = {}


def oracle_for(st, T):
    """the C oracle over the WHOLE target set (index of ~10^8 residues, built once per set and test session)"""
    from oracle_lib import Alp, Oracle
    if T not in _oracles:
        aa = st["dbs"][T]["aa"]
        o = Oracle()
        o.set_targets([aa.seq(i) for i in range(aa.n)])
        g = json.load(gzip.open(os.path.join(ROOT, "tests/golden/spacer.json.gz")))["ref"]
        o.set_alp(Alp.from_ref(g["alpGapped"]), Alp.from_ref(g["alpGapless"]))
        _oracles[T] = o
    return _oracles[T]


def test_the_run_has_the_configs_size_and_finds_what_was_planted(full):
    info, st = full
    assert info["spacers"] == 100000 and info["arrays"] == 5000
    assert info["target_residues"] > 9.0e7 and info["target_orfs"] > 900000 and info["query_peptides"] > 350000
    assert info["alignments_target"] > 2000000 and info["alignments_control"] > 100000
    # every real array matches its genome in (nearly) every replica; what else is reported stays near the FDR asked for (0.05)
    assert info["real_arrays"] == 2500 and info["planted_pairs"] == 25000
    assert info["true_matches_found"] >= 0.97 * info["planted_pairs"], info
    assert info["other_matches"] <= 0.08 * info["matches_reported"], info
    off = st["off"]
    assert off[0] == 0 and off[-1] == st["n"] and np.all(np.diff(off) >= 0) and np.all(np.diff(off) <= 300)


def test_sampled_peptides_against_the_oracle_at_size(full):
    info, st = full
    o = oracle_for(st, "t")
    q, h, off, aln, cig = st["dbs"]["q"]["aa"], st["h"], st["off"], st["aln"], st["cig"]
    rnd = random.Random(41)
    checked = 0
    for qi in rnd.sample(range(q.n), 250):
        qs = q.seq(qi)
        exp = o.prefilter_list(qs)
        got = [[int(x[0]), int(x[1]), ((int(x[2]) & 0xFFFF) ^ 0x8000) - 0x8000] for x in h[off[qi]:off[qi + 1]]]
        assert got == exp, qi
        for k, e in enumerate(exp):
            r = aln[off[qi] + k]
            a = o.align(qs, e[0])
            assert (r.raw_score, r.bit_score, r.q_end, r.db_end, r.passed) == (a.rawScore, a.score, a.qEndPos, a.dbEndPos, a.passed)
            if a.passed:
                assert (r.q_start, r.db_start, r.evalue) == (a.qStartPos, a.dbStartPos, a.eval)
                assert C.string_at(C.addressof(cig) + r.cigar_offset).decode() == a.backtrace.decode()
            checked += 1
    assert checked > 800


@pytest.mark.parametrize("T,name,arrays", [("t", "score", 10), ("c", "score_rev", 6)])
def test_score_entries_of_sampled_arrays_from_scratch(full, T, name, arrays):
    import sp_sets_oracle as so
    info, st = full
    dbs, got = st["dbs"], st[name]
    q, t = dbs["q"], dbs[T]
    o = oracle_for(st, T)
    ex = json.load(gzip.open(os.path.join(ROOT, "tests/golden/examples.json.gz")))["info"]["nucl_evalue"]
    ev = so.Evaluer([float.fromhex(x) for x in ex["alp"]], int(t["nucl"].off[-1]))       # nucleotide.out 10/2: the constants do not depend on the data
    m2s = {k: int(s) for k, s in enumerate(t["set"])}
    size = {int(s): int(c) for s, c in zip(*np.unique(t["set"], return_counts=True))}
    members = {}
    for k, s in enumerate(q["set"]):
        members.setdefault(int(s), []).append(k)
    rnd = random.Random(7)
    real = [a for a in range(info["arrays"]) if st["truth"][a] >= 0]
    rest = [a for a in range(info["arrays"]) if st["truth"][a] < 0]
    sample = rnd.sample(real, arrays - arrays // 3) + rnd.sample(rest, arrays // 3)
    nonempty = alignments = 0
    for a in sample:
        merged = ""
        for k in members.get(a, []):
            qs = q["aa"].seq(k)
            alns = [x for x in (o.align(qs, hh[0]) for hh in o.prefilter_list(qs)) if x.passed]
            alignments += len(alns)
            prot = o.alignment_text(alns)
            t_nuc = {x.dbKey: t["nucl"].seq(x.dbKey) for x in alns}
            t_star = {x.dbKey: t["aa"].seq(x.dbKey).startswith("*") for x in alns}
            nucl = so.protein_aln_to_nucl(prot, q["nucl"].seq(k), t_nuc, qs.startswith("*"), t_star, ev)
            merged += so.best_hit_per_set(so.combine_prot_nucl(prot, nucl), m2s, size, True)
        exp = so.combine_score(merged, len(members.get(a, [])), m2s)
        assert got.get(a, "") == exp, a
        nonempty += exp != ""
    assert nonempty >= arrays // 2 and alignments > 20 * arrays


def test_match_list_is_the_fdr_filter_of_the_two_score_dbs(full):
    import sp_sets_oracle as so
    info, st = full
    exp, thr = so.fdr_filter(st["score"], st["score_rev"], 0.05, 0)
    assert {k: v for k, v in exp.items() if v} == {k: v for k, v in st["match"].items() if v}
    assert thr > 0


def test_eight_emulated_target_shards_reproduce_the_unsplit_bytes(full):
    import bench
    import spacepharer_amd as sa
    from spacepharer_amd.dist import decompose_by_residues
    info, st = full
    q, t = st["dbs"]["q"]["aa"], st["dbs"]["t"]["aa"]
    lens = np.diff(t.off).astype(np.int64) + 2
    parts, keep = [], None
    for r in range(8):
        start, count = decompose_by_residues(lens, r, 8)
        c = sa.SearchContext()
        c.set_targets_shard_raw(t.ptr(), t.offs(), t.n, start, start + count)
        c.set_queries_raw(q.ptr(), q.offs(), q.n)
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
    assert n == st["n"] and np.array_equal(off, st["off"]) and np.array_equal(h, st["h"])
    aln, cig, na = keep.align_hits_raw()
    assert na == n and bench.record_fields(C.addressof(aln), n) == st["rec"]
    assert C.string_at(C.addressof(cig), len(st["pool"])) == st["pool"]
    keep.close()
    for o in _oracles.values():
        o.close()
    _oracles.clear()
