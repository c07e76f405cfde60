"""GPU parity of the probe-list cut of the target-sharded search (include/spacepharer_amd.h: sp_ctx_probe_*): every shard's context
enumerates a SLICE of the queries against the union of the shards' occupancy maps, the occupied probes travel to the shards that hold their slot, every shard
fetches the lists of the probes it received.  The candidates of every shard must equal -- as a set of 16-byte records -- what sp_ctx_prefilter_candidates()
finds when the shard enumerates every query itself, and the merged hits must equal the unsplit search (reference golden vectors / oracle) bit for bit."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _ranges(seqs, world):
    from spacepharer_amd.dist import decompose_by_residues
    lens = [len(s) + 2 for s in seqs]
    return [decompose_by_residues(lens, r, world) for r in range(world)]


def _probe_cut(sa, T, Q, world, slices=None, **kw):
    """-> (hits after the merge, per-shard candidate arrays (probe cut), per-shard candidate arrays (every shard enumerates), contexts)"""
    ctxs = []
    for start, count in _ranges(T, world):
        c = sa.SearchContext(**kw)
        c.set_targets_shard(T, start, start + count)
        c.set_queries(Q)
        ctxs.append(c)
    plain = []
    for c in ctxs:
        c.prefilter_candidates()
        plain.append(c.candidates().copy())
    maps = [c.shard_occupancy() for c in ctxs]
    for c in ctxs:
        c.probe_set_shards(maps)
    if slices is None:                                        # the ranks' slices tile the batch
        cut = [len(Q) * r // world for r in range(world + 1)]
        slices = [(cut[r], cut[r + 1]) for r in range(world)]
    sent = [c.probe_enumerate(a, b) for c, (a, b) in zip(ctxs, slices)]      # sent[src][dst] = (records, probes per query of the slice)
    cands = []
    for dst, c in enumerate(ctxs):
        order = list(range(world))[::-1] if dst % 2 else list(range(world))  # the order of the senders must not matter
        recs = np.concatenate([sent[src][dst][0] for src in order])
        per_query = np.zeros(len(Q), dtype=np.uint32)
        for src, (a, b) in enumerate(slices):
            per_query[a:b] = sent[src][dst][1]
        real = recs[recs[:, 0] != 0xFFFFFFFF]
        assert len(real) == per_query.sum()
        n = c.probe_fetch(recs, per_query)
        cd = c.candidates().copy()
        assert cd.shape == (n, 16)
        cands.append(cd)
    union = np.concatenate(cands, axis=0)
    ctxs[0].merge_candidates(union)
    return ctxs[0].hits(), cands, plain, ctxs, sent


def _as_set(cd):
    return sorted(bytes(r) for r in cd)


@pytest.mark.parametrize("name,world", [("small", 2), ("spacer", 3), ("family", 4), ("spacer", 8)])
def test_probe_cut_equals_reference_unsplit(name, world):
    import spacepharer_amd as sa
    g = load_golden(name)
    r = g["ref"]
    if max(len(q) for q in g["queries"]) > 64:
        pytest.skip("the cut serves spacer peptides")
    got, cands, plain, ctxs, sent = _probe_cut(sa, g["targets"], g["queries"], world, sensitivity=r["sens"], max_seqs=r["maxSeqs"])
    for s in range(world):
        assert _as_set(cands[s]) == _as_set(plain[s]), (name, s)
    for q in r["queries"]:
        assert [list(h) for h in got[q["id"]]] == q["hits"], (name, q["id"])
    for c in ctxs:
        c.close()


def test_probe_cut_with_ties_uneven_slices_and_an_empty_slice():
    from golden.make_golden import make_case
    import spacepharer_amd as sa
    T, Q = make_case(seed=99173, n_targets=2500, n_queries=300, tmin=40, tmax=260, qmin=9, qmax=40, families=6)
    T = T + [T[i % 50] for i in range(700)]
    ref = sa.SearchContext(max_seqs=25)
    ref.set_targets(T)
    ref.set_queries(Q)
    ref.prefilter()
    exp = ref.hits()
    got, cands, plain, ctxs, sent = _probe_cut(sa, T, Q, 5, slices=[(0, 7), (7, 7), (7, 150), (150, 299), (299, 300)], max_seqs=25)
    for s in range(5):
        assert _as_set(cands[s]) == _as_set(plain[s]), s
    assert got == exp
    # what travels: a probe names a query of the sender's slice, and a shard receives fewer probes than the queries have similar k-mers
    for src, (a, b) in enumerate([(0, 7), (7, 7), (7, 150), (150, 299), (299, 300)]):
        for dst in range(5):
            k = sent[src][dst][0][:, 0]
            real = k[k != 0xFFFFFFFF]
            assert len(real) == 0 or (real.min() >= a and real.max() < b)
    ref.close()
    for c in ctxs:
        c.close()


def test_probe_cut_rejects_long_queries_and_missing_maps():
    import spacepharer_amd as sa
    g = load_golden("longq")
    c = sa.SearchContext(sensitivity=g["ref"]["sens"])
    c.set_targets_shard(g["targets"], 0, len(g["targets"]))
    c.set_queries(g["queries"])
    with pytest.raises(sa.SpError, match="occupancy maps"):
        c.n_shards = 1
        c.probe_enumerate(0, 1)
    c.probe_set_shards([c.shard_occupancy()])
    with pytest.raises(sa.SpError, match="64 residues"):
        c.probe_enumerate(0, len(g["queries"]))
    c.close()


@pytest.mark.parametrize("ranks", [2, 3, 4])
def test_bench_probe_cut_on_several_ranks_of_one_gpu(ranks):
    """SP_SHARD=probes: bench.py's N > 1 path with the probe-list cut -- ranks share the GPU, the probe lists and the candidates travel through the library's host
    transport (sp_dist alltoallv / allgather).  Every timed step is compared inside bench.py with the pass in which every shard enumerates every query itself;
    here the aligned pairs over all ranks must add up to the one-rank search's."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "3", "--warmup", "1", "--regions", "2", "--spacers", "1500", "--targets", "6000", "--cpu-sample", "0", "--no-module-e2e"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--shard", "targets", "--exchange", "host"] + common, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    one = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--same-gpu", "--exchange", "host", "--no-replica", "--no-hybrid"] + common,
                       capture_output=True, text=True, timeout=1800, env=dict(env, SP_SHARD="probes"))
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == ranks
    assert out["decompositions"]["value_from"].startswith("target_shards_probe_cut")
    assert out["exchange"]["probe_cut"]["probe_exchange_ms_per_step"] > 0
    assert out["config"]["pairs_aligned_per_step"] == one["config"]["pairs_aligned_per_step"]
    assert out["results_checked"]["steps"] == 6 and out["results_checked"]["equal_to_single_context_result"] is True
