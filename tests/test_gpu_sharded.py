"""GPU parity of the target-sharded search: the target set is cut into contiguous ranges, every
range is searched by its own context (on a multi-GPU node: its own GPU), the per-shard candidate lists are concatenated
(what the RCCL all-gather delivers) and the selection runs on the union.  The result must equal the UNSPLIT search --
reference golden vectors and CPU oracle -- bit for bit: hit sets, scores, diagonals, order."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _ranges(seqs, world):
    from spacepharer_amd.dist import decompose_by_residues
    lens = [len(s) + 2 for s in seqs]
    return [decompose_by_residues(lens, r, world) for r in range(world)]


def _sharded_hits(sa, T, Q, world, pad_rows=0, **kw):
    ctxs, cands = [], []
    for start, count in _ranges(T, world):
        c = sa.SearchContext(**kw)
        c.set_targets_shard(T, start, start + count)
        c.set_queries(Q)
        n = c.prefilter_candidates()
        cd = c.candidates()
        assert cd.shape == (n, 16)
        if n:
            seq_id = cd.view(np.uint32).reshape(-1, 4)[:, 1]
            assert seq_id.min() >= start and seq_id.max() < start + count          # a shard only reports its own targets
        cands.append(cd)
        if pad_rows:
            cands.append(np.full((pad_rows, 16), 0xFF, dtype=np.uint8))           # all-gather padding is ignored
        ctxs.append(c)
    union = np.concatenate(cands, axis=0)
    rng = np.random.default_rng(7)
    union = union[rng.permutation(len(union))]                                     # arrival order must not matter
    ctxs[0].merge_candidates(union)
    got = ctxs[0].hits()
    return got, ctxs


@pytest.mark.parametrize("name,world", [("small", 2), ("spacer", 3), ("family", 4), ("longq", 8)])
def test_sharded_search_equals_reference_unsplit(name, world):
    import spacepharer_amd as sa
    g = load_golden(name)
    r = g["ref"]
    got, ctxs = _sharded_hits(sa, g["targets"], g["queries"], world, pad_rows=5, sensitivity=r["sens"], max_seqs=r["maxSeqs"])
    for q in r["queries"]:
        assert [list(h) for h in got[q["id"]]] == q["hits"], (name, q["id"])
    # the context that merged is in the post-prefilter state: align its hits and compare with the golden alignments
    if name == "small":
        hits, offs, total = ctxs[0].hits_compact()
        res, cig, n = ctxs[0].align_hits_raw()
        assert n == total == sum(len(q["hits"]) for q in r["queries"])
    for c in ctxs:
        c.close()


def test_sharded_search_equals_unsplit_gpu_and_oracle_with_ties():
    """many near-identical targets -> more candidates than --max-seqs with equal scores: the cut is decided by the
    (score, seqId & (BINSIZE-1), k-mer ordinal, seqId) order, which must come out the same from 5 shards"""
    from golden.make_golden import make_case
    from oracle_lib import Oracle
    import spacepharer_amd as sa
    T, Q = make_case(seed=99173, n_targets=2500, n_queries=300, tmin=40, tmax=260, qmin=9, qmax=40, families=6)
    T = T + [T[i % 50] for i in range(700)]              # exact duplicates: equal scores on many targets
    ref = sa.SearchContext(max_seqs=25)
    ref.set_targets(T)
    ref.set_queries(Q)
    ref.prefilter()
    exp = ref.hits()
    got, ctxs = _sharded_hits(sa, T, Q, 5, max_seqs=25)
    assert got == exp
    assert max(len(h) for h in exp) == 25
    o = Oracle(sens=5.7, max_seqs=25)
    o.set_targets(T)
    for i in range(0, len(Q), 7):
        assert [list(h) for h in got[i]] == o.prefilter_list(Q[i]), i
    o.close()
    # a query slice: queries outside [q_from, q_to) come back empty, the others unchanged
    union = np.concatenate([c.candidates() for c in ctxs], axis=0)
    ctxs[1].merge_candidates(union, q_from=100, q_to=200)
    part = ctxs[1].hits()
    assert part[100:200] == exp[100:200] and all(len(h) == 0 for h in part[:100] + part[200:])
    ref.close()
    for c in ctxs:
        c.close()


def test_bench_sharded_path_over_rccl_single_rank():
    """bench.py --shard targets on one GPU: candidates travel through a real RCCL all-gather (one-rank group) from and
    to HBM, then the selection + align; it must align exactly the pairs the unsplit path aligns"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--spacers", "600", "--targets", "4000",
            "--cpu-sample", "0"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    outs = []
    for extra in ([], ["--shard", "targets"], ["--shard", "probes"]):
        p = subprocess.run(base + extra, capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]))
    a, b, c = outs
    # the probe-list cut through the same one-rank RCCL group (its all-to-all degenerates to the rank's own block): every timed step equals the plain sharded pass
    assert c["config"]["pairs_aligned_per_step"] == a["config"]["pairs_aligned_per_step"] and c["counters_per_step"]["hits"] == a["counters_per_step"]["hits"]
    assert "probe_cut" in c["exchange"] and c["exchange"]["rccl_ranks"] == 1
    assert b["scaling"] == "strong" and "target shards" in b["config"]["parallelism"]
    assert a["config"]["pairs_aligned_per_step"] == b["config"]["pairs_aligned_per_step"] > 0
    assert a["counters_per_step"]["hits"] == b["counters_per_step"]["hits"]
    assert a["counters_per_step"]["similar_kmers"] == b["counters_per_step"]["similar_kmers"]


@pytest.mark.parametrize("ranks", [2, 3])
def test_bench_self_launches_the_target_sharded_run_with_several_ranks(ranks):
    """`python bench.py --gpus N` without a launcher starts its N ranks itself.  On a one-GPU box the ranks share the device
    (--same-gpu) and the candidates travel through host memory (--exchange gloo: RCCL refuses two ranks on one device); everything
    else is the N > 1 path the driver runs on a multi-GPU node: target shards, all-gather, selection on the union by query slice,
    align by query slice, the query-partition (replica) leg afterwards.  Every step's delivered bytes are checked inside bench.py
    against a single-context pass of the same rank; here the aligned pairs over all ranks must add up to the unsplit search's."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "3", "--warmup", "1", "--regions", "2", "--spacers", "1500", "--targets", "6000", "--cpu-sample", "0", "--no-module-e2e"]
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    one = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--same-gpu", "--exchange", "gloo"] + common,
                       capture_output=True, text=True, timeout=1800, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 alone prints the line
    out = json.loads(lines[0])
    assert out["n_gpus"] == ranks and out["scaling"] == "strong"
    # the headline is the best decomposition of the same job; all of them are in the line
    dec = out["decompositions"]
    # the headline at N > 1 is BASELINE configs[3]'s decomposition (target shards + candidate all-gather), fixed; the other cuts of the same job are side figures
    assert set(dec["spacer_queries_per_s"]) >= {"target_shards", "query_partition"} and dec["value_from"].startswith("target_shards") and dec["best_measured"] in dec["spacer_queries_per_s"]
    assert out["value"] == dec["spacer_queries_per_s"]["target_shards"]
    assert out["weak_replica"]["scaling"] == "weak" and "scaling_note" in out
    assert out["exchange"]["requested"] in ("host", "gloo") and out["exchange"]["as_requested"] is True and out["exchange"]["rccl_ranks"] == 0
    assert ("target shards x%d" % ranks) in out["config"]["parallelism"] or "target shards" in out["config"]["parallelism"]
    qp = out["query_partition"]
    assert qp["scaling"] == "strong" and qp["results_checked"] > 0 and qp["allgather_bytes_per_step"] > 0 and qp["exchange_ms_per_step"] > 0
    assert set(qp["step_modes"]) == {"device-driven x2", "host-driven x4"} and abs(qp["value"] - max(qp["step_modes"].values())) < 1e-6 * qp["value"]      # both step modes, the better one counts
    assert out["exchange"]["ranks"] == ranks and out["exchange"]["allgather_bytes_per_step"] > 0 and out["exchange"]["exchange_ms_per_step"] > 0
    assert out["config"]["pairs_aligned_per_step"] == one["config"]["pairs_aligned_per_step"]
    assert out["results_checked"]["steps"] == 6
    assert out["replica"]["scaling"] == "weak" and out["replica"]["value"] > 0 and out["replica"]["results_checked"] > 0
    # a --gpus that disagrees with the launcher's WORLD_SIZE is refused instead of silently measuring something else
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"] + common, capture_output=True, text=True, timeout=300,
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_query_beyond_the_hit_stream_limit_is_named_and_searchable_through_shards():
    """one query with more than 65 535 index hits (the 16-bit stream position of k_match; the reference's own buffer holds 2 x 10^6):
    the unsplit call refuses the batch and names the query, the same batch against target shards gives the oracle's hits"""
    import random
    import spacepharer_amd as sa
    from oracle_lib import Oracle
    rnd = random.Random(11)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    core = "".join(rnd.choice(aa) for _ in range(60))
    T = []
    for i in range(2400):
        s = list(core)
        for _ in range(3):
            s[rnd.randrange(60)] = rnd.choice(aa)
        T.append("".join(rnd.choice(aa) for _ in range(rnd.randrange(0, 20))) + "".join(s) + "".join(rnd.choice(aa) for _ in range(20)))
    Q = ["".join(rnd.choice(aa) for _ in range(12)), core, "".join(rnd.choice(aa) for _ in range(10))]
    c = sa.SearchContext()
    c.set_targets(T)
    c.set_queries(Q)
    with pytest.raises(sa.SpError) as e:
        c.prefilter()
    assert "position 1 of the batch" in str(e.value) and "65535" in str(e.value)
    c.set_queries([Q[0], Q[2]])      # the context stays usable
    c.prefilter()
    o = Oracle()
    o.set_targets(T)
    assert [[list(h) for h in hs] for hs in c.hits()] == [o.prefilter_list(Q[0]), o.prefilter_list(Q[2])]
    c.close()
    got, ctxs = _sharded_hits(sa, T, Q, 8)
    for q in range(3):
        assert [list(h) for h in got[q]] == o.prefilter_list(Q[q]), q
    assert len(got[1]) == 300
    for x in ctxs:
        x.close()
    o.close()


def test_bench_two_shards_times_two_query_groups_on_four_ranks():
    """four ranks as 2 target shards x 2 query groups (`--shards 2`): every group searches its half of the queries against both shards
    and all-gathers within the group -- the aligned pairs of all ranks add up to the unsplit search's; the default run of four ranks
    reports the same decomposition under the `hybrid` key next to the four-shard line"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "3", "--warmup", "1", "--regions", "2", "--spacers", "1500", "--targets", "6000", "--cpu-sample", "0", "--no-module-e2e"]
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    one = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--same-gpu", "--exchange", "gloo", "--shards", "2", "--no-replica"] + common,
                       capture_output=True, text=True, timeout=1800, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 4 and "target shards x2 x 2 query groups" in out["config"]["parallelism"]
    assert out["config"]["pairs_aligned_per_step"] == one["config"]["pairs_aligned_per_step"]
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--same-gpu", "--exchange", "gloo", "--no-replica"] + common,
                       capture_output=True, text=True, timeout=1800, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["hybrid"]["value"] > 0 and out["hybrid"]["results_checked"] > 0
    dec = out["decompositions"]["spacer_queries_per_s"]      # four shards and 2 x 2: the four shards (configs[3]) carry the headline, the hybrid stays beside it
    assert set(dec) == {"target_shards", "hybrid"} and out["value"] == dec["target_shards"] and "target shards" in out["config"]["parallelism"]
    assert out["config"]["pairs_aligned_per_step"] == one["config"]["pairs_aligned_per_step"]
