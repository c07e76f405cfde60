"""GPU: the multi-GPU data path of the PRODUCT (include/spacepharer_amd.h: sp_dist_*, sp_ctx_allgather_candidates, sp_ctx_allgather_hits) with two, three and four
ranks.  On a one-GPU box the ranks share the device, so the device buffers travel through the library's host transport (RCCL refuses two ranks on one device); on a node
with a GPU per rank the same calls run over RCCL -- the single-rank RCCL test below loads RCCL, sets the communicator up and runs both collectives through it.
Results must equal the UNSPLIT search's bytes: reference golden vectors."""
import binascii
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _rank_main(rank, world, uid_hex, transport, name, out):
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    import spacepharer_amd as sa
    from spacepharer_amd import dist as spd
    g = load_golden(name)
    r = g["ref"]
    T, Q = g["targets"], g["queries"]
    comm = spd.Comm(binascii.unhexlify(uid_hex), rank, world, transport)
    ok = True
    # ---- target shards: rank g indexes target range g, every rank matches ALL queries; candidates all-gathered, selection on the union for this rank's query slice
    t_from, t_cnt = spd.decompose_by_residues([len(s) + 2 for s in T], rank, world)
    q_from, q_cnt = spd.decompose_by_residues([len(s) + 2 for s in Q], rank, world)
    c = sa.SearchContext(sensitivity=r["sens"], max_seqs=r["maxSeqs"])
    c.set_targets_shard(T, t_from, t_from + t_cnt)
    c.set_queries(Q)
    info = spd.sharded_prefilter(c, comm, q_from, q_from + q_cnt)
    got = c.hits()
    for q in r["queries"]:
        exp = q["hits"] if q_from <= q["id"] < q_from + q_cnt else []
        ok = ok and [list(h) for h in got[q["id"]]] == exp
    ok = ok and info["union"] >= 1 and info["exchange_s"] > 0 and (world == 1 or info["bytes"] > 0)
    # the whole result on every rank: the slices' hit lists, all-gathered (the sharded contexts hold all queries: the slice is cut out for the exchange)
    hp, op, n = c.hits_view()
    import ctypes as C
    off = np.frombuffer(C.string_at(op, 4 * (len(Q) + 1)), dtype=np.uint32)
    c.close()
    # ---- query partition: the whole target set on every rank, rank g searches query slice g; hit lists all-gathered on device buffers
    c = sa.SearchContext(sensitivity=r["sens"], max_seqs=r["maxSeqs"])
    c.set_targets(T)
    c.set_queries(Q[q_from:q_from + q_cnt])
    c.prefilter()
    offs, hits = spd.allgather_hits(c, comm, q_from, len(Q))
    for q in r["queries"]:
        lo, hi = int(offs[q["id"]]), int(offs[q["id"] + 1])
        ok = ok and [[int(h["seq_id"]), int(h["score"]), int(h["diagonal"])] for h in hits[lo:hi]] == q["hits"]
    ok = ok and int(offs[-1]) == sum(len(q["hits"]) for q in r["queries"])
    # ... and from a delivered list in host memory (what a search step hands over)
    hp, op, n = c.hits_view()
    offs2, hits2 = spd.allgather_hits(c, comm, q_from, len(Q), my_hits=C.cast(hp, C.c_void_p), my_offsets=C.cast(op, C.c_void_p), my_n=n)
    ok = ok and np.array_equal(offs, offs2) and hits.tobytes() == hits2.tobytes()
    c.close()
    ok = ok and comm.allreduce([1.0], "sum")[0] == world
    comm.barrier()
    comm.close()
    open("%s.%d" % (out, rank), "w").write("ok" if ok else "mismatch")


@pytest.mark.parametrize("name,world", [("small", 2), ("spacer", 3), ("family", 4)])
def test_ranks_sharing_one_gpu_exchange_through_the_library(name, world, tmp_path):
    from spacepharer_amd import dist as spd
    uid = binascii.hexlify(spd.unique_id()).decode()
    out = str(tmp_path / "rank")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rank_main, args=(r, world, uid, spd.TRANSPORT_HOST, name, out)) for r in range(world)]
    for p in procs: p.start()
    for p in procs: p.join(600)
    assert [p.exitcode for p in procs] == [0] * world
    assert [open("%s.%d" % (out, r)).read() for r in range(world)] == ["ok"] * world


def test_rccl_transport_single_rank(tmp_path):
    """RCCL itself: librccl loaded on demand, ncclCommInitRank, both collectives through ncclAllGather (one rank: all this box has GPUs for)"""
    from spacepharer_amd import dist as spd
    uid = binascii.hexlify(spd.unique_id()).decode()
    out = str(tmp_path / "rccl")
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_rank_main, args=(0, 1, uid, spd.TRANSPORT_RCCL, "small", out))
    p.start(); p.join(600)
    assert p.exitcode == 0 and open(out + ".0").read() == "ok"


def _fallback_main(rank, world, uid_hex, out):
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    from spacepharer_amd import dist as spd
    comm = spd.Comm(binascii.unhexlify(uid_hex), rank, world, spd.TRANSPORT_RCCL)
    open("%s.t%d" % (out, rank), "w").write(str(comm.transport))
    comm.close()


def test_ranks_that_cannot_have_rccl_agree_on_the_host_transport(tmp_path, monkeypatch):
    """RCCL asked for by two ranks that share a GPU (ncclCommInitRank refuses that): every rank ends on the host transport -- none of them returns an error while the
    others wait in a collective -- and the search over it gives the unsplit bytes; with SP_DIST_STRICT=1 the error comes back instead, on every rank"""
    from spacepharer_amd import dist as spd
    ctx = mp.get_context("spawn")
    uid = binascii.hexlify(spd.unique_id()).decode()
    out = str(tmp_path / "fb")
    procs = [ctx.Process(target=_fallback_main, args=(r, 2, uid, out)) for r in range(2)]
    for p in procs: p.start()
    for p in procs: p.join(600)
    assert [p.exitcode for p in procs] == [0, 0]
    assert [open("%s.t%d" % (out, r)).read() for r in range(2)] == [str(spd.TRANSPORT_HOST)] * 2
    uid = binascii.hexlify(spd.unique_id()).decode()
    out = str(tmp_path / "rank")
    procs = [ctx.Process(target=_rank_main, args=(r, 2, uid, spd.TRANSPORT_RCCL, "small", out)) for r in range(2)]
    for p in procs: p.start()
    for p in procs: p.join(600)
    assert [p.exitcode for p in procs] == [0, 0] and [open("%s.%d" % (out, r)).read() for r in range(2)] == ["ok", "ok"]
    monkeypatch.setenv("SP_DIST_STRICT", "1")
    uid = binascii.hexlify(spd.unique_id()).decode()
    out = str(tmp_path / "strict")
    procs = [ctx.Process(target=_fallback_main, args=(r, 2, uid, out)) for r in range(2)]
    for p in procs: p.start()
    for p in procs: p.join(600)
    assert all(p.exitcode not in (0, None) for p in procs)      # (SpError in both ranks: nobody hangs)


def _overflow_rank(rank, world, uid_hex, out):
    """shard 0 holds 2 400 near-copies of one sequence (a query of that family has more than 65 535 index hits there), the other shards random sequences"""
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    import random
    import spacepharer_amd as sa
    from spacepharer_amd import dist as spd
    rnd = random.Random(11)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    core = "".join(rnd.choice(aa) for _ in range(60))
    fam = []
    for i in range(2400):
        s = list(core)
        for _ in range(3):
            s[rnd.randrange(60)] = rnd.choice(aa)
        fam.append("".join(rnd.choice(aa) for _ in range(rnd.randrange(0, 20))) + "".join(s) + "".join(rnd.choice(aa) for _ in range(20)))
    per = sum(len(s) + 2 for s in fam)
    T = list(fam)
    while sum(len(s) + 2 for s in T) < world * per:
        T.append("".join(rnd.choice(aa) for _ in range(90)))
    Q = ["".join(rnd.choice(aa) for _ in range(12)), core, "".join(rnd.choice(aa) for _ in range(10))]
    comm = spd.Comm(binascii.unhexlify(uid_hex), rank, world, spd.TRANSPORT_HOST)
    t_from, t_cnt = spd.decompose_by_residues([len(s) + 2 for s in T], rank, world)
    c = sa.SearchContext()
    c.set_targets_shard(T, t_from, t_from + t_cnt)
    c.set_queries(Q)
    ok = True
    try:
        spd.sharded_prefilter(c, comm, 0, len(Q))
        ok = False                                   # every rank must fail ...
    except sa.SpError as e:
        ok = "candidate pass of rank 0 failed" in str(e) and (rank != 0 or "65535" in str(e))      # ... with rank 0 named, and rank 0 says why
    # the job goes on: the same ranks, the batch without the heavy query
    c.set_queries([Q[0], Q[2]])
    spd.sharded_prefilter(c, comm, 0, 2)
    ref = sa.SearchContext()
    ref.set_targets(T)
    ref.set_queries([Q[0], Q[2]])
    ref.prefilter()
    ok = ok and c.hits() == ref.hits()
    c.close(); ref.close()
    comm.barrier()
    comm.close()
    open("%s.%d" % (out, rank), "w").write("ok" if ok else "mismatch")


@pytest.mark.parametrize("world", [2, 3])
def test_one_rank_beyond_the_hit_stream_limit_fails_all_ranks_together(world, tmp_path):
    """target shards, one rank's shard refuses a query (more than 65 535 index hits): the candidate all-gather behind it fails on EVERY rank with the offender named --
    no rank is left in a receive -- and the same communicator carries the next batch"""
    from spacepharer_amd import dist as spd
    uid = binascii.hexlify(spd.unique_id()).decode()
    out = str(tmp_path / "ovf")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_overflow_rank, args=(r, world, uid, out)) for r in range(world)]
    for p in procs: p.start()
    for p in procs: p.join(600)
    assert [p.exitcode for p in procs] == [0] * world
    assert [open("%s.%d" % (out, r)).read() for r in range(world)] == ["ok"] * world
