"""N > 1 path on CPU: two and three processes.  The ranks find each other through torch.distributed's gloo store (or SP_DIST_ID); the collectives are the library's
own (sp_dist_*: host transport here, RCCL on GPUs).  Checks the query partition (the arithmetic of the reference's DBReader::decomposeDomainByAminoAcid) and that the
all-gathered hit lists equal the unsplit oracle result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    # the id travels through torch.distributed's gloo store (what torchrun gives a job); every collective after that is the library's own (sp_dist_*, host transport)
    from golden.make_golden import make_case
    from oracle_lib import Oracle
    from spacepharer_amd import dist as spd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("SP_DIST_ID", None)
    comm = spd.init_from_env(spd.TRANSPORT_HOST, rank=rank, world=world, via_torch=True)
    assert (comm.rank, comm.world) == (rank, world)
    T, Q = make_case(seed=31, n_targets=400, n_queries=90, tmin=30, tmax=200, qmin=9, qmax=14)
    start, count = spd.decompose_by_residues([len(q) + 2 for q in Q], rank, world)
    # the checker stands in for the device here: this test is about the partition + exchange logic only
    o = Oracle()
    o.set_targets(T)
    offs, hits = [0], []
    for q in Q[start:start + count]:
        h = o.prefilter_list(q)
        hits += h
        offs.append(len(hits))
    go, gh = spd.allgather_hits_host(comm, start, np.array(offs, dtype=np.uint64), np.array(hits, dtype=np.int64).reshape(-1, 3))
    assert comm.allreduce([rank + 1.0, float(rank)], "sum") == [world * (world + 1) / 2.0, world * (world - 1) / 2.0]
    assert comm.allreduce([float(rank)], "max") == [world - 1.0] and comm.allreduce([float(rank)], "min") == [0.0]
    if rank == 0:
        exp_off, exp = [0], []
        for q in Q:
            exp += o.prefilter_list(q)
            exp_off.append(len(exp))
        ok = (go.tolist() == exp_off) and (gh.tolist() == exp)
        open(out, "w").write("ok" if ok else "mismatch")
    comm.barrier()
    comm.close()
    import torch.distributed as dist
    dist.destroy_process_group()


def test_query_partition_and_allgather_gloo(tmp_path):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_decompose_matches_reference_arithmetic():
    from spacepharer_amd.dist import decompose_by_residues
    lens = [12, 12, 30, 12, 11, 50, 12, 12, 12, 13]
    parts = [decompose_by_residues(lens, r, 3) for r in range(3)]
    # chunk = ceil(176/3) = 59: ranks take entries until their running sum reaches the chunk
    assert parts == [(0, 4), (4, 2), (6, 4)]
    assert sum(c for _, c in parts) == len(lens)
    assert decompose_by_residues(lens, 0, 1) == (0, 10)
    assert decompose_by_residues([5, 5], 1, 4) == (1, 1) and decompose_by_residues([5, 5], 3, 4) == (0, 0)


def _cand_records(rank):
    """deterministic stand-in for a shard's sp_candidate list: rank r contributes 37*r records (rank 0: none)"""
    n = 37 * rank
    rng = np.random.default_rng(1000 + rank)
    rec = np.zeros((n, 4), dtype=np.uint32)
    rec[:, 0] = rng.integers(0, 500, n)            # query
    rec[:, 1] = rank * 10000 + np.arange(n)        # seq_id inside the rank's target range
    rec[:, 2] = rng.integers(0, 1 << 20, n)        # k-mer ordinal
    rec[:, 3] = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)   # diagonal | score << 16
    return rec.view(np.uint8).reshape(n, 16)


def _cand_worker(rank, world, uid_hex, out):
    # here the id comes the other way: SP_DIST_ID in the environment (what the module fan-out gives its per-GPU workers)
    os.environ["SP_DIST_ID"] = uid_hex
    from spacepharer_amd import dist as spd
    comm = spd.init_from_env(spd.TRANSPORT_HOST, rank=rank, world=world)
    parts = comm.allgatherv_host(_cand_records(rank))
    ok = all(parts[r].tolist() == _cand_records(r).tolist() for r in range(world)) and parts[0].shape == (0, 16)
    # equal-size all-gather of a padded block, the shape the device path uses: padding rows are 0xFF and dropped by sp_ctx_merge_candidates
    longest = 37 * (world - 1)
    pad = np.full((longest, 16), 0xFF, dtype=np.uint8)
    mine = _cand_records(rank)
    pad[:len(mine)] = mine
    got = comm.allgather_host(pad).reshape(-1, 16)
    keep = got[got.view(np.uint32).reshape(-1, 4)[:, 0] != 0xFFFFFFFF]
    exp = np.concatenate([_cand_records(r) for r in range(world)], axis=0)
    ok = ok and got.shape[0] == world * longest and keep.tolist() == exp.tolist() and comm.bytes_moved() > 0
    open("%s.%d" % (out, rank), "w").write("ok" if ok else "mismatch")
    comm.barrier()
    comm.close()


def test_candidate_allgather_gloo(tmp_path):
    """target-sharded mode: ragged candidate lists (one rank empty) -> padded all-gather, padding rows are 0xFF"""
    import binascii
    from spacepharer_amd import dist as spd
    out = str(tmp_path / "cand")
    mp.spawn(_cand_worker, args=(3, binascii.hexlify(spd.unique_id()).decode(), out), nprocs=3, join=True)
    assert [open("%s.%d" % (out, r)).read() for r in range(3)] == ["ok"] * 3


def _a2a_block(src, dst):
    """what rank src sends to rank dst: (src + 2 * dst) % 5 * 1000 + 3 words (rank 1 sends rank 2 nothing)"""
    n = 0 if (src, dst) == (1, 2) else ((src + 2 * dst) % 5) * 1000 + 3
    return (np.arange(n, dtype=np.uint32) * 2654435761 + src * 97 + dst).astype(np.uint32)


def _a2a_worker(rank, world, uid_hex, out):
    os.environ["SP_DIST_ID"] = uid_hex
    from spacepharer_amd import dist as spd
    comm = spd.init_from_env(spd.TRANSPORT_HOST, rank=rank, world=world)
    before = comm.bytes_moved()
    got = comm.alltoallv_host([_a2a_block(rank, r) for r in range(world)])
    ok = all(got[r].tolist() == _a2a_block(r, rank).tolist() for r in range(world)) and comm.bytes_moved() > before
    got = comm.alltoallv_host([_a2a_block(rank, r)[:1] for r in range(world)])      # a second round on the same sockets, tiny blocks
    ok = ok and all(got[r].tolist() == _a2a_block(r, rank)[:1].tolist() for r in range(world))
    # a size mismatch (rank 0 expects one byte more from rank 1 than rank 1 sends): EVERY rank must fail, together, before any payload moves -- nobody is left in a receive
    import ctypes as C
    from spacepharer_amd import _lib
    blocks = [np.zeros(8, np.uint8) for _ in range(world)]
    recv = [np.zeros(9 if (rank == 0 and r == 1) else 8, np.uint8) for r in range(world)]
    vp, u64 = C.c_void_p * world, C.c_uint64 * world
    comm.L.sp_dist_alltoallv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    rc = comm.L.sp_dist_alltoallv(comm.h, vp(*[b.ctypes.data for b in blocks]), u64(*[b.nbytes for b in blocks]), vp(*[b.ctypes.data for b in recv]), u64(*[b.nbytes for b in recv]), 0)
    ok = ok and rc != 0 and b"rank 1 sends 8 bytes to rank 0, which expects 9" in _lib.load_library().sp_last_error()
    got = comm.alltoallv_host([_a2a_block(rank, r)[:2] for r in range(world)])      # ... and the sockets are still in step
    ok = ok and all(got[r].tolist() == _a2a_block(r, rank)[:2].tolist() for r in range(world))
    open("%s.%d" % (out, rank), "w").write("ok" if ok else "mismatch")
    comm.barrier()
    comm.close()


@pytest.mark.parametrize("world", [2, 4])
def test_alltoallv_host_transport(tmp_path, world):
    """sp_dist_alltoallv (the probe lists of the probe-list cut travel this way): ragged blocks, one of them empty, through rank 0 over the bootstrap sockets"""
    import binascii
    from spacepharer_amd import dist as spd
    out = str(tmp_path / "a2a")
    mp.spawn(_a2a_worker, args=(world, binascii.hexlify(spd.unique_id()).decode(), out), nprocs=world, join=True)
    assert [open("%s.%d" % (out, r)).read() for r in range(world)] == ["ok"] * world
