"""Multi-GPU helpers: one process per GPU.  Every collective of the data path is the library's own (include/spacepharer_amd.h, sp_dist_* / sp_ctx_allgather_*:
RCCL all-gather of device buffers over xGMI, counts and timings over the bootstrap sockets); this module is the ctypes binding plus the one thing a launcher has to do --
hand the 128-byte id that rank 0 makes to every rank.  Under torchrun that goes through torch.distributed's store (gloo), otherwise through SP_DIST_ID (hex) or a file.

Query partition (the reference's default, lib/mmseqs/src/prefiltering/Prefiltering.cpp:273-281,735-740): every rank holds the whole target index in its own HBM and
searches a contiguous slice of the queries (decompose_by_residues = DBReader::decomposeDomainByAminoAcid); `allgather_hits` brings all hit lists to every rank.
Target shards: rank g indexes target range g, all ranks match all queries, `sharded_prefilter` exchanges the 16-byte per-(query, target)
candidates and runs the selection on the union -- the unsplit result, exactly.
"""
import binascii
import ctypes as C
import os

import numpy as np

from . import _lib

TRANSPORT_RCCL, TRANSPORT_HOST = 0, 1
ID_BYTES = 128
CAND_BYTES = 16     # sizeof(sp_candidate)


def _L():
    L = _lib.load_library()
    if not getattr(L, "_dist_bound", False):
        vp = C.c_void_p
        L.sp_dist_unique_id.argtypes = [C.c_char_p]
        L.sp_dist_init.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.sp_dist_init.restype = vp
        L.sp_dist_destroy.argtypes = [vp]
        L.sp_dist_destroy.restype = None
        L.sp_dist_barrier.argtypes = [vp]
        L.sp_dist_allgather.argtypes = [vp, vp, C.c_uint64, vp, C.c_int]
        L.sp_dist_allreduce_f64.argtypes = [vp, C.POINTER(C.c_double), C.c_uint64, C.c_int]
        L.sp_dist_bytes_moved.argtypes = [vp]
        L.sp_dist_bytes_moved.restype = C.c_uint64
        L.sp_dist_decompose.argtypes = [C.POINTER(C.c_uint64), C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.sp_ctx_allgather_candidates.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        L.sp_ctx_allgather_hits.argtypes = [vp, vp, C.c_uint64, C.c_uint64, vp, vp, vp, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint64)]
        L._dist_bound = True
    return L


def unique_id():
    """the 128 bytes every rank of a job passes to Comm(); made once, by anybody"""
    buf = C.create_string_buffer(ID_BYTES)
    _lib._check(_L().sp_dist_unique_id(buf))
    return buf.raw


class Comm:
    """the ranks of one job (sp_comm).  Collective constructor."""

    def __init__(self, uid, rank, world, transport=TRANSPORT_RCCL):
        self.L = _L()
        self.rank, self.world, self.transport = int(rank), int(world), int(transport)
        self.h = self.L.sp_dist_init(C.c_char_p(bytes(uid)), self.rank, self.world, self.transport)
        if not self.h:
            raise _lib.SpError(self.L.sp_last_error().decode())
        self.L.sp_dist_transport.argtypes = [C.c_void_p]
        self.transport = int(self.L.sp_dist_transport(self.h))      # (RCCL asked for, host transport agreed on: see sp_dist_init)

    def close(self):
        if self.h:
            self.L.sp_dist_destroy(self.h)
            self.h = None

    def barrier(self):
        _lib._check(self.L.sp_dist_barrier(self.h))

    def allgather_host(self, arr):
        """equal-size all-gather of a numpy array: returns [world, *arr.shape]"""
        a = np.ascontiguousarray(arr)
        out = np.empty((self.world,) + a.shape, dtype=a.dtype)
        if a.nbytes:
            _lib._check(self.L.sp_dist_allgather(self.h, a.ctypes.data, a.nbytes, out.ctypes.data, 0))
        return out

    def allgather_device(self, send_ptr, nbytes, recv_ptr):
        _lib._check(self.L.sp_dist_allgather(self.h, send_ptr, nbytes, recv_ptr, 1))

    def allgatherv_host(self, arr):
        """ragged all-gather of numpy arrays that differ in their first dimension: list of world arrays (counts, then payloads padded to the longest)"""
        a = np.ascontiguousarray(arr)
        n = self.allgather_host(np.array([a.shape[0]], dtype=np.int64))[:, 0]
        longest = max(1, int(n.max()))
        pad = np.zeros((longest,) + a.shape[1:], dtype=a.dtype)
        pad[:a.shape[0]] = a
        allp = self.allgather_host(pad)
        return [allp[r][:int(n[r])] for r in range(self.world)]

    def alltoallv_host(self, blocks):
        """blocks[r]: the numpy array (1-D, one dtype for all) this rank sends to rank r; returns the list of the arrays the ranks sent to this one (sizes exchanged first)"""
        blocks = [np.ascontiguousarray(b) for b in blocks]
        assert len(blocks) == self.world
        dt = blocks[0].dtype
        sizes = self.allgather_host(np.array([b.nbytes for b in blocks], dtype=np.uint64))      # [src, dst]
        recv = [np.empty(int(sizes[r][self.rank]) // dt.itemsize, dtype=dt) for r in range(self.world)]
        vp = C.c_void_p * self.world
        u64 = C.c_uint64 * self.world
        self.L.sp_dist_alltoallv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib._check(self.L.sp_dist_alltoallv(self.h, vp(*[b.ctypes.data for b in blocks]), u64(*[b.nbytes for b in blocks]), vp(*[b.ctypes.data for b in recv]), u64(*[b.nbytes for b in recv]), 0))
        return recv

    def allreduce(self, values, op="sum"):
        v = (C.c_double * len(values))(*[float(x) for x in values])
        _lib._check(self.L.sp_dist_allreduce_f64(self.h, v, len(values), {"sum": 0, "max": 1, "min": 2}[op]))
        return [float(x) for x in v]

    def bytes_moved(self):
        return int(self.L.sp_dist_bytes_moved(self.h))

    def split(self, color):
        """a communicator of the ranks that pass the same color (collective over this one); ranks keep their order"""
        colors = self.allgather_host(np.array([int(color)], dtype=np.int64))[:, 0]
        members = [r for r in range(self.world) if colors[r] == color]
        mine = np.frombuffer(unique_id() if members[0] == self.rank else bytes(ID_BYTES), dtype=np.uint8)
        ids = self.allgather_host(mine)
        return Comm(ids[members[0]].tobytes(), members.index(self.rank), len(members), self.transport)


def init_from_env(transport=TRANSPORT_RCCL, rank=None, world=None, via_torch=False):
    """Comm of the job this process was launched into: RANK / WORLD_SIZE from the environment; the id from SP_DIST_ID (hex), else derived from the launcher's
    MASTER_ADDR / MASTER_PORT (id_from_launcher_env), or -- via_torch -- made by rank 0 and broadcast through torch.distributed's gloo store."""
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    hexid = os.environ.get("SP_DIST_ID")
    if hexid:
        uid = binascii.unhexlify(hexid)
    elif via_torch:
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group("gloo", rank=rank, world_size=world)
        box = [unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]
    else:
        uid = id_from_launcher_env()
    return Comm(uid, rank, world, transport)


def id_from_launcher_env():
    """the id every rank of a torchrun-style launch can write down by itself, without talking to anybody: rank 0 listens on MASTER_PORT + 1 (the launcher's own store sits
    on MASTER_PORT), the token is a hash of what all ranks of the job share (MASTER_ADDR, MASTER_PORT, the run id).  No torch in the process: PyTorch brings its own HIP
    runtime and its own RCCL, and a process that talks to the GPU through this library has no use for a second runtime."""
    import hashlib
    import struct
    port = (int(os.environ.get("MASTER_PORT", "29533")) + 1 + int(os.environ.get("SP_DIST_PORT_OFFSET", "0"))) % 65536
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    try:
        import socket
        addr = socket.gethostbyname(addr)
    except OSError:
        addr = "127.0.0.1"
    seed = "|".join([addr, os.environ.get("MASTER_PORT", ""), os.environ.get("TORCHELASTIC_RUN_ID", ""), os.environ.get("WORLD_SIZE", "")])
    token = struct.unpack("<Q", hashlib.sha256(seed.encode()).digest()[:8])[0]
    raw = b"SPDIST1\0" + struct.pack("<HHIQ", port, 0, 0, token) + addr.encode()[:63]
    return raw + bytes(ID_BYTES - len(raw))


def decompose_by_residues(entry_lengths, rank, world):
    """DBReader::decomposeDomainByAminoAcid (lib/mmseqs/src/commons/DBReader.cpp:1171-1213): contiguous slices balanced by data size.  entry_lengths are the index
    `length` fields (sequence length + 2).  Returns (start, count)."""
    a = np.ascontiguousarray(entry_lengths, dtype=np.uint64)
    s, c = C.c_uint64(), C.c_uint64()
    rc = _L().sp_dist_decompose(a.ctypes.data_as(C.POINTER(C.c_uint64)), len(a), int(rank), int(world), C.byref(s), C.byref(c))
    if rc != 0:
        raise ValueError(_L().sp_last_error().decode())
    return int(s.value), int(c.value)


def allgather_hits_host(comm, query_offset, offsets, hits):
    """Ragged all-gather of per-rank hit lists that already lie in host memory (the steps behind the search: per-set scores, text entries, ...).
    query_offset: index of this rank's first query in the global order; offsets uint64[nq_local + 1]; hits [n, k] of any dtype.
    Returns (global_offsets uint64[nq_total + 1], global_hits [n_total, k]) ordered by global query index."""
    counts = np.diff(np.asarray(offsets, dtype=np.int64))
    hits = np.ascontiguousarray(hits)
    metas = comm.allgather_host(np.array([int(query_offset), len(counts)], dtype=np.int64))
    call = comm.allgatherv_host(counts.astype(np.int64))
    hall = comm.allgatherv_host(hits)
    order = sorted(range(comm.world), key=lambda r: (int(metas[r][0]), r))
    gc = np.concatenate([call[r] for r in order]) if order else np.zeros(0, dtype=np.int64)
    go = np.zeros(len(gc) + 1, dtype=np.uint64)
    go[1:] = np.cumsum(gc).astype(np.uint64)
    gh = np.concatenate([hall[r] for r in order], axis=0)
    return go, gh


def allgather_hits(ctx, comm, query_offset, nq_total, hit_cap=None, my_hits=None, my_offsets=None, my_n=None):
    """query partition: the hit lists of all ranks' slices, in global query order, on every rank (sp_ctx_allgather_hits: device buffers, RCCL).
    my_hits / my_offsets: addresses of a delivered step's list in host memory (with my_n hits); default: the context's last prefilter, where it lies in HBM.
    Returns (offsets uint64[nq_total + 1], hits structured array of sp_hit_packed)."""
    L = _L()
    total = comm.allreduce([ctx.num_hits() if my_n is None else my_n], "sum")[0]
    cap = int(hit_cap if hit_cap is not None else total) + 1
    hits = np.zeros(cap, dtype=np.dtype([("seq_id", np.uint32), ("score", np.uint16), ("diagonal", np.int16)]))
    offs = np.zeros(int(nq_total) + 1, dtype=np.uint64)
    n = C.c_uint64()
    _lib._check(L.sp_ctx_allgather_hits(ctx.ctx, comm.h, int(query_offset), int(nq_total), my_hits, my_offsets, hits.ctypes.data, offs.ctypes.data_as(C.POINTER(C.c_uint64)), cap, C.byref(n)))
    return offs, hits[:int(n.value)]


K_SLOTS = 64000000      # 20^6 k-mer slots

# Which pass a target-sharded search runs on every rank -- decided from what was MEASURED on one MI355X playing shard 0 of S (profiles/r05_probe_cut.txt: 1e7 residues,
# profiles/r06_probe_cut_dense.txt: 2e8 residues; 100 000 / 20 000 spacers):
#     occupancy of the shard's slot table   S    plain k_match_grp<true>   enumerate 1/S + fetch   cut / plain   probes received per rank and 100 000 spacers
#     0.02  (1e7 residues / 8)              8    2.60 ms                   0.46 + 0.55 ms          2.5x          10 M  (161 MB)
#     0.04  (1e7 / 4)                       4    2.81                      0.74 + 0.72             1.9x          17 M
#     0.07  (1e7 / 2)                       2    3.10                      1.69 + 1.00             1.16x         31 M
#     0.31  (2e8 / 8)                       8    1.31 (20 000 spacers)     0.75 + 0.89             0.80x         119 M (1.9 GB)
# The cut removes the enumeration every shard repeats; what it adds grows with the number of occupied probes, i.e. with the occupancy.  On a sparse shard (phage-scale proxy) the
# enumeration is the matching kernel's time and the cut wins from four shards on; at the density of the real DB the index hits dominate, a k-mer is held by 2.5 of 8 shards and
# the probes drown the exchange: the plain pass, which there scales by itself (an eighth of 2e8 residues costs 0.36 of the unsplit matcher's time).
PROBE_CUT_MAX_OCCUPANCY = 0.08
PROBE_CUT_MIN_SHARDS = 4


def choose_sharded_pass(occupancy, shards, max_query_len=64):
    """'probes' or 'plain' for a target-sharded search over `shards` ranks whose fullest shard has `occupancy` of its k-mer slots occupied (see the table above)"""
    if shards >= PROBE_CUT_MIN_SHARDS and shards <= 8 and occupancy <= PROBE_CUT_MAX_OCCUPANCY and max_query_len <= 64:
        return "probes"
    return "plain"


def agree_sharded_pass(ctx, comm, max_query_len=64):
    """the ranks' verdict (one all-reduce of the occupancies: every rank decides from the fullest shard, hence alike)"""
    occ = ctx.stat("index_slots") / float(K_SLOTS)
    occ = comm.allreduce([occ], "max")[0] if comm.world > 1 else occ
    return choose_sharded_pass(occ, comm.world, max_query_len), occ


def sharded_prefilter(ctx, comm, q_from=0, q_to=None, probes=False):
    """One target-sharded prefilter pass on this rank's context (targets staged with set_targets_shard*): candidates of the own shard -> all-gather (RCCL: HBM -> xGMI ->
    HBM) -> selection on the union for queries [q_from, q_to).  Afterwards ctx.hits_compact() / ctx.align_hits_raw() see the merged hits.
    Returns {"union": rows of the gathered list, "bytes": bytes this rank received, "exchange_s": wall time of the exchange (counts + payload)}."""
    rows, secs, psecs = C.c_uint64(), C.c_double(), C.c_double()
    before = comm.bytes_moved()
    if probes:
        # the probe-list cut: this rank enumerates queries [q_from, q_to) only, the occupied probes travel to the shards that hold their slot (sp_ctx_probe_search; rank r
        # of `comm` is shard r) -- the candidates it leaves are those of prefilter_candidates(), record for record
        _lib._check(_L().sp_ctx_probe_search(ctx.ctx, comm.h, int(q_from), int(q_to if q_to is not None else ctx.n_queries), C.byref(psecs)))
    else:
        try:
            ctx.prefilter_candidates()
        except _lib.SpError:
            # rank-local (a query of THIS shard beyond the hit-stream limit, say): the context remembers it and the collective below fails on every rank, together --
            # nobody is left waiting in a receive for the rank that stopped here
            pass
    _lib._check(_L().sp_ctx_allgather_candidates(ctx.ctx, comm.h, int(q_from), int(q_to if q_to is not None else (1 << 62)), C.byref(rows), C.byref(secs)))
    return {"union": int(rows.value), "bytes": comm.bytes_moved() - before, "exchange_s": float(secs.value) + float(psecs.value), "probe_exchange_s": float(psecs.value)}
