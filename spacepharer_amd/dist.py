"""Multi-GPU helpers: one process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI on MI355X, "gloo" in
the CPU tests).

The search shards by QUERY: every rank holds the whole target index in its own 288 GB of HBM and searches a
contiguous slice of the query DB -- the reference's default QUERY_DB_SPLIT (lib/mmseqs/src/prefiltering/
Prefiltering.cpp:273-281,735-740), which keeps results identical to the unsplit run (a target split is not:
Prefiltering.cpp:343-346 applies a per-split hit quota).  No data-path collective is needed; the only exchange is
the variable-length all-gather of the per-rank hit lists so that every rank (or just rank 0) can write the result
DB / feed the host-side aggregation steps.

Target-sharded mode (for target sets that should be split across GPUs): rank g indexes the target
range decompose_by_residues(target_lengths, g, world), all ranks search all queries, `allgather_candidates` exchanges
the 16-byte per-(query, target) candidates (RCCL all-gather of device buffers over xGMI; gloo/host in the CPU tests) and
every rank runs the selection on the union for its slice of the queries (SearchContext.merge_candidates*), which
reproduces the unsplit result exactly.
"""
import math

import numpy as np


def decompose_by_residues(entry_lengths, rank, world):
    """DBReader::decomposeDomainByAminoAcid (lib/mmseqs/src/commons/DBReader.cpp:1171-1213): contiguous slices balanced
    by data size.  entry_lengths are the index `length` fields (sequence length + 2).  Returns (start, count)."""
    n = len(entry_lengths)
    data_size = int(sum(int(x) for x in entry_lengths))
    if world > max(1, data_size):
        raise ValueError("world size %d larger than the data size %d" % (world, data_size))
    if world == 1:
        return 0, n
    if n <= world:
        return (rank if rank < n else 0), (1 if rank < n else 0)
    chunk = int(math.ceil(float(data_size) / float(world)))
    per = [0] * world
    cur, acc = 0, 0
    for length in entry_lengths:
        if acc >= chunk:
            acc = 0
            cur += 1
        acc += int(length)
        per[cur] += 1
    return sum(per[:rank]), per[rank]


def allgather_hits(query_offset, offsets, hits, group=None, device=None):
    """Variable-length all-gather of per-rank hit lists.

    query_offset : index of this rank's first query in the global query order
    offsets      : uint64[nq_local + 1] -- hits of local query i are hits[offsets[i]:offsets[i+1]]
    hits         : int64[n_hits, 3]     -- (target index, score, diagonal)
    Returns (global_offsets uint64[nq_total + 1], global_hits int64[n, 3]) ordered by global query index.
    Two collectives: the counts (one int64 pair per rank), then the padded payloads -- payloads are a few MB, so on
    xGMI this is latency-bound whatever the algorithm."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    counts = np.diff(np.asarray(offsets, dtype=np.int64))
    meta = torch.tensor([int(query_offset), len(counts), int(counts.sum())], dtype=torch.int64, device=dev)
    metas = [torch.zeros(3, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    metas = [m.cpu().numpy() for m in metas]
    max_q = max(int(m[1]) for m in metas)
    max_h = max(int(m[2]) for m in metas)
    cpad = torch.zeros(max(1, max_q), dtype=torch.int64, device=dev)
    cpad[:len(counts)] = torch.from_numpy(counts).to(dev)
    hpad = torch.zeros((max(1, max_h), 3), dtype=torch.int64, device=dev)
    if len(hits):
        hpad[:len(hits)] = torch.from_numpy(np.ascontiguousarray(hits, dtype=np.int64)).to(dev)
    call = [torch.zeros_like(cpad) for _ in range(world)]
    hall = [torch.zeros_like(hpad) for _ in range(world)]
    dist.all_gather(call, cpad, group=group)
    dist.all_gather(hall, hpad, group=group)
    order = sorted(range(world), key=lambda r: int(metas[r][0]))
    all_counts, all_hits = [], []
    for r in order:
        nq, nh = int(metas[r][1]), int(metas[r][2])
        all_counts.append(call[r][:nq].cpu().numpy())
        all_hits.append(hall[r][:nh].cpu().numpy())
    gc = np.concatenate(all_counts) if all_counts else np.zeros(0, dtype=np.int64)
    go = np.zeros(len(gc) + 1, dtype=np.uint64)
    go[1:] = np.cumsum(gc).astype(np.uint64)
    gh = np.concatenate(all_hits) if all_hits else np.zeros((0, 3), dtype=np.int64)
    return go, gh


CAND_BYTES = 16     # sizeof(sp_candidate)


def allgather_candidates(local, group=None, device=None):
    """All-gather of the per-shard candidate lists.

    local : uint8[n, 16] -- numpy array (host exchange) or torch tensor (on `device`: RCCL moves HBM to HBM)
    Returns a torch uint8 tensor [world * max_n, 16] on `device`; unused rows are 0xFF (query == 0xFFFFFFFF = padding,
    ignored by sp_ctx_merge_candidates).  Two collectives: the counts, then the padded payloads."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    if not torch.is_tensor(local):
        local = torch.from_numpy(np.ascontiguousarray(local, dtype=np.uint8).reshape(-1, CAND_BYTES))
    local = local.to(dev)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    ns = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(ns, n, group=group)
    max_n = max(1, max(int(x.item()) for x in ns))
    pad = torch.full((max_n, CAND_BYTES), 0xFF, dtype=torch.uint8, device=dev)
    if local.shape[0]:
        pad[:local.shape[0]] = local
    out = torch.empty((world * max_n, CAND_BYTES), dtype=torch.uint8, device=dev)
    if hasattr(dist, "all_gather_into_tensor") and dev.type != "cpu":
        dist.all_gather_into_tensor(out, pad, group=group)
    else:
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        out = torch.cat(parts, dim=0)
    return out


def sharded_prefilter(ctx, q_from=0, q_to=None, group=None, device=None):
    """One target-sharded prefilter pass on this rank's context (targets staged with set_targets_shard*): candidates of
    the own shard -> all-gather (RCCL when `device` is the GPU: the payload goes HBM -> xGMI -> HBM) -> selection on the
    union for queries [q_from, q_to).  Afterwards ctx.hits_compact() / ctx.align_hits_raw() see the merged hits.
    Returns {"union": rows of the gathered list, "bytes": bytes this rank received, "exchange_s": wall time of the exchange
    step (counts + payload collectives, until the union is in place)}."""
    import time
    import torch
    n = ctx.prefilter_candidates()
    dev = device if device is not None else torch.device("cpu")
    local = torch.empty((n, CAND_BYTES), dtype=torch.uint8, device=dev)
    if n:
        ctx.candidates_into(local.data_ptr())          # synchronises the context's stream
    t0 = time.perf_counter()
    union = allgather_candidates(local, group=group, device=dev)
    if dev.type != "cpu":
        torch.cuda.synchronize(dev)                     # the collective ran on torch's stream
    dt = time.perf_counter() - t0
    if dev.type != "cpu":
        ctx.merge_candidates_device(union.data_ptr(), union.shape[0], q_from, q_to)
    else:
        ctx.merge_candidates(union.numpy(), q_from, q_to)
    return {"union": int(union.shape[0]), "bytes": int(union.shape[0]) * CAND_BYTES, "exchange_s": dt}
