#!/usr/bin/env python3
"""bench.py -- spacer-queries/s through prefilter+align on MI355X (BASELINE.json metric), one JSON line on rank 0.

A "step" is one pass of the hot path over one resident batch of synthetic spacer peptides:
    k_prep -> k_match -> k_score -> k_select -> SW forward -> E-value gate -> SW reverse + traceback -> record assembly
with the target index / residues / similar-k-mer tables and the query batch already resident in HBM; both results (the
prefilter hit list and the alignment records + backtraces) are delivered to host memory -- the hit list travels on a copy
stream while the align kernels run, the alignment records while the NEXT step's prefilter runs (sp_ctx_align_hits_begin /
_end); the last step's records are collected before the closing barrier, so all K steps complete inside the timed region.
The K steps are dealt to --pipeline (default 2) search contexts per GPU, each with its own HIP streams and host thread: the
matching kernel is latency-bound (a third of the VALU issue slots), the Smith-Waterman kernel VALU-bound, and two passes in
flight fill the machine better than one (two whole bench PROCESSES on one GPU reach 1.3x the throughput of one).

Workload (BASELINE.json configs[2], SURVEY.md section 8d "synthetic proxy"): 10 000 synthetic 30-nt spacers =
60 000 peptides of 10 aa (even index: window of a target with one substitution, odd: background i.i.d.) against
50 000 synthetic phage ORFs x 200 aa = 1e7 residues drawn from the VTML40 background, seed 12345.
N > 1 (torchrun): every rank holds the full target index and searches its own 10 000 spacers (weak scaling, query
partition like the reference's default QUERY_DB_SPLIT; no data-path collective) -- value = all spacers / max time.

--shard targets selects the target-sharded path instead (BASELINE.json configs[3]): the SAME 10 000
spacers on every rank, rank g indexes target range g, the per-shard candidates are all-gathered over RCCL/xGMI and each
rank finishes (selection on the union + align) its slice of the queries -- strong scaling, value = spacers / max time.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

AA = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
BG = np.array([0.0819, 0.0169, 0.0453, 0.0616, 0.0465, 0.0745, 0.0217, 0.0549, 0.0582, 0.0905, 0.0274, 0.0435,
               0.0506, 0.0375, 0.0447, 0.0533, 0.0569, 0.081, 0.0132, 0.0329])
BG = BG / BG.sum()


def make_workload(n_spacers, n_targets, target_len, seed, query_seed=None):
    """targets from `seed`; queries (half of them planted from THOSE targets) from `query_seed` (None: continue the target
    stream, the single-GPU workload) -- so every rank of a multi-GPU run searches its own spacers against the same phages"""
    rng = np.random.default_rng(seed)
    tres = AA[rng.choice(20, size=n_targets * target_len, p=BG)]
    toff = np.arange(n_targets + 1, dtype=np.uint64) * np.uint64(target_len)
    if query_seed is not None:
        rng = np.random.default_rng(query_seed)
    nq, ql = 6 * n_spacers, 10
    q = AA[rng.choice(20, size=nq * ql, p=BG)].reshape(nq, ql)
    tid = rng.integers(0, n_targets, size=nq)
    pos = rng.integers(0, target_len - ql + 1, size=nq)
    planted = tres.reshape(n_targets, target_len)[tid[:, None], pos[:, None] + np.arange(ql)[None, :]].copy()
    sub = rng.integers(0, ql, size=nq)
    planted[np.arange(nq), sub] = AA[rng.integers(0, 20, size=nq)]
    q[0::2] = planted[0::2]
    qoff = np.arange(nq + 1, dtype=np.uint64) * np.uint64(ql)
    return tres, toff, np.ascontiguousarray(q).reshape(-1), qoff


def u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def cpu_baseline_reference(tres, toff, qres, qoff, n_sample, procs, reps=5):
    """The REAL reference classes (oracle/_ref/ref_harness --bench: QueryMatcher::matchQuery + Matcher::getSWResult, AVX2,
    compiled from /root/reference by oracle/Makefile) on `procs` host cores: one single-threaded process per core, each
    searching its own slice of the sample against the full target set.  Only the search loops are timed (index build,
    similar-k-mer tables and the ALP fit are set-up, as for the GPU number) and they run `reps` times over the slice so that
    the sample is a few seconds per core; rate = sample / slowest process's mean time per pass."""
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import matrix_text
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    n_t, tl, ql = len(toff) - 1, int(toff[1]), int(qoff[1])
    per = (n_sample + procs - 1) // procs
    with tempfile.TemporaryDirectory() as d:
        m80, m40, tf = (os.path.join(d, x) for x in ("VTML80.out", "VTML40.out", "t.txt"))
        open(m80, "w").write(matrix_text("VTML80"))
        open(m40, "w").write(matrix_text("VTML40"))
        with open(tf, "wb") as f:
            f.write(b"\n".join(tres[i * tl:(i + 1) * tl].tobytes() for i in range(n_t)) + b"\n")
        jobs = []
        for p in range(procs):
            lo, hi = p * per, min(n_sample, (p + 1) * per)
            if lo >= hi:
                break
            qf, of = os.path.join(d, "q%d.txt" % p), os.path.join(d, "o%d.json" % p)
            with open(qf, "wb") as f:
                f.write(b"\n".join(qres[i * ql:(i + 1) * ql].tobytes() for i in range(lo, hi)) + b"\n")
            jobs.append((subprocess.Popen([harness, m80, m40, tf, qf, of, "--bench-reps", str(reps)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL), of))
        res = []
        for pr, of in jobs:
            if pr.wait() != 0:
                raise RuntimeError("ref_harness failed")
            res.append(json.load(open(of))["bench"])
    slowest = max(r["prefilter_s"] + r["align_s"] for r in res)
    return {"value": (n_sample / 6.0) / slowest, "unit": "spacer-queries/s", "cores": len(res), "kind": "reference",
            "pairs": sum(r["pairs"] for r in res),
            "sample": "%d of the workload's peptides (%d SW alignments) split over %d single-threaded processes of oracle/_ref/ref_harness "
                      "(reference QueryMatcher + Matcher classes, AVX2, g++ -O3); search loops only, %d passes each, slowest process %.2f s per pass "
                      "(prefilter %.2f s + align %.2f s)" % (n_sample, sum(r["pairs"] for r in res), len(res), reps, slowest,
                                                            max(r["prefilter_s"] for r in res), max(r["align_s"] for r in res))}


def cpu_baseline(tres, toff, qres, qoff, n_sample):
    """The oracle (scalar C port of the reference algorithm, one thread) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Alp, Oracle
    o = Oracle(sens=5.7, max_seqs=300)
    n_t = len(toff) - 1
    tl = int(toff[1])
    o.set_targets([tres[i * tl:(i + 1) * tl].tobytes() for i in range(n_t)])
    g = json.load(__import__("gzip").open(os.path.join(ROOT, "tests/golden/spacer.json.gz")))["ref"]
    o.set_alp(Alp.from_ref(g["alpGapped"]), Alp.from_ref(g["alpGapless"]))
    ql = int(qoff[1])
    t0 = time.perf_counter()
    pairs = 0
    for i in range(n_sample):
        q = qres[i * ql:(i + 1) * ql].tobytes().decode()
        for h in o.prefilter_list(q):
            o.align(q, h[0])
            pairs += 1
    dt = time.perf_counter() - t0
    o.close()
    return {"value": (n_sample / 6.0) / dt, "unit": "spacer-queries/s", "cores": 1, "kind": "port",
            "sample": "%d of the workload's peptides (prefilter + %d SW alignments), oracle/sp_oracle.c, single thread, "
                      "index build excluded" % (n_sample, pairs)}


def default_workload(args):
    return (args.spacers, args.targets, args.target_len) == (10000, 50000, 200)


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the newest committed counter pass (profiles/rNN_pmc_traffic.json, made by
    tools/gpu_pmc.sh on this same workload: FETCH_SIZE corrected by the calibration reads + WRITE_SIZE); None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        doc = json.load(open(files[-1]))
        k = doc["kernels"][kernel]
        pmc_traffic.measured_stream = doc.get("streaming_read_of_1GiB")
        return k["fetch_bytes_corrected"] + k.get("write_bytes_raw", 0.0), "profiles/" + os.path.basename(files[-1])
    except (KeyError, ValueError):
        return None, None


pmc_traffic.measured_stream = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--spacers", type=int, default=10000)
    ap.add_argument("--targets", type=int, default=50000)
    ap.add_argument("--target-len", type=int, default=200)
    ap.add_argument("--stage", default="full", choices=["full", "prefilter"])
    ap.add_argument("--cpu-sample", type=int, default=3000, help="peptides per core for the CPU baseline (0 = skip)")
    ap.add_argument("--cpu-cores", type=int, default=8, help="host cores for the reference CPU baseline (one process each)")
    ap.add_argument("--pipeline", type=int, default=2,
                    help="search contexts per GPU, each on its own HIP streams and host thread, taking the steps in turn: the latency-bound "
                         "matcher of one pass and the VALU-bound aligner of another share the machine (1 = strictly one pass at a time)")
    ap.add_argument("--shard", default="queries", choices=["queries", "targets"],
                    help="N>1 decomposition: query partition with replicated index (default) or target shards + RCCL all-gather")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    sharded = args.shard == "targets"
    if world > 1 or sharded:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if world == 1:      # single-GPU run of the sharded path: a one-rank RCCL group
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1)
        else:
            dist.init_process_group("nccl")

    import spacepharer_amd as sa
    from spacepharer_amd.dist import decompose_by_residues, sharded_prefilter
    if sharded:
        tres, toff, qres, qoff = make_workload(args.spacers, args.targets, args.target_len, 12345)
    else:
        # same target set on every rank; rank r > 0 searches its own spacers (planted from the common targets)
        tres, toff, qres, qoff = make_workload(args.spacers, args.targets, args.target_len, 12345, None if rank == 0 else 12345 + 7919 * rank)
    ctx = sa.SearchContext(device_mask=1 << local_rank)
    nq = len(qoff) - 1
    q_from, q_to = 0, nq
    t0 = time.perf_counter()
    if sharded:
        t_from, t_cnt = decompose_by_residues(np.diff(toff).astype(np.int64) + 2, rank, world)
        ctx.set_targets_shard_raw(tres.tobytes(), u64p(toff), len(toff) - 1, t_from, t_from + t_cnt)
        q_from, q_cnt = decompose_by_residues(np.diff(qoff).astype(np.int64) + 2, rank, world)
        q_to = q_from + q_cnt
    else:
        ctx.set_targets_raw(tres.tobytes(), u64p(toff), len(toff) - 1)
    t_index = time.perf_counter() - t0
    ctx.set_queries_raw(qres.tobytes(), u64p(qoff), nq)
    gpu_dev = None
    if dist is not None:
        gpu_dev = torch.device("cuda", local_rank)
    # further contexts of the pipeline: same resident targets and queries, own streams (the exchange step of the target-sharded
    # path is a collective on the process group: one context there)
    n_pipe = 1 if sharded else max(1, min(args.pipeline, args.steps))
    ctxs = [ctx]
    for _ in range(n_pipe - 1):
        c2 = sa.SearchContext(device_mask=1 << local_rank)
        c2.set_targets_raw(tres.tobytes(), u64p(toff), len(toff) - 1)
        c2.set_queries_raw(qres.tobytes(), u64p(qoff), nq)
        ctxs.append(c2)

    def step(c=ctx):
        if sharded:
            sharded_prefilter(c, q_from, q_to, device=gpu_dev)   # own shard -> RCCL all-gather -> selection on the union
        else:
            c.prefilter()
        npairs = c.num_hits()
        if args.stage == "full" and npairs:
            c.align_hits_begin()                          # alignment result DB content (result_t + backtraces per hit): returns when
                                                          # the kernels are done, the 5 MB travel while the next pass's prefilter runs
        c.hits_view()                                     # prefilter result DB content (hit_t triples per query) in host memory: its
        return npairs                                     # copy started underneath the align kernels

    def barrier():
        for c in ctxs:
            c.align_hits_end()                            # the last pass's alignment records are in host memory
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        for c in ctxs:
            c.sync()

    if n_pipe > 1:
        for c in ctxs:
            c.set_tuning("match_blocks_per_cu", 2)      # leaves LDS for the other context's aligner on every CU (14.7 M against 13.6 M)
    for c in ctxs:
        for _ in range(args.warmup):
            step(c)
    # timed region: HIP events bracket only the dominant kernel (k_match, the roofline figure) -- every event record between
    # two kernels costs ~6 us of idle GPU, so the per-stage breakdown is taken in a separate short pass after the clock stops.
    # The K steps are dealt to the contexts of the pipeline; each context runs its share on its own host thread (ctypes
    # releases the GIL for the duration of a call), so up to n_pipe passes are in flight on the GPU at any time.
    import threading
    for c in ctxs:
        c.set_timing(1)
    share = [args.steps // n_pipe + (1 if i < args.steps % n_pipe else 0) for i in range(n_pipe)]
    acc = [[0.0, 0] for _ in range(n_pipe)]          # per context: sum of k_match event times, pairs of its last step

    def run(i):
        c = ctxs[i]
        for _ in range(share[i]):
            acc[i][1] = step(c)
            v = c.stat("ms_match")
            acc[i][0] += 0.0 if v != v else v

    barrier()
    t0 = time.perf_counter()
    if n_pipe == 1:
        run(0)
    else:
        threads = [threading.Thread(target=run, args=(i,)) for i in range(n_pipe)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
    barrier()
    dt = time.perf_counter() - t0
    ms_match_sum = sum(a[0] for a in acc)
    pairs = acc[0][1]
    # per-stage breakdown: one context alone, HIP events around every stage, outside the timed region
    stats = {"ms_match": 0.0, "ms_prep": 0.0, "ms_score": 0.0, "ms_select": 0.0, "ms_sw_forward": 0.0, "ms_sw_reverse": 0.0, "ms_traceback": 0.0}
    ctx.set_timing(2)
    ctx.set_tuning("match_blocks_per_cu", 0)           # alone on the GPU the matcher wants all the LDS
    n_breakdown = 3
    for _ in range(n_breakdown):
        step()
        for k in stats:
            v = ctx.stat(k)
            stats[k] += 0.0 if v != v else v
    barrier()
    ctx.set_timing(1)
    if dist is not None:
        t = torch.tensor([dt, float(pairs)], device="cuda")
        dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
        dt, pairs = float(t[0].item()), int(t[1].item())
        dist.destroy_process_group()
    if rank != 0:
        return
    K, M, D, R, H = (ctx.stat(k) for k in ("kmers", "db_matches", "diagonals", "diag_residues", "hits"))
    ms_match = ms_match_sum / args.steps      # HIP events on the context's stream, inside the timed region
    # algorithmic bytes of the dominant kernel (k_match) per launch: 6 B similar-k-mer table entry + 16 B offsets pair
    # per similar k-mer, 6 B per index entry fetched, 7 B per double-hit diagonal written
    algo_bytes = 22.0 * K + 6.0 * M + 7.0 * D
    achieved = algo_bytes / (ms_match * 1e-3) / 1e9 if ms_match > 0 else 0.0
    traffic, traffic_src = pmc_traffic("k_match") if default_workload(args) else (None, None)
    out = {
        "metric": "spacer-queries/sec through prefilter+align; achieved HBM GB/s vs roofline",
        "value": args.spacers * (1 if sharded else world) * args.steps / dt,
        "unit": "spacer-queries/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
        "dtype": "u8/i16", "data": "synthetic",
        "config": {"workload": "%d synthetic 30-nt spacers (= %d peptides of 10 aa) vs %d synthetic phage ORFs x %d aa "
                               "(%.1e residues), 1xMI355X per rank, %s" % (args.spacers, nq, args.targets, args.target_len,
                                                                         args.targets * args.target_len,
                                                                         "prefilter + ungapped + banded-SW align" if args.stage == "full" else "prefilter only"),
                   "stage": args.stage, "pairs_aligned_per_step": pairs, "sensitivity": 5.7, "max_seqs": 300,
                   "parallelism": ("target shards x%d, RCCL all-gather of candidates, selection + align by query slice" % world) if sharded
                                  else ("query partition x%d, index replicated" % world),
                   "passes_in_flight_per_gpu": n_pipe,
                   "index_build_s": round(t_index, 3)},
        "kernels_ms_per_step": {k[3:]: round(v / n_breakdown, 4) for k, v in stats.items()},
        "kernels_ms_source": "HIP events around every stage in %d extra steps of ONE context AFTER the timed region (the timed steps bracket k_match only: "
                             "%.4f ms per launch with %d passes in flight)" % (n_breakdown, ms_match, n_pipe),
        "counters_per_step": {"similar_kmers": K, "index_entries_fetched": M, "diagonals_scored": D, "diag_residues": R, "hits": H},
        "roofline": {"bound": "hbm", "kernel": "k_match", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                     "frac": achieved / 8000.0, "traffic": traffic,
                     "traffic_source": traffic_src, "algorithmic_bytes_per_launch": algo_bytes,
                     "measured_streaming_read_GBps": pmc_traffic.measured_stream,
                     # the same kernel with the GPU to itself (one context, the breakdown pass after the timed region): its launches in the
                     # timed region share the machine with the other context's aligner and take longer each, while more work gets done
                     "exclusive": {"ms_per_launch": round(stats["ms_match"] / n_breakdown, 4),
                                   "achieved": algo_bytes / (stats["ms_match"] / n_breakdown * 1e-3) / 1e9 if stats["ms_match"] > 0 else 0.0,
                                   "frac": algo_bytes / (stats["ms_match"] / n_breakdown * 1e-3) / 1e9 / 8000.0 if stats["ms_match"] > 0 else 0.0},
                     "passes_in_flight": n_pipe},
    }
    if args.cpu_sample > 0 and world == 1:      # rank 0 at N = 1 only
        # the reference itself when its harness travelled with the repo (oracle/_ref, built by __graft_entry__.build()), else
        # the scalar port; either way a bounded sample
        cores = max(1, min(args.cpu_cores, os.cpu_count() or 1))
        if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_harness")):
            try:
                n_s = nq if args.cpu_sample >= 3000 else min(args.cpu_sample * cores, nq)
                out["cpu_baseline"] = cpu_baseline_reference(tres, toff, qres, qoff, n_s, cores)
                if n_s == nq and not sharded and world == 1:     # whole workload: the reference must align exactly the pairs the GPU aligned
                    out["cpu_baseline"]["pairs_equal_gpu"] = bool(out["cpu_baseline"].pop("pairs") == pairs)
            except Exception as e:      # noqa: BLE001 -- fall back to the port, say why
                out["cpu_baseline"] = cpu_baseline(tres, toff, qres, qoff, min(args.cpu_sample, nq))
                out["cpu_baseline"]["sample"] += " (reference harness unavailable: %s)" % e
        else:
            out["cpu_baseline"] = cpu_baseline(tres, toff, qres, qoff, min(args.cpu_sample, nq))
    C.CDLL(None).fflush(None)      # library chatter buffered in C stdio (RCCL prints its path) goes out BEFORE the result line
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
