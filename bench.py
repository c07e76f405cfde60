#!/usr/bin/env python3
"""bench.py -- spacer-queries/s through prefilter+align on MI355X (BASELINE.json metric), one JSON line on rank 0.

A "step" is one pass of the hot path over one resident batch of synthetic spacer peptides:
    k_prep -> k_match -> k_score -> k_select -> SW forward -> E-value gate -> SW reverse + traceback -> record assembly
with the target index / residues / similar-k-mer tables and the query batches already resident in HBM; both results (the
prefilter hit list and the alignment records + backtraces) are delivered to host memory.  By default the step is
DEVICE-DRIVEN (sp_ctx_search_begin / _end): every count stays in HBM, all kernels of the step and the four result copies are
queued without the host learning anything in between, two steps per context may be in flight (--in-flight), and a step whose
outputs outgrow the capacities is repeated the ordinary way inside _end ("steps_repeated" in the line; 0 on this workload).
--host-driven times the round-2 sequence instead (sp_ctx_prefilter, sp_ctx_align_hits_begin / _end: the host reads the hit
count between the two halves).  With no mode asked for (N = 1) BOTH are measured -- host-driven with four contexts first, then
device-driven with four -- and `value` is the host-driven one, a fixed configuration ("step_modes" holds both, with the k_match launch time of each; since
round 6 -- a step's results leave the device when its kernels are done instead of being queued behind events -- the two are level).
All K steps complete inside the timed region.
Every step delivers into its OWN page-locked buffers; after the clock has stopped every delivered result (hit list,
offsets, alignment records, backtraces) is digested and must equal the digest of the same batch searched by one context
alone -- a wrong byte anywhere fails the run.

N = 1 (BASELINE.json configs[2], SURVEY.md section 8d "synthetic proxy"): 10 000 synthetic 30-nt spacers = 60 000 peptides of
10 aa (even index: window of a target with one substitution, odd: background i.i.d.) against 50 000 synthetic phage ORFs x
200 aa = 1e7 residues drawn from the VTML40 background, seed 12345.  The K steps are dealt to --pipeline (default 2) search
contexts, each with its own HIP streams, host thread and its own DISTINCT query batch (so consecutive steps search different
spacers): the matching kernel is latency-bound, the Smith-Waterman kernel VALU-bound, two passes in flight fill the machine
better than one.  The K-step region is repeated --regions times (default 5); `value` is the median region, the spread is
reported next to it.

N > 1 (BASELINE.json configs[3]): the same 100 000 spacers (600 000 peptides) against the same phages, strong scaling, value =
spacers / max time over ranks.  The job is measured under every decomposition the path offers and `value` is the best of them
(`parallelism` names it, `decompositions` holds all): "target shards" (rank g indexes target range g, all ranks match all queries
against their shard, the per-shard candidates are all-gathered over RCCL/xGMI, each rank finishes -- selection on the union +
align -- its slice of the queries), "hybrid" at N >= 4 (2 target shards x N / 2 query groups; --shards S chooses any split: the
similar-k-mer enumeration is replicated per SHARD), and "query partition" (every rank holds the full index and searches 1/N of
the spacers; the hit lists and alignment records are all-gathered so that every rank ends with the whole result, the exchange
of one batch running beside the search of the next; measured with host-driven steps in four contexts per rank first and with
device-driven steps in two, the better one counts: "query_partition.step_modes").  Key "replica" is the weak-scaling side figure (every rank its own 10 000
spacers, no collective).
`python bench.py --gpus N` starts the N ranks itself (child processes, started before anything touches the GPU); under
torchrun it uses the ranks it is given.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

AA = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
BG = np.array([0.0819, 0.0169, 0.0453, 0.0616, 0.0465, 0.0745, 0.0217, 0.0549, 0.0582, 0.0905, 0.0274, 0.0435,
               0.0506, 0.0375, 0.0447, 0.0533, 0.0569, 0.081, 0.0132, 0.0329])
BG = BG / BG.sum()


def make_workload(n_spacers, n_targets, target_len, seed, query_seed=None):
    """targets from `seed`; queries (half of them planted from THOSE targets) from `query_seed` (None: continue the target
    stream, the single-GPU workload) -- so every batch / rank searches its own spacers against the same phages"""
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


def host_info():
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    avail = None
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable"):
                avail = int(ln.split()[1]) * 1024
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    quota = None      # cgroup CPU quota of this container (cpu.max: "<quota> <period>" in microseconds, or "max")
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            f = open(path).read().split()
            if path.endswith("cpu.max"):
                quota = None if f[0] == "max" else float(f[0]) / float(f[1])
            else:
                q = float(f[0])
                quota = None if q <= 0 else q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    cores = usable if quota is None else max(1, min(usable, int(quota + 0.5)))
    return {"nproc": os.cpu_count() or 1, "usable_cores": usable, "cpu_quota_cores": quota, "cores_for_cpu_baseline": cores, "cpu_model": model, "mem_available_bytes": avail}


def cpu_baseline_reference(tres, toff, qres, qoff, n_sample, threads, target_s=4.0, want_results=False):
    """The REAL reference classes (oracle/_ref/ref_harness --bench: QueryMatcher::matchQuery + Matcher::getSWResult, AVX2,
    compiled from /root/reference by oracle/Makefile) on `threads` host cores: ONE process whose worker threads share the index
    table, the sequence store and the similar-k-mer tables and own a QueryMatcher / Matcher each, like the reference's OpenMP
    regions; thread t searches query slice t.  Only the search loops are timed (index build, similar-k-mer tables and the ALP
    fit are set-up, as for the GPU number); every thread walks its slice `reps` times so that the timed part lasts a few
    seconds; rate = sample / wall time per pass.  (Round 1 ran one single-threaded PROCESS per core: with 256 of them, each
    with its own 1 GB of tables, the cores starve each other -- 0.5 k peptides/s per core instead of 15 k.)"""
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import matrix_text
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    n_t, tl, ql = len(toff) - 1, int(toff[1]), int(qoff[1])
    per = (n_sample + threads - 1) // threads
    reps = int(max(3, min(2000, round(target_s / max(1e-3, per * 1.0e-4)))))      # ~0.1 ms per peptide and core
    with tempfile.TemporaryDirectory() as d:
        m80, m40, tf, qf, of = (os.path.join(d, x) for x in ("VTML80.out", "VTML40.out", "t.txt", "q.txt", "o.json"))
        open(m80, "w").write(matrix_text("VTML80"))
        open(m40, "w").write(matrix_text("VTML40"))
        with open(tf, "wb") as f:
            f.write(b"\n".join(tres[i * tl:(i + 1) * tl].tobytes() for i in range(n_t)) + b"\n")
        with open(qf, "wb") as f:
            f.write(b"\n".join(qres[i * ql:(i + 1) * ql].tobytes() for i in range(n_sample)) + b"\n")
        df = os.path.join(d, "dump.txt")
        subprocess.run([harness, m80, m40, tf, qf, of, "--bench-reps", str(reps), "--bench-threads", str(threads)] + (["--bench-dump", df] if want_results else []),
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        r = json.load(open(of))["bench"]
        results = open(df).read().split("\n")[:n_sample] if want_results else None
    t_pass = r["prefilter_s"] + r["align_s"]
    # what a `prefilter` + `align` module pair of the reference pays for this sample on the same threads: the index is rebuilt in every prefilter
    # process (tantan masking + IndexTable, Prefiltering.cpp:499-537) and the similar-k-mer tables are computed at start-up; DB I/O and text formatting
    # are not in the harness and not counted
    t_module = r["index_build_s"] + r["table_s"] + t_pass
    return {"value": (n_sample / 6.0) / t_pass, "unit": "spacer-queries/s", "cores": r["threads"], "kind": "reference", "pairs": r["pairs"], "results": results,
            "index_build_s": r["index_build_s"], "kmer_tables_s": r["table_s"], "search_loops_s": t_pass, "module_wall_s": t_module,
            "module_value": (n_sample / 6.0) / t_module,
            "sample": "%d of the workload's peptides (%d SW alignments) in one process of oracle/_ref/ref_harness (reference QueryMatcher + Matcher classes, AVX2, "
                      "g++ -O3) with %d worker threads sharing one index; value = search loops only, %d passes, %.4f s per pass (prefilter %.4f s + align %.4f s); "
                      "module_value additionally counts what every prefilter process of the reference pays: index build %.3f s (masking + both index passes + list sort, "
                      "same threads) + similar-k-mer tables %.3f s" % (n_sample, r["pairs"], r["threads"], reps, t_pass, r["prefilter_s"], r["align_s"], r["index_build_s"], r["table_s"])}


def reference_results_equal_gpu(tres, toff, qres, qoff, n_sample, results):
    """The hit lists and alignments the reference classes produced in the timed loops of cpu_baseline_reference() (ref_harness --bench-dump: per query, per hit
    seqId, prefilter score, diagonal | bit score, starts, ends, length, E-value as a hex double, accepted, backtrace) against one search of the same sampled
    peptides on the GPU: every field, bit for bit.  Returns a small report; "equal" is what the bench line carries."""
    import spacepharer_amd as sa
    ql = int(qoff[1])
    ctx = sa.SearchContext()
    try:
        ctx.set_targets_raw(tres.tobytes(), u64p(toff), len(toff) - 1)
        sub_off = (np.arange(n_sample + 1, dtype=np.uint64) * np.uint64(ql))
        ctx.set_queries_raw(qres[:n_sample * ql].tobytes(), u64p(sub_off), n_sample)
        ctx.prefilter()
        hits = ctx.hits()
        pairs = [(q, h[0]) for q in range(n_sample) for h in hits[q]]
        aln = ctx.align(pairs) if pairs else []
        k = n_hits = n_aln = 0
        for q in range(n_sample):
            exp = [e for e in results[q].split(";") if e]
            if len(exp) != len(hits[q]):
                return {"equal": False, "first_difference": "query %d: %d hits on the GPU, %d in the reference" % (q, len(hits[q]), len(exp))}
            for h, e in zip(hits[q], exp):
                f = e.split(",")
                res, cig = aln[k]
                k += 1
                n_hits += 1
                if [int(h[0]), int(h[1]), int(h[2])] != [int(f[0]), int(f[1]), int(f[2])]:
                    return {"equal": False, "first_difference": "query %d: hit %s on the GPU, %s in the reference" % (q, list(h), f[:3])}
                ok = int(f[10])
                same = (res.bit_score, res.q_end, res.db_end, int(res.passed)) == (int(f[3]), int(f[5]), int(f[7]), ok)
                if same and ok:
                    same = (res.q_start, res.db_start, res.aln_len) == (int(f[4]), int(f[6]), int(f[8])) and res.evalue == float.fromhex(f[9]) and cig == f[11]
                    n_aln += 1
                if not same:
                    return {"equal": False, "first_difference": "query %d target %d: alignment differs (%s)" % (q, int(h[0]), e[:120])}
        return {"equal": True, "peptides": n_sample, "hits_compared": n_hits, "accepted_alignments_compared": n_aln}
    finally:
        ctx.close()


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


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the newest committed counter pass (profiles/rNN_pmc_traffic.json, made by
    tools/gpu_pmc.sh on this same workload: FETCH_SIZE corrected by the calibration reads + WRITE_SIZE); None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        doc = json.load(open(files[-1]))
        k = doc["kernels"][kernel]
        pmc_traffic.measured_stream = doc.get("streaming_read_of_1GiB")
        # the counter pass names the device sources it was collected on (tools/pmc_summary.py: binary_identity); another tree's kernels -> stale
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from pmc_summary import binary_identity
        here, there = binary_identity(), doc.get("binary") or {}
        pmc_traffic.stale = there.get("device_sources_sha256") != here["device_sources_sha256"]
        pmc_traffic.collected_on = there.get("device_sources_sha256")
        return k["fetch_bytes_corrected"] + k.get("write_bytes_raw", 0.0), "profiles/" + os.path.basename(files[-1])
    except (KeyError, ValueError):
        return None, None


pmc_traffic.measured_stream = None
pmc_traffic.stale = None
pmc_traffic.collected_on = None


# ---------------------------------------------------------------------------------------------------------------------
# result slots: page-locked buffers one step delivers into, and their digest
# ---------------------------------------------------------------------------------------------------------------------
REC_BYTES = 64      # sizeof(sp_alignment): 60 bytes of fields + 4 bytes of tail padding (never compared)


def record_fields(addr, n):
    """the 60 field bytes of n sp_alignment records (the 4 bytes of tail padding per record are not part of the result)"""
    a = np.frombuffer(C.string_at(addr, REC_BYTES * n), dtype=np.uint32).reshape(n, REC_BYTES // 4)
    return np.ascontiguousarray(a[:, :15]).tobytes()


class Slot:
    def __init__(self, L, nq, cap_hits, full):
        self.L, self.nq, self.cap = L, nq, cap_hits
        self.hits = L.sp_host_alloc(8 * cap_hits)
        self.off = L.sp_host_alloc(4 * (nq + 1))
        self.rec = L.sp_host_alloc(REC_BYTES * cap_hits) if full else None
        self.cig_cap = 24 * cap_hits
        self.cig = L.sp_host_alloc(self.cig_cap) if full else None
        self.used = C.c_uint64(0)
        self.n = 0                      # hits (= pairs aligned) of the step that last delivered here
        self.hits_ptr_ok = True
        self.into = (self.rec, self.cig, self.cig_cap, self.used) if full else None

    def digest(self, full):
        """crc32 over everything the step delivered: offsets, hit records, alignment records, backtrace pool"""
        n = self.n
        h = zlib.crc32(C.string_at(self.off, 4 * (self.nq + 1)))
        h = zlib.crc32(C.string_at(self.hits, 8 * n), h)
        if full and n:
            h = zlib.crc32(record_fields(self.rec, n), h)
            h = zlib.crc32(C.string_at(self.cig, int(self.used.value)), h)
        return h

    def free(self):
        for p in (self.hits, self.off, self.rec, self.cig):
            if p:
                self.L.sp_host_free(p)
        self.hits = self.off = self.rec = self.cig = None


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes (nothing in this process has touched
    the GPU), rank 0 prints the result line on the inherited stdout."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import binascii
    from spacepharer_amd import dist as spd
    uid = binascii.hexlify(spd.unique_id()).decode()      # (makes the id only: nothing here touches a GPU)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SP_DIST_ID=uid,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def probe_cut_proxy(*a, **kw):
    """the one-GPU proxies of the target-sharded passes (side figures of the line): tools/bench_extras.py"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_extras
    return bench_extras.probe_cut_proxy(*a, **kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None)
    ap.add_argument("--steps", type=int, default=100, help="steps per timed region (default 100: a region of the default workload lasts ~50 ms)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--regions", type=int, default=5, help="the K-step timed region is repeated this many times; value = median region")
    ap.add_argument("--spacers", type=int, default=None, help="default: 10 000 (N = 1, replica mode), 100 000 (target shards at N > 1)")
    ap.add_argument("--targets", type=int, default=50000)
    ap.add_argument("--target-len", type=int, default=200)
    ap.add_argument("--stage", default="full", choices=["full", "prefilter"])
    ap.add_argument("--cpu-sample", type=int, default=3000, help="0 = skip the CPU baseline; < 3000: that many peptides per core instead of the whole workload")
    ap.add_argument("--cpu-cores", type=int, default=0, help="host cores for the reference CPU baseline (one process each); 0 = all usable cores")
    ap.add_argument("--match-blocks", type=int, default=0, help="matching-kernel blocks per CU while several search contexts share the GPU (0 = the library's own choice: two for the grouped kernel; -1 = all that fit); "
                    "the small footprint fits four, three leave room for the other context's aligner")
    ap.add_argument("--tune", action="append", default=[], metavar="NAME=VALUE", help="sp_ctx_set_tuning(name, value) on every search context of the timed region (experiments)")
    ap.add_argument("--pipeline", type=int, default=None,
                    help="search contexts per GPU, each on its own HIP streams and host thread with its own query batch, taking the steps in turn "
                         "(1 = strictly one pass at a time)")
    ap.add_argument("--shard", default=None, choices=["queries", "targets", "probes"],
                    help="decomposition: target shards + RCCL all-gather (default at N > 1) or query partition with replicated index (default at N = 1); \"probes\" (or SP_SHARD=probes "
                         "in the environment): target shards with the probe-list cut -- every rank enumerates its query slice once and sends the occupied probes to the shards (opt-in)")
    ap.add_argument("--in-flight", type=int, default=2, choices=[1, 2], help="device-driven steps queued per context before the oldest is collected")
    ap.add_argument("--host-driven", action="store_true", help="the round-2 step: the host learns the hit count between prefilter and align and the record count after align "
                    "(sp_ctx_prefilter + sp_ctx_align_hits_begin / _end) instead of one device-driven step (sp_ctx_search_begin / _end)")
    ap.add_argument("--no-dense", action="store_true", help="N = 1, default workload: skip the additional line on the dense index (10 000 spacers vs 2e8 residues, the density of GenBank_phage_2018_09)")
    ap.add_argument("--unshared", action="store_true", help="every search context of a GPU stages its own copy of the target set and tables (the round-3 configuration; default: one staged set, shared)")
    ap.add_argument("--no-module-e2e", action="store_true", help="N = 1: skip the files-to-files timing of the module drop-ins (tools/module_e2e.py)")
    ap.add_argument("--shards", type=int, default=0, help="N > 1, target-sharded: S target shards x N / S query groups (all-gather within a group); 0 = N shards, every rank matches all queries (BASELINE configs[3])")
    ap.add_argument("--no-probe-proxy", action="store_true", help="N = 1, default workload: skip the one-GPU measurement of the probe-list cut (shard 0 of 8 at 100 000 spacers: plain pass against enumerate + place + fetch)")
    ap.add_argument("--no-hybrid", action="store_true", help="N >= 4: skip the additional 2-shard x N/2-group measurement")
    ap.add_argument("--no-replica", action="store_true", help="N > 1: skip the additional query-partition (replica) measurement")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "host", "gloo"], help="transport of the all-gathers: RCCL (HBM -> xGMI -> HBM), or through host memory and the library's bootstrap sockets (\"host\"; \"gloo\" is the old name) for ranks that share a GPU (tests)")
    ap.add_argument("--stagger-us", type=int, default=0, help="host-driven steps: context i enters a timed region i x this many microseconds after context 0 (experiment)")
    ap.add_argument("--strict-rccl", action="store_true", help="N > 1: fail instead of falling back to the host transport when RCCL cannot be set up on every rank (exchange.rccl_ranks == world or no line)")
    ap.add_argument("--same-gpu", action="store_true", help="all ranks use HIP device 0 (tests of the N > 1 path on a one-GPU box; needs --exchange gloo)")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if args.exchange == "gloo":
        args.exchange = "host"
    if args.same_gpu and args.exchange != "host":
        sys.exit("bench.py: --same-gpu needs --exchange host (RCCL refuses two ranks on one device)")
    if env_world is None and (args.gpus or 1) > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(env_world or "1")
    local_rank = 0 if args.same_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus is not None and args.gpus != world:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (start it without a launcher, or make the two agree)" % (args.gpus, world))
    mode = args.shard or (os.environ.get("SP_SHARD") if os.environ.get("SP_SHARD") in ("queries", "targets", "probes") else None) or ("targets" if world > 1 else "queries")
    probe_cut = mode == "probes"
    # N > 1 without an explicit --shard / SP_SHARD: the target-sharded pass is chosen by the library's rule from the shards' measured occupancy and the number of ranks
    # (spacepharer_amd.dist.choose_sharded_pass: the probe-list cut on sparse shards from four ranks on, the plain pass otherwise)
    auto_pass = args.shard is None and os.environ.get("SP_SHARD") not in ("queries", "targets", "probes")
    rule = {"on": probe_cut, "verdict": None, "occupancy": None}
    sharded = mode in ("targets", "probes")
    n_spacers = args.spacers or (100000 if (sharded and world > 1) else 10000)
    full = args.stage == "full"

    import spacepharer_amd as sa
    from spacepharer_amd import dist as spd
    from spacepharer_amd.dist import decompose_by_residues, sharded_prefilter
    L = sa.load_library()
    # N > 1: every collective of the data path is the library's (sp_dist_* / sp_ctx_allgather_*: RCCL all-gather of device buffers, counts and timings over its bootstrap
    # sockets).  The ranks' 128-byte id comes from SP_DIST_ID (this script's own launcher) or through torch.distributed's gloo store (torchrun) -- control plane only.
    comm = None
    if world > 1 or sharded:
        L.sp_init(1 << local_rank)      # this rank's GPU, before the communicator binds to it
        args.exchange_requested = args.exchange
        comm = spd.init_from_env(spd.TRANSPORT_RCCL if args.exchange == "rccl" else spd.TRANSPORT_HOST, rank=rank, world=world)
        if args.exchange == "rccl" and comm.transport != spd.TRANSPORT_RCCL:
            if args.strict_rccl:
                raise SystemExit("bench.py --strict-rccl: the ranks could not set up RCCL and agreed on the host transport (reason above); nothing was measured")
            args.exchange = "host"      # (all ranks agreed on the host transport: RCCL could not be set up on one of them -- the reason is on stderr; the line below says which transport ran)

    def all_max(x):
        return x if comm is None or world == 1 else comm.allreduce([x], "max")[0]

    def all_sum(x):
        return x if comm is None or world == 1 else comm.allreduce([x], "sum")[0]

    def measure(shard_targets, spacers, steps, regions, warmup, want_breakdown, shards=0, partition=False, pipeline=None, host_driven=None, targets=None):
        """one decomposition on this rank: set-up, reference digests, `regions` timed K-step regions with digest checks.
        shards (target-sharded mode): S target shards x world / S query groups -- group g = rank // S searches query slice g against the
        S shards and all-gathers within the group; 0 / world = every rank is a shard and matches all queries (BASELINE configs[3])."""
        pipeline = (args.pipeline or 2) if pipeline is None else pipeline
        host_driven = args.host_driven if host_driven is None else host_driven
        n_pipe = 1 if shard_targets else max(1, min(pipeline, steps))
        unshared = args.unshared
        # partition (N > 1, index replicated): the SAME job as the sharded one -- `spacers` spacers against the whole target set -- cut by QUERY: rank r
        # searches slice r of every batch against the full index in its own HBM and the per-rank hit lists are all-gathered (RCCL) after every step
        S = (shards or world) if shard_targets else 1
        G = world // S if shard_targets else 1
        sub = None
        if shard_targets and G > 1:
            assert world % S == 0, "--shards must divide the number of ranks"
            sub = comm.split(rank // S)      # the S shards of this rank's query group
        # batch b is searched by context b: batch 0 = the single-GPU workload's spacers; further batches / ranks draw their own
        batches = []
        for b in range(n_pipe):
            if shard_targets:
                qs = None                                         # the SAME spacers on every rank
            elif partition:
                qs = None if b == 0 else 12345 + 104729 * b       # the same batches on every rank, each rank keeps its slice
            else:
                qs = None if (rank == 0 and b == 0) else 12345 + 7919 * rank + 104729 * b
            batches.append(make_workload(spacers, targets or args.targets, args.target_len, 12345, qs))
        tres, toff = batches[0][0], batches[0][1]
        part_from, part_total = 0, 0
        if partition and world > 1:
            parts = []
            for b in range(n_pipe):
                tr, to, qr, qo = batches[b]
                p_from, p_cnt = decompose_by_residues(np.diff(qo).astype(np.int64) + 2, rank, world)
                part_from, part_total = p_from, len(qo) - 1
                parts.append((p_from, p_cnt, part_total))
                lo, hi = int(qo[p_from]), int(qo[p_from + p_cnt])
                batches[b] = (tr, to, qr[lo:hi].copy(), (qo[p_from:p_from + p_cnt + 1] - qo[p_from]).astype(qo.dtype))
            assert len({len(bb[3]) for bb in batches}) == 1 and len(set(parts)) == 1, "the batches decompose differently: exchange_hits() passes ONE (part_from, part_total) for every slot"
        if shard_targets and G > 1:      # this rank's group holds query slice g of every batch only
            for b in range(n_pipe):
                tr, to, qr, qo = batches[b]
                g_from, g_cnt = decompose_by_residues(np.diff(qo).astype(np.int64) + 2, rank // S, G)
                lo, hi = int(qo[g_from]), int(qo[g_from + g_cnt])
                batches[b] = (tr, to, qr[lo:hi].copy(), (qo[g_from:g_from + g_cnt + 1] - qo[g_from]).astype(qo.dtype))
        nq = len(batches[0][3]) - 1
        q_from, q_to = 0, nq
        ctxs = []
        t0 = time.perf_counter()
        for b in range(n_pipe):
            c = sa.SearchContext(device_mask=1 << local_rank)
            if shard_targets:
                t_from, t_cnt = decompose_by_residues(np.diff(toff).astype(np.int64) + 2, rank % S, S)
                if b == 0 or unshared:
                    c.set_targets_shard_raw(tres.tobytes(), u64p(toff), len(toff) - 1, t_from, t_from + t_cnt)
                else:
                    c.share_targets(ctxs[0])
                q_from, q_cnt = decompose_by_residues(np.diff(batches[b][3]).astype(np.int64) + 2, rank % S, S)
                q_to = q_from + q_cnt
            elif b == 0 or unshared:
                c.set_targets_raw(tres.tobytes(), u64p(toff), len(toff) - 1)
            else:
                c.share_targets(ctxs[0])      # ONE staged target set, several search contexts (streams, queries, result buffers of their own)
            if b == 0:
                t_index = time.perf_counter() - t0
            c.set_queries_raw(batches[b][2].tobytes(), u64p(batches[b][3]), nq)
            ctxs.append(c)
        ctx = ctxs[0]
        if shard_targets and auto_pass and world > 1:
            rule["verdict"], rule["occupancy"] = spd.agree_sharded_pass(ctxs[0], sub or comm, int(np.diff(batches[0][3]).max()))
            rule["on"] = rule["verdict"] == "probes"
        xch = {"s": 0.0, "bytes": 0, "n": 0, "probe_s": 0.0}
        cut_on = [False]      # (the probe-list cut: switched on AFTER the reference digests, which come from every shard enumerating every query itself)

        def search(c):
            if shard_targets:
                try:
                    info = sharded_prefilter(c, sub or comm, q_from, q_to, probes=rule["on"] and cut_on[0])     # own shard -> all-gather (within the query group) -> selection on the union
                except sa.SpError as e:
                    # the probe-list cut chosen by the RULE (nobody asked for it) and its exchange failed -- on all ranks together, the library agrees on failures before its
                    # collectives: the job goes on with the plain pass and says so
                    if not (rule["on"] and cut_on[0] and auto_pass):
                        raise
                    rule["on"], rule["fallback"] = False, str(e)[:300]
                    info = sharded_prefilter(c, sub or comm, q_from, q_to, probes=False)
                xch["probe_s"] += info["probe_exchange_s"]
                xch["s"] += info["exchange_s"]
                xch["bytes"] += info["bytes"]
                xch["n"] += 1
            else:
                c.prefilter()

        # sizing pass (also the first warm-up): how many hits does a batch deliver?
        caps = []
        for c in ctxs:
            search(c)
            caps.append(int(c.num_hits() * 1.05) + 1024)
        share = [steps // n_pipe + (1 if i < steps % n_pipe else 0) for i in range(n_pipe)]
        slots = [[Slot(L, nq, caps[i], full) for _ in range(share[i])] for i in range(n_pipe)]
        ref_slot = [Slot(L, nq, caps[i], full) for i in range(n_pipe)]

        device_driven = full and not shard_targets and not host_driven      # the whole step queued without a host round trip (sp_ctx_search_begin / _end)

        def exchange_hits(slot):
            """query partition: the step's hit list of every rank, all-gathered (counts, then the padded 8-byte records): what the host-side steps
            behind the search (besthitperset ... FDR) need in one place; RCCL HBM -> xGMI -> HBM, gloo through host memory in the tests"""
            if not (partition and world > 1):
                return
            t0 = time.perf_counter()
            before = comm.bytes_moved()
            spd.allgather_hits(ctxs[0], comm, part_from, part_total, my_hits=slot.hits, my_offsets=slot.off, my_n=slot.n)      # (any context of this rank: the list comes from the slot)
            xch["s"] += time.perf_counter() - t0
            xch["bytes"] += comm.bytes_moved() - before
            xch["n"] += 1

        def step(c, slot):
            if device_driven:      # one step at a time (reference digests, warm-up, breakdown): queue it, collect it
                c.search_begin(slot.hits, slot.off, slot.cap, slot.rec, slot.cig, slot.cig_cap)
                n, used = c.search_end()
                slot.n, slot.used.value, slot.hits_ptr_ok = n, used, True
                return n
            c.set_hit_sink(slot.hits, slot.off, slot.cap)
            search(c)
            n = slot.n = c.num_hits()
            if full and n:
                c.align_hits_begin(into=slot.into)         # alignment result DB content (result_t + backtraces per hit): returns when the
                                                           # kernels are done, the records travel while the next pass's prefilter runs
            hp, _, cnt = c.hits_view()                     # prefilter result DB content (hit_t triples per query) in host memory: its copy
            slot.hits_ptr_ok = (C.cast(hp, C.c_void_p).value == slot.hits) and cnt == n      # started underneath the align kernels
            return n

        def barrier():
            for c in ctxs:
                if not device_driven:
                    c.align_hits_end()                     # the last pass's alignment records are in host memory
            for c in ctxs:
                c.sync()
            if comm is not None and world > 1:
                comm.barrier()

        # reference digests: every batch searched by its context ALONE, default tuning, nothing else on the GPU
        ref = []
        for i, c in enumerate(ctxs):
            step(c, ref_slot[i])
            barrier()
            assert ref_slot[i].hits_ptr_ok
            ref.append(ref_slot[i].digest(full))
        cut_on[0] = True
        if n_pipe > 1:
            for c in ctxs:
                c.set_tuning("match_blocks_per_cu", args.match_blocks)     # leaves LDS / registers for the other context's aligner on every CU
                for kv in args.tune:
                    c.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
        for i, c in enumerate(ctxs):
            for _ in range(warmup):
                step(c, slots[i][0])
                exchange_hits(slots[i][0])      # (query partition at N > 1: the collective is part of the step -- its first calls set up connections / channels)
        barrier()
        import threading
        for c in ctxs:
            c.set_timing(1)
        acc = [[0.0, 0, 0.0] for _ in range(n_pipe)]      # k_match ms by HIP events, pairs of the last step, k_match ms by the kernel's own wall clock

        def run(i):
            c = ctxs[i]
            if device_driven:
                # two steps in flight per context: step j + 1 is queued before step j is collected, so the context's stream never waits for the
                # host; every step is complete (hit list, records, backtraces in ITS page-locked buffers) when its search_end() returns
                def collect(slot):
                    n, used = c.search_end()
                    slot.n, slot.used.value, slot.hits_ptr_ok = n, used, True
                    done[id(slot)].set()      # (the exchange thread takes the steps in a fixed order, the same on every rank)
                    acc[i][1] = n
                    v = c.stat("ms_match")
                    acc[i][0] += 0.0 if v != v else v
                    v = c.stat("ms_match_active")
                    acc[i][2] += 0.0 if v != v else v
                prev = None
                for j in range(share[i]):
                    s = slots[i][j]
                    c.search_begin(s.hits, s.off, s.cap, s.rec, s.cig, s.cig_cap)
                    if args.in_flight < 2:
                        collect(s)
                        continue
                    if prev is not None:
                        collect(prev)
                    prev = s
                if prev is not None:
                    collect(prev)
                return
            if args.stagger_us > 0 and i > 0:      # the contexts enter a region one after the other instead of all at once (their matching kernels would meet in phase)
                t_go = time.perf_counter() + i * args.stagger_us * 1e-6
                while time.perf_counter() < t_go:
                    pass
            for j in range(share[i]):
                acc[i][1] = step(c, slots[i][j])
                done[id(slots[i][j])].set()           # (host-driven: the hit list is in host memory when step() returns)
                v = c.stat("ms_match")
                acc[i][0] += 0.0 if v != v else v
                v = c.stat("ms_match_active")
                acc[i][2] += 0.0 if v != v else v

        region_dt, region_match, region_active, checked = [], [], [], 0
        done = {}

        def exchanger():      # collectives from ONE host thread per rank, steps in the order (round, context)
            for j in range(max(share)):
                for i2 in range(n_pipe):
                    if j < share[i2]:
                        done[id(slots[i2][j])].wait()
                        exchange_hits(slots[i2][j])
        # the contexts' host threads exist BEFORE the clock starts and wait at a gate (round 6: starting four Python threads inside the timed region cost 0.2-0.4 ms of a 20-step
        # region of 9 ms); a region = gate opens -> every thread runs its share of the K steps -> all are back
        jobs = [lambda i=i: run(i) for i in range(n_pipe)] + ([exchanger] if partition and world > 1 else [])
        gate_in, gate_out, stop, failed = threading.Barrier(len(jobs) + 1), threading.Barrier(len(jobs) + 1), [False], []

        def worker(job):
            while True:
                gate_in.wait()
                if stop[0]:
                    return
                try:
                    job()
                except BaseException as e:      # noqa: BLE001  (reported by the main thread after the region)
                    failed.append(e)
                gate_out.wait()
        pool = [threading.Thread(target=worker, args=(job,), daemon=True) for job in jobs] if len(jobs) > 1 else []
        for th in pool:
            th.start()
        for _ in range(regions):
            for a in acc:
                a[0] = 0.0
                a[2] = 0.0
            xch.update(s=0.0, bytes=0, n=0)
            done.clear()
            done.update({id(sl): threading.Event() for grp in slots for sl in grp})
            barrier()
            t0 = time.perf_counter()
            if not pool:
                run(0)
            else:
                gate_in.wait()
                gate_out.wait()
                if failed:
                    raise failed[0]
            barrier()
            dt = time.perf_counter() - t0
            region_dt.append(all_max(dt))
            region_match.append(sum(a[0] for a in acc) / steps)
            region_active.append(sum(a[2] for a in acc) / steps)
            # the clock has stopped: every result this region delivered must be the bytes the context delivered alone
            for i in range(n_pipe):
                for j in range(share[i]):
                    s = slots[i][j]
                    if not s.hits_ptr_ok or s.digest(full) != ref[i]:
                        raise RuntimeError("rank %d: result of step %d of context %d differs from the single-context result of the same batch" % (rank, j, i))
                    checked += 1
        if pool:
            stop[0] = True
            gate_in.wait()
            for th in pool:
                th.join()
        xch_region = dict(xch)
        pairs = [acc[i][1] for i in range(n_pipe)]
        stats = None
        if want_breakdown:      # per-stage breakdown: one context alone, HIP events around every stage, outside the timed regions
            stats = {"ms_match": 0.0, "ms_prep": 0.0, "ms_score": 0.0, "ms_select": 0.0, "ms_sw_forward": 0.0, "ms_sw_reverse": 0.0, "ms_traceback": 0.0}
            ctx.set_timing(2)
            ctx.set_tuning("match_blocks_per_cu", -1)      # alone on the GPU the matcher wants all the LDS (the other contexts of the region are idle now)
            for _ in range(3):
                step(ctx, ref_slot[0])
                for k in stats:
                    v = ctx.stat(k)
                    stats[k] += 0.0 if v != v else v
            barrier()
            stats = {k: v / 3 for k, v in stats.items()}
            ctx.set_timing(1)
        counters = {k: ctx.stat(k) for k in ("kmers", "db_matches", "diagonals", "diag_residues", "hits")}
        out = dict(region_dt=region_dt, region_match=region_match, region_active=region_active, checked=checked, pairs=pairs, stats=stats, counters=counters, n_pipe=n_pipe, nq=nq, device_driven=device_driven,
                   steps_repeated=sum(v for v in (c.stat("steps_repeated") for c in ctxs) if v == v),
                   t_index=t_index, batches=batches, xch=xch_region, q_slice=(q_from, q_to))
        for grp in slots + [ref_slot]:
            for s in grp:
                s.free()
        for c in ctxs:
            c.close()
        return out

    dual = world == 1 and not sharded and full and args.pipeline is None and not args.host_driven
    alt = None
    qpart_h = None
    if world > 1 and sharded and full and not args.no_replica and args.pipeline is None and not args.host_driven:
        # N > 1: the query partition with host-driven steps in four contexts per rank, measured before anything else for the same reason as at N = 1
        qpart_h = measure(False, n_spacers, max(4, args.steps // 2), max(1, min(3, args.regions)), args.warmup, False, partition=True, pipeline=4, host_driven=True)
    if dual:
        # N = 1, no mode asked for: the step is measured both ways -- device-driven (four contexts, two steps in flight each) and host-driven with four
        # contexts, whose idle moments between prefilter and align the other contexts fill -- `value` is the host-driven one (fixed); both stay in the line.
        # (The eight HIP streams of four contexts share the runtime's four hardware queues pairwise -- search stream with search stream, copy stream with copy
        # stream -- when they are created in order; the library keeps its stream pairs in a per-device pool for that, so the order of the two measurements
        # does not matter any more: SP_BENCH_DEVICE_DRIVEN_FIRST=1 swaps them, 19.1 M either way; before the pool the second place cost 10 %.)
        if not os.environ.get("SP_BENCH_DEVICE_DRIVEN_FIRST"):      # (experiment: the other order, to check that the library's stream pool makes it irrelevant)
            alt = measure(False, n_spacers, args.steps, args.regions, args.warmup, False, pipeline=4, host_driven=True)
    m = measure(sharded, n_spacers, args.steps, args.regions, args.warmup, True, shards=args.shards, pipeline=4 if dual else None)      # (dual: device-driven steps in FOUR contexts, two in flight each -- level with the host-driven four since round 6's late copies)
    if dual and alt is None:
        alt = measure(False, n_spacers, args.steps, args.regions, args.warmup, False, pipeline=4, host_driven=True)
    rep = hyb = qpart = None
    step_modes = None
    if dual:

        def med(mm):
            d = sorted(mm["region_dt"])
            return d[len(d) // 2] if len(d) % 2 else 0.5 * (d[len(d) // 2 - 1] + d[len(d) // 2])
        def launch_ms(mm):      # k_match per launch inside the median region (HIP events on the contexts' streams)
            o = sorted(range(len(mm["region_dt"])), key=lambda i: mm["region_dt"][i])
            return mm["region_match"][o[len(o) // 2]]

        def active_ms(mm):
            o = sorted(range(len(mm["region_dt"])), key=lambda i: mm["region_dt"][i])
            return mm["region_active"][o[len(o) // 2]]
        step_modes = {"device-driven x4": {"spacer_queries_per_s": n_spacers * args.steps / med(m), "ms_per_step": med(m) / args.steps * 1e3, "k_match_ms_per_launch": round(launch_ms(m), 4), "k_match_active_ms_per_launch": round(active_ms(m), 4)},
                      "host-driven x4": {"spacer_queries_per_s": n_spacers * args.steps / med(alt), "ms_per_step": med(alt) / args.steps * 1e3, "k_match_ms_per_launch": round(launch_ms(alt), 4), "k_match_active_ms_per_launch": round(active_ms(alt), 4)}}
        # ONE fixed configuration carries `value`: host-driven steps in four contexts (the faster one in every run of rounds 3-5: 21-24 M against 15-17 M; level since round 6); the
        # device-driven step -- the interface that needs no host round trip inside a step -- is measured beside it and stays in the line (step_modes)
        alt["stats"] = m["stats"]
        m = alt
        step_modes["value_from"] = "host-driven x4 (fixed configuration; the device-driven step is reported beside it)"
    if world > 1 and sharded and not args.no_replica:
        rep = measure(False, 10000, max(4, args.steps // 2), max(1, min(3, args.regions)), args.warmup, False)
        # the same 100 000-spacer job cut by QUERY (index replicated, hit lists all-gathered after every step): strong scaling like the sharded line
        qpart = measure(False, n_spacers, max(4, args.steps // 2), max(1, min(3, args.regions)), args.warmup, False, partition=True)
    if world >= 4 and world % 2 == 0 and sharded and not args.shards and not args.no_hybrid:
        # the same job with TWO target shards x world / 2 query groups: the enumeration of the similar k-mers is replicated twice, not world times
        hyb = measure(True, n_spacers, max(4, args.steps // 2), max(1, min(3, args.regions)), args.warmup, False, shards=2)
    dense = None
    if dual and not args.no_dense and (n_spacers, args.targets, args.target_len) == (10000, 50000, 200):
        # the density north_star's target DB has: the same 10 000 spacers against 10^6 x 200 residues (95 % of the k-mer slots occupied, 2 200 index hits per query):
        # host-driven steps in two contexts, fewer steps -- a step is ten times longer.  (The dense matching kernel fills every CU's LDS: more contexts LOSE there
        # -- three 1.4 M, four 1.7 M spacer-queries/s -- and the device-driven step in two contexts 2.1 M against 2.5 M: profiles/r04_dense_step_modes.txt)
        dense = measure(False, 10000, max(4, args.steps // 10), max(1, min(3, args.regions)), 2, True, pipeline=2, host_driven=True, targets=1000000)
    probe_proxy = None
    if dual and not args.no_probe_proxy and (n_spacers, args.targets, args.target_len) == (10000, 50000, 200):
        # the probe-list cut of the target-sharded search on ONE GPU playing shard 0 of 8 at configs[3]'s size: what the matching work of a step costs
        # when every shard enumerates every query (the N > 1 `value`'s decomposition) against enumerate-an-eighth + place + fetch; no N > 1 run has measured its exchange
        try:
            probe_proxy = probe_cut_proxy(sa, args)
        except Exception as e:      # (a side figure must not cost the line)
            probe_proxy = {"error": repr(e)[:300]}
    c3_one = c3_four = dense_shard = None
    if dual and not args.no_dense and (n_spacers, args.targets, args.target_len) == (10000, 50000, 200):
        # BASELINE configs[3]'s job -- 100 000 spacers of a step against the whole target set -- UNSPLIT on this one GPU: the N = 1 anchor of the N > 1 line (the same job, strong
        # scaling).  One context = what a rank of the N > 1 run has; four host-driven contexts = this GPU's best.
        k3 = max(3, args.steps // 20)
        c3_one = measure(False, 100000, k3, max(1, min(3, args.regions)), 2, False, pipeline=1, host_driven=True)
        c3_four = measure(False, 100000, 4 * k3, max(1, min(3, args.regions)), 2, False, pipeline=4, host_driven=True)
    if dual and not args.no_probe_proxy and (n_spacers, args.targets, args.target_len) == (10000, 50000, 200):
        # ... and the target-sharded passes at the density north_star's DB has: this GPU as shard 0 of 8 of 2e8 residues (2.5e7 per shard, 31 % of the slots occupied), 20 000
        # spacers: the plain pass against the probe-list cut -- the measurement behind spacepharer_amd.dist.choose_sharded_pass
        try:
            import argparse as _ap
            dense_shard = probe_cut_proxy(sa, _ap.Namespace(targets=1000000, target_len=200), shards=8, spacers=20000, probes_per_query=640)
        except Exception as e:
            dense_shard = {"error": repr(e)[:300]}
    pairs_total = all_sum(float(m["pairs"][0]))      # batch 0 of every rank (sharded: the ranks' query slices add up to the batch)
    if comm is not None:
        comm.barrier()
        comm.close()
    if rank != 0:
        return

    def region_summary(mm, spacers_per_region):
        dts = sorted(mm["region_dt"])
        med = dts[len(dts) // 2] if len(dts) % 2 else 0.5 * (dts[len(dts) // 2 - 1] + dts[len(dts) // 2])
        return med, {"n": len(dts), "spacer_queries_per_s": [round(spacers_per_region / d, 1) for d in mm["region_dt"]],
                     "min": spacers_per_region / dts[-1], "max": spacers_per_region / dts[0], "median": spacers_per_region / med}

    spacers_per_region = n_spacers * (1 if sharded else world) * args.steps
    dt, regions = region_summary(m, spacers_per_region)
    K, M, D, R, H = (m["counters"][k] for k in ("kmers", "db_matches", "diagonals", "diag_residues", "hits"))
    order = sorted(range(len(m["region_dt"])), key=lambda i: m["region_dt"][i])
    ms_match = m["region_match"][order[len(order) // 2]]      # HIP events on the context's stream, inside the (median) timed region
    ms_active = m["region_active"][order[len(order) // 2]]    # the same launches from their first wave's start to their last wave's end (the kernel's own wall clock)
    # algorithmic bytes of the dominant kernel (k_match) per launch: 6 B similar-k-mer table entry + 16 B offsets pair
    # per similar k-mer, 6 B per index entry fetched, 7 B per double-hit diagonal written
    algo_bytes = 22.0 * K + 6.0 * M + 7.0 * D
    achieved = algo_bytes / (ms_match * 1e-3) / 1e9 if ms_match > 0 else 0.0
    default_wl = (n_spacers, args.targets, args.target_len, world) == (10000, 50000, 200, 1)
    traffic, traffic_src = pmc_traffic("k_match") if default_wl else (None, None)
    stats, n_pipe, nq = m["stats"], m["n_pipe"], m["nq"]
    out = {
        "metric": "spacer-queries/sec through prefilter+align; achieved HBM GB/s vs roofline",
        "value": spacers_per_region / dt,
        "unit": "spacer-queries/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
        "dtype": "u8/i16", "data": "synthetic",
        "config": {"workload": "%d synthetic 30-nt spacers (= %d peptides of 10 aa) vs %d synthetic phage ORFs x %d aa "
                               "(%.1e residues), %s, %s" % (n_spacers, nq, args.targets, args.target_len, args.targets * args.target_len,
                                                            ("targets sharded over %d x MI355X" % world) if sharded else "1xMI355X per rank",
                                                            "prefilter + ungapped + banded-SW align" if full else "prefilter only"),
                   "stage": args.stage, "pairs_aligned_per_step": int(pairs_total), "pairs_aligned_per_step_by_batch_rank0": m["pairs"], "sensitivity": 5.7, "max_seqs": 300,
                   "parallelism": ("target shards x%d%s, %s all-gather of candidates, selection + align by query slice" % (args.shards or world, (" x %d query groups" % (world // args.shards)) if args.shards and args.shards != world else "", "RCCL" if args.exchange == "rccl" else "host-memory")) if sharded
                                  else ("query partition x%d, index replicated" % world),
                   "passes_in_flight_per_gpu": n_pipe, "distinct_query_batches": n_pipe,
                   "step": ("device-driven: the whole step is queued without a host round trip (sp_ctx_search_begin / _end), two steps in flight per context; %d steps met a capacity and were repeated the synchronous way" % int(m["steps_repeated"]))
                           if m["device_driven"] else "host-driven: the host learns the hit count between prefilter and align (sp_ctx_prefilter + sp_ctx_align_hits_begin / _end)",
                   "index_build_s": round(m["t_index"], 3)},
        "regions": regions,
        "results_checked": {"steps": m["checked"], "equal_to_single_context_result": True,
                            "what": "crc32 over offsets + hit records%s of EVERY timed step, compared after the clock stopped with the same batch searched by one context alone"
                                    % (" + alignment records + backtrace pool" if full else "")},
        "kernels_ms_per_step": {k[3:]: round(v, 4) for k, v in stats.items()},
        "kernels_ms_source": "HIP events around every stage in 3 extra steps of ONE context AFTER the timed regions (the timed steps bracket k_match only: "
                             "%.4f ms per launch with %d passes in flight)" % (ms_match, n_pipe),
        "counters_per_step": {"similar_kmers": K, "index_entries_fetched": M, "diagonals_scored": D, "diag_residues": R, "hits": H},
        "roofline": {"bound": "hbm", "kernel": "k_match_grp (the matching kernel for short queries on a sparse index; k_match otherwise)" if (args.target_len * args.targets <= 50000000 and not sharded) else "k_match", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                     "frac": achieved / 8000.0, "traffic": traffic,
                     "traffic_source": traffic_src, "traffic_stale": pmc_traffic.stale, "traffic_device_sources_sha256": pmc_traffic.collected_on, "algorithmic_bytes_per_launch": algo_bytes,
                     "measured_streaming_read_GBps": pmc_traffic.measured_stream,
                     # the same kernel with the GPU to itself (one context, the breakdown pass after the timed regions): its launches in the
                     # timed regions share the machine with the other context's aligner and take longer each, while more work gets done
                     "exclusive": {"ms_per_launch": round(stats["ms_match"], 4),
                                   "achieved": algo_bytes / (stats["ms_match"] * 1e-3) / 1e9 if stats["ms_match"] > 0 else 0.0,
                                   "frac": algo_bytes / (stats["ms_match"] * 1e-3) / 1e9 / 8000.0 if stats["ms_match"] > 0 else 0.0},
                     "passes_in_flight": n_pipe,
                     # the launch from its first wave's start to its last wave's end (device wall clock read by the kernel: what rocprofv3 reports as the kernel's
                     # duration); the event time above also holds the wait for the first free wave slots while other contexts' kernels fill the CUs
                     "kernel_active": {"ms_per_launch": round(ms_active, 4), "achieved": algo_bytes / (ms_active * 1e-3) / 1e9 if ms_active > 0 else 0.0,
                                       "frac": algo_bytes / (ms_active * 1e-3) / 1e9 / 8000.0 if ms_active > 0 else 0.0},
                     # launches of several contexts overlap: each takes longer than alone and more of them finish per second.  The same algorithmic bytes over
                     # the WALL time of the region (all contexts' launches, everything else of the step included) -- the kernel's share of the machine as the job sees it
                     "over_wall_time": {"achieved": algo_bytes * args.steps / dt / 1e9, "frac": algo_bytes * args.steps / dt / 1e9 / 8000.0}},
        "host": host_info(),
    }
    if dense is not None:
        dsteps = max(4, args.steps // 10)
        ddt, dregions = region_summary(dense, 10000 * dsteps)
        dK, dM, dD = (dense["counters"][k] for k in ("kmers", "db_matches", "diagonals"))
        dalgo = 22.0 * dK + 6.0 * dM + 7.0 * dD
        dexcl = dense["stats"]["ms_match"]
        out["dense"] = {"workload": "10000 synthetic 30-nt spacers vs 1000000 synthetic phage ORFs x 200 aa (2.0e+08 residues: the survey's estimate of GenBank_phage_2018_09), 1xMI355X, prefilter + align",
                        "value": 10000 * dsteps / ddt, "unit": "spacer-queries/s", "ms_per_step": ddt / dsteps * 1e3, "steps": dsteps, "regions": dregions, "results_checked": dense["checked"],
                        "pairs_aligned_per_step": dense["pairs"][0], "index_build_s": round(dense["t_index"], 3), "step": "host-driven, two contexts",
                        "kernels_ms_per_step": {k[3:]: round(v, 4) for k, v in dense["stats"].items()},
                        "counters_per_step": {"similar_kmers": dK, "index_entries_fetched": dM, "diagonals_scored": dD},
                        "roofline": {"kernel": "k_match<512,false,false> (large footprint, direct bucket table)", "algorithmic_bytes_per_launch": dalgo,
                                     "exclusive_ms_per_launch": round(dexcl, 4), "exclusive_frac": dalgo / (dexcl * 1e-3) / 1e9 / 8000.0 if dexcl > 0 else None}}
    if step_modes is not None:
        out["config"]["step"] += " -- four search contexts (host threads) on one shared target set; fixed configuration, the device-driven step is measured beside it (step_modes)"
        for k in ("device-driven x4", "host-driven x4"):      # the roofline fraction of a k_match launch in either mode (more passes in flight: longer launches, more work done)
            step_modes[k]["roofline_frac_per_launch"] = round(algo_bytes / (step_modes[k]["k_match_ms_per_launch"] * 1e-3) / 1e9 / 8000.0, 4) if step_modes[k]["k_match_ms_per_launch"] > 0 else None
        out["step_modes"] = step_modes
    if sharded:
        x = m["xch"]
        out["exchange"] = {"transport": "RCCL all-gather (HBM -> xGMI -> HBM), sp_ctx_allgather_candidates" if args.exchange == "rccl" else "all-gather through host memory and sockets (sp_dist host transport)",
                           "rccl_ranks": world if args.exchange == "rccl" else 0, "ranks": world, "requested": getattr(args, "exchange_requested", args.exchange),
                           "as_requested": getattr(args, "exchange_requested", args.exchange) == args.exchange,      # false: RCCL was asked for, the ranks fell back to the host transport together (stderr says why)
                           "allgather_bytes_per_step": x["bytes"] / max(1, x["n"]),
                           "exchange_ms_per_step": 1e3 * x["s"] / max(1, x["n"]),
                           "query_slice_of_rank0": list(m["q_slice"])}
        if rule["on"]:
            out["exchange"]["probe_cut"] = {"what": "every rank enumerates its query slice once (k_match_grp<true, 1>) and sends the occupied probes (12 bytes each) to the shards that hold the slot "
                                                    "(sp_ctx_probe_search: all-to-all); the shards fetch (k_match_grp<true, 2>); every timed step compared with the pass in which every shard enumerates every query",
                                            "probe_exchange_ms_per_step": 1e3 * x["probe_s"] / max(1, x["n"])}
    if probe_proxy is not None:
        out["probe_cut_one_gpu"] = probe_proxy
    if c3_one is not None:
        blk = {"what": "BASELINE configs[3]'s job unsplit on ONE GPU: 100 000 synthetic spacers (600 000 peptides) per step against the whole target set (1e7 residues), prefilter + align, "
                       "host-driven steps; the N = 1 anchor of the N > 1 line (same job, strong scaling)"}
        for tag, mm, k in (("one_context", c3_one, max(3, args.steps // 20)), ("four_contexts", c3_four, 4 * max(3, args.steps // 20))):
            cdt, cregions = region_summary(mm, 100000 * k)
            blk[tag] = {"value": 100000 * k / cdt, "unit": "spacer-queries/s", "ms_per_step": cdt / k * 1e3, "steps": k, "regions": cregions, "results_checked": mm["checked"]}
        out["configs3_one_gpu"] = blk
    if dense_shard is not None:
        out["dense_shard_proxy"] = dense_shard
    if hyb is not None:
        hdt, hregions = region_summary(hyb, n_spacers * max(4, args.steps // 2))
        out["hybrid"] = {"value": n_spacers * max(4, args.steps // 2) / hdt, "unit": "spacer-queries/s", "scaling": "strong", "ms_per_step": hdt / max(4, args.steps // 2) * 1e3,
                         "regions": hregions, "results_checked": hyb["checked"],
                         "parallelism": "2 target shards x %d query groups: RCCL all-gather of the candidates within each pair of ranks" % (world // 2)}
    if qpart is not None:
        k2 = max(4, args.steps // 2)
        qdt, qregions = region_summary(qpart, n_spacers * k2)
        x = qpart["xch"]
        modes = {"device-driven x2": n_spacers * k2 / qdt}
        step_kind = "device-driven steps, two contexts per rank"
        if qpart_h is not None:
            hdt2, hregions2 = region_summary(qpart_h, n_spacers * k2)
            modes["host-driven x4"] = n_spacers * k2 / hdt2
            if hdt2 < qdt:
                qdt, qregions, qpart, x = hdt2, hregions2, qpart_h, qpart_h["xch"]
                step_kind = "host-driven steps, four contexts per rank"
        out["query_partition"] = {"value": n_spacers * k2 / qdt, "unit": "spacer-queries/s", "scaling": "strong", "ms_per_step": qdt / k2 * 1e3, "regions": qregions, "step_modes": modes,
                                  "results_checked": qpart["checked"], "allgather_bytes_per_step": x["bytes"] / max(1, x["n"]), "exchange_ms_per_step": 1e3 * x["s"] / max(1, x["n"]),
                                  "parallelism": "the same job cut by query: every rank holds the full index (it fits HBM a hundred times over) and searches 1/%d of the spacers of a step, "
                                                 "%s, %s all-gather of the hit lists after every step" % (world, step_kind, "RCCL" if args.exchange == "rccl" else "host-memory")}
    if world > 1 and sharded:
        # the headline at N > 1 is BASELINE configs[3]'s decomposition, fixed: target shards + all-gather of the candidates (RCCL over xGMI).  The other cuts of the
        # SAME job (100 000 spacers against the whole target set, strong scaling) stay in the line as side figures; `weak_replica` is the N = 1 job on every rank.
        probe_cut = rule["on"]
        cands = {"target_shards_probe_cut" if probe_cut else "target_shards": out["value"]}
        if hyb is not None:
            cands["hybrid"] = out["hybrid"]["value"]
        if qpart is not None:
            cands["query_partition"] = out["query_partition"]["value"]
        out["decompositions"] = {"spacer_queries_per_s": cands, "value_from": (("target_shards_probe_cut" if probe_cut else "target_shards") + (" (BASELINE configs[3]; the pass on the shards chosen by spacepharer_amd.dist.choose_sharded_pass from the fullest shard's occupancy %.3f and %d ranks: %s)" % (rule["occupancy"], world, rule["verdict"]) if rule["verdict"] else " (asked for: --shard / SP_SHARD)")),
                                 "rule": {"occupancy_of_the_fullest_shard": rule["occupancy"], "ranks": world, "verdict": rule["verdict"], "probe_cut_max_occupancy": spd.PROBE_CUT_MAX_OCCUPANCY, "probe_cut_min_shards": spd.PROBE_CUT_MIN_SHARDS,
                                          "measured_on": "profiles/r05_probe_cut.txt, profiles/r06_probe_cut_dense.txt", "fell_back_to_plain": rule.get("fallback")}, "best_measured": max(cands, key=cands.get),
                                 "target_shards_ms_per_step": out["ms_per_step"]}
        out["scaling_note"] = ("N = 1 measures configs[2] (10 000 spacers per step on one GPU, `weak_replica` continues that job to N ranks); N > 1 measures configs[3] (100 000 spacers, "
                               "target shards, strong scaling) -- two different jobs, do not join them into one efficiency curve; use `weak_replica` for the weak series.  No 1 -> 8 curve "
                               "has been measured on hardware by the builder: every RCCL run so far had world = 1.")
    if rep is not None:
        rdt, rregions = region_summary(rep, 10000 * world * max(4, args.steps // 2))
        out["weak_replica"] = out["replica"] = {"value": 10000 * world * max(4, args.steps // 2) / rdt, "unit": "spacer-queries/s", "scaling": "weak",
                          "ms_per_step": rdt / max(4, args.steps // 2) * 1e3, "regions": rregions, "results_checked": rep["checked"],
                          "parallelism": "query partition x%d: every rank holds the full index and searches its own 10 000 spacers, no data-path collective" % world}
    if args.cpu_sample > 0 and world == 1:      # rank 0 at N = 1 only
        # the reference itself when its harness travelled with the repo (oracle/_ref, built by __graft_entry__.build()), else
        # the scalar port; either way a bounded sample
        hi = out["host"]
        # all the cores this container may use: the affinity mask, capped by its cgroup CPU quota (256 hardware threads are visible on
        # the pool's boxes but cpu.max grants 16 CPUs: 256 threads just get throttled -- 27 k spacer-queries/s, 8 of them do 21 k)
        cores = args.cpu_cores if args.cpu_cores > 0 else hi["cores_for_cpu_baseline"]
        cores = max(1, min(cores, hi["usable_cores"], 512))
        tres, toff, qres, qoff = m["batches"][0]
        if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_harness")):
            try:
                n_s = nq if args.cpu_sample >= 3000 else min(args.cpu_sample * cores, nq)
                out["cpu_baseline"] = cpu_baseline_reference(tres, toff, qres, qoff, n_s, cores, want_results=not sharded)
                res_ref = out["cpu_baseline"].pop("results", None)
                if n_s == nq and not sharded:     # whole workload: the reference must align exactly the pairs the GPU aligned
                    out["cpu_baseline"]["pairs_equal_gpu"] = bool(out["cpu_baseline"].pop("pairs") == m["pairs"][0])
                if res_ref is not None:           # ... and a sample of its hit lists and alignments, field by field, must be the GPU's (after the clock stopped)
                    n_cmp = min(n_s, 3000)
                    out["cpu_baseline"]["results_equal_gpu"] = reference_results_equal_gpu(tres, toff, qres, qoff, n_cmp, res_ref)
            except Exception as e:      # noqa: BLE001 -- fall back to the port, say why
                out["cpu_baseline"] = cpu_baseline(tres, toff, qres, qoff, min(args.cpu_sample, nq))
                out["cpu_baseline"]["sample"] += " (reference harness unavailable: %s)" % e
        else:
            out["cpu_baseline"] = cpu_baseline(tres, toff, qres, qoff, min(args.cpu_sample, nq))
        out["cpu_baseline"]["nproc"] = hi["nproc"]
        out["cpu_baseline"]["cpu_quota_cores"] = hi["cpu_quota_cores"]
        out["cpu_baseline"]["cpu_model"] = hi["cpu_model"]
    if dense is not None and args.cpu_sample > 0 and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_harness")):
        try:      # the reference on the same dense workload: a bounded sample of the peptides (its index build alone takes seconds there)
            tres_d, toff_d, qres_d, qoff_d = dense["batches"][0]
            n_d = min(len(qoff_d) - 1, 200 * cores)
            cb = cpu_baseline_reference(tres_d, toff_d, qres_d, qoff_d, n_d, cores, target_s=3.0, want_results=True)
            cb.pop("pairs", None)
            # the REAL reference classes at the density of the real DB: their hit lists and alignments for the sample, compared with the GPU's field by field
            cb["results_equal_gpu"] = reference_results_equal_gpu(tres_d, toff_d, qres_d, qoff_d, n_d, cb.pop("results"))
            out["dense"]["cpu_baseline"] = cb
            out["dense"]["x_reference_search_loops"] = out["dense"]["value"] / cb["value"]
        except Exception as e:      # noqa: BLE001
            out["dense"]["cpu_baseline"] = {"error": repr(e)[:300]}
    if world == 1 and not args.no_module_e2e and not sharded and (n_spacers, args.targets, args.target_len) == (10000, 50000, 200):
        # the boundary the workflow scripts use: module calls on DB triples, "DBs on disk" -> "result DB written", as fresh processes
        # of the front-end binary and warm inside one process; two modules (prefilter, align) and the fused search
        import subprocess
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_extras
        bench_extras.files_to_files(out, args, dense, cores if args.cpu_sample > 0 else 0, nq)
    # the figures a reader of the LAST 2 000 characters of this line needs (the side blocks above are long): one flat object, last key of the line; the dense kernel's
    # fraction also rides in `roofline` itself
    if "dense" in out and isinstance(out["dense"].get("roofline"), dict):
        out["roofline"]["dense_index"] = {"kernel": out["dense"]["roofline"]["kernel"], "exclusive_ms_per_launch": out["dense"]["roofline"]["exclusive_ms_per_launch"],
                                          "exclusive_frac": out["dense"]["roofline"]["exclusive_frac"], "ms_per_step": out["dense"]["ms_per_step"], "value": out["dense"]["value"]}
    summ = {"value": out["value"], "ms_per_step": out["ms_per_step"], "roofline_frac": out["roofline"]["frac"], "roofline_exclusive_frac": out["roofline"]["exclusive"]["frac"],
            "traffic_stale": out["roofline"].get("traffic_stale")}
    if "dense" in out:
        summ["dense"] = {"value": out["dense"]["value"], "ms_per_step": out["dense"]["ms_per_step"], "k_match_exclusive_ms": out["dense"]["roofline"]["exclusive_ms_per_launch"],
                         "exclusive_frac": out["dense"]["roofline"]["exclusive_frac"], "x_cpu_reference": (round(out["dense"]["value"] / out["dense"]["cpu_baseline"]["value"], 1) if (out["dense"].get("cpu_baseline") or {}).get("value") else None)}
    if "configs3_one_gpu" in out:
        summ["configs3_one_gpu"] = {k: {"value": v["value"], "ms_per_step": v["ms_per_step"]} for k, v in out["configs3_one_gpu"].items() if isinstance(v, dict)}
    if isinstance(out.get("dense_shard_proxy"), dict) and "x_plain" in out["dense_shard_proxy"]:
        summ["dense_shard_proxy"] = {k: out["dense_shard_proxy"][k] for k in ("plain_k_match_grp_ms", "cut_matching_ms", "x_plain", "probe_bytes_received")}
    if isinstance(out.get("probe_cut_one_gpu"), dict) and "x_plain" in out["probe_cut_one_gpu"]:
        summ["sparse_shard_proxy"] = {k: out["probe_cut_one_gpu"][k] for k in ("plain_k_match_grp_ms", "cut_matching_ms", "x_plain", "probe_bytes_received")}
    f2f = out.get("files_to_files") or {}
    for tag, blk in (("files_to_files_1e7", f2f), ("files_to_files_2e8", f2f.get("dense") or {})):
        xs = {m: (blk.get(m) or {}).get("x_reference") for m in ("fused_search", "prefilter_then_align") if isinstance(blk.get(m), dict)}
        if any(xs.values()):
            summ[tag] = {m: {k: round(v, 1) for k, v in x.items()} for m, x in xs.items() if x}
    lp = ((f2f.get("dense") or {}).get("fused_search") or {}).get("fresh_process_laps")
    if lp:
        summ["fresh_fused_process_2e8_laps_ms"] = {k: v for k, v in lp.items() if k != "note"}
    if "cpu_baseline" in out:
        summ["x_cpu_reference"] = round(out["value"] / out["cpu_baseline"]["value"], 1) if out["cpu_baseline"].get("value") else None
    out["summary"] = summ
    C.CDLL(None).fflush(None)      # library chatter buffered in C stdio (RCCL prints its path) goes out BEFORE the result line
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
