import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = os.environ.get("SP_LIB_NAME", "libspacepharer_amd.so")    # override: instrumented builds (tools/)


class SpError(RuntimeError):
    pass


def library_path():
    # the product library lives in the package directory; instrumented / sanitized builds (make variant | phase | ablate | asan) in csrc/build/
    here = os.path.join(HERE, LIB_NAME)
    alt = os.path.join(HERE, "csrc", "build", LIB_NAME)
    return alt if (LIB_NAME != "libspacepharer_amd.so" and not os.path.exists(here) and os.path.exists(alt)) else here


def build_library():
    """Compile the HIP extension in-tree (hipcc --offload-arch=gfx950)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "csrc")])
    return library_path()


class SearchParams(C.Structure):
    _fields_ = [("seed_sub_mat", C.c_char_p), ("sub_mat", C.c_char_p), ("sensitivity", C.c_float), ("kmer_score", C.c_int),
                ("max_seqs", C.c_int), ("min_ungapped_score", C.c_int), ("mask", C.c_int), ("gap_open", C.c_int),
                ("gap_extend", C.c_int), ("evalue", C.c_double), ("threads", C.c_int), ("bin_size", C.c_uint)]


class Hit(C.Structure):
    _fields_ = [("seq_id", C.c_uint32), ("score", C.c_int32), ("diagonal", C.c_int16), ("pad", C.c_int16)]


class HitPacked(C.Structure):
    """sp_hit_packed: the 8-byte hit record as the device delivers it"""
    _fields_ = [("seq_id", C.c_uint32), ("score", C.c_uint16), ("diagonal", C.c_int16)]


class Alignment(C.Structure):
    _fields_ = [("db_key", C.c_uint32), ("bit_score", C.c_int32), ("raw_score", C.c_int32), ("seq_id", C.c_float),
                ("evalue", C.c_double), ("q_start", C.c_int32), ("q_end", C.c_int32), ("q_len", C.c_uint32),
                ("db_start", C.c_int32), ("db_end", C.c_int32), ("db_len", C.c_uint32), ("aln_len", C.c_uint32),
                ("cigar_offset", C.c_uint32), ("passed", C.c_int32)]


class Candidate(C.Structure):
    """sp_candidate: best double-hit diagonal of one (query, target), exchanged between target shards"""
    _fields_ = [("query", C.c_uint32), ("seq_id", C.c_uint32), ("kmer_ordinal", C.c_uint32), ("diagonal", C.c_uint16), ("score", C.c_uint16)]


class Rescore(C.Structure):
    _fields_ = [("score", C.c_int32), ("start_pos", C.c_int32), ("end_pos", C.c_int32), ("diagonal_len", C.c_int32),
                ("diagonal", C.c_int32), ("dist_to_diagonal", C.c_uint32), ("evalue", C.c_double), ("bit_score", C.c_int32)]


HOST_MODULES = ("extractorfs", "translatenucs", "reverseseqbycodon", "proteinaln2nucl", "combineprotnuclaln", "besthitperset", "mergeresultsbyset",
                "combinescore", "filtermatchbyfdr", "summarizeresults")      # set-level steps: host-only module drop-ins

class SetLayout(C.Structure):
    _fields_ = [("q_nucl", C.c_char_p), ("q_nucl_off", C.POINTER(C.c_uint64)), ("t_nucl", C.c_char_p), ("t_nucl_off", C.POINTER(C.c_uint64)),
                ("q_set", C.POINTER(C.c_uint32)), ("t_set", C.POINTER(C.c_uint32)), ("nucl_sub_mat", C.c_char_p),
                ("nucl_gap_open", C.c_int), ("nucl_gap_extend", C.c_int), ("simple_best_hit", C.c_int)]


class TextDb(C.Structure):
    _fields_ = [("n", C.c_uint64), ("keys", C.POINTER(C.c_uint32)), ("offsets", C.POINTER(C.c_uint64)), ("text", C.c_void_p)]


_lib = None


def load_library():
    """Load the in-tree shared library.  Fails loudly when it has not been built -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise SpError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(or make -C spacepharer_amd/csrc); there is no CPU fallback" % path)
    L = C.CDLL(path)
    vp = C.c_void_p
    L.sp_init.argtypes = [C.c_int]
    L.sp_last_error.restype = C.c_char_p
    L.sp_version.restype = C.c_char_p
    L.sp_default_params.argtypes = [C.POINTER(SearchParams)]
    L.sp_device_memory.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.sp_ctx_create.restype = vp
    L.sp_ctx_create.argtypes = [C.POINTER(SearchParams)]
    L.sp_ctx_destroy.argtypes = [vp]
    L.sp_ctx_set_targets.argtypes = [vp, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_uint64]
    L.sp_ctx_set_queries.argtypes = [vp, C.c_char_p, C.POINTER(C.c_uint64), C.c_uint64]
    L.sp_ctx_prefilter.argtypes = [vp]
    L.sp_ctx_get_hits.argtypes = [vp, C.POINTER(Hit), C.POINTER(C.c_uint32)]
    L.sp_ctx_max_seqs.argtypes = [vp]
    L.sp_ctx_num_hits.restype = C.c_uint64
    L.sp_ctx_num_hits.argtypes = [vp]
    L.sp_ctx_get_hits_compact.argtypes = [vp, C.POINTER(Hit), C.POINTER(C.c_uint64)]
    L.sp_ctx_hits_view.argtypes = [vp, C.POINTER(C.POINTER(HitPacked)), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint64)]
    L.sp_ctx_set_hit_sink.argtypes = [vp, vp, vp, C.c_uint64]
    L.sp_ctx_set_scores.argtypes = [vp, C.POINTER(SetLayout), C.POINTER(Alignment), C.c_char_p, C.c_int, C.POINTER(TextDb)]
    L.sp_text_db_free.argtypes = [C.POINTER(TextDb)]
    L.sp_ctx_align.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(Alignment), C.c_char_p,
                               C.c_uint64, C.POINTER(C.c_uint64)]
    L.sp_ctx_align_hits.argtypes = [vp, C.POINTER(Alignment), C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.sp_ctx_align_hits_begin.argtypes = [vp, C.POINTER(Alignment), C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.sp_ctx_align_hits_end.argtypes = [vp]
    L.sp_ctx_search_begin.argtypes = [vp, vp, vp, C.c_uint64, vp, vp, C.c_uint64]
    L.sp_ctx_search_end.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.sp_ctx_rescore.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int16), C.c_uint64, C.c_int,
                                 C.POINTER(Rescore)]
    L.sp_ctx_set_targets_shard.argtypes = [vp, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_uint64, C.c_uint64, C.c_uint64]
    L.sp_ctx_share_targets.argtypes = [vp, vp]
    L.sp_ctx_prefilter_candidates.argtypes = [vp]
    L.sp_ctx_shard_occupancy.argtypes = [vp, vp, C.c_int]
    L.sp_ctx_probe_set_shards.argtypes = [vp, vp, C.c_int, C.c_int]
    L.sp_ctx_probe_enumerate.argtypes = [vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    L.sp_ctx_probe_get.argtypes = [vp, C.c_int, vp, vp, C.c_int]
    L.sp_ctx_probe_records_device.argtypes = [vp, C.c_int]; L.sp_ctx_probe_records_device.restype = vp
    L.sp_ctx_probe_query_counts_device.argtypes = [vp, C.c_int]; L.sp_ctx_probe_query_counts_device.restype = vp
    L.sp_ctx_probe_fetch.argtypes = [vp, vp, C.c_uint64, vp, C.c_int]
    L.sp_ctx_probe_search.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_double)]
    L.sp_ctx_num_candidates.restype = C.c_uint64
    L.sp_ctx_overflow_queries.restype = C.c_uint64
    L.sp_ctx_overflow_queries.argtypes = [vp, C.POINTER(C.c_uint32), C.c_uint64]
    L.sp_ctx_num_candidates.argtypes = [vp]
    L.sp_ctx_get_candidates.argtypes = [vp, C.c_void_p]
    L.sp_ctx_candidates_device.restype = vp
    L.sp_ctx_candidates_device.argtypes = [vp]
    L.sp_ctx_merge_candidates.argtypes = [vp, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64]
    L.sp_ctx_stat.restype = C.c_double
    L.sp_ctx_stat.argtypes = [vp, C.c_char_p]
    L.sp_ctx_sync.argtypes = [vp]
    L.sp_flat_db_open.restype = vp
    L.sp_flat_db_open.argtypes = [C.c_char_p]
    L.sp_flat_db_close.argtypes = [vp]
    L.sp_flat_db_entries.restype = C.c_uint64
    L.sp_flat_db_entries.argtypes = [vp]
    for name, rt in (("residues", C.c_void_p), ("offsets", C.POINTER(C.c_uint64)), ("keys", C.POINTER(C.c_uint32)), ("sets", C.POINTER(C.c_uint32))):
        fn = getattr(L, "sp_flat_db_" + name)
        fn.restype = rt
        fn.argtypes = [vp]
    L.sp_ctx_hits_text.argtypes = [vp, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    L.sp_ctx_align_hits_text.argtypes = [vp, C.c_double, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    L.sp_ctx_set_timing.argtypes = [vp, C.c_int]
    L.sp_ctx_set_tuning.argtypes = [vp, C.c_char_p, C.c_int]
    L.sp_ctx_stream.restype = vp
    L.sp_ctx_stream.argtypes = [vp]
    L.sp_ctx_index_digest.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.sp_ctx_get_masked.argtypes = [vp, C.c_char_p]
    L.sp_ctx_ext3_digest.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.sp_host_alloc.restype = vp
    L.sp_host_alloc.argtypes = [C.c_uint64]
    L.sp_host_free.argtypes = [vp]
    L.sp_ctx_diag_stream_read.argtypes = [vp, C.c_uint64, C.c_int, C.POINTER(C.c_double)]
    for f in ("sp_prefilter_run", "sp_align_run", "sp_rescorediagonal_run", "sp_search_run") + tuple("sp_%s_run" % m for m in HOST_MODULES):
        getattr(L, f).argtypes = [C.c_int, C.POINTER(C.c_char_p)]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise SpError("%s (code %d)" % (load_library().sp_last_error().decode(), rc))


def _concat(seqs):
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
    offs = (C.c_uint64 * (len(bs) + 1))()
    t = 0
    for i, b in enumerate(bs):
        offs[i] = t
        t += len(b)
    offs[len(bs)] = t
    return b"".join(bs), offs


def _run_module(fn, args):
    L = load_library()
    _check(L.sp_init(0))
    arr = (C.c_char_p * len(args))(*[a.encode() for a in args])
    rc = getattr(L, fn)(len(args), arr)
    if rc != 0:
        raise SpError("%s failed: %s" % (fn, L.sp_last_error().decode()))
    return rc


def module_host(name, args):
    """one of the set-level steps (HOST_MODULES) with the reference module's argv; needs no device"""
    if name not in HOST_MODULES:
        raise SpError("unknown host module " + name)
    L = load_library()
    arr = (C.c_char_p * len(args))(*[str(a).encode() for a in args])
    rc = getattr(L, "sp_%s_run" % name)(len(args), arr)
    if rc != 0:
        raise SpError("%s failed: %s" % (name, L.sp_last_error().decode()))
    return rc


def module_prefilter(args):
    """`prefilter <queryDB> <targetDB> <outDB> [flags]` -- same argv as the reference module."""
    return _run_module("sp_prefilter_run", list(args))


def module_align(args):
    return _run_module("sp_align_run", list(args))


def module_rescorediagonal(args):
    return _run_module("sp_rescorediagonal_run", list(args))


def module_search(args):
    """`search <queryDB> <targetDB> <alignmentDB> <tmpDir> [flags]`: prefilter + align on one resident context"""
    return _run_module("sp_search_run", list(args))


class SearchContext:
    """matrices -> target set (masked store + k-mer index in HBM) -> query batch -> prefilter / align."""

    def __init__(self, device_mask=0, **kw):
        self.L = load_library()
        _check(self.L.sp_init(device_mask))
        p = SearchParams()
        self.L.sp_default_params(C.byref(p))
        self._keep = []
        for k, v in kw.items():
            if isinstance(v, str):
                v = v.encode()
                self._keep.append(v)
            setattr(p, k, v)
        self.params = p
        self.ctx = self.L.sp_ctx_create(C.byref(p))
        if not self.ctx:
            raise SpError(self.L.sp_last_error().decode())
        self.n_targets = 0
        self.n_queries = 0

    def close(self):
        if self.ctx:
            for p in getattr(self, "_pinned", []):
                self.L.sp_host_free(p)
            self._pinned = []
            self._aln = self._cig = None
            self._aln_cap = 0
            self.L.sp_ctx_destroy(self.ctx)
            self.ctx = None

    def _result_buffers(self, n):
        """page-locked result buffers (sp_host_alloc), grown on demand and reused"""
        if getattr(self, "_aln_cap", 0) < n or getattr(self, "_aln", None) is None:
            self._aln_cap = n + n // 4 + 16
            self._cig_cap = 64 * self._aln_cap
            self._new_pinned_results()

    def reserve_results(self, n_hits):
        """page-locked result buffers for align_hits_raw() / align_hits_begin() with room for n_hits alignments, ahead of the first call (pinning a few hundred MB takes
        0.1-0.3 s: a caller that knows its scale does it beside something else)"""
        self._result_buffers(int(n_hits))

    def _new_pinned_results(self):
        if not hasattr(self, "_pinned"):
            self._pinned = []
        pa = self.L.sp_host_alloc(C.sizeof(Alignment) * self._aln_cap)
        pc = self.L.sp_host_alloc(self._cig_cap)
        self._pinned += [pa, pc]
        self._aln = (Alignment * self._aln_cap).from_address(pa)
        self._cig = (C.c_char * self._cig_cap).from_address(pc)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_targets(self, seqs, keys=None):
        buf, offs = _concat(seqs)
        karr = (C.c_uint32 * len(seqs))(*keys) if keys is not None else None
        _check(self.L.sp_ctx_set_targets(self.ctx, buf, offs, karr, len(seqs)))
        self.n_targets = len(seqs)

    def set_targets_raw(self, buf, offs_u64, n):
        _check(self.L.sp_ctx_set_targets(self.ctx, buf, offs_u64, None, n))
        self.n_targets = n

    # ---- target-sharded search: this context indexes targets [shard_from, shard_to) of the full set
    def share_targets(self, other):
        """search the target set `other` has staged (device tables and set shared, not copied)"""
        _check(self.L.sp_ctx_share_targets(self.ctx, other.ctx))

    def set_targets_shard(self, seqs, shard_from, shard_to, keys=None):
        buf, offs = _concat(seqs)
        karr = (C.c_uint32 * len(seqs))(*keys) if keys is not None else None
        _check(self.L.sp_ctx_set_targets_shard(self.ctx, buf, offs, karr, len(seqs), shard_from, shard_to))
        self.n_targets = len(seqs)

    def set_targets_shard_raw(self, buf, offs_u64, n, shard_from, shard_to):
        _check(self.L.sp_ctx_set_targets_shard(self.ctx, buf, offs_u64, None, n, shard_from, shard_to))
        self.n_targets = n

    def prefilter_candidates(self):
        """match + ungapped scoring against this shard; returns the number of candidates left in HBM"""
        _check(self.L.sp_ctx_prefilter_candidates(self.ctx))
        return int(self.L.sp_ctx_num_candidates(self.ctx))

    # ---- the probe-list cut of the target-sharded search (include/spacepharer_amd.h: sp_ctx_probe_*)
    def shard_occupancy(self):
        """this shard's occupancy map: numpy uint8[SP_SHARD_MAP_BYTES]"""
        import numpy as np
        out = np.zeros(8000000, dtype=np.uint8)
        _check(self.L.sp_ctx_shard_occupancy(self.ctx, out.ctypes.data, 0))
        return out

    def probe_set_shards(self, maps):
        """maps: the shards' occupancy maps in shard order (list of uint8 arrays)"""
        import numpy as np
        cat = np.ascontiguousarray(np.concatenate(maps), dtype=np.uint8)
        _check(self.L.sp_ctx_probe_set_shards(self.ctx, cat.ctypes.data, len(maps), 0))
        self.n_shards = len(maps)

    def probe_enumerate(self, q_from, q_to):
        """queries [q_from, q_to) -> per destination shard (records uint32[n, 4], probes per query of the slice uint32[q_to - q_from]) host copies"""
        import numpy as np
        counts = (C.c_uint64 * 8)()
        _check(self.L.sp_ctx_probe_enumerate(self.ctx, q_from, q_to, counts))
        out = []
        for s in range(self.n_shards):
            r = np.zeros((max(1, counts[s]), 4), dtype=np.uint32); qc = np.zeros(max(1, q_to - q_from), dtype=np.uint32)
            _check(self.L.sp_ctx_probe_get(self.ctx, s, r.ctypes.data, qc.ctypes.data, 0))
            out.append((r[:counts[s]], qc[:q_to - q_from]))
        return out

    def probe_fetch(self, records, query_counts):
        """the probes addressed to this shard + the number of probes per query of the batch -> its candidates (returns their number)"""
        import numpy as np
        records = np.ascontiguousarray(records, dtype=np.uint32).reshape(-1, 4); query_counts = np.ascontiguousarray(query_counts, dtype=np.uint32)
        assert len(query_counts) == self.n_queries
        _check(self.L.sp_ctx_probe_fetch(self.ctx, records.ctypes.data, len(records), query_counts.ctypes.data, 0))
        return int(self.L.sp_ctx_num_candidates(self.ctx))

    def probe_search(self, comm, q_from, q_to):
        """collective over `comm` (rank r = shard r); returns (candidates of this shard, seconds of exchange)"""
        x = C.c_double()
        _check(self.L.sp_ctx_probe_search(self.ctx, comm, q_from, q_to, C.byref(x)))
        return int(self.L.sp_ctx_num_candidates(self.ctx)), x.value

    def candidates(self):
        """numpy uint8[n, 16] view of this shard's sp_candidate records (host copy)"""
        import numpy as np
        n = int(self.L.sp_ctx_num_candidates(self.ctx))
        out = np.zeros((max(1, n), C.sizeof(Candidate)), dtype=np.uint8)
        _check(self.L.sp_ctx_get_candidates(self.ctx, out.ctypes.data))
        return out[:n]

    def index_digest(self):
        """(entries, FNV-1a digest) of the device-built index, comparable with the reference's IndexTable dump"""
        n, h = C.c_uint64(), C.c_uint64()
        _check(self.L.sp_ctx_index_digest(self.ctx, C.byref(n), C.byref(h)))
        return int(n.value), int(h.value)

    def ext3_digest(self):
        h = C.c_uint64()
        _check(self.L.sp_ctx_ext3_digest(self.ctx, C.byref(h)))
        return int(h.value)

    def masked_targets(self):
        """masked target residues (letters, concatenated) as the device holds them"""
        buf = C.create_string_buffer(int(self.stat("target_residues")) + 1)
        _check(self.L.sp_ctx_get_masked(self.ctx, buf))
        return buf.raw[:int(self.stat("target_residues"))].decode()

    def candidates_into(self, ptr):
        """copy this shard's candidates to `ptr` (host or device address with room for num_candidates * 16 bytes)"""
        _check(self.L.sp_ctx_get_candidates(self.ctx, ptr))

    def candidates_device(self):
        return self.L.sp_ctx_candidates_device(self.ctx), int(self.L.sp_ctx_num_candidates(self.ctx))

    def merge_candidates(self, cands, q_from=0, q_to=None):
        """cands: numpy uint8[n, 16] (host) -- the union of all shards' candidates"""
        import numpy as np
        cands = np.ascontiguousarray(cands, dtype=np.uint8)
        _check(self.L.sp_ctx_merge_candidates(self.ctx, cands.ctypes.data, len(cands), 0, q_from, self.n_queries if q_to is None else q_to))

    def merge_candidates_device(self, ptr, n, q_from=0, q_to=None):
        _check(self.L.sp_ctx_merge_candidates(self.ctx, ptr, n, 1, q_from, self.n_queries if q_to is None else q_to))

    def set_queries(self, seqs):
        buf, offs = _concat(seqs)
        _check(self.L.sp_ctx_set_queries(self.ctx, buf, offs, len(seqs)))
        self.n_queries = len(seqs)

    def set_queries_raw(self, buf, offs_u64, n):
        _check(self.L.sp_ctx_set_queries(self.ctx, buf, offs_u64, n))
        self.n_queries = n

    def prefilter(self):
        _check(self.L.sp_ctx_prefilter(self.ctx))

    def num_hits(self):
        return int(self.L.sp_ctx_num_hits(self.ctx))

    def hits_raw(self):
        ms = self.L.sp_ctx_max_seqs(self.ctx)
        hits = (Hit * (max(1, self.n_queries * ms)))()
        counts = (C.c_uint32 * max(1, self.n_queries))()
        _check(self.L.sp_ctx_get_hits(self.ctx, hits, counts))
        return hits, counts, ms

    def hits_compact(self):
        """(Hit array of all hits grouped by query, uint64 offsets[n_queries+1])"""
        total = int(self.L.sp_ctx_num_hits(self.ctx))
        if getattr(self, "_hit_cap", 0) < total or getattr(self, "_hit_nq", -1) != self.n_queries:      # buffers are reused from call to call
            self._hit_cap = total + total // 4 + 16
            self._hit_nq = self.n_queries
            self._hits = (Hit * self._hit_cap)()
            self._hit_offs = (C.c_uint64 * (self.n_queries + 1))()
        _check(self.L.sp_ctx_get_hits_compact(self.ctx, self._hits, self._hit_offs))
        return self._hits, self._hit_offs, total

    def hits_view(self):
        """(HitPacked array, uint32 offsets[n_queries+1], total) -- the library's page-locked delivery buffers, no copy;
        valid until the next prefilter()"""
        hp, op, n = C.POINTER(HitPacked)(), C.POINTER(C.c_uint32)(), C.c_uint64()
        _check(self.L.sp_ctx_hits_view(self.ctx, C.byref(hp), C.byref(op), C.byref(n)))
        return hp, op, int(n.value)

    def set_hit_sink(self, hits_ptr, offsets_ptr, cap_hits):
        """deliver the following passes' hit lists to caller-owned page-locked memory (addresses; None switches back)"""
        _check(self.L.sp_ctx_set_hit_sink(self.ctx, hits_ptr, offsets_ptr, cap_hits))

    def hits(self):
        hits, offs, _ = self.hits_compact()
        return [[(hits[i].seq_id, hits[i].score, hits[i].diagonal) for i in range(offs[q], offs[q + 1])]
                for q in range(self.n_queries)]

    def align(self, pairs):
        n = len(pairs)
        q = (C.c_uint32 * max(1, n))(*[p[0] for p in pairs])
        t = (C.c_uint32 * max(1, n))(*[p[1] for p in pairs])
        res = (Alignment * max(1, n))()
        cap = 1 << 16
        while True:
            buf = C.create_string_buffer(cap)
            used = C.c_uint64()
            rc = self.L.sp_ctx_align(self.ctx, q, t, n, res, buf, cap, C.byref(used))
            if rc == -5 and used.value > cap:
                cap = int(used.value) + 16
                continue
            _check(rc)
            break
        out = []
        for i in range(n):
            r = res[i]
            cig = C.string_at(C.addressof(buf) + r.cigar_offset).decode() if r.aln_len else ""
            out.append((r, cig))
        return out

    def _text(self, call):
        """result-DB text formatted on the device: list of entries (bytes, without the closing NUL), one per query"""
        lens = (C.c_uint32 * max(1, self.n_queries))()
        used = C.c_uint64()
        cap = 1 << 16
        while True:
            buf = C.create_string_buffer(cap)
            rc = call(buf, cap, C.byref(used), lens)
            if rc == -5 and used.value > cap:
                cap = int(used.value) + 16
                continue
            _check(rc)
            break
        raw, out, at = buf.raw, [], 0
        for q in range(self.n_queries):
            out.append(raw[at:at + lens[q] - 1])
            assert raw[at + lens[q] - 1] == 0
            at += lens[q]
        assert at == used.value
        return out

    def hits_text(self):
        return self._text(lambda buf, cap, used, lens: self.L.sp_ctx_hits_text(self.ctx, buf, cap, used, lens))

    def align_hits_text(self, min_seq_id=0.0, min_aln_len=0):
        return self._text(lambda buf, cap, used, lens: self.L.sp_ctx_align_hits_text(self.ctx, min_seq_id, min_aln_len, buf, cap, used, lens))

    def sync(self):
        _check(self.L.sp_ctx_sync(self.ctx))

    def set_tuning(self, name, value):
        """knobs that never change a result, e.g. ("match_blocks_per_cu", 2)"""
        _check(self.L.sp_ctx_set_tuning(self.ctx, name.encode(), int(value)))

    def set_timing(self, level):
        """0: no HIP events inside the pipeline, 1: around the matching kernel only (default), 2: every stage"""
        _check(self.L.sp_ctx_set_timing(self.ctx, level))

    def diag_stream_read(self, nbytes, lane_bytes):
        """profiling aid: milliseconds of one streaming read of nbytes (FETCH_SIZE calibration)"""
        ms = C.c_double()
        _check(self.L.sp_ctx_diag_stream_read(self.ctx, nbytes, lane_bytes, C.byref(ms)))
        return ms.value

    def align_raw(self, qidx, tidx):
        """numpy uint32 arrays in, (Alignment array, cigar buffer) out -- no per-pair Python work"""
        n = len(qidx)
        self._result_buffers(n)
        used = C.c_uint64()
        qp = qidx.ctypes.data_as(C.POINTER(C.c_uint32))
        tp = tidx.ctypes.data_as(C.POINTER(C.c_uint32))
        while True:
            rc = self.L.sp_ctx_align(self.ctx, qp, tp, n, self._aln, C.cast(self._cig, C.c_char_p), self._cig_cap, C.byref(used))
            if rc == -5 and used.value > self._cig_cap:
                self._cig_cap = int(used.value) * 2
                self._new_pinned_results()
                continue
            _check(rc)
            return self._aln, self._cig

    def align_hits_raw(self):
        """align all hits of the last prefilter(); results follow hits_compact() order"""
        n = int(self.L.sp_ctx_num_hits(self.ctx))
        self._result_buffers(n)
        used = C.c_uint64()
        while True:
            rc = self.L.sp_ctx_align_hits(self.ctx, self._aln, C.cast(self._cig, C.c_char_p), self._cig_cap, C.byref(used))
            if rc == -5 and used.value > self._cig_cap:
                self._cig_cap = int(used.value) * 2
                self._new_pinned_results()
                continue
            _check(rc)
            return self._aln, self._cig, n

    def align_hits_begin(self, into=None):
        """like align_hits_raw(), but returns while the results are still on their way; align_hits_end() collects them.
        into = (records address, cigar address, cigar capacity, ctypes c_uint64 for the cigar bytes used): caller-owned page-locked
        buffers with room for num_hits() records; no retry when the cigar pool is too small (SpError)"""
        n = int(self.L.sp_ctx_num_hits(self.ctx))
        if into is not None:
            _check(self.L.sp_ctx_align_hits_begin(self.ctx, C.cast(into[0], C.POINTER(Alignment)), C.cast(into[1], C.c_char_p), into[2], C.byref(into[3])))
            return n
        self._result_buffers(n)
        used = C.c_uint64()
        while True:
            rc = self.L.sp_ctx_align_hits_begin(self.ctx, self._aln, C.cast(self._cig, C.c_char_p), self._cig_cap, C.byref(used))
            if rc == -5 and used.value > self._cig_cap:
                _check(self.L.sp_ctx_align_hits_end(self.ctx))
                self._cig_cap = int(used.value) * 2
                self._new_pinned_results()
                continue
            _check(rc)
            return n

    def align_hits_end(self):
        _check(self.L.sp_ctx_align_hits_end(self.ctx))
        return getattr(self, "_aln", None), getattr(self, "_cig", None)

    def search_begin(self, hits_ptr, offsets_ptr, hit_cap, records_ptr, cigars_ptr, cigar_cap):
        """queue one whole step (prefilter + align of every hit + delivery into the caller's page-locked buffers: addresses) without a host
        round trip; up to two steps in flight; search_end() collects the oldest"""
        _check(self.L.sp_ctx_search_begin(self.ctx, hits_ptr, offsets_ptr, hit_cap, records_ptr, cigars_ptr, cigar_cap))

    def search_end(self):
        """-> (hits of the oldest step in flight, bytes of its backtrace pool); its results are complete in the buffers given to search_begin()"""
        n, used = C.c_uint64(), C.c_uint64()
        _check(self.L.sp_ctx_search_end(self.ctx, C.byref(n), C.byref(used)))
        return int(n.value), int(used.value)

    def set_scores(self, q_nucl, t_nucl, q_set, t_set, aln, cig, stage=0, simple_best_hit=True):
        """predictmatch's chain after the search on the records in memory (sp_ctx_set_scores); q_nucl / t_nucl: lists of nucleotide
        ORFs, q_set / t_set: set key per query / target, aln / cig: what align_hits_raw() returned.  -> {key: entry text}"""
        return self.set_scores_raw(_concat(q_nucl), _concat(t_nucl), (C.c_uint32 * max(1, len(q_set)))(*q_set), (C.c_uint32 * max(1, len(t_set)))(*t_set), aln, cig, stage, simple_best_hit)

    def set_scores_raw(self, q_nucl, t_nucl, q_set, t_set, aln, cig, stage=0, simple_best_hit=True):
        """set_scores() on buffers the caller already holds: q_nucl / t_nucl = (bytes, ctypes uint64 offsets[n + 1]) as _concat()
        returns them, q_set / t_set = ctypes uint32 arrays"""
        lay = SetLayout(q_nucl[0], q_nucl[1], t_nucl[0], t_nucl[1], q_set, t_set, None, 0, 0, 1 if simple_best_hit else 0)
        out = TextDb()
        _check(self.L.sp_ctx_set_scores(self.ctx, C.byref(lay), aln, C.cast(cig, C.c_char_p), stage, C.byref(out)))
        try:
            return {int(out.keys[i]): C.string_at(out.text + out.offsets[i]).decode() for i in range(out.n)}
        finally:
            self.L.sp_text_db_free(C.byref(out))

    def rescore(self, triples, mode):
        n = len(triples)
        q = (C.c_uint32 * max(1, n))(*[p[0] for p in triples])
        t = (C.c_uint32 * max(1, n))(*[p[1] for p in triples])
        d = (C.c_int16 * max(1, n))(*[p[2] for p in triples])
        res = (Rescore * max(1, n))()
        _check(self.L.sp_ctx_rescore(self.ctx, q, t, d, n, mode, res))
        return [res[i] for i in range(n)]

    def overflow_queries(self):
        """positions of the queries the last prefilter() refused (SP_ERR_OVERFLOW)"""
        n = int(self.L.sp_ctx_overflow_queries(self.ctx, None, 0))
        out = (C.c_uint32 * max(1, n))()
        self.L.sp_ctx_overflow_queries(self.ctx, out, n)
        return [int(out[i]) for i in range(n)]

    def stat(self, name):
        return self.L.sp_ctx_stat(self.ctx, name.encode())


class FlatDb:
    """A sequence DB read and flattened by the library's DB reader (sp_flat_db_open): residues / offsets / keys / sets as ctypes views that live as long as this object."""

    def __init__(self, path):
        self.L = load_library()
        self.h = self.L.sp_flat_db_open(path.encode())
        if not self.h:
            raise SpError(self.L.sp_last_error().decode())
        self.n = int(self.L.sp_flat_db_entries(self.h))
        self.residues = C.cast(self.L.sp_flat_db_residues(self.h), C.c_char_p)
        self.offsets = self.L.sp_flat_db_offsets(self.h)
        self.keys = self.L.sp_flat_db_keys(self.h)
        self.sets = self.L.sp_flat_db_sets(self.h)

    def close(self):
        if self.h:
            self.L.sp_flat_db_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001
            pass
