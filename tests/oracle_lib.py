"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE: the CPU restatement used as the checker)."""
import ctypes as C
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def matrix_text(name):
    """Render tests/golden/matrices.json back into the `.out` text layout the matrix parser reads."""
    m = json.load(open(os.path.join(GOLDEN, "matrices.json")))[name]
    lines = ["# %s" % name,
             "# Background (precomputed optional): " + " ".join(m["background"]),
             "# Lambda     (precomputed optional): " + m["lambda"],
             "\t" + "\t".join(m["letters"])]
    for l, row in zip(m["letters"], m["scores"]):
        lines.append(l + "\t" + "\t".join(str(x) for x in row))
    return "\n".join(lines) + "\n"


class Hit(C.Structure):
    _fields_ = [("seqId", C.c_uint32), ("score", C.c_int32), ("diagonal", C.c_uint16)]


class Aln(C.Structure):
    _fields_ = [("dbKey", C.c_uint32), ("score", C.c_int32), ("rawScore", C.c_int32), ("qcov", C.c_float),
                ("dbcov", C.c_float), ("seqId", C.c_float), ("eval", C.c_double), ("alnLength", C.c_uint32),
                ("qStartPos", C.c_int32), ("qEndPos", C.c_int32), ("qLen", C.c_uint32), ("dbStartPos", C.c_int32),
                ("dbEndPos", C.c_int32), ("dbLen", C.c_uint32), ("passed", C.c_int32), ("backtrace", C.c_char * 4096)]


class Alp(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("lambda_", "K", "a_I", "b_I", "alpha_I", "beta_I", "a_J", "b_J",
                                          "alpha_J", "beta_J", "sigma", "tau")]

    @classmethod
    def from_ref(cls, d):
        names = ["lambda", "K", "a_I", "b_I", "alpha_I", "beta_I", "a_J", "b_J", "alpha_J", "beta_J", "sigma", "tau"]
        return cls(*[float.fromhex(d[n]) for n in names])


class Local(C.Structure):
    _fields_ = [("score", C.c_int), ("startPos", C.c_int), ("endPos", C.c_int), ("diagonalLen", C.c_int),
                ("diagonal", C.c_int), ("distToDiagonal", C.c_uint)]


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(ORACLE_DIR, "sp_oracle.c")):
            build()
        L = C.CDLL(path)
        vp, u64p, u32p, u16p = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint16)
        L.spo_create.restype = vp
        L.spo_create.argtypes = [C.c_char_p, C.c_char_p, C.c_float, C.c_int, C.c_uint, C.c_uint64]
        L.spo_destroy.argtypes = [vp]
        L.spo_kmer_thr.argtypes = [vp]
        L.spo_bin_size.argtypes = [vp]; L.spo_bin_size.restype = C.c_uint
        for f in ("spo_seed_matrix", "spo_ungapped_matrix", "spo_aln_matrix", "spo_ext3_score", "spo_ext3_index",
                  "spo_masked_residues", "spo_target_offsets", "spo_index_offsets", "spo_index_seqids", "spo_index_positions"):
            getattr(L, f).restype = vp; getattr(L, f).argtypes = [vp]
        L.spo_set_targets.argtypes = [vp, C.c_char_p, u64p, C.c_size_t, C.c_int]
        L.spo_num_targets.restype = C.c_size_t; L.spo_num_targets.argtypes = [vp]
        for f in ("spo_aa_db_size", "spo_index_entries", "spo_index_digest", "spo_masked_count"):
            getattr(L, f).restype = C.c_uint64; getattr(L, f).argtypes = [vp]
        L.spo_kmer_list.restype = C.c_size_t
        L.spo_kmer_list.argtypes = [vp, C.c_char_p, C.c_int, u32p, C.c_size_t]
        L.spo_match.restype = C.c_size_t
        L.spo_match.argtypes = [vp, C.c_char_p, C.c_int, u32p, u16p, C.c_size_t, u64p, u64p]
        L.spo_prefilter_query.restype = C.c_size_t
        L.spo_prefilter_query.argtypes = [vp, C.c_char_p, C.c_int, C.POINTER(Hit), C.c_size_t]
        L.spo_ungapped_score.argtypes = [vp, C.c_char_p, C.c_int, C.c_uint32, C.c_uint16, C.POINTER(C.c_int)]
        L.spo_prefilter_text.restype = C.c_size_t
        L.spo_prefilter_text.argtypes = [C.POINTER(Hit), C.c_size_t, u32p, C.c_char_p, C.c_size_t]
        L.spo_set_alp.argtypes = [vp, C.POINTER(Alp), C.POINTER(Alp)]
        L.spo_evalue.restype = C.c_double; L.spo_evalue.argtypes = [vp, C.c_int, C.c_double, C.c_double]
        L.spo_bitscore.restype = C.c_double; L.spo_bitscore.argtypes = [vp, C.c_int, C.c_double]
        L.spo_align_pair.argtypes = [vp, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_uint32, C.c_int, C.c_int,
                                     C.c_double, C.POINTER(Aln)]
        L.spo_sw_forward.argtypes = [vp, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 4
        L.spo_alignment_text.restype = C.c_size_t
        L.spo_alignment_text.argtypes = [C.POINTER(Aln), C.c_size_t, C.c_char_p, C.c_size_t]
        L.spo_rescore_diagonal.argtypes = [vp, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_uint16, C.c_int, C.POINTER(Local)]
        L.spo_map_sequence.argtypes = [vp, C.c_char_p, C.c_int, C.c_char_p]
        L.spo_local_bias.argtypes = [vp, C.c_char_p, C.c_int, C.POINTER(C.c_float)]
        L.spo_window_thresholds.argtypes = [vp, C.c_char_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.spo_fnv1a.restype = C.c_uint64; L.spo_fnv1a.argtypes = [vp, C.c_size_t, C.c_uint64]
        _lib = L
    return _lib


class Matrix(C.Structure):
    _fields_ = [("alphabetSize", C.c_int), ("sub", (C.c_short * 21) * 21), ("pBack", C.c_double * 21),
                ("prob", (C.c_double * 21) * 21), ("lambda_", C.c_double), ("aa2num", C.c_ubyte * 256),
                ("num2aa", C.c_char * 22)]


class Oracle:
    """The whole CPU path for one target set (mirrors what prefilter/align do per process)."""

    def __init__(self, sens=5.7, max_seqs=300, bin_size=0, l2=2097152):
        self.L = lib()
        self.ctx = self.L.spo_create(matrix_text("VTML80").encode(), matrix_text("VTML40").encode(), sens, max_seqs, bin_size, l2)
        assert self.ctx, "matrix parse failed"
        self.targets = []

    def close(self):
        if self.ctx:
            self.L.spo_destroy(self.ctx); self.ctx = None

    def matrix(self, which):
        p = getattr(self.L, "spo_%s_matrix" % which)(self.ctx)
        return C.cast(p, C.POINTER(Matrix)).contents

    def set_targets(self, seqs, mask=1):
        self.targets = [s.encode() if isinstance(s, str) else s for s in seqs]
        offs = [0]
        for s in self.targets:
            offs.append(offs[-1] + len(s))
        buf = b"".join(self.targets)
        arr = (C.c_uint64 * len(offs))(*offs)
        rc = self.L.spo_set_targets(self.ctx, buf, arr, len(self.targets), mask)
        assert rc == 0
        self.offsets = offs

    def set_alp(self, gapped, gapless):
        self._g, self._gl = gapped, gapless
        self.L.spo_set_alp(self.ctx, C.byref(gapped), C.byref(gapless))

    def masked(self):
        n = self.offsets[-1]
        raw = C.string_at(self.L.spo_masked_residues(self.ctx), n)
        num2aa = self.matrix("seed").num2aa.decode()
        return ["".join(num2aa[b] for b in raw[self.offsets[i]:self.offsets[i + 1]]) for i in range(len(self.targets))]

    def match(self, q):
        q = q.encode()
        cap = 1 << 16
        ids = (C.c_uint32 * cap)(); dg = (C.c_uint16 * cap)()
        dbm = C.c_uint64(); kl = C.c_uint64()
        n = self.L.spo_match(self.ctx, q, len(q), ids, dg, cap, C.byref(dbm), C.byref(kl))
        assert n <= cap
        return [[ids[i], dg[i]] for i in range(n)], dbm.value, kl.value

    def prefilter(self, q):
        q = q.encode()
        cap = 4096
        hits = (Hit * cap)()
        n = self.L.spo_prefilter_query(self.ctx, q, len(q), hits, cap)
        assert n <= cap
        return hits, n

    def prefilter_list(self, q):
        hits, n = self.prefilter(q)
        return [[hits[i].seqId, hits[i].score, C.c_int16(hits[i].diagonal).value] for i in range(n)]

    def prefilter_text(self, q):
        hits, n = self.prefilter(q)
        buf = C.create_string_buffer(64 * (n + 1))
        k = self.L.spo_prefilter_text(hits, n, None, buf, len(buf))
        return buf.raw[:k].decode()

    def kmer_list(self, kmer6, thr):
        cap = 1 << 22
        out = (C.c_uint32 * cap)()
        n = self.L.spo_kmer_list(self.ctx, bytes(kmer6), thr, out, cap)
        return out, n

    def align(self, q, target_id, gap_open=16, gap_extend=2, evalue=200.0):
        r = Aln()
        t = self.targets[target_id]
        rc = self.L.spo_align_pair(self.ctx, q.encode(), len(q), t, len(t), target_id, gap_open, gap_extend, evalue, C.byref(r))
        assert rc == 0, rc
        return r

    def alignment_text(self, alns):
        arr = (Aln * len(alns))(*alns)
        buf = C.create_string_buffer(8192 * (len(alns) + 1))
        k = self.L.spo_alignment_text(arr, len(alns), buf, len(buf))
        return buf.raw[:k].decode()

    def rescore(self, q, target_id, diag, mode):
        r = Local()
        t = self.targets[target_id]
        self.L.spo_rescore_diagonal(self.ctx, q.encode(), len(q), t, len(t), diag & 0xFFFF, mode, C.byref(r))
        return r
