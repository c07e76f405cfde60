"""TEST INFRASTRUCTURE ONLY -- never imported by the product (spacepharer_amd/), only by tests/, __graft_entry__.smoke() and
bench.py's checking legs.

CPU restatement (pure Python: these are text / small-integer steps and the cases are small) of the set-level steps around
SpacePHARER's search hot path -- SURVEY.md section 8 rows f1-f4.  Entries are the text of MMseqs2 DB entries, DBs are
dicts {key: entry text}.  Parity PINNED: tests/test_oracle_sets.py checks every function against
tests/golden/examples.json.gz, which oracle/_ref/ref_sets produced with the reference's own classes
(tests/golden/make_examples.py).

Reference lines restated (paths under /root/reference):
  find_orfs            lib/mmseqs/src/commons/Orf.cpp:118-148 (setSequence), :200-330 (findForward), :437-457 (writeOrfHeader)
  extract_orfs         lib/mmseqs/src/util/extractorfs.cpp:66-125 + DBWriter::createRenumberedDB (commons/DBWriter.cpp:681-739)
  translate            lib/mmseqs/src/commons/TranslateNucl.h:334-503, lib/mmseqs/src/util/translatenucs.cpp:77-109
  reverse_by_codon     src/util/ReverseSeqByCodon.cpp:39-52
  protein_aln_to_nucl  lib/mmseqs/src/util/proteinaln2nucl.cpp:99-187; statistics lib/mmseqs/src/alignment/EvalueComputation.h:18-40,
                       lib/mmseqs/lib/alp/sls_pvalues.cpp:366-560; matrix lib/mmseqs/src/commons/NucleotideMatrix.cpp:18-63
  combine_prot_nucl    src/util/CombineProtNuclAln.cpp:55-97
  best_hit_per_set     lib/mmseqs/src/multihit/besthitperset.cpp:29-115, Aggregation.cpp:24-45
  merge_by_set         lib/mmseqs/src/util/mergeresultsbyset.cpp:32-50
  combine_score        src/util/CombineScore.cpp:41-98
  fdr_filter           src/util/FilterMatchbyFdr.cpp:49-252
  summarize            src/util/SummarizeResults.cpp:66-180 (no taxonomy DB, --report-fdr 0)
"""
import math
import struct
import sys

DBL_MIN = sys.float_info.min
DBL_MAX = sys.float_info.max
FLT_MAX = 3.4028234663852886e+38
CHAR_MAX = "\x7f"


def fmt_e(x):
    """SSTR(double): "%.3E" (lib/mmseqs/src/commons/Util.cpp:713-718)"""
    if math.isinf(x):
        return "INF" if x > 0 else "-INF"
    if math.isnan(x):
        return "NAN"
    return "%.3E" % x


def strtod(s):
    s = s.strip()
    end = 0
    for i in range(len(s), 0, -1):      # longest numeric prefix, like strtod
        try:
            float(s[:i])
            end = i
            break
        except ValueError:
            continue
    return float(s[:end]) if end else 0.0


def atou(s):
    n = 0
    for ch in s:
        if not ch.isdigit():
            break
        n = n * 10 + ord(ch) - 48
    return n


# ---------------------------------------------------------------------------------------------------------------------
# ORFs
# ---------------------------------------------------------------------------------------------------------------------
_COMP = dict(zip("ABCDGHKMNRSTUVWY", "TVGHCDMKNYSAABWR"))


def complement(c):
    if c in _COMP:
        return _COMP[c]
    if c.upper() in _COMP and c.islower():
        return _COMP[c.upper()].lower()
    return "."


def _strands(seq):
    fwd = "".join("t" if c == "u" else c for c in seq)
    rev = "".join("N" if complement(c) == "." else complement(c) for c in reversed(fwd))
    return fwd, rev


def _scan(s, L, strand, min_len, max_len, max_gaps, frames, start_mode, out):
    s = s + CHAR_MAX * 8
    inside, has_start = [True] * 3, [False] * 3
    gaps, count, start = [0] * 3, [0] * 3, [0, 1, 2]
    up = lambda c: c if c == CHAR_MAX else chr(ord(c) & ~0x20)      # noqa: E731
    for i in range(0, L - 2, 3):
        for pos in range(i, i + 3):
            c = [up(s[pos + k]) for k in range(3)]
            fr = pos % 3
            if not frames & (1 << fr):
                continue
            incomplete = CHAR_MAX in c
            last = not incomplete and CHAR_MAX in s[pos + 3:pos + 6]
            is_start = c == ["A", "T", "G"]
            if start_mode == 0:
                begin = (not inside[fr]) and is_start
            elif start_mode == 1:
                begin = not inside[fr]
            else:
                begin = is_start
            if begin:
                inside[fr], has_start[fr], start[fr], gaps[fr], count[fr] = True, True, pos, 0, 0
            stop = "".join(c) in ("TAA", "TAG", "TGA")
            if inside[fr]:
                if not stop:
                    count[fr] += 1
                if any(x == "N" or complement(x) == "." for x in c):
                    gaps[fr] += 1
            if inside[fr] and (stop or last):
                inside[fr] = False
                if count[fr] == 0 and stop:
                    continue
                to = pos + 2 if (last and not stop) else pos - 1
                if gaps[fr] > max_gaps or count[fr] > max_len or count[fr] < min_len:
                    continue
                out.append((start[fr], to, not has_start[fr], not stop, strand))


def find_orfs(seq, min_len, start_mode, max_len=32734, max_gaps=2 ** 31 - 1, frames=7):
    """[(sequence, from, to, incomplete_start, incomplete_end)] in the reference's order; from/to on the contig's plus strand"""
    L = len(seq)
    if L < 3:
        return []
    fwd, rev = _strands(seq)
    locs = []
    _scan(fwd, L, 1, min_len, max_len, max_gaps, frames, start_mode, locs)
    _scan(rev, L, -1, min_len, max_len, max_gaps, frames, start_mode, locs)
    out = []
    for a, b, inc_s, inc_e, strand in locs:
        s = (fwd if strand > 0 else rev)[a:b + 1]
        f, t = (a, b) if strand > 0 else ((L - 1) - a, (L - 1) - b)
        out.append((s, f, t, inc_s, inc_e))
    return out


def orf_header(key, f, t, inc_s, inc_e):
    h = "%d\t%d%s%d" % (key, f, "+" if f < t else "-", abs(f - t))
    complete = (1 if inc_s else 0) | (2 if inc_e else 0)
    return h + ("\t%d" % complete if complete else "") + "\n"


def extract_orfs(nucl_db, lookup_text, min_len, start_mode):
    """-> (orf sequence DB, header DB, lookup text); ORFs renumbered 0.. in (contig key, discovery) order"""
    set_of = {}
    for line in lookup_text.splitlines():
        f = line.split("\t")
        set_of[int(f[0])] = f[2]
    seqs, heads, lookup, k = {}, {}, [], 0
    for key in sorted(nucl_db):
        seq = nucl_db[key].rstrip("\n")
        for s, f, t, inc_s, inc_e in find_orfs(seq, min_len, start_mode):
            seqs[k] = s + "\n"
            heads[k] = orf_header(key, f, t, inc_s, inc_e)
            lookup.append("%d\t%d\t%s\n" % (k, key, set_of.get(key, "0")))
            k += 1
    return seqs, heads, "".join(lookup)


# ---------------------------------------------------------------------------------------------------------------------
# translation
# ---------------------------------------------------------------------------------------------------------------------
_CODE = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
_TCAG = {8: 0, 2: 1, 1: 2, 4: 3}      # ncbi4na bit -> position in TCAG order
_BASE = {}
for _i, _ch in enumerate("-ACMGRSVTWYHKDBN"):
    _BASE[_ch] = _i
    _BASE[_ch.lower()] = _i
_BASE.update({"U": 8, "u": 8, "X": 15, "x": 15})
for _i in range(16):
    _BASE[chr(_i)] = _i               # the ncbi4na byte values themselves (this makes '\n' read as 'Y')


def _codon(a, b, c):
    aa = None
    for x in (1, 2, 4, 8):
        if not x & a:
            continue
        for y in (1, 2, 4, 8):
            if not y & b:
                continue
            for z in (1, 2, 4, 8):
                if not z & c:
                    continue
                ch = _CODE[16 * _TCAG[x] + 4 * _TCAG[y] + _TCAG[z]]
                if aa is None:
                    aa = ch
                elif aa != ch:
                    if aa in "BDN" and ch in "DN":
                        aa = "B"
                    elif aa in "ZEQ" and ch in "EQ":
                        aa = "Z"
                    elif aa in "JIL" and ch in "IL":
                        aa = "J"
                    else:
                        aa = "X"
    return aa or "X"


def translate_entry(entry):
    """entry text of a nucleotide DB ("ACGT...\\n") -> entry text of the translated DB, or None when the entry is skipped"""
    if entry == "":
        return None
    data = entry + "\0\0\0"
    length = len(entry)
    if (data[length] != "\n" and length % 3 != 0) and (data[length - 1] == "\n" and (length - 1) % 3 != 0):
        length -= length % 3
    if length < 3:
        return None
    length = min(length, 3 * 65535)
    out = []
    for i in range(0, length - 2, 3):
        c = data[i:i + 3]
        r = _codon(*(_BASE.get(x, 0) for x in c))
        out.append(r.lower() if any(x.islower() for x in c) else r)
    return "".join(out[:length // 3]) + "\n"


def reverse_by_codon(entry):
    seq = entry.rstrip("\n")
    L = len(seq)
    return "".join(seq[L - d - 3:L - d] for d in range(0, L - 2, 3)) + "\n"


# ---------------------------------------------------------------------------------------------------------------------
# f1: proteinaln2nucl
# ---------------------------------------------------------------------------------------------------------------------
class Evaluer:
    """EvalueComputation with given ALP parameters (12 doubles: lambda K a_I b_I alpha_I beta_I a_J b_J alpha_J beta_J sigma tau)"""

    def __init__(self, alp, db_residues):
        (self.lam, self.K, self.a_I, self.b_I, self.alpha_I, self.beta_I, self.a_J, self.b_J, self.alpha_J, self.beta_J, self.sigma, self.tau) = alp
        self.m = float(db_residues)
        self.logK = math.log(self.K)

    def area(self, y, n):
        const = 1 / math.sqrt(2.0 * 3.1415926535897932384626433832795)
        vi_thr, vj_thr, c_thr = max(2.0 * self.alpha_I / self.lam, 0.0), max(2.0 * self.alpha_J / self.lam, 0.0), max(2.0 * self.sigma / self.lam, 0.0)
        m_li = self.m - (self.a_I * y + self.b_I)
        sq_vi = math.sqrt(max(vi_thr, self.alpha_I * y + self.beta_I))
        m_f = 1e100 if sq_vi == 0.0 else m_li / sq_vi
        p_mf = 0.5 * math.erfc(-math.sqrt(0.5) * m_f)
        e_mf = -const * math.exp(-0.5 * m_f * m_f)
        p1 = m_li * p_mf - sq_vi * e_mf
        n_lj = n - (self.a_J * y + self.b_J)
        sq_vj = math.sqrt(max(vj_thr, self.alpha_J * y + self.beta_J))
        n_f = 1e100 if sq_vj == 0.0 else n_lj / sq_vj
        p_nf = 0.5 * math.erfc(-math.sqrt(0.5) * n_f)
        e_nf = -const * math.exp(-0.5 * n_f * n_f)
        p2 = n_lj * p_nf - sq_vj * e_nf
        c_y = max(c_thr, self.sigma * y + self.tau)
        return p1 * p2 + c_y * (p_mf * p_nf)

    def evalue(self, score, n):
        return self.K * math.exp(-self.lam * score) * self.area(score, n)

    def bit_score(self, score):
        return (self.lam * score - self.logK) / math.log(2.0)


def _nucl_class(c):
    u = c.upper()
    if u == "A":
        return 0
    if u in "CMYH":
        return 1
    if u in "TUW":
        return 2
    if u in "GKBDVRS":
        return 3
    return 4


def nucl_score(a, b):
    """nucleotide.out through NucleotideMatrix(.., 1.0, 0.0): +2 for equal A/C/G/T classes, -3 otherwise"""
    x, y = _nucl_class(a), _nucl_class(b)
    return 2 if (x == y and x < 4) else -3


def seq_id_text(x):
    """Util::fastSeqIdToBuffer as the result writer uses it (lib/mmseqs/src/commons/Util.cpp:278-306)"""
    if x == 1.0:
        return "1.00"
    return "0." + ("0" if x < 0.10 else "") + ("0" if x < 0.01 else "") + str(int(_f32(x * 1000)))      # float * int: a float product


def _f32(x):
    import struct
    return struct.unpack("f", struct.pack("f", x))[0]


def protein_aln_to_nucl(entry, q_nuc, t_nuc_by_key, q_start_codon, t_start_codon_by_key, ev, gap_open=10, gap_extend=2):
    out = []
    qn = q_nuc.rstrip("\n")
    for line in entry.splitlines():
        f = line.split("\t")
        key = int(f[0])
        tn = t_nuc_by_key[key].rstrip("\n")
        qs, qe, ts, te = int(f[4]), int(f[5]), int(f[7]), int(f[8])
        t_sc = t_start_codon_by_key[key]
        ts, te = ts * 3 + (-3 if t_sc else 0), te * 3 + 2 + (-3 if t_sc else 0)
        qs, qe = qs * 3 + (-3 if q_start_codon else 0), qe * 3 + 2 + (-3 if q_start_codon else 0)
        ids = aln_len = score = 0
        qp, tp, bt = qs, ts, ""
        cig, i = f[10], 0
        while i < len(cig):
            j = i
            while cig[j].isdigit():
                j += 1
            cnt, op = (int(cig[i:j]) if j > i else 0), cig[j]
            i = j + 1
            if op == "M":
                for _ in range(cnt * 3):
                    ids += qn[qp] == tn[tp]
                    score += nucl_score(qn[qp], tn[tp])
                    qp += 1
                    tp += 1
            elif op == "D":
                tp += cnt * 3
                score -= gap_open + ((cnt - 1) * 3) * gap_extend
            elif op == "I":
                qp += cnt * 3
                score -= gap_open + ((cnt - 1) * 3) * gap_extend
            else:
                continue
            aln_len += cnt * 3
            bt += "%d%s" % (cnt * 3, op)
        bits = int(ev.bit_score(score))                      # a double assigned to an int: truncation
        seq_id = _f32(_f32(ids) / _f32(aln_len))
        out.append("%d\t%d\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t%d\t%s\n" % (key, bits, seq_id_text(seq_id), fmt_e(ev.evalue(score, len(qn))), qs, qe, len(qn), ts, te, len(tn), bt))
    return "".join(out)


# ---------------------------------------------------------------------------------------------------------------------
# f2: combineprotnuclaln, besthitperset, mergeresultsbyset, combinescore
# ---------------------------------------------------------------------------------------------------------------------
def combine_prot_nucl(prot_entry, nucl_entry):
    out = []
    nucl_lines = nucl_entry.splitlines()
    for line in prot_entry.splitlines():
        f = line.split("\t")
        target, prot_eval = atou(f[0]), strtod(f[3])
        updated, nucl_id = FLT_MAX, 0.0
        for nl in nucl_lines:
            if atou(nl) != target:
                continue
            g = nl.split("\t")
            nucl_id, nucl_eval = strtod(g[2]), strtod(g[3])
            lp = math.log(prot_eval) if prot_eval > 0 else float("-inf")
            ln = math.log(nucl_eval) if nucl_eval > 0 else float("-inf")
            mean = lp * 0.5 + ln * 0.5
            updated = math.exp(mean) if mean < ln else nucl_eval
        out.append("\t".join(f[:2] + [fmt_e(nucl_id), fmt_e(updated)] + f[4:]) + "\n")
    return "".join(out)


def _group_by_set(entry, member_to_set):
    groups = {}
    for line in entry.split("\n"):
        if line == "":
            continue
        cols = line.split("\t")
        groups.setdefault(member_to_set[atou(cols[0])], []).append(cols)
    return groups


def best_hit_per_set(entry, member_to_set, set_size, simple=True):
    out = []
    groups = _group_by_set(entry, member_to_set)
    for s in sorted(groups):
        lines, n_genes = groups[s], set_size[s]
        best_score = second = -DBL_MAX
        best_eval, best = DBL_MAX, None
        single = simple or len(lines) < 2
        for l in lines:
            ev = strtod(l[3])
            pval = ev / n_genes
            if pval == 0:
                pval = DBL_MIN
            score = -math.log(pval)
            if single:
                if best_eval > ev:
                    best_eval, best = ev, l
            elif score >= best_score:
                second, best_score, best = best_score, score, l
            elif score > second:
                second = score
        if single:
            if best_eval == 0:
                log_p = math.log(DBL_MIN)
            elif 0 < best_eval < 10e-4:
                log_p = math.log(best_eval)
            else:
                x = 1 - math.exp(-best_eval)
                log_p = math.log(x) if x > 0 else float("-inf")
        else:
            log_p = second - best_score
        out.append(("\t".join(fmt_e(log_p) if i == 1 else c for i, c in enumerate(best)) if best is not None else "") + "\n")
    return "".join(out)


def merge_by_set(set_to_member_entry, result_db):
    return "".join(result_db[atou(line)] for line in set_to_member_entry.splitlines())


def combine_score(entry, query_set_size, member_to_set):
    out = []
    log_thr = math.log(1.0 / (query_set_size + 1))
    groups = _group_by_set(entry, member_to_set)
    for s in sorted(groups):
        min_log = sum_log = min_id = sum_id = 0.0
        k = 0
        for l in groups[s]:
            log_p, seq_id = strtod(l[1]), strtod(l[2])
            if log_p < min_log:
                min_log = log_p if log_p == 0 else -log_p
                min_id = seq_id
            if log_p < log_thr:
                sum_log -= log_p - log_thr
                sum_id += seq_id
                k += 1
        if k == 0:
            out.append("%d\t%s\t%s\n" % (s, fmt_e(min_log), fmt_e(min_id)))
        else:
            out.append("%d\t%s\t%s\n" % (s, fmt_e(sum_log - log_thr), fmt_e(sum_id / k)))
    return "".join(out)


# ---------------------------------------------------------------------------------------------------------------------
# f4: filtermatchbyfdr, summarizeresults
# ---------------------------------------------------------------------------------------------------------------------
def _scores(db):
    return [strtod(line.split()[1]) for k in sorted(db) for line in db[k].splitlines()]


def _div(a, b):
    if b == 0:
        return float("nan") if a == 0 else math.copysign(float("inf"), a)
    return a / b


def fdr_threshold(P, N, have_control, cutoff, report):
    P = sorted(P, reverse=True) or [0.0]
    uniq, fdr_list = [], []
    cutoff = struct.unpack("f", struct.pack("f", cutoff))[0]      # a float in the reference (LocalParameters.h:57)
    if not have_control:
        if report:      # FilterMatchbyFdr.cpp:203-211: the running score is declared INSIDE the loop, so every score above DBL_MIN is listed
            for p in P:
                if DBL_MIN < p:
                    uniq.append(p)
                    fdr_list.append(0.0)
        return P[-1], uniq, fdr_list
    N = sorted(N, reverse=True)
    x, y, cur, nc = [], [], DBL_MAX, 0
    for pc, p in enumerate(P):
        if p < cur:
            cur = p
            while nc < len(N) and cur < N[nc]:
                nc += 1
            uniq.append(cur)
            y.append(1.0 * pc / len(P))
            x.append(1.0 * (nc + 0.5) / (len(N) + 1))
    x.append(1.0)
    y.append(1.0)
    slopes, hull, i = [], [], 0
    while i < len(x) - 1:
        j_max, best = i + 1, 0.0
        for j in range(i + 1, len(x)):
            s = _div(1.0 * (y[j] - y[i]), (x[j] - x[i]))
            if s >= best:
                j_max, best = j, s
        i = j_max
        slopes.append(best)
        hull.append(j_max)
    pi0 = (slopes[-2] + slopes[-1]) / 2 if len(slopes) >= 2 else float("nan")
    i, fdr = 0, 0.0
    if _div(x[hull[-1]] * pi0, y[hull[-1]]) >= cutoff:
        while fdr <= cutoff:
            fdr = _div(x[hull[i]] * pi0, y[hull[i]])
            i += 1
    if i < 2:
        threshold = P[-1]
    else:
        j = hull[i - 2]
        tpfp, fp, fdr = y[j], x[j] * pi0, 0.0
        while fdr <= cutoff:
            j += 1
            dx = x[j] - x[j - 1]
            tpfp += dx * slopes[i - 1]
            fp += dx * pi0
            fdr = _div(fp, tpfp)
        threshold = uniq[j]
    if report:
        fdr_list = [0.0] * hull[0]
        for h in range(len(hull) - 1):
            tpfp, fp = y[hull[h]], x[hull[h]] * pi0
            for j in range(hull[h], hull[h + 1]):
                if math.isinf(slopes[h]):
                    fdr_list.append(_div(x[hull[h]] * pi0, y[hull[h]]))
                else:
                    tpfp += (x[j] - x[j - 1]) * slopes[h]
                    fp += (x[j] - x[j - 1]) * pi0
                    fdr_list.append(_div(fp, tpfp))
    return threshold, uniq, fdr_list


def fdr_filter(pos_db, neg_db, cutoff, report):
    threshold, uniq, fdr_list = fdr_threshold(_scores(pos_db), _scores(neg_db), len(neg_db) > 0, cutoff, report)
    out = {}
    for k, entry in pos_db.items():
        kept = []
        for line in entry.splitlines():
            score = strtod(line.split()[1])
            if score >= threshold:
                f = line.split("\t")
                # (a score that is not in the unique-score list -- possible only without a control DB and for scores <= DBL_MIN -- makes the
                # reference read one element past its FDR vector; not pinned, 0 here)
                fdr = fdr_list[uniq.index(score)] if (report and score in uniq and uniq.index(score) < len(fdr_list)) else 0.0
                kept.append("\t".join(f[:3]) + ("\t" + fmt_e(fdr) if report else "") + "\n")
        out[k] = "".join(kept)
    return out, threshold


def summarize(match_entry, aln_entry, fmt, report_fdr=False):
    out = []
    aln_lines = aln_entry.splitlines()
    for m in match_entry.splitlines():
        mf = m.split("\t")
        target_set, head, tmp, lines = atou(mf[0]), "", [], 0
        for al in aln_lines:
            if atou(al) != target_set:
                continue
            e = al.split("\t")
            if lines == 0:
                head = "#%s\t%s\t%s\t" % (e[2], e[3], mf[1]) + (mf[3] + "\t" if report_fdr else "")      # SummarizeResults.cpp:91-116
            lines += 1
            if fmt in (1, 2):
                t = ">" + "\t".join([e[1], e[3], e[4], e[5], e[6], e[8], e[9]] + (e[13:15] if len(e) == 15 else []))
                if fmt == 2:
                    t += "\n" + e[10] + "\n" + e[11]
                tmp.append(t + "\n")
        if lines:
            out.append(head + "%d\n" % lines + "".join(tmp))
    return "".join(out)
