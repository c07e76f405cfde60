/*
 * sp_oracle.c -- TEST INFRASTRUCTURE ONLY (see sp_oracle.h).  Plain C99, scalar, single thread.
 * Compile with -fsigned-char -ffp-contract=off (the reference is built -mavx2 without -mfma).
 *
 * Each function cites the reference lines it restates (paths relative to
 * /root/reference/lib/mmseqs/src/).  No reference code is copied; SIMD routines are restated as
 * lane loops over small arrays so that lane-layout dependent corner cases stay identical.
 */
#include "sp_oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define X_RES 20

uint64_t spo_fnv1a(const void *p, size_t n, uint64_t h) {
    const unsigned char *c = (const unsigned char *) p;
    if (h == 0) h = 1469598103934665603ULL;
    for (size_t i = 0; i < n; i++) { h ^= c[i]; h *= 1099511628211ULL; }
    return h;
}

/* ------------------------------------------------------------------------------------------------
 * a1  SubstitutionMatrix  (commons/SubstitutionMatrix.cpp:13-51,258-299,327-441; BaseMatrix.cpp:96-156)
 * ---------------------------------------------------------------------------------------------- */
/* Util::getWordsOfLine (commons/Util.h:232-246): words are separated by ' ' or '\t' */
static size_t words_of_line(const char *data, const char **words, size_t maxElement) {
    size_t n = 0;
    while (*data != '\n' && *data != '\0') {
        while (*data == ' ' || *data == '\t') data++;
        words[n++] = data;
        if (n >= maxElement) return n;
        while (!(*data == ' ' || *data == '\t' || *data == '\n' || *data == '\0')) data++;
    }
    if (n < maxElement) words[n] = data;
    return n;
}

int spo_matrix_init(spo_matrix *m, const char *text, double bitFactor, double scoreBias) {
    memset(m, 0, sizeof(*m));
    for (int i = 0; i < 256; i++) m->aa2num[i] = 255;        /* BaseMatrix::BaseMatrix, BaseMatrix.cpp:12-20 */
    const char *words[256];
    /* setAaMappingDetectAlphSize, SubstitutionMatrix.cpp:409-441 */
    int alph = -1;
    for (const char *line = text; *line;) {
        const char *eol = strchr(line, '\n');
        size_t wc = words_of_line(line, words, 256);
        if (line[0] != '#' && wc > 1) {
            for (size_t i = 0; i < wc; i++) {
                if (!isalpha((unsigned char) words[i][0])) return -1;
                int aa = toupper((unsigned char) words[i][0]);
                m->aa2num[aa] = (unsigned char) i;
                if (i < SPO_ALPH) m->num2aa[i] = (char) aa;
            }
            alph = (int) wc;
            break;
        }
        if (!eol) break;
        line = eol + 1;
    }
    if (alph != SPO_ALPH) return -2;                          /* this restatement covers the 21-letter case */
    m->alphabetSize = alph;
    const int xIdx = m->aa2num['X'];
    /* readProbMatrix, SubstitutionMatrix.cpp:327-407 */
    double raw[SPO_ALPH][SPO_ALPH];
    memset(raw, 0, sizeof(raw));
    int started = 0, hasLambda = 0, hasBg = 0;
    for (const char *line = text; *line;) {
        const char *eol = strchr(line, '\n');
        size_t wc = words_of_line(line, words, 256);
        if (line[0] == '#') {
            if (strncmp(line, "# Background (precomputed optional):", 36) == 0) {
                for (size_t i = 4; i < wc && i - 4 < SPO_ALPH; i++) m->pBack[i - 4] = strtod(words[i], NULL);
                hasBg = 1;
            }
            if (strncmp(line, "# Lambda     (precomputed optional):", 36) == 0) {
                m->lambda = strtod(words[4], NULL);
                hasLambda = 1;
            }
        } else if (wc > 1 && !started) {
            started = 1;
        } else if (wc > 1) {
            if (!isalpha((unsigned char) words[0][0])) return -3;
            int aa = m->aa2num[toupper((unsigned char) words[0][0])];
            for (int i = 0; i < alph; i++) raw[aa][i] = strtod(words[i + 1], NULL);
        }
        if (!eol) break;
        line = eol + 1;
    }
    if (!hasLambda || !hasBg) return -4;   /* estimateLambdaAndBackground (lib/cacode) is not restated: VTML40/80 carry both */
    int xIsPositive = 0;
    for (int j = 0; j < alph; j++) if (raw[xIdx][j] > 0 || raw[j][xIdx] > 0) { xIsPositive = 1; break; }
    if (!xIsPositive) for (int i = 0; i < alph - 1; i++) m->pBack[i] = m->pBack[i] * (1.0 - m->pBack[xIdx]);
    for (int i = 0; i < alph; i++) for (int j = 0; j < alph; j++)
        m->prob[i][j] = exp(m->lambda * raw[i][j]) * m->pBack[i] * m->pBack[j];
    /* setupLetterMapping, SubstitutionMatrix.cpp:258-299 */
    for (int letter = 0; letter < 255; letter++) {
        int up = (letter < 128) ? toupper(letter) : letter;
        switch (up) {
            case 'A': case 'T': case 'G': case 'C': case 'D': case 'E': case 'F': case 'H': case 'I': case 'K':
            case 'L': case 'M': case 'N': case 'P': case 'Q': case 'R': case 'S': case 'V': case 'W': case 'Y': case 'X':
                m->aa2num[letter] = m->aa2num[up]; break;
            case 'J': m->aa2num[letter] = m->aa2num['L']; break;
            case 'U': case 'O': m->aa2num[letter] = m->aa2num['X']; break;
            case 'Z': m->aa2num[letter] = m->aa2num['E']; break;
            case 'B': m->aa2num[letter] = m->aa2num['D']; break;
            default: m->aa2num[letter] = m->aa2num['X']; break;
        }
    }
    m->aa2num[255] = m->aa2num['X'];
    /* generateSubMatrix, BaseMatrix.cpp:96-156 (local background = row sums, X := ANY_BACK) */
    double pb[SPO_ALPH];
    for (int i = 0; i < alph; i++) { pb[i] = 0; for (int j = 0; j < alph; j++) pb[i] += m->prob[i][j]; }
    pb[alph - 1] = 1E-5;
    for (int i = 0; i < alph; i++) for (int j = 0; j < alph; j++) {
        double sm = log2(m->prob[i][j] / (pb[i] * pb[j]));
        double v = (bitFactor * sm + scoreBias);
        m->sub[i][j] = (short) ((v < 0.0) ? v - 0.5 : v + 0.5);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * a2  ExtendedSubstitutionMatrix::calcScoreMatrix (prefiltering/ExtendedSubstitutionMatrix.cpp:20-69)
 *     partner k-mers enumerated with the FIRST letter slowest (createCartesianProduct, :105-126),
 *     k-mer index = sum letter[p]*20^p (Indexer.h), std::stable_sort by score descending.
 * ---------------------------------------------------------------------------------------------- */
static void ext_generic(const spo_matrix *m, int k, short *score, uint16_t *index) {
    const int A = 20;
    int size = 1; for (int i = 0; i < k; i++) size *= A;
    /* enumeration order e -> letters (first letter slowest) and k-mer index */
    int *idxOf = (int *) malloc(sizeof(int) * size);
    unsigned char *let = (unsigned char *) malloc((size_t) size * k);
    for (int e = 0; e < size; e++) {
        int r = e, id = 0, pw = 1;
        for (int p = k - 1; p >= 0; p--) { let[e * k + p] = (unsigned char) (r % A); r /= A; }
        for (int p = 0; p < k; p++) { id += let[e * k + p] * pw; pw *= A; }
        idxOf[e] = id;
    }
    short *tmp = (short *) malloc(sizeof(short) * size);
    int cnt[1024];
    for (int ei = 0; ei < size; ei++) {
        const unsigned char *li = let + ei * k;
        int lo = 32767, hi = -32768;
        for (int ej = 0; ej < size; ej++) {
            const unsigned char *lj = let + ej * k;
            int s = 0; for (int p = 0; p < k; p++) s += m->sub[li[p]][lj[p]];
            tmp[ej] = (short) s; if (s < lo) lo = s; if (s > hi) hi = s;
        }
        /* stable counting sort, descending score, ties keep enumeration order */
        int range = hi - lo + 1;
        memset(cnt, 0, sizeof(int) * (range + 1));
        for (int ej = 0; ej < size; ej++) cnt[hi - tmp[ej] + 1]++;
        for (int r = 0; r < range; r++) cnt[r + 1] += cnt[r];
        short *srow = score + (size_t) idxOf[ei] * size;
        uint16_t *irow = index + (size_t) idxOf[ei] * size;
        for (int ej = 0; ej < size; ej++) {
            int pos = cnt[hi - tmp[ej]]++;
            srow[pos] = tmp[ej]; irow[pos] = (uint16_t) idxOf[ej];
        }
    }
    free(tmp); free(let); free(idxOf);
}
void spo_ext3(const spo_matrix *m, short *score, uint16_t *index) { ext_generic(m, 3, score, index); }
void spo_ext2(const spo_matrix *m, short *score, uint16_t *index) { ext_generic(m, 2, score, index); }

/* ------------------------------------------------------------------------------------------------
 * a3  Sequence::mapSequence (commons/Sequence.cpp:476-489)
 * ---------------------------------------------------------------------------------------------- */
int spo_map_sequence(const spo_matrix *m, const char *seq, int maxLen, unsigned char *out) {
    int l = 0;
    while (l < maxLen && seq[l] != '\0' && seq[l] != '\n') { out[l] = m->aa2num[(unsigned char) seq[l]]; l++; }
    return l;
}

/* ------------------------------------------------------------------------------------------------
 * a4  SubstitutionMatrix::calcLocalAaBiasCorrection (commons/SubstitutionMatrix.cpp:85-114)
 * ---------------------------------------------------------------------------------------------- */
void spo_local_bias(const spo_matrix *m, const unsigned char *seq, int N, float *bias) {
    const int windowSize = 40;
    for (int i = 0; i < N; i++) {
        const int minPos = (i - windowSize / 2) > 0 ? (i - windowSize / 2) : 0;
        const int maxPos = (i + windowSize / 2) < N ? (i + windowSize / 2) : N;
        const int windowLength = maxPos - minPos;
        int sum = 0;
        const short *row = m->sub[seq[i]];
        for (int j = minPos; j < maxPos; j++) sum += row[seq[j]];
        sum -= row[seq[i]];
        float d = (float) sum;
        d = (float) ((double) d / (-1.0 * (double) ((float) windowLength)));
        for (int a = 0; a < m->alphabetSize; a++) d = (float) ((double) d + m->pBack[a] * (double) ((float) row[a]));
        bias[i] = d;
    }
}

/* ------------------------------------------------------------------------------------------------
 * a5  tantan::maskSequences, gap-less branch (commons/tantan.cpp:73-125,303-345,369-398,435-488)
 *     called with (50, 0.005, 0.05, 0.9, 0, 0, 0.9) from IndexBuilder.cpp:139-148;
 *     likelihood ratios = prob[i][j]/(pBack[i]*pBack[j]) (BaseMatrix.h:80-109)
 * ---------------------------------------------------------------------------------------------- */
static double first_repeat_offset_prob(double probMult, int maxRepeatOffset) {
    volatile double pm = probMult;   /* keep pow() a run-time libm call like the reference's */
    if (pm < 1 || pm > 1) return (1 - pm) / (1 - pow(pm, (double) maxRepeatOffset));
    return 1.0 / maxRepeatOffset;
}

int spo_tantan_mask(const spo_matrix *m, unsigned char *seq, int L) {
    enum { MAXOFF = 50, STEP = 16 };
    const double repeatProb = 0.005, repeatEndProb = 0.05, decay = 0.9, minMaskProb = 0.9;
    double lr[SPO_ALPH][SPO_ALPH];
    for (int i = 0; i < SPO_ALPH; i++) for (int j = 0; j < SPO_ALPH; j++) lr[i][j] = m->prob[i][j] / (m->pBack[i] * m->pBack[j]);
    const double b2b = 1 - repeatProb, f2b = repeatEndProb, f2f0 = 1 - repeatEndProb;
    const double b2fDecay = decay, b2fGrowth = 1 / decay;
    const double b2fFirst = repeatProb * first_repeat_offset_prob(b2fDecay, MAXOFF);
    (void) b2fGrowth;
    double b2f[MAXOFF], fg[MAXOFF];
    { double p = b2fFirst; for (int i = 0; i < MAXOFF; i++) { b2f[i] = p; p *= b2fDecay; } }
    if (L <= 0) return 0;
    float *probs = (float *) malloc(sizeof(float) * (size_t) L);
    double *scale = (double *) calloc((size_t) L / STEP + 1, sizeof(double));
    double bg = 1.0;
    for (int i = 0; i < MAXOFF; i++) fg[i] = 0.0;
    for (int pos = 0; pos < L; pos++) {                       /* calcForwardTransitionAndEmissionProbs */
        double b = bg, fromFg = 0;
        const double *row = lr[seq[pos]];
        int maxOffset = pos < MAXOFF ? pos : MAXOFF;
        for (int i = 0; i < maxOffset; i++) {
            double f = fg[i];
            fromFg += f;
            fg[i] = (b * b2f[i] + f * f2f0) * row[seq[pos - i - 1]];
        }
        bg = b * b2b + fromFg * f2b;
        if (pos % STEP == STEP - 1) {                         /* rescaleForward */
            double s = 1 / bg;
            scale[pos / STEP] = s;
            bg *= s;
            for (int i = 0; i < MAXOFF; i++) fg[i] *= s;
        }
        probs[pos] = (float) bg;
    }
    double z; { double acc = 0.0; for (int i = 0; i < MAXOFF; i++) acc += fg[i]; z = bg * b2b + acc * f2b; }
    bg = b2b;
    for (int i = 0; i < MAXOFF; i++) fg[i] = f2b;
    for (int pos = L - 1; pos >= 0; pos--) {
        double nonRepeat = (double) probs[pos] * bg / z;
        probs[pos] = 1 - (float) nonRepeat;
        if (pos % STEP == STEP - 1) { double s = scale[pos / STEP]; bg *= s; for (int i = 0; i < MAXOFF; i++) fg[i] *= s; }
        double toBg = f2b * bg, toFg = 0;                     /* calcEmissionAndBackwardTransitionProbs */
        const double *row = lr[seq[pos]];
        int maxOffset = pos < MAXOFF ? pos : MAXOFF;
        for (int i = 0; i < maxOffset; i++) {
            double f = fg[i] * row[seq[pos - i - 1]];
            toFg += b2f[i] * f;
            fg[i] = toBg + f2f0 * f;
        }
        bg = b2b * bg + toFg;
    }
    int masked = 0;
    for (int pos = 0; pos < L; pos++) if ((double) probs[pos] >= minMaskProb) { seq[pos] = X_RES; masked++; }
    free(scale); free(probs);
    return masked;
}

/* ------------------------------------------------------------------------------------------------
 * context
 * ---------------------------------------------------------------------------------------------- */
struct spo_ctx {
    spo_matrix seed, ungapped, aln;
    int kmerThr;
    int maxSeqs;
    unsigned binSize;
    uint64_t l2;
    short *e3score; uint16_t *e3index;
    /* targets */
    size_t n;
    uint64_t *toff;            /* n+1 */
    unsigned char *masked;     /* numeric masked residues */
    unsigned char *plainNum;   /* numeric unmasked residues (for align) */
    uint64_t maskedCount;
    /* index */
    uint32_t *ioff;            /* 20^6+1 */
    uint32_t *iseq; uint16_t *ipos;
    uint64_t nEntries;
    /* evalue */
    spo_alp alp[2];            /* [0]=gapless [1]=gapped */
    int haveAlp;
};

static const int PATTERN_POS[SPO_KMER] = {0, 1, 3, 4, 5, 7};    /* 11011101, Sequence.cpp:199-234 */
#define TABLE_SIZE 64000000u

spo_ctx *spo_create(const char *seedText, const char *scoreText, float sens, int maxSeqs, unsigned binSize, uint64_t l2) {
    spo_ctx *c = (spo_ctx *) calloc(1, sizeof(spo_ctx));
    /* Prefiltering.cpp:63-67 and Alignment.cpp:159: bit factors 8/2/2, score bias -0.2f/-0.2f/0.0f */
    if (spo_matrix_init(&c->seed, seedText, (double) 8.0f, (double) -0.2f) ||
        spo_matrix_init(&c->ungapped, scoreText, (double) 2.0f, (double) -0.2f) ||
        spo_matrix_init(&c->aln, scoreText, (double) 2.0f, (double) 0.0f)) { free(c); return NULL; }
    /* Prefiltering::getKmerThreshold, Prefiltering.cpp:987-1021 (k=6, sequence search) */
    { float base = 163.2; double best = base - (sens * 8.917); c->kmerThr = (int) best; }
    c->maxSeqs = maxSeqs;
    c->binSize = binSize;
    c->l2 = l2 ? l2 : 262144;
    c->e3score = (short *) malloc(sizeof(short) * 8000u * 8000u);
    c->e3index = (uint16_t *) malloc(sizeof(uint16_t) * 8000u * 8000u);
    spo_ext3(&c->seed, c->e3score, c->e3index);
    return c;
}

void spo_destroy(spo_ctx *c) {
    if (!c) return;
    free(c->e3score); free(c->e3index); free(c->toff); free(c->masked); free(c->plainNum);
    free(c->ioff); free(c->iseq); free(c->ipos); free(c);
}
int spo_kmer_thr(const spo_ctx *c) { return c->kmerThr; }
const spo_matrix *spo_seed_matrix(const spo_ctx *c) { return &c->seed; }
const spo_matrix *spo_ungapped_matrix(const spo_ctx *c) { return &c->ungapped; }
const spo_matrix *spo_aln_matrix(const spo_ctx *c) { return &c->aln; }
const short *spo_ext3_score(const spo_ctx *c) { return c->e3score; }
const uint16_t *spo_ext3_index(const spo_ctx *c) { return c->e3index; }
size_t spo_num_targets(const spo_ctx *c) { return c->n; }
uint64_t spo_aa_db_size(const spo_ctx *c) { return c->toff ? c->toff[c->n] : 0; }
const unsigned char *spo_masked_residues(const spo_ctx *c) { return c->masked; }
const uint64_t *spo_target_offsets(const spo_ctx *c) { return c->toff; }
uint64_t spo_index_entries(const spo_ctx *c) { return c->nEntries; }
const uint32_t *spo_index_offsets(const spo_ctx *c) { return c->ioff; }
const uint32_t *spo_index_seqids(const spo_ctx *c) { return c->iseq; }
const uint16_t *spo_index_positions(const spo_ctx *c) { return c->ipos; }
uint64_t spo_masked_count(const spo_ctx *c) { return c->maskedCount; }

/* QueryMatcher::initDiagonalMatcher (prefiltering/QueryMatcher.cpp:383-411) */
unsigned spo_bin_size(const spo_ctx *c) {
    if (c->binSize) return c->binSize;
    size_t db = c->n;
    for (unsigned b = 2; b <= 1024; b *= 2) if (db / b < c->l2) return b;
    return 2048;
}

/* ------------------------------------------------------------------------------------------------
 * a6  IndexBuilder::fillDatabase + IndexTable (prefiltering/IndexBuilder.cpp:53-256,
 *     IndexTable.h:130-170,202-226,341-394,179-186)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint32_t kmer; uint16_t pos; } kp_t;
static int kp_cmp(const void *a, const void *b) {
    const kp_t *x = (const kp_t *) a, *y = (const kp_t *) b;
    if (x->kmer != y->kmer) return x->kmer < y->kmer ? -1 : 1;
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    return 0;
}

/* distinct indexable k-mers of one masked target, each at its smallest position, sorted by k-mer */
static size_t target_kmers(const spo_ctx *c, const unsigned char *s, int L, kp_t *buf) {
    size_t n = 0;
    for (int i = 0; i + SPO_SPAN <= L; i++) {
        int hasX = 0, self = 0; uint32_t idx = 0, pw = 1;
        for (int p = 0; p < SPO_KMER; p++) {
            unsigned char r = s[i + PATTERN_POS[p]];
            if (r == X_RES) hasX = 1;
            self += (char) c->seed.sub[r][r];          /* getScoreLookup, IndexBuilder.cpp:8-19 */
            idx += r * pw; pw *= 20;
        }
        if (hasX) continue;                               /* IndexTable.h:138-140 */
        if (c->kmerThr > 0 && self < c->kmerThr) continue; /* IndexTable.h:141-149 */
        buf[n].kmer = idx; buf[n].pos = (uint16_t) i; n++;
    }
    if (n > 1) qsort(buf, n, sizeof(kp_t), kp_cmp);
    size_t w = 0;
    for (size_t k = 0; k < n; k++) if (k == 0 || buf[k].kmer != buf[k - 1].kmer) buf[w++] = buf[k];
    return w;
}

int spo_set_targets(spo_ctx *c, const char *residues, const uint64_t *offsets, size_t n, int mask) {
    free(c->toff); free(c->masked); free(c->plainNum); free(c->ioff); free(c->iseq); free(c->ipos);
    c->n = n;
    c->toff = (uint64_t *) malloc(sizeof(uint64_t) * (n + 1));
    memcpy(c->toff, offsets, sizeof(uint64_t) * (n + 1));
    uint64_t R = offsets[n];
    c->masked = (unsigned char *) malloc(R + 1);
    c->plainNum = (unsigned char *) malloc(R + 1);
    c->maskedCount = 0;
    int maxL = 0;
    for (size_t t = 0; t < n; t++) {
        int L = (int) (offsets[t + 1] - offsets[t]);
        if (L > maxL) maxL = L;
        for (int i = 0; i < L; i++) c->plainNum[offsets[t] + i] = c->seed.aa2num[(unsigned char) residues[offsets[t] + i]];
        memcpy(c->masked + offsets[t], c->plainNum + offsets[t], (size_t) L);
        if (mask) c->maskedCount += (uint64_t) spo_tantan_mask(&c->seed, c->masked + offsets[t], L);
    }
    c->ioff = (uint32_t *) calloc((size_t) TABLE_SIZE + 1, sizeof(uint32_t));
    kp_t *buf = (kp_t *) malloc(sizeof(kp_t) * (size_t) (maxL + 1));
    /* pass 1: count (addKmerCount) */
    for (size_t t = 0; t < n; t++) {
        size_t k = target_kmers(c, c->masked + offsets[t], (int) (offsets[t + 1] - offsets[t]), buf);
        for (size_t j = 0; j < k; j++) c->ioff[buf[j].kmer + 1]++;
    }
    for (size_t k = 0; k < TABLE_SIZE; k++) c->ioff[k + 1] += c->ioff[k];   /* IndexTable::init */
    c->nEntries = c->ioff[TABLE_SIZE];
    c->iseq = (uint32_t *) malloc(sizeof(uint32_t) * (c->nEntries + 1));
    c->ipos = (uint16_t *) malloc(sizeof(uint16_t) * (c->nEntries + 1));
    uint32_t *cursor = (uint32_t *) malloc(sizeof(uint32_t) * (size_t) TABLE_SIZE);
    memcpy(cursor, c->ioff, sizeof(uint32_t) * (size_t) TABLE_SIZE);
    /* pass 2: fill in target order => every list is already sorted by (seqId, pos) (sortDBSeqLists) */
    for (size_t t = 0; t < n; t++) {
        size_t k = target_kmers(c, c->masked + offsets[t], (int) (offsets[t + 1] - offsets[t]), buf);
        for (size_t j = 0; j < k; j++) { uint32_t at = cursor[buf[j].kmer]++; c->iseq[at] = (uint32_t) t; c->ipos[at] = buf[j].pos; }
    }
    free(cursor); free(buf);
    return 0;
}

uint64_t spo_index_digest(const spo_ctx *c) {
    uint64_t h = 0;
    for (size_t k = 0; k < TABLE_SIZE; k++)
        for (uint32_t e = c->ioff[k]; e < c->ioff[k + 1]; e++) {
            uint32_t rec[3] = {(uint32_t) k, c->iseq[e], c->ipos[e]};
            h = spo_fnv1a(rec, sizeof(rec), h);
        }
    return h;
}

/* ------------------------------------------------------------------------------------------------
 * a7  KmerGenerator::generateKmerList, k=6 => two 3-mer steps (prefiltering/KmerGenerator.cpp:107-216)
 * ---------------------------------------------------------------------------------------------- */
#define MAX_KMER_RESULT_SIZE (262144u * 32u)   /* KmerGenerator.h:45 */
size_t spo_kmer_list(const spo_ctx *c, const unsigned char *kmer, int threshold, uint32_t *out, size_t cap) {
    const uint32_t ia = kmer[0] + 20u * kmer[1] + 400u * kmer[2];
    const uint32_t ib = kmer[3] + 20u * kmer[4] + 400u * kmer[5];
    const short *sa = c->e3score + (size_t) ia * 8000, *sb = c->e3score + (size_t) ib * 8000;
    const uint16_t *xa = c->e3index + (size_t) ia * 8000, *xb = c->e3index + (size_t) ib * 8000;
    const short thr = (short) threshold;
    const short possibleRest0 = sb[0];                        /* highestScorePerArray[1] + 0 */
    const short cutoff1 = (short) (thr - possibleRest0);
    size_t counter = 0;
    for (size_t i = 0; i < 8000; i++) {                       /* calculateArrayProduct */
        const short si = sa[i];
        if (si < cutoff1) break;
        const short cutoff2 = (short) (thr - si - 0);
        for (size_t j = 0; j < 8000 && (counter + 1 < MAX_KMER_RESULT_SIZE) && sb[j] >= cutoff2; j++) {
            if (counter < cap) out[counter] = (uint32_t) xa[i] + (uint32_t) xb[j] * 8000u;
            counter++;
        }
        if (counter + 1 >= MAX_KMER_RESULT_SIZE) return counter;
    }
    return counter;
}

/* ------------------------------------------------------------------------------------------------
 * a8  per-window thresholds (prefiltering/QueryMatcher.cpp:200-219)
 * ---------------------------------------------------------------------------------------------- */
int spo_window_thresholds(const spo_ctx *c, const unsigned char *q, int L, const float *bias, int *thr) {
    int nw = 0;
    for (int i = 0; i + SPO_SPAN <= L; i++, nw++) {
        float bc = 0; int hasX = 0;
        for (int p = 0; p < SPO_KMER; p++) { bc += bias[i + PATTERN_POS[p]]; if (q[i + PATTERN_POS[p]] == X_RES) hasX = 1; }
        if (hasX) { thr[i] = -1; continue; }
        short b = (short) ((bc < 0.0) ? (double) bc - 0.5 : (double) bc + 0.5);
        int t = c->kmerThr - b; if (t < 0) t = 0;
        thr[i] = (short) t;
    }
    return nw;
}

/* ------------------------------------------------------------------------------------------------
 * a8+a9  QueryMatcher::match + CacheFriendlyOperations::findDuplicates, diagonal-scoring mode
 *        (prefiltering/QueryMatcher.cpp:188-307, CacheFriendlyOperations.cpp:38-49,146-235,298-308)
 *
 *  NOTE (pinned by the golden vectors): findDuplicates scans windows indexFrom <= i < indexTo,
 *  where indexTo is the LAST window's position, so the hits of the last window never take part
 *  (CacheFriendlyOperations.cpp:43 together with QueryMatcher.cpp:285-291).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint32_t id; uint16_t diag; } cr_t;

static size_t match_impl(const spo_ctx *c, const unsigned char *q, int L, const float *bias, cr_t **outp,
                         uint64_t *dbMatches, uint64_t *kmerListLen) {
    int *thr = (int *) malloc(sizeof(int) * (size_t) (L + 1));
    int nw = spo_window_thresholds(c, q, L, bias, thr);
    size_t cap = 1024, n = 0;
    cr_t *stream = (cr_t *) malloc(sizeof(cr_t) * cap);
    size_t *winStart = (size_t *) malloc(sizeof(size_t) * (size_t) (nw + 2));
    uint32_t *klist = NULL; size_t kcap = 0;
    uint64_t klen = 0;
    int indexTo = 0;
    for (int i = 0; i < nw; i++) {
        winStart[i] = n;
        indexTo = i;
        if (thr[i] < 0) continue;
        unsigned char kmer[SPO_KMER];
        for (int p = 0; p < SPO_KMER; p++) kmer[p] = q[i + PATTERN_POS[p]];
        size_t k = spo_kmer_list(c, kmer, thr[i], klist, kcap);
        if (k > kcap) { kcap = k; klist = (uint32_t *) realloc(klist, sizeof(uint32_t) * kcap); k = spo_kmer_list(c, kmer, thr[i], klist, kcap); }
        klen += k;
        for (size_t kk = 0; kk < k; kk++) {
            uint32_t from = c->ioff[klist[kk]], to = c->ioff[klist[kk] + 1];
            if (n + (to - from) > cap) { while (n + (to - from) > cap) cap *= 2; stream = (cr_t *) realloc(stream, sizeof(cr_t) * cap); }
            for (uint32_t e = from; e < to; e++) { stream[n].id = c->iseq[e]; stream[n].diag = (uint16_t) (i - c->ipos[e]); n++; }
        }
    }
    winStart[nw] = n;
    if (dbMatches) *dbMatches = n;
    if (kmerListLen) *kmerListLen = klen;
    /* only windows [0, indexTo) are scanned */
    size_t nScan = (nw > 0) ? winStart[indexTo] : 0;
    const unsigned B = spo_bin_size(c);
    unsigned shift = 0; while ((1u << shift) < B) shift++;
    size_t dupSize = (c->n >> shift) + 2;
    unsigned char *prev = (unsigned char *) calloc(dupSize, 1);
    cr_t *res = (cr_t *) malloc(sizeof(cr_t) * (nScan + 1));
    cr_t *tmp = (cr_t *) malloc(sizeof(cr_t) * (nScan + 1));
    size_t nres = 0;
    for (unsigned bin = 0; bin < B; bin++) {
        size_t nt = 0;
        for (size_t e = 0; e < nScan; e++) {                  /* hashIndexEntry keeps stream order inside a bin */
            if ((stream[e].id & (B - 1)) != bin) continue;
            const uint32_t h = stream[e].id >> shift;
            const unsigned char cur = (unsigned char) stream[e].diag;
            if (cur == prev[h]) tmp[nt++] = stream[e];         /* CacheFriendlyOperations.cpp:161-168 */
            prev[h] = cur;
        }
        for (size_t k = nt; k-- > 0;) prev[tmp[k].id >> shift] = (unsigned char) (((unsigned char) tmp[k].diag) + 1);   /* :200-205 */
        for (size_t k = 0; k < nt; k++) {                     /* :208-222 */
            const uint32_t h = tmp[k].id >> shift;
            res[nres] = tmp[k];
            nres += (prev[h] != (unsigned char) tmp[k].diag) ? 1 : 0;
            prev[h] = (unsigned char) tmp[k].diag;
        }
        for (size_t e = 0; e < nScan; e++) if ((stream[e].id & (B - 1)) == bin) prev[stream[e].id >> shift] = 0;        /* :224-232 */
    }
    free(tmp); free(prev); free(klist); free(winStart); free(stream); free(thr);
    *outp = res;
    return nres;
}

size_t spo_match(const spo_ctx *c, const char *query, int qLen, uint32_t *ids, uint16_t *diags, size_t cap,
                 uint64_t *dbMatches, uint64_t *kmerListLen) {
    unsigned char *q = (unsigned char *) malloc((size_t) qLen + 1);
    int L = spo_map_sequence(&c->seed, query, qLen, q);
    float *bias = (float *) malloc(sizeof(float) * (size_t) (L + 1));
    spo_local_bias(&c->seed, q, L, bias);
    cr_t *res; size_t n = match_impl(c, q, L, bias, &res, dbMatches, kmerListLen);
    for (size_t i = 0; i < n && i < cap; i++) { ids[i] = res[i].id; diags[i] = res[i].diag; }
    free(res); free(bias); free(q);
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * a10  UngappedAlignment (prefiltering/UngappedAlignment.cpp:295-355 createProfile, :37-51 scalar
 *      scoring, :216-233 long diagonals, :368-382 geometry).  The AVX2 u8-saturating path
 *      (:67-121) equals min(255-bias, exact) -- see DESIGN.md -- so one scalar routine covers both.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int L; int bias; signed char *prof; /* L x 32, signed char like the reference's `char` */ } uprof_t;

static void uprof_build(const spo_ctx *c, const unsigned char *q, int L, const float *biasCorr, uprof_t *p) {
    int bias = 0, aaBias = 0;
    for (int i = 0; i < SPO_ALPH; i++) for (int j = 0; j < SPO_ALPH; j++) if (c->ungapped.sub[i][j] < bias) bias = c->ungapped.sub[i][j];
    signed char *corr = (signed char *) malloc((size_t) L + 1);
    for (int pos = 0; pos < L; pos++) {
        float a = biasCorr[pos];
        a = (float) ((a < 0.0) ? (double) (a / 4) - 0.5 : (double) (a / 4) + 0.5);
        corr[pos] = (signed char) a;
        aaBias = (corr[pos] < aaBias) ? corr[pos] : aaBias;
    }
    if (aaBias > 0) aaBias = 0;
    bias = abs(bias) + abs(aaBias);
    p->L = L; p->bias = bias;
    p->prof = (signed char *) malloc((size_t) L * 32 + 32);
    memset(p->prof, bias, (size_t) L * 32 + 32);
    for (int pos = 0; pos < L; pos++) for (int i = 0; i < SPO_ALPH; i++)
        p->prof[pos * 32 + i] = (signed char) (c->ungapped.sub[q[pos]][i] + corr[pos] + bias);
    free(corr);
}

static int scalar_diag(const signed char *prof, int bias, unsigned len, const unsigned char *db) {
    int max = 0, score = 0;
    for (unsigned pos = 0; pos < len; pos++) {
        int curr = prof[pos * 32 + db[pos]];
        score = (curr - bias) + score;
        score = (score < 0) ? 0 : score;
        max = (score > max) ? score : max;
    }
    return max;
}

static int single_seq_score(const uprof_t *p, const unsigned char *db, unsigned dbLen, int diagonal, unsigned minDist) {
    int max = 0;
    if (diagonal >= 0 && minDist < (unsigned) p->L) {
        unsigned len = dbLen < (unsigned) p->L - minDist ? dbLen : (unsigned) p->L - minDist;
        int s = scalar_diag(p->prof + (size_t) minDist * 32, p->bias, len, db); if (s > max) max = s;
    } else if (diagonal < 0 && minDist < dbLen) {
        unsigned len = (dbLen - minDist) < (unsigned) p->L ? (dbLen - minDist) : (unsigned) p->L;
        int s = scalar_diag(p->prof, p->bias, len, db + minDist); if (s > max) max = s;
    }
    return max;
}

static int long_score(const uprof_t *p, const unsigned char *db, unsigned dbLen, uint16_t diagonal) {
    int total = 0;
    for (unsigned d = 1; d <= 1 + dbLen / 32768; d++) {
        int real = (int) (-(int64_t) d * 65536 + diagonal);
        int m = single_seq_score(p, db, dbLen, real, (unsigned) abs(real)); if (m > total) total = m;
    }
    for (unsigned d = 0; d <= (unsigned) p->L / 65536; d++) {
        int real = (int) (d * 65536 + diagonal);
        int m = single_seq_score(p, db, dbLen, real, (unsigned) abs(real)); if (m > total) total = m;
    }
    return total;
}

/* exact (unclamped) score: UngappedAlignment::scoreSingelSequenceByCounterResult (:385-399) */
static int exact_score(const spo_ctx *c, const uprof_t *p, uint32_t id, uint16_t diag) {
    const unsigned char *db = c->masked + c->toff[id];
    unsigned dbLen = (unsigned) (c->toff[id + 1] - c->toff[id]);
    uint16_t minDist = (uint16_t) ((uint16_t) (0 - diag) < diag ? (uint16_t) (0 - diag) : diag);
    if (p->L >= 32768 || dbLen >= 32768) return long_score(p, db, dbLen, diag);
    return single_seq_score(p, db, dbLen, (int) (short) diag, minDist);
}

/* the u8 `count` written by scoreDiagonalAndUpdateHits (:140-214) */
static unsigned char clamped_score(const spo_ctx *c, const uprof_t *p, uint32_t id, uint16_t diag) {
    int m = exact_score(c, p, id, diag);
    int lim = (p->L >= 32768) ? 255 : 255 - p->bias;
    return (unsigned char) (m < lim ? m : lim);
}

int spo_ungapped_score(const spo_ctx *c, const char *query, int qLen, uint32_t seqId, uint16_t diag, int *biasOut) {
    unsigned char *q = (unsigned char *) malloc((size_t) qLen + 1);
    int L = spo_map_sequence(&c->seed, query, qLen, q);
    float *bias = (float *) malloc(sizeof(float) * (size_t) (L + 1));
    spo_local_bias(&c->seed, q, L, bias);
    uprof_t p; uprof_build(c, q, L, bias, &p);
    int s = clamped_score(c, &p, seqId, diag);
    if (biasOut) *biasOut = p.bias;
    free(p.prof); free(bias); free(q);
    return s;
}

/* ------------------------------------------------------------------------------------------------
 * a11  QueryMatcher::matchQuery tail (prefiltering/QueryMatcher.cpp:100-186,318-381,459-505,
 *      QueryMatcher.h:199-209, CacheFriendlyOperations.cpp:70-76,283-295,311-341)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint32_t id; uint16_t diag; unsigned char count; } crs_t;

static size_t radix_by_score(const unsigned *sizes, crs_t *writePos, unsigned thr, const crs_t *results, size_t n) {
    crs_t *ptr[256]; crs_t *prev = writePos + n;
    for (unsigned i = 0; i < 256; i++) { ptr[i] = prev - sizes[i]; prev = ptr[i]; }
    size_t above = 0;
    for (size_t i = 0; i < n; i++) { unsigned s = results[i].count; if (s >= thr) { above++; *ptr[s] = results[i]; ptr[s]++; } }
    return above;
}

/* straightforward restatement of keepMaxElement: output order = bin asc, then input order */
static size_t keep_max_elements(const spo_ctx *c, const crs_t *in, size_t n, crs_t *outArr) {
    const unsigned B = spo_bin_size(c);
    unsigned shift = 0; while ((1u << shift) < B) shift++;
    unsigned char *best = (unsigned char *) calloc((c->n >> shift) + 2, 1);
    size_t out = 0;
    for (unsigned bin = 0; bin < B; bin++) {
        for (size_t i = 0; i < n; i++) if ((in[i].id & (B - 1)) == bin) { uint32_t h = in[i].id >> shift; if (in[i].count > best[h]) best[h] = in[i].count; }
        for (size_t i = 0; i < n; i++) if ((in[i].id & (B - 1)) == bin) {
            uint32_t h = in[i].id >> shift;
            int found = (best[h] == in[i].count);
            if (found) outArr[out++] = in[i];
            best[h] = (unsigned char) (best[h] * (1 - found));
        }
    }
    free(best);
    return out;
}

static int hit_cmp(const void *a, const void *b) {     /* hit_t::compareHitsByScoreAndId, QueryMatcher.h:40-50 */
    const spo_hit *x = (const spo_hit *) a, *y = (const spo_hit *) b;
    int ax = abs(x->score), ay = abs(y->score);
    if (ax != ay) return ax > ay ? -1 : 1;
    if (x->seqId != y->seqId) return x->seqId < y->seqId ? -1 : 1;
    return 0;
}

size_t spo_prefilter_query(const spo_ctx *c, const char *query, int qLen, spo_hit *outHits, size_t cap) {
    unsigned char *q = (unsigned char *) malloc((size_t) qLen + 1);
    int L = spo_map_sequence(&c->seed, query, qLen, q);
    float *bias = (float *) malloc(sizeof(float) * (size_t) (L + 1));
    spo_local_bias(&c->seed, q, L, bias);
    cr_t *fd; size_t n = match_impl(c, q, L, bias, &fd, NULL, NULL);
    uprof_t p; uprof_build(c, q, L, bias, &p);
    const unsigned minDiagScoreThr = 15;                                /* --min-ungapped-score */
    size_t maxHits = c->n < (size_t) c->maxSeqs ? c->n : (size_t) c->maxSeqs;   /* Prefiltering.cpp:170, QueryMatcher.cpp:44 */
    crs_t *a = (crs_t *) malloc(sizeof(crs_t) * (n + 1) * 2);
    crs_t *b = a + (n + 1);
    unsigned sizes[256];
    memset(sizes, 0, sizeof(sizes));
    for (size_t i = 0; i < n; i++) { a[i].id = fd[i].id; a[i].diag = fd[i].diag; a[i].count = clamped_score(c, &p, fd[i].id, fd[i].diag); sizes[a[i].count]++; }
    size_t nAbove = radix_by_score(sizes, b, minDiagScoreThr, a, n);   /* QueryMatcher.cpp:107 */
    size_t nMax = keep_max_elements(c, b, nAbove, a);                  /* :137 ; a now = foundDiagonals+resultSize */
    memset(sizes, 0, sizeof(sizes));
    for (size_t i = 0; i < nMax; i++) sizes[a[i].count]++;
    unsigned thr; { size_t found = 0; unsigned t; for (t = 255; t > 0; t--) { found += sizes[t]; if (found >= maxHits) break; } thr = t; }   /* QueryMatcher.h:199-209 */
    if (thr < minDiagScoreThr) thr = minDiagScoreThr;
    const unsigned maxDiagThr = (unsigned) (255 - p.bias);
    size_t nOut = 0;
    /* resultSize < foundDiagonalsSize/2 always holds here (foundDiagonalsSize >= 1e6, QueryMatcher.cpp:41,145) */
    size_t nThr = radix_by_score(sizes, b, thr, a, nMax);              /* :148 */
    if (thr >= maxDiagThr) {
        /* score is truncated: rescoreHits + rescale (:149-155, :486-505, getResult :367-370) */
        int maxSelf = scalar_diag(p.prof, p.bias, (unsigned) L, q);
        { /* scoreSingleSequence(query,0,0): long path if L >= 32768 -- not reachable for the restated sizes */ }
        int lower = (int) maxDiagThr;
        maxSelf = maxSelf - lower; if (maxSelf < 1) maxSelf = 1; if (maxSelf > 65535) maxSelf = 65535;
        float fltMax = (float) maxSelf;
        memset(sizes, 0, sizeof(sizes));
        size_t el = 0;
        for (size_t i = 0; i < nThr && b[i].count >= lower; i++) {
            unsigned ns = (unsigned) exact_score(c, &p, b[i].id, b[i].diag);
            ns -= (unsigned) lower;
            float sc = (float) (ns < 65535u ? ns : 65535u);
            b[i].count = (unsigned char) ((sc / fltMax) * (float) 255 + 0.5);
            sizes[b[i].count]++; el++;
        }
        size_t n2 = radix_by_score(sizes, a, 0, b, el);
        for (size_t i = 0; i < n2 && nOut < maxHits; i++) {
            if (nOut < cap) { outHits[nOut].seqId = a[i].id; outHits[nOut].diagonal = a[i].diag;
                outHits[nOut].score = (int) ((unsigned) (255 - p.bias) + ((unsigned) a[i].count * (unsigned) maxSelf / 255)); }
            nOut++;
        }
    } else {
        for (size_t i = 0; i < nThr && nOut < maxHits; i++) {          /* getResult<UNGAPPED_DIAGONAL_SCORE>, :347-378 */
            if (b[i].count < thr) continue;
            int sc = b[i].count;
            if (sc >= 255 - p.bias) sc = exact_score(c, &p, b[i].id, b[i].diag);
            if (nOut < cap) { outHits[nOut].seqId = b[i].id; outHits[nOut].score = sc; outHits[nOut].diagonal = b[i].diag; }
            nOut++;
        }
    }
    size_t nn = nOut < cap ? nOut : cap;
    if (nn > 1) qsort(outHits, nn, sizeof(spo_hit), hit_cmp);         /* QueryMatcher.cpp:178-184 (keys are unique: one hit per seqId) */
    free(a); free(p.prof); free(fd); free(bias); free(q);
    return nOut;
}

static char *u32toa(uint32_t v, char *p) { char t[12]; int n = 0; do { t[n++] = (char) ('0' + v % 10); v /= 10; } while (v); while (n) *p++ = t[--n]; return p; }
static char *i32toa(int32_t v, char *p) { if (v < 0) { *p++ = '-'; return u32toa((uint32_t) (-(int64_t) v), p); } return u32toa((uint32_t) v, p); }

size_t spo_prefilter_text(const spo_hit *hits, size_t n, const uint32_t *keys, char *buf, size_t cap) {
    char *p = buf;
    for (size_t i = 0; i < n; i++) {
        if ((size_t) (p - buf) + 40 > cap) break;
        p = u32toa(keys ? keys[hits[i].seqId] : hits[i].seqId, p); *p++ = '\t';
        p = i32toa(hits[i].score, p); *p++ = '\t';
        p = i32toa((int32_t) (short) hits[i].diagonal, p); *p++ = '\n';
    }
    *p = '\0';
    return (size_t) (p - buf);
}

/* ------------------------------------------------------------------------------------------------
 * a18  E-values: lib/alp sls_alignment_evaluer.hpp:139-167 (evaluePerArea, bitScore),
 *      sls_pvalues.cpp:342-364 (thresholds), :366-560 (area with finite-size correction),
 *      sls_basic.hpp:195-198 (normal_probability).  The Gumbel parameters themselves come from ALP's
 *      Monte-Carlo fit and are supplied by the caller (captured from the reference, see
 *      tests/golden/*.json "alpGapped"/"alpGapless").
 * ---------------------------------------------------------------------------------------------- */
void spo_set_alp(spo_ctx *c, const spo_alp *gapped, const spo_alp *gapless) { c->alp[1] = *gapped; c->alp[0] = *gapless; c->haveAlp = 1; }

static double alp_area(const spo_alp *p, double y, double m_, double n_) {
    const double pi = 3.1415926535897932384626433832795;
    const double const_val = 1 / sqrt(2.0 * pi);
    double vi_thr = 2.0 * p->alpha_I / p->lambda; if (vi_thr < 0.0) vi_thr = 0.0;
    double vj_thr = 2.0 * p->alpha_J / p->lambda; if (vj_thr < 0.0) vj_thr = 0.0;
    double c_thr = 2.0 * p->sigma / p->lambda; if (c_thr < 0.0) c_thr = 0.0;
    double tmp = p->a_I * y + p->b_I;
    double m_li_y = m_ - tmp;
    double vi_y = p->alpha_I * y + p->beta_I; if (vi_y < vi_thr) vi_y = vi_thr;
    double sqrt_vi_y = sqrt(vi_y);
    double m_F = (sqrt_vi_y == 0.0) ? 1e100 : m_li_y / sqrt_vi_y;
    double P_m_F = 0.5 * erfc(-sqrt(0.5) * m_F);
    double E_m_F = -const_val * exp(-0.5 * m_F * m_F);
    double p1 = m_li_y * P_m_F - sqrt_vi_y * E_m_F;
    tmp = p->a_J * y + p->b_J;
    double n_lj_y = n_ - tmp;
    double vj_y = p->alpha_J * y + p->beta_J; if (vj_y < vj_thr) vj_y = vj_thr;
    double sqrt_vj_y = sqrt(vj_y);
    double n_F = (sqrt_vj_y == 0.0) ? 1e100 : n_lj_y / sqrt_vj_y;
    double P_n_F = 0.5 * erfc(-sqrt(0.5) * n_F);
    double E_n_F = -const_val * exp(-0.5 * n_F * n_F);
    double p2 = n_lj_y * P_n_F - sqrt_vj_y * E_n_F;
    double c_y = p->sigma * y + p->tau; if (c_y < c_thr) c_y = c_thr;
    double P_m_F_P_n_F = P_m_F * P_n_F;
    double c_y_P = c_y * P_m_F_P_n_F;
    double p1_p2 = p1 * p2;
    return p1_p2 + c_y_P;
}

/* EvalueComputation::computeEvalue (alignment/EvalueComputation.h:32-40): area(score, qLen, dbResidues);
   AlignmentEvaluer::area passes (seqlen2_, seqlen1_) as (m_, n_) (sls_alignment_evaluer.cpp:1011-1024) */
double spo_evalue(const spo_ctx *c, int gapped, double score, double qLen) {
    const spo_alp *p = &c->alp[gapped ? 1 : 0];
    double epa = p->K * exp(-p->lambda * score);
    double a = alp_area(p, score, (double) spo_aa_db_size(c), qLen);
    return epa * a;
}
double spo_bitscore(const spo_ctx *c, int gapped, double score) {
    const spo_alp *p = &c->alp[gapped ? 1 : 0];
    double logK = log(p->K);
    return (p->lambda * score - logK) / log(2.0);
}

/* ------------------------------------------------------------------------------------------------
 * a13-a16  SmithWaterman (alignment/StripedSmithWaterman.cpp).  Vectors are restated as lane arrays
 *          with the AVX2 lane counts (32 x u8, 16 x i16) so striping-dependent results stay equal.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int qL; int bias;
    unsigned char *q, *qrev;
    signed char *comp, *comprev;
    signed char mat[SPO_ALPH * SPO_ALPH];
} swq_t;

static void swq_init(const spo_ctx *c, const unsigned char *q, int L, swq_t *s) {     /* ssw_init, :689-779 */
    s->qL = L;
    s->q = (unsigned char *) malloc((size_t) L + 1); s->qrev = (unsigned char *) malloc((size_t) L + 1);
    s->comp = (signed char *) malloc((size_t) L + 1); s->comprev = (signed char *) malloc((size_t) L + 1);
    memcpy(s->q, q, (size_t) L);
    float *tmp = (float *) malloc(sizeof(float) * (size_t) (L + 1));
    spo_local_bias(&c->aln, q, L, tmp);
    int compositionBias = 0;
    for (int i = 0; i < L; i++) {
        s->comp[i] = (signed char) ((tmp[i] < 0.0) ? (double) tmp[i] - 0.5 : (double) tmp[i] + 0.5);
        compositionBias = (compositionBias < s->comp[i]) ? compositionBias : s->comp[i];
    }
    if (compositionBias > 0) compositionBias = 0;
    free(tmp);
    int bias = 0;
    for (int i = 0; i < SPO_ALPH; i++) for (int j = 0; j < SPO_ALPH; j++) { s->mat[i * SPO_ALPH + j] = (signed char) c->aln.sub[i][j]; if (s->mat[i * SPO_ALPH + j] < bias) bias = s->mat[i * SPO_ALPH + j]; }
    s->bias = abs(bias) + abs(compositionBias);
    for (int i = 0; i < L; i++) { s->qrev[i] = s->q[L - 1 - i]; s->comprev[i] = s->comp[L - 1 - i]; }
}
static void swq_free(swq_t *s) { free(s->q); free(s->qrev); free(s->comp); free(s->comprev); }

/* createQueryProfile<.., SUBSTITUTIONMATRIX> (:92-127): striped profile, value for residue nt, segment i, lane k */
static void make_profile(int *prof, int E, const unsigned char *qseq, const signed char *comp, const signed char *mat,
                         int qlen, int bias, int offset) {
    int segLen = (qlen + E - 1) / E;
    int *t = prof;
    for (int nt = 0; nt < SPO_ALPH; nt++) for (int i = 0; i < segLen; i++) { int j = i; for (int k = 0; k < E; k++) {
        *t++ = (j >= qlen) ? bias : mat[nt * SPO_ALPH + qseq[j + offset]] + comp[j + offset] + bias; j += segLen; } }
}

static inline int sat_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

typedef struct { int score, ref, read; } aend_t;

/* sw_sse2_byte (:301-516); profile values are stored as the int8_t the reference writes, then read as u8 */
static aend_t sw_byte(const unsigned char *db, int ref_dir, int db_length, int query_length, int gapO, int gapE,
                      const int *profInt, int terminate, int bias) {
    enum { E = 32 };
    int segLen = (query_length + E - 1) / E;
    int max = 0, end_query = query_length - 1, end_db = -1;
    size_t vec = (size_t) segLen * E;
    unsigned char *H0 = (unsigned char *) calloc(vec, 1), *H1 = (unsigned char *) calloc(vec, 1), *Ev = (unsigned char *) calloc(vec, 1), *Hmax = (unsigned char *) calloc(vec, 1);
    unsigned char *pvHStore = H0, *pvHLoad = H1;
    unsigned char vMaxScore[E], vMaxMark[E];
    memset(vMaxScore, 0, E); memset(vMaxMark, 0, E);
    int begin = 0, end = db_length, step = 1;
    if (ref_dir == 1) { begin = db_length - 1; end = -1; step = -1; }
    for (int i = begin; i != end; i += step) {
        unsigned char vF[E], vMaxColumn[E], vH[E], e[E];
        memset(vF, 0, E); memset(vMaxColumn, 0, E);
        /* vH = shiftl(pvHStore[segLen-1], 1) */
        vH[0] = 0; for (int k = 1; k < E; k++) vH[k] = pvHStore[(size_t) (segLen - 1) * E + k - 1];
        const int *vP = profInt + (size_t) db[i] * segLen * E;
        unsigned char *pv = pvHLoad; pvHLoad = pvHStore; pvHStore = pv;
        for (int j = 0; j < segLen; j++) {
            for (int k = 0; k < E; k++) {
                int h = sat_u8(vH[k] + (unsigned char) (signed char) vP[j * E + k]);
                h = sat_u8(h - bias);
                e[k] = Ev[j * E + k];
                if (e[k] > h) h = e[k];
                if (vF[k] > h) h = vF[k];
                if (h > vMaxColumn[k]) vMaxColumn[k] = (unsigned char) h;
                pvHStore[j * E + k] = (unsigned char) h;
                h = sat_u8(h - gapO);
                int ee = sat_u8(e[k] - gapE); if (h > ee) ee = h;
                Ev[j * E + k] = (unsigned char) ee;
                int f = sat_u8(vF[k] - gapE); if (h > f) f = h;
                vF[k] = (unsigned char) f;
                vH[k] = pvHLoad[j * E + k];
            }
        }
        /* lazy F (:418-448) */
        int j = 0;
        for (int k = 0; k < E; k++) vH[k] = pvHStore[j * E + k];
        for (int k = E - 1; k > 0; k--) vF[k] = vF[k - 1];
        vF[0] = 0;
        for (;;) {
            int all = 1;
            for (int k = 0; k < E; k++) { int t = sat_u8(vH[k] - gapO); t = sat_u8(vF[k] - t); if (t != 0) all = 0; }
            if (all) break;
            for (int k = 0; k < E; k++) {
                if (vF[k] > vH[k]) vH[k] = vF[k];
                if (vH[k] > vMaxColumn[k]) vMaxColumn[k] = vH[k];
                pvHStore[j * E + k] = vH[k];
                vF[k] = (unsigned char) sat_u8(vF[k] - gapE);
            }
            j++;
            if (j >= segLen) { j = 0; for (int k = E - 1; k > 0; k--) vF[k] = vF[k - 1]; vF[0] = 0; }
            for (int k = 0; k < E; k++) vH[k] = pvHStore[j * E + k];
        }
        int differs = 0;
        for (int k = 0; k < E; k++) { if (vMaxColumn[k] > vMaxScore[k]) vMaxScore[k] = vMaxColumn[k]; if (vMaxMark[k] != vMaxScore[k]) differs = 1; }
        if (differs) {
            int temp = 0;
            for (int k = 0; k < E; k++) { vMaxMark[k] = vMaxScore[k]; if (vMaxScore[k] > temp) temp = vMaxScore[k]; }
            if (temp > max) {
                max = temp;
                if (max + bias >= 255) break;
                end_db = i;
                memcpy(Hmax, pvHStore, vec);
            }
        }
        int colMax = 0; for (int k = 0; k < E; k++) if (vMaxColumn[k] > colMax) colMax = vMaxColumn[k];
        if (colMax == terminate) break;
    }
    int column_len = segLen * E;
    for (int i = 0; i < column_len; i++) if (Hmax[i] == max) { int temp = i / E + i % E * segLen; if (temp < end_query) end_query = temp; }
    aend_t r; r.score = (max + bias >= 255) ? 255 : max; r.ref = end_db; r.read = end_query;
    free(H0); free(H1); free(Ev); free(Hmax);
    return r;
}

static inline int sat_i16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
static inline int subs_u16(int a, int b) { unsigned ua = (unsigned) a & 0xFFFF, ub = (unsigned) b & 0xFFFF; return (int) (short) (ua > ub ? ua - ub : 0); }

/* sw_sse2_word (:519-687) */
static aend_t sw_word(const unsigned char *db, int ref_dir, int db_length, int query_length, int gapO, int gapE,
                      const int *prof, int terminate) {
    enum { E = 16 };
    int segLen = (query_length + E - 1) / E;
    int max = 0, end_read = query_length - 1, end_ref = 0;
    size_t vec = (size_t) segLen * E;
    short *H0 = (short *) calloc(vec, 2), *H1 = (short *) calloc(vec, 2), *Ev = (short *) calloc(vec, 2), *Hmax = (short *) calloc(vec, 2);
    short *pvHStore = H0, *pvHLoad = H1;
    short vMaxScore[E], vMaxMark[E];
    memset(vMaxScore, 0, sizeof(vMaxScore)); memset(vMaxMark, 0, sizeof(vMaxMark));
    int begin = 0, end = db_length, step = 1;
    if (ref_dir == 1) { begin = db_length - 1; end = -1; step = -1; }
    for (int i = begin; i != end; i += step) {
        short vF[E], vMaxColumn[E], vH[E], e[E];
        memset(vF, 0, sizeof(vF)); memset(vMaxColumn, 0, sizeof(vMaxColumn));
        vH[0] = 0; for (int k = 1; k < E; k++) vH[k] = pvHStore[(size_t) (segLen - 1) * E + k - 1];
        short *pv = pvHLoad;
        const int *vP = prof + (size_t) db[i] * segLen * E;
        pvHLoad = pvHStore; pvHStore = pv;
        for (int j = 0; j < segLen; j++) {
            for (int k = 0; k < E; k++) {
                int h = sat_i16(vH[k] + (short) vP[j * E + k]);
                e[k] = Ev[j * E + k];
                if (e[k] > h) h = e[k];
                if (vF[k] > h) h = vF[k];
                if (h > vMaxColumn[k]) vMaxColumn[k] = (short) h;
                pvHStore[j * E + k] = (short) h;
                h = subs_u16(h, gapO);
                int ee = subs_u16(e[k], gapE); if (h > ee) ee = h;
                Ev[j * E + k] = (short) ee;
                int f = subs_u16(vF[k], gapE); if (h > f) f = h;
                vF[k] = (short) f;
                vH[k] = pvHLoad[j * E + k];
            }
        }
        /* lazy F (:612-624) */
        int done = 0;
        for (int kk = 0; kk < E && !done; kk++) {
            for (int k = E - 1; k > 0; k--) vF[k] = vF[k - 1];
            vF[0] = 0;
            for (int j = 0; j < segLen; j++) {
                int any = 0;
                for (int k = 0; k < E; k++) {
                    short h = pvHStore[j * E + k];
                    if (vF[k] > h) h = vF[k];
                    if (h > vMaxColumn[k]) vMaxColumn[k] = h;
                    pvHStore[j * E + k] = h;
                    int hh = subs_u16(h, gapO);
                    vF[k] = (short) subs_u16(vF[k], gapE);
                    if (vF[k] > hh) any = 1;
                }
                if (!any) { done = 1; break; }
            }
        }
        int differs = 0;
        for (int k = 0; k < E; k++) { if (vMaxColumn[k] > vMaxScore[k]) vMaxScore[k] = vMaxColumn[k]; if (vMaxMark[k] != vMaxScore[k]) differs = 1; }
        if (differs) {
            int temp = 0;   /* simd_hmax16: unsigned maximum */
            for (int k = 0; k < E; k++) { vMaxMark[k] = vMaxScore[k]; int u = (unsigned short) vMaxScore[k]; if (u > temp) temp = u; }
            if (temp > max) { max = temp; end_ref = i; memcpy(Hmax, pvHStore, vec * 2); }
        }
        int colMax = 0; for (int k = 0; k < E; k++) { int u = (unsigned short) vMaxColumn[k]; if (u > colMax) colMax = u; }
        if (colMax == terminate) break;
    }
    int column_len = segLen * E;
    for (int i = 0; i < column_len; i++) if ((unsigned short) Hmax[i] == max) { int temp = i / E + i % E * segLen; if (temp < end_read) end_read = temp; }
    aend_t r; r.score = max; r.ref = end_ref; r.read = end_read;
    free(H0); free(H1); free(Ev); free(Hmax);
    return r;
}

/* banded_sw<SUBSTITUTIONMATRIX> (:780-984) -> uncompressed M/I/D string in `bt` (forward order) */
static int banded_sw(const unsigned char *db, const unsigned char *q, const signed char *comp, int db_length, int query_length,
                     int score, int gap_open, int gap_extend, int band_width, const signed char *mat, char *bt, size_t btCap) {
    int max = 0;
    int64_t width = 0, width_d = 0;
    int *h_b = NULL, *e_b = NULL, *h_c = NULL; signed char *direction = NULL;
    do {
        width = (int64_t) band_width * 2 + 3; width_d = (int64_t) band_width * 2 + 1;
        h_b = (int *) realloc(h_b, sizeof(int) * (size_t) (width + 2)); e_b = (int *) realloc(e_b, sizeof(int) * (size_t) (width + 2)); h_c = (int *) realloc(h_c, sizeof(int) * (size_t) (width + 2));
        direction = (signed char *) realloc(direction, (size_t) (width_d * query_length * 3 + 16));
        max = 0;   /* the reference keeps `max` across band doublings; it is monotone, so restarting at 0 is equivalent */
        for (int64_t j = 1; j < width - 1; j++) h_b[j] = 0;
        for (int i = 0; i < query_length; i++) {
            int beg = 0, end = db_length - 1, u = 0, edge, j, e, f;
            j = i - band_width; beg = beg > j ? beg : j;
            j = i + band_width; end = end < j ? end : j;
            edge = end + 1 < width - 1 ? end + 1 : (int) (width - 1);
            f = h_b[0] = e_b[0] = h_b[edge] = e_b[edge] = h_c[0] = 0;
            signed char *dl = direction + width_d * i * 3;
            for (j = beg; j <= end; j++) {
                int b, e1, f1, d, de, df, dh, temp1, temp2;
#define SET_U(u, w, i, j) { int x = (i) - (w); x = x > 0 ? x : 0; (u) = (j) - x + 1; }
#define SET_D(u, w, i, j, p) { int x = (i) - (w); x = x > 0 ? x : 0; x = (j) - x; (u) = x * 3 + p; }
                SET_U(u, band_width, i, j); SET_U(e, band_width, i - 1, j);
                SET_U(b, band_width, i, j - 1); SET_U(d, band_width, i - 1, j - 1);
                SET_D(de, band_width, i, j, 0); SET_D(df, band_width, i, j, 1); SET_D(dh, band_width, i, j, 2);
                temp1 = i == 0 ? -gap_open : h_b[e] - gap_open;
                temp2 = i == 0 ? -gap_extend : e_b[e] - gap_extend;
                e_b[u] = temp1 > temp2 ? temp1 : temp2;
                dl[de] = temp1 > temp2 ? 3 : 2;
                temp1 = h_c[b] - gap_open;
                temp2 = f - gap_extend;
                f = temp1 > temp2 ? temp1 : temp2;
                dl[df] = temp1 > temp2 ? 5 : 4;
                e1 = e_b[u] > 0 ? e_b[u] : 0;
                f1 = f > 0 ? f : 0;
                temp1 = e1 > f1 ? e1 : f1;
                temp2 = h_b[d] + mat[q[i] * SPO_ALPH + db[j]] + comp[i];
                h_c[u] = temp1 > temp2 ? temp1 : temp2;
                if (h_c[u] > max) max = h_c[u];
                if (temp1 <= temp2) dl[dh] = 1; else dl[dh] = e1 > f1 ? dl[de] : dl[df];
            }
            for (j = 1; j <= u; j++) h_b[j] = h_c[j];
        }
        band_width *= 2;
    } while (max < score);
    band_width /= 2;
    /* trace back (:881-960): ops are produced end -> start */
    int i = query_length - 1, j = db_length - 1, temp2 = 2, rc = 0;
    size_t n = 0;
    signed char *dl = direction + width_d * i * 3;
    char *rev = (char *) malloc((size_t) query_length + (size_t) db_length + 8);
    while (i > 0 || j > 0) {
        int t1; SET_D(t1, band_width, i, j, temp2);
        char op;
        switch (dl[t1]) {
            case 1: --i; --j; temp2 = 2; dl -= width_d * 3; op = 'M'; break;
            case 2: --i; temp2 = 0; dl -= width_d * 3; op = 'I'; break;
            case 3: --i; temp2 = 2; dl -= width_d * 3; op = 'I'; break;
            case 4: --j; temp2 = 1; op = 'D'; break;
            case 5: --j; temp2 = 2; op = 'D'; break;
            default: rc = -1; goto done;
        }
        rev[n++] = op;
    }
    /* the cigar assembly (:930-960) always closes with one extra 'M' for the first cell */
    rev[n++] = 'M';
    if (n + 1 > btCap) { rc = -2; goto done; }
    for (size_t k = 0; k < n; k++) bt[k] = rev[n - 1 - k];
    bt[n] = '\0';
done:
    free(rev); free(direction); free(h_c); free(e_b); free(h_b);
    return rc;
#undef SET_U
#undef SET_D
}

static float compute_cov(unsigned startPos, unsigned endPos, unsigned len) {    /* computeCov, :1055-1057 */
    unsigned mx = startPos > endPos ? startPos : endPos, mn = startPos < endPos ? startPos : endPos;
    unsigned a = len < mx ? len : mx;
    return (a - mn + 1) / (float) len;
}

int spo_sw_forward(const spo_ctx *c, const char *query, int qLen, const char *target, int tLen,
                   int gapOpen, int gapExtend, int *score, int *refEnd, int *readEnd, int *usedWord) {
    unsigned char *q = (unsigned char *) malloc((size_t) qLen + 1), *t = (unsigned char *) malloc((size_t) tLen + 1);
    int L = spo_map_sequence(&c->aln, query, qLen, q), TL = spo_map_sequence(&c->aln, target, tLen, t);
    swq_t s; swq_init(c, q, L, &s);
    int seg8 = (L + 31) / 32, seg16 = (L + 15) / 16;
    int *pb = (int *) malloc(sizeof(int) * (size_t) SPO_ALPH * seg8 * 32), *pw = (int *) malloc(sizeof(int) * (size_t) SPO_ALPH * seg16 * 16);
    make_profile(pb, 32, s.q, s.comp, s.mat, L, s.bias, 0);
    make_profile(pw, 16, s.q, s.comp, s.mat, L, 0, 0);
    aend_t b = sw_byte(t, 0, TL, L, gapOpen, gapExtend, pb, 255, s.bias);
    int word = 0;
    if (b.score == 255) { b = sw_word(t, 0, TL, L, gapOpen, gapExtend, pw, 65535); word = 1; }
    *score = b.score; *refEnd = b.ref; *readEnd = b.read; if (usedWord) *usedWord = word;
    free(pb); free(pw); swq_free(&s); free(q); free(t);
    return 0;
}

/* Matcher::getSWResult (alignment/Matcher.cpp:61-187) + ssw_align (StripedSmithWaterman.cpp:130-270)
   + Alignment::checkCriteria (Alignment.cpp:555-574) for  -a 1 --alignment-mode 2 -c 0 --min-seq-id 0 */
int spo_align_pair(const spo_ctx *c, const char *query, int qLen, const char *target, int tLen, uint32_t dbKey,
                   int gapOpen, int gapExtend, double evalThr, spo_aln *r) {
    memset(r, 0, sizeof(*r));
    unsigned char *q = (unsigned char *) malloc((size_t) qLen + 1), *t = (unsigned char *) malloc((size_t) tLen + 1);
    int L = spo_map_sequence(&c->aln, query, qLen, q), TL = spo_map_sequence(&c->aln, target, tLen, t);
    swq_t s; swq_init(c, q, L, &s);
    int seg8 = (L + 31) / 32, seg16 = (L + 15) / 16;
    int *pb = (int *) malloc(sizeof(int) * (size_t) SPO_ALPH * seg8 * 32), *pw = (int *) malloc(sizeof(int) * (size_t) SPO_ALPH * seg16 * 16);
    make_profile(pb, 32, s.q, s.comp, s.mat, L, s.bias, 0);
    make_profile(pw, 16, s.q, s.comp, s.mat, L, 0, 0);
    int word = 0;
    aend_t best = sw_byte(t, 0, TL, L, gapOpen, gapExtend, pb, 255, s.bias);
    if (best.score == 255) { best = sw_word(t, 0, TL, L, gapOpen, gapExtend, pw, 65535); word = 1; }
    r->dbKey = dbKey; r->rawScore = best.score; r->qLen = (uint32_t) L; r->dbLen = (uint32_t) TL;
    r->qStartPos = -1; r->dbStartPos = -1; r->qEndPos = best.read; r->dbEndPos = best.ref;
    r->score = (int) (spo_bitscore(c, 1, best.score) + 0.5);
    int rc = 0;
    if (best.ref == -1) { r->passed = 0; goto out; }          /* no residue could be aligned (:188-190) */
    r->eval = spo_evalue(c, 1, best.score, L);
    r->qcov = compute_cov(0, (unsigned) best.read, (unsigned) L);
    r->dbcov = compute_cov(0, (unsigned) best.ref, (unsigned) TL);
    if (r->eval > evalThr) { r->passed = 0; goto out; }       /* hasLowerEvalue (:193-200); covThr = 0 */
    {
        /* the reference reverses L+1 bytes (seq_reverse(..., q->L), StripedSmithWaterman.h:185-194) and reads at
           offset L - qEnd; with a plain length-L reversal that is offset L - 1 - qEnd */
        int queryOffset = L - 1 - best.read;
        int rl = best.read + 1;
        aend_t rev;
        if (!word) {
            int segr = (rl + 31) / 32; int *pr = (int *) malloc(sizeof(int) * (size_t) SPO_ALPH * segr * 32);
            make_profile(pr, 32, s.qrev, s.comprev, s.mat, rl, s.bias, queryOffset);
            rev = sw_byte(t, 1, best.ref + 1, rl, gapOpen, gapExtend, pr, best.score, s.bias);
            free(pr);
        } else {
            int segr = (rl + 15) / 16; int *pr = (int *) malloc(sizeof(int) * (size_t) SPO_ALPH * segr * 16);
            make_profile(pr, 16, s.qrev, s.comprev, s.mat, rl, 0, queryOffset);
            rev = sw_word(t, 1, best.ref + 1, rl, gapOpen, gapExtend, pr, best.score);
            free(pr);
        }
        if (rev.score != best.score) { rc = -10; goto out; }
        r->dbStartPos = rev.ref;
        r->qStartPos = best.read - rev.read;
        if (r->dbStartPos == -1) { rc = -11; goto out; }
        r->qcov = compute_cov((unsigned) r->qStartPos, (unsigned) r->qEndPos, (unsigned) L);
        r->dbcov = compute_cov((unsigned) r->dbStartPos, (unsigned) r->dbEndPos, (unsigned) TL);
        int dbl = r->dbEndPos - r->dbStartPos + 1, ql = r->qEndPos - r->qStartPos + 1;
        int band = abs(dbl - ql) + 1;
        rc = banded_sw(t + r->dbStartPos, s.q + r->qStartPos, s.comp + r->qStartPos, dbl, ql, best.score, gapOpen, gapExtend, band, s.mat, r->backtrace, sizeof(r->backtrace));
        if (rc) goto out;
        int ids = 0, tp = r->dbStartPos, qp = r->qStartPos;
        size_t alnLen = strlen(r->backtrace);
        for (size_t k = 0; k < alnLen; k++) {
            if (r->backtrace[k] == 'M') { if (t[tp] == q[qp]) ids++; qp++; tp++; }
            else if (r->backtrace[k] == 'I') qp++; else tp++;
        }
        r->alnLength = (uint32_t) alnLen;
        r->seqId = (float) ids / (float) (int) alnLen;      /* Util::computeSeqId, SEQ_ID_ALN_LEN */
        r->passed = (r->eval <= evalThr) && (r->seqId >= 0.0);
    }
out:
    free(pb); free(pw); swq_free(&s); free(q); free(t);
    return rc;
}

static int aln_cmp(const void *a, const void *b) {     /* Matcher::compareHits, Matcher.h:156-167 */
    const spo_aln *x = (const spo_aln *) a, *y = (const spo_aln *) b;
    if (x->eval != y->eval) return x->eval < y->eval ? -1 : 1;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;
    if (x->dbLen != y->dbLen) return x->dbLen < y->dbLen ? -1 : 1;
    if (x->dbKey != y->dbKey) return x->dbKey < y->dbKey ? -1 : 1;
    return 0;
}

/* Matcher::resultToBuffer (Matcher.cpp:323-370), Util::fastSeqIdToBuffer (Util.cpp:278-306), compressAlignment (:211-230) */
size_t spo_alignment_text(spo_aln *res, size_t n, char *buf, size_t cap) {
    if (n > 1) qsort(res, n, sizeof(spo_aln), aln_cmp);
    char *p = buf;
    for (size_t i = 0; i < n; i++) {
        const spo_aln *r = &res[i];
        if ((size_t) (p - buf) + 256 + 2 * strlen(r->backtrace) > cap) break;
        p = u32toa(r->dbKey, p); *p++ = '\t';
        p = i32toa(r->score, p); *p++ = '\t';
        if (r->seqId == 1.0) { memcpy(p, "1.00", 4); p += 4; }   /* the caller overwrites the third '0' with the tab (Util.cpp:279-290 returns the address OF the terminator) */
        else { *p++ = '0'; *p++ = '.'; if (r->seqId < 0.10) *p++ = '0'; if (r->seqId < 0.01) *p++ = '0'; p = i32toa((int) (r->seqId * 1000), p); }
        *p++ = '\t';
        p += sprintf(p, "%.3E", r->eval); *p++ = '\t';
        p = i32toa(r->qStartPos, p); *p++ = '\t'; p = i32toa(r->qEndPos, p); *p++ = '\t'; p = i32toa((int) r->qLen, p); *p++ = '\t';
        p = i32toa(r->dbStartPos, p); *p++ = '\t'; p = i32toa(r->dbEndPos, p); *p++ = '\t'; p = i32toa((int) r->dbLen, p); *p++ = '\t';
        const char *bt = r->backtrace; char state = 'M'; size_t counter = 0;
        for (size_t k = 0; bt[k]; k++) { if (bt[k] != state) { p = u32toa((uint32_t) counter, p); *p++ = state; state = bt[k]; counter = 1; } else counter++; }
        p = u32toa((uint32_t) counter, p); *p++ = state;
        *p++ = '\n';
    }
    *p = '\0';
    return (size_t) (p - buf);
}

/* ------------------------------------------------------------------------------------------------
 * a20  DistanceCalculator::computeUngappedAlignment (alignment/DistanceCalculator.h:93-295, all five rescore modes) on ASCII
 *      residues with the 123x123 char matrix of SubstitutionMatrix::createAsciiSubMat
 * ---------------------------------------------------------------------------------------------- */
static void by_diagonal(const spo_ctx *c, const char *q, unsigned qLen, const char *t, unsigned tLen, int diagonal, int mode, spo_local *res) {
    unsigned minDist = (unsigned) abs(diagonal);
    res->score = 0; res->startPos = -1; res->endPos = -1; res->diagonalLen = 0;   /* LocalAlignment() defaults */
    res->distToDiagonal = minDist; res->diagonal = diagonal;
    const char *s1 = NULL, *s2 = NULL; unsigned len = 0;
    if (diagonal >= 0 && minDist < qLen) { len = tLen < qLen - minDist ? tLen : qLen - minDist; s1 = q + minDist; s2 = t; }
    else if (diagonal < 0 && minDist < tLen) { len = (tLen - minDist) < qLen ? (tLen - minDist) : qLen; s1 = q; s2 = t + minDist; }
    else return;
    res->diagonalLen = (int) len;
#define SPO_PAIR(pos) ((int) (signed char) c->aln.sub[c->aln.aa2num[(unsigned char) s1[pos] > 122 ? 'X' : (unsigned char) s1[pos]]][c->aln.aa2num[(unsigned char) s2[pos] > 122 ? 'X' : (unsigned char) s2[pos]]])
    if (mode == 0) {       /* RESCORE_MODE_HAMMING: computeInverseHammingDistance (:275-295) -- equal letters, bytes compared as they are */
        unsigned same = 0;
        for (unsigned pos = 0; pos < len; pos++) same += (s1[pos] == s2[pos]);
        res->score = (int) same;
        return;
    }
    if (mode == 3 || mode == 4) {
        unsigned first = (s1[0] == '*' || s2[0] == '*') ? 1u : 0u, last = len - 1;
        if (last > 0 && (s1[len - 1] == '*' || s2[len - 1] == '*')) last--;
        if (mode == 3) {   /* RESCORE_MODE_GLOBAL_ALIGNMENT: computeGlobalSubstitutionStartEndDistance (:203-220) */
            long long sum = 0;
            for (unsigned pos = first; pos <= last && last != 0xFFFFFFFFu; pos++) sum += SPO_PAIR(pos);
            if (sum < 0) sum = 0;
            res->score = (int) sum; res->startPos = (int) first; res->endPos = (int) last;
            return;
        }
        /* RESCORE_MODE_WINDOW_QUALITY_ALIGNMENT: computeWindowQualitySubstitutionStartEndDistance (:222-272), windowSize 20, historyErrors 5 */
        unsigned long long window = 0, windowMask = 1ull << 19;
        unsigned currErrors = 0, maxLength = 0, currLength = 0, maxEndPos = 0, maxStartPos = 0, startPos = first;
        int maxScore = 0;
        for (unsigned pos = first; pos <= last && last != 0xFFFFFFFFu; pos++) {
            int currMatch = s1[pos] == s2[pos];
            if (window & windowMask) currErrors -= 1;
            window <<= 1;
            if (!currMatch) { window |= 1ull; currErrors += 1; }
            currLength += 1;
            if (pos >= 19 && currErrors > 5) { startPos = pos - 20 + 2; currLength = 19; }
            if (currLength > maxLength) { maxStartPos = startPos; maxEndPos = pos; maxLength = currLength; }
        }
        for (unsigned pos = maxStartPos; pos < maxEndPos; pos++) maxScore += SPO_PAIR(pos);
        res->score = maxScore; res->startPos = (int) maxStartPos; res->endPos = (int) maxEndPos;
        return;
    }
    int maxScore = 0, maxEnd = 0, maxStart = 0, minPos = -1, score = 0;
    for (unsigned pos = 0; pos < len; pos++) {
        int curr = SPO_PAIR(pos);
        score = curr + score;
        int isMin = (score <= 0);
        score = isMin ? 0 : score;
        minPos = isMin ? (int) pos : minPos;
        int isNewMax = (score > maxScore);
        maxEnd = isNewMax ? (int) pos : maxEnd;
        maxStart = isNewMax ? minPos + 1 : maxStart;
        maxScore = isNewMax ? score : maxScore;
    }
    res->score = maxScore;
    if (mode == 2) { res->startPos = maxStart; res->endPos = maxEnd; }
}

void spo_rescore_diagonal(const spo_ctx *c, const char *q, int qLen, const char *t, int tLen, uint16_t diag, int mode, spo_local *out) {
    spo_local max; max.score = 0; max.startPos = -1; max.endPos = -1; max.diagonalLen = 0; max.diagonal = 0; max.distToDiagonal = 0;
    int have = 0;
    for (unsigned d = 1; d <= 1 + (unsigned) tLen / 32768; d++) {
        int real = (int) (-(int64_t) d * 65536 + diag);
        spo_local tmp; by_diagonal(c, q, (unsigned) qLen, t, (unsigned) tLen, real, mode, &tmp);
        if ((unsigned) tmp.score > (unsigned) max.score) { max = tmp; have = 1; }      /* LocalAlignment::score is unsigned (DistanceCalculator.h:45) */
    }
    for (unsigned d = 0; d <= (unsigned) qLen / 65536; d++) {
        int real = (int) (d * 65536 + diag);
        spo_local tmp; by_diagonal(c, q, (unsigned) qLen, t, (unsigned) tLen, real, mode, &tmp);
        if ((unsigned) tmp.score > (unsigned) max.score) { max = tmp; have = 1; }      /* LocalAlignment::score is unsigned (DistanceCalculator.h:45) */
    }
    (void) have;
    *out = max;
}
