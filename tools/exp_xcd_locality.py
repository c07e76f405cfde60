"""CPU simulation: pbits line accesses (one 128-byte line per 32 x 32 tile) of the similar k-mers of sampled query windows, under an LRU cache per XCD, for
(1) queries dealt to the 8 XCDs at random and (2) queries grouped by the Z-rank of their first window's first 3-mer."""
import sys, ctypes as C, random
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import bench
from oracle_lib import Oracle, Matrix
NQ = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
tres, toff, qres, qoff = bench.make_workload(10000, 50000, 200, 12345)
o = Oracle()
o.set_targets([b"ACDEFGHIKLMNPQRSTVWY" * 3])
L = o.L
L.spo_window_thresholds.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]
L.spo_local_bias.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_float)]
L.spo_map_sequence.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p]
seedm = L.spo_seed_matrix(o.ctx)
# Z-curve renumbering (sp_gpu.hip: perm3Host)
pi = [2, 0, 7, 8, 15, 1, 12, 18, 10, 17, 16, 6, 5, 9, 11, 3, 4, 19, 14, 13]
code = []
for idx in range(8000):
    d = [pi[idx % 20], pi[(idx // 20) % 20], pi[idx // 400]]
    z = 0
    for bit in range(5):
        for j in range(3):
            z |= ((d[j] >> bit) & 1) << (3 * bit + j)
    code.append((z, idx))
code.sort()
fwd = np.zeros(8000, dtype=np.int64)
for rank, (z, idx) in enumerate(code):
    fwd[idx] = rank
rnd = random.Random(1)
sample = rnd.sample(range(len(qoff) - 1), NQ)
lines_per_query, keyA = [], []
for qi in sample:
    q = qres[qi * 10:(qi + 1) * 10].tobytes()
    num = C.create_string_buffer(16)
    L.spo_map_sequence(seedm, q, 10, num)
    bias = (C.c_float * 10)(); thr = (C.c_int * 10)()
    L.spo_local_bias(seedm, num.raw[:10], 10, bias)
    L.spo_window_thresholds(o.ctx, num.raw[:10], 10, bias, thr)
    tiles = []
    for w in range(2):      # windows 0 .. L-9
        if thr[w] < 0: continue
        k6 = bytes(num.raw[w + p] for p in (0, 1, 3, 4, 5, 7))
        out, n = o.kmer_list(k6, thr[w])
        km = np.frombuffer(out, dtype=np.uint32, count=n).astype(np.int64)
        a, b = fwd[km % 8000], fwd[km // 8000]
        tiles.append((a >> 5) * 250 + (b >> 5))
    t = np.concatenate(tiles) if tiles else np.zeros(0, dtype=np.int64)
    lines_per_query.append(t)
    r = num.raw
    keyA.append(int(fwd[r[0] + 20 * r[1] + 400 * r[3]]))
print("queries", NQ, "probes/query", np.mean([len(t) for t in lines_per_query]), "distinct lines/query", np.mean([len(np.unique(t)) for t in lines_per_query]))
# tile rows touched per query
print("distinct tile rows per query (of 250):", np.mean([len(np.unique(t // 250)) for t in lines_per_query]))
from collections import OrderedDict
def lru_misses(order, cap):
    cache = OrderedDict(); miss = acc = 0
    for qi in order:
        for ln in np.unique(lines_per_query[qi]):      # (a query's own repeats hit L1 / are one request)
            acc += 1
            if ln in cache: cache.move_to_end(ln)
            else:
                miss += 1; cache[ln] = 1
                if len(cache) > cap: cache.popitem(last=False)
    return miss, acc
# scale: the real batch has 60000 queries over 8 XCDs; the sample has NQ -> per-XCD stream NQ/8 queries; cache capacity scaled so that (lines per query x concurrent queries) vs capacity matches:
# real: 384 waves per XCD x 4 queries in flight; LRU capacity for pbits in a 4 MB L2 shared with other data: try 8K, 16K, 24K lines
for cap in (8192, 16384, 24576):
    rand_order = list(range(NQ)); rnd.shuffle(rand_order)
    m1 = a1 = 0
    for x in range(8):
        m, a = lru_misses(rand_order[x::8], cap); m1 += m; a1 += a
    by_a = sorted(range(NQ), key=lambda i: keyA[i])
    m2 = a2 = 0
    per = (NQ + 7) // 8
    for x in range(8):
        grp = by_a[x * per:(x + 1) * per]; rnd.shuffle(grp)
        m, a = lru_misses(grp, cap); m2 += m; a2 += a
    print("LRU %5d lines/XCD: random deal miss rate %.3f | grouped by first 3-mer miss rate %.3f  (accesses %d)" % (cap, m1 / a1, m2 / a2, a1))
