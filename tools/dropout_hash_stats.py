"""Statistics of the dropout mask hash (csrc/vc_rt.h: vc_drop_hash) — the r05 two-multiply mixer (F) against r01-r04's murmur-fmix form (A) and the
two-multiply candidates that FAIL (B, D: key injected after the first multiply — masks of different sites correlate at 0.2; C: no second key word — the
mask streams of two sites are XOR-permutations of each other).  numpy restatement of the integer arithmetic; run:  python tools/dropout_hash_stats.py
Per key: keep rates of the two 12-bit draws at p = 0.1 (threshold 410 / 4096), chi-square of the 4096 draw values over 4 M consecutive pairs, the largest
|correlation| among (lo, hi of one hash) and strides 1, 2, 25, 32, 256, 768, 1536; then the largest |correlation| of the same indices under nearby keys."""
import numpy as np
M = np.uint32
def fmix(x):
    x = x.copy()
    x ^= x >> M(16); x *= M(0x85EBCA6B); x ^= x >> M(13); x *= M(0xC2B2AE35); x ^= x >> M(16); return x
def hashA(pair, key): return fmix((pair * M(0x9E3779B1)) ^ M(key))
def hashB(pair, key):
    x = (pair * M(0x9E3779B1)) ^ M(key)
    x ^= x >> M(16); x *= M(0x7FEB352D); x ^= x >> M(15); return x
def hashC(pair, key):    # one multiply only after xor
    x = (pair ^ M(key)) * M(0x9E3779B1); x ^= x >> M(15); x *= M(0x2C1B3C6D); x ^= x >> M(13); return x
def hashD(pair, key):    # B with different shifts, draws from top
    x = (pair * M(0x9E3779B1)) ^ M(key)
    x ^= x >> M(15); x *= M(0x2C1B3C6D); x ^= x >> M(12); return x
def draws(h): return ((h >> M(8)) & M(0xFFF)).astype(np.int64), (h >> M(20)).astype(np.int64)
def stats(name, hf):
    rng = np.random.default_rng(0)
    res = []
    for key in [1, 0x12345679, 0xDEADBEEF | 1, 3, 0x80000001]:
        n = 1 << 22
        pair = np.arange(n, dtype=np.uint32)
        h = hf(pair, key)
        lo, hi = draws(h)
        thr = 410
        klo, khi = (lo >= thr), (hi >= thr)
        p_lo, p_hi = 1 - klo.mean(), 1 - khi.mean()
        # chi-square uniformity of 12-bit draws
        cl = np.bincount(lo, minlength=4096); ch = np.bincount(hi, minlength=4096)
        e = n / 4096
        chi_lo = ((cl - e) ** 2 / e).sum(); chi_hi = ((ch - e) ** 2 / e).sum()
        # correlations of keep decisions: lo-hi same hash, adjacent pairs, stride 25 (row of 50), stride 256, 1536
        def corr(a, b): a = a - a.mean(); b = b - b.mean(); return float((a * b).mean() / np.sqrt((a * a).mean() * (b * b).mean()))
        c = [corr(klo.astype(float), khi.astype(float))]
        for s in (1, 2, 25, 32, 256, 768, 1536):
            c.append(corr(klo[:-s].astype(float), klo[s:].astype(float))); c.append(corr(khi[:-s].astype(float), klo[s:].astype(float)))
        res.append((key, p_lo, p_hi, chi_lo, chi_hi, max(abs(x) for x in c)))
    # across keys (same index, different site keys / consecutive step seeds)
    pair = np.arange(1 << 20, dtype=np.uint32)
    k1 = draws(hf(pair, 0x1234567 | 1))[0] >= 410; 
    worst = 0
    for dk in (2, 4, 0x100, 0x10000, 0x9E3779B8):
        k2 = draws(hf(pair, (0x1234567 | 1) + dk))[0] >= 410
        a = k1.astype(float) - k1.mean(); b = k2.astype(float) - k2.mean()
        worst = max(worst, abs(float((a * b).mean() / np.sqrt((a * a).mean() * (b * b).mean()))))
    print(name)
    for r in res: print("  key %08x p_lo %.5f p_hi %.5f chi2 lo %.0f hi %.0f (dof 4095, sd 90) max|corr| %.5f" % r)
    print("  across keys max |corr| %.5f  (noise level 1/sqrt(n): %.5f / %.5f)" % (worst, 1 / np.sqrt(1 << 20), 1 / np.sqrt(1 << 22)))
for nm, hf in (("A murmur fmix (current, 3 mul)", hashA), ("B 2 mul", hashB), ("C 2 mul, xor-first", hashC), ("D 2 mul alt", hashD)):
    stats(nm, hf)
def hashF_factory(k2f):
    def hashF(pair, key):
        k2 = k2f(key)
        x = (pair ^ M(key)) * M(0x9E3779B1); x ^= x >> M(15); x ^= M(k2); x *= M(0x2C1B3C6D); x ^= x >> M(13); return x
    return hashF
def k2_of(key):
    x = np.array([key], dtype=np.uint32); return int(fmix(x * M(0x9E3779B1) + M(0x7F4A7C15))[0])
print("---- F: C + second key word xored between the multiplies")
stats("F 2 mul, xor key, k2 mid", hashF_factory(k2_of))
# permutation structure check: for C, mask(site1, i) == mask(site2, i ^ k1 ^ k2); for F it must not be
k1, k2 = 0x1234567 | 1, (0x1234567 | 1) ^ 0x40
pair = np.arange(1 << 16, dtype=np.uint32)
for nm, hf in (("C", hashC), ("F", hashF_factory(k2_of))):
    a = draws(hf(pair, k1))[0] >= 410; b = draws(hf(pair ^ M(0x40), k2))[0] >= 410
    print(nm, "fraction equal under the xor-permutation:", float((a == b).mean()), "(independent: %.3f)" % (0.9 * 0.9 + 0.1 * 0.1))
