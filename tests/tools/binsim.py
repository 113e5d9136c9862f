"""Would an LDS counting sort do for the tiles of the sample sort (natural text)?  Simulation on the CPU (round 4):
the `words` chunk, buckets of 512 consecutive suffixes of the suffix array (what the splitters cut), every bucket's keys
-- the symbols behind the bucket's common prefix -- normalised to the bucket's own range and dropped into BINS bins.
Prints how crowded the bins get: sum c^2 / n (compares per element of an in-bin ranking), the largest bin per bucket, the
share of buckets / elements with a bin above 64.

    python tests/tools/binsim.py [bins per bucket = 512]
"""
import sys, numpy as np
sys.path.insert(0, '.')
from oracle import oracle as O
from pysubstringsearch_amd import _ffi
n = 1 << 22
t = np.empty(n, dtype=np.uint8)
_ffi.lib.pss_gen_corpus(1, t.ctypes.data, n, 0)
sa = O.sa(t)
# codes: dense 1..sigma
vals = np.unique(t)
lut = np.zeros(256, dtype=np.int64); lut[vals] = np.arange(1, len(vals)+1)
sig = len(vals) + 1
codes = np.concatenate([lut[t], np.zeros(64, dtype=np.int64)])
K = 20
B = 512          # bucket population
BINS = int(sys.argv[1]) if len(sys.argv) > 1 else 512
stats = []
rng = np.random.default_rng(1)
nb = n // B
picks = rng.choice(nb, 3000, replace=False)
tot = 0; sumsq = 0; mx = []
big = 0
for b in picks:
    idx = sa[b*B:(b+1)*B].astype(np.int64)
    sym = codes[idx[:, None] + np.arange(K)[None, :]]      # B x K
    # common prefix of the bucket
    same = (sym == sym[0]).all(axis=0)
    L = int(np.argmin(same)) if not same.all() else K
    # numeric value of symbols L..L+11 (float64 ~ 53 bits: 10-11 symbols exact)
    w = sym[:, L:min(K, L+11)].astype(np.float64)
    pw = float(sig) ** np.arange(w.shape[1]-1, -1, -1)
    x = (w * pw).sum(axis=1)
    lo, hi = x.min(), x.max()
    if hi == lo:
        bins = np.zeros(B, dtype=np.int64)
    else:
        bins = np.minimum(((x - lo) / (hi - lo) * BINS).astype(np.int64), BINS-1)
    c = np.bincount(bins, minlength=BINS)
    tot += B; sumsq += int((c.astype(np.int64)**2).sum()); mx.append(int(c.max()))
    big += int(c[c > 64].sum())
mx = np.array(mx)
print('bins per bucket', BINS, 'mean compares per element (sum c^2 / n):', sumsq / tot, 'max bin: median', np.median(mx), 'p90', np.percentile(mx, 90), 'p99', np.percentile(mx,99), 'frac buckets with a bin > 64:', (mx > 64).mean(), 'elements in bins>64:', big/tot)
