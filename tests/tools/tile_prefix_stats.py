"""How long is the common prefix of the suffixes one local-sort tile of the sample sort holds?  (tiles whose
elements share >= 35 of the 99 key bits can be sorted by ONE 64-bit window of the 128-bit [key | index] number.)
A sorted random sample of suffixes stands in for the splitters: a tile of ~3456 elements spans ~27 of 2^22 samples.
usage: python tests/tools/tile_prefix_stats.py <corpus kind | real> [logn=29]"""
import sys
import numpy as np
sys.path.insert(0, '.')
from pysubstringsearch_amd import _ffi  # noqa: E402

KINDS = {'lines': 0, 'words': 1, 'mixed': 6, 'source': 7}
kind = sys.argv[1] if len(sys.argv) > 1 else 'words'
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 29
n = 1 << logn
if kind == 'real':
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('real_text', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'real_text.py'))
    rt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rt)
    t = np.frombuffer(rt.collect(n), dtype=np.uint8).copy()
else:
    t = np.empty(n, dtype=np.uint8)
    _ffi.check(_ffi.lib.pss_gen_corpus(KINDS[kind], t.ctypes.data, n, 0))
n = t.size
S = 1 << 22
rng = np.random.default_rng(5)
pos = rng.integers(0, n - 40, S)
W = 24
win = np.stack([t[pos + k] for k in range(W)], axis=1)                    # S x 24 bytes
keys = win.view('>u8').reshape(S, 3)
order = np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0]))
win = win[order]
sigma = len(np.unique(t[: 1 << 24]))
bits_per_symbol = np.log2(sigma + 1)
for span, what in ((4, 'one bucket (4 samples)'), (27, 'one tile (~27 samples)')):
    a, b = win[:-span], win[span:]
    neq = a != b
    cp = np.where(neq.any(axis=1), neq.argmax(axis=1), W)                # common prefix in symbols
    need = 35.0 / bits_per_symbol
    print(f'{kind}: {what}: common prefix symbols p10/p50/p90 = {np.percentile(cp, [10, 50, 90]).tolist()}, '
          f'share with >= {need:.1f} symbols (35 key bits at {bits_per_symbol:.2f} bits/symbol): {(cp >= np.ceil(need)).mean():.3f}, '
          f'>= {47.0 / bits_per_symbol:.1f} symbols (47 bits): {(cp >= np.ceil(47.0 / bits_per_symbol)).mean():.3f}')
