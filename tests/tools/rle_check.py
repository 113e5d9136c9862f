"""Run-length path (rle_build.hip) against the oracle: forced on random texts of every shape, then the
full-size adversarial corpora against libsais' hashes, timed with and without the path."""
import os, sys, ctypes
sys.path.insert(0, '/root/repo')
import numpy as np
from pysubstringsearch_amd import _ffi
from oracle import oracle as O
import torch
lib = _ffi.lib


def build(host, flags=0):
    n = host.size
    dT = torch.from_numpy(host).cuda(); dSA = torch.empty(n, dtype=torch.int32, device='cuda')
    st = _ffi.SaStats()
    _ffi.check(lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, flags, ctypes.byref(st)))
    return dSA.cpu().numpy(), st.as_dict()


def runs_text(rng, n, alpha, maxrun):
    out = np.empty(n + maxrun, np.uint8); o = 0
    while o < n:
        L = int(rng.integers(1, maxrun + 1)); out[o:o + L] = 40 + int(rng.integers(0, alpha)); o += L
    return out[:n].copy()


rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 120
for trial in range(trials):
    os.environ['PSS_RLE'] = '1' if trial % 4 else ''
    if not os.environ['PSS_RLE']: os.environ.pop('PSS_RLE')
    os.environ.pop('PSS_RLE_SORT', None)
    if trial % 5 == 4: os.environ['PSS_RLE_SORT'] = '1'
    n = int(rng.choice([2, 3, 5, 17, 100, 4095, 4096, 4097, 8192, 20000, 70001, 300000, 1 << 20, (1 << 21) + 77]))
    kind = trial % 6
    if kind == 0: t = rng.integers(0, int(rng.choice([1, 2, 3, 255])), n).astype(np.uint8)          # no long runs at all (alphabet 1: one run)
    elif kind == 1: t = runs_text(rng, n, int(rng.choice([1, 2, 3, 50])), int(rng.choice([2, 9, 100, 5000])))
    elif kind == 2:
        p = int(rng.choice([2, 3, 16, 100, 4096])); t = np.full(n, 97, np.uint8); t[p - 1::p] = 10          # periodic
    elif kind == 3:
        unit = runs_text(rng, int(rng.choice([7, 50, 1000])), 2, 20); t = np.tile(unit, n // unit.size + 1)[:n].copy()   # periodic run pattern
    elif kind == 4:
        t = runs_text(rng, n, 2, 300); t[rng.integers(0, n, max(1, n // 1000))] = 10
    else:
        t = runs_text(rng, n, 256 - 40, 40) if n > 100 else runs_text(rng, n, 3, 4)
        t[:] = np.where(t == 40, 0, t); t[-1] = 255
    sa, st = build(t)
    ok = np.array_equal(sa, O.sa(t))
    print(trial, 'n', n, 'kind', kind, 'rle', st['rle'], 'runs', st['runs'], 'idbits', st['rle_id_bits'], 'rounds', st['rounds'],
          'OK' if ok else 'FAIL', flush=True)
    bad += (not ok)
os.environ.pop('PSS_RLE', None); os.environ.pop('PSS_RLE_SORT', None)
print('BAD', bad)
if len(sys.argv) > 3: sys.exit(bad)
import bench
gold = bench.load_big_goldens()
n = 1 << 29
for kind, name in ((2, 'runs'), (3, 'periodic')):
    t = np.empty(n, np.uint8); lib.pss_gen_corpus(kind, t.ctypes.data, n, 0)
    dT = torch.from_numpy(t).cuda(); dSA = torch.empty(n, dtype=torch.int32, device='cuda'); st = _ffi.SaStats()
    g = gold[(name, 0, n)]
    for env, flags in ((None, 0), (None, 0), (None, 1), ('sort', 0), ('sort', 1), ('0', 0)):
        os.environ.pop('PSS_RLE', None); os.environ.pop('PSS_RLE_SORT', None)
        if env == 'sort': os.environ['PSS_RLE_SORT'] = '1'
        elif env is not None: os.environ['PSS_RLE'] = env
        _ffi.check(lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, flags, ctypes.byref(st)))
        d = st.as_dict()
        print(name, 'PSS_RLE', env, 'flags', flags, 'ms', round(d['ms_total'], 2), 'rle', d['rle'], 'runs', d['runs'], 'idbits', d['rle_id_bits'],
              'rounds', d['rounds'], 'table', round(d['rle_ms_table'], 2), 'reduced', round(d['rle_ms_reduced'], 2), 'expand', round(d['rle_ms_expand'], 2),
              'verified', bench.sa_poly64_torch(dSA) == g['sa_poly64'], flush=True)
    os.environ.pop('PSS_RLE', None)
