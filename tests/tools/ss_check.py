"""The sample sort over 16-byte elements (ss_sort_impl.h) forced on random alphabets / sizes / repeat structures and on the
four corpora against the oracle, then `words` and `lines` 2^29 timed through it (checksum-verified against libsais)."""
import ctypes
import os
import sys

sys.path.insert(0, '.')
import numpy as np
import torch

from oracle import oracle as O
from pysubstringsearch_amd import _ffi

lib = _ffi.lib


def build(host, flags=0):
    n = host.size
    dT = torch.from_numpy(host).cuda()
    dSA = torch.empty(n, dtype=torch.int32, device='cuda')
    st = _ffi.SaStats()
    _ffi.check(lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, flags, ctypes.byref(st)))
    return dSA.cpu().numpy(), st.as_dict()


quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'
os.environ['PSS_SS'] = '1'
os.environ['PSS_MSD'] = '0'
rng = np.random.default_rng(0)
bad = 0
for trial in range(20 if quick else 70):
    os.environ.pop('PSS_MODE', None)
    if trial % 5 == 2:
        os.environ['PSS_MODE'] = ['dense', 'text'][trial % 2]
    n = int(rng.choice([1 << 16, 70001, 100003, 300000, 1 << 20, (1 << 21) + 77, (1 << 22) + 5]))
    alpha = int(rng.choice([1, 2, 3, 4, 16, 27, 39, 100, 255, 256]))
    t = (rng.integers(0, alpha, n).astype(np.uint16) + (0 if alpha > 200 else 40)).astype(np.uint8)
    kind = trial % 4
    if kind == 1:      # repeats: a vocabulary of short words
        words = [bytes(rng.integers(97, 97 + min(alpha, 26), int(rng.integers(2, 9))).astype(np.uint8)) for _ in range(50)]
        t = np.frombuffer(b' '.join(words[int(i)] for i in rng.integers(0, 50, n // 4)), dtype=np.uint8)[:n].copy()
    elif kind == 2:    # long duplicated blocks
        blk = t[:5000].copy()
        for o in rng.integers(0, n - 5000, 20):
            t[o:o + 5000] = blk
    if rng.random() < 0.3:
        t[rng.integers(0, t.size, max(1, t.size // 50))] = 10
    sa, st = build(t)
    ok = np.array_equal(sa, O.sa(t))
    print(trial, t.size, alpha, kind, os.environ.get('PSS_MODE'), 'ss', st['ss'], 'maxb', st['ss_max_bucket'], 'buckets', st['ss_buckets'],
          'tiles', st['ss_tiles'], 'key_chars', st['key_chars'], 'rounds', st['rounds'], 'OK' if ok else 'FAIL', flush=True)
    bad += (not ok) or (st['ss'] != 1 and t.size >= (1 << 16) and st['sigma'] > 1)
os.environ.pop('PSS_MODE', None)
for kind in (0, 1):
    n = 1 << 22
    t = np.empty(n, np.uint8)
    lib.pss_gen_corpus(kind, t.ctypes.data, n, 0)
    sa, st = build(t)
    ok = np.array_equal(sa, O.sa(t))
    bad += not ok
    print('corpus', kind, 'ss', st['ss'], 'maxb', st['ss_max_bucket'], 'OK' if ok else 'FAIL', flush=True)
print('BAD', bad, flush=True)
if quick:
    sys.exit(1 if bad else 0)
os.environ.pop('PSS_MSD')
os.environ.pop('PSS_SS')
import bench

n = 1 << 29
dSA = torch.empty(n, dtype=torch.int32, device='cuda')
st = _ffi.SaStats()
for kind, name, envs in ((1, 'words', ({'PSS_SS': '0'}, {})), (0, 'lines', ({}, {'PSS_SS': '1', 'PSS_MSD': '0'}))):
    t = np.empty(n, np.uint8)
    lib.pss_gen_corpus(kind, t.ctypes.data, n, 0)
    dT = torch.from_numpy(t).cuda()
    g = bench.load_big_goldens()[(name, 0, n)]
    for env in envs:
        for k in ('PSS_SS', 'PSS_MSD'):
            os.environ.pop(k, None)
        os.environ.update(env)
        for flags in (0, 0, 1):
            _ffi.check(lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, flags, ctypes.byref(st)))
            d = st.as_dict()
            print(name, env, 'flags', flags, 'ms', round(d['ms_total'], 2), 'initial', round(d['ms_initial'], 2), 'ss', d['ss'], 'msd', d['msd'],
                  'maxb', d['ss_max_bucket'], 'tiles', d['ss_tiles'], 'sample', round(d['ss_ms_sample'], 2), 'g1', round(d['ss_ms_g1'], 2),
                  'g2', round(d['ss_ms_g2'], 2), 'loc', round(d['ss_ms_local'], 2), 'rounds', d['rounds'], 'active', d['sum_active'],
                  'verified', bench.sa_poly64_torch(dSA) == g['sa_poly64'], flush=True)
    del dT
