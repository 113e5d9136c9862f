"""Size-independent checks of a device-built SA at large n: permutation + sampled sortedness."""
import ctypes, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from pysubstringsearch_amd import _ffi
kind = {'lines': 0, 'words': 1, 'runs': 2, 'periodic': 3}[sys.argv[1]]
n = int(sys.argv[2]) if not sys.argv[2].startswith('2^') else 1 << int(sys.argv[2][2:])
host = np.empty(n, dtype=np.uint8); _ffi.lib.pss_gen_corpus(kind, host.ctypes.data, n, 0)
dT = torch.from_numpy(host).cuda(); dSA = torch.empty(n, dtype=torch.int32, device='cuda')
st = _ffi.SaStats()
t0 = time.time(); _ffi.check(_ffi.lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, 0, ctypes.byref(st))); t1 = time.time()
print(f'n={n} build {1e3*(t1-t0):.1f} ms  {n/(t1-t0)/1e9:.2f} GB/s', {k: getattr(st, k) for k in ('key_chars', 'initial_passes', 'rounds', 'round_passes', 'sum_active', 'mode', 'text_rounds', 'big_elems', 'msd', 'msd_max_bucket', 'msd_buckets')})
sa = dSA.to(torch.int64)
cnt = torch.zeros(n, dtype=torch.int8, device='cuda'); cnt[sa] = 1
assert int(cnt.sum().item()) == n and int(sa.min()) == 0 and int(sa.max()) == n - 1, 'not a permutation'
sa_h = dSA.cpu().numpy(); text = host.tobytes()
rng = np.random.default_rng(0)
for j in rng.integers(1, n, 20000):
    a, b = int(sa_h[j - 1]), int(sa_h[j])
    x, y = text[a:a + 256], text[b:b + 256]
    assert x < y or (x == y and len(x) == 256), (j, a, b)
print('permutation + 20000 sampled adjacent pairs ordered: ok')
