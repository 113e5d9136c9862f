"""Parity + timing on real text: concatenated Python sources of the image (long repeated
license headers / boilerplate => long LCPs, the case the synthetic corpora lack)."""
import ctypes, glob, hashlib, os, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from pysubstringsearch_amd import _ffi
target = int(sys.argv[1]) if len(sys.argv) > 1 else 64 << 20
parts = []; size = 0
for root in ('/usr/lib/python3/dist-packages', '/usr/local/lib/python3.10/dist-packages', '/usr/lib/python3.10'):
    for f in sorted(glob.glob(root + '/**/*.py', recursive=True)):
        try:
            b = open(f, 'rb').read()
        except OSError:
            continue
        parts.append(b); size += len(b)
        if size >= target: break
    if size >= target: break
data = b''.join(parts)[:target]
host = np.frombuffer(data, dtype=np.uint8).copy(); n = host.size
print('bytes', n, 'distinct symbols', len(set(data[:1 << 20])))
dT = torch.from_numpy(host).cuda(); dSA = torch.empty(n, dtype=torch.int32, device='cuda')
st = _ffi.SaStats()
for rep in range(2):
    _ffi.check(_ffi.lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, 0, ctypes.byref(st)))
d = st.as_dict()
print(f'GPU build {d["ms_total"]:.1f} ms -> {n / d["ms_total"] / 1e6:.2f} GB/s', {k: d[k] for k in ('key_chars', 'initial_passes', 'rounds', 'text_rounds', 'round_passes', 'sum_active', 'big_elems', 'mode')})
from oracle import oracle as O
t0 = time.time(); exp = O.sa(host); t1 = time.time()
got = dSA.cpu().numpy()
print(f'libsais {t1 - t0:.1f} s -> {n / (t1 - t0) / 1e9:.4f} GB/s; equal = {bool((got == exp).all())}')
