"""A/B of the periodic keys of the rank rounds (PSS_PERIODIC=0|1) on `words` text with periodic stretches.

    python tests/tools/per_perf.py [logn=29] [period=60] [stretches=1]
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, '.')
import torch  # noqa: E402

from pysubstringsearch_amd import _ffi  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 29
period = int(sys.argv[2]) if len(sys.argv) > 2 else 60
stretches = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n = 1 << logn
t = np.empty(n, dtype=np.uint8)
_ffi.check(_ffi.lib.pss_gen_corpus(1, t.ctypes.data, n, 0))
line = t[1000:1000 + period - 1].copy()
line[line == 10] = 97
line = np.concatenate([line, np.array([10], np.uint8)])
third = n // 3
seg = third // stretches
for k in range(stretches):
    a = third + k * seg
    L = seg - seg // 4
    t[a:a + L] = np.resize(line, L)
dT = torch.from_numpy(t).cuda()
dSA = torch.empty(n, dtype=torch.int32, device='cuda')
res = {}
for per in ('1', '0', '1', '0'):
    os.environ['PSS_PERIODIC'] = per
    st = _ffi.SaStats()
    best = None
    for _ in range(2):
        _ffi.check(_ffi.lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, 0, ctypes.byref(st)))
        best = st.ms_total if best is None else min(best, st.ms_total)
    d = st.as_dict()
    import hashlib
    h = hashlib.sha256(dSA.cpu().numpy().tobytes()).hexdigest()[:16]
    res.setdefault(per, []).append(h)
    print(f'PSS_PERIODIC={per}: {best:.1f} ms  anchor_ms={d["anchor_ms"]:.1f} anchor_rounds={d["anchor_rounds"]} levels={d["anchor_levels"]} '
          f'periodic={d["periodic_rounds"]}/{d["periodic_members"]} rounds={d["rounds"]} sha={h}', flush=True)
assert len({x for v in res.values() for x in v}) == 1, res
print('same suffix array with and without')
