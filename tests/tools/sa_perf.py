"""Exploration: time pss_sa_build_device on synthetic corpora (device-resident)."""
import ctypes
import hashlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from pysubstringsearch_amd import _ffi  # noqa: E402

KINDS = {'lines': 0, 'words': 1, 'runs': 2, 'periodic': 3, 'repeat_line': 4, 'dup_blocks': 5, 'mixed': 6, 'source': 7}


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else 'lines'
    logn = int(sys.argv[2]) if len(sys.argv) > 2 else 26
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    check = len(sys.argv) > 4 and sys.argv[4] == 'check'
    n = 1 << logn
    host = np.empty(n, dtype=np.uint8)
    t0 = time.time()
    _ffi.check(_ffi.lib.pss_gen_corpus(KINDS[kind], host.ctypes.data, n, 0))
    print(f'gen {kind} n=2^{logn}: {time.time() - t0:.2f}s sha256={hashlib.sha256(host.tobytes()).hexdigest()[:16]}')
    dT = torch.from_numpy(host).cuda()
    dSA = torch.empty(n, dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    st = _ffi.SaStats()
    for r in range(reps):
        t0 = time.time()
        rc = _ffi.lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, 1 if (r == reps - 1 or os.environ.get('PSS_PROFILE_ALL')) else 0, ctypes.byref(st))
        _ffi.check(rc)
        wall = time.time() - t0
        d = st.as_dict()
        print(f'rep {r}: wall {wall * 1e3:.1f} ms  dev {d["ms_total"]:.1f} ms  -> {n / d["ms_total"] / 1e6:.3f} GB/s  {d}')
    if check:
        sa = dSA.cpu().numpy()
        print('SA sha256', hashlib.sha256(sa.tobytes()).hexdigest())
        from oracle import oracle as O
        t0 = time.time()
        exp = O.sa(host)
        print(f'oracle {time.time() - t0:.1f}s equal={bool((sa == exp).all())}')


if __name__ == '__main__':
    main()
