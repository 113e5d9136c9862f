"""BASELINE.json configs[4] at full chunk size: SA build of the `runs` and `periodic`
corpora, checked exactly.

  periodic  closed form: text = ("a"*4095 + "\n") * K, so the suffix at k*4096 + r
            sorts by r descending ('\n' < 'a'), then k descending (the shorter of two
            otherwise identical suffixes first).
  runs      against libsais (oracle/_ref, one host thread) -- about a minute at 2^29.

    python tests/tools/adversarial.py [logn=29] [--no-libsais]
"""
import ctypes
import json
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import torch  # noqa: E402

from pysubstringsearch_amd import _ffi  # noqa: E402

KEYS = ('sigma', 'key_chars', 'initial_passes', 'rounds', 'text_rounds', 'round_passes', 'sum_active', 'big_elems', 'mode')


def build(kind, n, reps=2):
    host = np.empty(n, dtype=np.uint8)
    _ffi.check(_ffi.lib.pss_gen_corpus(kind, host.ctypes.data, n, 0))
    dT = torch.from_numpy(host).cuda()
    dSA = torch.empty(n, dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    st = _ffi.SaStats()
    best = None
    for _ in range(reps):
        _ffi.check(_ffi.lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, 0, ctypes.byref(st)))
        best = st.ms_total if best is None else min(best, st.ms_total)
    d = st.as_dict()
    return host, dSA, {'build_ms': round(best, 2), 'index_build_gbs': round(n / best / 1e6, 4), **{k: d[k] for k in KEYS}}


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else 29
    n = 1 << logn
    out = {'chunk_bytes': n}

    host, dSA, info = build(3, n)
    K = n // 4096
    r = torch.arange(4095, -1, -1, dtype=torch.int64, device='cuda').repeat_interleave(K)
    k = torch.arange(K - 1, -1, -1, dtype=torch.int64, device='cuda').repeat(4096)
    info['exact'] = bool(torch.equal(dSA.to(torch.int64), k * 4096 + r))
    out['periodic'] = info
    del r, k, dSA

    host, dSA, info = build(2, n)
    if '--no-libsais' not in sys.argv:
        from oracle import oracle as O
        t0 = time.time()
        exp = O.sa_reference(host) if O.have_reference() else O.sa_restatement(host)
        info['libsais_s'] = round(time.time() - t0, 1)
        info['exact'] = bool((dSA.cpu().numpy() == exp).all())
    out['runs'] = info
    print(json.dumps(out))


if __name__ == '__main__':
    main()
