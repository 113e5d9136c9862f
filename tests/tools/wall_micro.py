import ctypes, sys, time
sys.path.insert(0, '.')
import torch
from pysubstringsearch_amd import _ffi
g = torch.Generator(device='cuda'); g.manual_seed(1)
ms = ctypes.c_double()
for logn in range(20, 26):
    n = 1 << logn
    for bits in (8, 32):
        keys = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device='cuda', generator=g)
        vals = torch.arange(n, dtype=torch.int32, device='cuda')
        best = 1e9; bw = 1e9
        for _ in range(6):
            k = keys.clone(); v = vals.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _ffi.check(_ffi.lib.pss_sort_pairs_device(k.data_ptr(), v.data_ptr(), n, bits, 0, None))
            t1 = time.perf_counter()
            bw = min(bw, (t1 - t0) * 1e3)
            _ffi.check(_ffi.lib.pss_sort_pairs_device(k.data_ptr(), v.data_ptr(), n, bits, 0, ctypes.byref(ms)))
            best = min(best, ms.value)
        p = bits // 8
        print(f'n=2^{logn} passes={p}: scatter-only {best*1e3:.1f} us, wall {bw*1e3:.1f} us', flush=True)
