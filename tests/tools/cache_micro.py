"""Does a scatter pass run faster when its ping-pong buffers fit the 256 MiB Infinity Cache?
One to four 8-bit passes of the generic (u64, u32) pair sort over 2^20 .. 2^28 random pairs;
prints ns per element and pass (scatter kernels only, HIP events inside the engine)."""
import ctypes, sys
sys.path.insert(0, '.')
import torch
from pysubstringsearch_amd import _ffi
g = torch.Generator(device='cuda'); g.manual_seed(1)
ms = ctypes.c_double()
for logn in range(20, 29):
    n = 1 << logn
    for bits in (8, 32):
        keys = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device='cuda', generator=g)
        vals = torch.arange(n, dtype=torch.int32, device='cuda')
        best = 1e9
        for _ in range(5):
            k = keys.clone(); v = vals.clone()
            torch.cuda.synchronize()
            _ffi.check(_ffi.lib.pss_sort_pairs_device(k.data_ptr(), v.data_ptr(), n, bits, 0, ctypes.byref(ms)))
            best = min(best, ms.value)
        p = bits // 8
        print(f'n=2^{logn} passes={p}: {best:.4f} ms  {best*1e6/n/p:.4f} ns/elem/pass  {24*n*p/best/1e6:.0f} GB/s', flush=True)
