"""Sort microbenchmark: does the scatter kernel care about run alignment?  Random 8-bit digits vs
keys whose every 4096-tile holds exactly 16 of each digit value (all runs 128 B, line aligned)."""
import ctypes, sys
import numpy as np
sys.path.insert(0, '.')
import torch
from pysubstringsearch_amd import _ffi
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 28)
g = torch.Generator(device='cuda'); g.manual_seed(1)
ms = ctypes.c_double()
def run(name, keys):
    vals = torch.arange(n, dtype=torch.int32, device='cuda')
    best = 1e9
    for _ in range(4):
        k = keys.clone(); v = vals.clone()
        torch.cuda.synchronize()   # the engine runs on its own stream: torch's copies must have landed
        _ffi.check(_ffi.lib.pss_sort_pairs_device(k.data_ptr(), v.data_ptr(), n, 8, 0, ctypes.byref(ms)))
        best = min(best, ms.value)
    print(f'{name:34s} one 8-bit pass over {n} pairs: scatter {best:.3f} ms -> {24*n/best/1e6:.0f} GB/s')
rnd = torch.randint(0, 256, (n,), dtype=torch.int64, device='cuda', generator=g)
run('uniform random digits', rnd)
# balanced: within each 4096 tile digit = (permuted position) % 256 -> exactly 16 each
perm = torch.randperm(4096, device='cuda', generator=g)
bal = (perm.repeat(n // 4096) % 256).to(torch.int64)
run('balanced per tile (aligned runs)', bal)
run('all equal digit (one bucket)', torch.zeros(n, dtype=torch.int64, device='cuda'))
run('sorted digits (contiguous)', (torch.arange(n, device='cuda') // (n // 256)).to(torch.int64))
