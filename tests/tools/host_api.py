"""PCIe-inclusive rate of the libsais-compatible host entry point pss_sa_build (host text in, host SA out)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from pysubstringsearch_amd import _ffi
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 29)
t = np.empty(n, dtype=np.uint8); _ffi.lib.pss_gen_corpus(0, t.ctypes.data, n, 0)
sa = np.empty(n, dtype=np.int32)
for rep in range(3):
    t0 = time.perf_counter(); _ffi.check(_ffi.lib.pss_sa_build(t.ctypes.data, sa.ctypes.data, n, 0)); dt = time.perf_counter() - t0
    print(f'pss_sa_build host->host n=2^{n.bit_length()-1}: {dt*1e3:.1f} ms -> {n/dt/1e9:.2f} GB/s')
