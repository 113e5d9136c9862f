"""Where a resident-kernel query spends its time (diagnostic build of search.hip, -DPSS_TRACE_RESIDENT):

    make -C pysubstringsearch_amd/csrc trace
    PSS_LIBPSS=pysubstringsearch_amd/libpss_trace.so PSS_TRACE_RESIDENT=1 python tests/tools/latency_trace.py [logn=29]
"""
import ctypes, sys
import numpy as np
sys.path.insert(0, '.')
import torch
from pysubstringsearch_amd import Reader, _ffi
lib = _ffi.lib
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 29
n = 1 << logn
h = ctypes.c_void_p(); _ffi.check(lib.pss_reader_create(0, ctypes.byref(h)))
r = Reader._from_handle(h)
dSA = torch.empty(n, dtype=torch.int32, device='cuda')
host = np.empty(n, dtype=np.uint8); lib.pss_gen_corpus(0, host.ctypes.data, n, 0)
dT = torch.from_numpy(host).cuda()
_ffi.check(lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, 0, None))
_ffi.check(lib.pss_reader_add_chunk_device(h, dT.data_ptr(), dSA.data_ptr(), n))
r.set_low_latency(True)
base = host[70000:70008].tobytes().replace(b'\n', b'a').decode()
for q in (base[:8], base[:5], base[:4], 'zzzzqqqq'):
    for _ in range(5):
        res = r.search(q)
    print(f'--- {len(q)} bytes, {len(res)} results', file=sys.stderr, flush=True)
    for _ in range(6):
        r.search(q)
