import ctypes, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from pysubstringsearch_amd import Reader, _ffi
from bench import make_queries
lib = _ffi.lib
n = 1 << 29
host = np.empty(n, dtype=np.uint8); lib.pss_gen_corpus(0, host.ctypes.data, n, 0)
dT = torch.from_numpy(host).cuda(); dSA = torch.empty(n, dtype=torch.int32, device='cuda')
qs = make_queries(host, 10000, 8)
for rep in range(4):
    t0 = time.perf_counter()
    _ffi.check(lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, 0, None))
    t1 = time.perf_counter()
    h = ctypes.c_void_p(); _ffi.check(lib.pss_reader_create(0, ctypes.byref(h)))
    r = Reader._from_handle(h)
    t2 = time.perf_counter()
    _ffi.check(lib.pss_reader_add_chunk_device(h, dT.data_ptr(), dSA.data_ptr(), n))
    t3 = time.perf_counter()
    e, c = r.search_batch_raw(qs)
    t4 = time.perf_counter()
    r.close()
    t5 = time.perf_counter()
    print(f'build {1e3*(t1-t0):.1f} | create {1e3*(t2-t1):.2f} | add_chunk {1e3*(t3-t2):.1f} | search {1e3*(t4-t3):.2f} | close {1e3*(t5-t4):.1f} ms')
