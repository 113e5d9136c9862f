"""Where does a hit-heavy batch spend its time? (C call vs result marshalling)"""
import ctypes, sys, time, os
import numpy as np
sys.path.insert(0, '.')
import torch
from pysubstringsearch_amd import Reader, _ffi, _pssglue
lib = _ffi.lib
n = 1 << 29
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4
h = ctypes.c_void_p(); _ffi.check(lib.pss_reader_create(0, ctypes.byref(h)))
r = Reader._from_handle(h)
dSA = torch.empty(n, dtype=torch.int32, device='cuda')
rng = np.random.default_rng(1)
qs = []
for c in range(C):
    host = np.empty(n, dtype=np.uint8); lib.pss_gen_corpus(0, host.ctypes.data, n, c)
    dT = torch.from_numpy(host).cuda()
    _ffi.check(lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, 0, None))
    _ffi.check(lib.pss_reader_add_chunk_device(h, dT.data_ptr(), dSA.data_ptr(), n))
    for _ in range(25000 // C):
        s = int(rng.integers(0, n - 40)); ln = int(rng.integers(4, 33)); q = host[s:s+ln].tobytes()
        if b'\n' not in q: qs.append(q)
for rep in range(3):
    t0 = time.perf_counter()
    nq = len(qs); blob = b''.join(qs)
    lens = np.fromiter(map(len, qs), dtype=np.uint64, count=nq)
    offs = np.zeros(nq + 1, dtype=np.uint64); np.cumsum(lens, out=offs[1:])
    t1 = time.perf_counter()
    res = ctypes.c_void_p()
    _ffi.check(lib.pss_reader_search_batch(h, blob, offs.ctypes.data, nq, ctypes.byref(res)))
    t2 = time.perf_counter()
    ne = lib.pss_result_num_entries(res)
    off = lib.pss_result_offsets(res); base = lib.pss_result_bytes(res)
    ents = _pssglue.entries_to_list(ctypes.cast(base, ctypes.c_void_p).value, ctypes.cast(off, ctypes.c_void_p).value, ne, True)
    t3 = time.perf_counter()
    lib.pss_result_free(res)
    t4 = time.perf_counter()
    del ents
    t5 = time.perf_counter()
    st = r.last_stats()
    print(f'pack {1e3*(t1-t0):.1f} ms | C call {1e3*(t2-t1):.1f} ms (device {st["ms_device"]:.1f}) | list {1e3*(t3-t2):.1f} ms | free {1e3*(t4-t3):.1f} | del {1e3*(t5-t4):.1f} | entries {ne} bytes {st["result_bytes"]}')
