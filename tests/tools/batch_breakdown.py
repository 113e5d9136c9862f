"""Host-side breakdown of one 10k x 8-byte batch on a 512 MiB chunk (bench config)."""
import ctypes, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from pysubstringsearch_amd import Reader, _ffi, _pssglue
from bench import make_queries
lib = _ffi.lib
n = 1 << 29
host = np.empty(n, dtype=np.uint8); lib.pss_gen_corpus(0, host.ctypes.data, n, 0)
dT = torch.from_numpy(host).cuda(); dSA = torch.empty(n, dtype=torch.int32, device='cuda')
_ffi.check(lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, 0, None))
h = ctypes.c_void_p(); _ffi.check(lib.pss_reader_create(0, ctypes.byref(h)))
r = Reader._from_handle(h)
_ffi.check(lib.pss_reader_add_chunk_device(h, dT.data_ptr(), dSA.data_ptr(), n))
qs = make_queries(host, 10000, 8)
for rep in range(4):
    t0 = time.perf_counter()
    nq = len(qs); blob = b''.join(qs)
    offs = (ctypes.c_uint64 * (nq + 1))()
    view = np.ctypeslib.as_array(offs)
    np.cumsum(np.fromiter(map(len, qs), dtype=np.uint64, count=nq), out=view[1:])
    t1 = time.perf_counter()
    res = ctypes.c_void_p()
    _ffi.check(lib.pss_reader_search_batch(h, blob, offs, nq, ctypes.byref(res)))
    t2 = time.perf_counter()
    ne = lib.pss_result_num_entries(res)
    counts = list(lib.pss_result_query_counts(res)[:nq])
    t3 = time.perf_counter()
    off = lib.pss_result_offsets(res); base = lib.pss_result_bytes(res)
    ents = _pssglue.entries_to_list(ctypes.cast(base, ctypes.c_void_p).value, ctypes.cast(off, ctypes.c_void_p).value, ne, False)
    t4 = time.perf_counter()
    lib.pss_result_free(res)
    t5 = time.perf_counter()
    st = r.last_stats()
    print(f'pack {1e3*(t1-t0):.3f} | C call {1e3*(t2-t1):.3f} (device {st["ms_device"]:.3f}, interval {st["ms_interval"]:.3f}) | counts list {1e3*(t3-t2):.3f} | entries {1e3*(t4-t3):.3f} | free {1e3*(t5-t4):.3f} ms')
t0 = time.perf_counter(); e, c = r.search_batch_raw(qs); print('search_batch_raw total', 1e3 * (time.perf_counter() - t0))
