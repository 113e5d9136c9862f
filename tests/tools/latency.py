"""Single-query latency breakdown on one resident 2^logn chunk."""
import ctypes, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from pysubstringsearch_amd import Reader, _ffi
lib = _ffi.lib
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 29
nchunks = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = 1 << logn
h = ctypes.c_void_p(); _ffi.check(lib.pss_reader_create(0, ctypes.byref(h)))
r = Reader._from_handle(h)
dSA = torch.empty(n, dtype=torch.int32, device='cuda')
for c in range(nchunks):
    host = np.empty(n, dtype=np.uint8); lib.pss_gen_corpus(0, host.ctypes.data, n, c)
    dT = torch.from_numpy(host).cuda()
    _ffi.check(lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, 0, None))
    _ffi.check(lib.pss_reader_add_chunk_device(h, dT.data_ptr(), dSA.data_ptr(), n))
if len(sys.argv) > 3 and sys.argv[3] == 'resident':
    r.set_low_latency(True)       # single queries through the resident kernel (include/pss.h)
hit = host[1000:1008].tobytes().replace(b'\n', b'a')
for name, q in [('miss', 'zzzzqqqq'), ('hit8', host[5000:5008].tobytes().decode()), ('hit4 (many)', host[7000:7004].tobytes().decode())]:
    if '\n' in q: q = q.replace('\n', 'a')
    for _ in range(20): r.search(q)
    ts = []; dev = []; hst = []
    for _ in range(300):
        t0 = time.perf_counter(); res = r.search(q); ts.append(time.perf_counter() - t0); ls = r.last_stats(); dev.append(ls['ms_device']); hst.append(ls['ms_host'])
    ts.sort(); dev.sort(); hst.sort()
    if len(sys.argv) > 3 and sys.argv[3] == 'resident':
        name += ' ' + str(r.low_latency_stats())
    print(f'{name:12s} results={len(res):6d}  wall median {ts[150]*1e6:7.1f} us  p90 {ts[270]*1e6:7.1f} us | inside the library {hst[150]*1e3:7.1f} us | device(events) median {dev[150]*1e3:7.1f} us')
