"""Single-query latency against the number of results (one resident 2^logn `lines` chunk, or C of them):
queries are prefixes of 2..8 bytes taken from the text, so the result count falls by ~38x per byte.
The reference's README quotes 14.9 us (159 results), 497 us (5 943) on a 500 MB index and 10.1 ms (62 834)
on 15 chunks (README.md:48-59).

    python tests/tools/latency_hits.py [logn] [chunks] [resident]

`resident`: the reader in low-latency mode (Reader.set_low_latency: resident search kernel, no launch per query).
"""
import ctypes, json, sys, time
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
resident = len(sys.argv) > 3 and sys.argv[3] == 'resident'
if resident:
    r.set_low_latency(True)
base = host[70000:70008].tobytes().replace(b'\n', b'a').decode()
out = []
for ln in (8, 5, 4, 3, 2):
    q = base[:ln]
    for _ in range(10): res = r.search(q)
    reps = 200 if len(res) < 100000 else 30
    ts, dev, hst, pk = [], [], [], []
    for _ in range(reps):
        t0 = time.perf_counter(); res = r.search(q); ts.append(time.perf_counter() - t0)
        ls = r.last_stats(); dev.append(ls['ms_device']); hst.append(ls['ms_host'])
        t0 = time.perf_counter(); p = r.search_batch_packed([q.encode()]); pk.append(time.perf_counter() - t0)
    for a in (ts, dev, hst, pk): a.sort()
    m = reps // 2
    row = {'query_bytes': ln, 'results': len(res), 'list_us': round(ts[m] * 1e6, 1), 'packed_us': round(pk[m] * 1e6, 1),
           'library_us': round(hst[m] * 1e3, 1), 'device_us': round(dev[m] * 1e3, 1)}
    out.append(row)
    print(row, flush=True)
print(json.dumps({'logn': logn, 'chunks': nchunks, 'low_latency_mode': resident,
                  'low_latency_stats': r.low_latency_stats() if resident else None, 'rows': out}))
