"""What the in-process multi-device reader costs on top of the search itself: 15 chunks of `lines` (2^logn bytes each) opened
on ONE device and on eight virtual ones ([0] * 8: the parts take turns on the GPU, so the device work is the same and the
difference is the host side -- the merge of the parts' results).   python tests/tools/multi_merge_perf.py [logn=26] [queries=100000]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import bench  # noqa: E402
import pysubstringsearch  # noqa: E402
from pysubstringsearch_amd import _ffi  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 26
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
n = 1 << logn
d = '/dev/shm' if os.path.isdir('/dev/shm') else '/tmp'
src, idx = os.path.join(d, 'mm_src.txt'), os.path.join(d, 'mm.idx')
texts = []
with open(src, 'wb') as f:
    for c in range(15):
        t = np.empty(n, dtype=np.uint8)
        _ffi.check(_ffi.lib.pss_gen_corpus(0, t.ctypes.data, n, c))
        texts.append(t)
        f.write(t.tobytes())
t0 = time.time()
w = pysubstringsearch.Writer(idx, n)
w.add_entries_from_file_lines(src)
w.close()
print(f'index of 15 x {n >> 20} MiB written in {time.time() - t0:.1f} s')
per = (nq // 2 + 14) // 15
sampled = [q for c, t in enumerate(texts) for q in bench.sample_chunk_queries(t, c, per, 4, 32)]
queries = bench.mixed_queries(sampled, nq, 4, 32)
for devs in ([0], [0] * 2, [0] * 8):
    with pysubstringsearch.Reader(idx, devices=devs) as r:
        best = None
        for _ in range(4):
            t0 = time.perf_counter()
            pk = r.search_batch_packed(queries)
            dt = (time.perf_counter() - t0) * 1e3
            ne, nb = len(pk.offsets) - 1, int(pk.offsets[-1])
            del pk
            best = dt if best is None else min(best, dt)
        st = r.last_stats()
        print(f'devices={len(devs)}: chunks {r.num_chunks} per device {r.chunks_per_device}; batch of {nq}: {best:.1f} ms = {nq / best / 1e3:.2f} M q/s; '
              f'{ne} entries, {nb / 1e6:.0f} MB; device {st["ms_device"]:.1f} ms, host {st["ms_host"]:.1f} ms')
os.remove(src)
os.remove(idx)
