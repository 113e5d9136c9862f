"""End-to-end through the file API: text file -> Writer.add_entries_from_file_lines -> .idx -> Reader.
Reports ingest+build+write GB/s (text bytes / wall) and Reader cold-start GB/s (.idx bytes / wall)."""
import os, sys, time, tempfile, hashlib
import numpy as np
sys.path.insert(0, '.')
import pysubstringsearch
from pysubstringsearch_amd import _ffi

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 28
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = 1 << logn
d = tempfile.mkdtemp(dir=os.environ.get('PSS_TMP', '/tmp'))
src = os.path.join(d, 'corpus.txt'); idx = os.path.join(d, 'out.idx')
with open(src, 'wb') as f:
    for c in range(chunks):
        buf = np.empty(n, dtype=np.uint8); _ffi.lib.pss_gen_corpus(0, buf.ctypes.data, n, c); f.write(buf.tobytes())
import json
rows = []
devs_list = [None, [0, 0]] if len(sys.argv) > 3 else [None]     # third argument: also the two-lane (virtual device) pipeline
for rep in range(2 * len(devs_list)):
    devs = devs_list[rep // 2]
    t0 = time.perf_counter()
    w = pysubstringsearch.Writer(idx, n) if devs is None else pysubstringsearch.Writer(idx, n, devices=devs)
    w.add_entries_from_file_lines(src)
    w.finalize(); w.close()
    t1 = time.perf_counter()
    r = pysubstringsearch.Reader(idx)
    t2 = time.perf_counter()
    got = r.search('abc12')
    t3 = time.perf_counter()
    sz = os.path.getsize(idx)
    print(f'rep {rep}: write {n*chunks/ (t1-t0)/1e9:.3f} GB/s text ({t1-t0:.2f}s, idx {sz/1e9:.2f} GB -> {sz/(t1-t0)/1e9:.2f} GB/s file) | '
          f'reader open {sz/(t2-t1)/1e9:.2f} GB/s ({t2-t1:.2f}s) chunks={r.num_chunks} | first search {1e3*(t3-t2):.2f} ms hits={len(got)}')
    rows.append({'devices': devs, 'text_gbs': round(n * chunks / (t1 - t0) / 1e9, 3), 'file_gbs': round(sz / (t1 - t0) / 1e9, 2),
                 'reader_open_gbs': round(sz / (t2 - t1) / 1e9, 2), 'chunks': r.num_chunks})
    r.close()
print(json.dumps({'chunk_bytes': n, 'chunks': chunks, 'tmp': d, 'runs': rows}))
os.remove(src); os.remove(idx); os.rmdir(d)
