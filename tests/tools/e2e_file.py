"""End-to-end through the file API: text file -> Writer.add_entries_from_file_lines -> .idx -> Reader.
Reports ingest + build + write GB/s (text bytes / wall) and Reader cold-start GB/s (.idx bytes / wall).

    python tests/tools/e2e_file.py [logn=28] [chunks=2] [multi] [dir=/dev/shm] [check]

`multi`: also the two-lane (virtual device) pipeline; `check`: the .idx is compared with a single-lane one by sha256.
"""
import hashlib
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, '.')
import pysubstringsearch  # noqa: E402
from pysubstringsearch_amd import _ffi  # noqa: E402


def sha(path):
    h = hashlib.sha256()
    with open(path, 'rb') as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    multi = 'multi' in sys.argv[3:]
    check = 'check' in sys.argv[3:]
    where = next((a[4:] for a in sys.argv[3:] if a.startswith('dir=')), os.environ.get('PSS_TMP', '/tmp'))
    n = 1 << logn
    d = tempfile.mkdtemp(dir=where)
    src = os.path.join(d, 'corpus.txt')
    idx = os.path.join(d, 'out.idx')
    with open(src, 'wb') as f:
        for c in range(chunks):
            buf = np.empty(n, dtype=np.uint8)
            _ffi.lib.pss_gen_corpus(0, buf.ctypes.data, n, c)
            f.write(buf.tobytes())
    rows = []
    first_sha = None
    devs_list = [None, [0, 0]] if multi else [None]
    for rep in range(2 * len(devs_list)):
        devs = devs_list[rep // 2]
        if os.path.exists(idx):
            os.remove(idx)
        t0 = time.perf_counter()
        w = pysubstringsearch.Writer(idx, n) if devs is None else pysubstringsearch.Writer(idx, n, devices=devs)
        w.add_entries_from_file_lines(src)
        w.finalize()
        w.close()
        t1 = time.perf_counter()
        r = pysubstringsearch.Reader(idx)
        t2 = time.perf_counter()
        got = r.search('abc12')
        t3 = time.perf_counter()
        sz = os.path.getsize(idx)
        row = {'devices': devs, 'dir': where, 'text_gbs': round(n * chunks / (t1 - t0) / 1e9, 3), 'write_s': round(t1 - t0, 3),
               'file_gbs': round(sz / (t1 - t0) / 1e9, 2), 'reader_open_gbs': round(sz / (t2 - t1) / 1e9, 2),
               'reader_open_s': round(t2 - t1, 3), 'chunks': r.num_chunks, 'first_search_ms': round(1e3 * (t3 - t2), 2),
               'hits': len(got)}
        if check:
            h = sha(idx)
            first_sha = first_sha or h
            row['same_bytes_as_first'] = h == first_sha
        print(f'rep {rep}: {row}', flush=True)
        rows.append(row)
        r.close()
    print(json.dumps({'chunk_bytes': n, 'chunks': chunks, 'runs': rows}))
    os.remove(src)
    os.remove(idx)
    os.rmdir(d)


if __name__ == '__main__':
    main()
