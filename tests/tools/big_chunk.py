"""Container format 2 beyond the reference format's limit: ONE chunk of more than 1 GiB of text
(default 1.25 GiB) through Writer(format_version=2) -> .idx -> Reader, checked by brute force:
every sampled query must return exactly the lines of the text that contain it.

    python tests/tools/big_chunk.py [bytes=1342177280]"""
import pathlib
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, '.')
import pysubstringsearch  # noqa: E402
from pysubstringsearch_amd import _ffi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5 << 28
d = tempfile.mkdtemp(dir=os.environ.get('PSS_TMP', '/tmp'))
src, idx = os.path.join(d, 'corpus.txt'), os.path.join(d, 'out.idx')
buf = np.empty(n, dtype=np.uint8)
_ffi.check(_ffi.lib.pss_gen_corpus(0, buf.ctypes.data, n, 7))
text = buf.tobytes()
del buf
pathlib.Path(src).write_bytes(text)
t0 = time.perf_counter()
w = pysubstringsearch.Writer(idx, n, format_version=2)
w.add_entries_from_file_lines(src)
w.close()
t1 = time.perf_counter()
size = os.path.getsize(idx)
assert size == 16 + 16 + 5 * n, (size, n)
r = pysubstringsearch.Reader(idx)
t2 = time.perf_counter()
assert r.num_chunks == 1
rng = np.random.default_rng(3)
checked = 0
for qlen in (5, 8, 12):
    for _ in range(8):
        s = int(rng.integers(0, n - qlen))
        q = text[s:s + qlen]
        if b'\n' in q:
            continue
        starts = set()
        pos = text.find(q)
        while pos >= 0:
            starts.add(text.rfind(b'\n', 0, pos) + 1)
            pos = text.find(q, pos + 1)
        want = sorted(text[a:text.find(b'\n', a)] for a in starts)
        got = sorted(e.encode() for e in r.search(q.decode()))
        assert got == want, (q, len(got), len(want))
        checked += 1
r.close()
print(f'one chunk of {n} bytes: write {t1 - t0:.1f} s ({n / (t1 - t0) / 1e9:.2f} GB/s of text), open {t2 - t1:.1f} s, '
      f'{checked} queries equal to brute force; .idx {size} bytes')
os.remove(src)
os.remove(idx)
os.rmdir(d)
