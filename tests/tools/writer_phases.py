"""Where a one-chunk Writer spends its wall clock: add_entries_from_file_lines vs finalize, for generated `lines` text and
for real files.   python tests/tools/writer_phases.py [logn=29]"""
import importlib.util
import os
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import pysubstringsearch_amd as pss  # noqa: E402
from pysubstringsearch_amd import _ffi  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 29
spec = importlib.util.spec_from_file_location('real_text', os.path.join(os.path.dirname(__file__), 'real_text.py'))
rt = importlib.util.module_from_spec(spec)
spec.loader.exec_module(rt)
n = (1 << logn) - 4096
t = np.empty(n, dtype=np.uint8)
_ffi.check(_ffi.lib.pss_gen_corpus(0, t.ctypes.data, n, 0))
open('/tmp/wp_lines.txt', 'wb').write(t.tobytes())
open('/tmp/wp_real.txt', 'wb').write(rt.collect(1 << logn))
for name in ('lines', 'real', 'lines', 'real'):
    src = f'/tmp/wp_{name}.txt'
    size = os.path.getsize(src)
    t0 = time.perf_counter()
    w = pss.Writer('/tmp/wp.idx')
    t1 = time.perf_counter()
    w.add_entries_from_file_lines(src)
    t2 = time.perf_counter()
    w.finalize()
    t3 = time.perf_counter()
    w.close()
    t4 = time.perf_counter()
    print(f'{name}: {size} bytes  open {1e3 * (t1 - t0):.0f} ms  add_entries_from_file_lines {1e3 * (t2 - t1):.0f} ms  finalize {1e3 * (t3 - t2):.0f} ms  '
          f'close {1e3 * (t4 - t3):.0f} ms  total {1e3 * (t4 - t0):.0f} ms = {size / (t4 - t0) / 1e9:.2f} GB/s of text', flush=True)
for f in ('/tmp/wp_lines.txt', '/tmp/wp_real.txt', '/tmp/wp.idx'):
    os.remove(f)
