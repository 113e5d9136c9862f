"""Python-side cost of Writer.add_entry (the reference pays one pyo3 call per entry)."""
import os, sys, tempfile, time
sys.path.insert(0, '.')
import pysubstringsearch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
entries = ['entry number %d with some text in it' % i for i in range(n)]
with tempfile.TemporaryDirectory() as d:
    w = pysubstringsearch.Writer(os.path.join(d, 'a.idx'))
    t0 = time.perf_counter()
    for e in entries:
        w.add_entry(e)
    t1 = time.perf_counter()
    w.finalize()
    t2 = time.perf_counter()
    w.close()
print(f'{n} add_entry calls: {1e9 * (t1 - t0) / n:.0f} ns per call ({n / (t1 - t0) / 1e6:.2f} M entries/s); finalize {t2 - t1:.2f} s')
