"""Time-boxed fuzzing of the whole path against the oracle (libsais + the CPU restatement of the
reference's Writer / Reader): random alphabets, repeat structure, chunk limits and queries.

    python tests/tools/fuzz.py [seconds=120] [seed0=<time>]

Every case checks (1) the suffix array of the raw text, (2) the .idx container byte for byte,
(3) search / search_multiple multisets and per-query counts.  Prints the failing seed and stops."""
import pathlib
import os
import random
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import pysubstringsearch  # noqa: E402
from oracle import oracle as O  # noqa: E402
from util import sa_gpu  # noqa: E402


def make_text(rng):
    sigma = rng.choice([1, 2, 3, 4, 16, 40, 100, 255, 256])
    big = os.environ.get('FUZZ_BIG')      # SA only, sizes that reach the sampled sizing (n >= 2^24)
    n = int(2 ** (rng.uniform(21, 25.3) if big else rng.uniform(0, 22 if rng.random() < 0.05 else 18)))
    kind = rng.choice(['iid', 'repeat', 'runs', 'periodic', 'mixed'])
    nprng = np.random.default_rng(rng.getrandbits(32))
    syms = nprng.permutation(256)[:sigma].astype(np.uint8)
    if kind == 'iid':
        t = syms[nprng.integers(0, sigma, n)]
    elif kind == 'runs':
        parts = []
        while sum(len(p) for p in parts) < n:
            parts.append(np.full(int(nprng.integers(1, 1 + max(1, n // 8))), syms[nprng.integers(0, sigma)], dtype=np.uint8))
        t = np.concatenate(parts)[:n]
    elif kind == 'periodic':
        # one word repeated, half of the time long enough for the closed form of rle_build.h (n >= 2^15), with a
        # tail of other bytes behind the repetition (up to past what that path accepts)
        if rng.random() < 0.5:
            n = max(n, int(2 ** rng.uniform(15, 17)))
        period = syms[nprng.integers(0, sigma, int(nprng.integers(1, rng.choice([8, 300, 1100]))))]
        t = np.tile(period, n // len(period) + 1)[:n]
        if rng.random() < 0.6:
            tail = syms[nprng.integers(0, sigma, int(nprng.integers(0, rng.choice([4, 200, 1100]))))]
            if len(tail) and len(tail) < n:
                t[n - len(tail):] = tail
    else:
        base = syms[nprng.integers(0, sigma, max(1, n // rng.choice([2, 3, 7, 50])))]
        parts = []
        while sum(len(p) for p in parts) < n:
            c = base.copy()
            if kind == 'mixed' and len(c) > 4:
                for p in nprng.integers(0, len(c), rng.randint(0, 5)):
                    c[p] = syms[nprng.integers(0, sigma)]
            parts.append(c[: int(nprng.integers(1, len(c) + 1))])
        t = np.concatenate(parts)[:n]
    return np.ascontiguousarray(t)


def entries_of(rng):
    alphabet = rng.choice(['ab', 'abc', 'abcdefgh', 'aé☃', '\x00a', 'ab \t.', 'abcdefghijklmnopqrstuvwxyz0123456789'])
    m = rng.randint(1, 600)
    L = rng.choice([2, 8, 40, 300])
    out = []
    for _ in range(m):
        if out and rng.random() < 0.3:
            out.append(rng.choice(out))
        else:
            out.append(''.join(rng.choice(alphabet) for _ in range(rng.randint(0, L))))
    return alphabet, out


def build(path, entries, limit, W):
    w = W(path, limit) if limit is not None else W(path)
    for e in entries:
        w.add_entry(e)
    w.finalize()
    if hasattr(w, 'close'):
        w.close()
    return pathlib.Path(path).read_bytes()


def random_knobs(rng):
    """Builder / search / file-path switches that must never change a result -- drawn mechanically from the library's own
    registry (csrc/knobs.h through pss_knob_info): every switch with a `fuzz` column, each with probability 1/4, any
    combination on any input.  A switch added to the registry is fuzzed from then on without a line here."""
    from pysubstringsearch_amd import _ffi
    table = _ffi.knobs()
    for k in table:
        os.environ.pop(k['name'], None)
    _ffi.lib.pss_reload_env()        # the search switches are read once, not per call
    if rng.random() < 0.5:
        return
    for k in table:
        if k['fuzz'] and rng.random() < 0.25:
            os.environ[k['name']] = rng.choice(k['fuzz'])
    _ffi.lib.pss_reload_env()


def file_case(rng, tmp):
    """add_entries_from_file_lines: LF / CRLF / lone CR / empty lines / no final newline, raw bytes."""
    pieces = []
    for _ in range(rng.randint(0, 300)):
        body = bytes(rng.choice([97, 98, 99, 0, 13, 200, 255, 32]) for _ in range(rng.randint(0, 30)))
        pieces.append(body + rng.choice([b'\n', b'\n', b'\r\n', b'\n\n', b'\r\r\n']))
    blob = b''.join(pieces)
    if rng.random() < 0.5 and blob.endswith(b'\n'):
        blob = blob[:-1]
    src = os.path.join(tmp, 'in.txt')
    pathlib.Path(src).write_bytes(blob)
    limit = rng.choice([None, 40, 100, 1000])
    tail = rng.random() < 0.3
    out = []
    for W, name in ((pysubstringsearch.Writer, 'g'), (O.OracleWriter, 'o')):
        path = os.path.join(tmp, name + 'f.idx')
        w = W(path, limit) if limit is not None else W(path)
        w.add_entries_from_file_lines(src)
        if tail:
            w.add_entry('tail entry')
        w.finalize()
        if hasattr(w, 'close'):
            w.close()
        out.append(pathlib.Path(path).read_bytes())
    return out


def one_case(seed, tmp):
    rng = random.Random(seed)
    random_knobs(rng)
    t = make_text(rng)
    assert (sa_gpu(t) == O.sa(t)).all(), 'suffix array differs'
    if os.environ.get('FUZZ_BIG'):
        return
    if rng.random() < 0.3:
        state = rng.getstate()
        g, o = file_case(random.Random(seed ^ 0x5bd1e995), tmp)
        rng.setstate(state)
        assert g == o, 'file ingest: container differs'
    alphabet, entries = entries_of(rng)
    limit = rng.choice([None, 64, 257, 5000, 70000])
    if limit is not None:
        limit = max(limit, max(len(e.encode()) for e in entries) + 1)
    p, q = os.path.join(tmp, 'g.idx'), os.path.join(tmp, 'o.idx')
    assert build(p, entries, limit, pysubstringsearch.Writer) == build(q, entries, limit, O.OracleWriter), 'container differs'
    text = '\n'.join(entries) + '\n'
    queries = ['', '\n']
    for _ in range(40):
        s = rng.randrange(len(text))
        queries.append(text[s:s + rng.randint(1, 12)])
    for _ in range(10):
        queries.append(''.join(rng.choice(alphabet) for _ in range(rng.randint(1, 5))))
    o = O.OracleReader(q)
    with pysubstringsearch.Reader(p) as r:
        for s in queries[:12]:
            assert sorted(r.search(s)) == sorted(o.search(s)), repr(s)
        ents, counts = r.search_batch_raw([s.encode() for s in queries])
        oe, oc = o.search_multiple_bytes([s.encode() for s in queries])
        assert counts == oc.tolist(), 'per-query counts differ'
        pos = 0
        for c in counts:
            assert sorted(ents[pos:pos + c]) == sorted(oe[pos:pos + c])
            pos += c
    o.close()


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    O.use_reference_sa(O.have_reference())
    t0 = time.time()
    cases = 0
    with tempfile.TemporaryDirectory() as tmp:
        while time.time() - t0 < budget:
            if os.environ.get('FUZZ_TRACE'):
                print('seed', seed, file=sys.stderr, flush=True)
            try:
                one_case(seed, tmp)
            except Exception as e:   # noqa: BLE001
                print(f'FAIL seed={seed}: {type(e).__name__}: {e}')
                raise
            cases += 1
            seed += 1
    print(f'fuzz: {cases} cases in {time.time() - t0:.0f} s, all equal to the oracle (last seed {seed - 1})')


if __name__ == '__main__':
    main()
