"""Time-boxed fuzzing of the whole path against the oracle (libsais + the CPU restatement of the
reference's Writer / Reader): random alphabets, repeat structure, chunk limits and queries.

    python tests/tools/fuzz.py [seconds=120] [seed0=<time>]

Every case checks (1) the suffix array of the raw text, (2) the .idx container byte for byte,
(3) search / search_multiple multisets and per-query counts.  Prints the failing seed and stops."""
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
    return open(path, 'rb').read()


KNOBS = ('PSS_MODE', 'PSS_KEY_CHARS', 'PSS_KEY_DROP', 'PSS_TEXT_ROUNDS', 'PSS_NO_TIES_PASS', 'PSS_NO_SMALL_PATH', 'PSS_MSD',
         'PSS_MSD_NO_FUSE', 'PSS_MSD_SLOW_LOCAL', 'PSS_NO_PINNED_RESULTS', 'PSS_NO_MID_TIER', 'PSS_RLE', 'PSS_RLE_SORT', 'PSS_PERIOD', 'PSS_ANCHOR', 'PSS_ANCHOR_OMEGA',
         'PSS_COUNT_SORT', 'PSS_NO_PROBE', 'PSS_PERIODIC', 'PSS_PROBE_SKIP_PCT', 'PSS_ANCHOR_MIN_OMEGA', 'PSS_DEVICES', 'PSS_WRITER_MMAP', 'PSS_IO_THREADS',
         'PSS_WRITER_MMAP_MIN', 'PSS_INGEST_BLOCK', 'PSS_INGEST_MIN_ROOM', 'PSS_ANCHOR_SIDE', 'PSS_BIG_MERGE', 'PSS_NO_MID_MERGE', 'PSS_RESULT_ORDER', 'PSS_SS_SEG', 'PSS_SS')


def random_knobs(rng):
    """Builder / search switches that must never change a result."""
    from pysubstringsearch_amd import _ffi
    for k in KNOBS:
        os.environ.pop(k, None)
    _ffi.lib.pss_reload_env()        # the search switches are read once, not per call
    if rng.random() < 0.5:
        return
    if rng.random() < 0.5:
        os.environ['PSS_MODE'] = rng.choice(['dense', 'sparse', 'text'])
    if rng.random() < 0.5:
        os.environ['PSS_KEY_CHARS'] = str(rng.randint(1, 16))
        if rng.random() < 0.6:
            os.environ['PSS_KEY_DROP'] = str(rng.randint(0, 8))
    if rng.random() < 0.3:
        os.environ['PSS_TEXT_ROUNDS'] = str(rng.randint(0, 3))
    if rng.random() < 0.15:
        os.environ['PSS_NO_TIES_PASS'] = '1'
    if rng.random() < 0.5:
        os.environ['PSS_NO_SMALL_PATH'] = '1'
    if rng.random() < 0.6:
        os.environ['PSS_MSD'] = rng.choice(['0', '1', '1'])      # hybrid MSD initial sort forced on / off
        if rng.random() < 0.3:
            os.environ['PSS_MSD_NO_FUSE'] = '1'
        if rng.random() < 0.3:
            os.environ['PSS_MSD_SLOW_LOCAL'] = '1'
    if rng.random() < 0.3:
        os.environ['PSS_NO_PINNED_RESULTS'] = '1'
    if rng.random() < 0.3:
        os.environ['PSS_NO_MID_TIER'] = '1'
    if rng.random() < 0.5:
        os.environ['PSS_RLE'] = rng.choice(['0', '1', '1'])      # run-length path forced on / off
        if rng.random() < 0.4:
            os.environ['PSS_RLE_SORT'] = '1'                     # ... with the radix-sort expansion
    if rng.random() < 0.3:
        os.environ['PSS_PERIOD'] = '0'                           # never the closed form for one repeated word
    if rng.random() < 0.6:                                       # round 4: the anchor round forced on / off, narrow windows
        os.environ['PSS_ANCHOR'] = rng.choice(['0', '1', '1'])
        if rng.random() < 0.5:
            os.environ['PSS_ANCHOR_OMEGA'] = str(rng.choice([9, 12, 17, 33]))
        if rng.random() < 0.3:
            os.environ['PSS_NO_PROBE'] = '1'
        if rng.random() < 0.3:
            os.environ['PSS_PROBE_SKIP_PCT'] = rng.choice(['0', '20', '90'])      # when the probe sends the ties straight to the anchors
        if rng.random() < 0.3:
            os.environ['PSS_ANCHOR_MIN_OMEGA'] = rng.choice(['3', '7', '9'])      # narrowest window taken without PSS_ANCHOR=1
    if rng.random() < 0.3:
        os.environ['PSS_COUNT_SORT'] = '1'                       # rank rounds: counting instead of the segmented merge sort
    if rng.random() < 0.2:
        os.environ['PSS_PERIODIC'] = '0'                         # rank rounds: no periodic keys for the large groups
    if rng.random() < 0.3:
        os.environ['PSS_DEVICES'] = rng.choice(['0', '0,0', '0,0,0', 'all'])      # the default device list of Writer / Reader
    if rng.random() < 0.3:
        os.environ['PSS_WRITER_MMAP'] = rng.choice(['0', '1'])   # records through a shared mapping / pwrite
        os.environ['PSS_WRITER_MMAP_MIN'] = '16'                 # ... whatever their size (round 5: the route never ran before)
    if rng.random() < 0.3:                                       # direct file ingest on tiny chunks, tails longer than a block
        os.environ['PSS_INGEST_BLOCK'] = rng.choice(['16', '40', '300'])
        os.environ['PSS_INGEST_MIN_ROOM'] = rng.choice(['1', '20'])
    if rng.random() < 0.3:
        os.environ['PSS_ANCHOR_SIDE'] = rng.choice(['0', '1', '1'])   # anchors sorted beside the text round (second stream / thread)
    if rng.random() < 0.2:
        os.environ['PSS_SS'] = '1'                               # the sample sort for every text of >= 2^16 bytes
    if rng.random() < 0.2:
        os.environ['PSS_SS_SEG'] = '0'                           # sample sort: every tile sorted as one array (the round-4 local sort)
    if rng.random() < 0.3:
        os.environ['PSS_BIG_MERGE'] = rng.choice(['1', '2'])     # groups above 4096 members through the segmented merge sort
    if rng.random() < 0.2:
        os.environ['PSS_NO_MID_MERGE'] = '1'
    if rng.random() < 0.2:
        os.environ['PSS_RESULT_ORDER'] = 'sa'                    # (multisets are compared: the order must not matter)
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
    open(src, 'wb').write(blob)
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
        out.append(open(path, 'rb').read())
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
