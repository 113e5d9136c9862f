"""Generates tests/golden/sa_big.json: known answers of the REAL libsais
(oracle/_ref/libsais.so, compiled from /root/reference/src/libsais/libsais.c) for
full-size chunks -- the 512 MiB default of src/lib.rs:57 -- of the synthetic
corpora of SURVEY 8(d).  Run in the dev container only (minutes of CPU per chunk):

    python tests/golden/make_golden_big.py [--workers 4] [--logn 29]

Per (corpus kind, chunk_index):
  text_sha256   sha256 of the generated text
  sa_sha256     sha256 of the little-endian int32 suffix array (what the .idx record holds)
  sa_poly64     sum over i of (SA[i] + 1) * (2 i + 1) mod 2^64 -- a positional checksum
                a GPU computes in a millisecond (bench.py / tests verify every timed
                build with it; the sha256 is checked where a 2 GiB D2H is affordable)
  sa_stride     SA[k * n / 256] for k in 0..255 (narrows down where a mismatch lives)

Data only: inputs are regenerated from the corpus specification, outputs are hashes.
"""
import pathlib
import argparse
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402

KIND_IDS = {'lines': 0, 'words': 1, 'runs': 2, 'periodic': 3, 'repeat_line': 4, 'dup_blocks': 5, 'mixed': 6, 'source': 7}

# Round 5: chunks ABOVE the default size.  The reference accepts any max_chunk_len below 2^30 (src/lib.rs:57,116: the
# suffix array's byte length is a u32); container format 2 lifts that to 2^31 - 1.  2^30 - 1 is the largest chunk the
# reference's own format holds, 1.25 GiB one only format 2 can.
BIG_N = (1 << 30) - 1
FORMAT2_N = 5 << 28


def poly64(sa: np.ndarray) -> int:
    """sum (SA[i] + 1) * (2 i + 1) mod 2^64, in blocks (uint64 arithmetic wraps)."""
    acc = np.uint64(0)
    blk = 1 << 24
    with np.errstate(over='ignore'):
        for s in range(0, sa.size, blk):
            v = sa[s:s + blk].astype(np.uint64) + np.uint64(1)
            w = np.arange(s, s + v.size, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
            acc = acc + (v * w).sum(dtype=np.uint64)
    return int(acc)


def one(job):
    kind, chunk, n = job
    from oracle import oracle as O
    from pysubstringsearch_amd import _ffi
    t0 = time.time()
    text = np.empty(n, dtype=np.uint8)
    _ffi.check(_ffi.lib.pss_gen_corpus(KIND_IDS[kind], text.ctypes.data, n, chunk))
    if kind == 'lines':   # the oracle's independent generator must agree with the product's
        assert (O.gen_lines(n, chunk) == text).all()
    sa = O.sa_reference(text)
    rec = {
        'kind': kind, 'chunk_index': chunk, 'n': n,
        'text_sha256': hashlib.sha256(text.tobytes()).hexdigest(),
        'sa_sha256': hashlib.sha256(sa.astype('<i4').tobytes()).hexdigest(),
        'sa_poly64': poly64(sa),
        'sa_stride': [int(x) for x in sa[:: n // 256][:256]],
        'libsais_seconds': round(time.time() - t0, 1),
    }
    print(kind, chunk, rec['sa_sha256'][:16], rec['libsais_seconds'], 's', flush=True)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workers', type=int, default=4)
    ap.add_argument('--logn', type=int, default=29)
    ap.add_argument('--lines-chunks', type=int, default=15)
    ap.add_argument('--words-chunks', type=int, default=15)
    ap.add_argument('--big', action='store_true', help='also the chunks of 2^30 - 1 and 5 * 2^28 bytes')
    ap.add_argument('--only', default='', help='comma-separated kinds to (re)generate')
    ap.add_argument('--out', default=os.path.join(HERE, 'sa_big.json'))
    args = ap.parse_args()
    from oracle import oracle as O
    assert O.have_reference(), 'needs oracle/_ref/libsais.so (make -C oracle)'
    n0 = 1 << args.logn
    jobs = [('lines', 0, n0), ('words', 0, n0), ('runs', 0, n0), ('periodic', 0, n0)]
    jobs += [('lines', c, n0) for c in range(1, args.lines_chunks)]
    jobs += [('words', c, n0) for c in range(1, args.words_chunks)]
    jobs += [('repeat_line', 0, n0), ('dup_blocks', 0, n0)]      # general repeats (round 3)
    jobs += [('mixed', 0, n0)]                                   # natural text with a repetitive middle (round 4)
    jobs += [('source', 0, n0), ('source', 1, n0)]               # source-like text (round 5)
    if args.big:                                                 # above the default chunk size (round 5; ~6 GB and 3-4 minutes each)
        jobs += [('lines', 0, BIG_N), ('words', 0, BIG_N), ('mixed', 0, BIG_N), ('runs', 0, BIG_N), ('source', 0, BIG_N), ('words', 0, FORMAT2_N)]
    done = {}
    if os.path.exists(args.out):
        for r in json.loads(pathlib.Path(args.out).read_text())['chunks']:
            done[(r['kind'], r['chunk_index'], r['n'])] = r
    if args.only:
        jobs = [j for j in jobs if j[0] in args.only.split(',')]
    jobs = [j for j in jobs if (j[0], j[1], j[2]) not in done]
    with mp.get_context('spawn').Pool(args.workers) as pool:
        for rec in pool.imap_unordered(one, jobs):
            done[(rec['kind'], rec['chunk_index'], rec['n'])] = rec
            out = {'note': 'libsais (oracle/_ref) known answers for full-size chunks; see make_golden_big.py',
                   'chunks': [done[k] for k in sorted(done)]}
            json.dump(out, open(args.out + '.tmp', 'w'), indent=1)
            os.replace(args.out + '.tmp', args.out)


if __name__ == '__main__':
    main()
