"""Repeat structure of a text, from libsais' suffix array (oracle/_ref) and Kasai's LCP array: how much of it is tied
after k symbols, how large the tied groups are -- the figures the `source` corpus (pss_gen_corpus kind 7) was fitted to.

    python tests/tools/lcp_stats.py real <logn>            files found on this machine (tests/tools/real_text.py)
    python tests/tools/lcp_stats.py <kind> <logn> [chunk]  a synthetic corpus

CPU only; minutes at logn = 26.
"""
import pathlib
import ctypes
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, '.')

SRC = r'''
#include <stdint.h>
#include <stdlib.h>
/* Kasai: lcp[i] = LCP(suffix SA[i-1], suffix SA[i]); lcp[0] = 0 */
void kasai(const uint8_t *T, const int32_t *SA, int32_t n, int32_t *lcp, int32_t *rank)
{
    for (int32_t i = 0; i < n; ++i) rank[SA[i]] = i;
    int32_t h = 0;
    for (int32_t i = 0; i < n; ++i) {
        if (rank[i] > 0) {
            const int32_t j = SA[rank[i] - 1];
            while (i + h < n && j + h < n && T[i + h] == T[j + h]) ++h;
            lcp[rank[i]] = h;
            if (h > 0) --h;
        } else { lcp[0] = 0; h = 0; }
    }
}
/* members of groups above `big` members at depth d0 that are still ONE group at depth d1 (every member's next d1 - d0
 * symbols equal: a text round finds nothing to sort in them) */
int64_t unsplit(const int32_t *lcp, int32_t n, int32_t d0, int32_t d1, int32_t big)
{
    int64_t run = 1, total = 0;
    int32_t minl = 0x7fffffff;
    for (int32_t i = 1; i <= n; ++i) {
        if (i < n && lcp[i] >= d0) { ++run; if (lcp[i] < minl) minl = lcp[i]; continue; }
        if (run > big && minl >= d1) total += run;
        run = 1;
        minl = 0x7fffffff;
    }
    return total;
}
/* members of groups (maximal runs of the suffix array sharing >= depth symbols) above each size in `sizes` */
void groups(const int32_t *lcp, int32_t n, int32_t depth, const int32_t *sizes, int ns, int64_t *members, int64_t *tied)
{
    int64_t run = 1;
    *tied = 0;
    for (int k = 0; k < ns; ++k) members[k] = 0;
    for (int32_t i = 1; i <= n; ++i) {
        if (i < n && lcp[i] >= depth) { ++run; continue; }
        if (run > 1) *tied += run;
        for (int k = 0; k < ns; ++k) if (run > sizes[k]) members[k] += run;
        run = 1;
    }
}
'''


def helper():
    d = tempfile.gettempdir()
    so = os.path.join(d, 'pss_lcp_stats.so')
    if not os.path.exists(so):
        c = os.path.join(d, 'pss_lcp_stats.c')
        pathlib.Path(c).write_text(SRC)
        subprocess.run(['gcc', '-O2', '-shared', '-fPIC', '-o', so, c], check=True)
    lib = ctypes.CDLL(so)
    lib.kasai.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    lib.unsplit.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
    lib.unsplit.restype = ctypes.c_int64
    lib.groups.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return lib


def text_of(kind, logn, chunk):
    n = 1 << logn
    if kind == 'real':
        sys.path.insert(0, 'tests/tools')
        import real_text
        raw = real_text.collect(n)
        t = np.frombuffer(raw, dtype=np.uint8).copy()
        t[-1] = 10
        return t
    import bench
    from pysubstringsearch_amd import _ffi
    t = np.empty(n, dtype=np.uint8)
    _ffi.check(_ffi.lib.pss_gen_corpus(bench.KINDS[kind], t.ctypes.data, n, chunk))
    return t


def main():
    kind = sys.argv[1]
    logn = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    from oracle import oracle as O
    t = text_of(kind, logn, chunk)
    n = t.size
    nl = t == 10
    starts = np.flatnonzero(nl)[:-1] + 1
    print(f'{kind}: {n} bytes, {len(np.unique(t))} byte values, {int(nl.sum())} lines of {n / max(1, int(nl.sum())):.1f} bytes, '
          f'{100.0 * np.isin(t[starts], (32, 9)).mean():.1f} % start with a blank, blanks {100.0 * (t == 32).mean():.1f} % of the bytes')
    sa = O.sa_reference(t) if O.have_reference() else O.sa_restatement(t)
    h = helper()
    lcp = np.empty(n, dtype=np.int32)
    rank = np.empty(n, dtype=np.int32)
    h.kasai(t.ctypes.data, sa.ctypes.data, n, lcp.ctypes.data, rank.ctypes.data)
    del rank
    print(f'LCP mean {lcp.mean():.1f}, max {int(lcp.max())}')
    for k in (4, 8, 12, 16, 20, 28, 37, 53, 64, 128, 512, 4096, 65536):
        # a suffix is tied at depth k when it shares k symbols with a neighbour in the suffix array
        tied = np.maximum(lcp, np.append(lcp[1:], 0)) >= k
        print(f'  tied after {k:6d} symbols: {100.0 * tied.mean():6.2f} %')
    sizes = np.array([1, 512, 4096, 65536], dtype=np.int32)
    mem = np.zeros(4, dtype=np.int64)
    tied = ctypes.c_int64()
    for big in (512, 4096):
        u = h.unsplit(lcp.ctypes.data, n, 12, 20, big)
        print(f'  groups above {big} members at depth 12 that are still one group at depth 20: {100.0 * u / n:.1f} % of the suffixes')
    for depth in (12, 20, 28):
        h.groups(lcp.ctypes.data, n, depth, sizes.ctypes.data, 4, mem.ctypes.data, ctypes.byref(tied))
        print(f'  depth {depth}: in groups of > 1 / 512 / 4096 / 65536 members: ' + ' / '.join(f'{100.0 * m / n:.1f} %' for m in mem))


if __name__ == '__main__':
    main()
