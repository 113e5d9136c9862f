"""Multi-chunk resident-corpus search benchmark (BASELINE.json configs[2] / [3] on one GPU).

Builds C chunks of 2^logn bytes on the GPU (device-resident hand-off, no file),
keeps all of them resident in one Reader, then times
  * one batched call with Q mixed-length (4..32 B) queries, 50% sampled from the corpus;
  * single-query latency (batch of 1), R repetitions.
Optionally checks a sample of the queries against the CPU oracle on chunk 0 and
times the oracle (SA in RAM, one thread, one query at a time) on that chunk.

    python tests/tools/bench_corpus.py --chunks 15 --logn 29 --queries 100000
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

ALPHA = b'abcdefghijklmnopqrstuvwxyz0123456789 .'
KINDS = {'lines': 0, 'words': 1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--chunks', type=int, default=15)
    ap.add_argument('--logn', type=int, default=29)
    ap.add_argument('--corpus', default='lines', choices=sorted(KINDS))
    ap.add_argument('--queries', type=int, default=100000)
    ap.add_argument('--single-reps', type=int, default=1000)
    ap.add_argument('--oracle-check', type=int, default=2000, help='queries checked against the oracle on chunk 0')
    args = ap.parse_args()

    import torch
    from pysubstringsearch_amd import Reader, _ffi
    lib = _ffi.lib
    n = 1 << args.logn
    rng = np.random.default_rng(1)

    h = ctypes.c_void_p()
    _ffi.check(lib.pss_reader_create(0, ctypes.byref(h)))
    reader = Reader._from_handle(h)
    dSA = torch.empty(n, dtype=torch.int32, device='cuda')
    st = _ffi.SaStats()
    sampled = []
    host0 = None
    sa0 = None
    t_build = 0.0
    per_chunk = (args.queries // 2 + args.chunks - 1) // args.chunks
    for c in range(args.chunks):
        host = np.empty(n, dtype=np.uint8)
        _ffi.check(lib.pss_gen_corpus(KINDS[args.corpus], host.ctypes.data, n, c))
        dT = torch.from_numpy(host).cuda()
        t0 = time.perf_counter()
        _ffi.check(lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, 0, ctypes.byref(st)))
        t_build += time.perf_counter() - t0
        _ffi.check(lib.pss_reader_add_chunk_device(h, dT.data_ptr(), dSA.data_ptr(), n))
        while len(sampled) < per_chunk * (c + 1):
            s = int(rng.integers(0, n - 40))
            ln = int(rng.integers(4, 33))
            cand = host[s:s + ln].tobytes()
            if b'\n' not in cand:
                sampled.append(cand)
        if c == 0:
            host0 = host
            if args.oracle_check:
                sa0 = dSA.cpu().numpy()
        del dT
    queries = sampled[:args.queries // 2]
    while len(queries) < args.queries:
        ln = int(rng.integers(4, 33))
        queries.append(bytes(ALPHA[int(i)] for i in rng.integers(0, len(ALPHA), ln)))
    order = rng.permutation(len(queries))
    queries = [queries[i] for i in order]

    out = {'corpus': args.corpus, 'chunks': args.chunks, 'chunk_bytes': n, 'corpus_bytes': n * args.chunks,
           'build_gbs': round(n * args.chunks / t_build / 1e9, 3), 'queries': len(queries)}
    reader.search_batch_raw(queries[:100])   # warm-up
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        entries, counts = reader.search_batch_raw(queries)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    stats = reader.last_stats()
    out.update({'batch_s': round(best, 5), 'batched_queries_per_sec': round(len(queries) / best, 1),
                'entries': len(entries), 'hits_per_query': round(stats['hits'] / len(queries), 3),
                'ms_device': round(stats['ms_device'], 3), 'ms_interval': round(stats['ms_interval'], 3)})
    t0 = time.perf_counter()
    pk = reader.search_batch_packed(queries)
    dt = time.perf_counter() - t0
    out.update({'packed_batch_s': round(dt, 5), 'packed_queries_per_sec': round(len(queries) / dt, 1),
                'packed_bytes': int(pk.data.size)})
    # single-query latency
    lat = []
    for q in queries[:args.single_reps]:
        t0 = time.perf_counter()
        reader.search_batch_raw([q])
        lat.append(time.perf_counter() - t0)
    lat.sort()
    out.update({'single_query_us_median': round(lat[len(lat) // 2] * 1e6, 1), 'single_query_us_p90': round(lat[int(len(lat) * 0.9)] * 1e6, 1),
                'single_queries_per_sec': round(len(lat) / sum(lat), 1)})
    # oracle on chunk 0: parity on a sample + CPU queries/s
    if args.oracle_check and sa0 is not None:
        import tempfile
        from oracle import oracle as O
        with tempfile.TemporaryDirectory() as d:
            p = os.path.join(d, 'c0.idx')
            with open(p, 'wb') as f:
                f.write(np.uint32(n).tobytes())
                f.write(host0.tobytes())
                f.write(np.uint32((4 * n) & 0xffffffff).tobytes())
                f.write(sa0.astype('<i4').tobytes())
            o = O.OracleReader(p)
            h1 = ctypes.c_void_p()
            _ffi.check(lib.pss_reader_create(0, ctypes.byref(h1)))
            r1 = Reader._from_handle(h1)
            d0 = torch.from_numpy(host0).cuda()
            s0 = torch.from_numpy(sa0).cuda()
            _ffi.check(lib.pss_reader_add_chunk_device(h1, d0.data_ptr(), s0.data_ptr(), n))
            qs = queries[:args.oracle_check]
            got, gc = r1.search_batch_raw(qs)
            t0 = time.perf_counter()
            exp, ec = o.search_multiple_bytes(qs)
            t_cpu = time.perf_counter() - t0
            assert gc == ec.tolist(), 'per-query counts differ from the oracle'
            assert sorted(got) == sorted(exp), 'result multiset differs from the oracle'
            out.update({'oracle_checked_queries': len(qs), 'oracle_parity': True,
                        'cpu_queries_per_sec_one_chunk_1thread': round(len(qs) / t_cpu, 1),
                        'cpu_queries_per_sec_scaled_to_corpus': round(len(qs) / t_cpu / args.chunks, 1)})
            r1.close()
            o.close()
    reader.close()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
