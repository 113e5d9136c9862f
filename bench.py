#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on N MI355X of one node.

One "step" = one pass of the hot path over one batch of synthetic input per
GPU (BASELINE.json configs[1]): build the 32-bit suffix array of one 512 MiB
`lines` chunk that is already resident in HBM (pss_sa_build_device), hand the
chunk to a device-resident Reader, and answer one batch of 10 000 8-byte
queries (5 000 sampled from the text, 5 000 random) through the batched search.

  value            index-build GB/s  = chunk bytes of all ranks / build time
  queries_per_sec  batched queries/s = queries / (H2D queries + kernels + D2H
                   results + Python list construction [+ gather to rank 0])
  roofline         dominant kernel rs_scatter_kernel<false, false>: 24 algorithmic
                   bytes per element (8 B key + 4 B value in, same out) over its
                   HIP-event duration, against the 8 TB/s HBM peak
  cpu_baseline     the reference's libsais (oracle/_ref) on a bounded sample,
                   rank 0, N = 1 only

N > 1: one process per GPU (torch.distributed, RCCL); chunk r lives on rank r,
no collective on the build path, results of every rank are gathered to rank 0
(weak scaling: corpus grows with N).
"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
KINDS = {'lines': 0, 'words': 1, 'runs': 2, 'periodic': 3}
ALPHA = b'abcdefghijklmnopqrstuvwxyz0123456789 .'


def make_queries(text: np.ndarray, nq: int, qlen: int, seed: int = 1):
    """half sampled from the chunk (no newline inside), half uniform over ALPHA."""
    rng = np.random.default_rng(seed)
    out = []
    raw = text.tobytes() if text.size <= (1 << 27) else None
    while len(out) < nq // 2:
        s = int(rng.integers(0, text.size - qlen))
        cand = raw[s:s + qlen] if raw is not None else text[s:s + qlen].tobytes()
        if b'\n' not in cand:
            out.append(cand)
    for _ in range(nq - len(out)):
        out.append(bytes(ALPHA[int(i)] for i in rng.integers(0, len(ALPHA), qlen)))
    return out


def cpu_baseline(host: np.ndarray, queries, sample_logn: int):
    """Reference CPU path on a bounded sample: libsais exactly as src/lib.rs:30-36
    calls it (1 thread), then the oracle's restatement of Reader::search, one
    query at a time like the reference's search_multiple loop."""
    from oracle import oracle as O
    m = min(host.size, 1 << sample_logn)
    sample = host[:m].copy()
    sample[-1] = 10
    kind = 'reference' if O.have_reference() else 'port'
    t0 = time.perf_counter()
    sa = O.sa_reference(sample) if kind == 'reference' else O.sa_restatement(sample)
    t_sa = time.perf_counter() - t0
    qps = None
    try:
        with tempfile.TemporaryDirectory() as d:
            p = os.path.join(d, 'sample.idx')
            with open(p, 'wb') as f:   # chunk record layout, src/lib.rs:112-119
                f.write(np.uint32(m).tobytes())
                f.write(sample.tobytes())
                f.write(np.uint32(4 * m).tobytes())
                f.write(sa.astype('<i4').tobytes())
            r = O.OracleReader(p)
            t0 = time.perf_counter()
            total = 0
            for q in queries:
                total += len(r.search_bytes(q))
            qps = len(queries) / (time.perf_counter() - t0)
            r.close()
    except OSError:
        pass
    return {
        'value': round(m / t_sa / 1e9, 6), 'unit': 'GB/s', 'cores': 1, 'kind': kind,
        'sample': f'SA build of the first {m >> 20} MiB of the same chunk '
                  f'({"libsais from oracle/_ref" if kind == "reference" else "oracle restatement"}, 1 thread, {t_sa:.1f} s); '
                  f'queries/s = oracle Reader::search restatement, SA in RAM, 1 thread, the same {len(queries)} queries one at a time',
        'queries_per_sec': None if qps is None else round(qps, 1),
        'host_cpus': os.cpu_count(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--corpus', default='lines', choices=sorted(KINDS))
    ap.add_argument('--logn', type=int, default=29, help='log2 of the chunk size (29 = the 512 MiB default chunk)')
    ap.add_argument('--queries', type=int, default=10000)
    ap.add_argument('--qlen', type=int, default=8)
    ap.add_argument('--cpu-sample-logn', type=int, default=26)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    # Test hook for 1-GPU boxes: PSS_BENCH_BACKEND=gloo lets every rank share GPU 0
    # so the N > 1 code path (sharding, gather, max-over-ranks timing) can be exercised.
    backend = os.environ.get('PSS_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    from pysubstringsearch_amd import Reader, _ffi
    from pysubstringsearch_amd import dist as pdist
    lib = _ffi.lib
    n = 1 << args.logn
    dev = local_rank

    host = np.empty(n, dtype=np.uint8)
    _ffi.check(lib.pss_gen_corpus(KINDS[args.corpus], host.ctypes.data, n, rank))
    dT = torch.from_numpy(host).cuda()
    dSA = torch.empty(n, dtype=torch.int32, device='cuda')
    if world > 1:
        box = [make_queries(host, args.queries, args.qlen) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        queries = box[0]
    else:
        queries = make_queries(host, args.queries, args.qlen)
    torch.cuda.synchronize()

    st = _ffi.SaStats()
    last = {}
    h = ctypes.c_void_p()
    _ffi.check(lib.pss_reader_create(dev, ctypes.byref(h)))
    reader = Reader._from_handle(h)   # device-resident index of this rank, refreshed every step

    def step(flags=0):
        t0 = time.perf_counter()
        _ffi.check(lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, dev, flags, ctypes.byref(st)))
        t1 = time.perf_counter()
        # Writer -> Reader hand-off through HBM (no file): the fresh text + SA replace the resident chunk
        _ffi.check(lib.pss_reader_set_chunk_device(h, 0, dT.data_ptr(), dSA.data_ptr(), n))
        t2 = time.perf_counter()
        entries, counts = reader.search_batch_raw(queries)
        if world > 1:
            merged = pdist.gather_results(entries, counts, dst=0)
            if merged is not None:
                entries = merged[0]
        t3 = time.perf_counter()
        last['entries'] = len(entries)
        last['search_stats'] = reader.last_stats()
        return t1 - t0, t3 - t2

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t_begin = time.perf_counter()
    build_s = search_s = 0.0
    for _ in range(args.steps):
        b, s = step()
        build_s += b
        search_s += s
    sync_all()
    total_s = time.perf_counter() - t_begin
    sa_stats = st.as_dict()
    t = torch.tensor([build_s, search_s, total_s], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    build_s, search_s, total_s = t.tolist()

    # roofline of the dominant kernel: one extra build in profile mode (HIP events
    # on the engine's own stream around every radix-pass launch), outside the timed region
    _ffi.check(lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, dev, 1, ctypes.byref(st)))
    prof = st.as_dict()

    # secondary corpus (natural-text-like LCP), outside the timed region, N = 1 only
    secondary = None
    if world == 1 and args.corpus == 'lines' and not os.environ.get('PSS_BENCH_NO_SECONDARY'):
        w_host = np.empty(n, dtype=np.uint8)
        _ffi.check(lib.pss_gen_corpus(KINDS['words'], w_host.ctypes.data, n, 0))
        w_dT = torch.from_numpy(w_host).cuda()
        wst = _ffi.SaStats()
        best = None
        for _ in range(3):
            _ffi.check(lib.pss_sa_build_device(w_dT.data_ptr(), dSA.data_ptr(), n, dev, 0, ctypes.byref(wst)))
            best = wst.ms_total if best is None else min(best, wst.ms_total)
        wd = wst.as_dict()
        secondary = {'corpus': 'words', 'chunk_bytes': n, 'build_ms': round(best, 3),
                     'index_build_gbs': round(n / best / 1e6, 4),
                     'sa_stats': {k: wd[k] for k in ('key_chars', 'initial_passes', 'rounds', 'text_rounds', 'round_passes',
                                                     'sum_active', 'big_elems', 'mode')}}
        del w_dT

    if rank == 0:
        roof = None
        if prof['pairs_launches']:
            bytes_per_launch = 24.0 * prof['pairs_elems'] / prof['pairs_launches']
            ms_per_launch = prof['ms_pairs'] / prof['pairs_launches']
            achieved = bytes_per_launch / (ms_per_launch * 1e-3) / 1e9
            traffic = None
            pmc = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get('rs_scatter_pairs_bytes_per_launch')
                except Exception:
                    traffic = None
            roof = {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                    'kernel': 'rs_scatter_kernel<false, false>', 'launches_per_build': prof['pairs_launches'],
                    'ms_per_launch': round(ms_per_launch, 4), 'algorithmic_bytes_per_launch': int(bytes_per_launch)}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(host, queries, args.cpu_sample_logn)
        out = {
            'metric': 'queries/sec (batched) + index-build GB/s on 512MB chunk, 1/2/4/8 GPU',
            'value': round(world * n * args.steps / build_s / 1e9, 4),
            'unit': 'GB/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(total_s / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'u64', 'data': 'synthetic',
            'config': {
                'workload': f'configs[1]: one {n >> 20} MiB synthetic `{args.corpus}` chunk per GPU, suffix-array build + '
                            f'{len(queries)} {args.qlen}-byte queries (50% sampled from the text) in one batch',
                'corpus': args.corpus, 'chunk_bytes': n, 'queries': len(queries), 'query_len': args.qlen,
                'value_is': 'index-build GB/s (text bytes of all ranks / suffix-array build time, inputs resident in HBM)',
            },
            'queries_per_sec': round(len(queries) * args.steps / search_s, 1),
            'build_ms': round(build_s / args.steps * 1e3, 3),
            'search_ms': round(search_s / args.steps * 1e3, 3),
            'entries_per_batch': last.get('entries'),
            'search_stats': last.get('search_stats'),
            'sa_stats': {k: sa_stats[k] for k in ('sigma', 'code_bits', 'key_chars', 'initial_passes', 'rounds',
                                                  'round_passes', 'sum_active', 'sort_launches', 'mode', 'ms_total')},
            'roofline': roof,
            'cpu_baseline': cpu,
            'secondary': secondary,
        }
        print(json.dumps(out))
    reader.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
