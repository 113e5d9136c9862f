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
  roofline         dominant kernel (the scatter instantiation with the largest summed
                   duration; `lines`: fs_scatter_kernel<4, 4>, 4 B key + 4 B value in,
                   same out = 16 algorithmic bytes per element) over its HIP-event
                   duration, against the 8 TB/s HBM peak
  build_roofline   the whole build against the same peak: modelled algorithmic bytes
                   (A_model) / build time, beside A_min = 5 n and the PMC-measured bytes
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
        if world > 1:
            # packed local result -> gather to rank 0 -> one Python list there
            pk = reader.search_batch_packed(queries)
            merged = pdist.gather_packed(pk.data, np.diff(pk.offsets.astype(np.int64)), pk.counts, dst=0, packed=True)
            entries = merged[1][:-1] if merged is not None else []      # one offset per entry (packed result)
        else:
            entries, counts = reader.search_batch_raw(queries)
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
        # dominant kernel = the scatter instantiation with the largest summed duration in the profiled
        # build: the passes of the initial sort are fs_scatter_kernel<KIN, KOUT> (key plane bytes in /
        # out, 4-byte values), the sorts of the rounds and of the sizing sample rs_scatter_kernel<false>
        cands = []
        for idx in range(9):
            if prof['fs_launches'][idx]:
                kin, kout = (idx // 3) * 4, (idx % 3) * 4
                bpe = (kin + 4 if kin else 1) + kout + 4          # the text pass reads 1 B / suffix
                cands.append((prof['fs_ms'][idx], f'fs_scatter_kernel<{kin}, {kout}>', prof['fs_launches'][idx],
                              prof['fs_elems'][idx], bpe))
        if prof['pairs_launches']:
            cands.append((prof['ms_pairs'], 'rs_scatter_kernel<false>', prof['pairs_launches'], prof['pairs_elems'], 24))
        roof = None
        if cands:
            ms_sum, kname, launches, elems, bpe = max(cands)
            bytes_per_launch = float(bpe) * elems / launches
            ms_per_launch = ms_sum / launches
            achieved = bytes_per_launch / (ms_per_launch * 1e-3) / 1e9
            traffic = None
            pmc = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
            if os.path.exists(pmc) and args.corpus == 'lines' and args.logn == 29:
                try:
                    traffic = json.load(open(pmc)).get('kernels', {}).get(kname, {}).get('bytes_per_launch')
                except Exception:
                    traffic = None
            roof = {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                    'kernel': kname, 'launches_per_build': launches, 'algorithmic_bytes_per_element': bpe,
                    'ms_per_launch': round(ms_per_launch, 4), 'algorithmic_bytes_per_launch': int(bytes_per_launch)}
        # whole-build roofline: A_min = read T once + write SA once; A_model = what this
        # algorithm must move: alphabet 3 n, the passes of the initial sort (below),
        # initial rerank 8 n, ~80 B per active suffix and round
        build_ms = build_s / args.steps * 1e3
        # initial sort: pass p reads its key plane twice (histogram + scatter) and the values once,
        # writes the next plane and the values; planes are 8 B while > 32 key bits remain, then 4 B
        a_sort, kin = 0, 0
        passes0 = sa_stats['initial_passes']
        for p in range(passes0):
            rem = sa_stats['key_bits'] - 8 * (p + 1)
            kout = 0 if rem <= 0 else (4 if rem <= 32 else 8)
            a_sort += ((2 * kin + 4) if kin else 2) * n + (kout + 4) * n
            kin = kout
        a_sort += 32 * max(0, sa_stats['sort_elems'] - passes0 * n)     # (u64, u32) pair passes of the rounds
        a_model = 3 * n + 8 * n + a_sort + 80 * sa_stats['sum_active']
        measured = None
        pmcb = os.path.join(ROOT, 'profiles', 'pmc_build_traffic.json')
        if os.path.exists(pmcb) and args.corpus == 'lines' and args.logn == 29:
            try:
                measured = json.load(open(pmcb)).get('total_bytes')
            except Exception:
                measured = None
        # SURVEY 8(d)'s yardstick (64-bit keys in every pass, P = 8): 5 n + sum over rounds of n_r (32 P + 44)
        a_survey = 5 * n + 300 * (n + sa_stats['sum_active'])
        build_roof = {'a_min_bytes': 5 * n, 'a_model_bytes': int(a_model),
                      'survey_model_bytes': int(a_survey), 'survey_model_frac': round(a_survey / build_ms / 1e6 / HBM_PEAK_GBS, 4),
                      'effective_gbs_a_min': round(5 * n / build_ms / 1e6, 1),
                      'achieved': round(a_model / build_ms / 1e6, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                      'frac': round(a_model / build_ms / 1e6 / HBM_PEAK_GBS, 4), 'traffic': measured,
                      'passes': sa_stats['initial_passes'], 'rounds': sa_stats['rounds'], 'sum_active': sa_stats['sum_active']}
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
                                                  'round_passes', 'sum_active', 'sort_launches', 'mode', 'key_bits', 'ms_total')},
            'roofline': roof,
            'build_roofline': build_roof,
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
