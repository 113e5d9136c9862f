#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on N MI355X of one node.

--config chunk (default; BASELINE.json configs[1], weak scaling)
    One "step" = one pass of the hot path over one batch of synthetic input per GPU:
    build the 32-bit suffix array of one 512 MiB `lines` chunk that is already resident
    in HBM (pss_sa_build_device), hand the chunk to a device-resident Reader, and answer
    one batch of 10 000 8-byte queries (5 000 sampled from the text, 5 000 random).

      value            index-build GB/s  = chunk bytes of all ranks / build time
      queries_per_sec  batched queries/s = queries / (H2D queries + kernels + D2H
                       results + Python list construction [+ gather to rank 0])
      verified         the suffix array of the LAST TIMED STEP equals libsais' (the
                       reference's builder, src/lib.rs:30-36): positional checksum on the
                       GPU + sha256 of the 2 GiB array against tests/golden/sa_big.json
                       (or against libsais run on the spot for small --logn).  A build
                       whose result is wrong prints no value.
      build_ms_first_chunk, plan_hint
                       the timed steps rebuild chunks of one kind of text, so from the second on the build
                       runs under the plan of the previous chunk (its alphabet and its choice of sort, checked
                       inside the sort's first pass: pss_sa_stats.plan_hint); the time of a build that
                       starts from nothing (alphabet pass, recode, sizing sample) is reported beside it
      roofline         dominant kernel (the scatter instantiation with the largest summed
                       duration) over its HIP-event duration, against the 8 TB/s HBM peak
      build_roofline   the whole build against the same peak
      cpu_baseline     the reference's libsais (oracle/_ref) on a bounded sample,
                       rank 0, N = 1 only

    N > 1: one process per GPU (torch.distributed, RCCL); chunk r lives on rank r, no
    collective on the build path, results of every rank are gathered to rank 0.
    `python3 bench.py --gpus N` starts the N ranks itself (fresh child processes, one per GPU,
    before anything touches a GPU); under torchrun (WORLD_SIZE set) it is one of the ranks.

    The same line carries BASELINE configs[2] / [3] under "corpus15" (the leg below with at most
    3 steps; --no-corpus15 skips it): queries/s on the 7.5 GB corpus next to the CPU path.  At N > 1 a
    failure or hang of that leg does not cost the line its configs[1] result ("corpus15": {"error": ...}).

--config corpus15 (BASELINE.json configs[2] / [3], strong scaling)
    The 7.5 GB corpus: 15 `lines` chunks of 512 MiB, chunk c on rank c mod N (built there,
    every suffix array verified against libsais' checksum, then resident).  One "step" =
    one batch of 100 000 queries of 4..32 bytes (50 % sampled from the corpus) answered
    through the drop-in list API -- every rank searches its chunks, results are gathered
    to rank 0 (device buffers over RCCL), which builds the Python list.

      value            batched queries/s through the list API (what search_multiple returns)
      packed_queries_per_sec   the same batch through the packed (numpy) result API
      single_query_us  latency of Reader.search for one query over all chunks (N = 1); .low_latency_mode: the same with
                       Reader.set_low_latency() (resident search kernel, no launch per query)
      index_build_gbs  15 chunks / max over ranks of the summed build time
      cpu_baseline     SURVEY 8(d)(ii): the oracle's restatement of Reader::search, queries
                       one at a time, one thread per chunk, suffix arrays in RAM and -- like
                       the reference -- probed on disk with lseek + read(8 KiB)
"""
import pathlib
import argparse
import ctypes
import hashlib
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
KINDS = {'lines': 0, 'words': 1, 'runs': 2, 'periodic': 3, 'repeat_line': 4, 'dup_blocks': 5, 'mixed': 6, 'source': 7}
ALPHA = b'abcdefghijklmnopqrstuvwxyz0123456789 .'
METRIC = 'queries/sec (batched) + index-build GB/s on 512MB chunk, 1/2/4/8 GPU'
MASK64 = (1 << 64) - 1


# ----------------------------------------------------------------------------- verification --

def evidence(name: str):
    """A counter file under profiles/ and whether it was measured on THIS tree: (json or None, stamp).  The files carry the
    content hash of the engine's sources (tests/tools/tree_hash.py); bench.py recomputes it."""
    p = os.path.join(ROOT, 'profiles', name)
    if not os.path.exists(p):
        return None, None
    try:
        d = json.loads(pathlib.Path(p).read_text())
    except Exception:
        return None, None
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location('tree_hash', os.path.join(ROOT, 'tests', 'tools', 'tree_hash.py'))
        th = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(th)
        now = th.csrc_hash()
    except Exception:
        now = None
    stamp = {'file': 'profiles/' + name, 'measured_on_sources': d.get('csrc_sha256_16'), 'commit': d.get('commit'),
             'these_sources': now, 'current': bool(now and d.get('csrc_sha256_16') == now)}
    return d, stamp


def load_big_goldens():
    """(kind, chunk_index, n) -> libsais known answers (tests/golden/make_golden_big.py)."""
    p = os.path.join(ROOT, 'tests', 'golden', 'sa_big.json')
    if not os.path.exists(p):
        return {}
    return {(r['kind'], r['chunk_index'], r['n']): r for r in json.loads(pathlib.Path(p).read_text())['chunks']}


def sa_poly64_torch(d_sa) -> int:
    """sum (SA[i] + 1) * (2 i + 1) mod 2^64 of a device int32 tensor (int64 arithmetic wraps)."""
    import torch
    acc, n, blk = 0, d_sa.numel(), 1 << 26
    for s in range(0, n, blk):
        v = d_sa[s:s + blk].to(torch.int64) + 1
        w = torch.arange(s, s + v.numel(), device=d_sa.device, dtype=torch.int64) * 2 + 1
        acc = (acc + int((v * w).sum().item())) & MASK64
    return acc


def sa_poly64_numpy(sa: np.ndarray) -> int:
    acc, blk = 0, 1 << 24
    with np.errstate(over='ignore'):
        for s in range(0, sa.size, blk):
            v = sa[s:s + blk].astype(np.uint64) + np.uint64(1)
            w = np.arange(s, s + v.size, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
            acc = (acc + int((v * w).sum(dtype=np.uint64))) & MASK64
    return acc


def verify_sa(d_sa, host_text: np.ndarray, kind: str, chunk_index: int, goldens, want_sha: bool, host_sa=None, live_max: int = 1 << 25):
    """Is the device suffix array libsais' (the reference's) suffix array of this chunk?
    Returns (verified, how): True / False when it could be decided, None when no known answer exists
    for this input and it is too large to run libsais on the spot."""
    n = host_text.size
    g = goldens.get((kind, chunk_index, n))
    if g is not None:
        ok = sa_poly64_torch(d_sa) == g['sa_poly64']
        how = 'positional checksum vs libsais golden (tests/golden/sa_big.json)'
        if ok and want_sha:
            sa_np = host_sa if host_sa is not None else d_sa.cpu().numpy()
            ok = hashlib.sha256(sa_np.astype('<i4', copy=False)).hexdigest() == g['sa_sha256']
            how = 'sha256 of the int32 array + positional checksum vs libsais golden (tests/golden/sa_big.json)'
        return bool(ok), how
    if n <= live_max:
        from oracle import oracle as O
        if O.have_reference():
            ref = O.sa_reference(host_text)
            return bool(sa_poly64_torch(d_sa) == sa_poly64_numpy(ref) and
                        (not want_sha or np.array_equal(d_sa.cpu().numpy(), ref))), 'libsais (oracle/_ref) run on the same text'
        ref = O.sa_restatement(host_text)
        return bool(np.array_equal(d_sa.cpu().numpy(), ref)), 'oracle restatement run on the same text (libsais not built)'
    return None, 'no libsais known answer for this input (add it with tests/golden/make_golden_big.py)'


# ----------------------------------------------------------------------------- queries --

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


def sample_chunk_queries(text: np.ndarray, chunk_index: int, count: int, qmin: int, qmax: int):
    """SURVEY 8(d) config #4, the sampled half: `count` substrings of 4..32 bytes at uniform offsets of
    one chunk (no newline inside).  Seeded by the chunk index alone, so the query set does not depend
    on how the chunks are spread over ranks."""
    rng = np.random.default_rng(1000 + chunk_index)
    out = []
    while len(out) < count:
        s = int(rng.integers(0, text.size - qmax - 1))
        ln = int(rng.integers(qmin, qmax + 1))
        cand = text[s:s + ln].tobytes()
        if b'\n' not in cand:
            out.append(cand)
    return out


def mixed_queries(sampled, nq: int, qmin: int, qmax: int):
    rng = np.random.default_rng(1)
    queries = list(sampled[:nq // 2])
    while len(queries) < nq:
        ln = int(rng.integers(qmin, qmax + 1))
        queries.append(bytes(ALPHA[int(i)] for i in rng.integers(0, len(ALPHA), ln)))
    order = rng.permutation(len(queries))
    return [queries[i] for i in order]


# ----------------------------------------------------------------------------- CPU baselines --

def cpu_baseline_chunk(host: np.ndarray, queries, sample_logn: int):
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
    r = O.OracleReader.from_arrays([sample], [sa])
    b = r.bench_search(queries, 1)
    r.close()
    return {
        'value': round(m / t_sa / 1e9, 6), 'unit': 'GB/s', 'cores': 1, 'kind': kind,
        'sample': f'SA build of the first {m >> 20} MiB of the same chunk '
                  f'({"libsais from oracle/_ref" if kind == "reference" else "oracle restatement"}, 1 thread, {t_sa:.1f} s); '
                  f'queries/s = oracle Reader::search restatement, SA in RAM, 1 thread, the same {len(queries)} queries one at a time',
        'queries_per_sec': round(len(queries) / b['seconds'], 1),
        'host_cpus': os.cpu_count(),
    }


def cpu_baseline_corpus(texts, sas, queries, sample_queries: int, want_disk: bool):
    """SURVEY 8(d)(ii) on the resident corpus: queries one at a time, one thread per chunk (up to
    nproc), suffix arrays in RAM -- and, like the reference (src/lib.rs:216-217), left in the index
    file and probed with lseek + read(8 KiB).  `sas` are the GPU-built suffix arrays, each already
    verified against libsais' checksum."""
    from oracle import oracle as O
    qs = queries[:sample_queries]
    nchunks, nproc = len(texts), os.cpu_count() or 1
    threads = min(nchunks, nproc)
    out = {'unit': 'queries/s', 'kind': 'port', 'cores': threads, 'host_cpus': nproc,
           'sample': f'the first {len(qs)} queries of the batch, one at a time, fanned out over the {nchunks} chunks on '
                     f'{threads} threads (oracle/pss_oracle.c orc_bench_search: rayon par_iter_mut of src/lib.rs:207 restated)'}
    r = O.OracleReader.from_arrays(texts, sas)
    r.bench_search(qs[:max(len(qs) // 10, 1)], threads)   # (warm-up: the first pass over a 37 GiB corpus pays for cold pages and TLBs)
    b = r.bench_search(qs, threads)                       # per-chunk dedupe by a hash set, like lib.rs:262
    bs = r.bench_search(qs, threads, dedupe='sort')       # the checker's sorted scratch list (rounds 1-4 timed this one)
    assert np.array_equal(bs['counts'], b['counts'])
    hash_qps, sort_qps = len(qs) / b['seconds'], len(qs) / bs['seconds']
    out['value'] = round(max(hash_qps, sort_qps), 1)      # the better of the two: nothing here shall understate the CPU
    out['value_is'] = ('suffix arrays in RAM (kinder than the reference, which probes them on disk); the faster of two per-chunk '
                       'dedupes: a hash set like the reference\'s AHashSet (src/lib.rs:262) and the sorted scratch list of the checker')
    out['hash_dedupe_queries_per_sec'] = round(hash_qps, 1)
    out['sort_dedupe_queries_per_sec'] = round(sort_qps, 1)
    out['entries_per_query'] = round(b['entries'] / max(len(qs), 1), 2)
    b1 = r.bench_search(qs[:max(len(qs) // 10, 1)], 1)
    out['one_thread_queries_per_sec'] = round(max(len(qs) // 10, 1) / b1['seconds'], 1)
    counts = b['counts']
    r.close()
    if want_disk:
        total = sum(int(t.size) * 5 + 8 for t in texts)
        tmp = tempfile.gettempdir()
        if shutil.disk_usage(tmp).free > total + (8 << 30):
            d = tempfile.mkdtemp(prefix='pss_bench_')
            try:
                p = os.path.join(d, 'corpus.idx')
                t0 = time.perf_counter()
                with open(p, 'wb') as f:   # chunk record layout, src/lib.rs:112-119
                    for t, s in zip(texts, sas):
                        f.write(np.uint32(t.size).tobytes())
                        f.write(memoryview(t))
                        f.write(np.uint32((4 * t.size) & 0xffffffff).tobytes())
                        f.write(memoryview(s))
                t_write = time.perf_counter() - t0
                rd = O.OracleReader(p, load_sa=False)
                bd = rd.bench_search(qs, threads, disk=True)
                rd.close()
                assert np.array_equal(bd['counts'], counts)
                out['disk_queries_per_sec'] = round(len(qs) / bd['seconds'], 1)
                out['disk_is'] = (f'the reference\'s access path: suffix arrays in the {total >> 30} GiB index file (page cache hot, '
                                  f'written in {t_write:.0f} s), lseek + read(8 KiB) per probe, src/lib.rs:216-217')
            finally:
                shutil.rmtree(d, ignore_errors=True)
        else:
            out['disk_queries_per_sec'] = None
            out['disk_is'] = f'skipped: {tmp} has less than {(total >> 30) + 8} GiB free'
    return out, counts


# ----------------------------------------------------------------------------- distributed --

class Dist:
    def __init__(self, args):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.dist = None
        # Test hook for 1-GPU boxes: PSS_BENCH_BACKEND=gloo lets every rank share GPU 0
        # so the N > 1 code path (sharding, gather, max-over-ranks timing) can be exercised.
        self.backend = os.environ.get('PSS_BENCH_BACKEND', 'nccl')
        if self.backend != 'nccl':
            self.local_rank = 0
        torch.cuda.set_device(self.local_rank)
        if self.world > 1:
            import torch.distributed as dist
            self.dist = dist
            if self.backend == 'nccl':
                dist.init_process_group('nccl', device_id=torch.device('cuda', self.local_rank))
            else:
                dist.init_process_group(self.backend)
        assert self.world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={self.world}'

    def sync_all(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, values):
        t = self.torch.tensor(values, dtype=self.torch.float64, device='cuda' if self.backend == 'nccl' else 'cpu')
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.tolist()

    def all_true(self, flag):
        """True / False / None (undecided anywhere and false nowhere) over all ranks."""
        code = 1 if flag is True else (0 if flag is False else 2)
        t = self.torch.tensor([code == 0, code == 2], dtype=self.torch.float64, device='cuda' if self.backend == 'nccl' else 'cpu')
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        bad, unknown = t.tolist()
        return False if bad else (None if unknown else True)

    def gather_floats(self, value: float):
        """[value of rank 0, value of rank 1, ...] on every rank."""
        if self.world == 1:
            return [float(value)]
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device='cuda' if self.backend == 'nccl' else 'cpu')
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(x.item()) for x in out]

    def finish(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()


# ------------------------------------------------------------------------- N GPUs, one process --

def run_inproc(args):
    """`--inproc`: the N-GPU line without torch.distributed and without RCCL -- ONE process, one host thread per device
    (ctypes releases the GIL inside the library, so the N builds and searches really run side by side).  Weak scaling
    like the ranks route: device r builds its own rotating chunks of the corpus and answers the query batch against its
    resident chunk; `value` = text bytes of all devices / the slowest device's build time.  Then the two handles a user
    of the drop-in API gets on a multi-GPU node -- Writer(devices=...) and Reader(devices=...) (pss_writer_open_multi /
    pss_reader_open_multi: chunk c on device c mod N, what the reference does with rayon inside one process,
    src/lib.rs:205-207) -- are taken through a small index and compared with the one-device handles.  A second,
    independent route to an N-GPU number: bench.py falls back to it by itself when the ranks route prints no line."""
    import threading
    import torch
    import pysubstringsearch
    from pysubstringsearch_amd import Reader, _ffi
    lib = _ffi.lib
    N = args.gpus
    n = 1 << args.logn
    nq = args.queries or 10000
    visible = torch.cuda.device_count()
    devs = [r % max(visible, 1) for r in range(N)]          # fewer devices than N (test boxes): they take turns
    gold = load_big_goldens()
    ROT = 3
    state = [None] * N
    errors = []
    start = threading.Barrier(N + 1)
    stop = threading.Barrier(N + 1)

    def work(r):
        try:
            dev = devs[r]
            torch.cuda.set_device(dev)
            ids = [(r + k * N) % 15 for k in range(ROT)]
            hosts, dTs, qsets = [], [], []
            for ci in ids:
                hk = np.empty(n, dtype=np.uint8)
                _ffi.check(lib.pss_gen_corpus(KINDS[args.corpus], hk.ctypes.data, n, ci))
                hosts.append(hk)
                dTs.append(torch.from_numpy(hk).to(f'cuda:{dev}'))
                qsets.append(make_queries(hk, nq, args.qlen))
            dSA = torch.empty(n, dtype=torch.int32, device=f'cuda:{dev}')
            h = ctypes.c_void_p()
            _ffi.check(lib.pss_reader_create(dev, ctypes.byref(h)))
            reader = Reader._from_handle(h)
            st = _ffi.SaStats()
            k = 0
            tb = ts = 0.0

            def step():
                nonlocal k, tb, ts
                k = (k + 1) % ROT
                t0 = time.perf_counter()
                _ffi.check(lib.pss_sa_build_device(dTs[k].data_ptr(), dSA.data_ptr(), n, dev, 0, ctypes.byref(st)))
                t1 = time.perf_counter()
                _ffi.check(lib.pss_reader_set_chunk_device(h, 0, dTs[k].data_ptr(), dSA.data_ptr(), n))
                t2 = time.perf_counter()
                ents, _ = reader.search_batch_raw(qsets[k])
                t3 = time.perf_counter()
                return t1 - t0, t3 - t2, len(ents)
            for _ in range(args.warmup):
                step()
            torch.cuda.synchronize(dev)
            start.wait()
            entries = 0
            for _ in range(args.steps):
                b, q, entries = step()
                tb += b
                ts += q
            torch.cuda.synchronize(dev)
            stop.wait()
            ok, how = verify_sa(dSA, hosts[k], args.corpus, ids[k], gold, want_sha=False)
            state[r] = {'build_s': tb, 'search_s': ts, 'verified': ok, 'how': how, 'entries': entries, 'device': dev,
                        'ms_total': st.ms_total}
            reader.close()
        except Exception as e:      # noqa: BLE001
            errors.append(f'device thread {r}: {type(e).__name__}: {e}')
            for b in (start, stop):
                try:
                    b.abort()
                except Exception:   # noqa: BLE001
                    pass

    threads = [threading.Thread(target=work, args=(r,)) for r in range(N)]
    for t in threads:
        t.start()
    try:
        start.wait()
        t_begin = time.perf_counter()
        stop.wait()
        total_s = time.perf_counter() - t_begin
    except threading.BrokenBarrierError:
        total_s = float('nan')
    for t in threads:
        t.join()
    if errors or any(x is None for x in state):
        return 1, {'metric': METRIC, 'value': None, 'unit': 'GB/s', 'n_gpus': N, 'route': 'inproc', 'error': '; '.join(errors)[:600]}
    build_s = max(x['build_s'] for x in state)
    search_s = max(x['search_s'] for x in state)
    verified = False if any(x['verified'] is False for x in state) else (None if any(x['verified'] is None for x in state) else True)

    # the multi-device handles of the drop-in API on a small index: same bytes, same results as the one-device handles
    handles = None
    try:
        import shutil
        import tempfile
        d = tempfile.mkdtemp(prefix='pss_inproc_')
        try:
            cn = 1 << min(args.logn, 22)
            src = os.path.join(d, 'c.txt')
            with open(src, 'wb') as f:
                for c in range(2 * N + 1):
                    buf = np.empty(cn, dtype=np.uint8)
                    _ffi.check(lib.pss_gen_corpus(KINDS[args.corpus], buf.ctypes.data, cn, c))
                    f.write(buf.data)
            one, many = os.path.join(d, 'one.idx'), os.path.join(d, 'many.idx')
            w = pysubstringsearch.Writer(one, cn, device=devs[0])
            w.add_entries_from_file_lines(src)
            w.close()
            t0 = time.perf_counter()
            w = pysubstringsearch.Writer(many, cn, devices=devs)
            w.add_entries_from_file_lines(src)
            w.close()
            t_w = time.perf_counter() - t0
            same_bytes = pathlib.Path(one).read_bytes() == pathlib.Path(many).read_bytes()
            qs = make_queries(np.fromfile(src, dtype=np.uint8, count=cn), 3000, args.qlen)
            with pysubstringsearch.Reader(one, device=devs[0]) as r1, pysubstringsearch.Reader(many, devices=devs) as rn:
                e1, c1 = r1.search_batch_raw(qs)
                t0 = time.perf_counter()
                en, cn_ = rn.search_batch_raw(qs)
                t_r = time.perf_counter() - t0
                same_results = list(c1) == list(cn_) and sorted(e1) == sorted(en)
            handles = {'devices': devs, 'chunks': 2 * N + 1, 'chunk_bytes': cn, 'writer_same_bytes_as_one_device': bool(same_bytes),
                       'reader_same_results_as_one_device': bool(same_results), 'writer_seconds': round(t_w, 3),
                       'reader_batch_seconds': round(t_r, 4)}
            if not (same_bytes and same_results):
                verified = False
        finally:
            shutil.rmtree(d, ignore_errors=True)
    except Exception as e:      # noqa: BLE001
        handles = {'error': f'{type(e).__name__}: {e}'[:300]}
    value = None if verified is False else round(N * n * args.steps / build_s / 1e9, 4)
    out = {
        'metric': METRIC, 'value': value, 'unit': 'GB/s', 'n_gpus': N, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(total_s / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'u64', 'data': 'synthetic', 'route': 'inproc', 'verified': verified,
        'verified_by': state[0]['how'],
        'config': {'workload': f'configs[1] on {N} GPUs inside ONE process (one host thread per device, no torch.distributed, no RCCL): '
                               f'one {n >> 20} MiB synthetic `{args.corpus}` chunk per GPU, suffix-array build + {nq} '
                               f'{args.qlen}-byte queries against the device\'s own chunk',
                   'corpus': args.corpus, 'chunk_bytes': n, 'queries': nq, 'query_len': args.qlen,
                   'devices': devs, 'visible_devices': visible,
                   'value_is': 'index-build GB/s (text bytes of all devices / the slowest device\'s build time, inputs resident in HBM)'},
        'summary': {'index_build_gbs_warm': value, 'per_device_build_ms': [round(x['build_s'] / args.steps * 1e3, 3) for x in state],
                    'queries_per_sec_all_devices': round(N * nq * args.steps / search_s, 1), 'route': 'inproc'},
        'build_ms': round(build_s / args.steps * 1e3, 3), 'search_ms': round(search_s / args.steps * 1e3, 3),
        'per_device': state, 'multi_device_handles': handles,
    }
    return (1 if verified is False else 0), out


# --------------------------------------------------------------------------------- file API --

def run_e2e(args):
    """What a user of `Writer` / `Reader` sees: a text file of `--e2e-chunks` chunks goes through
    Writer.add_entries_from_file_lines into an .idx file (ingest, upload, suffix-array build, download, record write --
    src/lib.rs:67-124), then Reader opens it (src/lib.rs:162-199: here text AND suffix arrays become resident).  Wall
    clock around the calls, files in the page cache of PSS_BENCH_E2E_DIR (default: the system's temp directory).  The
    records of the first and the last chunk are checked against libsais' sha256 (tests/golden/sa_big.json)."""
    import hashlib
    import shutil
    import tempfile
    import pysubstringsearch
    from pysubstringsearch_amd import _ffi
    n = 1 << args.logn
    chunks = args.e2e_chunks
    where = os.environ.get('PSS_BENCH_E2E_DIR') or tempfile.gettempdir()
    d = tempfile.mkdtemp(prefix='pss_e2e_', dir=where)
    try:
        src, idx = os.path.join(d, 'corpus.txt'), os.path.join(d, 'out.idx')
        with open(src, 'wb') as f:
            for c in range(chunks):
                buf = np.empty(n, dtype=np.uint8)
                _ffi.check(_ffi.lib.pss_gen_corpus(KINDS[args.corpus], buf.ctypes.data, n, c))
                f.write(buf.data)
        del buf
        best, first = None, None
        for rep in range(2):                 # the first pass also grows the workspaces and pins the staging rings
            if os.path.exists(idx):
                os.remove(idx)
            t0 = time.perf_counter()
            w = pysubstringsearch.Writer(idx, n)
            w.add_entries_from_file_lines(src)
            w.finalize()
            w.close()
            t1 = time.perf_counter()
            r = pysubstringsearch.Reader(idx)
            t2 = time.perf_counter()
            nchunks, devices = r.num_chunks, list(r.devices)
            r.close()
            if first is None:
                first = (t1 - t0, t2 - t1)
            if best is None or t1 - t0 < best[0]:
                best = (t1 - t0, t2 - t1)
        sz = os.path.getsize(idx)
        # the file itself: lengths of every record, libsais' sha256 for the first and the last chunk
        gold = load_big_goldens()
        ok, checked = sz == chunks * (8 + 5 * n), []
        with open(idx, 'rb') as f:
            for c in range(chunks):
                at = c * (8 + 5 * n)
                f.seek(at)
                if int.from_bytes(f.read(4), 'little') != n:
                    ok = False
                f.seek(at + 4 + n)
                if int.from_bytes(f.read(4), 'little') != (4 * n) & 0xffffffff:
                    ok = False
                g = gold.get((args.corpus, c, n))
                if g is not None and c in (0, chunks - 1):
                    f.seek(at + 4)
                    ht = hashlib.sha256(f.read(n)).hexdigest()
                    f.seek(at + 8 + n)
                    h = hashlib.sha256()
                    left = 4 * n
                    while left:
                        b = f.read(min(left, 1 << 26))
                        h.update(b)
                        left -= len(b)
                    ok = ok and ht == g['text_sha256'] and h.hexdigest() == g['sa_sha256']
                    checked.append(c)
        # the same text into container format 2 with the suffix arrays striped over eight files of their own
        # (Writer(..., format_version=2, striped=True), include/pss.h PSS_FORMAT_STRIPED): one file in the page cache is what
        # bounds the reference container on this box (profiles/r04_pagecache_micro.txt)
        striped = None
        try:
            os.remove(idx)                   # (checked above; the two indexes need not share the disk)
            sidx = os.path.join(d, 'striped.idx')
            sbest = None
            for rep in range(2):
                for f in [sidx] + [f'{sidx}.sa{j}' for j in range(64)]:
                    if os.path.exists(f):
                        os.remove(f)
                t0 = time.perf_counter()
                w = pysubstringsearch.Writer(sidx, n, format_version=2, striped=True)
                w.add_entries_from_file_lines(src)
                w.finalize()
                w.close()
                t1 = time.perf_counter()
                r = pysubstringsearch.Reader(sidx)
                t2 = time.perf_counter()
                s_chunks = r.num_chunks
                r.close()
                if sbest is None or t1 - t0 < sbest[0]:
                    sbest = (t1 - t0, t2 - t1)
            S, unit = 8, 1 << 24
            s_ok, s_checked = s_chunks == chunks, []
            units_per_chunk = (4 * n + unit - 1) // unit
            for c in (0, chunks - 1):
                g = gold.get((args.corpus, c, n))
                if g is None:
                    continue
                h = hashlib.sha256()
                left = 4 * n
                for u in range(c * units_per_chunk, (c + 1) * units_per_chunk):
                    with open(f'{sidx}.sa{u % S}', 'rb') as f:
                        f.seek((u // S) * unit)
                        b = f.read(min(left, unit))
                    h.update(b)
                    left -= len(b)
                s_ok = s_ok and left == 0 and h.hexdigest() == g['sa_sha256']
                s_checked.append(c)
            s_sz = os.path.getsize(sidx) + sum(os.path.getsize(f'{sidx}.sa{j}') for j in range(S))
            striped = {'writer_seconds': round(sbest[0], 3), 'writer_text_gbs': round(chunks * n / sbest[0] / 1e9, 3),
                       'writer_idx_gbs': round(s_sz / sbest[0] / 1e9, 2), 'reader_open_seconds': round(sbest[1], 3),
                       'reader_open_idx_gbs': round(s_sz / sbest[1] / 1e9, 2), 'stripe_files': S, 'bytes': s_sz,
                       'verified': bool(s_ok), 'verified_by': f'suffix-array sha256 of chunks {s_checked}, read back from the stripe files, '
                                                              'against libsais (tests/golden/sa_big.json)'}
        except Exception as e:      # noqa: BLE001
            striped = {'error': f'{type(e).__name__}: {e}'[:300]}
        return {'chunks': chunks, 'chunk_bytes': n, 'dir': where, 'text_bytes': chunks * n, 'idx_bytes': sz, 'striped_format_2': striped,
                'writer_seconds': round(best[0], 3), 'writer_text_gbs': round(chunks * n / best[0] / 1e9, 3),
                'writer_idx_gbs': round(sz / best[0] / 1e9, 2),
                'writer_seconds_first': round(first[0], 3), 'writer_text_gbs_first': round(chunks * n / first[0] / 1e9, 3),
                'reader_open_seconds': round(best[1], 3), 'reader_open_idx_gbs': round(sz / best[1] / 1e9, 2),
                'reader_chunks': nchunks, 'devices': devices,
                'verified': bool(ok), 'verified_by': f'record lengths of every chunk; text and suffix-array sha256 of chunks {checked} '
                                                     'against libsais (tests/golden/sa_big.json)',
                'what': 'wall clock of Writer(path, chunk).add_entries_from_file_lines(text file) + finalize + close, then of '
                        'Reader(path); best of 2 (the first Writer of a process also allocates its staging and text buffers: `_first`); '
                        'both files in the page cache'}
    finally:
        shutil.rmtree(d, ignore_errors=True)


# ----------------------------------------------------------------------------- configs[1] --

def run_chunk(args, D):
    import torch
    from pysubstringsearch_amd import Reader, _ffi
    from pysubstringsearch_amd import dist as pdist
    lib = _ffi.lib
    rank, world, dist = D.rank, D.world, D.dist
    n = 1 << args.logn
    dev = D.local_rank
    nq = args.queries or 10000

    # The timed steps ROTATE over distinct chunks of the corpus (round 4): a Writer never builds the same chunk twice,
    # and a build that follows a build of the same kind of text runs under the remembered plan (alphabet + choice of
    # sort, pss_sa_stats.plan_hint) -- what chunks 2..N of a corpus see.  Rotating keeps that honest: nothing about a
    # chunk's CONTENT can be remembered from step to step.  The build without any plan (a first chunk) is timed below
    # and reported beside it (`value_cold`).
    ROT = max(1, min(3, int(os.environ.get('PSS_BENCH_ROTATE', '3'))))
    chunk_ids = [(rank + k * world) % 15 for k in range(ROT)]
    hosts, dTs, qsets = [], [], []
    for ci in chunk_ids:
        hk = np.empty(n, dtype=np.uint8)
        _ffi.check(lib.pss_gen_corpus(KINDS[args.corpus], hk.ctypes.data, n, ci))
        hosts.append(hk)
        dTs.append(torch.from_numpy(hk).cuda())
    dSA = torch.empty(n, dtype=torch.int32, device='cuda')
    for k in range(ROT):
        if world > 1:
            box = [make_queries(hosts[k], nq, args.qlen) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            qsets.append(box[0])
        else:
            qsets.append(make_queries(hosts[k], nq, args.qlen))
    torch.cuda.synchronize()
    cur = {'k': 0}

    st = _ffi.SaStats()
    last = {}
    h = ctypes.c_void_p()
    _ffi.check(lib.pss_reader_create(dev, ctypes.byref(h)))
    reader = Reader._from_handle(h)   # device-resident index of this rank, refreshed every step

    def step(flags=0):
        k = cur['k'] = (cur['k'] + 1) % ROT
        dT, queries = dTs[k], qsets[k]
        t0 = time.perf_counter()
        _ffi.check(lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, dev, flags, ctypes.byref(st)))
        t1 = time.perf_counter()
        # Writer -> Reader hand-off through HBM (no file): the fresh text + SA replace the resident chunk
        _ffi.check(lib.pss_reader_set_chunk_device(h, 0, dT.data_ptr(), dSA.data_ptr(), n))
        t2 = time.perf_counter()
        if world > 1:
            # local result stays on the device -> gather to rank 0 -> one packed result there
            merged = pdist.gather_device(reader.search_batch_device(queries), dst=0)
            entries = merged[1][:-1] if merged is not None else []      # one offset per entry (packed result)
        else:
            entries, counts = reader.search_batch_raw(queries)
        t3 = time.perf_counter()
        last['entries'] = len(entries)
        last['search_stats'] = reader.last_stats()
        return t1 - t0, t3 - t2

    for _ in range(args.warmup):
        step()
    D.sync_all()
    t_begin = time.perf_counter()
    build_s = search_s = 0.0
    for _ in range(args.steps):
        b, s = step()
        build_s += b
        search_s += s
    D.sync_all()
    total_s = time.perf_counter() - t_begin
    sa_stats = st.as_dict()
    per_rank_build_ms = [round(x / args.steps * 1e3, 3) for x in D.gather_floats(build_s)]        # (collective: every rank)
    per_rank_search_ms = [round(x / args.steps * 1e3, 3) for x in D.gather_floats(search_s)]
    build_s, search_s, total_s = D.max_over_ranks([build_s, search_s, total_s])

    # the suffix array the last timed step left in dSA: is it the reference's?
    host, dT, queries = hosts[cur['k']], dTs[cur['k']], qsets[cur['k']]
    host_sa = dSA.cpu().numpy() if world == 1 else None
    verified, how = verify_sa(dSA, host, args.corpus, chunk_ids[cur['k']], load_big_goldens(), want_sha=(world == 1), host_sa=host_sa)
    verified = D.all_true(verified)

    # ... and are the results of the query leg the reference's?  A sample of the batch (sampled and random queries alike)
    # through the oracle's restatement of Reader::search (src/lib.rs:209-278) over the same text and the suffix array
    # just verified, compared per query as multisets (the reference's tests use assertCountEqual, tests:32-37).
    verified_search = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        pick = list(range(0, len(queries), max(1, len(queries) // 200)))
        qs = [queries[i] for i in pick]
        got, got_counts = reader.search_batch_raw(qs)
        o = O.OracleReader.from_arrays([host], [host_sa])
        exp, exp_counts = o.search_multiple_bytes(qs)
        o.close()
        ok, pos_g, pos_e = list(got_counts) == [int(c) for c in exp_counts], 0, 0
        for cg, ce in zip(got_counts, exp_counts):
            if not ok:
                break
            ok = sorted(got[pos_g:pos_g + cg]) == sorted(exp[pos_e:pos_e + int(ce)])
            pos_g += cg
            pos_e += int(ce)
        verified_search = {'ok': bool(ok), 'queries': len(qs), 'entries': int(sum(got_counts)),
                           'by': 'oracle Reader::search restatement on the same text + verified suffix array, per-query multisets'}
        if not ok:
            verified = False
    del host_sa

    # roofline of the dominant kernel: one extra build in profile mode (HIP events
    # on the engine's own stream around every radix-pass launch), outside the timed region
    _ffi.check(lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, dev, 1, ctypes.byref(st)))
    prof = st.as_dict()

    # The timed steps rebuild chunks of one corpus, so from the second on the build goes straight to the initial sort the
    # previous chunk took (pss_sa_stats.plan_hint).  The build of a first chunk, which takes the sizing sample, beside it:
    # (flags bit 3: the build forgets what earlier builds on the device left behind -- include/pss.h)
    sized_ms = None
    cold_hint = None
    for _ in range(3):
        _ffi.check(lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, dev, 8, ctypes.byref(st)))
        sized_ms = st.ms_total if sized_ms is None else min(sized_ms, st.ms_total)
        cold_hint = int(st.plan_hint)

    # What the builder's workspace holds after the builds above (grow-only slots: the high-water mark of this corpus'
    # route), and Seam 1: the libsais-shaped call with HOST pointers, which pays the text up (n bytes) and
    # the suffix array down (4 n bytes) around the build -- outside the timed region, rank 0, N = 1
    workspace = {'lines' if args.corpus == 'lines' else args.corpus: int(lib.pss_workspace_bytes(dev))}
    seam1 = None
    if world == 1 and not os.environ.get('PSS_BENCH_NO_SECONDARY'):
        def host_build(t_ptr, sa_ptr):
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                rc = lib.pss_sa_build(t_ptr, sa_ptr, n, dev)
                dt = (time.perf_counter() - t0) * 1e3
                _ffi.check(rc)
                best = dt if best is None else min(best, dt)
            return best
        try:
            h_sa = np.empty(n, dtype=np.int32)
            ms_pageable = host_build(host.ctypes.data, h_sa.ctypes.data)
            ok_page = bool(sa_poly64_torch(torch.from_numpy(h_sa[:n]).cuda()) == sa_poly64_torch(dSA)) if verified is not False else None
            del h_sa
            p_t = torch.from_numpy(host).pin_memory()
            p_sa = torch.empty(n, dtype=torch.int32).pin_memory()
            ms_pinned = host_build(p_t.data_ptr(), p_sa.data_ptr())
            del p_t, p_sa
            seam1 = {'call': 'pss_sa_build(T, SA, n, device) with host pointers (the drop-in for libsais(T, SA, n, 0, NULL), src/lib.rs:30-36)',
                     'build_ms_host_pointers_pageable': round(ms_pageable, 2), 'build_ms_host_pointers_pinned': round(ms_pinned, 2),
                     'same_suffix_array_as_the_device_call': ok_page,
                     'bytes_over_pcie': 5 * n,
                     'pcie_gbs_pinned': round(5 * n / max(ms_pinned - build_s / args.steps * 1e3, 1e-3) / 1e6, 1),
                     'note': 'n bytes up + 4 n bytes down around a build of build_ms_warm: the call is bound by the host link '
                             '(2.68 GB per 512 MiB chunk), not by the build; the Writer hides it behind the next chunk (e2e leg)'}
        except Exception as e:                       # (no room for 2 GiB of pinned memory: the figure is simply absent)
            seam1 = {'error': repr(e)[:200]}

    # secondary corpus (natural-text-like LCP), outside the timed region, N = 1 only
    secondary = None
    if world == 1 and args.corpus == 'lines' and not os.environ.get('PSS_BENCH_NO_SECONDARY'):
        # three distinct chunks of the corpus in turn, like the headline: chunk k+1 is cut by the splitters chunk k's sample
        # gave (pss_sa_stats.ss_planned), nothing about a chunk's own content is carried over; a first chunk beside it
        w_dTs, w_hosts = [], []
        for ci in range(3):
            w_host = np.empty(n, dtype=np.uint8)
            _ffi.check(lib.pss_gen_corpus(KINDS['words'], w_host.ctypes.data, n, ci))
            w_hosts.append(w_host)
            w_dTs.append(torch.from_numpy(w_host).cuda())
        wst = _ffi.SaStats()
        w_cold = None
        for _ in range(2):
            _ffi.check(lib.pss_sa_build_device(w_dTs[0].data_ptr(), dSA.data_ptr(), n, dev, 8, ctypes.byref(wst)))
            w_cold = wst.ms_total if w_cold is None else min(w_cold, wst.ms_total)
        w_ms, w_planned, w_ok, w_how = [], 0, True, ''
        for i in range(7):
            ci = (i + 1) % 3
            _ffi.check(lib.pss_sa_build_device(w_dTs[ci].data_ptr(), dSA.data_ptr(), n, dev, 0, ctypes.byref(wst)))
            if i:                       # (the first follows a cold build of another chunk: counted from the second on)
                w_ms.append(wst.ms_total)
                w_planned += int(wst.ss_planned)
            if i >= 4:                  # every one of the three chunks once, as built under its predecessor's plan
                ok_i, w_how = verify_sa(dSA, w_hosts[ci], 'words', ci, load_big_goldens(), want_sha=False)
                w_ok = None if (ok_i is None or w_ok is None) else (w_ok and ok_i)
        best = sum(w_ms) / len(w_ms)
        workspace['words'] = int(lib.pss_workspace_bytes(dev))
        del w_dTs, w_hosts
        wd = wst.as_dict()
        w_traffic, w_stamp = None, None
        if args.logn == 29:
            wj, w_stamp = evidence('pmc_build_traffic_words.json')
            w_traffic = wj.get('total_bytes') if wj else None
        secondary = {'corpus': 'words', 'chunk_bytes': n, 'build_ms': round(best, 3),
                     'build_ms_is': 'mean of 6 builds rotating over 3 distinct chunks, each under the plan its predecessor left',
                     'build_ms_min': round(min(w_ms), 3), 'build_ms_max': round(max(w_ms), 3),
                     'build_ms_cold': round(w_cold, 3), 'planned_builds': w_planned,
                     'index_build_gbs': round(n / best / 1e6, 4), 'verified': w_ok, 'verified_by': w_how,
                     'traffic': w_traffic, 'traffic_gbs': None if not w_traffic else round(w_traffic / best / 1e6, 1),
                     'traffic_evidence': w_stamp,
                     'initial_sort': ('sample sort over 16-byte [key | index] elements (ss_sort_impl.h)' if wd['ss'] else
                                      'LSD passes with shrinking keys'),
                     'initial_sort_ms': round(wd['ms_initial'], 3),
                     'sa_stats': {k: wd[k] for k in ('key_chars', 'initial_passes', 'rounds', 'text_rounds', 'round_passes',
                                                     'sum_active', 'big_elems', 'mode', 'ss', 'ss_buckets', 'ss_max_bucket',
                                                     'ss_tiles', 'ss_samples')}}

    # configs[4]: the adversarial corpora (long runs of equal bytes; period 4096), outside the timed region,
    # N = 1 only: the run-length path (rle_build.hip), and once the prefix-doubling path it replaces
    adversarial = None
    if world == 1 and args.corpus == 'lines' and not os.environ.get('PSS_BENCH_NO_SECONDARY'):
        adversarial = []
        for kind in ('runs', 'periodic', 'repeat_line', 'dup_blocks', 'mixed', 'source'):
            a_host = np.empty(n, dtype=np.uint8)
            _ffi.check(lib.pss_gen_corpus(KINDS[kind], a_host.ctypes.data, n, 0))
            a_dT = torch.from_numpy(a_host).cuda()
            ast = _ffi.SaStats()
            best = None
            for _ in range(3 if kind in ('runs', 'periodic', 'source') else 2):
                _ffi.check(lib.pss_sa_build_device(a_dT.data_ptr(), dSA.data_ptr(), n, dev, 0, ctypes.byref(ast)))
                best = ast.ms_total if best is None else min(best, ast.ms_total)
            ad = ast.as_dict()
            workspace[kind] = int(lib.pss_workspace_bytes(dev))
            a_ok, a_how = verify_sa(dSA, a_host, kind, 0, load_big_goldens(), want_sha=False)
            if kind in ('repeat_line', 'dup_blocks', 'mixed', 'source'):
                # repeats that are not runs of one byte: one word repeated (closed form), duplicated blocks, natural
                # text with a repetitive middle, source-like text (anchor round: anchor_impl.h)
                adversarial.append({'corpus': kind, 'chunk_bytes': n, 'build_ms': round(best, 3),
                                    'index_build_gbs': round(n / best / 1e6, 3), 'verified': a_ok, 'verified_by': a_how,
                                    'run_length_path': bool(ad['rle']), 'rounds': ad['rounds'], 'sum_active': ad['sum_active'],
                                    'initial_sort': 'sample sort' if ad['ss'] else ('hybrid MSD' if ad['msd'] else 'LSD'),
                                    'anchor_round': bool(ad['anchor']), 'anchors': ad['anchor_count'],
                                    'anchor_window': ad['anchor_omega'], 'anchor_rank_rounds': ad['anchor_rounds'],
                                    'anchor_ms': round(ad['anchor_ms'], 2), 'anchor_left_tied': ad['anchor_left'],
                                    'anchors_beside_the_text_round': int(ad['anchor_side']), 'text_rounds': ad['text_rounds']})
                if kind == 'source':
                    sj, s_stamp = evidence('pmc_build_traffic_source.json')
                    if sj and args.logn == 29:
                        adversarial[-1].update({'traffic': sj.get('total_bytes'), 'traffic_gbs': round(sj['total_bytes'] / best / 1e6, 1),
                                                'traffic_evidence': s_stamp})
                del a_dT
                continue
            os.environ['PSS_RLE'] = '0'
            try:
                pd_ms = None
                for _ in range(2):         # the first build grows the workspace of the rank rounds (allocation inside the events)
                    _ffi.check(lib.pss_sa_build_device(a_dT.data_ptr(), dSA.data_ptr(), n, dev, 0, ctypes.byref(ast)))
                    pd_ms = ast.ms_total if pd_ms is None else min(pd_ms, ast.ms_total)
            finally:
                del os.environ['PSS_RLE']
            d_ok, _ = verify_sa(dSA, a_host, kind, 0, load_big_goldens(), want_sha=False)
            adversarial.append({'corpus': kind, 'chunk_bytes': n, 'build_ms': round(best, 3),
                                'index_build_gbs': round(n / best / 1e6, 3), 'verified': a_ok, 'verified_by': a_how,
                                'run_length_path': bool(ad['rle']), 'runs': ad['runs'], 'reduced_rounds': ad['rounds'],
                                'expansion_key_bits': ad['rle_id_bits'],
                                'prefix_doubling_ms': round(pd_ms, 1), 'prefix_doubling_rounds': ast.rounds,
                                'prefix_doubling_verified': d_ok})
            del a_dT

    # Real files instead of generated text (N = 1 only): 256 MiB (--real-files-logn) of the Python sources and C / HIP headers
    # found on this machine, in sorted path order (tests/tools/real_text.py) -- licence headers copied thousands of times,
    # indentation, 200-odd byte values; four suffixes in five sit inside copies, so the build goes through the sample
    # sort, a text round and the anchor round.  Checked against libsais run on the same bytes, here and now.
    real_files = None
    if world == 1 and args.corpus == 'lines' and not os.environ.get('PSS_BENCH_NO_SECONDARY') and not args.no_real_files:
        try:
            import importlib.util
            spec = importlib.util.spec_from_file_location('real_text', os.path.join(ROOT, 'tests', 'tools', 'real_text.py'))
            rt = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(rt)
            t_c = time.perf_counter()
            raw = rt.collect(1 << args.real_files_logn)
            collect_s = time.perf_counter() - t_c
            if len(raw) >= (1 << 24):
                r_host = np.frombuffer(raw, dtype=np.uint8).copy()
                r_host[-1] = 10
                rn = int(r_host.size)
                r_dT = torch.from_numpy(r_host).cuda()
                r_dSA = dSA[:rn] if rn <= n else torch.empty(rn, dtype=torch.int32, device='cuda')
                rst = _ffi.SaStats()
                best = None
                r_cold = None
                for i in range(4):
                    _ffi.check(lib.pss_sa_build_device(r_dT.data_ptr(), r_dSA.data_ptr(), rn, dev, 8 if i == 0 else 0, ctypes.byref(rst)))
                    if i == 0:
                        r_cold = rst.ms_total
                    else:
                        best = rst.ms_total if best is None else min(best, rst.ms_total)
                rd = rst.as_dict()
                r_ok, r_how = verify_sa(r_dSA, r_host, 'real_files', 0, {}, want_sha=True, live_max=1 << 29)
                rj, r_stamp = evidence('pmc_build_traffic_real.json')
                real_files = {'bytes': rn, 'byte_values': int(len(np.unique(r_host))), 'lines': int((r_host == 10).sum()),
                              'build_ms': round(best, 3), 'index_build_gbs': round(rn / best / 1e6, 3),
                              'verified': r_ok, 'verified_by': r_how, 'collect_seconds': round(collect_s, 1),
                              'initial_sort': 'sample sort' if rd['ss'] else ('hybrid MSD' if rd['msd'] else 'LSD'),
                              'anchor_round': bool(rd['anchor']), 'anchors': rd['anchor_count'],
                              'tied_after_initial_sort': rd['sum_active'], 'text_rounds': rd['text_rounds'],
                              'build_ms_cold': round(r_cold, 3), 'anchors_beside_the_text_round': int(rd['anchor_side']),
                              'anchor_ms': round(rd['anchor_ms'], 2), 'initial_sort_ms': round(rd['ms_initial'], 2),
                              'a_min_frac': round(5 * rn / best / 1e6 / HBM_PEAK_GBS, 4),
                              'traffic_per_suffix_committed': None if not rj else rj.get('bytes_per_suffix'),
                              'traffic_evidence': r_stamp}
                del r_dT, r_dSA
        except Exception as e:                       # (a machine without such files, or without the tool)
            real_files = {'error': repr(e)[:200]}

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
        if prof['msd']:
            # hybrid MSD initial sort (msd_sort.hip): partition from the text (1 B in, 8 B out per suffix), partition of
            # the 8-byte elements (8 in, 8 out), LDS-resident local sort (8 in, 4 out)
            lb = bool(prof.get('msd_lookback'))      # round 6: digits in LSD order, the second pass in one sweep (look-back)
            cands.append((prof['msd_ms_g1'], 'msd_scatter2_kernel<true, 1024, true>' if lb else 'msd_scatter2_kernel<true, 1024>', 1, n, 9))
            cands.append((prof['msd_ms_g2'], 'msd_scatter_lb_kernel' if lb else 'msd_scatter2_kernel<false, 1024>', 1, n, 16))
            cands.append((prof['msd_ms_local'], 'msd_local_fast_kernel', 1, n, 12))
        roof = None
        if cands:
            ms_sum, kname, launches, elems, bpe = max(cands)
            bytes_per_launch = float(bpe) * elems / launches
            ms_per_launch = ms_sum / launches
            achieved = bytes_per_launch / (ms_per_launch * 1e-3) / 1e9
            traffic, t_stamp = None, None
            if args.corpus == 'lines' and args.logn == 29:
                tj, t_stamp = evidence('pmc_traffic.json')
                traffic = tj.get('kernels', {}).get(kname, {}).get('bytes_per_launch') if tj else None
            roof = {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                    'traffic_source': None if traffic is None else
                    'profiles/pmc_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this workload '
                    '(FETCH x 2 on gfx950), committed with the code -- a constant of the build, not measured in this run',
                    'traffic_evidence': t_stamp,
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
        if sa_stats['msd']:
            # MSD initial sort: (1 hist + 1 + 8) + (8 hist + 8 + 8) + (8 + 4) bytes per suffix; the first rerank is fused
            # into the local sort (no pass over the flagged array).  With the look-back pass (round 6) the second
            # histogram read is gone: (1 + 1 + 8) + (8 + 8) + (8 + 4) = 38
            a_model = 3 * n + (38 if sa_stats.get('msd_lookback') else 46) * n + 80 * sa_stats['sum_active']
        measured, m_stamp = None, None
        if args.corpus == 'lines' and args.logn == 29:
            mj, m_stamp = evidence('pmc_build_traffic.json')
            measured = mj.get('total_bytes') if mj else None
        # SURVEY 8(d)'s yardstick (64-bit keys in every pass, P = 8): 5 n + sum over rounds of n_r (32 P + 44)
        a_survey = 5 * n + 300 * (n + sa_stats['sum_active'])
        build_roof = {'a_min_bytes': 5 * n, 'a_model_bytes': int(a_model),
                      'survey_model_bytes': int(a_survey), 'survey_model_frac': round(a_survey / build_ms / 1e6 / HBM_PEAK_GBS, 4),
                      'effective_gbs_a_min': round(5 * n / build_ms / 1e6, 1),
                      'achieved': round(a_model / build_ms / 1e6, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                      'frac': round(a_model / build_ms / 1e6 / HBM_PEAK_GBS, 4), 'traffic': measured,
                      'traffic_source': None if measured is None else 'profiles/pmc_build_traffic.json (committed PMC runs, not measured in this run)',
                      'traffic_evidence': m_stamp,
                      'passes': sa_stats['initial_passes'], 'rounds': sa_stats['rounds'], 'sum_active': sa_stats['sum_active']}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline_chunk(host, queries, args.cpu_sample_logn)
        value = None if verified is False else round(world * n * args.steps / build_s / 1e9, 4)
        value_cold = None if verified is False else round(world * n / sized_ms / 1e6, 4)

        def pick(obj, *path):
            for k in path:
                if obj is None:
                    return None
                obj = obj[k] if not isinstance(obj, list) else next((x for x in obj if x.get('corpus') == k), None)
            return obj
        out = {
            'metric': METRIC,
            # a build whose suffix array is not the reference's has no throughput worth reporting
            'value': value,
            'unit': 'GB/s',
            # the figures a reader of the line needs first, flat (the legs below carry the detail; corpus15 / e2e are
            # filled in by the legs that run after this one)
            'summary': {
                'index_build_gbs_warm': value, 'index_build_gbs_cold': value_cold,
                'build_ms_warm': round(build_s / args.steps * 1e3, 3), 'build_ms_first_chunk': round(sized_ms, 3),
                'distinct_chunks_in_timed_loop': ROT,
                'dominant_kernel': None if roof is None else roof['kernel'], 'roofline_frac': None if roof is None else roof['frac'],
                'build_roofline_frac': build_roof['frac'],
                'queries_per_sec_1chunk': round(len(queries) * args.steps / search_s, 1),
                'cpu_libsais_gbs': pick(cpu, 'value'),
                'words_build_ms': pick(secondary, 'build_ms'),
                'dup_blocks_build_ms': pick(adversarial, 'dup_blocks', 'build_ms'),
                'mixed_build_ms': pick(adversarial, 'mixed', 'build_ms'),
                'runs_build_ms': pick(adversarial, 'runs', 'build_ms'),
                'periodic_build_ms': pick(adversarial, 'periodic', 'build_ms'),
                'repeat_line_build_ms': pick(adversarial, 'repeat_line', 'build_ms'),
                'source_build_ms': pick(adversarial, 'source', 'build_ms'),
                'real_files_build_gbs': pick(real_files, 'index_build_gbs') if real_files and 'error' not in real_files else None,
                'real_files_bytes': pick(real_files, 'bytes') if real_files and 'error' not in real_files else None,
                'build_ms_host_pointers_pinned': pick(seam1, 'build_ms_host_pointers_pinned') if seam1 and 'error' not in seam1 else None,
                'build_ms_host_pointers_pageable': pick(seam1, 'build_ms_host_pointers_pageable') if seam1 and 'error' not in seam1 else None,
                'workspace_bytes': workspace,
                'workspace_bytes_is': 'pss_workspace_bytes(device) after the builds of each corpus, in the order they ran in this '
                                      'process (grow-only slots: a later figure includes what the earlier routes reserved)',
            },
            'seam1_host_pointers': seam1,
            'value_cold': value_cold,
            'value_is_warm': 'steady state of a Writer: %d distinct chunks rotate through the timed loop, every build after '
                             'the first runs under the remembered plan (alphabet + choice of sort); value_cold = a build with '
                             'no plan at all (first chunk of a corpus)' % ROT,
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(total_s / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'u64', 'data': 'synthetic',
            'verified': verified, 'verified_by': how, 'verified_search': verified_search,
            'config': {
                'workload': f'configs[1]: one {n >> 20} MiB synthetic `{args.corpus}` chunk per GPU, suffix-array build + '
                            f'{len(queries)} {args.qlen}-byte queries (50% sampled from the text) in one batch',
                'corpus': args.corpus, 'chunk_bytes': n, 'queries': len(queries), 'query_len': args.qlen,
                'value_is': 'index-build GB/s (text bytes of all ranks / suffix-array build time, inputs resident in HBM)',
            },
            'queries_per_sec': round(len(queries) * args.steps / search_s, 1),
            'build_ms': round(build_s / args.steps * 1e3, 3),
            'build_ms_first_chunk': round(sized_ms, 3), 'plan_hint': int(sa_stats.get('plan_hint', 0)), 'plan_hint_first_chunk': cold_hint,
            'search_ms': round(search_s / args.steps * 1e3, 3),
            'per_rank_build_ms': per_rank_build_ms, 'per_rank_search_ms': per_rank_search_ms,
            'ranks': world, 'gather': None if world == 1 else f'{D.backend} (torch.distributed P2P, device-side merge on rank 0)',
            'entries_per_batch': last.get('entries'),
            'search_stats': last.get('search_stats'),
            'sa_stats': {k: sa_stats[k] for k in ('sigma', 'code_bits', 'key_chars', 'initial_passes', 'rounds',
                                                  'round_passes', 'sum_active', 'sort_launches', 'mode', 'key_bits', 'ms_total',
                                                  'msd', 'msd_buckets', 'msd_max_bucket', 'msd_tiles', 'msd_slow_tiles', 'msd_lookback')},
            'initial_sort': ('hybrid MSD (2 partition passes over 8-byte [key|index] elements, the second in one sweep by decoupled '
                             'look-back, + LDS local sort)' if sa_stats.get('msd_lookback') else
                             'hybrid MSD (2 partition passes over 8-byte [key|index] elements + LDS local sort)') if sa_stats['msd']
                            else 'LSD passes with shrinking keys',
            'kernel_ms': ({('msd_scatter2_kernel<true, 1024, true>' if prof.get('msd_lookback') else 'msd_scatter2_kernel<true, 1024>'): round(prof['msd_ms_g1'], 3),
                           ('msd_scatter_lb_kernel' if prof.get('msd_lookback') else 'msd_scatter2_kernel<false, 1024>'): round(prof['msd_ms_g2'], 3),
                           'msd_local_fast_kernel': round(prof['msd_ms_local'], 3)} if prof['msd'] else None),
            'roofline': roof,
            'build_roofline': build_roof,
            'cpu_baseline': cpu,
            'secondary': secondary,
            'adversarial': adversarial,
            'real_files': real_files,
        }
    else:
        out = None
    reader.close()
    del dT, dTs, dSA
    return (1 if verified is False else 0), out


# ----------------------------------------------------------------------------- configs[2] / [3] --

def run_corpus(args, D, steps=None, warmup=None):
    import torch
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    from pysubstringsearch_amd import Reader, _ffi
    from pysubstringsearch_amd import dist as pdist
    lib = _ffi.lib
    rank, world, dist = D.rank, D.world, D.dist
    n = 1 << args.logn
    dev = D.local_rank
    nq = args.queries or 100000
    chunks = args.chunks
    mine = [c for c in range(chunks) if pdist.chunk_owner(c, world) == rank]
    goldens = load_big_goldens()
    keep_host = world == 1 and not args.no_cpu_baseline      # the CPU baseline needs text + SA of every chunk

    h = ctypes.c_void_p()
    _ffi.check(lib.pss_reader_create(dev, ctypes.byref(h)))
    reader = Reader._from_handle(h)
    dSA = torch.empty(n, dtype=torch.int32, device='cuda')
    st = _ffi.SaStats()
    per_chunk = (nq // 2 + chunks - 1) // chunks
    sampled, texts, sas = {}, [], []
    build_s, verified, how = 0.0, True, ''
    for c in mine:
        host = np.empty(n, dtype=np.uint8)
        _ffi.check(lib.pss_gen_corpus(KINDS[args.corpus], host.ctypes.data, n, c))
        dT = torch.from_numpy(host).cuda()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _ffi.check(lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, dev, 0, ctypes.byref(st)))
        build_s += time.perf_counter() - t0
        ok, how = verify_sa(dSA, host, args.corpus, c, goldens, want_sha=False)
        verified = False if (ok is False or verified is False) else (None if (ok is None or verified is None) else True)
        _ffi.check(lib.pss_reader_add_chunk_device(h, dT.data_ptr(), dSA.data_ptr(), n))
        sampled[c] = sample_chunk_queries(host, c, per_chunk, args.qmin, args.qmax)
        if keep_host:
            texts.append(host)
            sas.append(dSA.cpu().numpy())
        del dT
    verified = D.all_true(verified)
    if world > 1:
        boxes = [None] * world
        dist.all_gather_object(boxes, sampled)
        sampled = {c: q for b in boxes for c, q in b.items()}
    queries = mixed_queries([q for c in range(chunks) for q in sampled[c]], nq, args.qmin, args.qmax)

    last = {}
    keep = []      # the result lists of the timed steps stay alive until the clock has stopped: tearing down
                   # 14.7 M str objects costs a third of building them and is not part of a search

    def step(packed=False):
        if world > 1:
            merged = pdist.gather_device(reader.search_batch_device(queries), dst=0)
            if merged is not None:
                blob, offsets, counts = merged
                last['entries'] = len(offsets) - 1
                if not packed:
                    keep.append(pdist.packed_to_list(blob, offsets, as_str=True))
        elif packed:
            pk = reader.search_batch_packed(queries)
            last['entries'] = len(pk.offsets) - 1
            last['bytes'] = int(pk.data.size)
        else:
            keep.append(reader.search_multiple_bytes_as_str(queries))
            last['entries'] = len(keep[-1])

    for _ in range(warmup):
        step()
    keep.clear()
    D.sync_all()
    t_begin = time.perf_counter()
    for _ in range(steps):
        step()
    D.sync_all()
    total_s = time.perf_counter() - t_begin
    keep.clear()
    stats = reader.last_stats()
    D.sync_all()
    t_p = time.perf_counter()
    for _ in range(steps):
        step(packed=True)
    D.sync_all()
    packed_s = time.perf_counter() - t_p
    per_rank_build_ms = [round(x * 1e3, 2) for x in D.gather_floats(build_s)]
    total_s, packed_s, build_s = D.max_over_ranks([total_s, packed_s, build_s])

    lat = None
    if world == 1:
        ts = []
        for q in queries[:1000]:
            t0 = time.perf_counter()
            reader.search_batch_raw([q])
            ts.append(time.perf_counter() - t0)
        ts.sort()
        lat = {'median': round(ts[len(ts) // 2] * 1e6, 1), 'p90': round(ts[int(len(ts) * 0.9)] * 1e6, 1),
               'queries_per_sec': round(len(ts) / sum(ts), 1)}
        # the same queries with the reader in low-latency mode (resident search kernel, include/pss.h): same results?
        want = [reader.search_batch_raw([q])[0] for q in queries[:200]]
        reader.set_low_latency(True)
        reader.search_batch_raw([queries[0]])
        ts = []
        for q in queries[:1000]:
            t0 = time.perf_counter()
            reader.search_batch_raw([q])
            ts.append(time.perf_counter() - t0)
        ts.sort()
        same = all(reader.search_batch_raw([q])[0] == w for q, w in zip(queries[:200], want))
        ll = reader.low_latency_stats()
        reader.set_low_latency(False)
        lat['low_latency_mode'] = {'median': round(ts[len(ts) // 2] * 1e6, 1), 'p90': round(ts[int(len(ts) * 0.9)] * 1e6, 1),
                                   'queries_per_sec': round(len(ts) / sum(ts), 1), 'same_results': bool(same),
                                   'kernels_started': ll['kernels_started'], 'queries_served': ll['queries_served']}
        if not same:
            verified = False
            how = 'low-latency mode returned different results'

    rc, out = 0, None
    if rank == 0:
        cpu = None
        if keep_host:
            cpu, cpu_counts = cpu_baseline_corpus(texts, sas, queries, args.cpu_sample_queries, not args.no_disk_baseline)
            got = reader.count_multiple_bytes(queries[:args.cpu_sample_queries])
            if not np.array_equal(np.asarray(got, dtype=np.uint64), cpu_counts):
                verified = False            # per-query entry counts differ from the oracle's
                how = 'per-query entry counts differ from the CPU oracle'
        hits_q = stats['hits'] / max(stats['queries'], 1)
        # A_query = chunks touched x 7.4 KB (58 dependent probes x 128 B) + 132 B per hit
        a_query = len(mine) * len(queries) * 7424 + stats['hits'] * 132
        ms_dev = stats['ms_device']
        # No fraction of the HBM peak is claimed for the search: SURVEY's 58-probe model is not what this implementation
        # moves (key samples + 64-ary steps), and its traffic has not been measured with PMC counters.
        # Traffic of the batch's kernels by PMC (profiles/pmc_search_corpus15.json, tests/tools/pmc_search.sh: separate
        # FETCH_SIZE / WRITE_SIZE runs of this workload at N = 1; raw request bytes -- the gather kernels read 64-byte sectors,
        # for which the x 2 of streaming reads does not apply) over the device time measured in THIS run.
        traffic = achieved = None
        pmcs = os.path.join(ROOT, 'profiles', 'pmc_search_corpus15.json')
        if os.path.exists(pmcs) and world == 1 and chunks == 15 and args.logn == 29 and len(queries) == 100000:
            try:
                traffic = json.loads(pathlib.Path(pmcs).read_text()).get('total_bytes_raw')
                achieved = round(traffic / ms_dev / 1e6, 1) if traffic and ms_dev else None
            except Exception:
                traffic = achieved = None
        roof = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': None if achieved is None else round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                'traffic_source': None if traffic is None else 'profiles/pmc_search_corpus15.json (committed PMC runs of this workload; '
                                  'raw FETCH_SIZE + WRITE_SIZE of the batch kernels, not measured in this run)',
                'kernel': 'search pipeline of one batch on rank 0 (interval search, entry recovery, emit)',
                'ms_device': round(ms_dev, 3), 'ms_interval': round(stats['ms_interval'], 3),
                'survey_model_bytes': int(a_query),
                'note': 'latency / transaction bound by construction (SURVEY 8(d)): graded on queries/s against the CPU path; '
                        'achieved = measured fabric bytes of the batch kernels / device time of the batch (the bytes of the SURVEY '
                        'model -- 58 dependent probes -- are not the bytes this search moves)'}
        qps = len(queries) * steps / total_s
        out = {
            'metric': METRIC, 'value': None if verified is False else round(qps, 1), 'unit': 'queries/s',
            'n_gpus': world, 'steps': steps, 'warmup': warmup,
            'ms_per_step': round(total_s / steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic',
            'verified': verified, 'verified_by': how + ('; per-query entry counts of a sample equal to the CPU oracle' if cpu else ''),
            'config': {
                'workload': f'configs[2]/[3]: {chunks} x {n >> 20} MiB `{args.corpus}` chunks ({chunks * n / 1e9:.2f} GB) resident, chunk c on '
                            f'rank c mod {world}; one batch of {len(queries)} queries of {args.qmin}..{args.qmax} bytes (50% sampled)',
                'corpus': args.corpus, 'chunks': chunks, 'chunk_bytes': n, 'queries': len(queries),
                'value_is': 'batched queries/s through the drop-in list API (Reader.search_multiple semantics: '
                            'list[str] on rank 0), including H2D, kernels, D2H / gather and Python list construction',
                'imbalance': f'{max(len([c for c in range(chunks) if c % world == r]) for r in range(world))} chunks on the '
                             f'fullest rank: best-case speed-up {chunks / max(len([c for c in range(chunks) if c % world == r]) for r in range(world)):.2f}x',
            },
            'entries_per_batch': last.get('entries'), 'hits_per_query': round(hits_q, 2),
            'packed_queries_per_sec': round(len(queries) * steps / packed_s, 1),
            'single_query_us': lat,
            'index_build_gbs': round(chunks * n / build_s / 1e9, 3),
            'per_rank_build_ms': per_rank_build_ms,
            'search_stats': stats,
            'roofline': roof, 'cpu_baseline': cpu,
        }
        if cpu and cpu.get('value'):
            # The CPU figures above stop at the packed result, like the packed API here.  The reference hands
            # Python a list[str] (pyo3, src/lib.rs:284-286): creating those str objects costs the same on
            # either side, so the list-level CPU figure adds the per-entry cost measured on the GPU path.
            list_s_per_entry = max(total_s - packed_s, 0.0) / steps / max(last.get('entries') or 1, 1)
            ns = min(len(queries), args.cpu_sample_queries)
            e_sample = cpu['entries_per_query'] * ns
            cpu['list_level_queries_per_sec'] = round(ns / (ns / cpu['value'] + e_sample * list_s_per_entry), 1)
            cpu['python_str_ns_per_entry'] = round(list_s_per_entry * 1e9, 1)
            out['gpu_over_cpu'] = {'packed_api_vs_cpu_packed': round(out['packed_queries_per_sec'] / cpu['value'], 2),
                                   'list_api_vs_cpu_list_level': round(qps / cpu['list_level_queries_per_sec'], 2),
                                   'list_api_vs_cpu_packed': round(qps / cpu['value'], 2)}
            if cpu.get('disk_queries_per_sec'):
                dl = ns / (ns / cpu['disk_queries_per_sec'] + e_sample * list_s_per_entry)
                out['gpu_over_cpu']['list_api_vs_reference_access_path'] = round(qps / dl, 2)
                out['gpu_over_cpu']['packed_api_vs_reference_access_path'] = round(
                    out['packed_queries_per_sec'] / cpu['disk_queries_per_sec'], 2)
        rc = 1 if verified is False else 0
    reader.close()
    return rc, out


def launch_ranks(n: int, printed=None) -> int:
    """`python3 bench.py --gpus N` without a launcher: this process never touches a GPU; it starts N fresh copies of
    itself, one per GPU, with the torch.distributed environment of a one-node job (rendezvous on 127.0.0.1), lets
    rank 0's stdout through (the one JSON line) and returns the worst exit code.  A rank that dies takes the others
    with it (they would wait in a collective forever)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))

    def forward():
        # rank 0's stdout: the JSON line goes to ours, anything else a library printed there (gloo's connection
        # banner ...) to stderr -- the driver reads ONE line from this process
        for line in procs[0].stdout:
            (sys.stdout if line.startswith('{') else sys.stderr).write(line)
            sys.stdout.flush()
            if printed is not None and line.startswith('{'):
                printed.append(1)

    import threading
    fw = threading.Thread(target=forward, daemon=True)
    fw.start()
    rc, live, killed = 0, set(range(n)), set()
    codes = {}
    while live:
        time.sleep(0.2)
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            codes[r] = code
            if code != 0:
                rc = rc or code
                for q in live:      # exactly the processes started above
                    if q not in killed:
                        procs[q].terminate()
                        killed.add(q)
    fw.join(timeout=10)
    # Rank 0 prints the line.  When it ended by itself, its code is the verdict on that line (a rank that failed in the
    # configs[2]/[3] leg leaves with 1 so that its peers are stopped at once; rank 0 then still prints the configs[1]
    # result it has and leaves with THAT result's code).
    if 0 not in killed and 0 in codes:
        return codes[0]
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', default='chunk', choices=['chunk', 'corpus15'])
    ap.add_argument('--corpus', default='lines', choices=sorted(KINDS))
    ap.add_argument('--logn', type=int, default=29, help='log2 of the chunk size (29 = the 512 MiB default chunk)')
    ap.add_argument('--queries', type=int, default=0, help='queries per batch (default 10 000 / 100 000 by config)')
    ap.add_argument('--qlen', type=int, default=8)
    ap.add_argument('--chunks', type=int, default=15, help='corpus15: chunks of the corpus')
    ap.add_argument('--qmin', type=int, default=4)
    ap.add_argument('--qmax', type=int, default=32)
    ap.add_argument('--cpu-sample-logn', type=int, default=26)
    ap.add_argument('--cpu-sample-queries', type=int, default=5000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-disk-baseline', action='store_true')
    ap.add_argument('--no-corpus15', action='store_true', help='chunk config: skip the configs[2]/[3] leg of the line')
    ap.add_argument('--corpus15-queries', type=int, default=100000)
    ap.add_argument('--no-real-files', action='store_true', help='chunk config: skip the build of real files found on the machine')
    ap.add_argument('--real-files-logn', type=int, default=28, help='... and how many bytes of them (2^this) make the chunk')
    ap.add_argument('--no-e2e', action='store_true', help='chunk config: skip the file-API leg (Writer -> .idx -> Reader)')
    ap.add_argument('--e2e-chunks', type=int, default=4)
    ap.add_argument('--inproc', action='store_true',
                    help='N GPUs inside ONE process (a host thread per device; no torch.distributed, no RCCL)')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and not args.inproc:
        printed = []
        rc = launch_ranks(args.gpus, printed)      # nothing has touched a GPU yet (torch is not even imported)
        if printed or os.environ.get('PSS_BENCH_NO_INPROC_FALLBACK'):
            sys.exit(rc)
        # The ranks route (one process per GPU, RCCL) printed no line: the same N GPUs inside this one process instead --
        # a second, independent route to the number (its line says `route: inproc`).
        sys.stderr.write('bench.py: the ranks route printed no line (exit code %d); falling back to --inproc\n' % rc)
        args.inproc = True
    if args.inproc and 'WORLD_SIZE' not in os.environ:
        rc, out = run_inproc(args)
        print(json.dumps(out), flush=True)
        sys.exit(rc)
    leg = {'corpus15': False, 'bail': None}
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        try:                                            # (a leftover of an earlier job on this port)
            os.remove(os.path.join('/tmp', 'pss_bench_abort_%s' % os.environ.get('MASTER_PORT', '0')))
        except OSError:
            pass
        # SIGTERM is what torchrun and launch_ranks send to every rank when one dies.  Blocking it (pthread_sigmask) is not
        # enough: a runtime thread that unblocks it takes the signal with the default action and the process is gone
        # before rank 0 has printed the line it already has (seen on the GPU box).  A handler changes the disposition for
        # the whole process, whichever thread the signal lands on, and CPython's C-level handler writes the signal number
        # to the wake-up descriptor at once -- even while the main thread sits inside a collective and runs no Python
        # code.  The watcher thread blocks on the other end of that pipe.  Set up HERE, before torch is imported.
        import signal
        import threading
        term_r, term_w = os.pipe()
        os.set_blocking(term_w, False)
        signal.signal(signal.SIGTERM, lambda *_: None)
        signal.set_wakeup_fd(term_w, warn_on_full_buffer=False)

        def watch_term():
            while True:
                try:
                    if not os.read(term_r, 1):
                        return
                except OSError:
                    return
                # A peer can fail in the corpus15 leg while this rank is still on its way there (assembling its configs[1]
                # result): give the main thread a moment to reach the leg, where the line it has can be printed.  A rank
                # that is stuck in a collective of configs[1] itself never gets there and leaves empty-handed.
                if leg.get('fallback'):
                    continue                # rank 0 is producing the line through the in-process route: the line comes first
                for _ in range(300):
                    if leg.get('fallback'):
                        break
                    if leg['corpus15'] and leg['bail'] is not None:
                        leg['bail']('terminated by the launcher during the corpus15 leg (another rank failed)')
                    if leg.get('past'):
                        break
                    time.sleep(0.1)
                if leg.get('fallback'):
                    continue
                os._exit(143)
        threading.Thread(target=watch_term, daemon=True).start()
    # The ranks route has never met real peers (the builder's boxes have one GPU).  If its configs[1] leg does not finish --
    # RCCL does not come up, a collective hangs, a rank dies with an exception -- the N-GPU number is still wanted: rank 0
    # then runs the SAME workload through the in-process route (a fresh process: this one may be stuck inside a
    # collective) over all the devices and prints that line (`route: inproc`, with the reason); the other ranks just leave.
    ranks_abort = os.path.join('/tmp', 'pss_bench_ranks_abort_%s' % os.environ.get('MASTER_PORT', '0'))
    chunk_done = None

    def ranks_fallback(reason):
        rank = int(os.environ.get('RANK', '0'))
        try:
            with open(ranks_abort, 'a') as f:
                f.write(f'rank {rank}: {reason}\n')
        except OSError:
            pass
        if rank != 0:
            os._exit(0)                 # (0: a launcher that sees a failed rank tears rank 0 down with it)
        leg['fallback'] = True          # (SIGTERM is ignored from here on: the line comes first)
        time.sleep(3.0)                 # the others leave their GPUs
        import subprocess
        env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR',
                                                                 'MASTER_PORT', 'GROUP_RANK', 'ROLE_RANK', 'TORCHELASTIC_RUN_ID')}
        cmd = [sys.executable, os.path.abspath(__file__), '--gpus', str(args.gpus), '--inproc', '--steps', str(args.steps),
               '--warmup', str(args.warmup), '--logn', str(args.logn), '--corpus', args.corpus, '--qlen', str(args.qlen)]
        if args.queries:
            cmd += ['--queries', str(args.queries)]
        code, line = 1, None
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
            code = r.returncode
            line = next((ln for ln in r.stdout.splitlines() if ln.startswith('{')), None)
        except Exception as e:      # noqa: BLE001
            reason += f'; the in-process route failed too: {type(e).__name__}: {e}'
        if line:
            try:
                d = json.loads(line)
                d['fallback_reason'] = f'the ranks route (one process per GPU, {os.environ.get("PSS_BENCH_BACKEND", "nccl")}) did not finish: ' + reason[:400]
                line = json.dumps(d)
            except Exception:       # noqa: BLE001
                pass
            print(line, flush=True)
        else:
            print(json.dumps({'metric': METRIC, 'value': None, 'unit': 'GB/s', 'n_gpus': args.gpus, 'error': reason[:600]}), flush=True)
        os._exit(code)

    if int(os.environ.get('WORLD_SIZE', '1')) > 1 and args.config == 'chunk' and not os.environ.get('PSS_BENCH_NO_INPROC_FALLBACK'):
        try:
            os.remove(ranks_abort)
        except OSError:
            pass
        chunk_done = threading.Event()
        limit = float(os.environ.get('PSS_BENCH_RANKS_TIMEOUT', '600'))

        def watch_ranks():
            t0 = time.time()
            while not chunk_done.is_set():
                time.sleep(0.5)
                if os.path.exists(ranks_abort) and not chunk_done.is_set():
                    ranks_fallback('another rank gave up')
                if time.time() - t0 > limit and not chunk_done.is_set():
                    ranks_fallback(f'configs[1] not finished after {limit:.0f} s')
        threading.Thread(target=watch_ranks, daemon=True).start()
    try:
        D = Dist(args)
        if args.config == 'chunk':
            if os.environ.get('PSS_BENCH_HANG_RANK') == os.environ.get('RANK', '0'):      # test hook: this rank never gets there
                time.sleep(3600)
            rc, out = run_chunk(args, D)
    except Exception as e:      # noqa: BLE001
        if chunk_done is None:
            raise
        import traceback
        traceback.print_exc()
        ranks_fallback(f'{type(e).__name__}: {e}'[:300])
    if chunk_done is not None:
        chunk_done.set()
    if args.config == 'chunk':
        if D.world == 1 and not args.no_e2e and out is not None:
            import torch
            torch.cuda.empty_cache()
            try:
                e2e = run_e2e(args)
            except Exception as e:      # noqa: BLE001   (a full temp directory must not cost the line its headline)
                e2e = {'error': f'{type(e).__name__}: {e}'[:300]}
            out['e2e'] = e2e
            out['summary'].update({'e2e_writer_text_gbs': e2e.get('writer_text_gbs'), 'e2e_writer_idx_gbs': e2e.get('writer_idx_gbs'),
                                   'e2e_striped_writer_text_gbs': (e2e.get('striped_format_2') or {}).get('writer_text_gbs'),
                                   'e2e_striped_reader_open_idx_gbs': (e2e.get('striped_format_2') or {}).get('reader_open_idx_gbs'),
                                   'e2e_reader_open_idx_gbs': e2e.get('reader_open_idx_gbs'), 'e2e_verified': e2e.get('verified')})
            if e2e.get('verified') is False:
                out['value'] = None
                rc = rc or 1
        if not args.no_corpus15:
            # BASELINE configs[2] / [3] in the same line: the 7.5 GB corpus (chunk c on rank c mod N), one batch of 100 000
            # queries of 4..32 bytes, at most 3 timed steps; the CPU path (one thread per chunk; RAM and the reference's
            # lseek + read probes) beside it at N = 1
            import torch
            torch.cuda.empty_cache()
            q0, args.queries = args.queries, args.corpus15_queries
            # With more than one rank this leg gathers results over RCCL: a rank that fails or hangs in it must not cost the
            # line its headline.  An exception here, a SIGTERM from the launcher (torchrun and launch_ranks send one to
            # every rank when one dies) or no end within PSS_BENCH_CORPUS15_TIMEOUT seconds (900), and rank 0 prints the
            # configs[1] result it already has, with the reason in place of the corpus15 object.  The watching is done by a
            # thread: the main thread may be stuck inside a collective, where no Python signal handler would ever run.
            flag = os.path.join('/tmp', 'pss_bench_abort_%s' % os.environ.get('MASTER_PORT', '0'))

            import threading
            bail_once = threading.Lock()

            def bail(reason, tell=True):
                if not bail_once.acquire(blocking=False):     # (the watcher and the main thread at once: one line)
                    time.sleep(3600)
                if tell and D.world > 1:                      # the other ranks of this node: stop waiting for me
                    try:
                        with open(flag, 'w') as f:
                            f.write(f'rank {D.rank}: {reason}')
                    except OSError:
                        pass
                if out is not None:
                    out['corpus15'] = {'error': reason}
                    print(json.dumps(out), flush=True)
                # rank 0 leaves with the code of the line it printed; every other rank with 1: a launcher sees a failed
                # rank and stops the others at once instead of leaving that to the flag file alone
                os._exit(rc if D.rank == 0 else (rc or 1))
            done = None
            if D.world > 1:
                done = threading.Event()
                leg['bail'] = bail
                leg['corpus15'] = True
                limit = time.time() + float(os.environ.get('PSS_BENCH_CORPUS15_TIMEOUT', '900'))

                def watch():
                    while not done.is_set():
                        time.sleep(0.25)
                        if os.path.exists(flag):
                            try:
                                why = pathlib.Path(flag).read_text()[:300]
                            except OSError:
                                why = 'another rank gave up'
                            bail(why, tell=False)
                        if time.time() > limit:
                            bail('the corpus15 leg did not finish in time')
                threading.Thread(target=watch, daemon=True).start()
            try:
                if os.environ.get('PSS_BENCH_FAIL_CORPUS15') == str(D.rank):      # test hook: this rank fails in the leg
                    raise RuntimeError('PSS_BENCH_FAIL_CORPUS15')
                rc2, out2 = run_corpus(args, D, steps=min(args.steps, 3), warmup=min(args.warmup, 1))
            except Exception as e:      # noqa: BLE001
                if D.world == 1:
                    raise
                bail(f'{type(e).__name__}: {e}'[:300])
            if done is not None:
                done.set()
                leg['corpus15'] = False
                leg['past'] = True
            args.queries = q0
            rc = rc or rc2
            if out is not None and out2 is not None:
                out['corpus15'] = {k: out2[k] for k in out2 if k not in ('metric', 'higher_is_better', 'vs_baseline', 'data')}
                if out2['value'] is None:
                    out['value'] = None
                c_cpu = out2.get('cpu_baseline') or {}
                lat = out2.get('single_query_us') or {}
                out['summary'].update({
                    'corpus15_list_api_queries_per_sec': out2.get('value'),
                    'corpus15_packed_api_queries_per_sec': out2.get('packed_queries_per_sec'),
                    'corpus15_cpu_ram_queries_per_sec': c_cpu.get('value'), 'corpus15_cpu_threads': c_cpu.get('cores'),
                    'corpus15_cpu_disk_queries_per_sec': c_cpu.get('disk_queries_per_sec'),
                    'corpus15_single_query_us': lat.get('median'),
                    'corpus15_single_query_us_low_latency': (lat.get('low_latency_mode') or {}).get('median'),
                    'corpus15_verified': out2.get('verified'),
                })
                # configs[2] / [3] on NATURAL-TEXT-LIKE chunks ("run on lines and words"): a leg of its own
                # (python bench.py --config corpus15 --corpus words --qmin 16: 15 words builds, 259 M result entries per batch,
                # minutes with its CPU baselines) -- quoted here from the committed run, not measured in this one
                try:
                    wl = json.loads(pathlib.Path(os.path.join(ROOT, 'profiles', 'r06_bench_corpus15_words.json')).read_text().strip().splitlines()[-1])
                    wc = wl.get('cpu_baseline') or {}
                    out['summary'].update({
                        'corpus15_words_packed_api_queries_per_sec': wl.get('packed_queries_per_sec'),
                        'corpus15_words_list_api_queries_per_sec': wl.get('value'),
                        'corpus15_words_entries_per_batch': wl.get('entries_per_batch'),
                        'corpus15_words_cpu_ram_queries_per_sec': wc.get('value'),
                        'corpus15_words_cpu_disk_queries_per_sec': wc.get('disk_queries_per_sec'),
                        'corpus15_words_verified': wl.get('verified'),
                        'corpus15_words_is': 'profiles/r06_bench_corpus15_words.json (committed run of --config corpus15 --corpus words '
                                             '--qmin 16 on one MI355X; NOT measured in this run)',
                    })
                except Exception:
                    pass
    else:
        rc, out = run_corpus(args, D)
    if out is not None:
        print(json.dumps(out), flush=True)
    D.finish()
    sys.exit(rc)


if __name__ == '__main__':
    main()
