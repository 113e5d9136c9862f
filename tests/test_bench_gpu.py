"""bench.py contract on a GPU box: the JSON line of a small single-GPU run, and the N > 1 code
path (sharding by rank, result gather, max-over-ranks timing) with two processes.  A 1-GPU box
cannot host two RCCL ranks, so the two-process run uses bench.py's test hook
PSS_BENCH_BACKEND=gloo: both ranks share GPU 0 and torch.distributed runs over gloo."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

REQUIRED = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_single_gpu_contract():
    r = subprocess.run([sys.executable, 'bench.py', '--steps', '2', '--warmup', '1', '--logn', '22', '--queries', '500',
                        '--cpu-sample-logn', '20', '--chunks', '5', '--corpus15-queries', '3000', '--cpu-sample-queries', '300',
                        '--real-files-logn', '24'],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for k in REQUIRED:
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['scaling'] == 'weak'
    assert d['unit'] == 'GB/s' and d['value'] > 0 and d['higher_is_better'] is True and d['vs_baseline'] is None
    assert 'workload' in d['config'] and 'model' not in d['config']
    roof = d['roofline']
    assert roof['bound'] == 'hbm' and roof['unit'] == 'GB/s' and roof['peak'] == 8000.0
    assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-3
    cpu = d['cpu_baseline']
    assert cpu['cores'] == 1 and cpu['kind'] in ('reference', 'port') and cpu['value'] > 0 and cpu['sample']
    assert d['entries_per_batch'] == d['search_stats']['entries']
    assert d['verified'] is True and 'libsais' in d['verified_by']      # small n: libsais run on the spot
    # the query leg of the line is checked too: a sample of the batch against the oracle, per query
    vs = d['verified_search']
    assert vs['ok'] is True and vs['queries'] >= 100 and vs['entries'] > 0
    assert roof['traffic'] is None or roof['traffic_source']
    # BASELINE configs[2] / [3] ride in the same line: queries/s on the multi-chunk corpus next to the CPU path
    c = d['corpus15']
    assert c['unit'] == 'queries/s' and c['value'] > 0 and c['scaling'] == 'strong' and c['verified'] is True
    assert c['config']['chunks'] == 5 and c['packed_queries_per_sec'] > 0 and c['single_query_us']['median'] > 0
    assert c['cpu_baseline']['value'] > 0 and c['cpu_baseline']['disk_queries_per_sec'] > 0
    assert c['roofline']['frac'] is None and len(c['per_rank_build_ms']) == 1
    # round 5: real files, cold and warm, checked against libsais in the run; the striped container beside the reference's
    rf = d['real_files']
    assert rf and 'error' not in rf and rf['verified'] is True and rf['build_ms'] > 0 and rf['build_ms_cold'] > 0 and 'note' not in rf
    assert d['e2e']['verified'] is True and d['e2e']['striped_format_2']['verified'] is True
    assert any(a['corpus'] == 'source' and a['verified'] for a in d['adversarial'])
    ll = c['single_query_us']['low_latency_mode']                     # the resident search kernel: same results, no launch per query
    assert ll['same_results'] is True and ll['median'] > 0 and ll['queries_served'] >= 1000 and ll['kernels_started'] >= 1


def _check_two_ranks(d):
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['value'] > 0
    assert d['cpu_baseline'] is None and d['secondary'] is None     # rank 0, N = 1 only
    # rank 1 holds a different chunk (seed + 1): the sampled half of the queries comes from rank
    # 0's text, so the gathered batch has at least those hits
    assert d['entries_per_batch'] >= 250
    c = d['corpus15']
    assert c['n_gpus'] == 2 and c['value'] > 0 and c['verified'] is True and len(c['per_rank_build_ms']) == 2
    assert '3 chunks on the fullest rank' in c['config']['imbalance']


def test_bench_two_ranks_self_launch():
    """`python bench.py --gpus 2` with no launcher in the command (the shape of the driver's command): bench.py starts
    the two ranks itself, rank 0's JSON line is the only line on stdout."""
    env = dict(os.environ, PSS_BENCH_BACKEND='gloo')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--logn', '22', '--queries', '500',
           '--chunks', '5', '--corpus15-queries', '3000']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len([l for l in r.stdout.splitlines() if l.strip()]) == 1, r.stdout
    _check_two_ranks(_last_json(r.stdout))


def test_bench_two_ranks_keep_the_headline_when_the_corpus_leg_fails():
    """A rank that fails inside the configs[2]/[3] leg (test hook) must not cost the line its configs[1] result: rank 0
    still prints one line, with the reason in place of the corpus15 object, and the run ends -- by the other rank's
    exception or, if that one hangs in a collective, by the timeout."""
    env = dict(os.environ, PSS_BENCH_BACKEND='gloo', PSS_BENCH_FAIL_CORPUS15='1', PSS_BENCH_CORPUS15_TIMEOUT='60')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--logn', '22', '--queries', '500',
           '--chunks', '5', '--corpus15-queries', '3000']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    d = _last_json(r.stdout)
    assert d['n_gpus'] == 2 and d['value'] > 0 and 'error' in d['corpus15']


def test_bench_two_ranks_under_torchrun():
    env = dict(os.environ, PSS_BENCH_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--logn', '22',
           '--queries', '500', '--chunks', '5', '--corpus15-queries', '3000']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    _check_two_ranks(_last_json(r.stdout))


def _gpus() -> int:
    import torch
    return torch.cuda.device_count()       # (does not initialise the GPUs)


def test_bench_inproc_route():
    """`--inproc`: N GPUs inside one process, a host thread per device, no torch.distributed (on the one-GPU test box the
    two "devices" take turns on GPU 0).  The line carries the weak-scaling build figure, the per-device times and the
    check of the multi-device Writer / Reader handles against the one-device ones."""
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'PSS_BENCH_BACKEND'):
        env.pop(k, None)
    cmd = [sys.executable, 'bench.py', '--gpus', '2', '--inproc', '--steps', '2', '--warmup', '1', '--logn', '22', '--queries', '500']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d['route'] == 'inproc' and d['n_gpus'] == 2 and d['value'] > 0 and d['verified'] is True
    assert len(d['per_device']) == 2 and all(x['verified'] is True for x in d['per_device'])
    h = d['multi_device_handles']
    assert h['writer_same_bytes_as_one_device'] is True and h['reader_same_results_as_one_device'] is True


def test_bench_falls_back_to_inproc_when_the_ranks_route_prints_nothing():
    """On a one-GPU box `python bench.py --gpus 2` over RCCL cannot start its second rank: no line from the ranks route,
    so the same command answers through the in-process route instead of leaving the driver without a number."""
    if _gpus() >= 2:
        pytest.skip('the ranks route works here')
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'PSS_BENCH_BACKEND'):
        env.pop(k, None)
    cmd = [sys.executable, 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--logn', '22', '--queries', '500',
           '--no-corpus15']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    d = _last_json(r.stdout)
    assert d['route'] == 'inproc' and d['n_gpus'] == 2 and d['value'] > 0, r.stderr[-2000:]


def test_bench_ranks_route_that_hangs_falls_back_on_rank_0():
    """The ranks route (one process per GPU) has never met real peers: when its configs[1] leg does not finish (test hook:
    rank 1 never arrives, rank 0 waits in the first collective), rank 0 runs the same workload through the in-process
    route in a fresh process and prints THAT line, with the reason; the other ranks leave quietly."""
    env = dict(os.environ, PSS_BENCH_BACKEND='gloo', PSS_BENCH_HANG_RANK='1', PSS_BENCH_RANKS_TIMEOUT='25')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--logn', '22', '--queries', '500',
           '--no-corpus15']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    d = _last_json(r.stdout)
    assert d['route'] == 'inproc' and d['n_gpus'] == 2 and d['value'] > 0 and d['verified'] is True, r.stderr[-2000:]
    assert 'did not finish' in d['fallback_reason']
