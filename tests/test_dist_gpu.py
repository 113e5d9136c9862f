"""The real multi-GPU paths: two processes, two GPUs, RCCL.  Skipped where fewer than two GPUs are visible (the one-GPU
test boxes); everything they would exercise on one GPU is covered with virtual devices / gloo elsewhere
(test_multi_device_reader_in_one_process, test_device_merge_equals_host_merge, test_sharded_gather_gloo,
test_bench_two_ranks_self_launch)."""
import pathlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import pysubstringsearch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _gpus() -> int:
    import torch
    return torch.cuda.device_count()       # (does not initialise the GPUs)


@pytest.mark.skipif(_gpus() < 2, reason='needs two GPUs')
def test_rccl_gather_and_multi_device_handles(tmp_path, oracle):
    from tests.util import gen_corpus
    text = gen_corpus(0, 1 << 19).tobytes()
    (tmp_path / 'c.idx.txt').write_bytes(text)
    idx = str(tmp_path / 'c.idx')
    w = pysubstringsearch.Writer(idx, 1 << 16)
    w.add_entries_from_file_lines(idx + '.txt')
    w.close()
    rng = np.random.default_rng(4)
    qs = [b'', b'e', b'zzzzzz', b'th']
    while len(qs) < 3000:
        s = int(rng.integers(0, len(text) - 20))
        cand = text[s:s + int(rng.integers(1, 14))]
        qs.append(cand)
    pathlib.Path(idx + '.queries.json').write_text(json.dumps([q.decode('latin-1') for q in qs]))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out = str(tmp_path / 'r0.json')
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', '_dist_gpu_worker.py'), str(r), '2', str(port), idx, out],
                              env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    got = json.loads(pathlib.Path(out).read_text())
    o = oracle.OracleReader(idx)
    for name, batch in (('one', qs[:1]), ('few', qs[:40]), ('all', qs)):
        oe, oc = o.search_multiple_bytes(batch)
        assert got[name]['counts'] == oc.tolist(), name
        pos = 0
        for c in oc.tolist():
            assert sorted(got[name]['entries'][pos:pos + c]) == sorted(e.decode('latin-1') for e in oe[pos:pos + c])
            pos += c
        assert got['cabi_' + name]['counts'] == oc.tolist(), name                     # the gather inside the C ABI
        assert sorted(got['cabi_' + name]['entries']) == sorted(e.decode('latin-1') for e in oe)
    assert got['multi_writer_identical'] is True
    # and the single-process reader over the two real devices
    with pysubstringsearch.Reader(idx, devices=[0, 1]) as r:
        ents, counts = r.search_batch_raw(qs)
        oe, oc = o.search_multiple_bytes(qs)
        assert counts == oc.tolist() and sorted(ents) == sorted(oe)
    # the unchanged drop-in call: every visible device without being told (src/lib.rs:205-207)
    for var in ('PSS_DEVICES', 'PSS_DEVICE', 'LOCAL_RANK'):
        os.environ.pop(var, None)
    with pysubstringsearch.Reader(idx) as r:
        assert r.devices == list(range(_gpus())) and len(r.devices) >= 2
        ents, counts = r.search_batch_raw(qs)
        assert counts == oc.tolist() and sorted(ents) == sorted(oe)
