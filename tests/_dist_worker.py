"""world_size-2 gloo worker for tests/test_host.py::test_sharded_gather_gloo."""
import pathlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch.distributed as dist  # noqa: E402

from pysubstringsearch_amd import dist as pdist  # noqa: E402


class StubReader:
    """Stands in for the per-rank device Reader: canned local results."""

    def __init__(self, table):
        self.table = table

    def search_batch_raw(self, patterns):
        entries, counts = [], []
        for p in patterns:
            got = self.table.get(p.decode(), [])
            entries.extend(e.encode() for e in got)
            counts.append(len(got))
        return entries, counts


class PackedStubReader(StubReader):
    """Same canned results through the packed interface (what the device Reader offers)."""

    def search_batch_packed(self, patterns):
        import numpy as np
        from pysubstringsearch_amd import PackedResult
        entries, counts = self.search_batch_raw(patterns)
        offs = np.zeros(len(entries) + 1, dtype=np.uint64)
        np.cumsum([len(e) for e in entries], out=offs[1:])
        return PackedResult(np.frombuffer(b''.join(entries), dtype=np.uint8), offs, np.asarray(counts, dtype=np.uint64))


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    tables = [
        {'ten': ['ten', 'tenten'], 'x': [], 'e': ['one', 'three']},
        {'ten': ['ten'], 'x': ['x'], 'e': []},
    ]
    r = pdist.ShardedReader('unused', reader=StubReader(tables[rank]))
    got = r.search_multiple(['ten', 'zzz', 'x', 'e', 'ten'])
    raw = r.search_multiple_bytes([b'e', b'ten'])
    rp = pdist.ShardedReader('unused', reader=PackedStubReader(tables[rank]))
    packed = rp.search_multiple(['ten', 'zzz', 'x', 'e', 'ten'])
    assert packed == got
    empty = rp.search_multiple_bytes([b'zzz'])
    assert (empty is None) if rank else (empty[0] == [] and empty[1].tolist() == [0])
    if rank == 0:
        pathlib.Path(out).write_text(json.dumps({'got': got, 'counts': raw[1].tolist(), 'raw': [e.decode() for e in raw[0]]}))
    else:
        assert got is None and raw is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
