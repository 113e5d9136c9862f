"""world_size-2 RCCL worker for tests/test_dist_gpu.py: one process per GPU, each a Reader over its shard of the index,
the packed results gathered device to device and merged on the collecting rank's GPU."""
import pathlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from pysubstringsearch_amd import Writer, dist as pdist  # noqa: E402


def main():
    rank, world, port, idx, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    torch.cuda.set_device(rank)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world,
                            device_id=torch.device('cuda', rank))
    r = pdist.ShardedReader(idx, device=rank)
    queries = json.loads(pathlib.Path(idx + '.queries.json').read_text())
    qs = [q.encode('latin-1') for q in queries]
    res = {}
    for name, batch in (('one', qs[:1]), ('few', qs[:40]), ('all', qs)):
        got = r.search_multiple_bytes(batch)
        if rank == 0:
            res[name] = {'counts': [int(c) for c in got[1]], 'entries': [e.decode('latin-1') for e in got[0]]}
        else:
            assert got is None
    # the same gather INSIDE the C ABI (pss_gather_packed_rccl: the engine's own communicator, no torch tensor in the path);
    # the 128-byte id travels over the process group that is already up
    def share(raw):
        box = [raw]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    comm = pdist.EngineComm(rank, world, rank, share)
    for name, batch in (('one', qs[:1]), ('few', qs[:40]), ('all', qs)):
        got = comm.gather(r.local, batch, dst=0)
        if rank == 0:
            ents = pdist.packed_to_list(got[0], got[1])
            res['cabi_' + name] = {'counts': [int(c) for c in got[2]], 'entries': [e.decode('latin-1') for e in ents]}
        else:
            assert got is None
    comm.close()
    # the multi-device Writer with two distinct ordinals: same bytes as the single-device file
    if rank == 0:
        w = Writer(idx + '.multi', 1 << 16, devices=[0, 1])
        w.add_entries_from_file_lines(idx + '.txt')
        w.close()
        res['multi_writer_identical'] = pathlib.Path(idx + '.multi').read_bytes() == pathlib.Path(idx).read_bytes()
        pathlib.Path(out).write_text(json.dumps(res))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
