"""Multi-GPU sharding of the search path: one process per GPU
(``torch.distributed``; backend "nccl" is RCCL on ROCm, "gloo" on CPU tests).

Chunks are independent (reference src/lib.rs:105-124 builds and src/lib.rs:207
searches each chunk alone), so chunk c lives on rank ``c % world_size`` and no
collective sits on the data path.  The only exchange is the gather of result
strings to one rank (the reference concatenates per-chunk results under a
mutex, src/lib.rs:280): per-query counts, entry lengths and entry bytes are
gathered and re-interleaved query-major.  Dedupe stays local to the owning
rank (per (query, chunk), src/lib.rs:262,274) -- entries live in exactly one
chunk, so a cross-rank dedupe would wrongly drop identical entries.
"""
import typing

import numpy as np


def chunk_owner(chunk_index: int, world_size: int) -> int:
    return chunk_index % world_size


def pack_entries(entries: typing.Sequence[bytes]) -> typing.Tuple[np.ndarray, np.ndarray]:
    lens = np.fromiter((len(e) for e in entries), dtype=np.int64, count=len(entries))
    blob = np.frombuffer(b''.join(entries), dtype=np.uint8).copy()   # writable: torch.from_numpy needs it
    return blob, lens


def merge_packed(per_rank: typing.Sequence[typing.Tuple[np.ndarray, np.ndarray, np.ndarray]]
                 ) -> typing.Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """per_rank[r] = (blob uint8, lens int64[E_r], counts int64[nq]) with rank r's entries
    query-major.  Returns the merged result packed the same way -- (blob, offsets int64[E + 1],
    summed per-query counts) -- query-major, inside a query rank-major, inside a rank in its
    local order.  Vectorised: no Python object per entry."""
    nq = len(per_rank[0][2])
    qid, rk, src, length = [], [], [], []
    base = 0
    for r, (blob, lens, counts) in enumerate(per_rank):
        lens = np.asarray(lens, dtype=np.int64)
        counts = np.asarray(counts, dtype=np.int64)
        assert len(counts) == nq and int(counts.sum()) == len(lens)
        qid.append(np.repeat(np.arange(nq, dtype=np.int64), counts))
        rk.append(np.full(len(lens), r, dtype=np.int64))
        off = np.zeros(len(lens) + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        src.append(off[:-1] + base)          # position in the concatenation of all blobs
        length.append(lens)
        base += len(blob)
    qid, rk = np.concatenate(qid), np.concatenate(rk)
    src, length = np.concatenate(src), np.concatenate(length)
    order = np.lexsort((rk, qid))            # stable: keeps each rank's local order inside (query, rank)
    src, length = src[order], length[order]
    offsets = np.zeros(len(length) + 1, dtype=np.int64)
    np.cumsum(length, out=offsets[1:])
    total_bytes = int(offsets[-1])
    big = np.concatenate([np.asarray(p[0], dtype=np.uint8) for p in per_rank]) if base else np.zeros(0, np.uint8)
    if total_bytes:
        # byte i of the output comes from big[src[e] + (i - offsets[e])] for the entry e that holds it
        idx = np.repeat(src - offsets[:-1], length) + np.arange(total_bytes, dtype=np.int64)
        out = big[idx]
    else:
        out = np.zeros(0, dtype=np.uint8)
    total = np.sum([np.asarray(p[2], dtype=np.int64) for p in per_rank], axis=0).astype(np.int64)
    return out, offsets, total


def merge_query_major(per_rank: typing.Sequence[typing.Tuple[np.ndarray, np.ndarray, np.ndarray]]
                      ) -> typing.Tuple[typing.List[bytes], np.ndarray]:
    """``merge_packed`` with the entries as a list of bytes."""
    blob, offsets, total = merge_packed(per_rank)
    data = blob.tobytes()
    o = offsets.tolist()
    return [data[o[i]:o[i + 1]] for i in range(len(o) - 1)], total


def gather_packed(blob, lens, counts, group=None, dst: int = 0, packed: bool = False):
    """Collective on packed local results: ``blob`` = this rank's entries back to
    back (uint8), ``lens`` their lengths, ``counts`` the per-query entry counts,
    all query-major.  Rank ``dst`` gets (all entries query-major, total counts)
    -- with ``packed=True`` (blob, offsets, total counts), no Python object per entry --,
    the others None.  On the nccl (= RCCL) backend the payload travels as one device
    tensor per rank; sizes are exchanged first so that it can be padded to a common shape."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.int64)
    cnt = np.ascontiguousarray(counts, dtype=np.int64)
    nq = len(cnt)
    # two collectives: the sizes, then ONE payload per rank (counts | lengths | bytes) padded to the largest
    sizes = torch.tensor([len(lens), len(blob)], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    all_sizes = torch.stack(all_sizes).cpu().numpy()
    width = int((nq * 8 + all_sizes[:, 0] * 8 + all_sizes[:, 1]).max())
    payload = np.zeros(max(width, 1), dtype=np.uint8)
    mine = np.concatenate([cnt.view(np.uint8), lens.view(np.uint8), blob])
    payload[:len(mine)] = mine
    t = torch.from_numpy(payload).to(dev)
    # all_gather (not gather): supported by every backend/version; the payload is
    # small (result strings), so the extra copies to non-destination ranks are noise
    g = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(g, t, group=group)
    if rank != dst:
        return None
    per_rank = []
    for r in range(world):
        e, b = int(all_sizes[r, 0]), int(all_sizes[r, 1])
        raw = g[r].cpu().numpy()
        o1, o2 = nq * 8, nq * 8 + e * 8
        per_rank.append((raw[o2:o2 + b], raw[o1:o2].view(np.int64), raw[:o1].view(np.int64)))
    return merge_packed(per_rank) if packed else merge_query_major(per_rank)


def gather_results(entries: typing.Sequence[bytes], counts: typing.Sequence[int], group=None, dst: int = 0):
    """``gather_packed`` for a local result given as a list of entries."""
    blob, lens = pack_entries(entries)
    return gather_packed(blob, lens, counts, group, dst)


class ShardedReader:
    """Reader over the chunks owned by this rank; ``search_multiple`` is a
    collective returning the full result on rank ``dst`` (None elsewhere)."""

    def __init__(self, index_file_path: str, group=None, device: typing.Optional[int] = None, reader=None):
        import torch.distributed as dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        if reader is None:
            from . import Reader
            reader = Reader(index_file_path, device=device, shard=(self.rank, self.world))
        self.local = reader

    def search_multiple_bytes(self, patterns: typing.Sequence[bytes], dst: int = 0):
        local = self.local
        if hasattr(local, 'search_batch_packed'):
            # no per-entry Python objects on the ranks that only contribute
            pk = local.search_batch_packed(list(patterns))
            return gather_packed(pk.data, np.diff(pk.offsets.astype(np.int64)), pk.counts, self.group, dst)
        entries, counts = local.search_batch_raw(list(patterns))
        return gather_results(entries, counts, self.group, dst)

    def search_multiple(self, substrings: typing.List[str], dst: int = 0):
        got = self.search_multiple_bytes([s.encode('utf-8') for s in substrings], dst)
        return None if got is None else [e.decode('utf-8') for e in got[0]]

    def search(self, substring: str, dst: int = 0):
        return self.search_multiple([substring], dst)
