"""Multi-GPU sharding of the search path: one process per GPU
(``torch.distributed``; backend "nccl" is RCCL on ROCm, "gloo" on CPU tests).

Chunks are independent (reference src/lib.rs:105-124 builds and src/lib.rs:207
searches each chunk alone), so chunk c lives on rank ``c % world_size`` and no
collective sits on the data path.  The only exchange is the gather of result
strings to one rank (the reference concatenates per-chunk results under a
mutex, src/lib.rs:280-284):

* every rank answers the whole batch for its chunks and LEAVES the packed result
  in HBM (``Reader.search_batch_device``);
* one tiny all_gather exchanges the sizes (entries, bytes per rank);
* the contributing ranks ``isend`` their three device buffers (per-query counts, entry
  starts, entry bytes) to the collecting rank, which ``irecv``s them into device buffers of
  exactly that size -- one grouped RCCL send / recv batch, nothing padded, nothing sent to
  ranks that do not need it, no host round trip on the contributing ranks;
* the collecting rank brings the buffers down through pinned memory and merges them
  query-major (rank-major inside a query) with one ``memcpy`` per (query, rank) segment
  (``pss_merge_packed`` in libpss).

Dedupe stays local to the owning rank (per (query, chunk), src/lib.rs:262,274)
-- entries live in exactly one chunk, so a cross-rank dedupe would wrongly drop
identical entries.
"""
import ctypes
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
    local order.  The work is one C loop over (query, rank) segments (pss_merge_packed): no
    per-entry Python object and no per-byte index."""
    starts = []
    for blob, lens, counts in per_rank:
        lens = np.asarray(lens, dtype=np.int64)
        off = np.zeros(len(lens) + 1, dtype=np.uint64)
        np.cumsum(lens, out=off[1:])
        starts.append(off[:-1])
    return merge_packed_starts([(np.asarray(b, dtype=np.uint8), s, np.asarray(c)) for (b, _, c), s in zip(per_rank, starts)])


def merge_packed_starts(per_rank: typing.Sequence[typing.Tuple[np.ndarray, np.ndarray, np.ndarray]]
                        ) -> typing.Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """The same merge for results given as (blob uint8, entry starts [E_r], counts [nq]) -- the
    layout of ``pss_device_result`` -- with 8-byte integers of either signedness."""
    from . import _ffi
    world = len(per_rank)
    nq = len(per_rank[0][2])
    blobs = [np.ascontiguousarray(p[0], dtype=np.uint8) for p in per_rank]
    starts = [np.ascontiguousarray(p[1]).view(np.uint64) if np.asarray(p[1]).dtype.itemsize == 8
              else np.ascontiguousarray(p[1], dtype=np.uint64) for p in per_rank]
    counts = [np.ascontiguousarray(p[2]).view(np.uint64) if np.asarray(p[2]).dtype.itemsize == 8
              else np.ascontiguousarray(p[2], dtype=np.uint64) for p in per_rank]
    for c, s in zip(counts, starts):
        assert len(c) == nq and int(c.sum()) == len(s)
    n_ent = np.array([len(s) for s in starts], dtype=np.uint64)
    n_byt = np.array([len(b) for b in blobs], dtype=np.uint64)
    E, B = int(n_ent.sum()), int(n_byt.sum())
    out_counts = np.zeros(max(nq, 1), dtype=np.uint64)
    out_offsets = np.zeros(E + 1, dtype=np.uint64)
    out_bytes = np.empty(max(B, 1), dtype=np.uint8)
    arr = ctypes.c_void_p * world
    _ffi.check(_ffi.lib.pss_merge_packed(
        world, nq, arr(*[c.ctypes.data for c in counts]), arr(*[s.ctypes.data for s in starts]),
        arr(*[b.ctypes.data for b in blobs]), n_ent.ctypes.data, n_byt.ctypes.data,
        out_counts.ctypes.data, out_offsets.ctypes.data, out_bytes.ctypes.data))
    return out_bytes[:B], out_offsets.view(np.int64), out_counts[:nq].view(np.int64)


def packed_to_list(blob: np.ndarray, offsets: np.ndarray, as_str: bool = False) -> list:
    """Entries of a packed result as a Python list (one C loop when the glue module is built)."""
    n = len(offsets) - 1
    if n <= 0:
        return []
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets).view(np.uint64)
    try:
        from . import _pssglue
        return _pssglue.entries_to_list(blob.ctypes.data if blob.size else 0, offsets.ctypes.data, n, as_str)
    except ImportError:   # pragma: no cover
        data = blob.tobytes()
        o = offsets.tolist()
        out = [data[o[i]:o[i + 1]] for i in range(n)]
        return [e.decode('utf-8') for e in out] if as_str else out


def merge_query_major(per_rank: typing.Sequence[typing.Tuple[np.ndarray, np.ndarray, np.ndarray]]
                      ) -> typing.Tuple[typing.List[bytes], np.ndarray]:
    """``merge_packed`` with the entries as a list of bytes."""
    blob, offsets, total = merge_packed(per_rank)
    return packed_to_list(blob, offsets), total


_pinned_cache: dict = {}


def _pinned(torch, nbytes: int, key: str):
    """Grow-only pinned host staging (pinning a gigabyte costs more than copying it)."""
    t = _pinned_cache.get(key)
    if t is None or t.numel() < nbytes:
        try:
            t = torch.empty(max(nbytes + nbytes // 8, 1 << 20), dtype=torch.uint8, pin_memory=True)
        except RuntimeError:      # no accelerator in this process (gloo tests on CPU)
            t = torch.empty(max(nbytes, 1), dtype=torch.uint8)
        _pinned_cache[key] = t
    return t[:nbytes]


def gather_device(result, group=None, dst: int = 0):
    """Collective on a ``DeviceResult`` (``Reader.search_batch_device``: data uint8, entry starts
    int64, per-query counts int64, all in HBM).  Rank ``dst`` gets the merged packed result
    (blob uint8, offsets int64[E + 1], counts int64[nq]) as numpy arrays, the others None.

    nccl (= RCCL) backend: the three buffers of every contributing rank travel device to device
    in one grouped send / recv batch straight into buffers of their exact size on ``dst``; no
    host copy on the contributing ranks, nothing padded, nothing broadcast.  gloo (CPU tests, or
    the 1-GPU test hook): the same exchange on host tensors."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    on_device = dist.get_backend(group) == 'nccl'
    data, starts, counts = result.data, result.starts, result.counts
    if not on_device:
        data, starts, counts = data.cpu(), starts.cpu(), counts.cpu()
    # (Reader.search_batch_device hands out tensors it owns -- copies of the engine's workspace slots -- so the
    # collective can use them as they are)
    dev = data.device
    nq = counts.numel()
    sizes = torch.tensor([starts.numel(), data.numel()], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    all_sizes = torch.stack(all_sizes).cpu().tolist()

    def peer(r):     # global rank of group rank r
        return dist.get_global_rank(group, r) if group is not None else r

    ops, recv = [], {}
    if rank == dst:
        for r in range(world):
            if r == dst:
                continue
            e, b = all_sizes[r]
            bufs = (torch.empty(b, dtype=torch.uint8, device=dev), torch.empty(e, dtype=torch.int64, device=dev),
                    torch.empty(nq, dtype=torch.int64, device=dev))
            recv[r] = bufs
            ops += [dist.P2POp(dist.irecv, t, peer(r), group) for t in (bufs[2], bufs[1], bufs[0]) if t.numel()]
    else:
        ops += [dist.P2POp(dist.isend, t, peer(dst), group) for t in (counts, starts, data) if t.numel()]
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if rank != dst:
        return None
    recv[dst] = (data, starts, counts)
    if on_device and world > 1:
        return merge_on_device([recv[r] for r in range(world)], nq)
    per_rank = []
    for r in range(world):
        d, s, c = recv[r]
        if d.is_cuda:      # down through pinned staging: one DMA per buffer at link speed
            hd = _pinned(torch, d.numel(), f'd{r}')
            hs = _pinned(torch, s.numel() * 8, f's{r}').view(torch.int64)
            hc = _pinned(torch, c.numel() * 8, f'c{r}').view(torch.int64)
            hd.copy_(d, non_blocking=True)
            hs.copy_(s, non_blocking=True)
            hc.copy_(c, non_blocking=True)
            per_rank.append((hd, hs, hc))
        else:
            per_rank.append((d, s, c))
    if any(t[0].is_pinned() for t in per_rank if t[0].numel()):
        torch.cuda.synchronize()
    return merge_packed_starts([(d.numpy(), s.numpy(), c.numpy()) for d, s, c in per_rank])


_merge_flip = 0


def merge_on_device(per_rank, nq: int):
    """``per_rank`` = [(data uint8, starts int64, counts int64)] device tensors of the same nq queries, all on one GPU
    (the collecting rank's own result and the ones RCCL delivered): merged there by the engine
    (``pss_merge_packed_device``: query-major, rank-major inside a query), then ONE download through pinned memory
    instead of one per rank.  Returns (blob uint8, offsets int64[E + 1], counts int64[nq]) as numpy arrays -- views of
    pinned staging buffers that alternate between two sets: a result stays valid until the call after the next one."""
    import ctypes
    import torch
    from . import _ffi
    world = len(per_rank)
    dev = per_rank[0][0].device
    E = sum(int(s.numel()) for _, s, _ in per_rank)
    B = sum(int(d.numel()) for d, _, _ in per_rank)
    out_counts = torch.empty(max(nq, 1), dtype=torch.int64, device=dev)
    out_offsets = torch.empty(E + 1, dtype=torch.int64, device=dev)
    out_bytes = torch.empty(max(B, 1), dtype=torch.uint8, device=dev)
    vp = ctypes.c_void_p * world
    u64 = ctypes.c_uint64 * world
    torch.cuda.synchronize(dev)          # the engine works on its own stream
    _ffi.check(_ffi.lib.pss_merge_packed_device(
        dev.index or 0, world, nq,
        vp(*[c.data_ptr() if c.numel() else None for _, _, c in per_rank]),
        vp(*[s.data_ptr() if s.numel() else None for _, s, _ in per_rank]),
        vp(*[d.data_ptr() if d.numel() else None for d, _, _ in per_rank]),
        u64(*[int(s.numel()) for _, s, _ in per_rank]), u64(*[int(d.numel()) for d, _, _ in per_rank]),
        out_counts.data_ptr(), out_offsets.data_ptr(), out_bytes.data_ptr()))
    global _merge_flip
    _merge_flip ^= 1
    hd = _pinned(torch, B, f'md{_merge_flip}')
    ho = _pinned(torch, (E + 1) * 8, f'mo{_merge_flip}').view(torch.int64)
    hc = _pinned(torch, nq * 8, f'mc{_merge_flip}').view(torch.int64)
    hd.copy_(out_bytes[:B], non_blocking=True)
    ho.copy_(out_offsets, non_blocking=True)
    hc.copy_(out_counts[:nq], non_blocking=True)
    torch.cuda.synchronize(dev)
    return hd.numpy(), ho.numpy(), hc.numpy()


def gather_packed(blob, lens, counts, group=None, dst: int = 0, packed: bool = False):
    """``gather_device`` for a local result that is already on the host: ``blob`` = this rank's
    entries back to back (uint8), ``lens`` their lengths, ``counts`` the per-query entry counts,
    all query-major.  Rank ``dst`` gets (all entries query-major, total counts) -- with
    ``packed=True`` (blob, offsets, total counts) --, the others None."""
    import torch
    import torch.distributed as dist
    from . import DeviceResult
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')
    lens = np.ascontiguousarray(lens, dtype=np.int64)
    st = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=st[1:])
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    res = DeviceResult(torch.from_numpy(blob.copy() if not blob.flags.writeable else blob).to(dev),
                       torch.from_numpy(st[:-1].copy()).to(dev),
                       torch.from_numpy(np.ascontiguousarray(counts).astype(np.int64)).to(dev), int(blob.size))
    merged = gather_device(res, group, dst)
    if merged is None:
        return None
    if packed:
        return merged
    return packed_to_list(merged[0], merged[1]), merged[2]


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

    def search_multiple_packed(self, patterns: typing.Sequence[bytes], dst: int = 0):
        """(blob, offsets, counts) on ``dst``, None elsewhere."""
        local = self.local
        if hasattr(local, 'search_batch_device'):
            return gather_device(local.search_batch_device(list(patterns)), self.group, dst)
        if hasattr(local, 'search_batch_packed'):
            pk = local.search_batch_packed(list(patterns))
            return gather_packed(pk.data, np.diff(pk.offsets.astype(np.int64)), pk.counts, self.group, dst, packed=True)
        entries, counts = local.search_batch_raw(list(patterns))
        blob, lens = pack_entries(entries)
        return gather_packed(blob, lens, counts, self.group, dst, packed=True)

    def search_multiple_bytes(self, patterns: typing.Sequence[bytes], dst: int = 0):
        got = self.search_multiple_packed(patterns, dst)
        return None if got is None else (packed_to_list(got[0], got[1]), got[2])

    def search_multiple(self, substrings: typing.List[str], dst: int = 0):
        got = self.search_multiple_packed([s.encode('utf-8') for s in substrings], dst)
        return None if got is None else packed_to_list(got[0], got[1], as_str=True)

    def search(self, substring: str, dst: int = 0):
        return self.search_multiple([substring], dst)


class EngineComm:
    """The engine's own RCCL communicator (include/pss.h, pss_comm_*): built from a 128-byte id that rank 0 makes and
    ``share`` hands to every rank -- ``share(id_bytes_or_None) -> id_bytes`` is any broadcast the caller has (a
    ``torch.distributed`` broadcast over gloo, MPI, a file).  With it the gather of per-rank results happens INSIDE the C
    ABI (``pss_gather_packed_rccl``: grouped ncclSend / ncclRecv on the device results, device-side merge), no torch
    tensor in the path: what a C or Rust host of the library calls."""

    def __init__(self, rank: int, world: int, device: int, share):
        import ctypes
        from . import _ffi
        buf = (ctypes.c_uint8 * 128)()
        if rank == 0:
            _ffi.check(_ffi.lib.pss_comm_unique_id(buf))
        raw = share(bytes(buf) if rank == 0 else None)
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(raw)
        self._h = ctypes.c_void_p()
        self.rank, self.world, self.device = rank, world, device
        _ffi.check(_ffi.lib.pss_comm_init(buf, world, rank, device, ctypes.byref(self._h)))

    def gather(self, reader, patterns: typing.Sequence[bytes], dst: int = 0):
        """Collective: every rank searches its own chunks (``reader``: a single-device Reader) and rank ``dst`` gets
        (blob uint8, offsets int64[E + 1], counts int64[nq]) as numpy arrays, the others None."""
        import ctypes
        from . import _ffi
        nq = len(patterns)
        blob = b''.join(patterns)
        offs = np.zeros(nq + 1, dtype=np.uint64)
        if nq:
            np.cumsum(np.fromiter(map(len, patterns), dtype=np.uint64, count=nq), out=offs[1:])
        dr = _ffi.DeviceResult()
        _ffi.check(_ffi.lib.pss_reader_search_batch_device(reader._handle(), blob, offs.ctypes.data, nq, ctypes.byref(dr)))
        res = ctypes.c_void_p()
        _ffi.check(_ffi.lib.pss_gather_packed_rccl(self._h, ctypes.byref(dr), dst, ctypes.byref(res)))
        if self.rank != dst:
            return None
        try:
            n = _ffi.lib.pss_result_num_entries(res)
            counts = np.ctypeslib.as_array(_ffi.lib.pss_result_query_counts(res), shape=(max(nq, 1),))[:nq].astype(np.int64)
            offsets = np.ctypeslib.as_array(_ffi.lib.pss_result_offsets(res), shape=(n + 1,)).astype(np.int64)
            nbytes = int(offsets[n])
            data = (np.ctypeslib.as_array(_ffi.lib.pss_result_bytes(res), shape=(max(nbytes, 1),))[:nbytes].copy()
                    if nbytes else np.zeros(0, dtype=np.uint8))
            return data, offsets, counts
        finally:
            _ffi.lib.pss_result_free(res)

    def close(self) -> None:
        from . import _ffi
        if self._h:
            h, self._h = self._h, None
            _ffi.lib.pss_comm_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001
            pass
