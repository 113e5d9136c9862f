"""pysubstringsearch_amd -- MI355X-native drop-in for ``pysubstringsearch``.

Same classes, method names, argument names and error behaviour as the
reference Python layer (pysubstringsearch/__init__.py:6-73 over
src/lib.rs:42-288); the work is done by hand-written HIP kernels for gfx950
behind the C ABI of ``libpss.so`` (include/pss.h):

* ``Writer`` builds each chunk's 32-bit suffix array on the GPU and writes the
  reference's ``.idx`` chunk records byte for byte.
* ``Reader`` keeps text and suffix arrays resident in HBM;
  ``search_multiple`` is ONE batched device call (one wavefront per query)
  instead of a Python loop.

Extras that do not change the reference signatures: ``device=`` keyword,
``close()`` / context-manager support, ``Reader(..., shard=(i, n))``.
"""
import ctypes
import os
import typing

from . import _ffi
from ._ffi import lib as _lib

try:   # C loop for result lists (csrc/pyglue.c); plain Python slicing if it was not built
    from . import _pssglue
except ImportError:   # pragma: no cover
    _pssglue = None

__all__ = ['Writer', 'Reader', 'PackedResult', 'DeviceResult', 'device_count', 'default_devices', 'release_workspace', 'workspace_bytes']


def device_count() -> int:
    return _lib.pss_device_count()


def release_workspace() -> None:
    """Free the engine's grow-only HBM workspace (suffix-array build buffers, search
    scratch) on every device; resident Reader chunks stay."""
    _ffi.check(_lib.pss_release_workspace())


def workspace_bytes(device: int = 0) -> int:
    """HBM the engine's grow-only workspaces of ``device`` hold right now (include/pss.h, pss_workspace_bytes): what a
    chunk in flight costs next to a resident Reader; ``release_workspace()`` gives it back."""
    return int(_lib.pss_workspace_bytes(int(device)))


def default_devices() -> typing.List[int]:
    """The devices ``Reader(path)`` / ``Writer(path)`` use when neither ``device`` nor ``devices`` is given
    (include/pss.h, pss_default_devices): ``PSS_DEVICES=all|0,1,...``; else the one a launcher pinned this process to
    (``PSS_DEVICE`` / ``LOCAL_RANK`` / ``SLURM_LOCALID`` / ``OMPI_COMM_WORLD_LOCAL_RANK``); else every visible device -- the reference's ``search`` uses every core of the
    machine without being asked (src/lib.rs:205-207)."""
    arr = (ctypes.c_int32 * 64)()
    k = _lib.pss_default_devices(arr, 64)
    if k < 1:
        raise ValueError(_ffi.last_error() or 'PSS_DEVICES does not name devices')
    return [int(arr[i]) for i in range(k)]


def _default_device() -> int:
    return default_devices()[0]


def _utf8(value, name: str) -> bytes:
    # pyo3 `&str` extraction: only str is accepted (bytes -> TypeError)
    if not isinstance(value, str):
        raise TypeError(f"argument '{name}': '{type(value).__name__}' object cannot be converted to 'PyString'")
    return value.encode('utf-8')


def _path(value, name: str) -> bytes:
    if not isinstance(value, str):
        raise TypeError(f"argument '{name}': '{type(value).__name__}' object cannot be converted to 'PyString'")
    return os.fsencode(value)


class Writer:
    """Reference: pysubstringsearch/__init__.py:6-41, src/lib.rs:42-144."""

    def __init__(
        self,
        index_file_path: str,
        max_chunk_len: typing.Optional[int] = None,
        *,
        device: typing.Optional[int] = None,
        devices: typing.Optional[typing.Sequence[int]] = None,
        format_version: int = 1,
        striped: bool = False,
    ) -> None:
        """``format_version=2`` (extension, opt-in) writes the container with 64-bit lengths: chunks of
        up to 2^31 - 1 bytes instead of the reference format's < 1 GiB.  Reader opens either.
        ``striped=True`` (with ``format_version=2``) keeps the suffix arrays out of the index file, in eight files
        ``<path>.sa0 .. .sa7`` written and read by a thread each (include/pss.h, PSS_FORMAT_STRIPED): one file in the
        page cache takes 11 - 14 GB/s on the test box however many threads write it, a file per writer four times that.
        ``devices=[0, 1, ...]`` (extension) builds chunk k of the file on ``devices[k % len(devices)]``,
        several chunks at once; the records are still written in chunk order, so the file is
        byte-identical to the single-device one."""
        if format_version not in (1, 2):
            raise ValueError('format_version must be 1 (the reference container) or 2')
        if striped and format_version != 2:
            raise ValueError('striped=True needs format_version=2 (the reference container has no place for the flag)')
        if max_chunk_len is not None:
            if not isinstance(max_chunk_len, int) or isinstance(max_chunk_len, bool):
                raise TypeError("argument 'max_chunk_len': must be an int or None")
            if max_chunk_len < 0:
                raise OverflowError("can't convert negative int to unsigned")   # Option<usize>
        self._h = ctypes.c_void_p()
        path = _path(index_file_path, 'index_file_path')
        if devices is not None:
            if device is not None:
                raise ValueError('pass either device or devices')
            devs = [int(d) for d in devices]
            if not devs:
                raise ValueError('devices must not be empty')
        else:
            devs = default_devices() if device is None else [device]      # no argument: every device the process may use
        self.devices = list(devs)
        arr = (ctypes.c_int32 * len(devs))(*devs)
        rc = _lib.pss_writer_open_multi(
            path, -1 if max_chunk_len is None else max_chunk_len, arr, len(devs), format_version | (0x100 if striped else 0),
            ctypes.byref(self._h))
        _ffi.check(rc, index_file_path)

    @property
    def writer(self) -> 'Writer':
        """The reference wrapper exposes `.writer` (the native object, __init__.py:12).  A property, not an attribute
        holding `self`: that would make every handle a reference cycle, and `del w` would flush the last chunk only
        when the cyclic collector gets round to it -- the reference's Drop flushes at once (src/lib.rs:138-144)."""
        return self

    def _handle(self):
        if not self._h:
            raise ValueError('I/O operation on closed Writer')
        return self._h

    @property
    def io_stats(self) -> dict:
        """Extension (diagnostics): records written through a shared mapping / pwritten, file-ingest bytes read straight
        into the chunk / through a block buffer (include/pss.h, pss_writer_io_stats)."""
        st = _ffi.WriterIo()
        _ffi.check(_lib.pss_writer_io_stats(self._handle(), ctypes.byref(st)))
        return {k: int(getattr(st, k)) for k, _ in st._fields_}

    def add_entries_from_file_lines(self, input_file_path: str) -> None:
        _ffi.check(_lib.pss_writer_add_file_lines(self._handle(), _path(input_file_path, 'input_file_path')),
                   input_file_path)

    def add_entry(self, text: str) -> None:
        b = _utf8(text, 'text')
        _ffi.check(_lib.pss_writer_add_entry(self._handle(), b, len(b)))

    def dump_data(self) -> None:
        _ffi.check(_lib.pss_writer_dump(self._handle()))

    def finalize(self) -> None:
        _ffi.check(_lib.pss_writer_finalize(self._handle()))

    def close(self) -> None:
        """Finalize and release the file (the reference does this on drop, src/lib.rs:138-144)."""
        if self._h:
            h, self._h = self._h, ctypes.c_void_p()
            _ffi.check(_lib.pss_writer_close(h))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _ResultOwner:
    """Keeps a pss_result alive for the numpy views handed out by search_batch_packed."""

    def __init__(self, handle) -> None:
        self._h = handle

    def view(self, ptr, count: int, dtype):
        import numpy as np
        if not count or not ptr:
            return np.zeros(0, dtype=dtype)
        nbytes = int(count) * np.dtype(dtype).itemsize
        buf = (ctypes.c_uint8 * nbytes).from_address(ctypes.addressof(ptr.contents))
        buf._owner = self               # numpy keeps `buf` as the array's base, `buf` keeps the result
        a = np.frombuffer(buf, dtype=dtype, count=int(count))
        a.flags.writeable = False
        return a

    def __del__(self):
        h, self._h = self._h, None
        if h:
            _lib.pss_result_free(h)


class PackedResult(typing.NamedTuple):
    data: typing.Any      # numpy uint8: all entries back to back
    offsets: typing.Any   # numpy uint64 [num_entries + 1]
    counts: typing.Any    # numpy uint64 [num_queries]


class _DevicePtr:
    """A raw HBM range as an object torch.as_tensor understands (CUDA array interface)."""

    def __init__(self, ptr: int, nbytes: int) -> None:
        self.__cuda_array_interface__ = {'shape': (nbytes,), 'typestr': '|u1', 'data': (ptr, False), 'version': 2}


class DeviceResult(typing.NamedTuple):
    """Packed result of one batch left in HBM (torch tensors owned by the caller)."""
    data: typing.Any      # torch uint8 [num_bytes]
    starts: typing.Any    # torch int64 [num_entries]: start of every entry in data
    counts: typing.Any    # torch int64 [num_queries]
    num_bytes: int


class Reader:
    """Reference: pysubstringsearch/__init__.py:44-73, src/lib.rs:146-288."""

    def __init__(
        self,
        index_file_path: str,
        *,
        device: typing.Optional[int] = None,
        devices: typing.Optional[typing.Sequence[int]] = None,
        shard: typing.Tuple[int, int] = (0, 1),
        order: typing.Optional[str] = None,
    ) -> None:
        """``order='sa'`` (extension): the entries a chunk contributes to a query come in the reference's order --
        suffix-array order of their first hit, src/lib.rs:262-276 -- so that ``search(s)`` of a one-chunk index equals the
        reference's list element by element (default ``'text'``: the same multiset, see ``set_result_order``).
        ``devices=[0, 1, ...]`` (extension) makes chunk c of the file resident on ``devices[c % len(devices)]`` and
        answers every search on all of them at once, inside this process -- no launcher, no ``torch.distributed``: one
        host thread per device, results merged on the host (the reference fans a search over its chunks with rayon,
        src/lib.rs:207).  ``shard=(i, n)`` is the one-process-per-GPU form of the same split (``dist.ShardedReader``)."""
        self._h = ctypes.c_void_p()
        path = _path(index_file_path, 'index_file_path')
        if devices is not None:
            if device is not None or tuple(shard) != (0, 1):
                raise ValueError('pass devices, or device / shard')
            devs = [int(d) for d in devices]
            if not devs:
                raise ValueError('devices must not be empty')
        elif device is None and tuple(shard) == (0, 1):
            devs = default_devices()        # no argument: every device the process may use (one under a launcher)
        else:
            devs = [_default_device() if device is None else int(device)]
        self.devices = list(devs)
        if len(devs) > 1:
            arr = (ctypes.c_int32 * len(devs))(*devs)
            rc = _lib.pss_reader_open_multi(path, arr, len(devs), ctypes.byref(self._h))
        else:
            rc = _lib.pss_reader_open(path, devs[0], shard[0], shard[1], ctypes.byref(self._h))
        _ffi.check(rc, index_file_path)
        if order is not None:
            self.set_result_order(order)

    @property
    def reader(self) -> 'Reader':
        """The reference wrapper exposes `.reader` (__init__.py:49); a property for the reason given at Writer.writer
        (a dropped Reader gives its HBM back at once)."""
        return self

    @classmethod
    def _from_handle(cls, handle) -> 'Reader':
        r = cls.__new__(cls)
        r._h = handle
        r.devices = []
        return r

    def _handle(self):
        if not self._h:
            raise ValueError('I/O operation on closed Reader')
        return self._h

    @property
    def num_chunks(self) -> int:
        return _lib.pss_reader_num_chunks(self._handle())

    @property
    def chunks_per_device(self) -> typing.List[int]:
        """Extension: how many chunks every part of the reader holds -- chunk c of the file lives on ``devices[c % G]``
        (SURVEY 8(e)), so 15 chunks over 8 devices read ``[2, 2, 2, 2, 2, 2, 2, 1]``; one number for a single-device reader."""
        g = int(_lib.pss_reader_part_chunks(self._handle(), None, 0))
        buf = (ctypes.c_uint64 * max(g, 1))()
        _lib.pss_reader_part_chunks(self._handle(), buf, g)
        return [int(buf[i]) for i in range(g)]

    @property
    def residency(self) -> dict:
        """Extension: where the resident index lives -- bytes in HBM, bytes of suffix arrays kept in
        pinned host memory (chunks beyond the HBM budget) and how many chunks that concerns."""
        hbm, host, nhost = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        _ffi.check(_lib.pss_reader_residency(self._handle(), ctypes.byref(hbm), ctypes.byref(host), ctypes.byref(nhost)))
        return {'hbm_bytes': hbm.value, 'host_bytes': host.value, 'host_chunks': nhost.value}

    @property
    def chunk_tiers(self) -> typing.List[str]:
        """Extension: where the suffix array of every chunk lives right now -- 'hbm' or 'host' (pinned memory)."""
        n = self.num_chunks
        buf = (ctypes.c_uint8 * max(n, 1))()
        _ffi.check(_lib.pss_reader_chunk_tiers(self._handle(), buf, n, None))
        return ['host' if buf[i] else 'hbm' for i in range(n)]

    @property
    def residency_moves(self) -> int:
        """Extension: exchanges / promotions the residency manager has made on its own (pss_reader_set_auto_residency)."""
        moves = ctypes.c_uint64()
        _ffi.check(_lib.pss_reader_chunk_tiers(self._handle(), None, 0, ctypes.byref(moves)))
        return moves.value

    def set_auto_residency(self, on: bool = True) -> None:
        """Extension: between batches the hottest suffix array of the host tier changes places with
        the coldest one in HBM (include/pss.h).  On by default; ``evict`` / ``promote`` override by hand."""
        _ffi.check(_lib.pss_reader_set_auto_residency(self._handle(), 1 if on else 0))

    def evict(self, chunk: int) -> None:
        """Extension: move the suffix array of resident chunk ``chunk`` out of HBM into pinned host
        memory (searches keep working, the kernels read it over PCIe); ``promote`` brings it back."""
        _ffi.check(_lib.pss_reader_evict_chunk(self._handle(), int(chunk)))

    def promote(self, chunk: int) -> None:
        _ffi.check(_lib.pss_reader_promote_chunk(self._handle(), int(chunk)))

    def set_result_order(self, order: str) -> None:
        """Extension: ``'sa'`` -- the reference's order inside a chunk (suffix-array order of every entry's first hit,
        src/lib.rs:262-276); ``'text'`` (default) -- suffix-array order of every entry's leftmost match.  The multiset is
        the same; they differ when an entry holds the pattern more than once.  ``PSS_RESULT_ORDER=sa`` sets the default."""
        if order not in ('sa', 'text'):
            raise ValueError("order must be 'sa' or 'text'")
        _ffi.check(_lib.pss_reader_set_result_order(self._handle(), 1 if order == 'sa' else 0))

    @property
    def result_order(self) -> str:
        return 'sa' if _lib.pss_reader_result_order(self._handle()) == 1 else 'text'

    def set_low_latency(self, on: bool = True) -> None:
        """Extension: single queries (``search``) through a resident search kernel that waits for them in a pinned
        mailbox -- no kernel launch, no stream synchronisation per query; same results.  The kernel leaves by itself
        1 ms after the last query (include/pss.h, pss_reader_set_low_latency).  Off by default."""
        _ffi.check(_lib.pss_reader_set_low_latency(self._handle(), 1 if on else 0))

    def low_latency_stats(self) -> dict:
        launches, served = ctypes.c_uint64(), ctypes.c_uint64()
        _ffi.check(_lib.pss_reader_low_latency_stats(self._handle(), ctypes.byref(launches), ctypes.byref(served)))
        return {'kernels_started': launches.value, 'queries_served': served.value}

    def _search_batch(self, patterns: typing.Sequence[bytes], as_str: bool):
        nq = len(patterns)
        if _pssglue is not None:
            blob, offs = _pssglue.pack_queries(patterns)      # one C loop: blob + u64 offsets
        else:
            blob = b''.join(patterns)
            arr = (ctypes.c_uint64 * (nq + 1))()
            pos = 0
            for i, p in enumerate(patterns):
                arr[i] = pos
                pos += len(p)
            arr[nq] = pos
            offs = bytes(arr)
        res = ctypes.c_void_p()
        rc = _lib.pss_reader_search_batch(self._handle(), blob, offs, nq, ctypes.byref(res))
        _ffi.check(rc)
        try:
            n = _lib.pss_result_num_entries(res)
            if not nq:
                counts = []
            elif _pssglue is not None:
                counts = _pssglue.u64_list(ctypes.cast(_lib.pss_result_query_counts(res), ctypes.c_void_p).value, nq)
            else:
                counts = list(_lib.pss_result_query_counts(res)[:nq])
            entries = []
            if n:
                off = _lib.pss_result_offsets(res)
                base = _lib.pss_result_bytes(res)
                if _pssglue is not None:
                    entries = _pssglue.entries_to_list(ctypes.cast(base, ctypes.c_void_p).value,
                                                       ctypes.cast(off, ctypes.c_void_p).value, n, as_str)
                else:
                    # not ctypes.string_at: its size argument is a C int (results beyond 2 GiB)
                    addr = ctypes.cast(base, ctypes.c_void_p).value
                    data = bytes((ctypes.c_char * off[n]).from_address(addr)) if off[n] else b''
                    o = off[:n + 1]
                    entries = [data[o[i]:o[i + 1]] for i in range(n)]
                    if as_str:
                        entries = [e.decode('utf-8') for e in entries]
            return entries, counts
        finally:
            _lib.pss_result_free(res)

    def count_multiple(self, substrings: typing.List[str]) -> typing.List[int]:
        """Extension (not in the reference API): ``len(search(s))`` for every s, in one batched
        device call that materialises no entry -- only the counters come back."""
        import numpy as np
        return self.count_multiple_bytes([_utf8(s, 'substring') for s in substrings])

    def count(self, substring: str) -> int:
        """Extension: ``len(search(substring))`` without building the entries."""
        return self.count_multiple([substring])[0]

    def search_batch_packed(self, patterns: typing.Sequence[bytes]) -> 'PackedResult':
        """One batched device call, zero per-entry Python objects: read-only numpy
        views of the packed result, which lives as long as they do (entry i of the
        batch = data[offsets[i]:offsets[i+1]], entries are query-major, counts[q] of
        them belong to query q).  For
        hit-heavy batches the Python list of ``search_multiple`` costs more than
        the search itself (~50 ns per entry); this is the bulk alternative."""
        import numpy as np
        nq = len(patterns)
        blob = b''.join(patterns)
        offs = np.zeros(nq + 1, dtype=np.uint64)
        if nq:
            np.cumsum(np.fromiter(map(len, patterns), dtype=np.uint64, count=nq), out=offs[1:])
        res = ctypes.c_void_p()
        _ffi.check(_lib.pss_reader_search_batch(self._handle(), blob, offs.ctypes.data, nq, ctypes.byref(res)))
        owner = _ResultOwner(res)      # the arrays below are views of the C result; it lives as long as they do
        n = _lib.pss_result_num_entries(res)
        counts = owner.view(_lib.pss_result_query_counts(res), nq, np.uint64)
        offsets = owner.view(_lib.pss_result_offsets(res), n + 1, np.uint64)
        data = owner.view(_lib.pss_result_bytes(res), int(offsets[n]), np.uint8)
        return PackedResult(data, offsets, counts)

    def search_batch_device(self, patterns: typing.Sequence[bytes]) -> 'DeviceResult':
        """One batched device call whose packed result STAYS in HBM (torch tensors, no copy): what the
        multi-GPU gather sends over RCCL (``dist.gather_device``).  Needs torch."""
        import numpy as np
        import torch
        nq = len(patterns)
        blob = b''.join(patterns)
        offs = np.zeros(nq + 1, dtype=np.uint64)
        if nq:
            np.cumsum(np.fromiter(map(len, patterns), dtype=np.uint64, count=nq), out=offs[1:])
        dr = _ffi.DeviceResult()
        _ffi.check(_lib.pss_reader_search_batch_device(self._handle(), blob, offs.ctypes.data, nq, ctypes.byref(dr)))
        dev = torch.device('cuda', dr.device)

        def wrap(ptr, nbytes, dtype):
            if not nbytes or not ptr:
                return torch.empty(0, dtype=dtype, device=dev)
            return torch.as_tensor(_DevicePtr(ptr, nbytes), device=dev).view(dtype)

        # The three ranges are slots of the device's shared workspace: another handle or thread on the device may
        # reuse them as soon as the call has returned.  Own them before anyone else gets in (device-to-device, a few
        # microseconds for the sizes a batch produces).
        out = DeviceResult(wrap(dr.d_bytes, dr.num_bytes, torch.uint8).clone(), wrap(dr.d_offsets, dr.num_entries * 8, torch.int64).clone(),
                           wrap(dr.d_counts, nq * 8, torch.int64).clone(), int(dr.num_bytes))
        torch.cuda.current_stream(dev).synchronize()
        return out

    def search_multiple_bytes_as_str(self, patterns: typing.Sequence[bytes]) -> typing.List[str]:
        """``search_multiple`` for queries that are already UTF-8 bytes."""
        return self._search_batch(patterns, True)[0]

    def count_multiple_bytes(self, patterns: typing.Sequence[bytes]) -> typing.List[int]:
        """``count_multiple`` for queries that are already bytes."""
        import numpy as np
        nq = len(patterns)
        offs = np.zeros(nq + 1, dtype=np.uint64)
        if nq:
            np.cumsum(np.fromiter(map(len, patterns), dtype=np.uint64, count=nq), out=offs[1:])
        counts = np.zeros(max(nq, 1), dtype=np.uint64)
        _ffi.check(_lib.pss_reader_count_batch(self._handle(), b''.join(patterns), offs.ctypes.data, nq, counts.ctypes.data))
        return [int(c) for c in counts[:nq]]

    def search_batch_raw(self, patterns: typing.Sequence[bytes]):
        """One batched device call.  Returns (entries, per_query_counts): the
        entry byte strings query-major, and how many belong to each query."""
        return self._search_batch(patterns, False)

    def last_stats(self) -> dict:
        st = _ffi.SearchStats()
        _ffi.check(_lib.pss_reader_last_stats(self._handle(), ctypes.byref(st)))
        return st.as_dict()

    def search(self, substring: str) -> typing.List[str]:
        return self._search_batch([_utf8(substring, 'substring')], True)[0]

    def search_multiple(self, substrings: typing.List[str]) -> typing.List[str]:
        return self._search_batch([_utf8(s, 'substring') for s in substrings], True)[0]

    def close(self) -> None:
        if self._h:
            h, self._h = self._h, ctypes.c_void_p()
            _lib.pss_reader_close(h)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
