"""ctypes front-end of the CPU oracle (oracle/pss_oracle.c) and of the real
reference suffix-array builder (oracle/_ref/libsais.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py -- never by ``pysubstringsearch_amd``.

``OracleWriter`` / ``OracleReader`` mirror the reference Python classes
(/root/reference/pysubstringsearch/__init__.py:6-73) so parity tests read like
the reference's own tests.
"""
import ctypes
import os
import subprocess
import typing

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'liboracle.so')
_REF_PATH = os.path.join(_HERE, '_ref', 'libsais.so')

ORC_OK, ORC_EINVAL, ORC_ENOMEM, ORC_EIO, ORC_ETOOBIG = 0, -1, -2, -3, -4


def build(force: bool = False) -> None:
    """Compile liboracle.so (and _ref/libsais.so when /root/reference exists)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(os.path.join(_HERE, 'pss_oracle.c')) > os.path.getmtime(_LIB_PATH)
    ) or (not os.path.exists(_REF_PATH) and os.path.exists('/root/reference/src/libsais/libsais.c')):
        subprocess.run(['make', '-C', _HERE], check=True, capture_output=True)


_lib = None
_ref = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH, use_errno=True)
        vp, u8p = ctypes.c_void_p, ctypes.c_char_p
        L.orc_sa_build.argtypes = [vp, vp, ctypes.c_int32]
        L.orc_sa_build.restype = ctypes.c_int32
        L.orc_set_external_sa.argtypes = [vp]
        L.orc_set_external_sa.restype = None
        L.orc_writer_open.argtypes = [u8p, ctypes.c_int64, ctypes.POINTER(vp)]
        L.orc_writer_add_entry.argtypes = [vp, u8p, ctypes.c_size_t]
        L.orc_writer_add_file_lines.argtypes = [vp, u8p]
        L.orc_writer_dump.argtypes = [vp]
        L.orc_writer_finalize.argtypes = [vp]
        L.orc_writer_close.argtypes = [vp]
        L.orc_writer_capacity.argtypes = [vp]
        L.orc_writer_capacity.restype = ctypes.c_size_t
        L.orc_reader_open.argtypes = [u8p, ctypes.POINTER(vp)]
        L.orc_reader_open_ex.argtypes = [u8p, ctypes.c_int, ctypes.POINTER(vp)]
        L.orc_reader_from_arrays.argtypes = [ctypes.c_size_t, vp, vp, vp, ctypes.POINTER(vp)]
        L.orc_bench_search.argtypes = [vp, vp, vp, ctypes.c_uint32, ctypes.c_int, ctypes.c_int,
                                       ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64),
                                       ctypes.POINTER(ctypes.c_uint64), vp]
        L.orc_reader_close.argtypes = [vp]
        L.orc_reader_close.restype = None
        L.orc_reader_num_chunks.argtypes = [vp]
        L.orc_reader_num_chunks.restype = ctypes.c_size_t
        L.orc_reader_search.argtypes = [vp, u8p, ctypes.c_size_t, ctypes.POINTER(vp)]
        L.orc_reader_search_multiple.argtypes = [vp, vp, vp, ctypes.c_uint32, vp, ctypes.POINTER(vp)]
        L.orc_result_count.argtypes = [vp]
        L.orc_result_count.restype = ctypes.c_size_t
        L.orc_result_offsets.argtypes = [vp]
        L.orc_result_offsets.restype = ctypes.POINTER(ctypes.c_uint64)
        L.orc_result_bytes.argtypes = [vp]
        L.orc_result_bytes.restype = ctypes.POINTER(ctypes.c_uint8)
        L.orc_result_free.argtypes = [vp]
        L.orc_result_free.restype = None
        L.orc_reader_chunk_data.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        L.orc_reader_chunk_data.restype = ctypes.POINTER(ctypes.c_uint8)
        L.orc_reader_chunk_sa.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        L.orc_reader_chunk_sa.restype = ctypes.POINTER(ctypes.c_int32)
        L.orc_set_hash_dedupe.argtypes = [ctypes.c_int]
        L.orc_set_hash_dedupe.restype = None
        L.orc_get_hash_dedupe.restype = ctypes.c_int
        L.orc_gen_lines.argtypes = [vp, ctypes.c_size_t, ctypes.c_uint64]
        L.orc_gen_lines.restype = None
        _lib = L
    return _lib


def have_reference() -> bool:
    build()
    return os.path.exists(_REF_PATH)


def ref() -> ctypes.CDLL:
    """The real libsais, signature per /root/reference/src/lib.rs:14-22."""
    global _ref
    if _ref is None:
        if not have_reference():
            raise RuntimeError('oracle/_ref/libsais.so not built (reference tree absent)')
        R = ctypes.CDLL(_REF_PATH)
        R.libsais.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
        R.libsais.restype = ctypes.c_int32
        _ref = R
    return _ref


def use_reference_sa(enable: bool) -> None:
    """Make OracleWriter build its suffix arrays with the real libsais."""
    if enable:
        fn = ctypes.cast(ref().libsais, ctypes.c_void_p)
        lib().orc_set_external_sa(fn)
    else:
        lib().orc_set_external_sa(None)


def _as_u8(data) -> np.ndarray:
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


def sa_restatement(data) -> np.ndarray:
    """Suffix array by the oracle's own prefix-doubling routine."""
    t = _as_u8(data)
    sa = np.empty(t.size, dtype=np.int32)
    rc = lib().orc_sa_build(t.ctypes.data, sa.ctypes.data, t.size)
    if rc:
        raise RuntimeError(f'orc_sa_build failed: {rc}')
    return sa


def sa_reference(data) -> np.ndarray:
    """Suffix array by the real libsais, called as lib.rs:30-36 does (fs=0, freq=NULL)."""
    t = _as_u8(data)
    sa = np.empty(t.size, dtype=np.int32)
    rc = ref().libsais(t.ctypes.data, sa.ctypes.data, t.size, 0, None)
    if rc:
        raise RuntimeError(f'libsais failed: {rc}')
    return sa


def sa(data) -> np.ndarray:
    """Best available oracle SA: libsais when built, else the restatement."""
    return sa_reference(data) if have_reference() else sa_restatement(data)


def set_hash_dedupe(on: bool) -> None:
    """Per-chunk dedupe of the search restatement: hash set (lib.rs:262 as written) or the sorted scratch list the
    checker has used since round 1.  Same results, same order; tests run both."""
    lib().orc_set_hash_dedupe(1 if on else 0)


def gen_lines(n: int, chunk_index: int = 0) -> np.ndarray:
    out = np.empty(n, dtype=np.uint8)
    lib().orc_gen_lines(out.ctypes.data, n, chunk_index)
    return out


def _raise(rc: int, what: str):
    if rc == ORC_EIO:
        e = ctypes.get_errno()
        raise OSError(e, os.strerror(e), what)
    if rc == ORC_ETOOBIG:
        raise ValueError('entry is too big')
    if rc == ORC_ENOMEM:
        raise MemoryError(what)
    raise RuntimeError(f'{what}: oracle error {rc}')


class OracleWriter:
    def __init__(self, index_file_path: str, max_chunk_len: typing.Optional[int] = None) -> None:
        self._h = ctypes.c_void_p()
        rc = lib().orc_writer_open(
            os.fsencode(index_file_path), -1 if max_chunk_len is None else max_chunk_len, ctypes.byref(self._h))
        if rc:
            _raise(rc, index_file_path)

    def add_entries_from_file_lines(self, input_file_path: str) -> None:
        rc = lib().orc_writer_add_file_lines(self._h, os.fsencode(input_file_path))
        if rc:
            _raise(rc, input_file_path)

    def add_entry(self, text: str) -> None:
        b = text.encode('utf-8')
        rc = lib().orc_writer_add_entry(self._h, b, len(b))
        if rc:
            _raise(rc, 'add_entry')

    def dump_data(self) -> None:
        rc = lib().orc_writer_dump(self._h)
        if rc:
            _raise(rc, 'dump_data')

    def finalize(self) -> None:
        rc = lib().orc_writer_finalize(self._h)
        if rc:
            _raise(rc, 'finalize')

    @property
    def capacity(self) -> int:
        return lib().orc_writer_capacity(self._h)

    def close(self) -> None:
        if self._h:
            lib().orc_writer_close(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OracleReader:
    def __init__(self, index_file_path: str, load_sa: bool = True) -> None:
        """load_sa=False leaves the suffix arrays in the file, as the reference does
        (lib.rs:179-182); such a reader only serves bench_search(disk=True)."""
        self._h = ctypes.c_void_p()
        self._keep = None
        rc = lib().orc_reader_open_ex(os.fsencode(index_file_path), 1 if load_sa else 0, ctypes.byref(self._h))
        if rc:
            _raise(rc, index_file_path)

    @classmethod
    def from_arrays(cls, texts: typing.Sequence[np.ndarray], sas: typing.Sequence[np.ndarray]) -> 'OracleReader':
        """Reader over in-memory chunks (uint8 text + int32 suffix array each); the arrays
        are borrowed and kept alive by the reader object."""
        assert len(texts) == len(sas)
        texts = [np.ascontiguousarray(t, dtype=np.uint8) for t in texts]
        sas = [np.ascontiguousarray(s, dtype=np.int32) for s in sas]
        for t, s in zip(texts, sas):
            assert t.size == s.size
        k = len(texts)
        tp = (ctypes.c_void_p * max(k, 1))(*[t.ctypes.data for t in texts])
        sp = (ctypes.c_void_p * max(k, 1))(*[s.ctypes.data for s in sas])
        ln = (ctypes.c_uint64 * max(k, 1))(*[t.size for t in texts])
        r = cls.__new__(cls)
        r._h = ctypes.c_void_p()
        r._keep = (texts, sas)
        rc = lib().orc_reader_from_arrays(k, tp, ln, sp, ctypes.byref(r._h))
        if rc:
            _raise(rc, 'from_arrays')
        return r

    def bench_search(self, patterns: typing.Sequence[bytes], threads: int, disk: bool = False, dedupe: str = 'hash') -> dict:
        """The reference-shaped CPU baseline: queries one at a time, each
        fanned out over the chunks on `threads` workers (lib.rs:207), suffix array probed in
        RAM or -- disk=True -- in the index file with lseek + read(8 KiB) per probe
        (lib.rs:216-217).  Returns seconds, entries, bytes and the per-query counts.
        dedupe='hash' (default): the per-chunk dedupe is a hash set, like the reference's AHashSet (lib.rs:262);
        'sort': the checker's sorted scratch list (O(h log h): slower on high-hit queries, same output)."""
        assert dedupe in ('hash', 'sort')
        was = lib().orc_get_hash_dedupe()
        lib().orc_set_hash_dedupe(1 if dedupe == 'hash' else 0)
        try:
            return self._bench_search(patterns, threads, disk, dedupe)
        finally:
            lib().orc_set_hash_dedupe(was)

    def _bench_search(self, patterns, threads, disk, dedupe) -> dict:
        blob = b''.join(patterns)
        off = np.zeros(len(patterns) + 1, dtype=np.uint64)
        if patterns:
            off[1:] = np.cumsum([len(p) for p in patterns], dtype=np.uint64)
        buf = np.frombuffer(blob + b'\0', dtype=np.uint8)
        counts = np.zeros(max(len(patterns), 1), dtype=np.uint64)
        sec, ent, byt = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64()
        rc = lib().orc_bench_search(self._h, buf.ctypes.data, off.ctypes.data, len(patterns), threads, 1 if disk else 0,
                                    ctypes.byref(sec), ctypes.byref(ent), ctypes.byref(byt), counts.ctypes.data)
        if rc:
            _raise(rc, 'bench_search')
        return {'seconds': sec.value, 'entries': ent.value, 'bytes': byt.value, 'counts': counts[:len(patterns)],
                'threads': min(threads, max(self.num_chunks, 1)), 'disk': bool(disk), 'dedupe': dedupe}

    @property
    def num_chunks(self) -> int:
        return lib().orc_reader_num_chunks(self._h)

    def chunk(self, c: int) -> typing.Tuple[bytes, np.ndarray]:
        n = ctypes.c_size_t()
        p = lib().orc_reader_chunk_data(self._h, c, ctypes.byref(n))
        data = ctypes.string_at(p, n.value)
        p = lib().orc_reader_chunk_sa(self._h, c, ctypes.byref(n))
        sa_ = np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.empty(0, np.int32)
        return data, sa_

    @staticmethod
    def _unpack(res) -> typing.List[bytes]:
        L = lib()
        n = L.orc_result_count(res)
        out = []
        if n:
            off = L.orc_result_offsets(res)
            base = L.orc_result_bytes(res)
            # not ctypes.string_at: its size argument is a C int (results beyond 2 GiB)
            blob = bytes((ctypes.c_char * off[n]).from_address(ctypes.cast(base, ctypes.c_void_p).value)) if off[n] else b''
            out = [blob[off[i]:off[i + 1]] for i in range(n)]
        L.orc_result_free(res)
        return out

    def search_bytes(self, pattern: bytes) -> typing.List[bytes]:
        res = ctypes.c_void_p()
        rc = lib().orc_reader_search(self._h, pattern, len(pattern), ctypes.byref(res))
        if rc:
            _raise(rc, 'search')
        return self._unpack(res)

    def search(self, substring: str) -> typing.List[str]:
        return [b.decode('utf-8') for b in self.search_bytes(substring.encode('utf-8'))]

    def search_multiple_bytes(self, patterns: typing.Sequence[bytes]) -> typing.Tuple[typing.List[bytes], np.ndarray]:
        blob = b''.join(patterns)
        off = np.zeros(len(patterns) + 1, dtype=np.uint64)
        if patterns:
            off[1:] = np.cumsum([len(p) for p in patterns], dtype=np.uint64)
        counts = np.zeros(max(len(patterns), 1), dtype=np.uint64)
        buf = np.frombuffer(blob + b'\0', dtype=np.uint8)
        res = ctypes.c_void_p()
        rc = lib().orc_reader_search_multiple(
            self._h, buf.ctypes.data, off.ctypes.data, len(patterns), counts.ctypes.data, ctypes.byref(res))
        if rc:
            _raise(rc, 'search_multiple')
        return self._unpack(res), counts[:len(patterns)]

    def search_multiple(self, substrings: typing.List[str]) -> typing.List[str]:
        ents, _ = self.search_multiple_bytes([s.encode('utf-8') for s in substrings])
        return [b.decode('utf-8') for b in ents]

    def close(self) -> None:
        if self._h:
            lib().orc_reader_close(self._h)
            self._h = ctypes.c_void_p()
        self._keep = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
