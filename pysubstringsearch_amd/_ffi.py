"""ctypes binding of libpss.so (C ABI: include/pss.h).

The engine has no CPU fallback: if the shared library is missing this module
raises at import, and every compute call raises ``RuntimeError`` when no HIP
device is usable.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PSS_LIBPSS') or os.path.join(_HERE, 'libpss.so')   # override: kernel A/B experiments

PSS_OK, PSS_EINVAL, PSS_ENOMEM, PSS_EIO, PSS_ETOOBIG, PSS_EDEVICE, PSS_EFORMAT = 0, -1, -2, -3, -4, -5, -6

CORPUS_LINES, CORPUS_WORDS, CORPUS_RUNS, CORPUS_PERIODIC = 0, 1, 2, 3
CORPUS_REPEAT_LINE, CORPUS_DUP_BLOCKS, CORPUS_MIXED, CORPUS_SOURCE = 4, 5, 6, 7


class SaStats(ctypes.Structure):
    _fields_ = [
        ('sigma', ctypes.c_uint32),
        ('code_bits', ctypes.c_uint32),
        ('key_chars', ctypes.c_uint32),
        ('initial_passes', ctypes.c_uint32),
        ('rounds', ctypes.c_uint32),
        ('round_passes', ctypes.c_uint32),
        ('sum_active', ctypes.c_uint64),
        ('sort_elems', ctypes.c_uint64),
        ('ms_total', ctypes.c_double),
        ('ms_sort', ctypes.c_double),
        ('sort_launches', ctypes.c_uint64),
        ('ms_pairs', ctypes.c_double),
        ('pairs_launches', ctypes.c_uint64),
        ('pairs_elems', ctypes.c_uint64),
        ('ms_text', ctypes.c_double),
        ('text_launches', ctypes.c_uint64),
        ('mode', ctypes.c_uint64),
        ('text_rounds', ctypes.c_uint64),
        ('big_elems', ctypes.c_uint64),
        ('key_bits', ctypes.c_uint64),
        ('fs_ms', ctypes.c_double * 9),
        ('fs_launches', ctypes.c_uint64 * 9),
        ('fs_elems', ctypes.c_uint64 * 9),
        ('msd', ctypes.c_uint64),
        ('msd_buckets', ctypes.c_uint64),
        ('msd_max_bucket', ctypes.c_uint64),
        ('msd_tiles', ctypes.c_uint64),
        ('msd_ms_g1', ctypes.c_double),
        ('msd_ms_g2', ctypes.c_double),
        ('msd_ms_local', ctypes.c_double),
        ('msd_slow_tiles', ctypes.c_uint64),
        ('runs', ctypes.c_uint64),
        ('rle', ctypes.c_uint64),
        ('rle_id_bits', ctypes.c_uint64),
        ('rle_ms_table', ctypes.c_double),
        ('rle_ms_reduced', ctypes.c_double),
        ('rle_ms_expand', ctypes.c_double),
        ('ss', ctypes.c_uint64),
        ('ss_buckets', ctypes.c_uint64),
        ('ss_max_bucket', ctypes.c_uint64),
        ('ss_tiles', ctypes.c_uint64),
        ('ss_samples', ctypes.c_uint64),
        ('ss_ms_sample', ctypes.c_double),
        ('ss_ms_g1', ctypes.c_double),
        ('ss_ms_g2', ctypes.c_double),
        ('ss_ms_local', ctypes.c_double),
        ('ms_initial', ctypes.c_double),
        ('period', ctypes.c_uint64),
        ('period_extent', ctypes.c_uint64),
        ('period_path', ctypes.c_uint64),
        ('plan_hint', ctypes.c_uint64),
        ('anchor', ctypes.c_uint64),
        ('anchor_omega', ctypes.c_uint64),
        ('anchor_w', ctypes.c_uint64),
        ('anchor_count', ctypes.c_uint64),
        ('anchor_active', ctypes.c_uint64),
        ('anchor_depth', ctypes.c_uint64),
        ('anchor_text_rounds', ctypes.c_uint64),
        ('anchor_rounds', ctypes.c_uint64),
        ('anchor_sum_active', ctypes.c_uint64),
        ('anchor_left', ctypes.c_uint64),
        ('anchor_levels', ctypes.c_uint64),
        ('probe_pairs', ctypes.c_uint64),
        ('probe_same', ctypes.c_uint64),
        ('periodic_rounds', ctypes.c_uint64),
        ('periodic_members', ctypes.c_uint64),
        ('ss_planned', ctypes.c_uint64),
        ('dup_screen', ctypes.c_uint64),
        ('ss_plan_refused', ctypes.c_uint64),
        ('ss_declined_nomem', ctypes.c_uint64),
        ('anchor_side', ctypes.c_uint64),
        ('anchor_ms', ctypes.c_double),
        ('ms_restarts', ctypes.c_double),
        ('msd_lookback', ctypes.c_uint64),
    ]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        for k in ('fs_ms', 'fs_launches', 'fs_elems'):
            d[k] = list(d[k])
        return d


def knobs():
    """The library's environment switches (csrc/knobs.h) as a list of dicts: name, default, fuzz (values a fuzzer may set), what."""
    out = []
    for i in range(lib.pss_knob_count()):
        f = [ctypes.c_char_p() for _ in range(4)]
        check(lib.pss_knob_info(i, *[ctypes.byref(x) for x in f]))
        name, dflt, fuzz, what = [x.value.decode() for x in f]
        out.append({'name': name, 'default': dflt, 'fuzz': [v for v in fuzz.split('|') if v], 'what': what})
    return out


class SearchStats(ctypes.Structure):
    _fields_ = [
        ('queries', ctypes.c_uint64),
        ('hits', ctypes.c_uint64),
        ('entries', ctypes.c_uint64),
        ('result_bytes', ctypes.c_uint64),
        ('ms_device', ctypes.c_double),
        ('ms_interval', ctypes.c_double),
        ('ms_host', ctypes.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class DeviceResult(ctypes.Structure):
    """pss_device_result (include/pss.h): packed result of one batch left in HBM."""
    _fields_ = [
        ('num_queries', ctypes.c_uint64),
        ('num_entries', ctypes.c_uint64),
        ('num_bytes', ctypes.c_uint64),
        ('d_counts', ctypes.c_void_p),
        ('d_offsets', ctypes.c_void_p),
        ('d_bytes', ctypes.c_void_p),
        ('device', ctypes.c_int32),
    ]


def _preload_hip_runtime() -> None:
    """One HIP runtime per process.  The PyTorch-ROCm wheel bundles its own
    libamdhip64.so (SONAME libamdhip64.so.7) and libhsa-runtime64.so; if libpss
    pulled in /opt/rocm's copy first, a later ``import torch`` would load a
    second runtime that cannot see the GPU.  Loading torch's copy first (by
    path, without importing torch) makes both sides share it; without torch
    the system ROCm runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], 'lib', 'libamdhip64.so')
    if os.path.exists(cand):
        try:
            global _hip
            _hip = ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


_hip = None


def hip_runtime() -> ctypes.CDLL:
    """The HIP runtime this process uses (torch's bundled copy when torch is installed, else the system's): for tools
    and tests that call the runtime themselves next to the library."""
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL('libamdhip64.so')
    return _hip


class WriterIo(ctypes.Structure):
    _fields_ = [('records_mapped', ctypes.c_uint64), ('records_pwritten', ctypes.c_uint64),
                ('ingest_direct_bytes', ctypes.c_uint64), ('ingest_copied_bytes', ctypes.c_uint64)]


class RcclUniqueId(ctypes.Structure):
    _fields_ = [('internal', ctypes.c_char * 128)]


class RcclApiTable(ctypes.Structure):
    """pss_rccl_api (include/pss.h): the collectives library as a table of entry points -- an application that links
    RCCL itself hands them in, the tests make them fail."""
    _vp, _sz, _i = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    GET_UNIQUE_ID = ctypes.CFUNCTYPE(_i, ctypes.POINTER(RcclUniqueId))
    COMM_INIT_RANK = ctypes.CFUNCTYPE(_i, ctypes.POINTER(_vp), _i, RcclUniqueId, _i)
    COMM_FN = ctypes.CFUNCTYPE(_i, _vp)
    ASYNC_ERROR = ctypes.CFUNCTYPE(_i, _vp, ctypes.POINTER(_i))
    GROUP_FN = ctypes.CFUNCTYPE(_i)
    SEND = ctypes.CFUNCTYPE(_i, _vp, _sz, _i, _i, _vp, _vp)
    RECV = ctypes.CFUNCTYPE(_i, _vp, _sz, _i, _i, _vp, _vp)
    ALL_GATHER = ctypes.CFUNCTYPE(_i, _vp, _vp, _sz, _i, _vp, _vp)
    # (the C side sees `const char *(*)(int)`; a c_void_p result lets a Python callback hand back the address of a buffer it
    #  keeps alive -- a c_char_p result makes ctypes warn "memory leak in callback function" on every call)
    ERROR_STRING = ctypes.CFUNCTYPE(ctypes.c_void_p, _i)
    _fields_ = [('get_unique_id', GET_UNIQUE_ID), ('comm_init_rank', COMM_INIT_RANK), ('comm_destroy', COMM_FN),
                ('comm_abort', COMM_FN), ('comm_get_async_error', ASYNC_ERROR), ('group_start', GROUP_FN),
                ('group_end', GROUP_FN), ('send', SEND), ('recv', RECV), ('all_gather', ALL_GATHER),
                ('get_error_string', ERROR_STRING)]


def _load() -> ctypes.CDLL:
    _preload_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} is missing: build the HIP engine first '
            '(python -c "import __graft_entry__ as g; g.build()" or make -C pysubstringsearch_amd/csrc). '
            'pysubstringsearch_amd has no CPU fallback.')
    L = ctypes.CDLL(LIB_PATH, use_errno=True)
    vp, cp, i32, i64, u32, u64 = (ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int32, ctypes.c_int64,
                                  ctypes.c_uint32, ctypes.c_uint64)
    pvp = ctypes.POINTER(vp)
    sig = {
        'pss_device_count': (ctypes.c_int, []),
        'pss_default_devices': (i32, [ctypes.POINTER(i32), i32]),
        'pss_release_workspace': (ctypes.c_int, []),
        'pss_sa_stats_size': (u64, []),
        'pss_search_stats_size': (u64, []),
        'pss_last_error': (ctypes.c_size_t, [cp, ctypes.c_size_t]),
        'pss_sa_build': (i32, [vp, vp, i32, i32]),
        'pss_sa_build_device': (i32, [vp, vp, i32, i32, u32, ctypes.POINTER(SaStats)]),
        'pss_sort_pairs_device': (i32, [vp, vp, u32, i32, i32, ctypes.POINTER(ctypes.c_double)]),
        'pss_writer_open': (ctypes.c_int, [cp, i64, i32, pvp]),
        'pss_writer_open_format': (ctypes.c_int, [cp, i64, i32, i32, pvp]),
        'pss_writer_open_multi': (ctypes.c_int, [cp, i64, ctypes.POINTER(i32), i32, i32, pvp]),
        'pss_writer_add_entry': (ctypes.c_int, [vp, cp, u64]),
        'pss_writer_add_file_lines': (ctypes.c_int, [vp, cp]),
        'pss_writer_dump': (ctypes.c_int, [vp]),
        'pss_writer_finalize': (ctypes.c_int, [vp]),
        'pss_writer_close': (ctypes.c_int, [vp]),
        'pss_writer_chunk_limit': (u64, [vp]),
        'pss_writer_io_stats': (ctypes.c_int, [vp, ctypes.POINTER(WriterIo)]),
        'pss_reader_open': (ctypes.c_int, [cp, i32, i32, i32, pvp]),
        'pss_reader_open_multi': (ctypes.c_int, [cp, ctypes.POINTER(i32), i32, pvp]),
        'pss_reader_set_auto_residency': (ctypes.c_int, [vp, i32]),
        'pss_reader_chunk_tiers': (ctypes.c_int, [vp, vp, u64, ctypes.POINTER(u64)]),
        'pss_reader_evict_chunk': (ctypes.c_int, [vp, u64]),
        'pss_reader_promote_chunk': (ctypes.c_int, [vp, u64]),
        'pss_reader_set_low_latency': (ctypes.c_int, [vp, i32]),
        'pss_reader_set_result_order': (ctypes.c_int, [vp, i32]),
        'pss_reader_result_order': (i32, [vp]),
        'pss_reader_low_latency_stats': (ctypes.c_int, [vp, ctypes.POINTER(u64), ctypes.POINTER(u64)]),
        'pss_reader_create': (ctypes.c_int, [i32, pvp]),
        'pss_reader_add_chunk_device': (ctypes.c_int, [vp, vp, vp, u32]),
        'pss_reader_set_chunk_device': (ctypes.c_int, [vp, u64, vp, vp, u32]),
        'pss_workspace_bytes': (u64, [i32]),
        'pss_knob_count': (i32, []),
        'pss_knob_info': (ctypes.c_int, [i32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p),
                                           ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p)]),
        'pss_reader_num_chunks': (u64, [vp]),
        'pss_reader_part_chunks': (u64, [vp, ctypes.POINTER(u64), u64]),
        'pss_reader_residency': (ctypes.c_int, [vp, ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(u64)]),
        'pss_reader_search_batch': (ctypes.c_int, [vp, vp, vp, u32, pvp]),
        'pss_reader_count_batch': (ctypes.c_int, [vp, vp, vp, u32, vp]),
        'pss_reader_search_batch_device': (ctypes.c_int, [vp, vp, vp, u32, ctypes.POINTER(DeviceResult)]),
        'pss_merge_packed': (ctypes.c_int, [u32, u64, vp, vp, vp, vp, vp, vp, vp, vp]),
        'pss_merge_packed_device': (ctypes.c_int, [i32, u32, u64, vp, vp, vp, vp, vp, vp, vp, vp]),
        'pss_comm_unique_id': (ctypes.c_int, [vp]),
        'pss_comm_init': (ctypes.c_int, [vp, i32, i32, i32, pvp]),
        'pss_comm_destroy': (ctypes.c_int, [vp]),
        'pss_gather_packed_rccl': (ctypes.c_int, [vp, ctypes.POINTER(DeviceResult), i32, pvp]),
        'pss_comm_set_timeout_ms': (ctypes.c_int, [vp, u32]),
        'pss_comm_status': (ctypes.c_int, [vp, ctypes.POINTER(u64), ctypes.POINTER(u64)]),
        'pss_rccl_inject': (ctypes.c_int, [ctypes.POINTER(RcclApiTable)]),
        'pss_comm_adopt': (ctypes.c_int, [vp, i32, i32, i32, pvp]),
        'pss_reload_env': (ctypes.c_int, []),
        'pss_reader_last_stats': (ctypes.c_int, [vp, ctypes.POINTER(SearchStats)]),
        'pss_reader_close': (ctypes.c_int, [vp]),
        'pss_result_num_queries': (u64, [vp]),
        'pss_result_num_entries': (u64, [vp]),
        'pss_result_query_counts': (ctypes.POINTER(u64), [vp]),
        'pss_result_offsets': (ctypes.POINTER(u64), [vp]),
        'pss_result_bytes': (ctypes.POINTER(ctypes.c_uint8), [vp]),
        'pss_result_free': (None, [vp]),
        'pss_gen_corpus': (ctypes.c_int, [ctypes.c_int, vp, u64, u64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    # the stats structs are declared twice (include/pss.h and above): a field added on one side only would let the
    # library write past the end of the Python object
    for what, have, want in (('pss_sa_stats', ctypes.sizeof(SaStats), L.pss_sa_stats_size()),
                             ('pss_search_stats', ctypes.sizeof(SearchStats), L.pss_search_stats_size())):
        if have != want:
            raise ImportError(f'{LIB_PATH}: {what} is {want} bytes in the library and {have} in _ffi.py -- rebuild the '
                              'library (make -C pysubstringsearch_amd/csrc) or update the binding')
    return L


lib = _load()


def last_error() -> str:
    buf = ctypes.create_string_buffer(2048)
    lib.pss_last_error(buf, len(buf))
    return buf.value.decode('utf-8', 'replace')


def check(rc: int, what: str = '') -> None:
    """Maps a C-ABI status to the exception the reference raises for it
    (io::Error -> OSError by errno, src/lib.rs:55,71; ValueError, src/lib.rs:93)."""
    if rc == PSS_OK:
        return
    msg = last_error()
    if rc == PSS_EIO:
        e = ctypes.get_errno()
        raise OSError(e, os.strerror(e), what or None)
    if rc == PSS_ETOOBIG:
        raise ValueError('entry is too big')
    if rc == PSS_ENOMEM:
        raise MemoryError(msg)
    if rc == PSS_EFORMAT:
        raise OSError(msg)
    if rc == PSS_EINVAL:
        raise ValueError(msg)
    raise RuntimeError(msg or f'libpss error {rc}')
