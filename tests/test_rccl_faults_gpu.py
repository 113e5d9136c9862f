"""The RCCL gather inside the C ABI (pss_gather_packed_rccl, include/pss.h) when the collectives library misbehaves.

A one-GPU box cannot host two RCCL ranks, so the peer is played by an injected table of entry points (pss_rccl_inject +
pss_comm_adopt -- the seam an application that links RCCL itself would use): a world of two in which this process is
rank 0 (the collecting rank) or rank 1 (a contributing one), the other rank's buffers come out of numpy arrays, and
Send / Recv / GroupEnd / AllGather can be told to fail, a Recv to never complete, RCCL's asynchronous error to fire.
What must hold: the status code, no hang (bounded by PSS_RCCL_TIMEOUT_MS / pss_comm_set_timeout_ms), every opened group
closed, a timed-out communicator aborted and refusing further calls, readers and NEW communicators working afterwards.
The path these replace in the reference: rayon tasks appending to one Vec under a mutex, src/lib.rs:205-207, 280-284.
"""
import ctypes
import threading
import time

import numpy as np
import pytest

import pysubstringsearch
from pysubstringsearch_amd import _ffi

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(120)]

H2D, D2D = 1, 3
U64 = 5


@pytest.fixture(scope='module')
def hip():
    h = _ffi.hip_runtime()
    vp = ctypes.c_void_p
    h.hipMemcpyAsync.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_int, vp]
    h.hipMemsetAsync.argtypes = [vp, ctypes.c_int, ctypes.c_size_t, vp]
    h.hipLaunchHostFunc.argtypes = [vp, vp, vp]
    h.hipStreamWaitValue32.argtypes = [vp, vp, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
    h.hipHostMalloc.argtypes = [ctypes.POINTER(vp), ctypes.c_size_t, ctypes.c_uint]
    h.hipHostFree.argtypes = [vp]
    return h


class FakePeer:
    """The other rank of a world of two, and the switches that break it."""

    def __init__(self, hip, my_rank, peer_counts, peer_starts, peer_bytes):
        self.hip, self.me, self.peer = hip, my_rank, 1 - my_rank
        self.counts = np.ascontiguousarray(peer_counts, dtype=np.uint64)
        self.starts = np.ascontiguousarray(peer_starts, dtype=np.uint64)
        self.bytes = np.ascontiguousarray(peer_bytes, dtype=np.uint8)
        self.fail = None                  # 'send' | 'recv' | 'group_end' | 'all_gather' | 'hang' | 'async'
        self.peer_go = 0                  # the peer's go / no-go word
        self.peer_nq_delta = 0
        self.calls = {k: 0 for k in ('send', 'recv', 'group_start', 'group_end', 'all_gather', 'abort', 'async')}
        self.release = threading.Event()  # frees a Recv that "never completes" (the abort sets it)
        # A Recv that never completes is a wait INSIDE the stream -- hipStreamWaitValue32 on a word of pinned memory that
        # the abort sets: what a collective kernel waiting for its peer is to the rest of the device.  (Where the runtime
        # has no stream memory operations, a host function that blocks stands in; it also holds up the runtime's helper
        # thread, so the test about other work on the device skips there.)
        self.flag = ctypes.c_void_p()
        self.device_wait = hip.hipHostMalloc(ctypes.byref(self.flag), 64, 0) == 0
        if self.device_wait:
            ctypes.memset(self.flag.value, 0, 64)
        self.sent = []
        self._keep = []
        self._recv_seq = 0
        T = _ffi.RcclApiTable
        self._blocker = ctypes.CFUNCTYPE(None, ctypes.c_void_p)(self._block)
        self.table = T(T.GET_UNIQUE_ID(), T.COMM_INIT_RANK(), T.COMM_FN(), T.COMM_FN(self._abort), T.ASYNC_ERROR(self._async),
                       T.GROUP_FN(self._group_start), T.GROUP_FN(self._group_end), T.SEND(self._send), T.RECV(self._recv),
                       T.ALL_GATHER(self._all_gather), T.ERROR_STRING(self._errstr))

    # -- table entries --------------------------------------------------------------------------------------------
    _ERRSTR = ctypes.create_string_buffer(b'injected failure')

    def _errstr(self, rc):
        return ctypes.addressof(self._ERRSTR)

    def _group_start(self):
        self.calls['group_start'] += 1
        return 0

    def _group_end(self):
        self.calls['group_end'] += 1
        return 3 if self.fail == 'group_end' else 0

    def _abort(self, comm):
        self.calls['abort'] += 1
        self.release.set()
        if self.flag.value:
            ctypes.c_uint32.from_address(self.flag.value).value = 1
        return 0

    def _async(self, comm, perr):
        self.calls['async'] += 1
        perr[0] = 6 if self.fail == 'async' else 0
        return 0

    def _block(self, _):
        self.release.wait(30.0)           # (bounded whatever happens: the stream must not stay blocked for good)

    def _h2d(self, dst, arr, stream):
        self._keep.append(arr)
        assert self.hip.hipMemcpyAsync(dst, arr.ctypes.data, arr.nbytes, H2D, stream) == 0

    def _all_gather(self, send, recv, count, dtype, comm, stream):
        self.calls['all_gather'] += 1
        if self.fail == 'all_gather':
            return 2
        assert dtype == U64
        assert self.hip.hipMemcpyAsync(recv + self.me * count * 8, send, count * 8, D2D, stream) == 0
        if count == 4:                    # (entries, bytes, queries, 0)
            blk = np.array([self.starts.size, self.bytes.size, self.counts.size + self.peer_nq_delta, 0], dtype=np.uint64)
        else:                             # go / no-go
            assert count == 1
            blk = np.array([self.peer_go], dtype=np.uint64)
        self._h2d(recv + self.peer * count * 8, blk, stream)
        return 0

    def _send(self, buf, count, dtype, peer, comm, stream):
        self.calls['send'] += 1
        if self.fail == 'send':
            return 2
        assert peer == self.peer
        self.sent.append((count, dtype))
        return 0

    def _recv(self, buf, count, dtype, peer, comm, stream):
        self.calls['recv'] += 1
        if self.fail == 'recv':
            return 2
        assert peer == self.peer
        if self.fail in ('hang', 'async'):
            if self.device_wait and self.hip.hipStreamWaitValue32(stream, self.flag, 1, 1, 0xffffffff) != 0:      # 1 = equal
                self.device_wait = False
            if not self.device_wait:
                assert self.hip.hipLaunchHostFunc(stream, ctypes.cast(self._blocker, ctypes.c_void_p), None) == 0
        seq = [a for a in (self.counts, self.starts, self.bytes) if a.size]      # the order the collecting rank posts them in
        src = seq[self._recv_seq % len(seq)]
        self._recv_seq += 1
        assert src.size == count
        self._h2d(buf, src, stream)
        return 0


def _packed(reader, qs):
    pk = reader.search_batch_packed(qs)
    return np.array(pk.counts), np.array(pk.offsets), np.array(pk.data)


def _device_result(reader, qs):
    nq = len(qs)
    blob = b''.join(qs)
    offs = np.zeros(nq + 1, dtype=np.uint64)
    np.cumsum(np.fromiter(map(len, qs), dtype=np.uint64, count=nq), out=offs[1:])
    dr = _ffi.DeviceResult()
    _ffi.check(_ffi.lib.pss_reader_search_batch_device(reader._handle(), blob, offs.ctypes.data, nq, ctypes.byref(dr)))
    return dr


def _gather(comm, dr, dst=0):
    res = ctypes.c_void_p()
    rc = _ffi.lib.pss_gather_packed_rccl(comm, ctypes.byref(dr), dst, ctypes.byref(res))
    return rc, res


def _status(comm):
    g, a = ctypes.c_uint64(), ctypes.c_uint64()
    return _ffi.lib.pss_comm_status(comm, ctypes.byref(g), ctypes.byref(a)), g.value, a.value


def _adopt(peer, rank, timeout_ms=20000):
    _ffi.check(_ffi.lib.pss_rccl_inject(ctypes.byref(peer.table)))
    comm = ctypes.c_void_p()
    _ffi.check(_ffi.lib.pss_comm_adopt(ctypes.c_void_p(0x1234), 2, rank, 0, ctypes.byref(comm)))
    _ffi.check(_ffi.lib.pss_comm_set_timeout_ms(comm, timeout_ms))
    return comm


@pytest.fixture()
def two_indexes(tmp_path):
    """Two small indexes -- "my" chunks and the peer's -- and a batch that hits both."""
    rng = np.random.default_rng(5)
    words = ['alpha', 'beta', 'gamma', 'delta', 'epsilon', 'zeta', 'eta', 'theta']
    paths = []
    for k in range(2):
        p = str(tmp_path / f'part{k}.idx')
        w = pysubstringsearch.Writer(p, 2048, device=0)
        for i in range(400):
            w.add_entry(' '.join(words[int(x)] for x in rng.integers(0, 8, size=int(rng.integers(1, 6)))) + f' {k}:{i}')
        w.close()
        paths.append(p)
    qs = [b'alpha', b'eta', b'zz', b'a b', b'gamma delta', b'', b'1:3', b'0:39']
    yield paths, qs
    _ffi.lib.pss_rccl_inject(None)


def _expected_merge(mine, theirs, me_first=True):
    (c0, o0, d0), (c1, o1, d1) = (mine, theirs) if me_first else (theirs, mine)
    out, counts = [], []
    p0 = p1 = 0
    for q in range(len(c0)):
        for c, o, d, p in ((c0, o0, d0, p0), (c1, o1, d1, p1)):
            for e in range(p, p + int(c[q])):
                out.append(bytes(d[int(o[e]):int(o[e + 1])]))
        p0 += int(c0[q])
        p1 += int(c1[q])
        counts.append(int(c0[q]) + int(c1[q]))
    return out, counts


def _unpack(res, nq):
    n = _ffi.lib.pss_result_num_entries(res)
    counts = np.ctypeslib.as_array(_ffi.lib.pss_result_query_counts(res), shape=(nq,)).astype(np.int64).tolist()
    off = np.ctypeslib.as_array(_ffi.lib.pss_result_offsets(res), shape=(n + 1,)).astype(np.int64)
    data = bytes(np.ctypeslib.as_array(_ffi.lib.pss_result_bytes(res), shape=(max(int(off[n]), 1),))[:int(off[n])])
    _ffi.lib.pss_result_free(res)
    return [data[off[i]:off[i + 1]] for i in range(n)], counts


def test_gather_through_an_injected_table_collecting_rank(hip, two_indexes):
    paths, qs = two_indexes
    with pysubstringsearch.Reader(paths[0], device=0) as mine, pysubstringsearch.Reader(paths[1], device=0) as other:
        pm, pt = _packed(mine, qs), _packed(other, qs)
        peer = FakePeer(hip, 0, pt[0], pt[1][:-1], pt[2])
        comm = _adopt(peer, 0)
        rc, res = _gather(comm, _device_result(mine, qs))
        assert rc == 0, _ffi.last_error()
        got, counts = _unpack(res, len(qs))
        want, wcounts = _expected_merge(pm, pt)
        assert got == want and counts == wcounts and sum(counts) > 100
        assert peer.calls['group_start'] == peer.calls['group_end'] == 1 and peer.calls['all_gather'] == 2
        assert _status(comm) == (0, 1, 0)
        _ffi.lib.pss_comm_destroy(comm)


def test_gather_through_an_injected_table_contributing_rank(hip, two_indexes):
    paths, qs = two_indexes
    with pysubstringsearch.Reader(paths[1], device=0) as mine:
        pm = _packed(mine, qs)
        peer = FakePeer(hip, 1, np.zeros(len(qs)), np.zeros(0), np.zeros(0))
        comm = _adopt(peer, 1)
        res = ctypes.c_void_p()
        dr = _device_result(mine, qs)
        rc = _ffi.lib.pss_gather_packed_rccl(comm, ctypes.byref(dr), 0, ctypes.byref(res))
        assert rc == 0 and not res.value
        assert [c for c, _ in peer.sent] == [len(qs), len(pm[1]) - 1, len(pm[2])]       # counts, entry starts, bytes
        assert peer.calls['recv'] == 0 and peer.calls['group_start'] == peer.calls['group_end'] == 1
        _ffi.lib.pss_comm_destroy(comm)


@pytest.mark.parametrize('what,rank', [('send', 1), ('recv', 0), ('group_end', 0), ('group_end', 1), ('all_gather', 0)])
def test_a_failing_entry_point_is_an_error_code_and_the_group_is_closed(hip, two_indexes, what, rank):
    paths, qs = two_indexes
    with pysubstringsearch.Reader(paths[rank], device=0) as mine, pysubstringsearch.Reader(paths[1 - rank], device=0) as other:
        pt = _packed(other, qs)
        peer = FakePeer(hip, rank, pt[0], pt[1][:-1], pt[2])
        comm = _adopt(peer, rank)
        peer.fail = what
        rc, res = _gather(comm, _device_result(mine, qs))
        assert rc == _ffi.PSS_EDEVICE and not res.value
        assert 'injected failure' in _ffi.last_error()
        assert peer.calls['group_start'] == peer.calls['group_end']          # never left inside a group
        assert _status(comm)[0] == 0                                          # a synchronous error does not abort the communicator
        # ... and the same communicator serves the next call
        peer.fail = None
        peer._recv_seq = 0
        rc, res = _gather(comm, _device_result(mine, qs))
        assert rc == 0, _ffi.last_error()
        if rank == 0:
            got, _ = _unpack(res, len(qs))
            assert got == _expected_merge(_packed(mine, qs), pt)[0]
        assert sorted(mine.search('alpha')) == sorted(e.decode() for e in _unpack_list(mine, b'alpha'))
        _ffi.lib.pss_comm_destroy(comm)


def _unpack_list(reader, q):
    c, o, d = _packed(reader, [q])
    return [bytes(d[int(o[i]):int(o[i + 1])]) for i in range(int(c[0]))]


@pytest.mark.parametrize('mode', ['hang', 'async'])
def test_a_recv_that_never_completes_is_cut_off_and_the_communicator_aborted(hip, two_indexes, mode):
    paths, qs = two_indexes
    with pysubstringsearch.Reader(paths[0], device=0) as mine, pysubstringsearch.Reader(paths[1], device=0) as other:
        pt = _packed(other, qs)
        peer = FakePeer(hip, 0, pt[0], pt[1][:-1], pt[2])
        comm = _adopt(peer, 0, timeout_ms=400 if mode == 'hang' else 20000)
        peer.fail = mode
        t0 = time.time()
        rc, res = _gather(comm, _device_result(mine, qs))
        took = time.time() - t0
        assert rc == _ffi.PSS_EDEVICE and not res.value
        msg = _ffi.last_error()
        assert 'aborted' in msg and ('within 400 ms' in msg if mode == 'hang' else 'asynchronous RCCL error' in msg), msg
        assert took < 10.0 and (mode == 'async' or took >= 0.4)
        assert peer.calls['abort'] == 1 and peer.release.is_set()
        assert _status(comm) == (_ffi.PSS_EDEVICE, 0, 1)
        # the dead communicator refuses at once, without touching the library
        before = dict(peer.calls)
        rc, res = _gather(comm, _device_result(mine, qs))
        assert rc == _ffi.PSS_EDEVICE and 'aborted by an earlier failure' in _ffi.last_error() and peer.calls == before
        _ffi.lib.pss_comm_destroy(comm)
        # the reader on that device is unharmed, and a NEW communicator gathers
        assert sorted(mine.search('beta')) == sorted(e.decode() for e in _unpack_list(mine, b'beta'))
        peer2 = FakePeer(hip, 0, pt[0], pt[1][:-1], pt[2])
        comm2 = _adopt(peer2, 0)
        rc, res = _gather(comm2, _device_result(mine, qs))
        assert rc == 0, _ffi.last_error()
        assert _unpack(res, len(qs))[0] == _expected_merge(_packed(mine, qs), pt)[0]
        _ffi.lib.pss_comm_destroy(comm2)


def test_go_no_go_is_collective(hip, two_indexes):
    """A rank that cannot go through with the exchange says so BEFORE anybody sends: the collecting rank out of memory
    used to return without posting its receives and leave the others inside ncclSend."""
    paths, qs = two_indexes
    with pysubstringsearch.Reader(paths[1], device=0) as mine:
        peer = FakePeer(hip, 1, np.zeros(len(qs)), np.zeros(0), np.zeros(0))
        comm = _adopt(peer, 1)
        peer.peer_go = 2                                           # the collecting rank: -PSS_ENOMEM
        rc, res = _gather(comm, _device_result(mine, qs))
        assert rc == _ffi.PSS_ENOMEM and 'rank 0 gave up' in _ffi.last_error()
        assert peer.calls['send'] == 0 and peer.calls['group_start'] == 0
        # ranks that answered different numbers of queries: every rank sees the same table and refuses
        peer.peer_go, peer.peer_nq_delta = 0, 1
        rc, res = _gather(comm, _device_result(mine, qs))
        assert rc == _ffi.PSS_EINVAL and 'queries' in _ffi.last_error() and peer.calls['send'] == 0
        assert _status(comm)[0] == 0
        _ffi.lib.pss_comm_destroy(comm)


def test_a_waiting_gather_holds_no_lock(hip, two_indexes):
    """While a gather waits for a peer that never answers it holds no device context: calls that only need the reader
    side's lock return at once (round 4 held it across the collectives: everything on the device queued behind a dead
    peer), and a suffix-array build -- which has a context of its own -- and a search run to their end.  How soon GPU
    work of OTHER streams gets through is the runtime's business: HIP maps its streams onto a handful of hardware
    queues, and a stream that shares its queue with the blocked one waits with it -- bounded, like the gather itself, by
    the communicator's timeout."""
    paths, qs = two_indexes
    with pysubstringsearch.Reader(paths[0], device=0) as mine, pysubstringsearch.Reader(paths[1], device=0) as other:
        pt = _packed(other, qs)
        peer = FakePeer(hip, 0, pt[0], pt[1][:-1], pt[2])
        comm = _adopt(peer, 0, timeout_ms=3000)
        peer.fail = 'hang'
        n = 1 << 20
        text = np.empty(n, dtype=np.uint8)
        _ffi.check(_ffi.lib.pss_gen_corpus(_ffi.CORPUS_WORDS, text.ctypes.data, n, 0))
        sa = np.empty(n, dtype=np.int32)
        _ffi.check(_ffi.lib.pss_sa_build(text.ctypes.data, sa.ctypes.data, n, 0))      # (workspaces grown beforehand)
        other.search('gamma')
        out = {}
        dr = _device_result(mine, qs)
        th = threading.Thread(target=lambda: out.setdefault('rc', _gather(comm, dr)[0]))
        th.start()
        time.sleep(0.3)                                             # the gather is blocked in its receive by now
        t0 = time.time()
        for _ in range(50):
            other.low_latency_stats()                               # takes the reader side's lock of device 0, nothing else
            assert other.num_chunks > 0
        locked_for = time.time() - t0
        assert th.is_alive() and locked_for < 0.5, locked_for       # ... which the waiting gather does not hold
        sa[:] = 0
        t0 = time.time()
        _ffi.check(_ffi.lib.pss_sa_build(text.ctypes.data, sa.ctypes.data, n, 0))
        built_in = time.time() - t0
        assert sorted(other.search('gamma')) == sorted(e.decode() for e in _unpack_list(other, b'gamma'))
        th.join(20)
        assert not th.is_alive() and out['rc'] == _ffi.PSS_EDEVICE and built_in < 10.0, built_in
        assert np.array_equal(np.sort(sa), np.arange(n, dtype=np.int32))
        print(f'build beside a blocked gather: {built_in * 1e3:.0f} ms (device-side wait: {peer.device_wait})')
        _ffi.lib.pss_comm_destroy(comm)
