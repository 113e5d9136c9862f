"""CPU tests of the host side: the C ABI surface, the Python mirror of the
reference API, the corpus generators and the multi-process result gather.
No compute call needs a GPU here (the engine has no CPU fallback)."""
import pathlib
import ctypes
import hashlib
import inspect
import json
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    hdr = pathlib.Path(os.path.join(ROOT, 'include', 'pss.h')).read_text()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    names = sorted(set(re.findall(r'\b(pss_[a-z0-9_]+)\s*\(', hdr)))
    assert len(names) >= 25
    lib = ctypes.CDLL(os.path.join(ROOT, 'pysubstringsearch_amd', 'libpss.so'))
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/pss.h but not exported'


def test_stats_structs_are_declared_alike_on_both_sides():
    """pss_sa_stats / pss_search_stats exist twice -- include/pss.h and the ctypes classes of _ffi.py: same fields in the same
    order with the same types, and the library reports the size its build had (the binding refuses to load otherwise)."""
    from pysubstringsearch_amd import _ffi
    hdr = pathlib.Path(os.path.join(ROOT, 'include', 'pss.h')).read_text()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    ctype = {'uint64_t': ctypes.c_uint64, 'uint32_t': ctypes.c_uint32, 'int32_t': ctypes.c_int32, 'double': ctypes.c_double,
             'float': ctypes.c_float, 'int64_t': ctypes.c_int64}
    for name, cls in (('pss_sa_stats', _ffi.SaStats), ('pss_search_stats', _ffi.SearchStats)):
        body = re.search(r'typedef struct(?:\s+\w+)?\s*\{([^}]*)\}\s*' + name + r'\s*;', hdr, flags=re.S).group(1)
        fields = []
        for decl in body.split(';'):
            decl = decl.strip()
            if not decl:
                continue
            m = re.match(r'(\w+)\s+(\w+)(?:\[(\d+)\])?$', decl)
            assert m, decl
            t = ctype[m.group(1)]
            fields.append((m.group(2), t * int(m.group(3)) if m.group(3) else t))
        have = [(k, t) for k, t in cls._fields_]
        assert [k for k, _ in fields] == [k for k, _ in have], name
        for (k, t), (_, u) in zip(fields, have):
            assert ctypes.sizeof(t) == ctypes.sizeof(u) and (t is u or getattr(t, '_length_', 0) == getattr(u, '_length_', 0)), (name, k)
    assert _ffi.lib.pss_sa_stats_size() == ctypes.sizeof(_ffi.SaStats)
    assert _ffi.lib.pss_search_stats_size() == ctypes.sizeof(_ffi.SearchStats)


def test_api_signatures_match_reference(pss):
    import pysubstringsearch
    assert pysubstringsearch.Writer is pss.Writer and pysubstringsearch.Reader is pss.Reader
    # names/order from pysubstringsearch/__init__.py:7-73 of the reference
    assert list(inspect.signature(pss.Writer.__init__).parameters)[:3] == ['self', 'index_file_path', 'max_chunk_len']
    assert inspect.signature(pss.Writer.__init__).parameters['max_chunk_len'].default is None
    assert list(inspect.signature(pss.Writer.add_entry).parameters) == ['self', 'text']
    assert list(inspect.signature(pss.Writer.add_entries_from_file_lines).parameters) == ['self', 'input_file_path']
    assert list(inspect.signature(pss.Writer.dump_data).parameters) == ['self']
    assert list(inspect.signature(pss.Writer.finalize).parameters) == ['self']
    assert list(inspect.signature(pss.Reader.__init__).parameters)[:2] == ['self', 'index_file_path']
    assert list(inspect.signature(pss.Reader.search).parameters) == ['self', 'substring']
    assert list(inspect.signature(pss.Reader.search_multiple).parameters) == ['self', 'substrings']


def test_handles_are_not_reference_cycles(pss, tmp_path):
    """`.writer` / `.reader` exist like in the reference wrapper (__init__.py:12,49) but do not make the object refer to
    itself: dropping the last reference runs __del__ at once (the reference's Drop, src/lib.rs:138-144), without
    waiting for the cyclic collector."""
    import gc
    import weakref
    gc.disable()
    try:
        p = str(tmp_path / 'c.idx')
        w = pss.Writer(p)
        assert w.writer is w
        ref = weakref.ref(w)
        del w
        assert ref() is None           # freed by reference counting alone
        assert os.path.getsize(p) == 0   # nothing was added: an empty, finalized file
    finally:
        gc.enable()
    assert isinstance(pss.Reader.reader, property) and isinstance(pss.Writer.writer, property)


def test_file_not_found(pss):
    # reference tests/test_pysubstringsearch.py:48-56
    with pytest.raises(FileNotFoundError):
        pss.Reader(index_file_path='missing_index_file_path')
    with pytest.raises(FileNotFoundError):
        pss.Writer(index_file_path='/nonexistent_dir_xyz/out.idx')


def test_writer_argument_errors(pss, tmp_path):
    p = str(tmp_path / 'w.idx')
    w = pss.Writer(p, 4)
    assert os.path.getsize(p) == 0                      # File::create truncates immediately
    with pytest.raises(ValueError, match='entry is too big'):   # src/lib.rs:92-94
        w.add_entry('abcdef')
    with pytest.raises(TypeError):
        w.add_entry(b'bytes')                            # pyo3 &str accepts only str
    with pytest.raises(UnicodeEncodeError):
        w.add_entry('\ud800')                            # lone surrogate
    with pytest.raises(FileNotFoundError):
        w.add_entries_from_file_lines(str(tmp_path / 'nope.txt'))
    with pytest.raises(OverflowError):
        pss.Writer(p, -1)
    with pytest.raises(TypeError):
        pss.Writer(123)
    w.finalize()                                         # nothing buffered: no device needed
    w.close()
    w.close()
    with pytest.raises(ValueError):
        w.add_entry('x')
    assert w.writer is w
    with pss.Writer(p) as w2:
        assert w2._h
    assert os.path.getsize(p) == 0


def test_chunk_limit_growth_rule(pss, tmp_path):
    from pysubstringsearch_amd import _ffi
    w = pss.Writer(str(tmp_path / 'g.idx'), 4)
    assert _ffi.lib.pss_writer_chunk_limit(w._h) == 4
    w.add_entry('abcd')      # len == limit passes the check, the appended '\n' grows the "Vec": 4 -> 8
    assert _ffi.lib.pss_writer_chunk_limit(w._h) == 8
    w2 = pss.Writer(str(tmp_path / 'z.idx'), 0)
    w2.add_entry('')         # Vec::with_capacity(0) -> first push -> 8
    assert _ffi.lib.pss_writer_chunk_limit(w2._h) == 8
    # drop without a GPU: the dump of a real chunk fails loudly, the file handle is still released
    with pytest.raises(RuntimeError, match='no usable HIP device') if not pss.device_count() else _noraise():
        w.close()
    w2.close()               # a one-byte chunk has the suffix array [0] by contract (libsais.c:6603-6607): no device
    assert pathlib.Path(str(tmp_path / 'z.idx')).read_bytes().hex() == '010000000a0400000000000000'


class _noraise:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def test_compute_fails_loudly_without_gpu(pss, tmp_path):
    if pss.device_count() > 0:
        pytest.skip('a GPU is present')
    from tests.util import sa_gpu
    with pytest.raises(RuntimeError, match='no usable HIP device'):
        sa_gpu(b'banana')
    idx = tmp_path / 'x.idx'
    idx.write_bytes(bytes.fromhex('010000000a0400000000000000'))
    with pytest.raises(RuntimeError, match='no usable HIP device'):
        pss.Reader(str(idx))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'pysubstringsearch_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.cpp', '.hip', '.h')):
                src = pathlib.Path(os.path.join(dirpath, f)).read_text(encoding='utf-8')
                assert 'oracle' not in src.replace('oracle restatement', ''), f'{f} mentions the oracle'


def test_every_environment_switch_is_in_the_one_registry():
    """csrc/knobs.h: every switch the engine reads goes through knob("PSS_..."), and knob() only answers for names in the
    registry -- so the registry (which the C ABI lists, and which tests/tools/fuzz.py draws from) is complete by
    construction.  The sources may not call getenv("PSS_...") directly, may not ask knob() for a name the table lacks,
    and the table lists nothing the sources never read (PSS_DEVICE is read through the launchers' loop in capi.cpp)."""
    import re
    from pysubstringsearch_amd import _ffi
    table = _ffi.knobs()
    names = [k['name'] for k in table]
    assert len(names) == len(set(names)) >= 60
    assert all(k['what'] and k['default'] for k in table)
    read = set()
    csrc = os.path.join(ROOT, 'pysubstringsearch_amd', 'csrc')
    for f in os.listdir(csrc):
        if f.endswith(('.cpp', '.hip', '.h')):
            src = pathlib.Path(os.path.join(csrc, f)).read_text(encoding='utf-8')
            assert not re.search(r'getenv\("PSS_', src), f'{f} reads a switch behind the registry'
            read |= set(re.findall(r'knob\("(PSS_[A-Z0-9_]+)"\)', src))
    assert read - set(names) == set(), f'not registered: {sorted(read - set(names))}'
    assert set(names) - read == {'PSS_DEVICE'}, f'registered but never read: {sorted(set(names) - read)}'


# ---- corpus generators: C++ (libpss) vs an independent Python restatement of SURVEY 8(d) ----

M64 = (1 << 64) - 1


class Xs:
    def __init__(self, s):
        self.s = s

    def nx(self):
        s = self.s
        s ^= (s << 13) & M64
        s ^= s >> 7
        s ^= (s << 17) & M64
        self.s = s
        return s >> 32


def py_lines(n, chunk=0):
    alpha = b'abcdefghijklmnopqrstuvwxyz0123456789 .'
    g = Xs(88172645463325252 + chunk)
    out = bytearray(n)
    for i in range(n):
        r = g.nx()
        out[i] = 10 if r % 40 == 0 else alpha[(r >> 8) % 38]
    out[-1] = 10
    return bytes(out)


def py_words(n, chunk=0):
    v = Xs(0x2545F4914F6CDD1D)
    vocab = []
    for _ in range(65536):
        ln = 3 + v.nx() % 8
        vocab.append(bytes(97 + v.nx() % 26 for _ in range(ln)))
    g = Xs(88172645463325252 + chunk)
    out = bytearray()
    while len(out) < n:
        k = 1 + g.nx() % 12
        ws = []
        for _ in range(k):
            a = g.nx() % 65536
            sh = g.nx() % 16
            ws.append(vocab[a >> sh])
        out += b' '.join(ws) + b'\n'
    out = out[:n]
    out[-1] = 10
    return bytes(out)


def _py_lines_raw(n, seed):
    alpha = b'abcdefghijklmnopqrstuvwxyz0123456789 .'
    g = Xs(seed)
    out = bytearray(n)
    for i in range(n):
        r = g.nx()
        out[i] = 10 if r % 40 == 0 else alpha[(r >> 8) % 38]
    return out


def py_repeat_line(n, chunk=0):
    line = _py_lines_raw(40, 88172645463325252 + chunk)
    line = bytes(32 if c == 10 else c for c in line[:39]) + b'\n'
    out = bytearray((line * (n // 40 + 1))[:n])
    out[-1] = 10
    return bytes(out)


def py_dup_blocks(n, chunk=0):
    alpha = b'abcdefghijklmnopqrstuvwxyz0123456789 .'
    blk = 1 << 20
    base = _py_lines_raw(min(blk, n), 88172645463325252 + chunk)
    out = bytearray(base)
    g = Xs(((88172645463325252 ^ 0xD1B54A32D192ED03) + chunk) & M64)
    o = blk
    while o < n:
        ln = min(blk, n - o)
        copy = bytearray(base[:ln])
        for _ in range(16):
            pos = g.nx() % blk
            val = alpha[g.nx() % 38]
            if pos < ln:
                copy[pos] = val
        out += copy
        o += blk
    out[-1] = 10
    return bytes(out)


def test_generators_match_python_spec():
    from tests.util import gen_corpus
    n = 20000
    assert gen_corpus(0, n, 3).tobytes() == py_lines(n, 3)
    assert gen_corpus(1, n, 2).tobytes() == py_words(n, 2)
    assert gen_corpus(4, n, 1).tobytes() == py_repeat_line(n, 1)
    m = (1 << 20) + 70000           # the block, then a copy with (some of) its 16 edits
    assert gen_corpus(5, m, 2).tobytes() == py_dup_blocks(m, 2)
    per = gen_corpus(3, 10000).tobytes()
    assert per[:4096] == b'a' * 4095 + b'\n' and per[-1:] == b'\n' and set(per) == {97, 10}
    runs = gen_corpus(2, 50000).tobytes()
    assert set(runs) <= {97, 98, 10} and runs[-1:] == b'\n'
    assert all(len(set(line)) <= 1 and len(line) <= 8192 for line in runs.split(b'\n'))
    kats = {k['name']: k for k in json.loads(pathlib.Path(os.path.join(ROOT, 'tests', 'golden', 'sa_kats.json')).read_text())['kats']}
    for kind, name in [(0, 'lines_1MiB'), (1, 'words_1MiB'), (2, 'runs_1MiB'), (3, 'periodic_1MiB')]:
        assert hashlib.sha256(gen_corpus(kind, 1 << 20).tobytes()).hexdigest() == kats[name]['text_sha256']


# ---- multi-process gather (N > 1 path), gloo, world_size 2 ----

def test_merge_query_major_unit():
    from pysubstringsearch_amd import dist as pdist
    a = pdist.pack_entries([b'ten', b'tenten', b'one']) + (np.array([2, 0, 1]),)
    b = pdist.pack_entries([b'ten', b'x']) + (np.array([1, 1, 0]),)
    out, total = pdist.merge_query_major([a, b])
    assert out == [b'ten', b'tenten', b'ten', b'x', b'one'] and total.tolist() == [3, 1, 1]
    assert pdist.chunk_owner(9, 8) == 1


def test_sharded_gather_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out = str(tmp_path / 'r0.json')
    env = dict(os.environ, PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', '_dist_worker.py'), str(r), '2', str(port), out],
                              env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=180) == 0
    got = json.loads(pathlib.Path(out).read_text())
    # query-major; inside a query rank 0's entries then rank 1's; duplicates across queries kept
    assert got['got'] == ['ten', 'tenten', 'ten', 'x', 'one', 'three', 'ten', 'tenten', 'ten']
    assert got['counts'] == [2, 3] and got['raw'] == ['one', 'three', 'ten', 'tenten', 'ten']


def test_cabi_argument_contract_without_gpu():
    """libsais' argument contract (libsais.c:6599-6607) holds before any device is touched."""
    from pysubstringsearch_amd import _ffi
    lib = _ffi.lib
    t = np.frombuffer(b'x', dtype=np.uint8).copy()
    sa = np.full(4, -7, dtype=np.int32)
    assert lib.pss_sa_build(None, sa.ctypes.data, 1, 0) == _ffi.PSS_EINVAL
    assert lib.pss_sa_build(t.ctypes.data, None, 1, 0) == _ffi.PSS_EINVAL
    assert lib.pss_sa_build(t.ctypes.data, sa.ctypes.data, -1, 0) == _ffi.PSS_EINVAL
    assert lib.pss_sa_build(t.ctypes.data, sa.ctypes.data, 0, 0) == _ffi.PSS_OK and sa[0] == -7   # n == 0 writes nothing
    assert lib.pss_sa_build(t.ctypes.data, sa.ctypes.data, 1, 0) == _ffi.PSS_OK and sa[0] == 0     # n == 1 -> SA[0] = 0
    assert _ffi.last_error() != '' or True
    h = ctypes.c_void_p()
    assert lib.pss_writer_open(None, -1, 0, ctypes.byref(h)) == _ffi.PSS_EINVAL
    assert lib.pss_reader_open(b'x', 0, 2, 2, ctypes.byref(h)) == _ffi.PSS_EINVAL      # shard index out of range
    assert lib.pss_gen_corpus(99, t.ctypes.data, 1, 0) == _ffi.PSS_EINVAL
    assert lib.pss_result_num_entries(None) == 0 and lib.pss_reader_num_chunks(None) == 0
    # the observers of round 6: a null reader has no parts, a device nobody used holds no workspace, knob indices are checked
    assert lib.pss_reader_part_chunks(None, None, 0) == 0
    assert lib.pss_workspace_bytes(-1) == 0 and lib.pss_workspace_bytes(63) == 0 and lib.pss_workspace_bytes(1 << 20) == 0
    assert lib.pss_knob_info(-1, None, None, None, None) == _ffi.PSS_EINVAL
    assert lib.pss_knob_info(lib.pss_knob_count(), None, None, None, None) == _ffi.PSS_EINVAL
    assert lib.pss_knob_info(0, None, None, None, None) == _ffi.PSS_OK
    assert lib.pss_reader_set_low_latency(None, 1) == _ffi.PSS_EINVAL
    assert lib.pss_reader_low_latency_stats(None, None, None) == _ffi.PSS_EINVAL
    assert lib.pss_reader_evict_chunk(None, 0) == _ffi.PSS_EINVAL
    lib.pss_result_free(None)
    assert lib.pss_reader_close(None) == _ffi.PSS_OK and lib.pss_writer_close(None) == _ffi.PSS_OK


def test_merge_packed_random():
    """The vectorised merge of per-rank packed results equals the naive one: query-major, inside a
    query rank-major, inside a rank the local order; empty blobs, empty entries, one rank."""
    import random
    from pysubstringsearch_amd import dist as pdist
    rng = random.Random(1)
    for _ in range(100):
        nq, world = rng.randint(1, 6), rng.randint(1, 4)
        per, naive = [], [[[] for _ in range(world)] for _ in range(nq)]
        for r in range(world):
            ents, counts = [], []
            for q in range(nq):
                k = rng.randint(0, 3)
                counts.append(k)
                for _ in range(k):
                    e = bytes(rng.randrange(256) for _ in range(rng.randint(0, 5)))
                    ents.append(e)
                    naive[q][r].append(e)
            blob, lens = pdist.pack_entries(ents)
            per.append((blob, lens, np.array(counts)))
        want = [e for q in range(nq) for r in range(world) for e in naive[q][r]]
        out, total = pdist.merge_query_major(per)
        assert out == want
        assert total.tolist() == [sum(len(naive[q][r]) for r in range(world)) for q in range(nq)]
        blob, offsets, total2 = pdist.merge_packed(per)
        assert bytes(blob) == b''.join(want) and len(offsets) == len(want) + 1 and total2.tolist() == total.tolist()


def test_tools_compile():
    """The measurement / fuzzing helpers under tests/tools are not imported by any test: at least
    keep them syntactically valid."""
    import glob
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools')
    files = sorted(glob.glob(os.path.join(root, '*.py')))
    assert len(files) > 10
    for f in files:
        compile(pathlib.Path(f).read_text(encoding='utf-8'), f, 'exec')


def test_drop_in_name_ships_type_stubs():
    """The reference ships pysubstringsearch/pysubstringsearch.pyi and py.typed beside its __init__.py."""
    pkg = os.path.join(ROOT, 'pysubstringsearch')
    assert os.path.exists(os.path.join(pkg, 'py.typed')) and os.path.exists(os.path.join(pkg, '__init__.pyi'))
    stub = pathlib.Path(os.path.join(ROOT, 'pysubstringsearch_amd', '__init__.pyi')).read_text()
    for name in ('add_entries_from_file_lines', 'add_entry', 'dump_data', 'finalize', 'search', 'search_multiple'):
        assert f'def {name}(' in stub, name


def test_merge_packed_c_helper_rejects_inconsistent_input():
    from pysubstringsearch_amd import _ffi
    counts = np.array([2, 1], dtype=np.uint64)
    starts = np.array([0, 3], dtype=np.uint64)          # 2 entries, but the counts claim 3
    blob = np.frombuffer(b'abcdef', dtype=np.uint8)
    arr = ctypes.c_void_p * 1
    ne, nb = np.array([2], dtype=np.uint64), np.array([6], dtype=np.uint64)
    oc, oo, ob = np.zeros(2, np.uint64), np.zeros(4, np.uint64), np.zeros(6, np.uint8)
    rc = _ffi.lib.pss_merge_packed(1, 2, arr(counts.ctypes.data), arr(starts.ctypes.data), arr(blob.ctypes.data),
                                   ne.ctypes.data, nb.ctypes.data, oc.ctypes.data, oo.ctypes.data, ob.ctypes.data)
    assert rc == _ffi.PSS_EINVAL


def test_io_pool_is_done_with_a_batch_before_its_waiter_returns(tmp_path):
    """A Batch of the I/O pool lives on its waiter's stack (the Writer's record thread, the Reader's load paths): the worker
    that finishes its last piece must not touch it once the waiter can run on.  tests/native/io_pool_stack_batch.cpp
    repeats the record thread's pattern 40 000 times and watches the stack frame that follows; the version that notified
    after unlocking (found by round 6's fuzz campaign as a stack-smashing abort on the GPU box) rewrites words of that
    frame or hangs here within a few thousand rounds.  Host code only: common.cpp built with g++, no device touched."""
    csrc = os.path.join(ROOT, 'pysubstringsearch_amd', 'csrc')
    exe = str(tmp_path / 'io_pool_stack_batch')
    r = subprocess.run(['g++', '-O1', '-std=c++17', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include', '-I' + csrc,
                        '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'native', 'io_pool_stack_batch.cpp'),
                        os.path.join(csrc, 'common.cpp'), '-o', exe, '-L/opt/rocm/lib', '-lamdhip64', '-lpthread'],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, LD_LIBRARY_PATH='/opt/rocm/lib:' + os.environ.get('LD_LIBRARY_PATH', ''))
    r = subprocess.run([exe, '40000'], capture_output=True, text=True, env=env, timeout=180)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-1000:]


def test_host_logic_under_sanitizers(tmp_path):
    """The host side of the library (capi.cpp: container writer / reader, argument checks, packed-result merge;
    common.cpp; corpus.cpp) rebuilt with -fsanitize=address,undefined (`make asan`) and driven through the host tests of
    this file in a child process: no report, same results.  CPU only -- GPU sanitizers are not available."""
    import shutil
    import subprocess
    if not shutil.which('g++'):
        pytest.skip('no g++')
    asan_rt = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip()
    ubsan_rt = subprocess.run(['gcc', '-print-file-name=libubsan.so'], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan_rt) or not os.path.exists(asan_rt):
        pytest.skip('no libasan')
    r = subprocess.run(['make', '-C', os.path.join(ROOT, 'pysubstringsearch_amd', 'csrc'), 'asan'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = os.path.join(ROOT, 'build', 'asan', 'libpss_asan.so')
    env = dict(os.environ, PSS_LIBPSS=lib, LD_PRELOAD=asan_rt + ':' + ubsan_rt,
               ASAN_OPTIONS='detect_leaks=0:abort_on_error=1', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    tests = ['test_cabi_argument_contract_without_gpu', 'test_writer_argument_errors', 'test_chunk_limit_growth_rule',
             'test_merge_packed_random', 'test_merge_packed_c_helper_rejects_inconsistent_input', 'test_generators_match_python_spec',
             'test_file_not_found', 'test_handles_are_not_reference_cycles']
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_host.py'), '-x', '-q', '-m', 'not gpu',
                        '-k', ' or '.join(tests)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert 'AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stderr[-3000:]
    assert f'{len(tests)} passed' in r.stdout, r.stdout[-500:]



def test_default_device_list(monkeypatch):
    """pss_default_devices: what `Reader(path)` / `Writer(path)` use when no device is named -- PSS_DEVICES=all|list
    first, else the pin of a launcher (PSS_DEVICE / LOCAL_RANK), else every visible device (the reference's search
    fans over every core without being asked: src/lib.rs:205-207)."""
    import pysubstringsearch_amd as P
    for var in ('PSS_DEVICES', 'PSS_DEVICE', 'LOCAL_RANK', 'SLURM_LOCALID', 'OMPI_COMM_WORLD_LOCAL_RANK', 'MV2_COMM_WORLD_LOCAL_RANK'):
        monkeypatch.delenv(var, raising=False)
    n = P.device_count()
    assert P.default_devices() == (list(range(n)) if n else [0])
    monkeypatch.setenv('LOCAL_RANK', '3')
    assert P.default_devices() == [3 % n if n else 3]
    monkeypatch.setenv('PSS_DEVICE', '1')
    assert P.default_devices() == [1 % n if n else 1]              # PSS_DEVICE before LOCAL_RANK
    monkeypatch.setenv('PSS_DEVICES', '0,0,0')
    assert P.default_devices() == [0, 0, 0]                         # the list wins; an ordinal may repeat
    monkeypatch.setenv('PSS_DEVICES', 'all')
    assert P.default_devices() == (list(range(n)) if n else [0])
    # a list that does not parse is an error (round 5: it used to count as unset, and under a launcher every rank would
    # have opened its handles on all visible GPUs); so is an ordinal beyond the device count
    for bad in ('0,x', '0,,1', ',', '-1', 'every'):
        monkeypatch.setenv('PSS_DEVICES', bad)
        with pytest.raises(ValueError, match='PSS_DEVICES'):
            P.default_devices()
        with pytest.raises(ValueError, match='PSS_DEVICES'):
            P.Reader('/nonexistent/never-opened.idx')
    # set to nothing counts as unset -- and the launcher's pin is honoured then (it sat in the else-branch before)
    monkeypatch.setenv('PSS_DEVICES', '')
    assert P.default_devices() == [1 % n if n else 1]
    monkeypatch.delenv('PSS_DEVICE')
    monkeypatch.delenv('LOCAL_RANK')
    monkeypatch.setenv('SLURM_LOCALID', '2')
    assert P.default_devices() == [2 % n if n else 2]
    monkeypatch.delenv('SLURM_LOCALID')
    monkeypatch.setenv('OMPI_COMM_WORLD_LOCAL_RANK', '5')
    assert P.default_devices() == [5 % n if n else 5]
    arr = (ctypes.c_int32 * 2)()
    from pysubstringsearch_amd import _ffi
    monkeypatch.setenv('PSS_DEVICES', '0,0,0')
    assert _ffi.lib.pss_default_devices(arr, 2) == 2                # never more than the caller's capacity
    assert _ffi.lib.pss_default_devices(None, 4) == 0
    monkeypatch.setenv('PSS_DEVICES', 'x')
    assert _ffi.lib.pss_default_devices(arr, 2) == -1 and 'PSS_DEVICES' in _ffi.last_error()


def test_file_ingest_reads_straight_into_the_chunk(pss, tmp_path, monkeypatch):
    """pss_writer_add_file_lines (src/lib.rs:67-86): a file of plain '\\n' lines is read straight into the chunk being
    filled, block after block -- the unterminated tail of one read stays in place and the next read continues it (round 4
    carried it over in a side buffer, which sent every later block through the copying path).  Lines with a '\\r' go
    line by line through a block buffer.  No device is needed while nothing is flushed."""
    import numpy as np
    rng = np.random.default_rng(1)
    lines = [bytes(rng.integers(97, 123, int(rng.integers(0, 90)), dtype=np.uint8)) for _ in range(40000)]
    plain = b'\n'.join(lines) + b'\nlast line without a newline'
    src = tmp_path / 'plain.txt'
    src.write_bytes(plain)
    monkeypatch.setenv('PSS_INGEST_BLOCK', '65536')            # many direct reads instead of one of 32 MiB
    monkeypatch.setenv('PSS_INGEST_MIN_ROOM', '4096')
    w = pss.Writer(str(tmp_path / 'a.idx'), 64 << 20, device=0)
    w.add_entries_from_file_lines(str(src))
    st = w.io_stats
    assert st['ingest_copied_bytes'] == 0, st
    assert st['ingest_direct_bytes'] == len(plain) - len(b'last line without a newline'), st
    try:
        w.close()                                               # (the flush needs a GPU: the container of the CPU suite has none)
    except (RuntimeError, OSError, ValueError):
        pass
    crlf = tmp_path / 'crlf.txt'
    crlf.write_bytes(b'\r\n'.join(lines[:5000]) + b'\r\n')
    w = pss.Writer(str(tmp_path / 'b.idx'), 64 << 20, device=0)
    w.add_entries_from_file_lines(str(crlf))
    st = w.io_stats
    assert st['ingest_direct_bytes'] == 0 and st['ingest_copied_bytes'] == crlf.stat().st_size, st
    try:
        w.close()
    except (RuntimeError, OSError, ValueError):
        pass
