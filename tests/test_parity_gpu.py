"""GPU parity (through the C ABI / the Python mirror of the reference API):
the reference's own tests re-expressed as data, the container goldens, and
fuzzing against the CPU oracle.  Result lists are compared as multisets
(assertCountEqual in the reference, tests/test_pysubstringsearch.py:32-37);
.idx files byte for byte."""
import pathlib
import hashlib
import json
import os
import random

import numpy as np
import pytest

import pysubstringsearch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, 'golden')


def load(name):
    return json.loads(pathlib.Path(os.path.join(GOLD, name)).read_text(encoding='utf-8'))


def build(path, entries, max_chunk_len=None, dump_after=(), W=None):
    w = (W or pysubstringsearch.Writer)(path, max_chunk_len)
    for i, e in enumerate(entries):
        w.add_entry(e)
        if i in dump_after:
            w.dump_data()
    w.finalize()
    w.close()
    return pathlib.Path(path).read_bytes()


def test_reference_tests_as_data(tmp_path):
    for case in load('reference_cases.json')['cases']:
        if 'missing_path' in case:
            with pytest.raises(FileNotFoundError):
                pysubstringsearch.Reader(index_file_path=case['missing_path'])
            continue
        p = str(tmp_path / (case['name'] + '.idx'))
        writer = pysubstringsearch.Writer(index_file_path=p)
        for s in case['entries']:
            writer.add_entry(text=s)
        writer.finalize()
        idx = pathlib.Path(p).read_bytes()
        assert hashlib.sha256(idx).hexdigest() == case['idx_sha256'], case['name']
        reader = pysubstringsearch.Reader(index_file_path=p)
        for s in case['searches']:
            assert sorted(reader.search(substring=s['substring'])) == sorted(s['expected']), (case['name'], s)
        for s in case['search_multiple']:
            assert sorted(reader.search_multiple(substrings=s['substrings'])) == sorted(s['expected'])
        reader.close()
        writer.close()


def test_container_goldens(tmp_path):
    gold = load('container_cases.json')
    for case in gold['cases']:
        p = str(tmp_path / (case['name'] + '.idx'))
        idx = build(p, case['entries'], case['max_chunk_len'], case['dump_after'])
        assert idx.hex() == case['idx_hex'], case['name']
        with pysubstringsearch.Reader(p) as r:
            for s in case['searches']:
                assert sorted(r.search(s['substring'])) == s['expected'], (case['name'], s['substring'])
    for case in gold['file_ingest']:
        src = tmp_path / (case['name'] + '.txt')
        src.write_bytes(bytes.fromhex(case['input_hex']))
        p = str(tmp_path / (case['name'] + '.idx'))
        w = pysubstringsearch.Writer(p, case.get('max_chunk_len'))
        w.add_entries_from_file_lines(str(src))
        w.close()
        assert pathlib.Path(p).read_bytes().hex() == case['idx_hex'], case['name']


def test_writer_reuse_after_finalize_and_truncation(tmp_path, oracle):
    p = str(tmp_path / 'r.idx')
    w = pysubstringsearch.Writer(p)
    w.add_entry('alpha')
    w.finalize()
    w.finalize()                      # idempotent (src/lib.rs:126-135)
    w.add_entry('beta')               # a Writer stays usable: appends one more chunk
    w.close()
    with pysubstringsearch.Reader(p) as r:
        assert r.num_chunks == 2 and sorted(r.search('a')) == ['alpha', 'beta']
    o = oracle.OracleReader(p)
    assert sorted(o.search('a')) == ['alpha', 'beta']
    data = pathlib.Path(p).read_bytes()
    t = tmp_path / 't.idx'
    t.write_bytes(data[:-3])          # truncated SA -> UnexpectedEof -> OSError
    with pytest.raises(OSError):
        pysubstringsearch.Reader(str(t))
    e = tmp_path / 'e.idx'
    e.write_bytes(b'')                # empty file: zero chunks, every search is empty
    with pysubstringsearch.Reader(str(e)) as r:
        assert r.search('a') == [] and r.search_multiple(['a', '']) == []


def _fuzz_corpus(rng, alphabet, n_entries, max_len):
    return [''.join(rng.choice(alphabet) for _ in range(rng.randint(0, max_len))) for _ in range(n_entries)]


@pytest.mark.parametrize('small_path', [True, False])
@pytest.mark.parametrize('seed', range(6))
def test_fuzz_against_oracle(tmp_path, oracle, seed, small_path, search_env):
    # both search paths: the fused small-batch kernel and the general multi-kernel pipeline
    if not small_path:
        search_env(PSS_NO_SMALL_PATH=1)
    rng = random.Random(seed)
    alphabet = rng.choice(['ab', 'abc', 'ab \n'.replace('\n', ''), 'abcdefgh', 'aé☃', '\x00a'])
    entries = _fuzz_corpus(rng, alphabet, rng.randint(1, 400), rng.choice([3, 12, 60]))
    limit = rng.choice([None, 64, 257, 5000])
    if limit is not None:
        limit = max(limit, max(len(e.encode()) for e in entries) + 1)
    p = str(tmp_path / 'f.idx')
    q = str(tmp_path / 'o.idx')
    idx = build(p, entries, limit)
    oracle.use_reference_sa(False)
    ref = build(q, entries, limit, W=oracle.OracleWriter)
    assert idx == ref                                   # byte-identical container
    text = '\n'.join(entries) + '\n'
    queries = ['', 'a', 'b', 'ab', 'ba', 'zz', '\n', 'a\n', '\na']
    for _ in range(60):
        s = rng.randrange(len(text))
        queries.append(text[s:s + rng.randint(1, 9)])
    for _ in range(20):
        queries.append(''.join(rng.choice(alphabet) for _ in range(rng.randint(1, 6))))
    o = oracle.OracleReader(q)
    with pysubstringsearch.Reader(p) as r:
        for s in queries:
            assert sorted(r.search(s)) == sorted(o.search(s)), repr(s)
        got = r.search_multiple(queries)                # one batched device call
        exp = o.search_multiple(queries)
        assert sorted(got) == sorted(exp)
        ents, counts = r.search_batch_raw([s.encode() for s in queries])
        oe, oc = o.search_multiple_bytes([s.encode() for s in queries])
        assert counts == oc.tolist()                    # per-query counts, query-major order
        pos = 0
        for c in counts:                                # per-query multisets line up query by query
            assert sorted(ents[pos:pos + c]) == sorted(oe[pos:pos + c])
            pos += c


def test_long_entries_and_patterns(tmp_path, oracle):
    rng = random.Random(99)
    entries = ['a' * 5000, 'a' * 4999 + 'b', ('ab' * 3000), 'x' * 70 + 'needle' + 'y' * 70,
               ''.join(rng.choice('abc') for _ in range(20000))]
    p = str(tmp_path / 'l.idx')
    build(p, entries)
    o = oracle.OracleReader(p)
    qs = ['a' * 100, 'a' * 4999 + 'b', 'ab' * 40, 'needle', 'x' * 65, 'a' * 5001, 'ba' * 33 + 'b', entries[4][777:777 + 300]]
    with pysubstringsearch.Reader(p) as r:
        for s in qs:
            assert sorted(r.search(s)) == sorted(o.search(s)), s[:20]
        assert sorted(r.search_multiple(qs)) == sorted(o.search_multiple(qs))


def test_one_mib_chunks_batch(tmp_path, oracle):
    """3 chunks x 1 MiB of the `lines` corpus via file ingest; 2000 mixed queries in one batch."""
    from tests.util import gen_corpus
    src = tmp_path / 'corpus.txt'
    with open(src, 'wb') as f:
        for c in range(3):
            f.write(gen_corpus(0, 1 << 20, c).tobytes())
    p = str(tmp_path / 'c.idx')
    w = pysubstringsearch.Writer(p, 1 << 20)
    w.add_entries_from_file_lines(str(src))
    w.close()
    q = str(tmp_path / 'o.idx')
    oracle.use_reference_sa(oracle.have_reference())
    ow = oracle.OracleWriter(q, 1 << 20)
    ow.add_entries_from_file_lines(str(src))
    ow.close()
    oracle.use_reference_sa(False)
    assert hashlib.sha256(pathlib.Path(p).read_bytes()).hexdigest() == hashlib.sha256(pathlib.Path(q).read_bytes()).hexdigest()
    text = pathlib.Path(src).read_bytes()
    rng = np.random.default_rng(1)
    qs = []
    while len(qs) < 1000:
        s = int(rng.integers(0, len(text) - 40))
        ln = int(rng.integers(2, 12))
        cand = text[s:s + ln]
        if b'\n' not in cand:
            qs.append(cand)
    alpha = b'abcdefghijklmnopqrstuvwxyz0123456789 .'
    for _ in range(1000):
        qs.append(bytes(alpha[int(i)] for i in rng.integers(0, 38, int(rng.integers(4, 9)))))
    o = oracle.OracleReader(q)
    with pysubstringsearch.Reader(p) as r:
        assert r.num_chunks == 3
        ents, counts = r.search_batch_raw(qs)
        oe, oc = o.search_multiple_bytes(qs)
        assert counts == oc.tolist()
        assert sorted(ents) == sorted(oe)
        st = r.last_stats()
        assert st['queries'] == len(qs) and st['entries'] == len(ents)
    # sharded readers: each owns chunk c % 2, union == whole
    with pysubstringsearch.Reader(p, shard=(0, 2)) as r0, pysubstringsearch.Reader(p, shard=(1, 2)) as r1:
        assert (r0.num_chunks, r1.num_chunks) == (2, 1)
        e0, c0 = r0.search_batch_raw(qs[:200])
        e1, c1 = r1.search_batch_raw(qs[:200])
        from pysubstringsearch_amd import dist as pdist
        merged, total = pdist.merge_query_major([pdist.pack_entries(e0) + (np.array(c0),),
                                                 pdist.pack_entries(e1) + (np.array(c1),)])
        oe2, oc2 = o.search_multiple_bytes(qs[:200])
        assert total.tolist() == oc2.tolist() and sorted(merged) == sorted(oe2)


def test_sa_kats_gpu():
    from tests.test_oracle import _kat_input
    from tests.util import sa_gpu
    for k in load('sa_kats.json')['kats']:
        data = _kat_input(k, None)
        sa = sa_gpu(data)
        assert hashlib.sha256(sa.tobytes()).hexdigest() == k['sa_sha256'], k['name']
        if 'sa' in k:
            assert sa.tolist() == k['sa']


def test_sa_properties_at_scale():
    """64 MiB (size-independent properties + the pinned libsais hash of SURVEY 8(c)(6))."""
    import torch
    from pysubstringsearch_amd import _ffi
    from tests.util import gen_corpus
    gold = load('sa_kats.json')['generated_big']
    for kind, name in [(0, 'lines_64MiB'), (1, 'words_64MiB')]:
        n = 1 << 26
        host = gen_corpus(kind, n)
        assert hashlib.sha256(host.tobytes()).hexdigest() == gold[name]['text_sha256']
        dT = torch.from_numpy(host).cuda()
        dSA = torch.empty(n, dtype=torch.int32, device='cuda')
        _ffi.check(_ffi.lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, 0, None))
        sa = dSA.cpu().numpy()
        assert hashlib.sha256(sa.tobytes()).hexdigest() == gold[name]['sa_sha256']
        # permutation + sortedness on a strided sample of adjacent pairs
        assert np.array_equal(np.sort(sa[:: 1]), np.arange(n, dtype=np.int32)) if n <= (1 << 26) else True
        text = host.tobytes()
        for j in range(1, n, n // 20000):
            a, b = int(sa[j - 1]), int(sa[j])
            assert text[a:a + 64] <= text[b:b + 64]


@pytest.mark.parametrize('seed', range(8))
def test_file_ingest_fuzz(tmp_path, oracle, seed, monkeypatch):
    """add_entries_from_file_lines (direct reads into the chunk with the unterminated tail kept in place, bulk fast path,
    per-line path) vs the oracle's line-by-line restatement: byte-identical .idx for random files and chunk limits.
    Seeds 5 .. 7 shrink the direct reads to a few dozen bytes (PSS_INGEST_BLOCK / _MIN_ROOM), so that path -- which
    otherwise wants a megabyte of room -- runs against limits of a few hundred bytes, tails longer than a block included."""
    rng = random.Random(100 + seed)
    if seed >= 5:
        monkeypatch.setenv('PSS_INGEST_BLOCK', str(rng.choice([16, 50, 700])))
        monkeypatch.setenv('PSS_INGEST_MIN_ROOM', str(rng.choice([1, 16, 100])))
    parts = []
    for _ in range(rng.randint(1, 3000)):
        ln = rng.choice([0, 1, 2, 5, 17, 40, 80, 300])
        body = bytes(rng.choice(b'abcxyz \t') for _ in range(rng.randint(0, ln)))
        term = rng.choices([b'\n', b'\r\n', b'\r'], weights=[90, 7 if seed % 2 else 0, 3 if seed % 2 else 0])[0]
        parts.append(body + term)
    raw = b''.join(parts)
    if rng.random() < 0.5:
        raw = raw.rstrip(b'\n') + b'tail-without-newline'
    src = tmp_path / 'in.txt'
    src.write_bytes(raw)
    for limit in (None, rng.choice([64, 100, 257]), rng.choice([1000, 4096, 30000]), 7):
        p, q = str(tmp_path / 'g.idx'), str(tmp_path / 'o.idx')
        w = pysubstringsearch.Writer(p, limit)
        w.add_entry('a')
        w.add_entries_from_file_lines(str(src))
        w.add_entry('z')
        w.close()
        oracle.use_reference_sa(False)
        ow = oracle.OracleWriter(q, limit)
        ow.add_entry('a')
        ow.add_entries_from_file_lines(str(src))
        ow.add_entry('z')
        ow.close()
        assert pathlib.Path(p).read_bytes() == pathlib.Path(q).read_bytes(), (seed, limit)


@pytest.mark.parametrize('stripes', [None, '3'])
def test_striped_container(tmp_path, oracle, monkeypatch, stripes):
    """Format 2 with the suffix arrays striped over files of their own (Writer(..., format_version=2, striped=True);
    include/pss.h, PSS_FORMAT_STRIPED): the index file holds header and texts, `<path>.sa<j>` the arrays -- every chunk's
    array starts a new 16 MiB unit, unit u in file u mod S.  Same chunks, same arrays (byte for byte against the
    reference container of the same entries), same search results; a missing or short stripe file is an error like the
    reference's UnexpectedEof; one, two and several devices write the same files."""
    if stripes:
        monkeypatch.setenv('PSS_STRIPES', stripes)
    S = int(stripes) if stripes else 8
    rng = random.Random(21)
    entries = [''.join(rng.choice('abcdefg \t') for _ in range(rng.randint(0, 50))) for _ in range(9000)] + ['', 'x']
    p1, p2 = str(tmp_path / 'v1.idx'), str(tmp_path / 'striped.idx')
    b1 = build(p1, entries, 20000)
    for devices in (None, [0, 0, 0]):
        w = pysubstringsearch.Writer(p2, 20000, format_version=2, striped=True, **({'devices': devices} if devices else {}))
        for e in entries:
            w.add_entry(e)
        w.close()
        b2 = pathlib.Path(p2).read_bytes()
        assert b2[:8] == b'PSSIDX\x02\x00' and int.from_bytes(b2[8:12], 'little') == (1 | (S << 8) | (24 << 16))
        files = [pathlib.Path(f'{p2}.sa{j}').read_bytes() for j in range(S)]
        assert not os.path.exists(f'{p2}.sa{S}')
        unit = 1 << 24
        o1, o2, chunks, u = 0, 16, 0, 0
        while o1 < len(b1):
            n = int.from_bytes(b1[o1:o1 + 4], 'little')
            assert int.from_bytes(b2[o2:o2 + 8], 'little') == n and b1[o1 + 4:o1 + 4 + n] == b2[o2 + 8:o2 + 8 + n]
            assert int.from_bytes(b2[o2 + 8 + n:o2 + 16 + n], 'little') == 4 * n
            sa1 = b1[o1 + 8 + n:o1 + 8 + 5 * n]
            f, off = files[u % S], (u // S) * unit                 # (chunks this small take one unit each)
            assert f[off:off + 4 * n] == sa1, chunks
            o1, o2, chunks, u = o1 + 8 + 5 * n, o2 + 16 + n, chunks + 1, u + 1
        assert o2 == len(b2) and chunks > 5
    text = '\n'.join(entries) + '\n'
    qs = [text[s:s + rng.randint(1, 6)] for s in (rng.randrange(len(text)) for _ in range(300))] + ['', 'zz', '\n']
    o = oracle.OracleReader(p1)
    oe, oc = o.search_multiple_bytes([q.encode() for q in qs])
    for kw in ({}, {'devices': [0, 0]}, {'shard': (1, 3)}):
        with pysubstringsearch.Reader(p2, **kw) as r:
            if 'shard' in kw:
                assert r.num_chunks == len(range(1, chunks, 3))
                continue
            assert r.num_chunks == chunks
            ents, counts = r.search_batch_raw([q.encode() for q in qs])
            assert counts == oc.tolist() and sorted(ents) == sorted(oe)
    os.truncate(f'{p2}.sa1', max(0, os.path.getsize(f'{p2}.sa1') - 3))
    with pytest.raises(OSError):
        pysubstringsearch.Reader(p2)
    os.remove(f'{p2}.sa0')
    with pytest.raises(FileNotFoundError):
        pysubstringsearch.Reader(p2)
    with pytest.raises(ValueError):
        pysubstringsearch.Writer(str(tmp_path / 'bad.idx'), striped=True)          # the reference container has no flag


def test_records_through_a_shared_mapping(oracle, monkeypatch):
    """On tmpfs large records go into the index file through a shared mapping of their range (write_record, capi_writer_impl.h).
    Round 4 opened the file write-only, like File::create does (src/lib.rs:55) -- and mmap(PROT_READ | PROT_WRITE,
    MAP_SHARED) on a write-only descriptor fails with EACCES: the route never ran.  The mapping now has a descriptor of
    its own (O_RDWR); pss_writer_io_stats says which way the records went, and the bytes are the pwrite route's."""
    import shutil
    import tempfile
    if not os.path.isdir('/dev/shm') or not os.access('/dev/shm', os.W_OK):
        pytest.skip('no tmpfs at /dev/shm')
    d = tempfile.mkdtemp(prefix='pss_mmap_', dir='/dev/shm')
    try:
        rng = random.Random(3)
        entries = [''.join(rng.choice('abcdefgh ') for _ in range(rng.randint(0, 60))) for _ in range(30000)]
        files = {}
        for route, env in (('map', {'PSS_WRITER_MMAP_MIN': '4096'}), ('pwrite', {'PSS_WRITER_MMAP': '0'})):
            for k in ('PSS_WRITER_MMAP_MIN', 'PSS_WRITER_MMAP'):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            p = os.path.join(d, route + '.idx')
            w = pysubstringsearch.Writer(p, 100000)
            for e in entries:
                w.add_entry(e)
            w.finalize()
            st = w.io_stats
            w.close()
            files[route] = pathlib.Path(p).read_bytes()
            if route == 'map':
                assert st['records_mapped'] >= 5 and st['records_pwritten'] <= 1, st      # (the last record may be below the threshold)
            else:
                assert st['records_mapped'] == 0 and st['records_pwritten'] >= 5, st
        assert files['map'] == files['pwrite']
        q = os.path.join(d, 'oracle.idx')
        ow = oracle.OracleWriter(q, 100000)
        for e in entries:
            ow.add_entry(e)
        ow.close()
        assert files['map'] == pathlib.Path(q).read_bytes()
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_real_files_through_the_file_api(tmp_path, oracle):
    """Real files (tests/tools/real_text.py: Python sources and headers found on the machine, bytes above 127, CRLF files,
    tabs, empty lines) through Writer.add_entries_from_file_lines -> .idx -> Reader: the container is byte-identical with
    the oracle's, and substrings sampled from the files -- and some that are not there -- return the oracle's entries."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('real_text', os.path.join(os.path.dirname(__file__), 'tools', 'real_text.py'))
    rt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rt)
    raw = rt.collect(5 << 20)
    if len(raw) < (2 << 20):
        pytest.skip('not enough text files on this machine')
    src = tmp_path / 'real.txt'
    src.write_bytes(raw)
    p, q = str(tmp_path / 'g.idx'), str(tmp_path / 'o.idx')
    limit = 1 << 20
    w = pysubstringsearch.Writer(p, limit)
    w.add_entries_from_file_lines(str(src))
    w.close()
    oracle.use_reference_sa(oracle.have_reference())
    ow = oracle.OracleWriter(q, limit)
    ow.add_entries_from_file_lines(str(src))
    ow.close()
    oracle.use_reference_sa(False)
    assert pathlib.Path(p).read_bytes() == pathlib.Path(q).read_bytes()
    rng = np.random.default_rng(3)
    qs = []
    for s0, k in zip(rng.integers(0, len(raw) - 64, 150), rng.integers(2, 40, 150)):
        cand = raw[int(s0):int(s0) + int(k)]
        if b'\n' not in cand and b'\r' not in cand:
            qs.append(cand)
    qs += [b'import ', b'    return', b'zq#zq#zq', b'\xc3\xa9', b'Copyright']
    o = oracle.OracleReader(p)
    oe, oc = o.search_multiple_bytes(qs)
    with pysubstringsearch.Reader(p) as r:
        ents, counts = r.search_batch_raw(qs)
        assert counts == [int(c) for c in oc]
        a = b = 0
        for cg, ce in zip(counts, oc):
            assert sorted(ents[a:a + cg]) == sorted(oe[b:b + int(ce)])
            a += cg
            b += int(ce)
        for i in (0, 5, len(qs) - 5, len(qs) - 1):
            one, c1 = r.search_batch_raw([qs[i]])
            assert c1 == [counts[i]]


def test_packed_result_api(tmp_path, oracle):
    entries = ['alpha beta', 'beta gamma', 'gamma', 'alphabet', '']
    p = str(tmp_path / 'p.idx')
    build(p, entries)
    o = oracle.OracleReader(p)
    qs = [b'alpha', b'gamma', b'zzz', b'', b'a']
    with pysubstringsearch.Reader(p) as r:
        pk = r.search_batch_packed(qs)
        ents, counts = r.search_batch_raw(qs)
        assert pk.counts.tolist() == counts
        blob = pk.data.tobytes()
        got = [blob[int(pk.offsets[i]):int(pk.offsets[i + 1])] for i in range(len(pk.offsets) - 1)]
        assert got == ents
        oe, oc = o.search_multiple_bytes(qs)
        assert counts == oc.tolist() and sorted(got) == sorted(oe)
        empty = r.search_batch_packed([])
        assert empty.counts.size == 0 and empty.offsets.tolist() == [0]


def test_concurrent_handles_from_threads(tmp_path, oracle):
    """ctypes drops the GIL: several Python threads inside the engine at once (shared
    per-device workspace) must still give oracle results."""
    import threading
    import pysubstringsearch_amd
    rng = random.Random(5)
    corpora = []
    for k in range(3):
        entries = [''.join(rng.choice('abcd ') for _ in range(rng.randint(1, 30))) for _ in range(400)]
        p = str(tmp_path / f't{k}.idx')
        build(p, entries, 2000)
        corpora.append((p, entries))
    errors = []

    def worker(k):
        try:
            p, entries = corpora[k]
            o = oracle.OracleReader(p)
            qs = ['a', 'ab', 'abc', 'd d', ' ', 'zz'] + [e[:3] for e in entries[:40]]
            with pysubstringsearch.Reader(p) as r:
                for _ in range(15):
                    for s in qs[:12]:
                        if sorted(r.search(s)) != sorted(o.search(s)):
                            errors.append((k, s))
                    if sorted(r.search_multiple(qs)) != sorted(o.search_multiple(qs)):
                        errors.append((k, 'batch'))
            q = str(tmp_path / f'w{k}.idx')      # and a Writer in the same thread
            build(q, entries, 1500)
            oracle.use_reference_sa(False)
            assert pathlib.Path(q).read_bytes() == build(str(tmp_path / f'ow{k}.idx'), entries, 1500, W=oracle.OracleWriter)
        except Exception as e:   # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
    pysubstringsearch_amd.release_workspace()
    with pysubstringsearch.Reader(corpora[0][0]) as r:      # still works after the workspace was dropped
        assert sorted(r.search('a')) == sorted(oracle.OracleReader(corpora[0][0]).search('a'))


def test_large_batch_lane_search(tmp_path, oracle):
    """>= 8192 (query, chunk) pairs switch the interval search to one lane per pair."""
    from tests.util import gen_corpus
    src = tmp_path / 'c.txt'
    src.write_bytes(gen_corpus(1, 1 << 19).tobytes())
    p = str(tmp_path / 'c.idx')
    w = pysubstringsearch.Writer(p, 1 << 17)
    w.add_entries_from_file_lines(str(src))
    w.close()
    text = src.read_bytes()
    rng = np.random.default_rng(9)
    qs = [b'', b'a', b'\n', b'zzzzzz']
    while len(qs) < 9000:
        s = int(rng.integers(0, len(text) - 20))
        qs.append(text[s:s + int(rng.integers(1, 14))])
    o = oracle.OracleReader(p)
    with pysubstringsearch.Reader(p) as r:
        assert r.num_chunks * len(qs) >= 32768
        ents, counts = r.search_batch_raw(qs)
        oe, oc = o.search_multiple_bytes(qs)
        assert counts == oc.tolist()
        assert sorted(ents) == sorted(oe)


def test_count_only_matches_search(tmp_path, oracle):
    """Extension API: count / count_multiple = len(search) per query, on a multi-chunk index, for
    batches on both sides of the lane-search threshold; counts come from the oracle."""
    rng = random.Random(4)
    entries = [''.join(rng.choice('abc') for _ in range(rng.randint(0, 12))) for _ in range(3000)]
    p, q = str(tmp_path / 'c.idx'), str(tmp_path / 'o.idx')
    build(p, entries, 2000)
    oracle.use_reference_sa(False)
    build(q, entries, 2000, W=oracle.OracleWriter)
    o = oracle.OracleReader(q)
    text = '\n'.join(entries) + '\n'
    with pysubstringsearch.Reader(p) as r:
        assert r.count_multiple([]) == []
        assert r.count('zzz') == 0 and r.count('') == len(entries)
        for nq in (7, 4000):
            qs = [text[s:s + rng.randint(1, 5)] for s in (rng.randrange(len(text)) for _ in range(nq))] + ['', 'a', 'q']
            _, oc = o.search_multiple_bytes([s.encode() for s in qs])
            assert r.count_multiple(qs) == oc.tolist()
        assert r.count('ab') == len(r.search('ab'))
        with pytest.raises(TypeError):
            r.count(b'ab')


def test_packed_result_outlives_reader(tmp_path):
    """search_batch_packed hands out views of the C result: they stay valid after the reader is
    closed and collected, are read-only, and empty results are well-formed."""
    import gc
    p = str(tmp_path / 'p.idx')
    build(p, ['alpha', 'beta', 'alphabet'], None)
    r = pysubstringsearch.Reader(p)
    pk = r.search_batch_packed([b'alpha', b'zzz', b'bet'])
    none = r.search_batch_packed([b'zzz'])
    r.close()
    del r
    gc.collect()
    assert pk.counts.tolist() == [2, 0, 2]
    ents = [bytes(pk.data[pk.offsets[i]:pk.offsets[i + 1]]) for i in range(len(pk.offsets) - 1)]
    assert sorted(ents) == sorted([b'alpha', b'alphabet', b'beta', b'alphabet'])
    assert not pk.data.flags.writeable
    with pytest.raises(ValueError):
        pk.data[0] = 0
    assert none.counts.tolist() == [0] and none.data.size == 0 and none.offsets.tolist() == [0]
    data = pk.data
    del pk
    gc.collect()
    assert bytes(data[:5]) in (b'alpha', b'beta\x00'[:5], b'alpha')   # still backed by the result


@pytest.mark.parametrize('budget_chunks', [0, 2])
def test_suffix_arrays_beyond_the_hbm_budget_stay_on_the_host(tmp_path, oracle, monkeypatch, budget_chunks):
    """Residency tiers: with PSS_READER_HBM_BUDGET exhausted the suffix arrays of the later chunks
    live in pinned host memory (read over PCIe by the kernels); every search path (fused single
    query, wave-per-pair, lane-per-pair, count-only) must return what the oracle returns."""
    from tests.util import gen_corpus
    src = tmp_path / 'c.txt'
    src.write_bytes(gen_corpus(0, 1 << 18).tobytes())
    p = str(tmp_path / 'c.idx')
    w = pysubstringsearch.Writer(p, 1 << 16)
    w.add_entries_from_file_lines(str(src))
    w.close()
    text = src.read_bytes()
    rng = np.random.default_rng(12)
    qs = [b'', b'e', b'\n', b'zzzzzz', b'a\n']
    while len(qs) < 9000:
        s = int(rng.integers(0, len(text) - 20))
        qs.append(text[s:s + int(rng.integers(1, 14))])
    o = oracle.OracleReader(p)
    # one chunk: 64 KiB text + 256 KiB suffix array + samples
    monkeypatch.setenv('PSS_READER_HBM_BUDGET', str(budget_chunks * ((1 << 16) * 5 + 4096)))
    with pysubstringsearch.Reader(p) as r:
        res = r.residency
        assert r.num_chunks >= 4
        assert res['host_chunks'] == r.num_chunks - budget_chunks and res['host_bytes'] > 0
        # fused paths (<= 64, <= 1024 pairs), 16 lanes per pair (2048 .. 8191 pairs), one lane per pair
        for batch in (qs[:1], qs[5:6], qs[:40], qs[:1000], qs):
            ents, counts = r.search_batch_raw(batch)
            oe, oc = o.search_multiple_bytes(batch)
            assert counts == oc.tolist()
            assert sorted(ents) == sorted(oe)
        assert r.count_multiple([q.decode() for q in qs[:50]]) == o.search_multiple_bytes(qs[:50])[1].tolist()
    monkeypatch.delenv('PSS_READER_HBM_BUDGET')
    with pysubstringsearch.Reader(p) as r:
        assert r.residency['host_chunks'] == 0


def test_multi_device_reader_in_one_process(tmp_path, oracle):
    """Reader(path, devices=[...]): chunk c resident on devices[c % G], one host thread per device answers the batch,
    the results are merged on the host -- no launcher, no torch.  On the one GPU of the test box the devices are
    "virtual" ([0, 0, 0]: three parts taking turns on GPU 0).  Every API of the reader must return what the oracle and
    the single-device reader return, per query, whatever G is; the part-major order inside a query is the only
    difference (the reference's inter-chunk order is unspecified, src/lib.rs:280-284)."""
    from tests.util import gen_corpus
    src = tmp_path / 'c.txt'
    src.write_bytes(gen_corpus(0, 1 << 19).tobytes())
    p = str(tmp_path / 'c.idx')
    w = pysubstringsearch.Writer(p, 1 << 16)
    w.add_entries_from_file_lines(str(src))
    w.close()
    text = src.read_bytes()
    rng = np.random.default_rng(21)
    qs = [b'', b'e', b'\n', b'zzzzzz', b'a\n', b'th']
    while len(qs) < 3000:
        s = int(rng.integers(0, len(text) - 20))
        qs.append(text[s:s + int(rng.integers(1, 14))])
    o = oracle.OracleReader(p)
    with pysubstringsearch.Reader(p) as single:
        nchunks = single.num_chunks
        for devs in ([0, 0], [0, 0, 0], [0] * 5):
            with pysubstringsearch.Reader(p, devices=devs) as r:
                assert r.num_chunks == nchunks and r.residency['hbm_bytes'] == single.residency['hbm_bytes']
                for batch in (qs[:1], qs[5:6], qs[:50], qs):
                    ents, counts = r.search_batch_raw(batch)
                    oe, oc = o.search_multiple_bytes(batch)
                    assert counts == oc.tolist()
                    pos = 0
                    se, _ = single.search_batch_raw(batch)
                    for c in counts:      # per query: the same multiset as the oracle and the single-device reader
                        assert sorted(ents[pos:pos + c]) == sorted(oe[pos:pos + c]) == sorted(se[pos:pos + c])
                        pos += c
                assert r.count_multiple([q.decode() for q in qs[:60]]) == o.search_multiple_bytes(qs[:60])[1].tolist()
                assert sorted(r.search('th')) == sorted(o.search('th'))
                pk = r.search_batch_packed(qs[:200])
                assert int(pk.counts.sum()) == len(pk.offsets) - 1 == sum(o.search_multiple_bytes(qs[:200])[1].tolist())
                st = r.last_stats()
                assert st['queries'] == 200 and st['entries'] == len(pk.offsets) - 1
                with pytest.raises(ValueError):
                    r.search_batch_device(qs[:3])
    with pytest.raises(ValueError):
        pysubstringsearch.Reader(p, devices=[])
    with pytest.raises(ValueError):
        pysubstringsearch.Reader(p, devices=[0], device=0)


def test_eight_way_placement_of_fifteen_chunks(tmp_path, oracle):
    """BASELINE configs[2] / [3] shape on the one GPU of the box: a 15-chunk index opened over EIGHT parts ([0] * 8) -- chunk
    c on part c mod 8, hence 2,2,2,2,2,2,2,1 chunks (best case 7.5 x at 8 GPUs) -- with every API answering
    what the oracle answers.  No multi-GPU box has run this build; the first 8-GPU run shall not also be the first 8-way run."""
    from tests.util import gen_corpus
    src = tmp_path / 'c.txt'
    text = gen_corpus(0, 15 << 15).tobytes()
    src.write_bytes(text)
    p = str(tmp_path / 'c.idx')
    w = pysubstringsearch.Writer(p, 1 << 15)
    w.add_entries_from_file_lines(str(src))
    w.close()
    o = oracle.OracleReader(p)
    rng = np.random.default_rng(8)
    qs = [b'', b'e', b'zzzzzz', b'a\n']
    while len(qs) < 1500:
        s0 = int(rng.integers(0, len(text) - 20))
        qs.append(text[s0:s0 + int(rng.integers(1, 14))])
    with pysubstringsearch.Reader(p, devices=[0] * 8) as r:
        nc = r.num_chunks
        assert nc >= 15
        assert r.chunks_per_device == [len(range(g, nc, 8)) for g in range(8)]
        if nc == 15:
            assert r.chunks_per_device == [2, 2, 2, 2, 2, 2, 2, 1]
        for batch in (qs[:1], qs[:40], qs):
            ents, counts = r.search_batch_raw(batch)
            oe, oc = o.search_multiple_bytes(batch)
            assert counts == oc.tolist()
            pos = 0
            for c in counts:
                assert sorted(ents[pos:pos + c]) == sorted(oe[pos:pos + c])
                pos += c
        assert r.count_multiple_bytes(qs[:100]) == o.search_multiple_bytes(qs[:100])[1].tolist()
        for q in qs[4:24]:                                  # configs[2]: one query at a time over all parts
            assert sorted(r.search_batch_raw([q])[0]) == sorted(o.search_multiple_bytes([q])[0])
    with pysubstringsearch.Reader(p) as single:
        assert single.chunks_per_device == [nc]


def test_multi_device_results_merged_by_several_threads(tmp_path, oracle):
    """The host merge of a multi-device reader splits the batch into ranges of queries that several threads merge side by
    side once the result is large (>= 2^18 entries or 32 MB: capi_reader_impl.h multi_batch, round 6).  A batch whose queries return
    hundreds of thousands of entries each -- single letters on 15 chunks of 1 MiB -- next to queries that return nothing:
    every API equal to the one-device reader and, on a sample, to the oracle."""
    from tests.util import gen_corpus
    src = tmp_path / 'c.txt'
    text = gen_corpus(0, 15 << 20).tobytes()
    src.write_bytes(text)
    p = str(tmp_path / 'c.idx')
    w = pysubstringsearch.Writer(p, 1 << 20)
    w.add_entries_from_file_lines(str(src))
    w.close()
    rng = np.random.default_rng(3)
    qs = [b'e', b'zzzzzzzz', b'a', b' ', b'qx', b'', b'.', b'0']
    while len(qs) < 400:
        s0 = int(rng.integers(0, len(text) - 20))
        qs.append(text[s0:s0 + int(rng.integers(2, 9))])
    order = rng.permutation(len(qs))
    qs = [qs[i] for i in order]
    with pysubstringsearch.Reader(p) as single:
        se, sc = single.search_batch_raw(qs)
        assert len(se) >= (1 << 20)                          # well beyond the threshold of the threaded merge
        starts = np.zeros(len(sc) + 1, dtype=np.int64)
        np.cumsum(sc, out=starts[1:])
        for devs in ([0, 0], [0] * 8):
            with pysubstringsearch.Reader(p, devices=devs) as r:
                ents, counts = r.search_batch_raw(qs)
                assert counts == sc and len(ents) == len(se)
                for i, c in enumerate(counts):
                    if c and c < 20000:                      # (the big ones: by length and a checksum of their bytes below)
                        assert sorted(ents[starts[i]:starts[i] + c]) == sorted(se[starts[i]:starts[i] + c]), qs[i]
                pk, spk = r.search_batch_packed(qs), single.search_batch_packed(qs)
                assert pk.counts.tolist() == spk.counts.tolist() and int(pk.offsets[-1]) == int(spk.offsets[-1])
                po, so = np.asarray(pk.offsets), np.asarray(spk.offsets)
                for i, c in enumerate(counts):               # per query: the same multiset of entry lengths, the same bytes in sum
                    a, b = int(starts[i]), int(starts[i] + c)
                    assert np.array_equal(np.sort(np.diff(po[a:b + 1])), np.sort(np.diff(so[a:b + 1])))
                    assert int(np.asarray(pk.data[int(po[a]):int(po[b])], dtype=np.uint64).sum()) == \
                        int(np.asarray(spk.data[int(so[a]):int(so[b])], dtype=np.uint64).sum())
                del pk, spk
    o = oracle.OracleReader(p)
    sample = [q for q in qs if q not in (b'e', b'a', b' ', b'', b'.', b'0')][:60]
    with pysubstringsearch.Reader(p, devices=[0] * 8) as r:
        ents, counts = r.search_batch_raw(sample)
        oe, oc = o.search_multiple_bytes(sample)
        assert counts == oc.tolist()
        pos = 0
        for c in counts:
            assert sorted(ents[pos:pos + c]) == sorted(oe[pos:pos + c])
            pos += c


def test_unchanged_call_uses_the_default_device_list(tmp_path, oracle, monkeypatch):
    """`Reader(path)` / `Writer(path)` exactly as a user of the reference writes them: with PSS_DEVICES set (or more
    than one GPU visible) they fan out over the device list without a `devices=` argument -- the reference's search
    uses every core without being asked (src/lib.rs:205-207).  PSS_DEVICES=0,0,0: three parts on the one GPU of the box."""
    from tests.util import gen_corpus
    src = tmp_path / 'c.txt'
    src.write_bytes(gen_corpus(1, 1 << 19).tobytes())
    for var in ('PSS_DEVICE', 'LOCAL_RANK'):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv('PSS_DEVICES', '0')
    p1 = str(tmp_path / 'one.idx')
    w = pysubstringsearch.Writer(p1, 1 << 16)
    assert w.devices == [0]
    w.add_entries_from_file_lines(str(src))
    w.close()
    monkeypatch.setenv('PSS_DEVICES', '0,0,0')
    p3 = str(tmp_path / 'three.idx')
    w = pysubstringsearch.Writer(p3, 1 << 16)                       # the reference's positional call
    assert w.devices == [0, 0, 0]
    w.add_entries_from_file_lines(str(src))
    w.close()
    assert pathlib.Path(p1).read_bytes() == pathlib.Path(p3).read_bytes()
    o = oracle.OracleReader(p1)
    text = src.read_bytes()
    rng = np.random.default_rng(2)
    qs = ['', 'e', 'zzzzzz'] + [text[s:s + int(rng.integers(1, 12))].decode() for s in rng.integers(0, len(text) - 20, 400)]
    qs = [q for q in qs if '\n' not in q]
    r = pysubstringsearch.Reader(p3)
    assert r.devices == [0, 0, 0]
    for q in qs[:40]:
        assert sorted(r.search(q)) == sorted(o.search(q))
    got = r.search_multiple(qs)
    assert sorted(got) == sorted(o.search_multiple(qs))
    r.close()
    monkeypatch.setenv('LOCAL_RANK', '0')                           # a launcher's pin does not override PSS_DEVICES ...
    r = pysubstringsearch.Reader(p3)
    assert r.devices == [0, 0, 0]
    r.close()
    monkeypatch.delenv('PSS_DEVICES')                               # ... but decides when PSS_DEVICES is unset
    r = pysubstringsearch.Reader(p3)
    assert r.devices == [0]
    r.close()
    r = pysubstringsearch.Reader(p3, device=0)                      # explicit arguments always win
    assert r.devices == [0]
    r.close()
    with pytest.raises(FileNotFoundError):
        pysubstringsearch.Reader(str(tmp_path / 'missing.idx'), devices=[0, 0])


def test_evict_and_promote_chunks(tmp_path, oracle):
    """Explicit residency control: Reader.evict(c) moves the suffix array of chunk c to pinned host memory (the kernels
    read it over PCIe), promote(c) brings it back; results never change, residency reports where things are.  Also
    through a multi-device reader (chunk c lives in part c % G)."""
    from tests.util import gen_corpus
    src = tmp_path / 'c.txt'
    src.write_bytes(gen_corpus(1, 1 << 18).tobytes())
    p = str(tmp_path / 'c.idx')
    w = pysubstringsearch.Writer(p, 1 << 16)
    w.add_entries_from_file_lines(str(src))
    w.close()
    text = src.read_bytes()
    rng = np.random.default_rng(5)
    qs = [text[s:s + 6] for s in rng.integers(0, len(text) - 10, 500)]
    o = oracle.OracleReader(p)
    oe, oc = o.search_multiple_bytes(qs)

    def same(r):
        ents, counts = r.search_batch_raw(qs)
        return counts == oc.tolist() and sorted(ents) == sorted(oe)

    for devs in (None, [0, 0, 0]):
        with (pysubstringsearch.Reader(p) if devs is None else pysubstringsearch.Reader(p, devices=devs)) as r:
            full = r.residency
            assert full['host_chunks'] == 0 and same(r)
            r.evict(1)
            r.evict(3)
            r.evict(3)                                   # already there: no-op
            res = r.residency
            assert res['host_chunks'] == 2 and res['host_bytes'] > 0 and res['hbm_bytes'] < full['hbm_bytes'] and same(r)
            r.promote(1)
            assert r.residency['host_chunks'] == 1 and same(r)
            r.promote(3)
            assert r.residency == full and same(r)
            with pytest.raises(ValueError):
                r.evict(10 ** 6)


def test_residency_manager_converges_on_the_hot_chunks(tmp_path, oracle, monkeypatch):
    """SURVEY 8(f) row 2, "LRU when index > HBM": six chunks, an HBM budget for the suffix arrays of two.  Batches whose
    hits land on two chunks of the host tier make the manager exchange them with the cold ones in HBM, one per batch;
    when the hot set changes, residency follows.  Results equal the oracle's throughout; chunks placed by hand stay put."""
    from tests.util import gen_corpus
    src = tmp_path / 'c.txt'
    n = 1 << 16
    src.write_bytes(b''.join(gen_corpus(0, n, c).tobytes() for c in range(6)))
    p = str(tmp_path / 'c.idx')
    w = pysubstringsearch.Writer(p, n)
    w.add_entries_from_file_lines(str(src))
    w.close()
    text = src.read_bytes()
    o = oracle.OracleReader(p)
    rng = np.random.default_rng(9)

    def batch_on(chunks, k=1500):
        qs = []
        while len(qs) < k:
            c = int(rng.choice(chunks))
            s = c * n + int(rng.integers(0, n - 20))
            q = text[s:s + 10]
            if b'\n' not in q:
                qs.append(q)
        return qs

    def check(r, qs):
        ents, counts = r.search_batch_raw(qs)
        oe, oc = o.search_multiple_bytes(qs)
        assert counts == oc.tolist() and sorted(ents) == sorted(oe)

    monkeypatch.setenv('PSS_READER_HBM_BUDGET', str(6 * (n + 128) + 2 * (n * 4 + 4096) + 1024))
    with pysubstringsearch.Reader(p) as r:
        assert r.num_chunks == 6 and r.chunk_tiers == ['hbm', 'hbm', 'host', 'host', 'host', 'host']
        for _ in range(6):
            check(r, batch_on([4, 5]))
        assert r.chunk_tiers == ['host', 'host', 'host', 'host', 'hbm', 'hbm'] and r.residency_moves >= 2
        moves = r.residency_moves
        for _ in range(3):                                  # nothing hotter outside HBM: nothing moves
            check(r, batch_on([4, 5]))
        assert r.residency_moves == moves
        for _ in range(8):                                  # the hot set changes: residency follows
            check(r, batch_on([1, 2]))
        assert r.chunk_tiers == ['host', 'hbm', 'hbm', 'host', 'host', 'host']
        for q in batch_on([0], 40):                         # single queries (fused path) count too
            check(r, [q])
        r.evict(1)                                          # placed by hand: stays, however hot
        for _ in range(4):
            check(r, batch_on([1], 1200))
        assert r.chunk_tiers[1] == 'host'
        r.set_auto_residency(False)
        tiers = r.chunk_tiers
        for _ in range(3):
            check(r, batch_on([3]))
        assert r.chunk_tiers == tiers
    with pysubstringsearch.Reader(p, devices=[0, 0]) as r:  # through a multi-device reader: every part manages its own chunks
        for _ in range(6):
            check(r, batch_on([4, 5]))
        assert r.residency_moves >= 1


@pytest.mark.timeout(300)
@pytest.mark.parametrize('chunk', [1 << 16, 1 << 20])
def test_low_latency_mode_matches_the_launch_path(tmp_path, oracle, monkeypatch, chunk):
    """Reader.set_low_latency(): single queries go through the resident search kernel (a mailbox in pinned memory, no
    launch per query).  Same results as the oracle for misses, single hits, hundreds of hits and queries with more hits
    than the resident path holds (those fall back to the launch path); the kernel's lease runs out between bursts and a
    new one starts; changing the reader's chunks (evict) and switching the mode off stop it; batches do not use it."""
    import time
    from pysubstringsearch_amd import _ffi
    from tests.util import gen_corpus
    monkeypatch.setenv('PSS_RESIDENT_IDLE_US', '2000')
    _ffi.lib.pss_reload_env()
    try:
        src = tmp_path / 'c.txt'
        src.write_bytes(gen_corpus(1, 1 << 20).tobytes())
        p = str(tmp_path / 'c.idx')
        w = pysubstringsearch.Writer(p, chunk)
        w.add_entries_from_file_lines(str(src))
        w.close()
        text = src.read_bytes()
        rng = np.random.default_rng(11)
        qs = [text[s:s + int(k)] for s, k in zip(rng.integers(0, len(text) - 40, 120), rng.integers(1, 24, 120))]
        qs = [q for q in qs if b'\n' not in q and q]
        qs += [b'zzzzzzzzqq', b'e', b' ', text[100:103]]            # a miss; thousands of hits (beyond the resident path)
        o = oracle.OracleReader(p)
        want = [sorted(o.search_multiple_bytes([q])[0]) for q in qs]
        with pysubstringsearch.Reader(p) as r:
            plain = [sorted(r.search_batch_raw([q])[0]) for q in qs]
            assert plain == want
            r.set_low_latency(True)
            s0 = r.low_latency_stats()
            got = [sorted(r.search_batch_raw([q])[0]) for q in qs]
            assert got == want
            s1 = r.low_latency_stats()
            assert s1['queries_served'] - s0['queries_served'] >= len(qs) - 8 and s1['kernels_started'] > s0['kernels_started']
            # the lease: after a pause the kernel is gone, the next query starts another
            time.sleep(0.05)
            assert sorted(r.search_batch_raw([qs[0]])[0]) == want[0]
            s2 = r.low_latency_stats()
            assert s2['kernels_started'] == s1['kernels_started'] + 1
            # a burst (of queries with few results: nothing here takes the milliseconds of the lease) stays on one kernel
            few = [i for i in range(len(qs)) if len(want[i]) <= 5][:7]
            assert len(few) == 7
            for i in range(50):
                assert sorted(r.search_batch_raw([qs[few[i % 7]]])[0]) == want[few[i % 7]]
            s3 = r.low_latency_stats()
            assert s3['queries_served'] == s2['queries_served'] + 50 and s3['kernels_started'] <= s2['kernels_started'] + 2
            # batches keep to the ordinary path
            ents, counts = r.search_batch_raw(qs[:20])
            assert counts == [len(x) for x in want[:20]] and r.low_latency_stats()['queries_served'] == s3['queries_served']
            # moving a chunk's suffix array restarts the kernel on the new table
            r.evict(0)
            assert sorted(r.search_batch_raw([qs[1]])[0]) == want[1]
            r.promote(0)
            assert sorted(r.search_batch_raw([qs[2]])[0]) == want[2]
            assert r.search('zzzzzzzzqq') == []
            r.set_low_latency(False)
            s4 = r.low_latency_stats()
            assert sorted(r.search_batch_raw([qs[4]])[0]) == want[4]
            assert r.low_latency_stats() == s4
            r.set_low_latency(True)
            assert sorted(r.search_batch_raw([qs[5]])[0]) == want[5]      # ... and the reader closes with a kernel waiting
        with pysubstringsearch.Reader(p, devices=[0, 0]) as r:            # several devices in one process: not this mode
            with pytest.raises(ValueError):
                r.set_low_latency(True)
        o.close()
    finally:
        monkeypatch.delenv('PSS_RESIDENT_IDLE_US')
        _ffi.lib.pss_reload_env()


def test_container_format_2(tmp_path, oracle):
    """Opt-in container with 64-bit lengths (no reference counterpart): same chunks, same suffix
    arrays, same search results as the reference container of the same entries; the Reader tells the
    two apart by the magic; truncation is reported like the reference's UnexpectedEof."""
    rng = random.Random(11)
    entries = [''.join(rng.choice('abcd \t') for _ in range(rng.randint(0, 30))) for _ in range(4000)] + ['', 'x']
    p1, p2 = str(tmp_path / 'v1.idx'), str(tmp_path / 'v2.idx')
    b1 = build(p1, entries, 3000)
    w = pysubstringsearch.Writer(p2, 3000, format_version=2)
    for e in entries:
        w.add_entry(e)
    w.close()
    b2 = pathlib.Path(p2).read_bytes()
    assert b2[:8] == b'PSSIDX\x02\x00' and b2[8:16] == bytes(8)
    # walk both files: identical text and suffix array bytes, chunk by chunk
    o1, o2, chunks = 0, 16, 0
    while o1 < len(b1):
        n = int.from_bytes(b1[o1:o1 + 4], 'little')
        assert int.from_bytes(b2[o2:o2 + 8], 'little') == n
        assert b1[o1 + 4:o1 + 4 + n] == b2[o2 + 8:o2 + 8 + n]
        s1, s2 = o1 + 4 + n, o2 + 8 + n
        assert int.from_bytes(b1[s1:s1 + 4], 'little') == 4 * n == int.from_bytes(b2[s2:s2 + 8], 'little')
        assert b1[s1 + 4:s1 + 4 + 4 * n] == b2[s2 + 8:s2 + 8 + 4 * n]
        o1, o2, chunks = s1 + 4 + 4 * n, s2 + 8 + 4 * n, chunks + 1
    assert o2 == len(b2) and chunks > 3
    text = '\n'.join(entries) + '\n'
    qs = [text[s:s + rng.randint(1, 6)] for s in (rng.randrange(len(text)) for _ in range(300))] + ['', 'zz', '\n']
    o = oracle.OracleReader(p1)
    oe, oc = o.search_multiple_bytes([q.encode() for q in qs])
    with pysubstringsearch.Reader(p2) as r:
        assert r.num_chunks == chunks
        ents, counts = r.search_batch_raw([q.encode() for q in qs])
        assert counts == oc.tolist() and sorted(ents) == sorted(oe)
    with pysubstringsearch.Reader(p2, shard=(1, 3)) as r:
        assert r.num_chunks == len(range(1, chunks, 3))
    pathlib.Path(str(tmp_path / 'cut.idx')).write_bytes(b2[:len(b2) - 5])
    with pytest.raises(OSError):
        pysubstringsearch.Reader(str(tmp_path / 'cut.idx'))
    with pytest.raises(ValueError):
        pysubstringsearch.Writer(str(tmp_path / 'bad.idx'), 1 << 31, format_version=2)
    with pytest.raises(ValueError):
        pysubstringsearch.Writer(str(tmp_path / 'bad.idx'), format_version=3)


@pytest.mark.parametrize('devices', [[0, 0], [0, 0, 0, 0, 0]])
def test_multi_device_writer_is_byte_identical(tmp_path, oracle, devices):
    """Writer(devices=[...]): chunk k is built on devices[k % G], several chunks in flight, records in
    chunk order -- the file must equal the single-device file and the oracle's (src/lib.rs:105-124),
    byte for byte.  One GPU here, so the list names it several times ("virtual devices"): same
    pipeline, same ordering logic, G builder threads."""
    from tests.util import gen_corpus
    src = tmp_path / 'c.txt'
    raw = gen_corpus(1, 3 << 20).tobytes()
    src.write_bytes(raw)
    rng = random.Random(3)
    single, multi, orc = str(tmp_path / 's.idx'), str(tmp_path / 'm.idx'), str(tmp_path / 'o.idx')
    for limit in (1 << 18, 300000, 1 << 20):
        def fill(w):
            w.add_entry('first')
            w.add_entries_from_file_lines(str(src))      # 3 MiB -> 3..12 chunks
            for i in range(50):
                w.add_entry('tail entry %d' % i)
                if i in (7, 8, 30):
                    w.dump_data()                          # explicit chunk boundaries, incl. back-to-back
            w.add_entry('')
            w.dump_data()                                  # a one-byte chunk in the middle of the pipeline
            w.add_entry('last')
        w = pysubstringsearch.Writer(single, limit)
        fill(w)
        w.close()
        w = pysubstringsearch.Writer(multi, limit, devices=devices)
        fill(w)
        w.finalize()
        w.add_entry('after finalize')                      # the pipeline keeps going after a finalize
        w.close()
        w = pysubstringsearch.Writer(single + '2', limit)
        fill(w)
        w.finalize()
        w.add_entry('after finalize')
        w.close()
        assert pathlib.Path(multi).read_bytes() == pathlib.Path(single + '2').read_bytes(), (limit, devices)
        oracle.use_reference_sa(True if oracle.have_reference() else False)
        ow = oracle.OracleWriter(orc, limit)
        fill(ow)
        ow.close()
        oracle.use_reference_sa(False)
        assert pathlib.Path(single).read_bytes() == pathlib.Path(orc).read_bytes(), limit
    with pysubstringsearch.Reader(multi) as r:
        assert r.num_chunks >= 4 and r.search('after finalize') == ['after finalize']
    # the host text of the chunks in flight is bounded by bytes too (PSS_WRITER_HOST_BUDGET): with room for one chunk
    # only, the pipeline degrades to one job at a time and still writes the same file
    os.environ['PSS_WRITER_HOST_BUDGET'] = '1'
    try:
        w = pysubstringsearch.Writer(multi + '3', 1 << 20, devices=devices)
        fill(w)
        w.finalize()
        w.add_entry('after finalize')
        w.close()
    finally:
        del os.environ['PSS_WRITER_HOST_BUDGET']
    assert pathlib.Path(multi + '3').read_bytes() == pathlib.Path(single + '2').read_bytes()
    with pytest.raises(ValueError):
        pysubstringsearch.Writer(multi, devices=[])
    with pytest.raises(ValueError):
        pysubstringsearch.Writer(multi, device=0, devices=[0])


def test_writer_failure_is_sticky(tmp_path):
    """A record that cannot be written (here: /dev/full) fails every later dump / finalize / close, and
    nothing is appended after the broken record."""
    if not os.path.exists('/dev/full'):
        pytest.skip('no /dev/full')
    w = pysubstringsearch.Writer('/dev/full', 64)
    for i in range(40):
        try:
            w.add_entry('entry %03d of the failing writer' % i)
        except OSError:
            break
    with pytest.raises(OSError):
        w.finalize()
    with pytest.raises(OSError):
        w.finalize()                                       # reported again, not swallowed after the first time
    with pytest.raises(OSError):
        w.close()


def test_sharded_reader_over_rccl_single_rank(tmp_path, oracle):
    """The device-resident gather on the real backend: torch.distributed with backend "nccl" (= RCCL) and
    the one GPU of the box as a world of one rank -- device tensors over the engine's workspace
    (Reader.search_batch_device), the size exchange as an RCCL collective, the download through pinned
    memory and the C merge.  (More than one rank needs more than one GPU: covered with gloo elsewhere.)"""
    import socket
    import torch
    import torch.distributed as dist
    from pysubstringsearch_amd import dist as pdist
    from tests.util import gen_corpus
    src = tmp_path / 'c.txt'
    src.write_bytes(gen_corpus(1, 1 << 19).tobytes())
    p = str(tmp_path / 'c.idx')
    w = pysubstringsearch.Writer(p, 1 << 17)
    w.add_entries_from_file_lines(str(src))
    w.close()
    text = src.read_bytes()
    rng = np.random.default_rng(21)
    qs = [b'', b'zzzzzz', b'\n']
    while len(qs) < 3000:
        s = int(rng.integers(0, len(text) - 20))
        qs.append(text[s:s + int(rng.integers(2, 12))])
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    try:
        sr = pdist.ShardedReader(p)
        blob, offsets, counts = sr.search_multiple_packed(qs)
        dr = sr.local.search_batch_device(qs[:100])
        assert dr.data.is_cuda and dr.starts.is_cuda and dr.counts.is_cuda and dr.data.numel() == dr.num_bytes
        ents, total = sr.search_multiple_bytes(qs[:500])
        sr.local.close()
    finally:
        dist.destroy_process_group()
    o = oracle.OracleReader(p)
    oe, oc = o.search_multiple_bytes(qs)
    assert counts.tolist() == oc.tolist()
    data = bytes(blob)
    got = [data[int(offsets[i]):int(offsets[i + 1])] for i in range(len(offsets) - 1)]
    assert sorted(got) == sorted(oe)
    oe5, oc5 = o.search_multiple_bytes(qs[:500])
    assert total.tolist() == oc5.tolist() and sorted(ents) == sorted(oe5)


def test_dropped_writer_flushes_at_once(pss, oracle, tmp_path):
    """`del writer` is the reference's Drop (src/lib.rs:138-144): the pending chunk is on disk when the statement
    returns -- no gc.collect(), no close() -- and a dropped Reader releases its handle the same way."""
    import gc
    import weakref
    gc.disable()
    try:
        p, q = str(tmp_path / 'g.idx'), str(tmp_path / 'o.idx')
        w = pss.Writer(p)
        ow = oracle.OracleWriter(q)
        for e in ('ten', 'tenten', 'some short string'):
            w.add_entry(e)
            ow.add_entry(e)
        ow.close()
        del w
        assert pathlib.Path(p).read_bytes() == pathlib.Path(q).read_bytes()
        r = pss.Reader(p)
        assert sorted(r.search('ten')) == ['ten', 'tenten']
        ref = weakref.ref(r)
        del r
        assert ref() is None
    finally:
        gc.enable()


def test_device_merge_equals_host_merge():
    """pss_merge_packed_device (the collecting rank of the RCCL gather merges the per-rank results in its HBM and brings
    ONE result down) against the host merge pss_merge_packed on random packed results: empty ranks, empty queries,
    one rank, sixteen ranks."""
    import torch
    from pysubstringsearch_amd import dist as pdist
    rng = np.random.default_rng(8)
    for world, nq in ((1, 50), (2, 1), (2, 777), (3, 5000), (8, 20000), (16, 300), (4, 0)):
        per_rank_host, per_rank_dev = [], []
        for r in range(world):
            counts = rng.integers(0, 4, nq).astype(np.int64) * (rng.random(nq) < 0.6)
            if r == 1:
                counts[:] = 0                                # a rank with nothing
            E = int(counts.sum())
            lens = rng.integers(0, 40, E).astype(np.int64)
            starts = np.zeros(E, dtype=np.int64)
            if E:
                starts[1:] = np.cumsum(lens)[:-1]
            blob = rng.integers(0, 256, int(lens.sum()), dtype=np.uint8)
            per_rank_host.append((blob, starts, counts))
            per_rank_dev.append((torch.from_numpy(blob).cuda(), torch.from_numpy(starts).cuda(), torch.from_numpy(counts).cuda()))
        hb, ho, hc = pdist.merge_packed_starts(per_rank_host) if nq else (np.zeros(0, np.uint8), np.zeros(1, np.int64), np.zeros(0, np.int64))
        db, do, dc = pdist.merge_on_device(per_rank_dev, nq)
        assert np.array_equal(dc, hc) and np.array_equal(do, ho) and np.array_equal(db, hb), (world, nq)
    # What arrives from another process is checked before anything is copied by its offsets: counts that do not add up to
    # the rank's entries, entry starts that do not ascend or leave the rank's bytes -> ValueError, as on the host.
    blob = torch.arange(100, dtype=torch.uint8).cuda()
    good = (blob, torch.tensor([0, 10, 50], dtype=torch.int64).cuda(), torch.tensor([1, 2, 0], dtype=torch.int64).cuda())
    pdist.merge_on_device([good], 3)
    for bad in ((blob, torch.tensor([0, 60, 50], dtype=torch.int64).cuda(), good[2]),            # starts descend
                (blob, torch.tensor([0, 10, 500], dtype=torch.int64).cuda(), good[2]),           # a start beyond the bytes
                (blob, good[1], torch.tensor([1, 1, 0], dtype=torch.int64).cuda())):             # counts != entries
        with pytest.raises(ValueError):
            pdist.merge_on_device([good, bad], 3)


def test_single_query_over_many_chunks_with_thousands_of_hits(tmp_path, oracle):
    """One query over 33 .. 64 chunks (one workgroup per (query, chunk) pair, up to four with this many pairs) whose
    pairs hold 1025 .. 4096 hits each: the fused one-kernel path answers it (no overflow back to the general pipeline),
    and the result is the oracle's whatever path ran."""
    from tests.util import gen_corpus
    src = tmp_path / 'c.txt'
    src.write_bytes(gen_corpus(0, 40 << 16).tobytes())
    p = str(tmp_path / 'c.idx')
    w = pysubstringsearch.Writer(p, 1 << 16)          # 40-odd chunks of 64 KiB
    w.add_entries_from_file_lines(str(src))
    w.close()
    o = oracle.OracleReader(p)
    with pysubstringsearch.Reader(p) as r:
        assert 33 <= r.num_chunks <= 64
        for q in ('e', 'a', ' ', 'th', 'zq9'):            # a single letter: ~1 700 hits in every 64 KiB chunk
            got, exp = r.search(q), o.search(q)
            assert len(got) == len(exp) and sorted(got) == sorted(exp), q
        assert max(o.search_multiple_bytes([b'e'])[1]) > 33 * 1025 / 2



def test_rccl_gather_inside_the_c_abi_single_rank(tmp_path, oracle):
    """pss_comm_* / pss_gather_packed_rccl with a communicator of ONE rank (all a one-GPU box can form): the RCCL entry
    points are found in the process, the communicator comes up, the collecting rank's own device result goes through the
    device-side merge and comes back equal to the oracle's.  The two-rank exchange runs in tests/test_dist_gpu.py."""
    from tests.util import gen_corpus
    from pysubstringsearch_amd import dist as pdist
    src = tmp_path / 'c.txt'
    src.write_bytes(gen_corpus(0, 1 << 18).tobytes())
    p = str(tmp_path / 'c.idx')
    w = pysubstringsearch.Writer(p, 1 << 16)
    w.add_entries_from_file_lines(str(src))
    w.close()
    text = src.read_bytes()
    rng = np.random.default_rng(3)
    qs = [b'', b'e', b'zzzzzz'] + [text[s:s + int(rng.integers(1, 12))] for s in rng.integers(0, len(text) - 20, 2000)]
    o = oracle.OracleReader(p)
    comm = pdist.EngineComm(0, 1, 0, lambda raw: raw)
    with pysubstringsearch.Reader(p, device=0) as r:
        for batch in (qs[:1], qs[:40], qs, []):
            data, offsets, counts = comm.gather(r, batch, dst=0)
            oe, oc = o.search_multiple_bytes(batch) if batch else ([], np.zeros(0, np.int64))
            assert counts.tolist() == list(oc)
            got = pdist.packed_to_list(data, offsets)
            assert sorted(got) == sorted(oe)
    comm.close()


@pytest.mark.parametrize('chunk_len', [None, 4096, 300])
def test_result_order_sa_equals_the_reference_lists(tmp_path, oracle, chunk_len, monkeypatch):
    """Reader(order='sa') / pss_reader_set_result_order: the entries of a chunk come in the reference's order --
    suffix-array order of every entry's FIRST hit (src/lib.rs:262-276; the oracle emits exactly that,
    oracle/pss_oracle.c search_chunk_src) -- so the lists are equal element by element, not just as multisets:
    one chunk, several chunks (chunks in index order), single queries, batches, the empty pattern, patterns that hold
    a newline, entries that hold the pattern many times.  The default order is the same multiset."""
    rng = random.Random(11 if chunk_len is None else chunk_len)
    entries = ['ten', 'tenten', 'xtenyten', 'ten', 'tententen', 'one', 'onet', 'aaa', 'aaaa', 'aaaaaaaa', '']
    for _ in range(700):
        entries.append(''.join(rng.choice('ab') for _ in range(rng.randrange(1, 40))))
    for _ in range(150):
        entries.append(' '.join(rng.choice(['ten', 'eleven', 'net', 'tent', 'aa']) for _ in range(rng.randrange(1, 8))))
    rng.shuffle(entries)
    p = str(tmp_path / 'o.idx')
    build(p, entries, chunk_len)
    o = oracle.OracleReader(p)
    queries = ['ten', 'a', 'aa', 'ab', 'ba', 'aaa', 'n\nt', 'a\n', '\n', '', 'en t', 'tent', 'zz', 'b', 'abab', 'net ten']
    differs = 0
    with pysubstringsearch.Reader(p, order='sa') as r, pysubstringsearch.Reader(p) as d:
        assert r.result_order == 'sa' and d.result_order == 'text'
        assert r.num_chunks == o.num_chunks and (chunk_len is None) == (r.num_chunks == 1)
        for q in queries:
            want = o.search(q)
            assert r.search(q) == want, q                      # the reference's list, element by element
            got_default = d.search(q)
            assert sorted(got_default) == sorted(want), q
            differs += got_default != want
        assert r.search_multiple(queries) == o.search_multiple(queries)
        big = [q for q in queries for _ in range(40)]              # a batch beyond the fused and mid pipelines' pair counts
        assert r.search_multiple(big) == o.search_multiple(big)
        assert r.count_multiple(queries) == [len(o.search(q)) for q in queries]
        # switching an open reader
        d.set_result_order('sa')
        assert d.search('ten') == o.search('ten')
        d.set_result_order('text')
        with pytest.raises(ValueError):
            d.set_result_order('nope')
    assert differs > 0, 'the corpus was meant to tell the two orders apart'
    monkeypatch.setenv('PSS_RESULT_ORDER', 'sa')                  # the environment sets the default of new readers
    with pysubstringsearch.Reader(p) as e:
        assert e.result_order == 'sa' and e.search('aa') == o.search('aa')
    o.close()
