"""CPU tests: the oracle restatement against the committed golden vectors
(reference tests re-expressed as data, libsais-generated SAs and .idx bytes)."""
import pathlib
import hashlib
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, 'golden')


def load(name):
    return json.loads(pathlib.Path(os.path.join(GOLD, name)).read_text(encoding='utf-8'))


def build_idx(W, path, entries, max_chunk_len=None, dump_after=()):
    w = W(path, max_chunk_len)
    for i, e in enumerate(entries):
        w.add_entry(e)
        if i in dump_after:
            w.dump_data()
    w.finalize()
    w.close()
    return pathlib.Path(path).read_bytes()


def test_little_endian_host():
    assert sys.byteorder == 'little'   # raw int32 writes == i32le (src/lib.rs:117-119)


def test_reference_cases(oracle, tmp_path):
    oracle.use_reference_sa(False)     # the restatement's own SA, not libsais
    for case in load('reference_cases.json')['cases']:
        if 'missing_path' in case:
            with pytest.raises(FileNotFoundError):
                oracle.OracleReader(str(tmp_path / case['missing_path']))
            continue
        p = str(tmp_path / (case['name'] + '.idx'))
        idx = build_idx(oracle.OracleWriter, p, case['entries'])
        assert hashlib.sha256(idx).hexdigest() == case['idx_sha256'], case['name']
        if case['idx_hex']:
            assert idx.hex() == case['idx_hex']
        r = oracle.OracleReader(p)
        for s in case['searches']:
            assert sorted(r.search(s['substring'])) == sorted(s['expected']), (case['name'], s['substring'])
        for s in case['search_multiple']:
            assert sorted(r.search_multiple(s['substrings'])) == sorted(s['expected'])
        r.close()


def test_container_cases(oracle, tmp_path):
    oracle.use_reference_sa(False)
    gold = load('container_cases.json')
    for case in gold['cases']:
        p = str(tmp_path / (case['name'] + '.idx'))
        idx = build_idx(oracle.OracleWriter, p, case['entries'], case['max_chunk_len'], case['dump_after'])
        assert idx.hex() == case['idx_hex'], case['name']
        r = oracle.OracleReader(p)
        for s in case['searches']:
            assert sorted(r.search(s['substring'])) == s['expected'], (case['name'], s['substring'])
        r.close()
    for case in gold['file_ingest']:
        src = tmp_path / (case['name'] + '.txt')
        src.write_bytes(bytes.fromhex(case['input_hex']))
        p = str(tmp_path / (case['name'] + '.idx'))
        w = oracle.OracleWriter(p, case.get('max_chunk_len'))
        w.add_entries_from_file_lines(str(src))
        w.close()
        assert pathlib.Path(p).read_bytes().hex() == case['idx_hex'], case['name']


def test_known_multi_chunk_layout(oracle, tmp_path):
    # three chunks with these exact suffix arrays
    p = str(tmp_path / 'm.idx')
    build_idx(oracle.OracleWriter, p, ['ten', 'ten', 'tenten', 'x'], 8)
    r = oracle.OracleReader(p)
    assert r.num_chunks == 3
    exp = [(b'ten\nten\n', [7, 3, 5, 1, 6, 2, 4, 0]), (b'tenten\n', [6, 4, 1, 5, 2, 3, 0]), (b'x\n', [1, 0])]
    for c, (data, sa) in enumerate(exp):
        d, s = r.chunk(c)
        assert d == data and s.tolist() == sa
    assert sorted(r.search('ten')) == ['ten', 'ten', 'tenten']
    assert r.search_multiple(['te', 'en']).count('ten') == 4   # duplicates across queries are kept


def _kat_input(k, oracle):
    from tests.util import gen_corpus
    fib = [b'a', b'ab']
    while len(fib[-1]) < 987:
        fib.append(fib[-1] + fib[-2])
    rng = np.random.default_rng(12345)
    perm = rng.permutation(256).astype(np.uint8).tobytes()
    zfn = np.array([0, 255, 10], dtype=np.uint8)[rng.integers(0, 3, 4096)].tobytes()
    table = {
        'a1000_nl': b'a' * 1000 + b'\n', 'fibonacci_987': fib[-1], 'perm256': perm, 'zero_ff_nl_4k': zfn,
        'periodic_64x64': (b'a' * 63 + b'\n') * 64, 'all_zero_5000': b'\x00' * 5000,
        'zeros_tail': b'\xff' * 4097 + b'\x00' * 17,
        'readme_chunk': b'some short string\nanother but now a longer string\nmore text to add\n',
    }
    if k['name'] in table:
        return table[k['name']]
    kinds = {'lines_1MiB': 0, 'words_1MiB': 1, 'runs_1MiB': 2, 'periodic_1MiB': 3}
    return gen_corpus(kinds[k['name']], 1 << 20).tobytes()


def test_sa_kats_restatement(oracle):
    for k in load('sa_kats.json')['kats']:
        data = _kat_input(k, oracle)
        assert hashlib.sha256(data).hexdigest() == k['text_sha256'], k['name']
        if k['name'] in ('runs_1MiB', 'periodic_1MiB'):
            continue   # O(n log n) rounds x 1 MiB on CPU: covered by the GPU test
        sa = oracle.sa_restatement(data)
        assert hashlib.sha256(sa.tobytes()).hexdigest() == k['sa_sha256'], k['name']
        if 'sa' in k:
            assert sa.tolist() == k['sa']


def test_restatement_matches_reference_when_built(oracle):
    if not oracle.have_reference():
        pytest.skip('oracle/_ref/libsais.so not present')
    rng = np.random.default_rng(5)
    for alpha in (1, 2, 3, 39, 256):
        for n in (0, 1, 2, 3, 17, 255, 1000, 5000):
            t = rng.integers(0, alpha, size=n, dtype=np.uint8)
            assert (oracle.sa_restatement(t) == oracle.sa_reference(t)).all()
    t = oracle.gen_lines(1 << 18)
    assert (oracle.sa_restatement(t) == oracle.sa_reference(t)).all()


def test_generator_golden(oracle):
    from tests.util import gen_corpus
    t = gen_corpus(0, 1 << 20)
    assert hashlib.sha256(t.tobytes()).hexdigest() == \
        '33246a0e40f61c2a592fd16dc644ade2e25eae811ca2d7457ce4154000485e99'
    assert (t == oracle.gen_lines(1 << 20)).all()   # independent C restatement of the generator


def test_thread_per_chunk_baseline_equals_plain_search(oracle, tmp_path):
    """orc_bench_search (the reference-shaped CPU baseline of SURVEY 8(d)(ii): queries one at a time,
    one worker per chunk, results appended under a mutex, suffix arrays in RAM or probed in the index
    file with lseek + read(8 KiB)) returns, per query, exactly what the plain restatement returns."""
    texts = [oracle.gen_lines(1 << 16, c) for c in range(5)]
    sas = [oracle.sa(t) for t in texts]
    r = oracle.OracleReader.from_arrays(texts, sas)
    assert r.num_chunks == 5
    rng = np.random.default_rng(2)
    qs = [b'', b'\n', b'zzzzzz', b'e']
    for _ in range(600):
        c, s, ln = int(rng.integers(0, 5)), int(rng.integers(0, (1 << 16) - 40)), int(rng.integers(2, 12))
        qs.append(texts[c][s:s + ln].tobytes())
    ents, counts = r.search_multiple_bytes(qs)
    for threads in (1, 2, 5, 64):
        for dedupe in ('hash', 'sort'):
            b = r.bench_search(qs, threads, dedupe=dedupe)
            assert np.array_equal(b['counts'], counts) and b['entries'] == len(ents) and b['bytes'] == sum(map(len, ents))
            assert b['threads'] == min(threads, 5) and b['seconds'] > 0 and b['dedupe'] == dedupe
    # the hash-set dedupe (lib.rs:262 as written: what the timed baseline uses) returns the checker's lists, entry by
    # entry and in the same order -- SA order of the first hit -- on high-hit queries too
    try:
        oracle.set_hash_dedupe(True)
        e_h, c_h = r.search_multiple_bytes(qs)
    finally:
        oracle.set_hash_dedupe(False)
    assert np.array_equal(c_h, counts) and e_h == ents and int(counts.max()) > 1000
    p = str(tmp_path / 'five.idx')
    with open(p, 'wb') as f:            # chunk records, src/lib.rs:112-119
        for t, s in zip(texts, sas):
            f.write(np.uint32(t.size).tobytes() + t.tobytes() + np.uint32(4 * t.size).tobytes() + s.astype('<i4').tobytes())
    on_disk = oracle.OracleReader(p, load_sa=False)     # suffix arrays stay in the file (src/lib.rs:179-182)
    for threads in (1, 5):
        b = on_disk.bench_search(qs, threads, disk=True)
        assert np.array_equal(b['counts'], counts) and b['entries'] == len(ents)
    with pytest.raises(RuntimeError):
        on_disk.search_bytes(b'a')                      # no suffix array in RAM: only the disk path serves it
    loaded = oracle.OracleReader(p)
    e2, c2 = loaded.search_multiple_bytes(qs)
    assert np.array_equal(c2, counts) and sorted(e2) == sorted(ents)


def test_big_goldens_are_well_formed_and_the_checksum_is_libsais(oracle):
    """tests/golden/sa_big.json (libsais on full 512 MiB chunks, generated by make_golden_big.py):
    every BASELINE chunk is present; the positional checksum formula the GPU side evaluates in torch
    is the one the generator evaluated in numpy (re-derived here on 1 MiB with the real libsais)."""
    import bench
    gold = bench.load_big_goldens()
    n = 1 << 29
    for c in range(15):
        assert ('lines', c, n) in gold, c
    for kind in ('words', 'runs', 'periodic'):
        assert (kind, 0, n) in gold
    for g in gold.values():
        assert len(g['sa_sha256']) == 64 and len(g['sa_stride']) == 256 and 0 <= g['sa_poly64'] < 1 << 64
        assert sorted(set(g['sa_stride'])) == sorted(g['sa_stride'])      # distinct suffixes
    if not oracle.have_reference():
        pytest.skip('oracle/_ref/libsais.so not present')
    from tests.golden.make_golden_big import poly64
    t = oracle.gen_lines(1 << 20, 3)
    sa = oracle.sa_reference(t)
    assert poly64(sa) == bench.sa_poly64_numpy(sa)
    assert hashlib.sha256(sa.astype('<i4').tobytes()).hexdigest() != hashlib.sha256(np.sort(sa).tobytes()).hexdigest()
