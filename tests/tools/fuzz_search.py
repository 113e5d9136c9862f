"""Time-boxed fuzzing of the search path on mid-size multi-chunk indexes: batch sizes from 1 to
tens of thousands of queries (fused small-batch kernel, wave-per-pair and lane-per-pair interval
search, arena overflow, > 1024 hits per (query, chunk)), long and newline-crossing patterns,
sharded readers, Writers and Readers over several lanes of one GPU (container formats 1, 2, striped).  Everything is
compared with the oracle's restatement of Reader::search.

    python tests/tools/fuzz_search.py [seconds=120] [seed0=<time>]"""
import pathlib
import os
import random
import sys
import tempfile
import time

sys.path.insert(0, '.')
import pysubstringsearch  # noqa: E402
from oracle import oracle as O  # noqa: E402
from pysubstringsearch_amd import Reader  # noqa: E402


def one_case(seed, tmp):
    rng = random.Random(seed)
    alphabet = rng.choice(['ab', 'abc', 'abcdefgh', 'aé☃b', 'ab \t.', 'abcdefghijklmnopqrstuvwxyz'])
    m = int(2 ** rng.uniform(3, 16))
    L = rng.choice([3, 10, 40, 200])
    pool = [''.join(rng.choice(alphabet) for _ in range(rng.randint(0, L))) for _ in range(max(1, m // rng.choice([1, 2, 20])))]
    entries = [rng.choice(pool) for _ in range(m)]
    total = sum(len(e.encode()) + 1 for e in entries)
    chunks = rng.choice([1, 1, 2, 7, 40])
    limit = max(total // chunks + 1, max(len(e.encode()) for e in entries) + 1)
    p, q = os.path.join(tmp, 'g.idx'), os.path.join(tmp, 'o.idx')
    for path, W in ((p, pysubstringsearch.Writer), (q, O.OracleWriter)):
        w = W(path, limit)
        for e in entries:
            w.add_entry(e)
        w.finalize()
        if hasattr(w, 'close'):
            w.close()
    assert pathlib.Path(p).read_bytes() == pathlib.Path(q).read_bytes(), 'container differs'
    text = '\n'.join(entries) + '\n'
    nq = rng.choice([1, 3, 50, 1500, 40000 if total < 300000 else 3000])
    queries = []
    for _ in range(nq):
        r = rng.random()
        if r < 0.6:
            s = rng.randrange(len(text))
            queries.append(text[s:s + rng.choice([1, 2, 3, 5, 9, 17, 60])])
        elif r < 0.9:
            queries.append(''.join(rng.choice(alphabet) for _ in range(rng.randint(1, 7))))
        else:
            queries.append(rng.choice(['', '\n', alphabet[0], alphabet[0] + '\n', '\n' + alphabet[0]]))
    if rng.random() < 0.5:
        os.environ['PSS_NO_SMALL_PATH'] = '1'
    else:
        os.environ.pop('PSS_NO_SMALL_PATH', None)
    if rng.random() < 0.3:
        os.environ['PSS_WAVE_SEARCH'] = '1'
    else:
        os.environ.pop('PSS_WAVE_SEARCH', None)
    if rng.random() < 0.4:
        os.environ['PSS_NO_BLOCK_PATH'] = '1'
    else:
        os.environ.pop('PSS_NO_BLOCK_PATH', None)
    if rng.random() < 0.3:
        # suffix arrays beyond this many bytes of HBM stay in pinned host memory
        os.environ['PSS_READER_HBM_BUDGET'] = str(rng.choice([0, total * 2, total * 4]))
    else:
        os.environ.pop('PSS_READER_HBM_BUDGET', None)
    if rng.random() < 0.3:
        os.environ['PSS_NO_MID_PIPELINE'] = '1'
    else:
        os.environ.pop('PSS_NO_MID_PIPELINE', None)
    if rng.random() < 0.3:
        os.environ['PSS_NO_SEARCH_STAGE'] = '1'
    else:
        os.environ.pop('PSS_NO_SEARCH_STAGE', None)
    if rng.random() < 0.3:
        os.environ['PSS_NO_GROUP_SEARCH'] = '1'
    else:
        os.environ.pop('PSS_NO_GROUP_SEARCH', None)
    os.environ.pop('PSS_NO_KEY_SAMPLES', None)
    os.environ.pop('PSS_SAMPLE_SHIFT', None)
    r = rng.random()
    if r < 0.15:
        os.environ['PSS_NO_KEY_SAMPLES'] = '1'
    elif r < 0.85:
        os.environ['PSS_SAMPLE_SHIFT'] = str(rng.randint(0, 8))     # dense key-sample tables on these small chunks
    from pysubstringsearch_amd import _ffi
    _ffi.lib.pss_reload_env()        # the search switches are read once, not per call
    qb = [s.encode() for s in queries]
    o = O.OracleReader(q)
    oe, oc = o.search_multiple_bytes(qb)
    with Reader(p) as r:
        ents, counts = r.search_batch_raw(qb)
        assert counts == oc.tolist(), 'per-query counts differ'
        pos = 0
        for c in counts:
            assert sorted(ents[pos:pos + c]) == sorted(oe[pos:pos + c]), 'multiset differs'
            pos += c
        pk = r.search_batch_packed(qb)
        assert pk.counts.tolist() == counts and len(pk.offsets) == len(ents) + 1
        assert bytes(pk.data) == b''.join(ents)
        # single queries, on the launch path and through the resident kernel of the low-latency mode (a short lease
        # now and then, so that kernels come and go between queries)
        if rng.random() < 0.5:
            os.environ['PSS_RESIDENT_IDLE_US'] = str(rng.choice([20, 200, 1000]))
            _ffi.lib.pss_reload_env()
        pos, starts = 0, []
        for c in counts:
            starts.append(pos)
            pos += c
        pick = [rng.randrange(len(qb)) for _ in range(min(len(qb), 60))]
        for mode in (False, True):
            r.set_low_latency(mode)
            for i in pick:
                e1, c1 = r.search_batch_raw([qb[i]])
                assert c1 == [counts[i]] and sorted(e1) == sorted(oe[starts[i]:starts[i] + counts[i]]), ('single query differs', mode)
        r.set_low_latency(False)
        os.environ.pop('PSS_RESIDENT_IDLE_US', None)
    # several lanes in one process (round 6: added to the campaign after it found the I/O-pool race): a Writer whose
    # chunks go round-robin to k builder lanes of this GPU must write the same bytes, in format 1 and -- format 2, striped
    # or not -- an index that reads back equal; a Reader over k lanes (host merge of the lanes' results, by several
    # threads when large) must return what the one-lane reader returned
    if rng.random() < 0.5:
        k = rng.choice([2, 3, 8])
        fmt = rng.choice([1, 1, 2])
        striped = fmt == 2 and rng.random() < 0.5
        pm = os.path.join(tmp, 'm.idx')
        kw = dict(devices=[0] * k)
        if fmt == 2:
            kw.update(format_version=2, striped=striped)
        w = pysubstringsearch.Writer(pm, limit, **kw)
        for e in entries:
            w.add_entry(e)
        w.finalize()
        w.close()
        if fmt == 1:
            assert pathlib.Path(pm).read_bytes() == pathlib.Path(q).read_bytes(), 'container of a %d-lane writer differs' % k
        nsub = min(len(qb), 2000)
        want = sum(oc.tolist()[:nsub])
        for path, devs in ((pm, None), (p, [0] * rng.choice([2, 5, 8])), (pm, [0] * k)):
            with (Reader(path, devices=devs) if devs else Reader(path)) as r:
                e, c = r.search_batch_raw(qb[:nsub])
                assert c == oc.tolist()[:nsub], ('lanes: counts differ', fmt, striped, devs)
                pos = 0
                for cc in c:
                    assert sorted(e[pos:pos + cc]) == sorted(oe[pos:pos + cc]), ('lanes: multiset differs', fmt, striped, devs)
                    pos += cc
                assert pos == want
                assert r.count_multiple_bytes(qb[:nsub]) == c, ('lanes: count API differs', fmt, striped, devs)
        for f in os.listdir(tmp):
            if f.startswith('m.idx'):
                os.remove(os.path.join(tmp, f))
    # shards: the union over ranks is the whole result, per query
    k = rng.choice([2, 3, 8])
    per = [0] * len(qb)
    allents = []
    for i in range(k):
        with Reader(p, shard=(i, k)) as r:
            e, c = r.search_batch_raw(qb[:200])
            per = [a + b for a, b in zip(per, c + [0] * (len(per) - len(c)))]
            allents += e
    assert per[:min(200, len(qb))] == oc.tolist()[:200], 'sharded counts differ'
    assert sorted(allents) == sorted(oe[:sum(oc.tolist()[:200])]), 'sharded multiset differs'
    o.close()


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    O.use_reference_sa(O.have_reference())
    t0 = time.time()
    cases = 0
    with tempfile.TemporaryDirectory() as tmp:
        while time.time() - t0 < budget:
            if os.environ.get('FUZZ_TRACE'):
                print('seed', seed, file=sys.stderr, flush=True)
            try:
                one_case(seed, tmp)
            except Exception as e:   # noqa: BLE001
                print(f'FAIL seed={seed}: {type(e).__name__}: {e}')
                raise
            cases += 1
            seed += 1
    print(f'fuzz_search: {cases} cases in {time.time() - t0:.0f} s, all equal to the oracle (last seed {seed - 1})')


if __name__ == '__main__':
    main()
