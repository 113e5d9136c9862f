"""Property tests (hypothesis): the oracle against brute force on CPU; the HIP engine
against the oracle on the GPU, over arbitrary byte strings and entry lists."""
import pathlib
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

small_alphabet_bytes = st.lists(st.sampled_from([0, 1, 10, 97, 98, 255]), min_size=0, max_size=200).map(bytes)
any_bytes = st.binary(min_size=0, max_size=300)


def brute_sa(t: bytes):
    return sorted(range(len(t)), key=lambda i: t[i:])


@settings(max_examples=150, deadline=None)
@given(st.one_of(small_alphabet_bytes, any_bytes))
def test_oracle_sa_is_the_sorted_suffix_order(oracle, t):
    assert oracle.sa_restatement(t).tolist() == brute_sa(t)


def brute_search(entries, pattern: str):
    # reference semantics: entries joined with '\n'; a hit may start inside an entry and run
    # across newlines; the entry containing the START of the hit is returned, once per entry
    text = ('\n'.join(entries) + '\n').encode()
    pat = pattern.encode()
    starts = []
    pos = 0
    for e in entries:
        starts.append(pos)
        pos += len(e.encode()) + 1
    out = []
    for k, e in enumerate(entries):
        lo, hi = starts[k], starts[k] + len(e.encode())      # hit start may be anywhere in [lo, hi] (hi = the '\n')
        if any(text.startswith(pat, p) for p in range(lo, hi + 1)):
            out.append(e)
    return out


def chunks_of(entries, cap):
    """Writer::add_entry's flush rule (src/lib.rs:96-100), restated in Python: an entry that does not fit
    (len + entry + 1 > capacity) starts a new chunk.  (Entries here are far below the capacity, so the Vec
    growth corner does not arise.)"""
    chunks, cur, size = [], [], 0
    for e in entries:
        need = len(e.encode()) + 1
        if size + need > cap and cur:
            chunks.append(cur)
            cur, size = [], 0
        cur.append(e)
        size += need
    if cur:
        chunks.append(cur)
    return chunks


entry = st.text(alphabet=st.sampled_from('ab é'), min_size=0, max_size=12)


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(st.lists(entry, min_size=1, max_size=30), st.text(alphabet=st.sampled_from('ab é\n'), min_size=0, max_size=4))
def test_oracle_search_matches_brute_force(oracle, tmp_path_factory, entries, pattern):
    p = str(tmp_path_factory.mktemp('h') / 'o.idx')
    w = oracle.OracleWriter(p, 64)
    for e in entries:
        w.add_entry(e)
    w.close()
    r = oracle.OracleReader(p)
    # every chunk is searched on its own (src/lib.rs:207): a hit -- also one that runs across newlines, or
    # starts on an entry's terminating newline -- must lie inside one chunk's text
    want = [e for ch in chunks_of(entries, 64) for e in brute_search(ch, pattern)]
    assert r.num_chunks == len(chunks_of(entries, 64))
    assert sorted(r.search(pattern)) == sorted(want)
    r.close()


@pytest.mark.gpu
@settings(max_examples=120, deadline=None)
@given(st.one_of(small_alphabet_bytes, any_bytes, st.binary(min_size=4000, max_size=9000)))
def test_gpu_sa_matches_oracle(oracle, t):
    from tests.util import sa_gpu
    assert (sa_gpu(t) == oracle.sa_restatement(t)).all()


@pytest.mark.gpu
@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(st.lists(entry, min_size=1, max_size=60), st.lists(st.text(alphabet=st.sampled_from('ab é\n'), min_size=0, max_size=5),
                                                           min_size=1, max_size=8), st.sampled_from([None, 16, 40]))
def test_gpu_container_and_search_match_oracle(oracle, tmp_path_factory, entries, patterns, limit):
    import pysubstringsearch
    d = tmp_path_factory.mktemp('g')
    p, q = str(d / 'g.idx'), str(d / 'o.idx')
    if limit is not None:
        limit = max(limit, max(len(e.encode()) for e in entries) + 1)
    w, ow = pysubstringsearch.Writer(p, limit), oracle.OracleWriter(q, limit)
    for e in entries:
        w.add_entry(e)
        ow.add_entry(e)
    w.close()
    ow.close()
    assert pathlib.Path(p).read_bytes() == pathlib.Path(q).read_bytes()
    o = oracle.OracleReader(q)
    with pysubstringsearch.Reader(p) as r:
        assert sorted(r.search_multiple(patterns)) == sorted(o.search_multiple(patterns))
        for s in patterns:
            assert sorted(r.search(s)) == sorted(o.search(s))
    o.close()


# ---- the argument behind the run-length path (pysubstringsearch_amd/csrc/rle_build.hip), on the CPU ----

def rle_suffix_array(t: bytes):
    """Suffix array by way of the run-length reduced string, exactly as rle_build.hip derives it (its
    header comment): (1) maximal runs; (2) one symbol per run, meta = (byte, type, type ? -L : L) with
    type = (next byte > byte), the end of the text counting as a byte below all; the run heads are ordered
    like the suffixes of the meta string (sorted here by brute force); (3) every suffix gets the key
    (byte, type, type ? -r : r) with r = bytes left in its run, and equal keys are ordered by the rank of
    the next run head -- realised, as on the GPU, by generating the suffixes in that order and sorting
    STABLY by the key alone."""
    n = len(t)
    starts = [i for i in range(n) if i == 0 or t[i] != t[i - 1]] + [n]
    S = len(starts) - 1
    meta = []
    for k in range(S):
        c, L = t[starts[k]], starts[k + 1] - starts[k]
        typ = 1 if starts[k + 1] < n and t[starts[k + 1]] > c else 0
        meta.append((c, typ, -L if typ else L))
    sar = sorted(range(S), key=lambda k: meta[k:])                 # suffix array of the reduced string
    order = [S - 1] + [k - 1 for k in sar if k > 0]                # runs by the rank of their successor's head
    gen = []
    for k in order:
        c, typ, _ = meta[k]
        for x in range(starts[k], starts[k + 1]):
            r = starts[k + 1] - x
            gen.append(((c, typ, -r if typ else r), x))
    gen.sort(key=lambda kv: kv[0])                                 # stable
    return [x for _, x in gen]


run_text = st.lists(st.tuples(st.sampled_from([0, 10, 97, 98, 255]), st.integers(1, 9)), min_size=1, max_size=40).map(
    lambda runs: b''.join(bytes([c]) * length for c, length in runs))


@settings(max_examples=300, deadline=None)
@given(st.one_of(run_text, small_alphabet_bytes.filter(lambda b: len(b) > 0)))
def test_run_length_reduction_gives_the_suffix_array(t):
    assert rle_suffix_array(t) == brute_sa(t)


# ---- the argument behind the closed form for texts that repeat one word (rle_build.hip, period_suffix_array), on the CPU ----

def periodic_suffix_array(t: bytes):
    """Suffix array of a text whose first m bytes have smallest period p (>= 2), exactly as period_suffix_array derives it:
    suffixes that start more than margin = 2p + 2(n - m) + 1 bytes before the repetition ends form one block per rotation
    of the word, the blocks ordered like the rotations, inside a block by position -- ascending or descending, decided by the
    byte that ends the repetition; the remaining suffixes are sorted directly and fall between the blocks (compared with
    a block = with its rotation repeated for as long as the late suffix lasts, the late one first on a tie).  Returns None
    when the text is not of that shape."""
    n = len(t)
    p = next((q for q in range(1, n) if all(t[i] == t[i + q] for i in range(min(n - q, 4 * q + 8)))), 0)
    if p < 2 or 4 * p > n:
        return None
    m = next((i + p for i in range(n - p) if t[i] != t[i + p]), n)
    # p must be the smallest period of the repetition
    for q in range(1, p):
        if all(t[i] == t[i + q] for i in range(m - q)):
            return None
    tail = n - m
    margin = 2 * p + 2 * tail + 1
    if 4 * margin > m:
        return None
    long_end = m - margin
    W = t[:p]
    desc = m == n or t[m] < W[m % p]
    blocks = [c for c in range(p) if c < long_end]
    late = list(range(long_end, n))

    def rot(c, k):
        return W[(c + k) % p]

    import functools

    def cmp(a, b):                      # items: ('b', class) or ('l', start)
        if a[0] == 'b' and b[0] == 'b':
            x, y = [rot(a[1], k) for k in range(p)], [rot(b[1], k) for k in range(p)]
            return -1 if x < y else (1 if x > y else 0)
        if a[0] == 'l' and b[0] == 'l':
            x, y = t[a[1]:], t[b[1]:]
            return -1 if x < y else (1 if x > y else 0)
        late_first = a[0] == 'l'
        x, blk = (a, b) if late_first else (b, a)
        s = t[x[1]:]
        r = bytes(rot(blk[1], k) for k in range(len(s)))
        c = -1 if s <= r else 1        # the late suffix first on a tie (it is the shorter one)
        return c if late_first else -c

    items = sorted([('b', c) for c in blocks] + [('l', i) for i in late], key=functools.cmp_to_key(cmp))
    sa = []
    for kind, v in items:
        if kind == 'l':
            sa.append(v)
        else:
            members = list(range(v, long_end, p))
            sa.extend(reversed(members) if desc else members)
    return sa


periodic_text = st.tuples(st.binary(min_size=2, max_size=7).filter(lambda w: len(set(w)) > 1), st.integers(30, 140),
                          st.binary(min_size=0, max_size=5)).map(lambda x: (x[0] * (x[1] // len(x[0]) + 2))[:x[1]] + x[2])
two_letter_periodic = st.tuples(st.lists(st.sampled_from([97, 98]), min_size=2, max_size=9).map(bytes).filter(lambda w: len(set(w)) > 1),
                                st.integers(40, 160), st.lists(st.sampled_from([97, 98]), min_size=0, max_size=6).map(bytes)).map(
    lambda x: (x[0] * (x[1] // len(x[0]) + 2))[:x[1]] + x[2])


@settings(max_examples=400, deadline=None)
@given(st.one_of(periodic_text, two_letter_periodic))
def test_closed_form_for_one_repeated_word_gives_the_suffix_array(t):
    sa = periodic_suffix_array(t)
    if sa is not None:                  # (texts that are not of the shape are somebody else's business)
        assert sa == brute_sa(t)


def test_closed_form_is_exercised():
    """The property above is not vacuous: plain repetitions, with and without a tail, are of the shape."""
    for t in (b'ab' * 40, b'abc' * 30 + b'x', b'aab' * 25 + b'a', (b'abaab' * 30)[:-2] + b'bb', b'ba' * 35 + b'ab'):
        assert periodic_suffix_array(t) == brute_sa(t), t
