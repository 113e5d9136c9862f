"""GPU parity: pss_sa_build (HIP, through the C ABI) == the oracle's suffix
array, bit for bit (libsais from oracle/_ref when built, else the restatement)."""
import hashlib
import random

import numpy as np
import pytest

from tests.util import gen_corpus, sa_gpu

pytestmark = pytest.mark.gpu

ALPHABETS = [b'a', b'ab', b'ab\n', b'\x00', b'\x00\n', b'\x00\xff\n', bytes(range(256)), bytes(range(1, 256)),
             b'abcdefghijklmnopqrstuvwxyz0123456789 .\n']


def test_tiny_contract(oracle):
    assert sa_gpu(b'').size == 0
    assert sa_gpu(b'x').tolist() == [0]
    assert sa_gpu(b'\x00\x00').tolist() == [1, 0]
    assert sa_gpu(b'ba').tolist() == [1, 0]
    assert sa_gpu(b'ab\n').tolist() == oracle.sa(b'ab\n').tolist()


def test_random_small(oracle):
    rng = random.Random(7)
    for it in range(300):
        a = rng.choice(ALPHABETS)
        n = rng.randint(2, 300)
        s = bytes(rng.choice(a) for _ in range(n))
        got = sa_gpu(s)
        exp = oracle.sa(s)
        assert (got == exp).all(), (s, got.tolist(), exp.tolist())


@pytest.mark.parametrize('n', [4095, 4096, 4097, 8191, 65536, 100003, 1 << 20])
@pytest.mark.parametrize('alpha', [2, 4, 39, 256])
def test_random_sizes(oracle, n, alpha):
    rng = np.random.default_rng(n * 1000 + alpha)
    t = rng.integers(0, alpha, size=n, dtype=np.uint8)
    if alpha == 39:
        t = np.frombuffer(b'abcdefghijklmnopqrstuvwxyz0123456789 .\n', dtype=np.uint8)[t]
    got = sa_gpu(t)
    exp = oracle.sa(t)
    assert (got == exp).all()


@pytest.mark.parametrize('kind,n', [(0, 1 << 20), (1, 1 << 20), (2, 1 << 18), (3, 1 << 16), (3, 1 << 18)])
def test_corpora(oracle, kind, n):
    t = gen_corpus(kind, n)
    got = sa_gpu(t)
    exp = oracle.sa(t)
    assert (got == exp).all()


def test_structured(oracle):
    fib = [b'a', b'ab']
    while len(fib[-1]) < 987:
        fib.append(fib[-1] + fib[-2])
    cases = [b'a' * 1000 + b'\n', fib[-1], (b'a' * 63 + b'\n') * 64, b'\x00' * 5000, b'ab' * 5000,
             bytes(range(256)) * 40, b'\xff' * 4097 + b'\x00' * 17]
    for s in cases:
        got = sa_gpu(s)
        exp = oracle.sa(s)
        assert (got == exp).all(), s[:32]


@pytest.mark.parametrize('mode', ['dense', 'sparse', 'text', 'text1'])
@pytest.mark.parametrize('key_chars', [None, '2', '5'])
def test_forced_modes(oracle, monkeypatch, mode, key_chars):
    """Every tie-resolution mode (inverse SA doubling, hash table + key search,
    text rounds, text rounds falling back to doubling after one round) and short
    initial keys (many rounds) give the same suffix array."""
    monkeypatch.setenv('PSS_MODE', mode.rstrip('1'))
    if mode == 'text1':
        monkeypatch.setenv('PSS_TEXT_ROUNDS', '1')
    if key_chars:
        monkeypatch.setenv('PSS_KEY_CHARS', key_chars)
    rng = np.random.default_rng(11)
    cases = [gen_corpus(0, 300000), gen_corpus(1, 200000), gen_corpus(2, 70000), gen_corpus(3, 50000),
             rng.integers(0, 256, 100000, dtype=np.uint8), rng.integers(0, 2, 150000, dtype=np.uint8),
             np.frombuffer((b'abcde' * 2000 + b'\n') * 7 + b'xyz' * 100, dtype=np.uint8)]
    for t in cases:
        assert (sa_gpu(t) == oracle.sa(t)).all()


@pytest.mark.parametrize('mode', ['dense', 'sparse', 'text'])
@pytest.mark.parametrize('key_chars,drop', [('3', '1'), ('5', '2'), ('4', '3'), ('7', '5'), ('2', '7')])
def test_coarsened_last_symbol(oracle, monkeypatch, mode, key_chars, drop):
    """The sort key may leave out low bits of its last symbol (the sampled sizing does that to
    save a pass at n >= 2^24); groups then share one symbol less than the key packs.  Forced
    here at small sizes, over every alphabet width (drop >= code_bits is ignored)."""
    monkeypatch.setenv('PSS_MODE', mode)
    monkeypatch.setenv('PSS_KEY_CHARS', key_chars)
    monkeypatch.setenv('PSS_KEY_DROP', drop)
    rng = np.random.default_rng(5)
    cases = [gen_corpus(0, 250000), gen_corpus(1, 150000), gen_corpus(2, 60000), gen_corpus(3, 40000),
             rng.integers(0, 256, 90000, dtype=np.uint8), rng.integers(0, 2, 120000, dtype=np.uint8),
             rng.integers(0, 100, 120000, dtype=np.uint8)]
    for t in cases:
        assert (sa_gpu(t) == oracle.sa(t)).all()


def test_sampled_sizing_sizes(oracle):
    """n >= 2^24: the initial sort is sized from a sorted sample of suffix keys; both outcomes
    (fewer passes + coarsened key on `lines`, full-width key on `words`) against libsais."""
    for kind, n in ((0, (1 << 24) + 12345), (1, 1 << 24)):
        t = gen_corpus(kind, n)
        assert (sa_gpu(t) == oracle.sa(t)).all()
    rng = np.random.default_rng(17)
    # all 256 byte values (9-bit codes computed on the fly, 5 of them dropped from a 5-symbol key),
    # a binary alphabet (32-bit key at most) and a text whose second half repeats the first
    half = rng.integers(97, 123, 1 << 23, dtype=np.uint8)
    for t in (rng.integers(0, 256, (1 << 24) + 7, dtype=np.uint8), rng.integers(0, 2, 1 << 24, dtype=np.uint8),
              np.concatenate([half, half])):
        assert (sa_gpu(t) == oracle.sa(t)).all()


def test_all_ones_key_in_small_sort(oracle):
    """Regression (found by tests/tools/fuzz.py, seed 1279): 15 symbols -> 4-bit codes, so 16 symbols of the
    largest code pack to an all-ones 64-bit text key -- the value the one-workgroup sort pads with.
    The padding overtook real elements and the chained sort of the large groups read out of bounds."""
    import os
    t = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'regress_small_sort_padding.npy'))
    assert (sa_gpu(t) == oracle.sa(t)).all()
    crafted = np.frombuffer(b'\x0f' * 3000 + bytes(range(1, 16)) * 20 + b'\x0e' * 700 + b'\x0f' * 900, dtype=np.uint8)
    assert (sa_gpu(crafted) == oracle.sa(crafted)).all()


def test_long_repeats(oracle):
    """Duplicated blocks with small edits: LCPs in the tens of thousands (text rounds must
    hand over to rank rounds; large and small groups mixed)."""
    rng = np.random.default_rng(3)
    base = gen_corpus(1, 200000).copy()
    parts = [base]
    for k in range(5):
        c = base.copy()
        for p in rng.integers(0, c.size, 40):
            c[p] = rng.integers(97, 123)
        parts.append(c[: int(rng.integers(50000, 200000))])
    parts.append(np.frombuffer(b'the quick brown fox\n' * 9000, dtype=np.uint8))
    t = np.concatenate(parts)
    assert (sa_gpu(t) == oracle.sa(t)).all()


@pytest.mark.parametrize('n,bits', [(1, 8), (100, 64), (4096, 17), (4097, 40), (70001, 64), (1 << 20, 33)])
def test_radix_sort_pairs_stable(n, bits):
    """The device radix sort alone: stable sort of (u64 key, u32 value) pairs == numpy's stable argsort."""
    import ctypes
    import torch
    from pysubstringsearch_amd import _ffi
    rng = np.random.default_rng(n)
    mask = (1 << bits) - 1 if bits < 64 else (1 << 64) - 1
    keys = (rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)) & np.uint64(mask)
    if n > 1000:
        keys[rng.integers(0, n, n // 3)] = keys[0]          # plenty of ties
    vals = np.arange(n, dtype=np.uint32)
    dk, dv = torch.from_numpy(keys.view(np.int64)).cuda(), torch.from_numpy(vals.view(np.int32)).cuda()
    _ffi.check(_ffi.lib.pss_sort_pairs_device(dk.data_ptr(), dv.data_ptr(), n, bits, 0, None))
    order = np.argsort(keys, kind='stable')
    assert (dk.cpu().numpy().view(np.uint64) == keys[order]).all()
    if n > 8192:                                             # the one-tile bitonic path is not stable by design
        assert (dv.cpu().numpy().view(np.uint32) == vals[order]).all()


# ---- hybrid MSD initial sort (msd_sort.hip) ----

def _sa_device(host, stats=None, flags=0):
    import ctypes
    import torch
    from pysubstringsearch_amd import _ffi
    n = host.size
    dT = torch.from_numpy(host).cuda()
    dSA = torch.empty(n, dtype=torch.int32, device='cuda')
    st = _ffi.SaStats()
    _ffi.check(_ffi.lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, flags, ctypes.byref(st)))
    if stats is not None:
        stats.update(st.as_dict())
    return dSA.cpu().numpy()


@pytest.mark.parametrize('slow_local', [False, True])
def test_msd_initial_sort_forced_matches_libsais(oracle, monkeypatch, slow_local):
    """PSS_MSD=1 takes the hybrid MSD initial sort (two 10-bit partition passes over 8-byte elements +
    LDS-resident local sort) whenever the key fits; sizes around the tile (8192) and range boundaries,
    alphabets from 2 symbols to all 256 byte values, with and without the general local-sort kernel.
    The suffix array is unique, so every path must reproduce libsais' bytes."""
    monkeypatch.setenv('PSS_MSD', '1')
    if slow_local:
        monkeypatch.setenv('PSS_MSD_SLOW_LOCAL', '1')
    rng = np.random.default_rng(7)
    took = 0
    for n in (2, 3, 17, 100, 4095, 4097, 8191, 8192, 8193, 20000, 70001, 131072, 131073, 300000, (1 << 20) + 5, (1 << 21) + 77):
        for alpha in (2, 3, 4, 16, 39, 100, 256):
            if n > 300000 and alpha not in (4, 39, 256):
                continue
            t = rng.integers(0, alpha, n).astype(np.uint8)
            if alpha < 200:
                t += 40
            if rng.random() < 0.4:
                t[rng.integers(0, n, max(1, n // 50))] = 10
            st = {}
            sa = _sa_device(t, st)
            took += int(st['msd'])
            assert np.array_equal(sa, oracle.sa(t)), (n, alpha, st['msd'])
    assert took > 40        # the path really ran (it declines only when a key or a bucket does not fit)


def test_msd_crowded_bins_and_oversized_buckets(oracle, monkeypatch):
    """Inputs the fast local kernel must hand over or the whole path must decline: thousands of equal
    lines (every joint bucket one crowded bin: the general local kernel takes those tiles), and a text
    whose 20-bit prefixes are so skewed that a bucket exceeds a tile (the LSD passes run instead)."""
    monkeypatch.setenv('PSS_MSD', '1')
    monkeypatch.setenv('PSS_PERIOD', '0')          # (one line repeated: the closed form would take it from the sort under test)
    rng = np.random.default_rng(11)
    line = bytes(rng.integers(97, 123, 200).astype(np.uint8)) + b'\n'
    t = np.frombuffer(line * 3000, dtype=np.uint8).copy()              # 3000 copies: buckets of <= 3000 equal keys
    st = {}
    sa = _sa_device(t, st)
    assert np.array_equal(sa, oracle.sa(t))
    assert st['msd'] == 1 and st['msd_slow_tiles'] > 0
    mix = np.concatenate([rng.integers(97, 123, 1 << 20).astype(np.uint8), np.full(1 << 18, 97, np.uint8),
                          rng.integers(97, 123, 1 << 18).astype(np.uint8)])
    st = {}
    sa = _sa_device(mix, st)
    assert np.array_equal(sa, oracle.sa(mix))
    assert st['msd'] == 0 and st['msd_max_bucket'] > 4088               # declined after the exact bucket count


def test_msd_is_chosen_for_high_entropy_text_only(oracle):
    """Without the switch the sorted key sample decides: the `lines` corpus (uniform symbols) takes the
    MSD path at 2^24, natural-text-like `words` (crowded prefixes) keeps the LSD passes."""
    from tests.util import gen_corpus
    st = {}
    t = gen_corpus(0, 1 << 24)
    sa = _sa_device(t, st)
    assert st['msd'] == 1 and st['msd_max_bucket'] <= 4088 and st['msd_slow_tiles'] == 0
    assert hashlib.sha256(sa.tobytes()).hexdigest() == hashlib.sha256(oracle.sa(t).tobytes()).hexdigest()
    st = {}
    t = gen_corpus(1, 1 << 24)
    sa = _sa_device(t, st)
    assert st['msd'] == 0
    assert hashlib.sha256(sa.tobytes()).hexdigest() == hashlib.sha256(oracle.sa(t).tobytes()).hexdigest()


# ---- sample sort over 16-byte elements (ss_sort_impl.h) ----

@pytest.mark.parametrize('local', ['buckets', 'tile', 'window'])
def test_sample_sort_forced_matches_libsais(oracle, monkeypatch, local):
    """PSS_SS=1 sends every text of >= 2^16 bytes through the sample sort (splitters from a sorted sample, two partition
    passes over 16-byte [key | index] elements, merge sort of every tile in LDS, ties as flags + the tile-boundary pass):
    alphabets of 2 .. 256 symbols (keys of 11 .. 32 symbols), word-like repeats, long duplicated blocks (tie groups that
    cross buckets and tiles), both bucket-count shapes (B1 = B2 and B1 = 2 B2), every tie-resolution mode afterwards.
    The local sort in its three forms: bucket by bucket on a plan of padded lengths (round 5, the default), the whole
    tile as one array (PSS_SS_SEG=0), and that on the window plan (PSS_SS_WINDOW_PLAN)."""
    monkeypatch.setenv('PSS_SS', '1')
    monkeypatch.setenv('PSS_MSD', '0')
    if local == 'tile':
        monkeypatch.setenv('PSS_SS_SEG', '0')
    elif local == 'window':
        monkeypatch.setenv('PSS_SS_WINDOW_PLAN', '1')
    rng = np.random.default_rng(3)
    took = 0
    for trial, n in enumerate((1 << 16, 70001, 100003, 300000, 1 << 20, (1 << 21) + 77, (1 << 22) + 5, 1 << 16, 200001, 1 << 19)):
        alpha = (2, 3, 4, 16, 27, 39, 100, 255, 256, 27)[trial]
        t = (rng.integers(0, alpha, n).astype(np.uint16) + (0 if alpha > 200 else 40)).astype(np.uint8)
        if trial % 3 == 1:
            words = [bytes(rng.integers(97, 97 + min(alpha, 26), int(rng.integers(2, 9))).astype(np.uint8)) for _ in range(50)]
            t = np.frombuffer(b' '.join(words[int(i)] for i in rng.integers(0, 50, n // 4)), dtype=np.uint8)[:n].copy()
        elif trial % 3 == 2:
            blk = t[:5000].copy()
            for o in rng.integers(0, n - 5000, 20):
                t[o:o + 5000] = blk
        t[-1] = 10
        if trial % 4 == 3:
            monkeypatch.setenv('PSS_MODE', ('dense', 'text')[trial % 2])
        else:
            monkeypatch.delenv('PSS_MODE', raising=False)
        st = {}
        sa = _sa_device(t, st)
        took += st['ss']
        assert st['ss'] == 1 and st['ss_max_bucket'] <= 4088, (trial, st['ss'], st['ss_max_bucket'])
        assert np.array_equal(sa, oracle.sa(t)), (trial, n, alpha)
    assert took == 10


def test_sample_sort_is_chosen_for_natural_text(oracle):
    """Without switches at n >= 2^24: `words` (some 20-bit prefix far beyond a tile) takes the sample sort, `lines` the
    radix MSD sort; both give libsais' bytes."""
    from tests.util import gen_corpus
    for kind, want_ss, want_msd in ((1, 1, 0), (0, 0, 1)):
        t = gen_corpus(kind, 1 << 24)
        st = {}
        sa = _sa_device(t, st)
        assert (st['ss'], st['msd']) == (want_ss, want_msd), (kind, st['ss'], st['msd'])
        assert hashlib.sha256(sa.tobytes()).hexdigest() == hashlib.sha256(oracle.sa(t).tobytes()).hexdigest()


@pytest.mark.parametrize('front', [True, False])
def test_initial_sort_plan_is_reused_and_checked(oracle, monkeypatch, front):
    """The second build of the same kind of text (same size class, every switch at its default) goes straight to the MSD
    sort with the remembered alphabet: no alphabet pass, no recode pass, no sizing sample -- the sort's first histogram
    pass recodes the raw text and checks that every byte has a code (plan_hint = 2; with PSS_NO_PLAN_FRONT=1 the
    alphabet is looked at first and only the sample is skipped: plan_hint = 1).  Same bytes out.  A text with the same
    byte values but natural text's distribution gets the plan too; the MSD sort's exact bucket check refuses it and the
    build goes on (plan_hint 1) or starts over (2) -- still libsais' bytes -- and forgets the plan.  So does a text with
    a byte the plan has no code for, and a text over a subset of the remembered alphabet is sorted with the larger table.
    PSS_NO_PLAN_CACHE=1 switches the memory off."""
    if not front:
        monkeypatch.setenv('PSS_NO_PLAN_FRONT', '1')
    n = 1 << 24
    lines = gen_corpus(0, n)
    want = hashlib.sha256(oracle.sa(lines).tobytes()).hexdigest()
    st = {}
    _sa_device(lines, st, flags=8)       # (flags bit 3: a cold build, whatever earlier tests left on the device)
    # a first chunk whose symbols are close to uniform goes to the MSD sort on its symbol counts alone (plan_hint 3)
    assert (st['plan_hint'], st['msd']) == (3 if front else 0, 1)
    st = {}
    sa = _sa_device(lines, st)
    assert (st['plan_hint'], st['msd']) == (2 if front else 1, 1)
    assert hashlib.sha256(sa.tobytes()).hexdigest() == want
    if front:
        # an unaligned text pointer keeps the separate passes
        import torch
        from pysubstringsearch_amd import _ffi
        import ctypes
        buf = torch.from_numpy(np.concatenate([np.zeros(3, np.uint8), lines])).cuda()
        dSA = torch.empty(n, dtype=torch.int32, device='cuda')
        s2 = _ffi.SaStats()
        _ffi.check(_ffi.lib.pss_sa_build_device(buf.data_ptr() + 3, dSA.data_ptr(), n, 0, 0, ctypes.byref(s2)))
        assert s2.plan_hint == 1 and hashlib.sha256(dSA.cpu().numpy().tobytes()).hexdigest() == want
        # a byte the remembered alphabet does not have ('~' at one place): refused inside the first pass, rebuilt from scratch
        odd = lines.copy()
        odd[n // 3] = 126
        assert 126 not in np.unique(lines)
        st = {}
        sa = _sa_device(odd, st)
        assert (st['plan_hint'], st['msd']) == (3, 1)           # (rebuilt without the plan: a first chunk again)
        assert hashlib.sha256(sa.tobytes()).hexdigest() == hashlib.sha256(oracle.sa(odd).tobytes()).hexdigest()
        # ... which is the plan now; the old text is over a subset of that alphabet and takes the larger table
        st = {}
        sa = _sa_device(lines, st)
        assert (st['plan_hint'], st['msd'], st['sigma']) == (2, 1, len(np.unique(odd)))
        assert hashlib.sha256(sa.tobytes()).hexdigest() == want
    # same alphabet, crowded prefixes: a small vocabulary over the bytes of `lines`
    rng = np.random.default_rng(9)
    alphabet = np.unique(lines)
    vocab = [alphabet[rng.integers(0, len(alphabet), int(rng.integers(3, 9)))] for _ in range(300)]
    vocab.append(alphabet)                                   # every byte value occurs
    picks = rng.integers(0, len(vocab), n // 4)
    crowded = np.concatenate([vocab[-1]] + [vocab[i] for i in picks])[:n].copy()
    assert len(crowded) == n and np.array_equal(np.unique(crowded), alphabet)
    st = {}
    sa = _sa_device(crowded, st)
    if front:
        assert (st['plan_hint'], st['msd'], st['ss']) == (0, 0, 1)                                 # planned, refused, rebuilt
    else:
        assert (st['plan_hint'], st['msd'], st['ss']) == (1, 0, 1) and st['msd_max_bucket'] > 4088    # hinted, refused
    assert hashlib.sha256(sa.tobytes()).hexdigest() == hashlib.sha256(oracle.sa(crowded).tobytes()).hexdigest()
    st = {}
    _sa_device(lines, st)
    assert (st['plan_hint'], st['msd']) == (3 if front else 0, 1)      # the refusal cleared the plan
    _sa_device(lines, {})
    monkeypatch.setenv('PSS_NO_PLAN_CACHE', '1')              # (the build reads its switches on every call)
    st = {}
    sa = _sa_device(lines, st)
    assert (st['plan_hint'], st['msd']) == (0, 1)
    assert hashlib.sha256(sa.tobytes()).hexdigest() == want


def test_sample_sort_plan_is_reused_across_chunks(oracle, monkeypatch):
    """Natural text: the sorted sample of one chunk cuts the next chunks of the corpus (same sample size, bucket counts,
    index bits and alphabet -- the chunks of one Writer differ by an entry or two in length) -- no sizing sample, no sample
    of their own (ss_planned = 1) -- and the suffix array is libsais' all the same: any splitters give the exact result,
    the previous chunk's only have to keep the buckets inside a tile, which the exact bucket check confirms on every
    build.  A chunk of another size class, a cold build (flags bit 3) and PSS_NO_PLAN_CACHE=1 draw their own sample; a
    chunk whose equal keys sit where no remembered splitter separates them starts over without the plan
    (ss_plan_refused) and the plan rests for a build."""
    n = (1 << 25) - 4096
    a, b_ = gen_corpus(1, n, 0), gen_corpus(1, n, 1)
    assert np.array_equal(np.unique(a), np.unique(b_)) and not np.array_equal(a, b_)

    def check(t, st):
        sa = _sa_device(t, st)
        assert hashlib.sha256(sa.tobytes()).hexdigest() == hashlib.sha256(oracle.sa(t).tobytes()).hexdigest()

    st = {}
    _sa_device(a, st, flags=8)
    assert (st['plan_hint'], st['ss'], st['ss_planned']) == (0, 1, 0)
    for t in (b_, a):
        st = {}
        check(t, st)
        assert (st['plan_hint'], st['ss'], st['ss_planned'], st['msd']) == (2, 1, 1, 0)
    # a cold build forgets, and leaves a new plan
    st = {}
    _sa_device(b_, st, flags=8)
    assert (st['plan_hint'], st['ss_planned']) == (0, 0)
    # a chunk a few entries shorter: the same geometry, the plan holds (splitters whose index lies beyond this chunk's end
    # are numbers like any other)
    c = gen_corpus(1, n - 70001, 2)
    st = {}
    check(c, st)
    assert (st['ss'], st['ss_planned'], st['ss_plan_refused']) == (1, 1, 0)
    # another size class: own sample
    c = gen_corpus(1, (1 << 25) + 4096, 2)
    st = {}
    check(c, st)
    assert (st['ss'], st['ss_planned']) == (1, 0)
    # equal keys where no remembered splitter separates them (300 000 times one byte in the middle of the chunk): refused,
    # the build starts over with a sample of its own; the next chunk does not try the plan at once, the one after does
    _sa_device(a, {}, flags=8)
    odd = a.copy()
    odd[1 << 23:(1 << 23) + 300000] = a[0]
    st = {}
    check(odd, st)
    assert (st['ss'], st['ss_planned'], st['ss_plan_refused']) == (1, 0, 1) and st['ms_restarts'] > 0, (st['ss_planned'], st['ss_plan_refused'])
    st = {}
    _sa_device(b_, st)
    assert (st['plan_hint'], st['ss_planned'], st['ss_plan_refused']) == (2, 0, 0)
    st = {}
    check(a, st)
    assert (st['ss_planned'], st['ss_plan_refused']) == (1, 0)
    monkeypatch.setenv('PSS_NO_PLAN_CACHE', '1')
    _sa_device(a, {})
    st = {}
    _sa_device(b_, st)
    assert (st['plan_hint'], st['ss'], st['ss_planned']) == (0, 1, 0)


def test_sample_sort_samples_the_whole_text(oracle):
    """Regression (tests/tools/real_text.py, 412 MB of real source files): the sample took one position from every
    stride of floor(n / S) bytes, so the last n mod S positions of a text whose length is no multiple of the sample size
    were never sampled; a run of equal bytes there -- 18 432 of them in that chunk -- has equal keys and consecutive
    indices, landed in one bucket beyond a tile, and the sample sort declined (the build fell back to the LSD passes
    with keys of 8 symbols instead of 12: 168 ms instead of 132).  The strata now cover the text to its last byte."""
    n = (1 << 24) + 70000                     # S = 131 072: floor(n / S) = 128 leaves the last 70 000 bytes out
    t = gen_corpus(1, n)
    t[n - 50000:n - 20000] = ord('q')         # a run longer than 7 tiles where no sample used to fall
    st = {}
    sa = _sa_device(t, st, flags=8)
    assert st['ss'] == 1 and st['ss_max_bucket'] <= 4088, (st['ss'], st['ss_max_bucket'])
    assert hashlib.sha256(sa.tobytes()).hexdigest() == hashlib.sha256(oracle.sa(t).tobytes()).hexdigest()


def test_real_files_chunk(oracle):
    """Real files instead of generated text: 20 MB of the Python sources, headers and documentation found on the
    machine, in sorted path order (tests/tools/real_text.py) -- licence headers copied hundreds of times, rules of
    dashes, indentation runs, bytes above 127, 200-odd byte values.  The suffix array is libsais'; the build takes the
    sample sort, a text round and the anchor round, like the 412 MB chunk of profiles/r04_real_files.txt."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('real_text', os.path.join(os.path.dirname(__file__), 'tools', 'real_text.py'))
    rt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rt)
    raw = rt.collect(20 << 20)
    if len(raw) < (17 << 20):
        pytest.skip('not enough text files on this machine')
    t = np.frombuffer(raw, dtype=np.uint8).copy()
    t[-1] = 10
    st = {}
    sa = _sa_device(t, st, flags=8)
    assert hashlib.sha256(sa.tobytes()).hexdigest() == hashlib.sha256(oracle.sa(t).tobytes()).hexdigest()
    assert st['anchor_left'] == 0            # (which sorts the build took depends on the files; on this image: sample sort + anchors)


# ---- texts that repeat one word (rle_build.hip, periodic prefix) ----

def _periodic(word, n, tail=b''):
    body = (word * (n // len(word) + 2))[:n - len(tail)]
    return np.frombuffer(body + tail, dtype=np.uint8).copy()


@pytest.mark.parametrize('case', range(15))
def test_periodic_text_closed_form(oracle, monkeypatch, case):
    """A text that repeats one word of 2 .. 1024 bytes (with up to 1024 other bytes behind the repetition) gets its
    suffix array in closed form -- rotation blocks, ascending or descending by the byte that ends the repetition, the
    last suffixes sorted on the host -- and it is libsais' for: words whose rotations share long prefixes, tails that
    sort before / after the word's continuation, tails that look like the word for a while, the pure repetition (no
    tail), repetitions cut mid-word, the longest word and tail the path takes, and shapes it must decline (the
    repetition ends early; the word is longer than 1024)."""
    rng = np.random.default_rng(case)
    n = 1 << 17
    taken = True
    if case == 0:
        t = _periodic(b'ab', n)
    elif case == 1:
        t = _periodic(b'abc', n + 1, b'\n')
    elif case == 2:
        t = _periodic(b'aab' * 5 + b'aac', n, b'a')                       # tail byte < continuation
    elif case == 3:
        t = _periodic(b'aab' * 5 + b'aac', n, b'z')                       # tail byte > continuation
    elif case == 4:
        w = bytes(rng.integers(97, 100, 40).astype(np.uint8))
        t = _periodic(w, n - 3, b'\n')
    elif case == 5:
        w = bytes(rng.integers(97, 123, 1024).astype(np.uint8))           # longest word
        t = _periodic(w, n, bytes(rng.integers(97, 123, 1024).astype(np.uint8)))      # longest tail
    elif case == 6:
        w = b'the quick brown fox jumps over the lazy dog\n'
        t = _periodic(w, n, w[5:17] + b'X')                                # tail = the word again, from elsewhere, then a break
    elif case == 7:
        w = b'xyxyxz'
        t = _periodic(w, n, b'xyxyxy')                                     # tail continues a rotation's prefix
    elif case == 8:
        t = _periodic(bytes([0, 255, 0, 7]), n, bytes([255, 0]))
    elif case == 9:
        t = _periodic(b'abcabd', 40000)                                    # small text
    elif case == 10:
        t = gen_corpus(4, 1 << 20)                                         # `repeat_line`: the generator's last byte breaks the word
    elif case == 11:
        w = bytes(rng.integers(97, 123, 1025).astype(np.uint8))           # word too long: declined
        t, taken = _periodic(w, n), False
    elif case == 12:
        t = _periodic(b'abcde', n)
        t[n // 2] = ord('z')                                               # the repetition ends half way: declined
        taken = False
    elif case == 14:
        monkeypatch.setenv('PSS_RLE', '0')                                 # one byte repeated, the run-length path switched off:
        t, taken = _periodic(b'a', 40000), False                           # "aa" is not a word (period 1): declined
    else:
        t = _periodic(b'ab', n, b'a' + bytes(rng.integers(97, 99, 1024).astype(np.uint8)))   # ('a' where 'b' was due) tail too long: declined
        taken = False
    st = {}
    sa = _sa_device(t, st)
    assert st['period_path'] == (1 if taken else 0), (case, st['period'], st['period_extent'])
    assert hashlib.sha256(sa.tobytes()).hexdigest() == hashlib.sha256(oracle.sa(t).tobytes()).hexdigest(), case


def test_periodic_text_random_words_and_tails(oracle):
    """Eighty random shapes of the same: word length 2 .. 300 over 2 .. 4 letters (rotations with long common prefixes),
    tails of 0 .. 200 bytes over the same letters (so they often continue the word for a while), lengths that cut the
    word anywhere."""
    rng = np.random.default_rng(2024)
    took = 0
    for _ in range(80):
        sigma = int(rng.integers(2, 5))
        w = bytes(rng.integers(97, 97 + sigma, int(rng.integers(2, 301))).astype(np.uint8))
        tail = bytes(rng.integers(97, 97 + sigma, int(rng.integers(0, 201))).astype(np.uint8))
        n = int(rng.integers(4 * 8192, 60000))
        t = _periodic(w, n, tail)
        st = {}
        sa = _sa_device(t, st)
        took += st['period_path']
        assert np.array_equal(sa, oracle.sa(t)), (w, tail, n, st['period'], st['period_extent'])
    assert took >= 60          # (a random word that is itself a repetition has a shorter period; a few are declined for size)


# ---- run-length path (rle_build.hip) ----

def _runs_text(rng, n, alpha, maxrun, base=40):
    out = np.empty(n + maxrun, np.uint8)
    o = 0
    while o < n:
        length = int(rng.integers(1, maxrun + 1))
        out[o:o + length] = base + int(rng.integers(0, alpha))
        o += length
    return out[:n].copy()


@pytest.mark.parametrize('expansion', ['columns', 'sort'])
def test_rle_path_forced_matches_libsais(oracle, monkeypatch, expansion):
    """PSS_RLE=1 sends EVERY text through the run-length path (run table -> one symbol per run -> suffix
    sort of the reduced string by rank rounds -> stable radix sort of all suffixes by (class, remaining
    run length)): texts without runs, one single run, runs of every length around the tile and sort
    thresholds, periodic texts (the reduced string is periodic too), byte values 0 and 255, and the
    end-of-text cases (last run of either type).  The expansion is the matrix walk where its table fits
    (stats rle == 2) and the stable radix sort otherwise or with PSS_RLE_SORT.  The result must be
    libsais' bytes."""
    monkeypatch.setenv('PSS_RLE', '1')
    if expansion == 'sort':
        monkeypatch.setenv('PSS_RLE_SORT', '1')
    walked = 0
    rng = np.random.default_rng(5)
    cases = []
    for n in (2, 3, 5, 17, 100, 4095, 4096, 4097, 8192, 20000, 70001, 300000):
        cases.append(rng.integers(0, 3, n).astype(np.uint8) + 97)                      # hardly any runs
        cases.append(np.full(n, 97, np.uint8))                                         # one run
        cases.append(_runs_text(rng, n, 2, 9))
        cases.append(_runs_text(rng, n, 3, 5000))
        t = np.full(n, 97, np.uint8)
        t[15::16] = 10                                                                 # periodic, period 16
        cases.append(t)
        t = _runs_text(rng, n, 200, 40, base=0)
        t[-1] = 255
        cases.append(t)
        unit = _runs_text(rng, 50, 2, 20)
        cases.append(np.tile(unit, n // 50 + 1)[:n].copy())                            # a periodic pattern of runs
        t = _runs_text(rng, n, 2, 300)
        t[-1] = 10                                                                     # the reference's texts end in a newline
        cases.append(t)
    cases.append(_runs_text(rng, (1 << 21) + 77, 2, 8192))
    for t in cases:
        st = {}
        sa = _sa_device(t, st)
        assert st['rle'] in (1, 2)
        walked += st['rle'] == 2
        assert np.array_equal(sa, oracle.sa(t)), (t.size, st['runs'])
    assert (walked > 20) if expansion == 'columns' else (walked == 0)


@pytest.mark.parametrize('rle', ['0', '1'])
def test_unaligned_text_pointer(oracle, monkeypatch, rle):
    """The text pointer handed to pss_sa_build_device need not be 16-byte aligned and n need not be a multiple of 16
    (a chunk inside a larger device buffer): the run count of sa_symbols (shuffled 16-byte vectors) and the run
    table of rle_count / rle_starts must agree -- the build compares them -- and the result is libsais'."""
    import ctypes
    import torch
    from pysubstringsearch_amd import _ffi
    monkeypatch.setenv('PSS_RLE', rle)
    rng = np.random.default_rng(11)
    for off, n in ((1, 100003), (3, 65537), (7, 4099), (8, 300001), (15, 1 << 20)):
        t = _runs_text(rng, n, 3, 200) if rle == '1' else rng.integers(97, 101, n).astype(np.uint8)
        t[-1] = 10
        big = torch.zeros(n + 64, dtype=torch.uint8, device='cuda')
        big[off:off + n] = torch.from_numpy(t).cuda()
        dSA = torch.empty(n, dtype=torch.int32, device='cuda')
        st = _ffi.SaStats()
        _ffi.check(_ffi.lib.pss_sa_build_device(big.data_ptr() + off, dSA.data_ptr(), n, 0, 0, ctypes.byref(st)))
        assert bool(st.rle) == (rle == '1')
        assert np.array_equal(dSA.cpu().numpy(), oracle.sa(t)), (off, n)


def test_rle_path_is_chosen_for_long_runs_only(oracle, monkeypatch):
    """Without the switch: runs averaging >= 8 bytes take the path (`runs`, `periodic`, a zero-padded
    binary-like text), anything else (lines, words, short runs) keeps the key sort + rounds; PSS_RLE=0
    turns it off and the doubling rounds give the same bytes."""
    from tests.util import gen_corpus
    for kind, want in ((0, 0), (1, 0), (2, 1), (3, 1)):
        st = {}
        t = gen_corpus(kind, 1 << 22)
        sa = _sa_device(t, st)
        assert (st['rle'] > 0) == bool(want), kind
        assert hashlib.sha256(sa.tobytes()).hexdigest() == hashlib.sha256(oracle.sa(t).tobytes()).hexdigest()
    rng = np.random.default_rng(9)
    t = rng.integers(1, 256, 1 << 20).astype(np.uint8)
    for o in rng.integers(0, t.size - 30000, 10):
        t[o:o + int(rng.integers(1000, 20000))] = 0                                    # zero padding inside random bytes
    st = {}
    sa = _sa_device(t, st)
    assert st['rle'] == 0 and st['runs'] * 8 > t.size                                 # runs are long but few: not worth it
    assert np.array_equal(sa, oracle.sa(t))
    t = _runs_text(rng, 1 << 20, 4, 64)
    st = {}
    sa = _sa_device(t, st)
    assert st['rle'] == 2
    assert np.array_equal(sa, oracle.sa(t))
    monkeypatch.setenv('PSS_RLE', '0')
    st = {}
    sa0 = _sa_device(t, st)
    assert st['rle'] == 0 and np.array_equal(sa0, sa)


def _repeat_cases(rng, nrng):
    """Texts whose ties outlive the text rounds: copies of blocks (with and without edits), periodic stretches inside
    other text, two stretches of the same word with different ends, copies mixed with runs of one byte."""
    out = []
    for alpha, n in ((2, 70000), (4, 200000), (26, 300000), (39, 1 << 20), (200, 150000), (256, 400000)):
        base = nrng.integers(0, alpha, n, dtype=np.uint8)
        if alpha == 26:
            base = base + np.uint8(97)
        blk = int(rng.choice([300, 4000, 50000]))
        t = np.resize(base[:blk], n).copy()                      # one block repeated ...
        for p in nrng.integers(0, n, 25):                        # ... with sparse edits
            t[p] = base[p]
        out.append(t)
        t = base.copy()                                          # a periodic stretch inside random text
        a, b = sorted(int(x) for x in nrng.integers(0, n, 2))
        t[a:b] = np.resize(base[:int(rng.choice([2, 7, 60, 333]))], b - a)
        out.append(t)
        t = base.copy()                                          # the same word twice, different lengths and ends
        q = n // 4
        word = base[:int(rng.choice([5, 13, 64]))]
        t[q:2 * q] = np.resize(word, q)
        t[3 * q:3 * q + q // 2] = np.resize(word, q // 2)
        out.append(t)
        t = base.copy()                                          # blocks copied from elsewhere + runs
        for _ in range(12):
            s_, d_ = (int(x) for x in nrng.integers(0, n - 5000, 2))
            t[d_:d_ + 5000] = t[s_:s_ + 5000].copy()
        for _ in range(10):
            s_ = int(nrng.integers(0, n - 700))
            t[s_:s_ + int(nrng.integers(20, 700))] = t[s_]
        out.append(t)
    return out


@pytest.mark.parametrize('omega', [None, '9', '17', '40'])
@pytest.mark.parametrize('count_sort', [False, True])
def test_anchor_round_forced(oracle, monkeypatch, omega, count_sort):
    """PSS_ANCHOR=1: whenever ties outlive the text rounds, ONE round keyed by the ranks of the anchors (minimizers of
    the text, anchor_impl.h) finishes the suffix array -- at any size, with the window forced narrower than the known
    common prefix allows, with either group sort of the rank rounds.  Nothing may be left tied by that round."""
    import ctypes
    import random

    import torch

    from pysubstringsearch_amd import _ffi
    monkeypatch.setenv('PSS_ANCHOR', '1')
    if omega:
        monkeypatch.setenv('PSS_ANCHOR_OMEGA', omega)
    if count_sort:
        monkeypatch.setenv('PSS_COUNT_SORT', '1')
    rng = random.Random(5)
    nrng = np.random.default_rng(5)
    took = 0
    for t in _repeat_cases(rng, nrng) + [gen_corpus(6, 1 << 20), gen_corpus(5, 1 << 21), gen_corpus(6, 3 << 20, 2)]:
        t = np.ascontiguousarray(t)
        dT = torch.from_numpy(t).cuda()
        dSA = torch.empty(t.size, dtype=torch.int32, device='cuda')
        st = _ffi.SaStats()
        _ffi.check(_ffi.lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), t.size, 0, 0, ctypes.byref(st)))
        assert st.anchor_left == 0
        took += int(st.anchor)
        assert (dSA.cpu().numpy() == oracle.sa(t)).all()
    assert took >= (10 if omega != "9" else 4)      # (a window of 9 chooses about n / 5 anchors: the limit at which the path declines)


@pytest.mark.parametrize('omega', [None, '12'])
def test_anchors_sorted_beside_the_text_round(oracle, monkeypatch, omega):
    """PSS_ANCHOR_SIDE=1 (round 5, SideAnchors in sa_build.hip): the anchors are selected and sorted by a second host
    thread on a second stream, in the helper context, while the main line runs the text round that precedes the anchor
    round -- at any size here (by default: texts of >= 2^24 bytes whose sampled ties show copies).  Same suffix arrays;
    the side line's keys are used when the text round reached the depth they were made for (anchor_side = 1) and thrown
    away when it gave up half-way (2); a second build of the same text starts its side line beside the initial sort
    (the plan the first one left) and must agree as well."""
    import ctypes
    import random

    import torch

    from pysubstringsearch_amd import _ffi
    monkeypatch.setenv('PSS_ANCHOR', '1')
    monkeypatch.setenv('PSS_ANCHOR_SIDE', '1')
    if omega:
        monkeypatch.setenv('PSS_ANCHOR_OMEGA', omega)
    rng = random.Random(7)
    nrng = np.random.default_rng(7)
    used = thrown = 0
    cases = _repeat_cases(rng, nrng)[:14] + [gen_corpus(6, 1 << 20), gen_corpus(7, 1 << 21), gen_corpus(7, 3 << 20, 1), gen_corpus(5, 1 << 21)]
    for t in cases:
        t = np.ascontiguousarray(t)
        want = oracle.sa(t)
        dT = torch.from_numpy(t).cuda()
        dSA = torch.empty(t.size, dtype=torch.int32, device='cuda')
        st = _ffi.SaStats()
        for again in range(2):
            dSA.zero_()
            _ffi.check(_ffi.lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), t.size, 0, 8 if again == 0 else 0, ctypes.byref(st)))
            assert st.anchor_left == 0
            assert (dSA.cpu().numpy() == want).all(), (t.size, again, st.anchor_side)
            used += int(st.anchor_side == 1)
            thrown += int(st.anchor_side == 2)
    assert used >= 6, (used, thrown)
    # the default: small texts and natural text never start a side line
    monkeypatch.delenv('PSS_ANCHOR_SIDE')
    monkeypatch.delenv('PSS_ANCHOR')
    for kind, n in ((1, 1 << 22), (7, 1 << 21)):
        t = gen_corpus(kind, n)
        dT = torch.from_numpy(t).cuda()
        dSA = torch.empty(n, dtype=torch.int32, device='cuda')
        st = _ffi.SaStats()
        _ffi.check(_ffi.lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, 8, ctypes.byref(st)))
        assert st.anchor_side == 0 and (dSA.cpu().numpy() == oracle.sa(t)).all()


@pytest.mark.parametrize('merge', ['1', '2', None])
def test_large_groups_take_the_segmented_merge_sort(oracle, monkeypatch, merge):
    """Groups of more than 4096 tied suffixes (bg_*_kernel in sa_build.hip, round 5): every 4096-element tile of a group
    sorted in LDS, then merge passes inside the group -- for every round (PSS_BIG_MERGE=1), for text rounds only (2), and
    the chained radix sorts that stayed the default (the merge sort measured no faster at full size) -- against libsais.  Texts made of a few lines written thousands of times in random order:
    every position of a line is a group of as many suffixes as the line has copies -- 5 000 (one merge pass), 20 000
    (three), 70 000 (five), with equal and unequal keys inside them, partial tiles at the ends, in text rounds (PSS_ANCHOR=0
    keeps the rank rounds going instead of the anchor round) and rank rounds."""
    import ctypes

    import torch

    from pysubstringsearch_amd import _ffi
    if merge is not None:                      # (None: the default -- the chained radix sorts)
        monkeypatch.setenv('PSS_BIG_MERGE', merge)
    rng = np.random.default_rng(23)
    cases = []
    for nlines, copies, width in ((37, 5000, 24), (9, 20011, 17), (3, 70003, 11), (120, 4500, 9)):
        pool = [bytes(rng.integers(97, 101, int(rng.integers(3, width)), dtype=np.uint8)) + b'\n' for _ in range(nlines)]
        order = rng.integers(0, nlines, nlines * copies)
        cases.append(np.frombuffer(b''.join(pool[int(i)] for i in order), dtype=np.uint8).copy())
    big_seen = 0
    for anchor in ('0', None):
        if anchor is None:
            monkeypatch.delenv('PSS_ANCHOR', raising=False)
        else:
            monkeypatch.setenv('PSS_ANCHOR', anchor)
        for t in cases:
            want = oracle.sa(t)
            dT = torch.from_numpy(t).cuda()
            dSA = torch.empty(t.size, dtype=torch.int32, device='cuda')
            st = _ffi.SaStats()
            _ffi.check(_ffi.lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), t.size, 0, 8, ctypes.byref(st)))
            assert (dSA.cpu().numpy() == want).all(), (t.size, anchor, merge)
            big_seen += int(st.big_elems)
    assert big_seen > 1000000


def test_anchor_round_is_chosen_for_long_repeats_only(oracle):
    """Default switches: texts of >= 2^20 bytes whose ties outlive the text rounds take the anchor round; natural text
    and high-entropy lines do not."""
    import ctypes

    import torch

    from pysubstringsearch_amd import _ffi
    for kind, n, want in ((5, 1 << 22, 1), (6, 1 << 22, 1), (1, 1 << 22, 0), (0, 1 << 22, 0), (5, 1 << 18, 0)):
        t = gen_corpus(kind, n)
        dT = torch.from_numpy(t).cuda()
        dSA = torch.empty(n, dtype=torch.int32, device='cuda')
        st = _ffi.SaStats()
        _ffi.check(_ffi.lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, 0, ctypes.byref(st)))
        assert int(st.anchor) == want and st.anchor_left == 0, (kind, n, st.anchor)
        assert (dSA.cpu().numpy() == oracle.sa(t)).all()


def _periodic_stretch_cases():
    """Texts whose large tied groups are the phases of periodic runs: several runs of one word -- equal lengths among
    them (ties on type and extent: the rank at the break decides), breaks above and below the periodic symbol, one run
    that ends with the text, runs of another word with the same first symbols -- inside random text."""
    rng = np.random.default_rng(12)
    out = []
    for alpha, word_len, run_lens, n in ((4, 3, (30000, 30000, 12000, 47000), 300000), (26, 7, (50000, 20000, 20000), 250000),
                                         (2, 5, (40000, 25000, 40000), 200000), (200, 60, (90000, 90000), 400000),
                                         (3, 1, (20000, 20000, 35000), 150000), (39, 12, (80000,), 200000)):
        t = rng.integers(0, alpha, n, dtype=np.uint8) + (97 if alpha <= 26 else 0)
        word = t[:word_len].copy()
        at = 1000
        for k, L in enumerate(run_lens):
            t[at:at + L] = np.resize(np.roll(word, -k), L)
            # the symbol that breaks the period: once below, once above the periodic one, then whatever the text holds
            if k < 2:
                per = int(t[at + L - word_len])
                t[at + L] = per - 1 if (k == 0 and per > int(t.min())) else min(255, per + 1)
            at += L + int(rng.integers(1, 5000))
        tail = min(n // 6, 25000)
        t[n - tail:] = np.resize(word, tail)                       # periodic up to the end of the text
        other = word.copy()
        other[-1] = word[-1] + 1 if word[-1] < 255 else word[-1] - 1
        t[at:at + 9000] = np.resize(other, 9000)                   # another word with the same first symbols
        out.append(np.ascontiguousarray(t))
    return out


@pytest.mark.parametrize('anchor', ['0', '1'])
def test_periodic_runs_inside_rank_rounds(oracle, monkeypatch, anchor):
    """Rank rounds (of the text with PSS_ANCHOR=0, of the anchors' names with 1): large groups whose members are the
    phases of periodic runs take the key (type of the break, how far the period goes on, rank at the break) and come
    apart in that one round (`periodic_rounds`, `periodic_members`) -- libsais' order, the same as with PSS_PERIODIC=0,
    which takes log2(run / depth) rounds over them."""
    monkeypatch.setenv('PSS_ANCHOR', anchor)
    monkeypatch.setenv('PSS_PERIOD', '0')      # (no closed form for a text that starts periodic: the rounds are under test)
    monkeypatch.setenv('PSS_RLE', '0')
    took = 0
    for t in _periodic_stretch_cases():
        want = oracle.sa(t)
        st = {}
        got = _sa_device(t, st)
        assert (got == want).all()
        assert st['anchor_left'] == 0
        took += int(st['periodic_rounds'] > 0)
        monkeypatch.setenv('PSS_PERIODIC', '0')
        st0 = {}
        got0 = _sa_device(t, st0)
        monkeypatch.delenv('PSS_PERIODIC')
        assert (got0 == want).all() and st0['periodic_rounds'] == 0
        if st['periodic_rounds'] and anchor == '0':
            assert st['rounds'] <= st0['rounds']
    assert took >= 4


def test_anchor_round_on_a_text_with_fewer_anchors_than_symbols_per_key(oracle, monkeypatch):
    """Regression (tests/tools/fuzz.py, seed 4512): 891 bytes over two symbols with PSS_ANCHOR=1 -- the anchors' own sort
    (subset mode) counts h in symbols of the text while its text rounds run, and a few dozen anchors are fewer than the 32
    symbols of one key: the "h >= n" guard compared symbols with elements and refused the build."""
    import os
    monkeypatch.setenv('PSS_ANCHOR', '1')
    monkeypatch.setenv('PSS_MODE', 'text')
    monkeypatch.setenv('PSS_PERIOD', '0')
    t = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'regress_anchor_tiny_subset.npy'))
    assert (sa_gpu(t) == oracle.sa(t)).all()
    rng = np.random.default_rng(4)
    for n in (70, 200, 900, 3000):          # tiny texts full of repeats, narrow alphabets
        for alpha in (2, 3):
            blk = rng.integers(0, alpha, max(8, n // 5), dtype=np.uint8) + 190
            t = np.resize(blk, n).copy()
            t[rng.integers(0, n, 3)] = 190
            assert (sa_gpu(t) == oracle.sa(t)).all(), (n, alpha)
