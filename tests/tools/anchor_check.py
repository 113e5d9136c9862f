"""Exploration / fuzz: the anchor round (anchor_impl.h) against the oracle on texts full of repeats.

    python tests/tools/anchor_check.py [cases=60] [seed=1]

Every case is built twice through pss_sa_build_device -- PSS_ANCHOR=1 (the round is taken whenever ties outlive the
text rounds, whatever the size of the text, with a random window) and PSS_ANCHOR=0 (rank rounds) -- and compared with
the oracle's suffix array; prints which cases took the round and what it left.
"""
import ctypes
import os
import random
import sys

import numpy as np

sys.path.insert(0, '.')
import torch  # noqa: E402

from pysubstringsearch_amd import _ffi  # noqa: E402


def build(t, **env):
    old = {k: os.environ.get(k) for k in env}
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    try:
        n = t.size
        dT = torch.from_numpy(t).cuda()
        dSA = torch.empty(n, dtype=torch.int32, device='cuda')
        st = _ffi.SaStats()
        _ffi.check(_ffi.lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, 0, ctypes.byref(st)))
        return dSA.cpu().numpy(), st.as_dict()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def make_case(rng: random.Random):
    nrng = np.random.default_rng(rng.randrange(1 << 30))
    kind = rng.randrange(9)
    alpha = rng.choice([2, 3, 4, 26, 39, 200, 256])
    n = rng.choice([3000, 20000, 70000, 300000, 1 << 20])
    base = nrng.integers(0, alpha, n, dtype=np.uint8)
    if alpha == 26:
        base += 97
    if kind == 0:       # duplicated blocks with edits
        blk = min(n, rng.choice([200, 1000, 5000, 40000]))
        src = base[:blk].copy()
        parts = []
        while sum(p.size for p in parts) < n:
            c = src.copy()
            for _ in range(rng.randrange(0, 4)):
                c[rng.randrange(blk)] = nrng.integers(0, alpha)
            parts.append(c)
        t = np.concatenate(parts)[:n]
    elif kind == 1:     # a periodic stretch inside random text
        p = rng.choice([2, 3, 7, 40, 60, 333])
        word = base[:p].copy()
        a, b = sorted(rng.sample(range(n), 2))
        t = base.copy()
        t[a:b] = np.resize(word, b - a)
    elif kind == 2:     # blocks copied from elsewhere in the text
        t = base.copy()
        blk = rng.choice([100, 500, 4096, 30000])
        for _ in range(rng.randrange(1, 30)):
            if n <= 2 * blk:
                break
            s, d = rng.randrange(n - blk), rng.randrange(n - blk)
            t[d:d + blk] = t[s:s + blk].copy()
    elif kind == 3:     # two periodic stretches of the same word, different ends
        p = rng.choice([5, 13, 64])
        word = base[:p].copy()
        t = base.copy()
        q = n // 4
        t[q:2 * q] = np.resize(word, q)
        t[3 * q:3 * q + q // 2] = np.resize(word, q // 2)
    elif kind == 4:     # whole text = one block repeated (no edits) + tail
        blk = min(n, rng.choice([37, 1000, 9999]))
        t = np.resize(base[:blk], n).copy()
        t[-rng.randrange(1, 50):] = nrng.integers(0, alpha)
    elif kind == 5:     # runs mixed with copies
        t = base.copy()
        for _ in range(20):
            s = rng.randrange(n - 600)
            t[s:s + rng.randrange(20, 600)] = t[s]
        half = n // 2
        t[half:half + n // 4] = t[:n // 4]
    elif kind in (7, 8):   # several runs of one word (equal lengths among them, one up to the end of the text), other words too
        t = base.copy()
        words = [base[s:s + rng.choice([1, 2, 3, 4, 7, 12, 60, 200])].copy() for s in (0, 300, 900)]
        lens = [rng.choice([5000, 20000, 70000]) for _ in range(3)]
        at = 0
        for r in range(rng.randrange(2, 9)):
            w = words[rng.randrange(1 if kind == 7 else 3)]
            L = min(lens[rng.randrange(3)], n // 10)
            at += rng.randrange(1, max(2, n // 10))
            if at + L > n:
                break
            ph = rng.randrange(w.size)
            t[at:at + L] = np.resize(np.roll(w, -ph), L)
            at += L
        if rng.random() < 0.4:
            L = min(n // 8, 30000)
            t[n - L:] = np.resize(words[0], L)
    else:               # natural-text-like corpus with a repetitive middle
        t = np.empty(n, dtype=np.uint8)
        _ffi.check(_ffi.lib.pss_gen_corpus(6, t.ctypes.data, n, rng.randrange(4)))
    return np.ascontiguousarray(t), kind, alpha


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from oracle import oracle as O
    rng = random.Random(seed)
    bad = 0
    took = 0
    for c in range(cases):
        t, kind, alpha = make_case(rng)
        exp = O.sa(t)
        omega = rng.choice([None, 2, 3, 5, 9, 17, 33])
        got, st = build(t, PSS_ANCHOR=1, PSS_ANCHOR_OMEGA=omega)
        ok = bool((got == exp).all())
        got0, st0 = build(t, PSS_ANCHOR=0, PSS_ANCHOR_OMEGA=None)
        ok0 = bool((got0 == exp).all())
        took += int(st['anchor'])
        flag = '' if (ok and ok0 and st['anchor_left'] == 0) else '  <-- MISMATCH'
        bad += 0 if not flag else 1
        print(f'case {c}: kind={kind} alpha={alpha} n={t.size} omega={omega} ok={ok} ok0={ok0} anchor={st["anchor"]} '
              f'w={st["anchor_w"]} om={st["anchor_omega"]} depth={st["anchor_depth"]} anchors={st["anchor_count"]} '
              f'active={st["anchor_active"]} left={st["anchor_left"]} arounds={st["anchor_text_rounds"]}+{st["anchor_rounds"]} lv={st["anchor_levels"]} '
              f'rounds={st["rounds"]} vs {st0["rounds"]} per={st["periodic_rounds"]}/{st["periodic_members"]} vs {st0["periodic_rounds"]}/{st0["periodic_members"]} '
              f'ms={st["ms_total"]:.2f} vs {st0["ms_total"]:.2f}{flag}', flush=True)
    print(f'{cases} cases, {took} took the anchor round, {bad} bad')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
