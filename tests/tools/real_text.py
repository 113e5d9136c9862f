"""Real files through the builder: source code, headers and documentation found on the machine (Python packages, ROCm
headers), concatenated in sorted path order into one chunk -- licence headers copied thousands of times, ASCII rules,
tables of numbers, bytes above 127 -- built on the GPU and compared with libsais (oracle/_ref) byte for byte.

    python tests/tools/real_text.py [logn=26] [reps=3]
"""
import pathlib
import ctypes
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import torch  # noqa: E402

from pysubstringsearch_amd import _ffi  # noqa: E402

ROOTS = ('/usr/lib/python3.10', '/usr/local/lib/python3.10/dist-packages', '/opt/rocm/include', '/usr/include', '/usr/share/doc')
EXTS = ('.py', '.h', '.hpp', '.hip', '.txt', '.md', '.rst', '.json', '.cmake', '.c', '.cpp', '.pyi', '.yaml', '.cfg', '.inc')


def collect(limit):
    parts, total = [], 0
    for root in ROOTS:
        for dp, dn, fn in os.walk(root):
            dn.sort()
            for f in sorted(fn):
                if not f.endswith(EXTS):
                    continue
                p = os.path.join(dp, f)
                try:
                    if os.path.islink(p) or os.path.getsize(p) > (8 << 20):
                        continue
                    b = pathlib.Path(p).read_bytes()
                except OSError:
                    continue
                if not b:
                    continue
                if not b.endswith(b'\n'):
                    b += b'\n'
                parts.append(b)
                total += len(b)
                if total >= limit:
                    return b''.join(parts)[:limit]
    return b''.join(parts)


def two_chunks(logn):
    """The plan of the sample sort on real files: the text in two halves of equal geometry, built one after the other
    a few times -- is the second half cut by the first half's sample, or refused?"""
    n = 1 << logn
    raw = collect(n)
    half = (len(raw) // 2) & ~0xfff
    parts = [np.frombuffer(raw[:half], dtype=np.uint8).copy(), np.frombuffer(raw[half:2 * half - 4097], dtype=np.uint8).copy()]
    for t in parts:
        t[-1] = 10
    from oracle import oracle as O
    want = [hashlib.sha256(O.sa(t).tobytes()).hexdigest() for t in parts]
    dTs = [torch.from_numpy(t).cuda() for t in parts]
    dSA = torch.empty(half, dtype=torch.int32, device='cuda')
    st = _ffi.SaStats()
    ok = True
    for r in range(8):
        k = r % 2
        nn = parts[k].size
        _ffi.check(_ffi.lib.pss_sa_build_device(dTs[k].data_ptr(), dSA.data_ptr(), nn, 0, 8 if r == 0 else 0, ctypes.byref(st)))
        d = st.as_dict()
        same = hashlib.sha256(dSA[:nn].cpu().numpy().tobytes()).hexdigest() == want[k]
        ok = ok and same
        print(f'build {r} (half {k}, {nn} bytes): {st.ms_total:.1f} ms  plan={d["plan_hint"]} ss={d["ss"]} planned={d["ss_planned"]} '
              f'refused={d["ss_plan_refused"]} max bucket {d["ss_max_bucket"]} equal to libsais: {same}', flush=True)
    sys.exit(0 if ok else 1)


def main():
    if len(sys.argv) > 2 and sys.argv[2] == 'halves':
        return two_chunks(int(sys.argv[1]))
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    n = 1 << logn
    t0 = time.time()
    cache = f'/tmp/real_text_{logn}.bin'
    if os.path.exists(cache):
        raw = pathlib.Path(cache).read_bytes()
    else:
        raw = collect(n)
        pathlib.Path(cache).write_bytes(raw)
    t = np.frombuffer(raw, dtype=np.uint8).copy()
    if t.size and t[-1] != 10:
        t[-1] = 10
    n = t.size
    print(f'{n} bytes of real files collected in {time.time() - t0:.1f} s; {len(np.unique(t))} byte values, '
          f'{int((t == 10).sum())} lines', flush=True)
    dT = torch.from_numpy(t).cuda()
    dSA = torch.empty(n, dtype=torch.int32, device='cuda')
    st = _ffi.SaStats()
    for r in range(reps):
        _ffi.check(_ffi.lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, 8 if r == 0 else 0, ctypes.byref(st)))
        d = st.as_dict()
        print(f'build {r}: {st.ms_total:.1f} ms = {n / st.ms_total / 1e6:.2f} GB/s  ss={d["ss"]} msd={d["msd"]} rle={d["rle"]} '
              f'anchor={d["anchor"]} (anchors {d["anchor_count"]}, levels {d["anchor_levels"]}, {d["anchor_ms"]:.1f} ms) '
              f'rounds={d["rounds"]} text_rounds={d["text_rounds"]} periodic={d["periodic_rounds"]}/{d["periodic_members"]} '
              f'sum_active={d["sum_active"]} plan={d["plan_hint"]} restarts={d["ms_restarts"]:.1f} ms | sigma={d["sigma"]} key_chars={d["key_chars"]} '
              f'initial {d["ms_initial"]:.1f} ms, ss buckets {d["ss_buckets"]} max {d["ss_max_bucket"]} samples {d["ss_samples"]} nomem {d["ss_declined_nomem"]} big_elems {d["big_elems"]}', flush=True)
    if len(sys.argv) > 3 and sys.argv[3] == 'nocheck':
        return
    got = dSA.cpu().numpy()
    from oracle import oracle as O
    t0 = time.time()
    if O.have_reference():
        ref = O.sa_reference(t)
        how = 'libsais (oracle/_ref)'
    else:
        ref = O.sa_restatement(t)
        how = 'oracle restatement'
    cpu_s = time.time() - t0
    ok = bool(np.array_equal(got, ref))
    print(f'{how}: {cpu_s:.1f} s = {n / cpu_s / 1e9:.3f} GB/s; equal: {ok}; sha256 {hashlib.sha256(got.tobytes()).hexdigest()[:16]}')
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
