"""A/B of one builder switch on a full-size chunk, hash-checked against libsais' golden:
python tests/tools/words_ab.py <corpus> <ENV_SWITCH> [reps]"""
import ctypes, os, sys
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from pysubstringsearch_amd import _ffi
kind, switch = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = 1 << 29
t = np.empty(n, np.uint8); _ffi.lib.pss_gen_corpus(bench.KINDS[kind], t.ctypes.data, n, 0)
g = bench.load_big_goldens()[(kind, 0, n)]
dT = torch.from_numpy(t).cuda(); dSA = torch.empty(n, dtype=torch.int32, device='cuda'); st = _ffi.SaStats()
for rep in range(reps):
    for on in (True, False):
        if on: os.environ[switch] = '1'
        else: os.environ.pop(switch, None)
        _ffi.check(_ffi.lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, 0, ctypes.byref(st)))
        d = st.as_dict()
        print(f"{switch}={'1' if on else '-'} ms {d['ms_total']:.2f} rounds {d['rounds']} round_passes {d['round_passes']} big {d['big_elems']} verified {bench.sa_poly64_torch(dSA) == g['sa_poly64']}", flush=True)
