"""Re-run one fuzz_search case by seed (prints the switches it drew)."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, '.')
import fuzz_search as F  # noqa: E402
from oracle import oracle as O  # noqa: E402

O.use_reference_sa(O.have_reference())
seed = int(sys.argv[1])
with tempfile.TemporaryDirectory() as tmp:
    try:
        F.one_case(seed, tmp)
        print('ok', {k: v for k, v in os.environ.items() if k.startswith('PSS_')})
    except Exception as e:   # noqa: BLE001
        print('FAIL', type(e).__name__, e, {k: v for k, v in os.environ.items() if k.startswith('PSS_')})
