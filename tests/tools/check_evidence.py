"""Do the counter files under profiles/ describe this tree?  Every pmc_*.json written by the round-5 tools carries the
content hash of the engine's sources it was measured on (tests/tools/tree_hash.py); a file whose hash differs from the
current sources -- or that carries none -- is stale.

    python tests/tools/check_evidence.py [--strict]     exit 1 when a file is stale (--strict: also for files without a hash)
"""
import pathlib
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tree_hash  # noqa: E402

now = tree_hash.csrc_hash()
strict = '--strict' in sys.argv
bad = 0
for f in sorted(glob.glob(os.path.join(tree_hash.ROOT, 'profiles', 'pmc_*.json'))):
    try:
        d = json.loads(pathlib.Path(f).read_text())
    except Exception as e:                      # noqa: BLE001
        print(f'{os.path.basename(f)}: unreadable ({e})')
        bad += 1
        continue
    h = d.get('csrc_sha256_16')
    if h == now:
        state = 'current'
    elif h is None:
        state = 'no hash (written before round 5)'
        bad += 1 if strict else 0
    else:
        state = f'STALE: measured on {h} (commit {d.get("commit")}), the sources are {now}'
        bad += 1
    print(f'{os.path.basename(f):40s} {state}')
sys.exit(1 if bad else 0)
