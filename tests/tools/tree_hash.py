"""Content hash of the engine's sources (pysubstringsearch_amd/csrc/*.hip, *.h, *.cpp, *.c, Makefile + include/pss.h): what a
counter file under profiles/ was measured on.  The GPU box has no .git, so a commit hash cannot be taken there; a hash of
the files themselves can, and bench.py / check_evidence.py recompute it to say whether the evidence describes THIS tree.

    python tests/tools/tree_hash.py            prints the hash (and the commit given in PSS_TREE_COMMIT, if any)
"""
import pathlib
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def csrc_hash() -> str:
    files = []
    for pat in ('*.hip', '*.h', '*.cpp', '*.c', 'Makefile'):
        files += glob.glob(os.path.join(ROOT, 'pysubstringsearch_amd', 'csrc', pat))
    files.append(os.path.join(ROOT, 'include', 'pss.h'))
    h = hashlib.sha256()
    for f in sorted(files):
        h.update(os.path.relpath(f, ROOT).encode() + b'\0')
        h.update(pathlib.Path(f).read_bytes())
        h.update(b'\0')
    return h.hexdigest()[:16]


def stamp() -> dict:
    """Fields every evidence json carries."""
    return {'csrc_sha256_16': csrc_hash(), 'commit': os.environ.get('PSS_TREE_COMMIT') or None}


if __name__ == '__main__':
    print(csrc_hash(), os.environ.get('PSS_TREE_COMMIT', ''))
