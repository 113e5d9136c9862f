"""Prints the library's environment switches (csrc/knobs.h through the C ABI) as the markdown table of README.md: python tests/tools/knobs_md.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pysubstringsearch_amd import _ffi  # noqa: E402

print('| Switch | Unset means | Fuzzed with | Effect |')
print('|---|---|---|---|')
for k in _ffi.knobs():
    print(f"| `{k['name']}` | {k['default']} | {', '.join(k['fuzz']) or '–'} | {k['what']} |".replace('|  |', '| |'))
