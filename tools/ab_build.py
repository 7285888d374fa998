"""Development helper: build librlsted_<name>.so with extra compiler flags for an in-process A/B (tools/gpu/ab_bench.py).

    python3 tools/ab_build.py NAME -DRL_FOO=1 -DRL_BAR=2 ...

Not part of the product build: `python -m rescan_line_sted_amd._build` knows only the named study variants (q16, qbf16).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rescan_line_sted_amd import _build  # noqa: E402

if __name__ == '__main__':
    print(_build.build_variant(sys.argv[1], verbose=False, flags=sys.argv[2:]))
