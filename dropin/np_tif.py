"""Import shim for `import np_tif` (line_sted_figure_2.py:12, line_sted_figure_3.py:12)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from rescan_line_sted_amd.np_tif import tif_to_array, array_to_tif, parse_tif  # noqa: E402,F401
