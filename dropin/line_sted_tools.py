"""Import shim: put this directory first on sys.path (or PYTHONPATH) and the
reference's figure scripts (`import line_sted_tools as st`, `from
line_sted_tools import psf_report, tune_psf`; line_sted_figure_1.py:8,
line_sted_figure_2.py:13, line_sted_figure_a1.py:8) run against the MI355X
implementation unchanged."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from rescan_line_sted_amd.line_sted_tools import (  # noqa: E402,F401
    Deconvolver, logarithmic_progress, simulate, deconvolve)
from rescan_line_sted_amd.psf import (  # noqa: E402,F401
    psf_report, generate_psfs, tune_psf, get_width)
