"""rescan_line_sted_amd -- MI355X-native rescan line-STED image formation.

`line_sted_tools` mirrors the reference module of the same name
(figure_generation/line_sted_tools.py); `dropin/` holds import shims so the
reference's figure scripts find it under the original module names.
"""
__version__ = '0.1.0'
