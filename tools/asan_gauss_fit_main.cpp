// C entry for the host build of csrc/gauss_fit.cpp that tests/test_asan_gauss_fit.py loads (plain g++, or the
// sanitized build of tools/asan_emu.sh).  Test infrastructure only.
#include "../rescan_line_sted_amd/csrc/gauss_fit.hpp"

extern "C" int host_gauss_fit(const double* y, int m, double* p3) { return rl::gauss_fit_lmdif(y, m, p3); }
