// gauss_fit.hpp -- MINPACK-lmdif Gaussian fit used by get_width (host side)
#pragma once
namespace rl {
// Fits A*exp(-(x-mu)^2/(2 sigma^2)) to y[0..m-1] at x = 0..m-1 from the start
// [1, m/2, 1] with scipy.optimize.curve_fit's defaults; p = {A, mu, sigma}.
// Returns MINPACK's info code (1..4 = converged).
int gauss_fit_lmdif(const double* y, int m, double p[3]);
}  // namespace rl
