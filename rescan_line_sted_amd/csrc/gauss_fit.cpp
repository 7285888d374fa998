// gauss_fit.cpp -- host-side Gaussian width fit of get_width
// (figure_generation/line_sted_tools.py:653-668).
//
// The reference calls scipy.optimize.curve_fit, i.e. MINPACK lmdif with scipy's
// defaults (ftol = xtol = 1.49012e-8, gtol = 0, factor = 100, forward-difference
// Jacobian, mode-1 scaling).  That iteration stops up to ~1e-6 short of the true
// least-squares minimum, and the result decides an integer (the line rescan
// ratio, :252-256), so the same published algorithm (More, Garbow, Hillstrom:
// lmdif / fdjac2 / qrfac / lmpar / qrsolv / enorm) is restated here for the
// 3-parameter model A*exp(-(x-mu)^2 / (2 sigma^2)).  Scalar host code: it is
// control logic between device phases, not part of the data-parallel path.
#include "gauss_fit.hpp"

#include <cmath>
#include <vector>

namespace rl {
namespace {

constexpr int N = 3;
constexpr double kEps = 2.220446049250313e-16;
constexpr double kDwarf = 2.2250738585072014e-308;

double enorm(const double* v, int n) {
    const double rdwarf = 3.834e-20, rgiant = 1.304e19;
    const double agiant = rgiant / n;
    double s1 = 0, s2 = 0, s3 = 0, x1max = 0, x3max = 0;
    for (int i = 0; i < n; ++i) {
        const double t = std::fabs(v[i]);
        if (t > rdwarf && t < agiant) {
            s2 += t * t;
        } else if (t <= rdwarf) {
            if (t > x3max) {
                const double q = x3max / t;
                s3 = 1.0 + s3 * q * q;
                x3max = t;
            } else if (t != 0.0) {
                const double q = t / x3max;
                s3 += q * q;
            }
        } else {
            if (t > x1max) {
                const double q = x1max / t;
                s1 = 1.0 + s1 * q * q;
                x1max = t;
            } else {
                const double q = t / x1max;
                s1 += q * q;
            }
        }
    }
    if (s1 != 0.0) return x1max * std::sqrt(s1 + (s2 / x1max) / x1max);
    if (s2 != 0.0) {
        if (s2 >= x3max) return std::sqrt(s2 * (1.0 + (x3max / s2) * (x3max * s3)));
        return std::sqrt(x3max * ((s2 / x3max) + (x3max * s3)));
    }
    return x3max * std::sqrt(s3);
}

struct Model {
    const double* y;
    int m;
    void residual(const double* p, double* f) const {
        const double A = p[0], mu = p[1], s = p[2];
        for (int i = 0; i < m; ++i) {
            const double d = (double)i - mu;
            f[i] = A * std::exp(-(d * d) / (2. * (s * s))) - y[i];
        }
    }
};

// column-major m x 3 Jacobian: a[j*m + i]
void qrfac(double* a, int m, int* ipvt, double* rdiag, double* acnorm) {
    double wa[N];
    for (int j = 0; j < N; ++j) {
        acnorm[j] = enorm(a + j * m, m);
        rdiag[j] = acnorm[j];
        wa[j] = rdiag[j];
        ipvt[j] = j;
    }
    for (int j = 0; j < N && j < m; ++j) {
        int kmax = j;
        for (int k = j; k < N; ++k)
            if (rdiag[k] > rdiag[kmax]) kmax = k;
        if (kmax != j) {
            for (int i = 0; i < m; ++i) std::swap(a[j * m + i], a[kmax * m + i]);
            rdiag[kmax] = rdiag[j];
            wa[kmax] = wa[j];
            std::swap(ipvt[j], ipvt[kmax]);
        }
        double ajnorm = enorm(a + j * m + j, m - j);
        if (ajnorm != 0.0) {
            if (a[j * m + j] < 0.0) ajnorm = -ajnorm;
            for (int i = j; i < m; ++i) a[j * m + i] /= ajnorm;
            a[j * m + j] += 1.0;
            for (int k = j + 1; k < N; ++k) {
                double sum = 0.0;
                for (int i = j; i < m; ++i) sum += a[j * m + i] * a[k * m + i];
                const double temp = sum / a[j * m + j];
                for (int i = j; i < m; ++i) a[k * m + i] -= temp * a[j * m + i];
                if (rdiag[k] != 0.0) {
                    const double t = a[k * m + j] / rdiag[k];
                    rdiag[k] *= std::sqrt(std::fmax(0.0, 1.0 - t * t));
                    const double q = rdiag[k] / wa[k];
                    if (0.05 * (q * q) <= kEps) {
                        rdiag[k] = enorm(a + k * m + j + 1, m - j - 1);
                        wa[k] = rdiag[k];
                    }
                }
            }
        }
        rdiag[j] = -ajnorm;
    }
}

// r[i][j] row-major 3x3; upper triangle = R, strict lower overwritten with S^T
void qrsolv(double r[N][N], const int* ipvt, const double* diag, const double* qtb, double* x, double* sdiag) {
    double wa[N];
    for (int j = 0; j < N; ++j) {
        for (int i = j; i < N; ++i) r[i][j] = r[j][i];
        x[j] = r[j][j];
        wa[j] = qtb[j];
    }
    for (int j = 0; j < N; ++j) {
        const int l = ipvt[j];
        if (diag[l] != 0.0) {
            for (int k = j; k < N; ++k) sdiag[k] = 0.0;
            sdiag[j] = diag[l];
            double qtbpj = 0.0;
            for (int k = j; k < N; ++k) {
                if (sdiag[k] == 0.0) continue;
                double sn, cs;
                if (std::fabs(r[k][k]) < std::fabs(sdiag[k])) {
                    const double cotan = r[k][k] / sdiag[k];
                    sn = 0.5 / std::sqrt(0.25 + 0.25 * cotan * cotan);
                    cs = sn * cotan;
                } else {
                    const double tn = sdiag[k] / r[k][k];
                    cs = 0.5 / std::sqrt(0.25 + 0.25 * tn * tn);
                    sn = cs * tn;
                }
                r[k][k] = cs * r[k][k] + sn * sdiag[k];
                const double temp = cs * wa[k] + sn * qtbpj;
                qtbpj = -sn * wa[k] + cs * qtbpj;
                wa[k] = temp;
                for (int i = k + 1; i < N; ++i) {
                    const double t = cs * r[i][k] + sn * sdiag[i];
                    sdiag[i] = -sn * r[i][k] + cs * sdiag[i];
                    r[i][k] = t;
                }
            }
        }
        sdiag[j] = r[j][j];
        r[j][j] = x[j];
    }
    int nsing = N;
    for (int j = 0; j < N; ++j) {
        if (sdiag[j] == 0.0 && nsing == N) nsing = j;
        if (nsing < N) wa[j] = 0.0;
    }
    for (int k = 1; k <= nsing; ++k) {
        const int j = nsing - k;
        double sum = 0.0;
        for (int i = j + 1; i < nsing; ++i) sum += r[i][j] * wa[i];
        wa[j] = (wa[j] - sum) / sdiag[j];
    }
    for (int j = 0; j < N; ++j) x[ipvt[j]] = wa[j];
}

void lmpar(double r[N][N], const int* ipvt, const double* diag, const double* qtb, double delta, double* par,
           double* x, double* sdiag) {
    double wa1[N], wa2[N];
    int nsing = N;
    for (int j = 0; j < N; ++j) {
        wa1[j] = qtb[j];
        if (r[j][j] == 0.0 && nsing == N) nsing = j;
        if (nsing < N) wa1[j] = 0.0;
    }
    for (int k = 1; k <= nsing; ++k) {
        const int j = nsing - k;
        wa1[j] /= r[j][j];
        const double temp = wa1[j];
        for (int i = 0; i < j; ++i) wa1[i] -= r[i][j] * temp;
    }
    for (int j = 0; j < N; ++j) x[ipvt[j]] = wa1[j];
    for (int j = 0; j < N; ++j) sdiag[j] = 0.0;
    int iter = 0;
    for (int j = 0; j < N; ++j) wa2[j] = diag[j] * x[j];
    double dxnorm = enorm(wa2, N);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) {
        *par = 0.0;
        return;
    }
    double parl = 0.0;
    if (nsing >= N) {
        for (int j = 0; j < N; ++j) {
            const int l = ipvt[j];
            wa1[j] = diag[l] * (wa2[l] / dxnorm);
        }
        for (int j = 0; j < N; ++j) {
            double sum = 0.0;
            for (int i = 0; i < j; ++i) sum += r[i][j] * wa1[i];
            wa1[j] = (wa1[j] - sum) / r[j][j];
        }
        const double temp = enorm(wa1, N);
        parl = ((fp / delta) / temp) / temp;
    }
    for (int j = 0; j < N; ++j) {
        double sum = 0.0;
        for (int i = 0; i <= j; ++i) sum += r[i][j] * qtb[i];
        wa1[j] = sum / diag[ipvt[j]];
    }
    const double gnorm = enorm(wa1, N);
    double paru = gnorm / delta;
    if (paru == 0.0) paru = kDwarf / std::fmin(delta, 0.1);
    *par = std::fmax(*par, parl);
    *par = std::fmin(*par, paru);
    if (*par == 0.0) *par = gnorm / dxnorm;
    for (;;) {
        ++iter;
        if (*par == 0.0) *par = std::fmax(kDwarf, 0.001 * paru);
        const double sq = std::sqrt(*par);
        for (int j = 0; j < N; ++j) wa1[j] = sq * diag[j];
        qrsolv(r, ipvt, wa1, qtb, x, sdiag);
        for (int j = 0; j < N; ++j) wa2[j] = diag[j] * x[j];
        dxnorm = enorm(wa2, N);
        const double temp = fp;
        fp = dxnorm - delta;
        if (std::fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || iter == 10) break;
        for (int j = 0; j < N; ++j) {
            const int l = ipvt[j];
            wa1[j] = diag[l] * (wa2[l] / dxnorm);
        }
        for (int j = 0; j < N; ++j) {
            wa1[j] /= sdiag[j];
            const double t = wa1[j];
            for (int i = j + 1; i < N; ++i) wa1[i] -= r[i][j] * t;
        }
        const double tn = enorm(wa1, N);
        const double parc = ((fp / delta) / tn) / tn;
        if (fp > 0.0) parl = std::fmax(parl, *par);
        if (fp < 0.0) paru = std::fmin(paru, *par);
        *par = std::fmax(parl, *par + parc);
    }
}

}  // namespace

int gauss_fit_lmdif(const double* y, int m, double p[3]) {
    const double ftol = 1.49012e-8, xtol = 1.49012e-8, gtol = 0.0, factor = 100.0;
    const int maxfev = 200 * (N + 1);
    Model model{y, m};
    double x[N] = {1.0, m / 2.0, 1.0};                     // p0 of :665
    std::vector<double> fvec(m), wa4(m), fjac((size_t)m * N), tmp(m);
    model.residual(x, fvec.data());
    int nfev = 1, iter = 1, info = 0;
    double fnorm = enorm(fvec.data(), m);
    double par = 0.0, xnorm = 0.0, delta = 0.0;
    double diag[N] = {1, 1, 1};
    const double eps = std::sqrt(kEps);
    for (;;) {
        for (int j = 0; j < N; ++j) {                       // fdjac2
            const double keep = x[j];
            double h = eps * std::fabs(keep);
            if (h == 0.0) h = eps;
            x[j] = keep + h;
            model.residual(x, tmp.data());
            x[j] = keep;
            for (int i = 0; i < m; ++i) fjac[(size_t)j * m + i] = (tmp[i] - fvec[i]) / h;
        }
        nfev += N;
        int ipvt[N];
        double rdiag[N], acnorm[N];
        qrfac(fjac.data(), m, ipvt, rdiag, acnorm);
        if (iter == 1) {
            double wa3[N];
            for (int j = 0; j < N; ++j) {
                diag[j] = acnorm[j] != 0.0 ? acnorm[j] : 1.0;
                wa3[j] = diag[j] * x[j];
            }
            xnorm = enorm(wa3, N);
            delta = factor * xnorm;
            if (delta == 0.0) delta = factor;
        }
        for (int i = 0; i < m; ++i) wa4[i] = fvec[i];
        double qtf[N];
        for (int j = 0; j < N; ++j) {
            double* col = fjac.data() + (size_t)j * m;
            if (col[j] != 0.0) {
                double sum = 0.0;
                for (int i = j; i < m; ++i) sum += col[i] * wa4[i];
                const double temp = -sum / col[j];
                for (int i = j; i < m; ++i) wa4[i] += col[i] * temp;
            }
            col[j] = rdiag[j];
            qtf[j] = wa4[j];
        }
        double r[N][N];
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) r[i][j] = fjac[(size_t)j * m + i];
        double gnorm = 0.0;
        if (fnorm != 0.0) {
            for (int j = 0; j < N; ++j) {
                const int l = ipvt[j];
                if (acnorm[l] != 0.0) {
                    double sum = 0.0;
                    for (int i = 0; i <= j; ++i) sum += r[i][j] * (qtf[i] / fnorm);
                    gnorm = std::fmax(gnorm, std::fabs(sum / acnorm[l]));
                }
            }
        }
        if (gnorm <= gtol) {
            info = 4;
            break;
        }
        for (int j = 0; j < N; ++j) diag[j] = std::fmax(diag[j], acnorm[j]);
        for (;;) {
            double step[N], sdiag[N], wa1[N], wa2[N], wa3[N];
            lmpar(r, ipvt, diag, qtf, delta, &par, step, sdiag);
            for (int j = 0; j < N; ++j) {
                wa1[j] = -step[j];
                wa2[j] = x[j] + wa1[j];
                wa3[j] = diag[j] * wa1[j];
            }
            const double pnorm = enorm(wa3, N);
            if (iter == 1) delta = std::fmin(delta, pnorm);
            model.residual(wa2, wa4.data());
            ++nfev;
            const double fnorm1 = enorm(wa4.data(), m);
            double actred = -1.0;
            if (0.1 * fnorm1 < fnorm) {
                const double q = fnorm1 / fnorm;
                actred = 1.0 - q * q;
            }
            for (int j = 0; j < N; ++j) wa3[j] = 0.0;
            for (int j = 0; j < N; ++j) {
                const double temp = wa1[ipvt[j]];
                for (int i = 0; i <= j; ++i) wa3[i] += r[i][j] * temp;
            }
            const double temp1 = enorm(wa3, N) / fnorm;
            const double temp2 = (std::sqrt(par) * pnorm) / fnorm;
            const double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
            const double dirder = -(temp1 * temp1 + temp2 * temp2);
            double ratio = 0.0;
            if (prered != 0.0) ratio = actred / prered;
            if (ratio <= 0.25) {
                double temp = actred >= 0.0 ? 0.5 : 0.5 * dirder / (dirder + 0.5 * actred);
                if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                delta = temp * std::fmin(delta, pnorm / 0.1);
                par = par / temp;
            } else if (par == 0.0 || ratio >= 0.75) {
                delta = pnorm / 0.5;
                par = 0.5 * par;
            }
            if (ratio >= 1e-4) {
                double w[N];
                for (int j = 0; j < N; ++j) {
                    x[j] = wa2[j];
                    w[j] = diag[j] * x[j];
                }
                for (int i = 0; i < m; ++i) fvec[i] = wa4[i];
                xnorm = enorm(w, N);
                fnorm = fnorm1;
                ++iter;
            }
            const bool small = std::fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0;
            if (small) info = 1;
            if (delta <= xtol * xnorm) info = 2;
            if (small && info == 2) info = 3;
            if (info != 0) break;
            if (nfev >= maxfev) info = 5;
            if (std::fabs(actred) <= kEps && prered <= kEps && 0.5 * ratio <= 1.0) info = 6;
            if (delta <= kEps * xnorm) info = 7;
            if (gnorm <= kEps) info = 8;
            if (info != 0) break;
            if (ratio >= 1e-4) break;
        }
        if (info != 0) break;
    }
    for (int j = 0; j < N; ++j) p[j] = x[j];
    return info;
}

}  // namespace rl
