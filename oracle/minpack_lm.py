"""CPU ORACLE -- test infrastructure only.

Restatement of the published MINPACK `lmdif` Levenberg-Marquardt driver
(More, Garbow, Hillstrom 1980: lmdif / fdjac2 / qrfac / lmpar / qrsolv /
enorm) as used by scipy.optimize.curve_fit -> leastsq with its defaults
(ftol = xtol = 1.49012e-8, gtol = 0, factor = 100, mode 1 scaling, forward
difference Jacobian with epsfcn = machine epsilon, maxfev = 200*(n+1)).

The reference's get_width (line_sted_tools.py:653-668) stops wherever this
iteration stops, which is up to ~1e-6 away from the true least-squares
minimum, so parity with the reference needs the same iteration, not merely
the same minimum.  MINPACK is third-party code absent from /root/reference;
scipy 1.15.3 was the version behind the goldens.
"""
import math
import numpy as np

EPSMCH = 2.220446049250313e-16
DWARF = 2.2250738585072014e-308
_RDWARF = 3.834e-20
_RGIANT = 1.304e19


def enorm(v):
    v = np.asarray(v, dtype=np.float64)
    n = v.size
    if n == 0:
        return 0.0
    a = np.abs(v)
    agiant = _RGIANT / n
    mid = (a > _RDWARF) & (a < agiant)
    if mid.all():
        s2 = 0.0
        for t in a:                       # sequential sum, as the Fortran
            s2 += t * t
        return math.sqrt(s2)
    s1 = s2 = s3 = 0.0
    x1max = x3max = 0.0
    for t in a:
        t = float(t)
        if _RDWARF < t < agiant:
            s2 += t * t
        elif t <= _RDWARF:
            if t > x3max:
                s3 = 1.0 + s3 * (x3max / t) ** 2
                x3max = t
            elif t != 0.0:
                s3 += (t / x3max) ** 2
        else:
            if t > x1max:
                s1 = 1.0 + s1 * (x1max / t) ** 2
                x1max = t
            else:
                s1 += (t / x1max) ** 2
    if s1 != 0.0:
        return x1max * math.sqrt(s1 + (s2 / x1max) / x1max)
    if s2 != 0.0:
        if s2 >= x3max:
            return math.sqrt(s2 * (1.0 + (x3max / s2) * (x3max * s3)))
        return math.sqrt(x3max * ((s2 / x3max) + (x3max * s3)))
    return x3max * math.sqrt(s3)


def _dot(a, b):
    s = 0.0
    for x, y in zip(a, b):
        s += x * y
    return s


def qrfac(a):
    """Householder QR with column pivoting, in place on a (m x n).
    Returns ipvt, rdiag, acnorm."""
    m, n = a.shape
    acnorm = np.array([enorm(a[:, j]) for j in range(n)])
    rdiag = acnorm.copy()
    wa = rdiag.copy()
    ipvt = list(range(n))
    for j in range(min(m, n)):
        kmax = j
        for k in range(j, n):
            if rdiag[k] > rdiag[kmax]:
                kmax = k
        if kmax != j:
            a[:, [j, kmax]] = a[:, [kmax, j]]
            rdiag[kmax] = rdiag[j]
            wa[kmax] = wa[j]
            ipvt[j], ipvt[kmax] = ipvt[kmax], ipvt[j]
        ajnorm = enorm(a[j:, j])
        if ajnorm != 0.0:
            if a[j, j] < 0.0:
                ajnorm = -ajnorm
            a[j:, j] /= ajnorm
            a[j, j] += 1.0
            for k in range(j + 1, n):
                s = _dot(a[j:, j], a[j:, k])
                temp = s / a[j, j]
                a[j:, k] -= temp * a[j:, j]
                if rdiag[k] != 0.0:
                    temp = a[j, k] / rdiag[k]
                    rdiag[k] *= math.sqrt(max(0.0, 1.0 - temp * temp))
                    if 0.05 * (rdiag[k] / wa[k]) ** 2 <= EPSMCH:
                        rdiag[k] = enorm(a[j + 1:, k])
                        wa[k] = rdiag[k]
        rdiag[j] = -ajnorm
    return ipvt, rdiag, acnorm


def qrsolv(r, ipvt, diag, qtb):
    """r: n x n, upper triangle holds R (full diagonal).  The strict lower
    triangle is overwritten with the transposed strict upper triangle of S.
    Returns x, sdiag."""
    n = r.shape[0]
    x = np.zeros(n)
    sdiag = np.zeros(n)
    wa = np.array(qtb[:n], dtype=np.float64)
    for j in range(n):
        for i in range(j, n):
            r[i, j] = r[j, i]
        x[j] = r[j, j]
    for j in range(n):
        l = ipvt[j]
        if diag[l] != 0.0:
            sdiag[j:] = 0.0
            sdiag[j] = diag[l]
            qtbpj = 0.0
            for k in range(j, n):
                if sdiag[k] == 0.0:
                    continue
                if abs(r[k, k]) < abs(sdiag[k]):
                    cotan = r[k, k] / sdiag[k]
                    sin = 0.5 / math.sqrt(0.25 + 0.25 * cotan * cotan)
                    cos = sin * cotan
                else:
                    tan = sdiag[k] / r[k, k]
                    cos = 0.5 / math.sqrt(0.25 + 0.25 * tan * tan)
                    sin = cos * tan
                r[k, k] = cos * r[k, k] + sin * sdiag[k]
                temp = cos * wa[k] + sin * qtbpj
                qtbpj = -sin * wa[k] + cos * qtbpj
                wa[k] = temp
                for i in range(k + 1, n):
                    temp = cos * r[i, k] + sin * sdiag[i]
                    sdiag[i] = -sin * r[i, k] + cos * sdiag[i]
                    r[i, k] = temp
        sdiag[j] = r[j, j]
        r[j, j] = x[j]
    nsing = n
    for j in range(n):
        if sdiag[j] == 0.0 and nsing == n:
            nsing = j
        if nsing < n:
            wa[j] = 0.0
    for k in range(1, nsing + 1):
        j = nsing - k
        s = 0.0
        for i in range(j + 1, nsing):
            s += r[i, j] * wa[i]
        wa[j] = (wa[j] - s) / sdiag[j]
    for j in range(n):
        x[ipvt[j]] = wa[j]
    return x, sdiag


def lmpar(r, ipvt, diag, qtb, delta, par):
    """Returns par, x (the step, before negation), sdiag."""
    n = r.shape[0]
    wa1 = np.zeros(n)
    nsing = n
    for j in range(n):
        wa1[j] = qtb[j]
        if r[j, j] == 0.0 and nsing == n:
            nsing = j
        if nsing < n:
            wa1[j] = 0.0
    for k in range(1, nsing + 1):
        j = nsing - k
        wa1[j] /= r[j, j]
        temp = wa1[j]
        for i in range(j):
            wa1[i] -= r[i, j] * temp
    x = np.zeros(n)
    for j in range(n):
        x[ipvt[j]] = wa1[j]
    sdiag = np.zeros(n)
    it = 0
    wa2 = diag * x
    dxnorm = enorm(wa2)
    fp = dxnorm - delta
    if fp <= 0.1 * delta:
        return 0.0, x, sdiag
    parl = 0.0
    if nsing >= n:
        for j in range(n):
            l = ipvt[j]
            wa1[j] = diag[l] * (wa2[l] / dxnorm)
        for j in range(n):
            s = 0.0
            for i in range(j):
                s += r[i, j] * wa1[i]
            wa1[j] = (wa1[j] - s) / r[j, j]
        temp = enorm(wa1)
        parl = ((fp / delta) / temp) / temp
    for j in range(n):
        s = 0.0
        for i in range(j + 1):
            s += r[i, j] * qtb[i]
        wa1[j] = s / diag[ipvt[j]]
    gnorm = enorm(wa1)
    paru = gnorm / delta
    if paru == 0.0:
        paru = DWARF / min(delta, 0.1)
    par = max(par, parl)
    par = min(par, paru)
    if par == 0.0:
        par = gnorm / dxnorm
    while True:
        it += 1
        if par == 0.0:
            par = max(DWARF, 0.001 * paru)
        temp = math.sqrt(par)
        wa1 = temp * diag
        x, sdiag = qrsolv(r, ipvt, wa1, qtb)
        wa2 = diag * x
        dxnorm = enorm(wa2)
        temp = fp
        fp = dxnorm - delta
        if (abs(fp) <= 0.1 * delta or
                (parl == 0.0 and fp <= temp and temp < 0.0) or it == 10):
            break
        wa1 = np.zeros(n)
        for j in range(n):
            l = ipvt[j]
            wa1[j] = diag[l] * (wa2[l] / dxnorm)
        for j in range(n):
            wa1[j] /= sdiag[j]
            temp = wa1[j]
            for i in range(j + 1, n):
                wa1[i] -= r[i, j] * temp
        temp = enorm(wa1)
        parc = ((fp / delta) / temp) / temp
        if fp > 0.0:
            parl = max(parl, par)
        if fp < 0.0:
            paru = min(paru, par)
        par = max(parl, par + parc)
    return par, x, sdiag


def lmdif(fcn, x0, ftol=1.49012e-8, xtol=1.49012e-8, gtol=0.0, maxfev=0,
          epsfcn=EPSMCH, factor=100.0):
    """Minimise sum(fcn(x)**2).  Returns (x, info, nfev)."""
    x = np.array(x0, dtype=np.float64)
    n = x.size
    if maxfev == 0:
        maxfev = 200 * (n + 1)
    fvec = np.asarray(fcn(x), dtype=np.float64)
    m = fvec.size
    nfev = 1
    fnorm = enorm(fvec)
    par = 0.0
    it = 1
    info = 0
    eps = math.sqrt(max(epsfcn, EPSMCH))
    diag = np.ones(n)
    xnorm = 0.0
    delta = 0.0
    while True:
        # forward-difference Jacobian (fdjac2)
        fjac = np.empty((m, n))
        for j in range(n):
            temp = x[j]
            h = eps * abs(temp)
            if h == 0.0:
                h = eps
            x[j] = temp + h
            wa = np.asarray(fcn(x), dtype=np.float64)
            x[j] = temp
            fjac[:, j] = (wa - fvec) / h
        nfev += n
        ipvt, rdiag, acnorm = qrfac(fjac)
        if it == 1:
            for j in range(n):
                diag[j] = acnorm[j] if acnorm[j] != 0.0 else 1.0
            xnorm = enorm(diag * x)
            delta = factor * xnorm
            if delta == 0.0:
                delta = factor
        # (Q^T fvec), first n components
        wa4 = fvec.copy()
        qtf = np.zeros(n)
        for j in range(n):
            if fjac[j, j] != 0.0:
                s = _dot(fjac[j:, j], wa4[j:])
                temp = -s / fjac[j, j]
                wa4[j:] += fjac[j:, j] * temp
            fjac[j, j] = rdiag[j]
            qtf[j] = wa4[j]
        gnorm = 0.0
        if fnorm != 0.0:
            for j in range(n):
                l = ipvt[j]
                if acnorm[l] != 0.0:
                    s = 0.0
                    for i in range(j + 1):
                        s += fjac[i, j] * (qtf[i] / fnorm)
                    gnorm = max(gnorm, abs(s / acnorm[l]))
        if gnorm <= gtol:
            info = 4
            break
        diag = np.maximum(diag, acnorm)
        r = fjac[:n, :n].copy()
        while True:
            par, step, _ = lmpar(r, ipvt, diag, qtf, delta, par)
            wa1 = -step
            wa2 = x + wa1
            wa3 = diag * wa1
            pnorm = enorm(wa3)
            if it == 1:
                delta = min(delta, pnorm)
            wa4 = np.asarray(fcn(wa2), dtype=np.float64)
            nfev += 1
            fnorm1 = enorm(wa4)
            actred = -1.0
            if 0.1 * fnorm1 < fnorm:
                actred = 1.0 - (fnorm1 / fnorm) ** 2
            wa3 = np.zeros(n)
            for j in range(n):
                temp = wa1[ipvt[j]]
                for i in range(j + 1):
                    wa3[i] += r[i, j] * temp
            temp1 = enorm(wa3) / fnorm
            temp2 = (math.sqrt(par) * pnorm) / fnorm
            prered = temp1 * temp1 + temp2 * temp2 / 0.5
            dirder = -(temp1 * temp1 + temp2 * temp2)
            ratio = 0.0
            if prered != 0.0:
                ratio = actred / prered
            if ratio <= 0.25:
                if actred >= 0.0:
                    temp = 0.5
                else:
                    temp = 0.5 * dirder / (dirder + 0.5 * actred)
                if 0.1 * fnorm1 >= fnorm or temp < 0.1:
                    temp = 0.1
                delta = temp * min(delta, pnorm / 0.1)
                par = par / temp
            elif par == 0.0 or ratio >= 0.75:
                delta = pnorm / 0.5
                par = 0.5 * par
            if ratio >= 1e-4:
                x = wa2
                fvec = wa4
                xnorm = enorm(diag * x)
                fnorm = fnorm1
                it += 1
            if abs(actred) <= ftol and prered <= ftol and 0.5 * ratio <= 1.0:
                info = 1
            if delta <= xtol * xnorm:
                info = 2
            if (abs(actred) <= ftol and prered <= ftol and 0.5 * ratio <= 1.0
                    and info == 2):
                info = 3
            if info != 0:
                break
            if nfev >= maxfev:
                info = 5
            if abs(actred) <= EPSMCH and prered <= EPSMCH and 0.5 * ratio <= 1.0:
                info = 6
            if delta <= EPSMCH * xnorm:
                info = 7
            if gnorm <= EPSMCH:
                info = 8
            if info != 0:
                break
            if ratio >= 1e-4:
                break
        if info != 0:
            break
    return x, info, nfev
