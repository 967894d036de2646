"""Derive the coefficients of the branch-free fp32 erf used by the GELU
epilogue (csrc/common.h: erf_f32) and measure its error against scipy's
double-precision erf with every operation rounded to fp32.

  |x| <= SPLIT : erf(x) = x + x*P(s),  s = x^2, P(0) = 2/sqrt(pi) - 1
  |x| >  SPLIT : erf(x) = sign(x) * (1 - 2^(t*R(t))),   t = min(|x|, 4)
                 where t*R(t) = log2(erfc(t))
Both branches are evaluated and selected (no divergence on the GPU).
"""
import numpy as np
from numpy.polynomial import chebyshev as Ch
from scipy import special

SPLIT = 0.92
f32 = np.float32


def fit(fun, lo, hi, deg, n=4000):
    k = np.arange(n)
    x = np.cos(np.pi * (k + 0.5) / n)                 # Chebyshev nodes
    xs = 0.5 * (hi - lo) * x + 0.5 * (hi + lo)
    c = Ch.chebfit(x, fun(xs), deg)
    # convert to a plain polynomial in the original variable
    p = Ch.cheb2poly(c)
    # substitute x = (2*v - (hi+lo)) / (hi-lo)
    a, b = 2.0 / (hi - lo), -(hi + lo) / (hi - lo)
    out = np.zeros(1)
    lin = np.array([b, a])
    powr = np.ones(1)
    for coef in p:
        out = np.polynomial.polynomial.polyadd(out, coef * powr)
        powr = np.polynomial.polynomial.polymul(powr, lin)
    return out                                        # ascending coefficients


# small branch: P(s) = erf(x)/x - 1 on s in [0, SPLIT^2]
def small(s):
    x = np.sqrt(np.maximum(s, 1e-300))
    return np.where(s < 1e-12, 2 / np.sqrt(np.pi) - 1, special.erf(x) / x - 1.0)


# large branch: R(t) = log2(erfc(t)) / t on [SPLIT, 4]
def large(t):
    return np.log2(special.erfc(t)) / t


P = fit(small, 0.0, SPLIT * SPLIT, 6)
R = fit(large, SPLIT, 4.0, 8)


def horner32(coefs, v):
    r = np.full_like(v, f32(coefs[-1]))
    for c in coefs[-2::-1]:
        r = (r * v + f32(c)).astype(f32)              # fma modelled as mul+add in fp32 (pessimistic)
    return r


def erf32(x):
    x = x.astype(f32)
    t = np.minimum(np.abs(x), f32(4.0)).astype(f32)
    s = (x * x).astype(f32)
    small_v = (x + x * horner32(P, s)).astype(f32)
    e = np.exp2((t * horner32(R, t)).astype(f32)).astype(f32)
    large_v = np.copysign((f32(1.0) - e).astype(f32), x)
    return np.where(np.abs(x) <= f32(SPLIT), small_v, large_v)


x = np.concatenate([np.linspace(-6, 6, 2000001), np.linspace(-1e-3, 1e-3, 20001)])
ref = special.erf(x.astype(f32).astype(np.float64))
got = erf32(x).astype(np.float64)
err = np.abs(got - ref)
ulp = np.spacing(np.abs(ref).astype(f32)).astype(np.float64)
print('max abs err %.3e  max ulp err %.2f at x=%.4f' % (err.max(), (err / ulp).max(),
                                                          x[(err / ulp).argmax()]))
g_ref = 0.5 * x * (1 + special.erf(x / np.sqrt(2)))
xs = x.astype(f32)
g_got = (f32(0.5) * xs * (f32(1.0) + erf32((xs * f32(0.70710678118654752440)).astype(f32)))).astype(np.float64)
print('gelu max abs err %.3e' % np.abs(g_got - g_ref).max())
print('P =', ', '.join('%.9ef' % c for c in P))
print('R =', ', '.join('%.9ef' % c for c in R))


# ---------------------------------------------------------------------------
# Single-formula GELU (csrc/common.h: gelu_erf, round 2).  GELU adds erf to 1, so only
# ABSOLUTE erf accuracy matters and the small-|x| branch above is unnecessary:
#   gelu(v) = max(v, 0) - |v| 2^(w Q(w) - 1),   erfc(w / sqrt 2) = 2^(w Q(w)),
#   w = min(|v|, T),  Q ~ log2(erfc(w / sqrt 2)) / w on [0, T], T = 5.5
# Round 3: degree 5, minimax on the GELU error (round 2: degree 9, uniform in Q).
T = 5.5
S2 = np.sqrt(2.0)


def qfun(w):
    with np.errstate(divide='ignore', invalid='ignore'):
        r = np.log2(special.erfc(w / S2)) / w
    return np.where(w < 1e-9, -2 / np.sqrt(np.pi) / np.log(2) / S2, r)


def fit_gelu_weighted(deg, iters=200, n=8000):
    """Minimax fit of Q on [0, T] with the weight d gelu / d Q = 0.5 w^2 erfc(w/sqrt 2) ln 2
    (Lawson's iteratively reweighted least squares): the error that matters is the GELU's."""
    x = np.cos(np.pi * (np.arange(n) + 0.5) / n) * T / 2 + T / 2
    y = qfun(x)
    wgt = 0.5 * x * x * special.erfc(x / S2) * np.log(2) + 1e-13
    A = np.vander(x / T, deg + 1, increasing=True)
    lw = np.ones_like(x)
    for _ in range(iters):
        sw = np.sqrt(lw) * wgt
        c, *_ = np.linalg.lstsq(A * sw[:, None], y * sw, rcond=None)
        r = np.abs((A @ c - y) * wgt)
        lw = lw * (r + 1e-30)
        lw /= lw.sum()
    return c / (T ** np.arange(deg + 1))


def gelu_err(Q):
    v = np.concatenate([np.linspace(-12, 12, 4000001), np.linspace(-1e-2, 1e-2, 200001)]).astype(f32)
    w = np.minimum(np.abs(v), f32(T)).astype(f32)
    e = np.exp2((w * horner32(Q, w) - f32(1)).astype(f32)).astype(f32)        # 0.5 erfc
    g = (np.maximum(v, f32(0)) - (np.abs(v) * e).astype(f32)).astype(f32)
    g_ref = 0.5 * v.astype(np.float64) * (1 + special.erf(v.astype(np.float64) / S2))
    err = np.abs(g.astype(np.float64) - g_ref)
    return err.max(), v[err.argmax()], err[np.abs(v) < 2].max()


Q9 = fit(qfun, 0.0, T, 9, n=6000)                      # round 2: uniform fit of Q, degree 9
print('uniform degree-9 Q:   gelu max abs err %.3e at v=%.3f; |v|<2: %.3e' % gelu_err(Q9))
for deg in (4, 5, 6):
    Qd = fit_gelu_weighted(deg)
    print('weighted degree-%d Q: gelu max abs err %.3e at v=%.3f; |v|<2: %.3e' % ((deg,) + gelu_err(Qd)))
Q = fit_gelu_weighted(5)                               # csrc/common.h: GELU_Q
print('Q =', ', '.join('%.9ef' % c for c in Q))
