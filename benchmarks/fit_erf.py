"""The polynomial of salience_head_core.h erf_pos: erf(t) = 1 - 2^(-t p(t)); weighted least squares towards the minimax of the
erf error, evaluated in emulated fp32 Horner arithmetic.  python benchmarks/fit_erf.py  (degree 7 is the one in use)."""
import numpy as np
from scipy import special
# erf(t) = 1 - 2^(-(t * p(t))) on [0, T]; fit q(t) = -log2(erfc(t)) / t by weighted least squares, weight ~ erfc(t) * t * ln2 (error of erf per error of q)
T = 4.0
t = np.concatenate([np.linspace(1e-6, 0.5, 20000), np.linspace(0.5, T, 60000)])
q = -np.log2(special.erfc(t)) / t
w = special.erfc(t) * t * np.log(2.0)
best = None
for deg in (7, 8, 9, 10, 11):
    # iteratively reweighted to approximate minimax on erf error
    ww = w.copy()
    for it in range(60):
        V = np.vander(t, deg + 1, increasing=True)
        c, *_ = np.linalg.lstsq(V * ww[:, None], q * ww, rcond=None)
        err = (V @ c - q) * w          # ~ erf error (first order)
        ww = ww * (1 + 2.0 * np.abs(err) / np.abs(err).max())
    # evaluate in float32 arithmetic (Horner with fma-like float32 ops)
    c32 = c.astype(np.float32)
    tt = np.linspace(0, 6, 2000001).astype(np.float32)
    p = np.full_like(tt, c32[-1])
    for k in range(deg - 1, -1, -1):
        p = (p.astype(np.float64) * tt + c32[k]).astype(np.float32)   # fma rounding
    y = (-(tt.astype(np.float64) * p)).astype(np.float32)
    e = (1.0 - np.exp2(y.astype(np.float64))).astype(np.float32)
    ref = special.erf(tt.astype(np.float64))
    abs_err = np.abs(e.astype(np.float64) - ref)
    print(deg, "max abs err %.3g at t=%.3f" % (abs_err.max(), tt[abs_err.argmax()]), "coeffs", [float(x) for x in c32])
