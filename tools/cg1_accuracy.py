#!/usr/bin/env python3
"""
Attainable accuracy of the single-reduction (Chronopoulos-Gear) CG of stencil_march.hpp MODE_CG1 in fp32. Runs the two-launch recurrence
(PhiML's cg) and two closures of the single-reduction recurrence in NumPy float32 on the closed-box 2-D Poisson problem of the smoke-plume
configurations (5-point Neumann Laplacian, localized dipole rhs, true-residual refresh every 50 iterations like PhiML) and prints the
TRUE relative residual |y - A x| / |y| (evaluated in float64) they reach.   python tools/cg1_accuracy.py 512 2500
  textbook : p.Ap = delta - beta gamma / alpha_prev         (rounds 1-2: stalls 1-2 digits early -- the closure assumes exact orthogonality)
  five_sum : p.Ap = delta + beta (mu + nu) + beta^2 sigma   (r3: an identity of the stored vectors; what the kernels compute now)
"""
import sys

import numpy as np


def laplace_neumann(p, h2):
    q = np.pad(p, 1, mode='edge')
    return ((q[2:, 1:-1] - p) - (p - q[:-2, 1:-1]) + (q[1:-1, 2:] - p) - (p - q[1:-1, :-2])) / h2


def solve(n, iters, variant, dtype=np.float32, refresh=50):
    h2 = dtype((100.0 / n) ** 2)
    A = lambda p: laplace_neumann(p, h2)
    y = np.zeros((n, n), dtype)
    y[n // 2 - 5:n // 2 + 5, 5:15], y[n // 2 - 5:n // 2 + 5, 15:25] = 0.1, -0.1
    y -= y.mean(dtype=dtype)
    ysq = float((y.astype(np.float64) ** 2).sum())
    true_res = lambda x: np.sqrt(float(((y.astype(np.float64) - laplace_neumann(x.astype(np.float64), float(h2))) ** 2).sum()) / ysq)
    x = np.zeros_like(y)
    hist = []
    if variant == "two_launch":
        r = y - A(x); d = r.copy(); q = A(d); rsq = (r * r).sum(dtype=dtype)
        for k in range(1, iters + 1):
            al = dtype(rsq / (d * q).sum(dtype=dtype))
            x = x + al * d
            r = y - A(x) if k % refresh == 0 else r - al * q
            rn = (r * r).sum(dtype=dtype); be = dtype(rn / rsq); rsq = rn
            d = r + be * d; q = A(d)
            hist.append(true_res(x))
    else:
        dsum = lambda a, b: float((a.astype(np.float64) * b).sum())          # the kernels add rows in fp32, rows and planes in double
        r = y - A(x); w = A(r); g = dsum(r, r); dl = dsum(w, r)
        p = np.zeros_like(y); s = np.zeros_like(y); al = 0.0; g_old = None; mu = nu = sg = 0.0
        for k in range(1, iters + 1):
            be = 0.0 if g_old is None else g / g_old
            if g_old is None: den = dl
            elif variant == "textbook": den = dl - be * g / al
            else: den = dl + be * (mu + nu) + be * be * sg
            al = g / den
            p = r + dtype(be) * p; s = w + dtype(be) * s; x = x + dtype(al) * p; r = r - dtype(al) * s
            g_old = g
            if k % refresh == 0:
                r = y - A(x)
            w = A(r); g = dsum(r, r); dl = dsum(w, r); mu = dsum(r, s); nu = dsum(w, p); sg = dsum(p, s)
            hist.append(true_res(x))
    return np.asarray(hist)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
    for variant in ("two_launch", "textbook", "five_sum"):
        h = solve(n, iters, variant)
        first = lambda t: int(np.argmax(h < t)) + 1 if (h < t).any() else None
        print(f"{n}^2 fp32 {variant:10s} floor {h.min():.3e} at iteration {h.argmin() + 1}; after {iters}: {h[-1]:.3e}; first < 1e-3: {first(1e-3)}; first < 1e-4: {first(1e-4)}")
