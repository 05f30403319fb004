"""TEST INFRASTRUCTURE ONLY (never imported by the product: the product solves this QP on the device, clhip_gem_qp) —
Goldfarb-Idnani dual active-set solver for the small strictly convex QP of GEM, host float64.

The reference calls quadprog.solve_qp(P, q, G, h) (quadprog==0.1.6, not vendored; call site
rehearsal/model/gem.py:78), which implements D. Goldfarb & A. Idnani, "A numerically stable dual method
for solving strictly convex quadratic programs", Math. Programming 27 (1983).  This is a restatement of
that published algorithm in float64 numpy for the sizes GEM produces (t <= 9 unknowns):

    minimise 1/2 x^T G x - a^T x     subject to  C^T x >= b

Parity with quadprog itself is UNPINNED (the package is absent); tests check KKT residuals and agreement
with an exhaustive active-set enumeration and scipy.
"""
import numpy as np


def solve_qp(G, a, C, b, tol=1e-12, max_iter=200):
    G = np.asarray(G, dtype=np.float64)
    a = np.asarray(a, dtype=np.float64)
    C = np.asarray(C, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    n, m = G.shape[0], C.shape[1]
    L = np.linalg.cholesky(G)
    Linv = np.linalg.inv(L)
    Ginv = Linv.T @ Linv
    x = Ginv @ a                                  # unconstrained minimum
    active = []                                    # indices of active constraints
    u = np.zeros(0)                                # their multipliers
    for _ in range(max_iter):
        s = C.T @ x - b
        viol = [(s[i], i) for i in range(m) if i not in active and s[i] < -tol * max(1.0, abs(b[i]))]
        if not viol:
            lam = np.zeros(m)
            lam[active] = u
            return x, lam
        p = min(viol)[1]                           # most violated constraint
        np_ = C[:, p]
        u_plus = np.append(u, 0.0)
        while True:
            if active:
                N = C[:, active]
                Nstar = np.linalg.solve(N.T @ Ginv @ N, N.T @ Ginv)      # (N^T G^-1 N)^-1 N^T G^-1
                H = Ginv - Ginv @ N @ Nstar
                z = H @ np_
                r = Nstar @ np_
            else:
                z = Ginv @ np_
                r = np.zeros(0)
            # partial step length t1 (dual feasibility), full step length t2 (primal)
            t1, drop = np.inf, -1
            for j in range(len(active)):
                if r[j] > tol and u_plus[j] / r[j] < t1:
                    t1, drop = u_plus[j] / r[j], j
            zn = float(z @ np_)
            t2 = -float(s[p]) / zn if zn > tol else np.inf
            t = min(t1, t2)
            if not np.isfinite(t):
                raise ValueError("QP infeasible")
            if np.isfinite(t2):
                x = x + t * z
            u_plus[:len(active)] -= t * r
            u_plus[-1] += t
            if t == t2:                            # full step: constraint p becomes active
                active.append(p)
                u = u_plus
                break
            # partial step: drop constraint `drop`, recompute and continue with the same p
            del active[drop]
            u_plus = np.delete(u_plus, drop)
            s = C.T @ x - b
    raise RuntimeError("solve_qp: iteration limit")


def project2cone2_coefficients(gram, t_index, mem_indices, margin, eps=1e-3):
    """QP of gem.py:58-80 from the Gram matrix of [memory rows..., current row]:
    P = 1/2 (MM^T + (MM^T)^T) + eps I ; q = -M g ; min 1/2 v^T P v - q^T v ... s.t. v >= margin.
    Returns v (len(mem_indices))."""
    gram = np.asarray(gram, dtype=np.float64)
    MMt = gram[np.ix_(mem_indices, mem_indices)]
    t = len(mem_indices)
    P = 0.5 * (MMt + MMt.T) + np.eye(t) * eps
    q = -gram[mem_indices, t_index]
    v, _ = solve_qp(P, q, np.eye(t), np.zeros(t) + margin)
    return v
