"""Oracle: GEM gradient memory + projection (numpy float64, as the reference computes it on the CPU).

Restates rehearsal/model/gem.py:20-35 (store_grad), :38-55 (overwrite_grad), :58-80 (project2cone2)
and :275-277 (violation test).  The QP itself lives in the un-vendored quadprog==0.1.6
(requirements.txt:51): PARITY UNPINNED — here it is solved by exhaustive active-set enumeration
(t <= 9 unknowns => <= 512 candidate active sets), which is exact for a strictly convex QP and shares no
code with the product's Goldfarb-Idnani solver.
"""
import itertools

import numpy as np


def store_grad(grads_list, G, tid):
    """gem.py:20-35: flatten all parameter gradients into column tid of G[P][n_tasks]."""
    G = G.copy()
    G[:, tid] = 0.0
    G[:, tid] = np.concatenate([g.reshape(-1) for g in grads_list])
    return G


def overwrite_grad(newgrad, shapes):
    """gem.py:38-55: scatter a flat vector back to parameter shapes."""
    out, off = [], 0
    for s in shapes:
        n = int(np.prod(s))
        out.append(newgrad[off:off + n].reshape(s))
        off += n
    return out


def solve_bound_qp(P, q, margin):
    """min 1/2 v^T P v - q^T v  s.t. v >= margin (I^T v >= h), by enumeration of active sets."""
    t = P.shape[0]
    best = None
    for k in range(t + 1):
        for act in itertools.combinations(range(t), k):
            act = list(act)
            free = [i for i in range(t) if i not in act]
            v = np.full(t, float(margin))
            if free:
                rhs = q[free] - P[np.ix_(free, act)] @ v[act] if act else q[free]
                v[free] = np.linalg.solve(P[np.ix_(free, free)], rhs)
            grad = P @ v - q                      # = multipliers on active bounds (must be >= 0)
            if np.all(v[free] >= margin - 1e-12) and np.all(grad[act] >= -1e-10):
                f = 0.5 * v @ P @ v - q @ v
                if best is None or f < best[0]:
                    best = (f, v)
    return best[1]


def project2cone2(gradient, memories, margin=0.5, eps=1e-3):
    """gem.py:58-80. gradient [P], memories [P][t] (float32 in, float64 inside) -> projected gradient [P] f32."""
    M = memories.T.astype(np.float64)
    g = gradient.reshape(-1).astype(np.float64)
    t = M.shape[0]
    P = M @ M.T
    P = 0.5 * (P + P.T) + np.eye(t) * eps
    q = -(M @ g)
    v = solve_bound_qp(P, q, margin)
    x = v @ M + g
    return x.astype(np.float32), v


def violations(G, t, past):
    """gem.py:275-277 (float32 mm)."""
    dotp = G[:, t].astype(np.float32) @ G[:, past].astype(np.float32)
    return dotp, int((dotp < 0).sum())
