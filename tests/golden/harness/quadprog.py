"""Stand-in for quadprog.solve_qp (un-vendored dependency, quadprog==0.1.6).
Solves  min 1/2 x^T G x - a^T x  s.t.  C^T x >= b  with scipy (SLSQP, tight tol).
Used only to let the reference's gem.py import/run in the dev container."""
import numpy as np
from scipy.optimize import minimize


def solve_qp(G, a, C, b, meq=0):
    n = G.shape[0]
    cons = [{"type": "ineq", "fun": lambda x, C=C, b=b: C.T @ x - b,
             "jac": lambda x, C=C: C.T}]
    x0 = np.linalg.solve(G, a)
    x0 = np.maximum(x0, b + 0.0) if np.allclose(C, np.eye(n)) else x0
    r = minimize(lambda x: 0.5 * x @ G @ x - a @ x, x0,
                 jac=lambda x: G @ x - a, constraints=cons, method="SLSQP",
                 options={"ftol": 1e-15, "maxiter": 1000})
    return (r.x, r.fun, x0, r.nit, None, None)
