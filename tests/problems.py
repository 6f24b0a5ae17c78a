"""Small end-to-end NLPs used for boundary (API) parity.

Textbook problems -- the same set the reference exercises in
ipsolver/tests/test_minimized_constrained.py:16-329 -- written here from
their mathematical statements.  ``constraints(ns)`` builds the constraint
objects from a namespace so the golden generator can feed the identical
inputs to the reference while the tests feed them to this package.
"""
import numpy as np
import scipy.sparse as sps
from scipy.linalg import block_diag


class Problem:
    x_opt = None
    hess_mode = None      # None -> use self.hess; else a finite-difference tag

    def hess_arg(self):
        return self.hess_mode if self.hess_mode is not None else self.hess

    def constraints(self, ns):
        return ()


class Maratos(Problem):
    """Nocedal & Wright problem 15.4: min 2(x0^2+x1^2-1) - x0 on the circle."""
    name = "maratos"
    x_opt = np.array([1.0, 0.0])

    def __init__(self, degrees=60):
        t = degrees / 180 * np.pi
        self.x0 = [np.cos(t), np.sin(t)]

    def fun(self, x):
        return 2 * (x[0] ** 2 + x[1] ** 2 - 1) - x[0]

    def grad(self, x):
        return np.array([4 * x[0] - 1, 4 * x[1]])

    def hess(self, x):
        return 4 * np.eye(2)

    def constraints(self, ns):
        # The reference test supplies this (inexact) Jacobian 4x for the
        # circle x'x; kept so both solvers see the same callbacks.
        return ns.NonlinearConstraint(
            lambda x: x[0] ** 2 + x[1] ** 2, ("equals", 1),
            lambda x: [[4 * x[0], 4 * x[1]]],
            lambda x, v: 2 * v[0] * np.eye(2))


class HyperbolicIneq(Problem):
    """Nocedal & Wright problem 15.1 (the README example)."""
    name = "hyperbolic_ineq"
    x0 = [0, 0]
    x_opt = [1.952823, 0.088659]

    def fun(self, x):
        return 0.5 * (x[0] - 2) ** 2 + 0.5 * (x[1] - 0.5) ** 2

    def grad(self, x):
        return [x[0] - 2, x[1] - 0.5]

    def hess(self, x):
        return np.eye(2)

    def constraints(self, ns):
        nl = ns.NonlinearConstraint(
            lambda x: 1 / (x[0] + 1) - x[1], ("greater", 0.25),
            lambda x: [[-1 / (x[0] + 1) ** 2, -1]],
            lambda x, v: 2 * v[0] * np.array([[1 / (x[0] + 1) ** 3, 0],
                                              [0, 0]]))
        return (nl, ns.BoxConstraint(("greater",)))


class Rosenbrock(Problem):
    name = "rosenbrock"

    def __init__(self, n=2, random_state=0):
        self.x0 = np.random.RandomState(random_state).uniform(-1, 1, n)
        self.x_opt = np.ones(n)
        self.name = "rosenbrock%d" % n

    def fun(self, x):
        x = np.asarray(x)
        return np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2)

    def grad(self, x):
        x = np.asarray(x)
        g = np.zeros_like(x)
        g[:-1] += -400 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
        g[1:] += 200 * (x[1:] - x[:-1] ** 2)
        return g

    def hess(self, x):
        x = np.atleast_1d(x)
        off = -400 * x[:-1]
        dg = np.zeros(len(x))
        dg[:-1] = 1200 * x[:-1] ** 2 - 400 * x[1:] + 2
        dg[1:] += 200
        return np.diag(dg) + np.diag(off, 1) + np.diag(off, -1)


class IneqRosenbrock(Rosenbrock):
    """min rosenbrock s.t. x0 + 2 x1 <= 1 (fmincon documentation example)."""

    def __init__(self):
        Rosenbrock.__init__(self, 2)
        self.name = "ineq_rosenbrock"
        self.x0 = [-1, -0.5]
        self.x_opt = [0.5022, 0.2489]

    def constraints(self, ns):
        return ns.LinearConstraint([[1, 2]], ("less", 1))


class EqIneqRosenbrock(Rosenbrock):
    """... and additionally 2 x0 + x1 = 1."""

    def __init__(self):
        Rosenbrock.__init__(self, 2)
        self.name = "eq_ineq_rosenbrock"
        self.x0 = [-1, -0.5]
        self.x_opt = [0.41494, 0.17011]

    def constraints(self, ns):
        return (ns.LinearConstraint([[1, 2]], ("less", 1)),
                ns.LinearConstraint([[2, 1]], ("equals", 1)))


class Elec(Problem):
    """COPS problem 2: Coulomb energy of electrons kept inside the unit ball
    (sparse Jacobian)."""

    def __init__(self, n_electrons=10, random_state=0):
        self.ne = n_electrons
        self.name = "elec%d" % n_electrons
        rng = np.random.RandomState(random_state)
        phi = rng.uniform(0, 2 * np.pi, self.ne)
        theta = rng.uniform(-np.pi, np.pi, self.ne)
        self.x0 = np.hstack((np.cos(theta) * np.cos(phi),
                             np.cos(theta) * np.sin(phi), np.sin(theta)))

    def _deltas(self, x):
        P = np.reshape(x, (3, self.ne))
        return [c[:, None] - c for c in P]

    def fun(self, x):
        dx, dy, dz = self._deltas(x)
        with np.errstate(divide='ignore'):
            inv = (dx ** 2 + dy ** 2 + dz ** 2) ** -0.5
        np.fill_diagonal(inv, 0)
        return 0.5 * np.sum(inv)

    def grad(self, x):
        dx, dy, dz = self._deltas(x)
        with np.errstate(divide='ignore'):
            inv3 = (dx ** 2 + dy ** 2 + dz ** 2) ** -1.5
        np.fill_diagonal(inv3, 0)
        return np.hstack([-np.sum(dd * inv3, axis=1) for dd in (dx, dy, dz)])

    def hess(self, x):
        D = self._deltas(x)
        dist = (D[0] ** 2 + D[1] ** 2 + D[2] ** 2) ** 0.5
        with np.errstate(divide='ignore'):
            inv3, inv5 = dist ** -3, dist ** -5
        np.fill_diagonal(inv3, 0)
        np.fill_diagonal(inv5, 0)

        def block(a, b):
            B = -3 * D[a] * D[b] * inv5 + (inv3 if a == b else 0)
            B[np.diag_indices(self.ne)] = -np.sum(B, axis=1)
            return B

        return np.block([[block(a, b) for b in range(3)] for a in range(3)])

    def constraints(self, ns):
        ne = self.ne

        def fun(x):
            return np.sum(np.reshape(x, (3, ne)) ** 2, axis=0) - 1

        def jac(x):
            P = np.reshape(x, (3, ne))
            return sps.csc_matrix(np.hstack([2 * np.diag(c) for c in P]))

        def hess(x, v):
            D = 2 * np.diag(v)
            return block_diag(D, D, D)

        return ns.NonlinearConstraint(fun, ("less",), jac, hess)


def _with_fd(cls, tag, suffix):
    class FD(cls):
        hess_mode = tag

        def __init__(self, *a, **k):
            cls.__init__(self, *a, **k)
            self.name = self.name + suffix
    FD.__name__ = cls.__name__ + suffix
    return FD


def exact_hessian_problems():
    return [Maratos(), HyperbolicIneq(), Rosenbrock(2), Rosenbrock(10),
            IneqRosenbrock(), EqIneqRosenbrock(), Elec(10)]


def fd_hessian_problems():
    return [_with_fd(Maratos, '3-point', "_fd3")(),
            _with_fd(HyperbolicIneq, '3-point', "_fd3")(),
            _with_fd(Maratos, '2-point', "_fd2")(),
            _with_fd(HyperbolicIneq, '2-point', "_fd2")(),
            _with_fd(Rosenbrock, '2-point', "_fd2")(10),
            _with_fd(EqIneqRosenbrock, '2-point', "_fd2")(),
            _with_fd(Elec, '2-point', "_fd2")(10)]


# ---- device-callback twins (x is a CUDA tensor; see ipsolver/device_mode.py) ----------
def _row_csr(values):
    """1 x n Jacobian as a DeviceCSR on a fixed (cached) pattern."""
    import torch
    from ipsolver.device import CSRPattern, DeviceCSR
    n = len(values)
    pat = _row_csr.patterns.get(n)
    if pat is None:
        pat = _row_csr.patterns[n] = CSRPattern(np.array([0, n], dtype=np.int32),
                                                np.arange(n, dtype=np.int32), (1, n))
    return DeviceCSR(pat, torch.stack(list(values)).to(torch.float64))


_row_csr.patterns = {}


class DeviceMaratos(Maratos):
    """Maratos with torch callbacks (same expressions as the host problem)."""

    def device_x0(self):
        import torch
        return torch.tensor(self.x0, dtype=torch.float64, device="cuda")

    def fun(self, x):
        return float(2 * (x[0] ** 2 + x[1] ** 2 - 1) - x[0])

    def grad(self, x):
        import torch
        return torch.stack([4 * x[0] - 1, 4 * x[1]])

    def constraints(self, ns):
        import torch
        return ns.NonlinearConstraint(
            lambda x: (x[0] ** 2 + x[1] ** 2).reshape(1), ("equals", 1),
            lambda x: _row_csr([4 * x[0], 4 * x[1]]),
            lambda x, v: torch.stack([2 * v[0], 2 * v[0]]))


class DeviceHyperbolicIneq(HyperbolicIneq):
    def device_x0(self):
        import torch
        return torch.tensor(self.x0, dtype=torch.float64, device="cuda")

    def fun(self, x):
        return float(0.5 * (x[0] - 2) ** 2 + 0.5 * (x[1] - 0.5) ** 2)

    def grad(self, x):
        import torch
        return torch.stack([x[0] - 2, x[1] - 0.5])

    def constraints(self, ns):
        import torch
        nl = ns.NonlinearConstraint(
            lambda x: (1 / (x[0] + 1) - x[1]).reshape(1), ("greater", 0.25),
            lambda x: _row_csr([-1 / (x[0] + 1) ** 2, -torch.ones_like(x[0])]),
            lambda x, v: torch.stack([2 * v[0] / (x[0] + 1) ** 3, torch.zeros_like(x[0])]))
        return (nl, ns.BoxConstraint(("greater",)))


class SparseBarrierQP(Problem):
    """Separable convex QP under linear inequalities with a Jacobian of RANDOM sparsity (``per``
    entries per row at random columns: no band, no reordering makes J J' narrow) and a box on
    every variable -- the barrier problem shape of BASELINE config 5 without its band.  The
    reference factors the augmented system of any pattern with SuperLU (projections.py:93-172);
    here it exercises the box-Schur elimination over a dense / iterative Schur solve."""
    name = "sparse_barrier_qp"

    def __init__(self, n=1200, m=800, per=4, seed=0):
        rng = np.random.default_rng(seed)
        cols = np.concatenate([rng.choice(n, per, replace=False) for _ in range(m)])
        J = sps.csr_matrix((rng.standard_normal(m * per), cols, np.arange(0, m * per + 1, per)),
                           shape=(m, n))
        J.sort_indices()
        self.n, self.m, self.J = n, m, J
        self.d = rng.uniform(1.0, 2.0, n)
        self.q = rng.standard_normal(n)
        self.x0 = rng.uniform(-0.5, 0.5, n)
        self.bnd = J.dot(self.x0) + rng.uniform(0.0, 0.5, m)      # strictly feasible at x0
        self.H = sps.diags(self.d, format="csr")

    def fun(self, x):
        return 0.5 * x.dot(self.d * x) - self.q.dot(x)

    def grad(self, x):
        return self.d * x - self.q

    def hess(self, x):
        return self.H

    def constraints(self, ns):
        return (ns.LinearConstraint(self.J, ("less", self.bnd)),
                ns.BoxConstraint(("interval", -0.8, 0.8)))
