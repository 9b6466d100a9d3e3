"""Acquisition optimiser of the MI355X path, mirroring the class name and ``optimize`` contract of
/root/reference/src/utils_functions/causal_optimizer.py:15-71.

The reference samples ``num_anchor_points`` uniform random anchors, scores them with one batched
``acquisition.evaluate`` (:52-55), keeps the best and refines it with L-BFGS (:59-65).  Here the batched
scoring IS the optimiser: a deterministic regular grid over the space (BASELINE.json's "N-candidate sweep",
SURVEY.md §0.7), scored on the GPU, arg-max taken on the GPU.  Gradient refinement is SURVEY.md §8 f3.

``anchors="uniform"`` is the reference's own mode, for use under runCBO.py where the trajectory of numpy's global
generator matters (``--seed``, src/ArgumentParser.py:47): ``num_anchor_points`` (100) uniform anchors drawn from the
GLOBAL generator in emukit's order, scored by one batched ``acquisition.evaluate`` (one device sweep), the top one by
``argsort()[::-1][:1]`` as emukit takes it, then L-BFGS from that anchor (scipy ``fmin_l_bfgs_b``, bounds, maxfun=1000 --
emukit's ``OptLbfgs``), whose result is returned as it comes (:59-65: no comparison with the anchor's own value).
emukit is absent here: its order of draws -- ``ParameterSpace.sample_uniform`` samples parameter by
parameter, each ``np.random.uniform(low, high, (n, 1))`` -- is recalled from emukit 0.4.10, parity unpinned.
"""
from __future__ import annotations

import numpy as np

from ..graphs import meshgrid_candidates
from .causal_acquisition_functions import CandidateGrid


class CausalGradientAcquisitionOptimizer:
    def __init__(self, space, num_anchor_points=None, grid_shape=None, anchors="grid"):
        """``space``: emukit ParameterSpace / object with ``parameters`` / list of (lo, hi).  ``anchors="grid"``
        (default): ``grid_shape`` (points per dimension) defaults to ``default_grid_shape``; ``num_anchor_points``, if
        given, is used as the candidate budget instead of 16384.  ``anchors="uniform"``: the reference's 100 (or
        ``num_anchor_points``) uniform random anchors and L-BFGS from the best (module docstring)."""
        from .utils import default_grid_shape, space_bounds
        if anchors not in ("grid", "uniform"):
            raise ValueError(f"anchors must be 'grid' or 'uniform', not {anchors!r}")
        self.space = space
        self.bounds = space_bounds(space)
        self.anchors = anchors
        self.num_anchor_points = 100 if num_anchor_points is None else int(num_anchor_points)    # (causal_optimizer.py:19)
        self.grid_shape = list(grid_shape) if grid_shape is not None else default_grid_shape(
            len(self.bounds), budget=num_anchor_points)
        self._grid, self._grid_model = None, None

    def sample_uniform(self, point_count):
        """emukit ``ParameterSpace.sample_uniform`` on the GLOBAL numpy generator: one ``np.random.uniform(low, high,
        (point_count, 1))`` per parameter, in the parameters' order, stacked as columns."""
        return np.hstack([np.random.uniform(low=lo, high=hi, size=(point_count, 1)) for lo, hi in self.bounds])

    def optimize_from_uniform_anchors(self, acquisition):
        """The reference's ``_optimize`` (:26-65) for a space without context or constraints."""
        x_anchors = self.sample_uniform(self.num_anchor_points)
        scores = np.asarray(acquisition.evaluate(x_anchors), dtype=np.float64)[:, 0]      # ONE batched device sweep (:52-55)
        anchor = x_anchors[np.argsort(scores)[::-1][:1]]                                # emukit AnchorPointsGenerator.get
        return self.refine(acquisition, anchor)                                         # OptLbfgs from it (:59-65)

    def candidates(self):
        return meshgrid_candidates(self.bounds, self.grid_shape)

    def refine(self, acquisition, x0):
        """The reference's second stage (causal_optimizer.py:59-65; emukit OptLbfgs = scipy fmin_l_bfgs_b with
        bounds and maxfun=1000, minimising -acquisition with its gradient) started from ``x0`` ((1,d))."""
        from scipy.optimize import fmin_l_bfgs_b

        def f_df(v):
            f, df = acquisition.evaluate_with_gradients(v[None, :])
            return -float(f[0, 0]), -df[0]

        x, fx, _ = fmin_l_bfgs_b(f_df, np.asarray(x0, dtype=np.float64).reshape(-1), bounds=self.bounds, maxfun=1000)
        return x[None, :], np.array([[-fx]])

    def refine_batched(self, acquisition, x0, max_rounds=200, history=8):
        """Multi-start refinement ("batched multi-start L-BFGS from top-k grid points"): every row of
        ``x0`` ((k,d)) is the start of its own projected L-BFGS ascent, and the k searches advance in lockstep -- one
        ``evaluate_with_gradients`` call on a (k,d) batch per round (one pair of batched device solves), whatever each
        search is doing (a first trial step, a backtracked one).  Returns (points (k,d), values (k,))."""
        lo = np.array([b[0] for b in self.bounds], dtype=np.float64)
        hi = np.array([b[1] for b in self.bounds], dtype=np.float64)
        return lockstep_lbfgs(lambda X: _values_and_gradients(acquisition, X), np.asarray(x0, dtype=np.float64), lo, hi,
                              max_rounds=max_rounds, history=history)

    def optimize(self, acquisition, context=None, refine=False, num_starts=1):
        """(x_max (1,d), acquisition value at x_max (1,1)) -- emukit ``AcquisitionOptimizerBase.optimize``.
        ``acquisition`` is ``CausalExpectedImprovement(...) / Cost(...)`` (an ``AcquisitionQuotient``) or a bare
        ``CausalExpectedImprovement``.  ``refine=True`` adds the reference's gradient stage: from the best grid point
        with scipy's L-BFGS-B (``num_starts=1``, what the reference does with its best anchor), or from the
        ``num_starts`` best grid points at once (``refine_batched``)."""
        if self.anchors == "uniform":
            return self.optimize_from_uniform_anchors(acquisition)
        # the grid of this optimiser stays on the device while the model object is the same (no allocation per call)
        model = acquisition.model
        if self._grid is None or self._grid_model is not model:
            if self._grid is not None:
                self._grid.close()
            self._grid, self._grid_model = CandidateGrid(self.candidates(), model), model
        grid, pts = self._grid, self._grid.points
        k = max(1, min(int(num_starts), pts.shape[0])) if refine else 1
        res = acquisition.sweep(grid, want_acq=k > 1)
        x = pts[res["best_idx"]][None, :].copy()
        fx = np.array([[res["best_val"]]])
        if refine and k == 1:
            xr, fr = self.refine(acquisition, x)
            if fr[0, 0] >= fx[0, 0]:
                x, fx = xr, fr
        elif refine:
            acq = np.asarray(res["acq"], dtype=np.float64).reshape(-1)
            top = np.argpartition(-acq, k - 1)[:k]
            top = top[np.argsort(-acq[top], kind="stable")]
            xs, fs = self.refine_batched(acquisition, pts[top])
            best = int(np.argmax(fs))
            if fs[best] >= fx[0, 0]:
                x, fx = xs[best][None, :].copy(), np.array([[fs[best]]])
        return x, fx


def _values_and_gradients(acquisition, X):
    """(values (k,), gradients (k,d)) of every row of X taken as a batch of ONE point: the posterior and its gradients
    for all rows come from one batched device call; a cost that depends on the point (cost_functions.py:11-17 sums
    |x| over the batch it is given) is evaluated row by row, as k single-point calls of the reference would."""
    num = getattr(acquisition, "numerator", None)
    if num is None:
        f, df = acquisition.evaluate_with_gradients(X)
        return f[:, 0], df
    f, df = num.evaluate_with_gradients(X)
    c = np.array([float(acquisition.denominator.evaluate(X[i:i + 1])) for i in range(X.shape[0])])
    return f[:, 0] / c, df / c[:, None]


def lockstep_lbfgs(fun, x0, lo, hi, max_rounds=200, history=8, pgtol=1e-5, ftol=2.2e-9, c1=1e-4):
    """Maximise ``fun`` over the box [lo, hi] from every row of ``x0`` at once.  ``fun(X)`` -> (values (k,),
    gradients (k,d)) is called ONCE per round on the (k,d) array of the searches' current trial points.  Each search
    is a projected L-BFGS iteration with Armijo backtracking (two-loop recursion over ``history`` pairs, steps
    clipped to the box, curvature pairs with s.y <= 0 dropped); a search that has converged (projected gradient below
    ``pgtol``, relative gain below ``ftol`` -- scipy's L-BFGS-B defaults -- or a vanishing step) keeps its point and
    is re-evaluated along with the others (the batch shape stays fixed).  Returns (points (k,d), values (k,))."""
    X = np.clip(np.asarray(x0, dtype=np.float64).copy(), lo, hi)
    k, d = X.shape
    F, G = fun(X)
    F, G = -np.asarray(F, dtype=np.float64), -np.asarray(G, dtype=np.float64)          # minimise -fun
    S = [[] for _ in range(k)]
    Y = [[] for _ in range(k)]
    done = np.zeros(k, dtype=bool)
    D = np.zeros((k, d))
    T = np.zeros(k)

    def projected_gradient(x, g):
        pg = g.copy()
        pg[(x <= lo) & (g > 0)] = 0.0                # cannot go below the lower bound
        pg[(x >= hi) & (g < 0)] = 0.0
        return pg

    def direction(i):
        pg = projected_gradient(X[i], G[i])
        q = pg.copy()
        alphas = []
        for s, y in zip(reversed(S[i]), reversed(Y[i])):
            a = s.dot(q) / y.dot(s)
            alphas.append(a)
            q -= a * y
        if S[i]:
            q *= S[i][-1].dot(Y[i][-1]) / Y[i][-1].dot(Y[i][-1])
        for (s, y), a in zip(zip(S[i], Y[i]), reversed(alphas)):
            q += (a - y.dot(q) / y.dot(s)) * s
        dvec = -q
        if dvec.dot(pg) >= 0.0:                      # not a descent direction (stale curvature at a bound)
            dvec = -pg
            S[i].clear(); Y[i].clear()
        return dvec, pg

    for i in range(k):
        D[i], pg = direction(i)
        n = np.linalg.norm(pg)
        done[i] = np.max(np.abs(pg)) <= pgtol
        T[i] = min(1.0, 1.0 / n) if n > 0 else 0.0
    for _ in range(max_rounds):
        if done.all():
            break
        trial = np.where(done[:, None], X, np.clip(X + T[:, None] * D, lo, hi))
        Ft, Gt = fun(trial)
        Ft, Gt = -np.asarray(Ft, dtype=np.float64), -np.asarray(Gt, dtype=np.float64)
        for i in range(k):
            if done[i]:
                continue
            step = trial[i] - X[i]
            if not np.isfinite(Ft[i]) or Ft[i] > F[i] + c1 * G[i].dot(step):
                T[i] *= 0.5                          # Armijo failed: shorter step next round
                if T[i] * np.max(np.abs(D[i])) < 1e-14:
                    done[i] = True
                continue
            y = Gt[i] - G[i]
            gain = F[i] - Ft[i]
            if step.dot(y) > 1e-12 * np.linalg.norm(step) * np.linalg.norm(y):
                S[i].append(step.copy()); Y[i].append(y)
                if len(S[i]) > history:
                    S[i].pop(0); Y[i].pop(0)
            small = gain <= ftol * max(abs(F[i]), abs(Ft[i]), 1.0)
            X[i], F[i], G[i] = trial[i], Ft[i], Gt[i]
            D[i], pg = direction(i)
            T[i] = 1.0
            if small or np.max(np.abs(pg)) <= pgtol:
                done[i] = True
    return X, -F

