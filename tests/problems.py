"""Seeded synthetic inputs for the BASELINE.json configurations.

Pure numpy; shared by the golden-fixture generator, the parity tests and
bench.py so that every leg sees the same problems.  Env parameter dictionaries
use the reference's JSON config keys (``tfmpc/envs/*/*.config.json``).
"""

import numpy as np


# --------------------------------------------------------------------- LQR ---
def make_lqr_instance(seed, n, m):
    """cfg1 / cfg3 instance: ``np.random.seed(seed)`` then the draw order of the
    reference's ``make_lqr`` (``tfmpc/envs/__init__.py:9-18``): F, f,
    ``make_spd_matrix``, c.  Returns float64 arrays, vectors flattened."""
    from sklearn.datasets import make_spd_matrix

    d = n + m
    np.random.seed(seed)
    F = np.random.normal(size=(n, d))
    f = np.random.normal(size=(n, 1))
    C = make_spd_matrix(d)
    c = np.random.normal(size=(d, 1))
    return F, f[:, 0], C, c[:, 0]


def make_lqr_batch(B, n, m, seed0=1000, x0_seed=7):
    """cfg3: instance i uses seed ``seed0 + i``; ``x0 ~ N(0,1)^n``."""
    Fs, fs, Cs, cs = zip(*[make_lqr_instance(seed0 + i, n, m) for i in range(B)])
    x0 = np.random.default_rng(x0_seed).normal(size=(B, n))
    return np.stack(Fs), np.stack(fs), np.stack(Cs), np.stack(cs), x0


def make_lqr_batch_fast(B, n, m, seed=0):
    """Fast vectorised generator for timing runs and large-batch tests: F, f, c ~ N(0,1) as in
    ``make_lqr``; C = U diag(1 + U(0,1)) U^T with U from the SVD of a random Gram matrix -- SPD
    and well conditioned (eigenvalues in [1, 2]).  It does NOT reproduce ``make_spd_matrix``'s
    spectrum (0.04 .. n+m); the seeded ``make_lqr_instance`` does and defines the parity cases."""
    rng = np.random.default_rng(seed)
    d = n + m
    F = rng.normal(size=(B, n, d))
    f = rng.normal(size=(B, n))
    c = rng.normal(size=(B, d))
    A = rng.uniform(size=(B, d, d))
    # make_spd_matrix: A = rand(d,d); U,_,Vt = svd(A^T A); X = U (1 + diag(rand(d))) Vt
    s = 1.0 + rng.uniform(size=(B, d))
    if B >= 4096:
        # the same matrices through the symmetric eigendecomposition (A^T A is symmetric positive definite: its left singular vectors ARE its
        # eigenvectors, descending order = eigh's order reversed): a third of the SVD's time; agrees with the SVD form to fp64 rounding.
        # (numpy, chunked: torch's batched CPU eigh spawns a thread per host core and took 110 s inside the GPU box's 16-CPU share)
        C = np.empty((B, d, d))
        for lo in range(0, B, 2048):
            Ac = A[lo:lo + 2048]
            _, V = np.linalg.eigh(np.matmul(np.ascontiguousarray(np.swapaxes(Ac, 1, 2)), Ac))
            V = V[:, :, ::-1]
            C[lo:lo + 2048] = np.matmul(V * s[lo:lo + 2048, None, :], np.ascontiguousarray(np.swapaxes(V, 1, 2)))
    else:
        U, _, Vt = np.linalg.svd(np.matmul(np.swapaxes(A, 1, 2), A))
        C = np.matmul(U * s[:, None, :], Vt)
    C = 0.5 * (C + np.swapaxes(C, 1, 2))
    x0 = rng.normal(size=(B, n))
    return F, f, C, c, x0


def make_lqr_batch_spd(B, n, m, seed=0, chunk=8192):
    """``make_lqr`` (``tfmpc/envs/__init__.py:9-18``) vectorised over the batch: F, f, c ~ N(0,1) and
    C by the very formula of sklearn's ``make_spd_matrix`` -- ``A ~ U(0,1)^{d x d}``, ``U, _, Vt =
    svd(A^T A)``, ``C = U (1.0 + diag(r)) Vt`` with ``r ~ U(0,1)^d``, where the ``1.0`` is added to EVERY
    entry of the middle factor, not just its diagonal.  Same distribution as ``make_lqr_instance``
    (eigenvalues of C from ~1e-3 .. 0.04 up to ~n+m, rho(F_x) ~ 5); the draws come from ``default_rng(seed)``
    instead of one global ``RandomState`` per instance, so the VALUES differ from the seeded instances.
    The generator of bench.py's headline batch and of the full-size parity tests."""
    rng = np.random.default_rng(seed)
    d = n + m
    F = rng.normal(size=(B, n, d))
    f = rng.normal(size=(B, n))
    c = rng.normal(size=(B, d))
    x0 = rng.normal(size=(B, n))
    C = np.empty((B, d, d))
    for lo in range(0, B, chunk):
        hi = min(B, lo + chunk)
        A = rng.uniform(size=(hi - lo, d, d))
        U, _, Vt = np.linalg.svd(np.einsum("bki,bkj->bij", A, A))
        mid = 1.0 + rng.uniform(size=(hi - lo, d))[:, :, None] * np.eye(d)
        C[lo:hi] = U @ mid @ Vt
    return F, f, C, c, x0


def make_navlin_batch(B, beta, seed=2):
    """cfg2: ``x0_i, goal_i ~ U(-10,10)^2``; instance 0 is the README pair."""
    rng = np.random.default_rng(seed)
    x0 = rng.uniform(-10, 10, size=(B, 2))
    goal = rng.uniform(-10, 10, size=(B, 2))
    x0[0] = (0.0, 0.0)
    goal[0] = (8.0, -9.0)
    F = np.concatenate([np.eye(2), np.eye(2)], axis=1)
    f = np.zeros(2)
    C = np.diag([2.0, 2.0, 2.0 * beta, 2.0 * beta])
    c = np.concatenate([-2.0 * goal, np.zeros((B, 2))], axis=1)
    return F, f, C, c, x0, goal


# -------------------------------------------------------------------- envs ---
NAV_CONFIG = {                       # tfmpc/envs/navigation/nav.config.json:5-11
    "goal": [[8.0], [9.0]],
    "deceleration": {"center": [[[5.0], [4.5]], [[1.5], [3.0]]], "decay": [1.15, 1.2]},
    "low": [[-1.0], [-1.0]],
    "high": [[1.0], [1.0]],
}

HVAC6_CONFIG = {                     # tfmpc/envs/hvac/hvac6.config.json:5-30
    "temp_outside": [[6.0]] * 6,
    "temp_hall": [[10.0]] * 6,
    "temp_lower_bound": [[20.0]] * 6,
    "temp_upper_bound": [[23.5]] * 6,
    "R_outside": [[4.0]] * 6,
    "R_hall": [[2.0]] * 6,
    "R_wall": [[1.5] * 6] * 6,
    "capacity": [[80.0]] * 6,
    "air_max": [[10.0]] * 6,
    "adj": [[False, True, False, True, False, False],
            [False, False, True, False, True, False],
            [False, False, False, False, False, True],
            [False, False, False, False, True, False],
            [False, False, False, False, False, True],
            [False, False, False, False, False, False]],
    "adj_outside": [[True], [False], [True], [True], [False], [True]],
    "adj_hall": [[True]] * 6,
}
HVAC6_X0 = [[10.0]] * 6

RES4_CONFIG = {                      # tfmpc/envs/reservoir/res4.config.json:5-18
    "max_res_cap": [[1000.0]] * 4,
    "low_penalty": [[-5.0]] * 4,
    "high_penalty": [[-100.0]] * 4,
    "set_point_penalty": [[-0.1]] * 4,
    "rain_shape": [[16.0]] * 4,
    "rain_scale": [[1.25]] * 4,
    "lower_bound": [[20.0], [30.0], [40.0], [60.0]],
    "upper_bound": [[80.0], [180.0], [380.0], [480.0]],
    "downstream": [[0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1], [0, 0, 0, 0]],
}
RES4_X0 = [[75.0], [50.0], [50.0], [50.0]]


def hvac_config(n, seed=5):
    """cfg5 HVAC at n rooms: the recipe of the reference's ``tests/conftest.py:35-63``
    with numpy's ``default_rng`` instead of TF's RNG."""
    rng = np.random.default_rng(seed)
    col = lambda a: np.asarray(a, dtype=np.float64).reshape(n, 1).tolist()
    lower = rng.normal(20.0, 1.5, size=n)
    R_wall = rng.normal(1.5, 0.1, size=(n, n))
    R_wall = 0.5 * (R_wall + R_wall.T)
    adj = rng.uniform(size=(n, n)) >= 0.4
    adj = np.triu(np.logical_and(adj, ~np.eye(n, dtype=bool)))
    return {
        "temp_outside": col(rng.normal(6.0, 1.0, size=n)),
        "temp_hall": col(rng.normal(10.0, 1.0, size=n)),
        "temp_lower_bound": col(lower),
        "temp_upper_bound": col(lower + rng.uniform(3.5, 55.5)),
        "R_outside": col(rng.normal(4.0, 1.0, size=n)),
        "R_hall": col(rng.normal(2.0, 0.5, size=n)),
        "R_wall": R_wall.tolist(),
        "capacity": col(rng.normal(80.0, 2.0, size=n)),
        "air_max": col(rng.normal(15.0, 1.0, size=n)),
        "adj": adj.tolist(),
        "adj_outside": (rng.normal(size=(n, 1)) >= 0.0).tolist(),
        "adj_hall": (rng.normal(size=(n, 1)) >= 0.0).tolist(),
    }


def reservoir_config(n, seed=5, max_res_cap=100.0):
    """cfg5 Reservoir at n reservoirs: recipe of ``tests/conftest.py:83-127``."""
    rng = np.random.default_rng(seed)
    col = lambda a: np.asarray(a, dtype=np.float64).reshape(n, 1).tolist()
    downstream = np.zeros((n, n))
    for i in range(n - 1):
        downstream[i, i + 1] = 1.0          # linear topology
    rain_mean = 0.20 * max_res_cap
    rain_var = 0.05 * max_res_cap ** 2
    return {
        "max_res_cap": col([max_res_cap] * n),
        "lower_bound": col(max_res_cap * rng.uniform(0.0, 0.5, size=n)),
        "upper_bound": col(max_res_cap * rng.uniform(0.5, 1.0, size=n)),
        "low_penalty": col([-5.0] * n),
        "high_penalty": col([-100.0] * n),
        "set_point_penalty": col([-0.1] * n),
        "downstream": downstream.tolist(),
        "rain_shape": col([rain_mean ** 2 / rain_var] * n),
        "rain_scale": col([rain_var / rain_mean] * n),
    }


def scalar_uniform_actions(T, low, high, rng):
    """``u_t = lo' + r_t (hi' - lo')`` with one scalar uniform per step (the
    reference's ``iLQR.start``, ``ilqr.py:59-70``).  Returns ``[T, m, 1]``."""
    low = np.asarray(low, dtype=np.float64).reshape(-1, 1)
    high = np.asarray(high, dtype=np.float64).reshape(-1, 1)
    lo = np.where(np.isinf(low), -1.0, low)
    hi = np.where(np.isinf(high), 1.0, high)
    r = rng.uniform(size=T)
    return np.stack([lo + r[t] * (hi - lo) for t in range(T)])
