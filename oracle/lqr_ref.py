"""TEST INFRASTRUCTURE ONLY -- numpy restatement of ``tfmpc/solvers/lqr.py``.

Single instance, column vectors ``[n, 1]``, the reference's operation order.
``dtype=np.float64`` is the oracle of record; ``dtype=np.float32`` stands in for
the reference's fp32 TensorFlow path when an fp32 error budget is needed
(SURVEY.md F4).  Citations are into ``/root/reference``.
"""

import numpy as np


def make_lqr(state_size, action_size):
    """Random LQR problem, RNG draw order of ``tfmpc/envs/__init__.py:9-18``.

    Uses the GLOBAL numpy RNG exactly like the reference (seed it first).
    Returns float64 ``(F, f, C, c)``; the reference casts to fp32 in
    ``lqr.py:19-22``.
    """
    from sklearn.datasets import make_spd_matrix

    n_dim = state_size + action_size
    F = np.random.normal(size=(state_size, n_dim))
    f = np.random.normal(size=(state_size, 1))
    C = make_spd_matrix(n_dim)
    c = np.random.normal(size=(n_dim, 1))
    return F, f, C, c


def make_lqr_linear_navigation(goal, beta):
    """``tfmpc/envs/__init__.py:21-30`` (x' = x + u, cost |x-g|^2 + beta |u|^2
    with the constant |g|^2 dropped).  ``F = [I I]`` for any n (quirk Q7)."""
    goal = np.asarray(goal, dtype=np.float64).reshape(-1, 1)
    n = goal.shape[0]
    F = np.concatenate([np.identity(n), np.identity(n)], axis=1)
    f = np.zeros((n, 1))
    C = np.diag([2.0] * n + [2.0 * beta] * n)
    c = np.concatenate([-2.0 * goal, np.zeros((n, 1))], axis=0)
    return F, f, C, c


def transition(F, f, x, u):
    """lqr.py:36-39"""
    z = np.concatenate([x, u], axis=0)
    return F @ z + f


def cost(C, c, x, u):
    """lqr.py:41-47 -> shape [1, 1]"""
    z = np.concatenate([x, u], axis=0)
    return 0.5 * (z.T @ C) @ z + z.T @ c


def final_cost(C, c, x):
    """lqr.py:49-57 -> shape [1, 1]"""
    n = x.shape[0]
    return 0.5 * (x.T @ C[:n, :n]) @ x + x.T @ c[:n]


def backward(F, f, C, c, T, dtype=np.float64, terminal="cost"):
    """lqr.py:59-129.  Returns ``(policy, value_fn)``: ``policy[t] = (K, k)``,
    ``value_fn[t] = (V, v, const)`` for t = 0..T-1.

    ``terminal="cost"`` is the reference at v0.7.0 (``V=C_xx, v=c_x``,
    lqr.py:67-68); ``terminal="zero"`` starts from a zero value function, which
    is what produced the table in the reference's ``README.md:75-90`` (SURVEY.md
    F3) and exists only to check that known answer.
    """
    F = np.asarray(F, dtype=dtype)
    f = np.asarray(f, dtype=dtype).reshape(-1, 1)
    C = np.asarray(C, dtype=dtype)
    c = np.asarray(c, dtype=dtype).reshape(-1, 1)
    n = F.shape[0]

    if terminal == "cost":
        V = C[:n, :n]
        v = c[:n]
    elif terminal == "zero":
        V = np.zeros((n, n), dtype=dtype)
        v = np.zeros((n, 1), dtype=dtype)
    else:
        raise ValueError(terminal)
    const = np.zeros((1, 1), dtype=dtype)

    policy, value_fn = [], [(V, v, const)]
    half = dtype(0.5)
    for _ in reversed(range(T)):
        Ft_V = F.T @ V                                     # :74
        Q = C + Ft_V @ F                                   # :75
        q = c + Ft_V @ f + F.T @ v                         # :76-78
        Q_uu, Q_ux, q_u = Q[n:, n:], Q[n:, :n], q[n:]      # :80-82
        inv_Q_uu = np.linalg.inv(Q_uu)                     # :84 (general inverse)
        K = -(inv_Q_uu @ Q_ux)                             # :86
        k = -(inv_Q_uu @ q_u)                              # :87
        Q_xx, Q_xu, q_x = Q[:n, :n], Q[:n, n:], q[:n]
        Kt_Quu = K.T @ Q_uu                                # :95
        V_new = Q_xx + Q_xu @ K + K.T @ Q_ux + Kt_Quu @ K  # :97-100
        v_new = q_x + Q_xu @ k + K.T @ q_u + Kt_Quu @ k    # :102-105
        # :107-121 -- W, w are recomputed from the PREVIOUS value function and
        # equal Q, q above; const accumulates the policy-independent terms.
        V_f = V @ f
        const1 = half * (k.T @ (Q_uu @ k))
        const2 = k.T @ q_u
        const3 = half * (f.T @ V_f) + f.T @ v
        const = const + (const1 + const2 + const3)
        V, v = V_new, v_new
        policy.append((K, k))
        value_fn.append((V, v, const))

    policy = list(reversed(policy))                        # :126
    value_fn = list(reversed(value_fn[1:]))                # :127
    return policy, value_fn


def forward(F, f, C, c, policy, x0, T, dtype=np.float64):
    """lqr.py:131-161 -> ``states[T+1,n,1], actions[T,m,1], costs[T+1,1,1]``."""
    F = np.asarray(F, dtype=dtype)
    f = np.asarray(f, dtype=dtype).reshape(-1, 1)
    C = np.asarray(C, dtype=dtype)
    c = np.asarray(c, dtype=dtype).reshape(-1, 1)
    x = np.asarray(x0, dtype=dtype).reshape(-1, 1)
    states, actions, costs = [x], [], []
    for t in range(T):
        K, k = policy[t]
        u = K @ x + k                                      # :143
        nx = transition(F, f, x, u)                        # :145
        costs.append(cost(C, c, x, u))                     # :146
        x = nx
        states.append(x)
        actions.append(u)
    costs.append(final_cost(C, c, x))                      # :154
    return np.stack(states), np.stack(actions), np.stack(costs)


def solve(F, f, C, c, x0, T, dtype=np.float64, terminal="cost"):
    """lqr.py:163-166.  Returns ``(states[T+1,n], actions[T,m], costs[T+1])``
    squeezed the way ``Trajectory`` stores them (trajectory.py:12-15), plus the
    policy and value function."""
    policy, value_fn = backward(F, f, C, c, T, dtype=dtype, terminal=terminal)
    x, u, cs = forward(F, f, C, c, policy, x0, T, dtype=dtype)
    return x[..., 0], u[..., 0], cs.reshape(-1), policy, value_fn
