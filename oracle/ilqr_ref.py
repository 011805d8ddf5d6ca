"""TEST INFRASTRUCTURE ONLY -- numpy restatement of ``tfmpc/solvers/ilqr.py``
(control-limited iLQR, Tassa et al. 2014), single instance, reference operation
order and reference quirks kept:

* Q1  ``start`` draws ONE scalar uniform per timestep (``ilqr.py:70``);
* Q2  the regularisation bump made after a Cholesky failure inside ``_backward``
      is local and not returned to ``solve`` (``ilqr.py:285-315``);
* Q3  ``residual < atol`` accepts a step the line search rejected (``:253-257``);
* Q4  11 step sizes ``geomspace(1, alpha_min, 11)`` (``:322``);
* Q5  ``backward`` defaults to ``mu=1.0`` while ``solve`` starts at 0 (``:95,215``).

PARITY UNPINNED: the reference's tests hold no numeric iLQR answer (see
``oracle/__init__.py``); with ``dtype=np.float64`` this file is the oracle of
record for K, k, line-search decisions, iteration counts and trajectories.
"""

import numpy as np

from . import boxqp_ref


class CholeskyFailure(Exception):
    """Stands for ``tf.errors.InvalidArgumentError`` from ``tf.linalg.cholesky``."""


def _cholesky(A):
    try:
        return np.linalg.cholesky(A)
    except np.linalg.LinAlgError as e:
        raise CholeskyFailure(str(e))


def _cholesky_solve(L, B):
    return np.linalg.solve(L.T, np.linalg.solve(L, B))


class ILQRRef:

    def __init__(self, env, dtype=np.float64, **kwargs):
        self.env = env
        self.dtype = np.dtype(dtype).type
        self.atol = kwargs.get("atol", 5e-3)                       # ilqr.py:28-37
        self.max_iterations = kwargs.get("max_iterations", 100)
        self.mu_min = kwargs.get("mu_min", 1e-6)
        self.delta_0 = kwargs.get("delta_0", 2.0)
        self.c1 = kwargs.get("c1", 0.0)
        self.alpha_min = kwargs.get("alpha_min", 1e-3)
        self.trace = []            # per backward attempt: dict(iteration, mu, alpha, accepted, ...)
        self.on_bang_bang = None   # optional callback(t, Q_u, |l_u| + |f_u|^T |V_x|) where the bang-bang controller of ilqr.py:140-141 is taken

    @property
    def low(self):
        return np.asarray(self.env.action_space.low, dtype=self.dtype)

    @property
    def high(self):
        return np.asarray(self.env.action_space.high, dtype=self.dtype)

    # ------------------------------------------------------------------ :53-82
    def start(self, x0, T, r=None, u_init=None, rng=None):
        """Random initial rollout.  ``r[T]`` are the per-step scalar uniforms
        (Q1); alternatively inject ``u_init[T,m,1]``."""
        dt = self.dtype
        low, high = self.low, self.high
        minval = np.where(np.isinf(low), -np.ones_like(low), low)
        maxval = np.where(np.isinf(high), np.ones_like(high), high)
        if u_init is None:
            if r is None:
                rng = rng or np.random.default_rng()
                r = rng.uniform(size=T)
            r = np.asarray(r, dtype=dt)
            u_init = np.stack([minval + r[t] * (maxval - minval) for t in range(T)])
        u_init = np.asarray(u_init, dtype=dt).reshape(T, -1, 1)
        x = np.asarray(x0, dtype=dt).reshape(-1, 1)
        states, costs = [x], []
        for t in range(T):
            costs.append(dt(self.env.cost(x, u_init[t])))
            x = np.asarray(self.env.transition(x, u_init[t]), dtype=dt)
            states.append(x)
        costs.append(dt(self.env.final_cost(x)))
        return np.stack(states), u_init, np.asarray(costs, dtype=dt)

    # ------------------------------------------------------------------ :84-92
    def derivatives(self, states, actions):
        tm = self.env.get_linear_transition(states[:-1], actions)
        cm = self.env.get_quadratic_cost(states[:-1], actions)
        fm = self.env.get_quadratic_final_cost(states[-1])
        cast = lambda tup: type(tup)(*[np.asarray(a, dtype=self.dtype) for a in tup])
        return cast(tm), cast(cm), cast(fm)

    # ----------------------------------------------------------------- :94-172
    def backward(self, T, actions, transition_model, cost_model, final_cost_model, mu=1.0):
        dt = self.dtype
        n, m = self.env.state_size, self.env.action_size
        mu = dt(mu)
        K = np.zeros((T, m, n), dtype=dt)
        k = np.zeros((T, m, 1), dtype=dt)
        V_x = final_cost_model.l_x                                    # :101-106
        V_xx = final_cost_model.l_xx
        J = dt(final_cost_model.l)
        dV1 = dt(0.0)
        dV2 = dt(0.0)
        bounded = self.env.action_space.is_bounded()
        low, high = self.low, self.high
        eye = np.eye(n, dtype=dt)
        half = dt(0.5)

        for t in range(T - 1, -1, -1):                                # :108
            f_x, f_u = transition_model.f_x[t], transition_model.f_u[t]
            l = cost_model.l[t]
            l_x, l_u = cost_model.l_x[t], cost_model.l_u[t]
            l_xx, l_uu, l_xu = cost_model.l_xx[t], cost_model.l_uu[t], cost_model.l_xu[t]

            Q_x = l_x + f_x.T @ V_x                                   # :122-123
            Q_u = l_u + f_u.T @ V_x
            fxT_Vxx = f_x.T @ V_xx                                    # :125-127
            fuT_Vxx = f_u.T @ V_xx
            fuT_Vxx_reg = f_u.T @ (V_xx + mu * eye)
            Q_xx = l_xx + fxT_Vxx @ f_x                               # :129-131
            Q_uu = l_uu + fuT_Vxx @ f_u
            Q_ux = l_xu.T + fuT_Vxx @ f_x
            Q_uu_reg = l_uu + fuT_Vxx_reg @ f_u                       # :133-134
            Q_ux_reg = l_xu.T + fuT_Vxx_reg @ f_x

            if bounded:                                               # :136-143
                if np.count_nonzero(V_xx) > 0:
                    self._step = t                                    # (test hooks below: which step the controller is for)
                    K_t, k_t = self._get_constrained_controller(actions[t], Q_uu_reg, Q_ux_reg, Q_u)
                else:
                    K_t = np.zeros((m, n), dtype=dt)
                    k_t = np.where(Q_u >= 0.0, low - actions[t], high - actions[t])
                    if self.on_bang_bang is not None:          # test hook: the selector's operand and the size of the terms it is
                        self.on_bang_bang(t, Q_u, np.abs(l_u) + np.abs(f_u).T @ np.abs(V_x))     # the sum of (tests/trace_oracle.py)
            else:
                K_t, k_t = self._get_unconstrained_controller(Q_uu_reg, Q_ux_reg, Q_u)

            KtT_Quu = K_t.T @ Q_uu                                    # :147
            V_x = Q_x + Q_ux.T @ k_t + K_t.T @ Q_u + KtT_Quu @ k_t    # :149-154
            V_xx = Q_xx + Q_ux.T @ K_t + K_t.T @ Q_ux + KtT_Quu @ K_t  # :156-161
            V_xx = half * (V_xx + V_xx.T)                             # :162
            J = J + l                                                 # :164
            dV1 = dV1 + (k_t.T @ Q_u).reshape(())                     # :166
            dV2 = dV2 + half * ((k_t.T @ Q_uu) @ k_t).reshape(())     # :167
            K[t], k[t] = K_t, k_t

        return K, k, dt(J), dt(dV1), dt(dV2)

    # ---------------------------------------------------------------- :174-212
    def forward(self, x, u, K, k, alpha=1.0):
        dt = self.dtype
        T = x.shape[0] - 1
        alpha = dt(alpha)
        low, high = self.low, self.high
        state = x[0]
        states, actions, costs = [state], [], []
        J = dt(0.0)
        residual = dt(0.0)
        for t in range(T):
            delta_x = state - x[t]                                    # :193
            delta_u = alpha * k[t] + K[t] @ delta_x                   # :194
            action = np.clip(u[t] + delta_u, low, high)               # :196-197
            cost = dt(self.env.cost(state, action))
            state = np.asarray(self.env.transition(state, action), dtype=dt)
            actions.append(action)
            costs.append(cost)
            states.append(state)
            J = J + cost                                              # :205
            residual = max(residual, dt(np.max(np.abs(delta_u))))     # :206
        fc = dt(self.env.final_cost(state))
        costs.append(fc)
        J = J + fc
        return np.stack(states), np.stack(actions), np.asarray(costs, dtype=dt), dt(J), dt(residual)

    # ---------------------------------------------------------------- :214-283
    def solve(self, x0, T, r=None, u_init=None, rng=None):
        """Returns ``(states[T+1,n], actions[T,m], costs[T+1], iteration)`` --
        the trajectory squeezed as ``Trajectory`` stores it."""
        dt = self.dtype
        mu, delta = 0.0, 1.0                                          # :215-216
        self.trace = []
        x_hat, u_hat, c_hat = self.start(x0, T, r=r, u_init=u_init, rng=rng)

        iteration = 0
        for iteration in range(self.max_iterations):                  # :227
            models = self.derivatives(x_hat, u_hat)                   # :234
            converged = False
            while True:                                               # :238
                K, k, J_hat, dV1, dV2 = self._backward(T, u_hat, *models, mu, delta)
                g_norm = np.mean(np.max(np.abs(k) / (np.abs(u_hat) + dt(1.0)), axis=1), axis=0)[0]  # :243
                rec = dict(iteration=iteration, mu=mu, delta=delta, J_hat=float(J_hat),
                           g_norm=float(g_norm), alpha=None, accepted=None, residual=None)
                self.trace.append(rec)
                if g_norm < self.atol:                                # :245-248
                    rec["converged"] = "g_norm"
                    converged = True
                    break
                done, x, u, c, residual, alpha = self._forward(x_hat, u_hat, J_hat, K, k, dV1, dV2)
                rec.update(alpha=float(alpha), accepted=bool(done), residual=float(residual))
                if residual < self.atol:                              # :253-257 (Q3)
                    rec["converged"] = "residual"
                    converged = True
                    x_hat, u_hat, c_hat = x, u, c
                    break
                if done:                                              # :259-266
                    delta = min(1 / self.delta_0, delta / self.delta_0)
                    mu = mu * delta * (mu * delta > self.mu_min)
                    x_hat, u_hat, c_hat = x, u, c
                    break
                else:                                                 # :267-270
                    delta = max(self.delta_0, delta * self.delta_0)
                    mu = max(self.mu_min, mu * delta)
                    if not np.isfinite(mu) or mu > 1e30:
                        raise RuntimeError("regularisation diverged (reference would loop forever)")
            if converged:                                             # :276-277
                break

        return x_hat[..., 0], u_hat[..., 0], c_hat, iteration

    # ---------------------------------------------------------------- :285-315
    def _backward(self, T, u_hat, transition_model, cost_model, final_cost_model, mu, delta):
        while True:
            try:
                return self.backward(T, u_hat, transition_model, cost_model, final_cost_model,
                                     self.dtype(mu))
            except CholeskyFailure:
                delta = max(self.delta_0, delta * self.delta_0)       # :308-309 (local: Q2)
                mu = max(self.mu_min, mu * delta)
                if mu > 1e30:
                    raise

    # ---------------------------------------------------------------- :317-355
    def _forward(self, x_hat, u_hat, J_hat, K, k, dV1, dV2):
        dt = self.dtype
        accept = False
        for alpha in np.geomspace(1.0, self.alpha_min, 11):           # :322
            alpha_ = dt(alpha)
            x, u, c, J, residual = self.forward(x_hat, u_hat, K, k, alpha_)
            # :339 -- python float alpha times fp32 tensors -> fp32 arithmetic
            delta_J = -alpha_ * (dV1 + alpha_ * dV2)
            dcost = J_hat - J
            if delta_J > 0:                                           # :342-346
                z = dcost / delta_J
            else:
                z = np.sign(dcost)
            if z >= self.c1:                                          # :351-353
                accept = True
                break
        return accept, x, u, c, residual, alpha_

    # ---------------------------------------------------------------- :357-362
    def _get_unconstrained_controller(self, Q_uu, Q_ux, Q_u):
        R = _cholesky(Q_uu)
        kK = -_cholesky_solve(R, np.concatenate([Q_u, Q_ux], axis=1))
        return kK[:, 1:], kK[:, :1]

    # ---------------------------------------------------------------- :364-387
    def _get_constrained_controller(self, u, Q_uu, Q_ux, Q_u):
        dt = self.dtype
        low = self.low - u
        high = self.high - u
        k_0 = (low + high) / 2                                        # :369
        try:
            k, Hfree, free, clamped = boxqp_ref.projected_newton_qp(Q_uu, Q_u, low, high, k_0, dtype=dt)
        except boxqp_ref.BoxQPFactorizationError:
            # Reference: UnboundLocalError inside py_function -> exit(-1).  The
            # restatement (and the product) treat it like the unconstrained
            # Cholesky failure: raise regularisation and retry.
            raise CholeskyFailure("box-QP: H_ff not positive definite")
        m, n = self.env.action_size, self.env.state_size
        K = np.zeros((m, n), dtype=dt)
        fr = free[:, 0]
        t = getattr(self, "_step", None)
        if getattr(self, "own_free", None) is not None and t is not None:      # test hook: the free set this restatement's QP ended on
            self.own_free[t] = fr.copy()
        forced = getattr(self, "forced_free", None)
        if forced is not None and t is not None:
            # TEST HOOK (tests/teacher_forced.py): K_t on ANOTHER program's free set -- the feedback gains of ilqr.py:375-385 are a
            # discrete function of which actions the QP left free, so two fp32 programs are compared on the same set; k_t stays this
            # restatement's own QP solution (the minimiser is unique: the programs agree on it to the QP's tolerance)
            fr = np.asarray(forced[t], dtype=bool)
            if np.count_nonzero(fr) > 0:
                try:
                    Hfree = np.linalg.cholesky(np.asarray(Q_uu[np.ix_(fr, fr)], dtype=dt)).astype(dt)
                except np.linalg.LinAlgError:
                    raise CholeskyFailure("box-QP (forced free set): H_ff not positive definite")
        if np.count_nonzero(fr) > 0:                                  # :377-383
            K[fr] = -_cholesky_solve(Hfree, Q_ux[fr])
        return K, k
