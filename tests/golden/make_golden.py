"""Generates the golden fixtures in this directory.

The reference (thiagopbueno/tf-mpc v0.7.0) cannot be imported in the build
container (TensorFlow / gym / tuneconfig are absent, no network), so these
vectors come from two sources, both recorded in each file's ``source`` field:

* ``reference-known-answer``: numbers the reference itself publishes or asserts
  (README table, box-QP solutions in its tests) typed in as DATA;
* ``oracle-fp64``: outputs of ``oracle/`` (the fp64 CPU restatement of the
  reference equations) on seeded inputs.  PARITY UNPINNED for the iLQR ones.

Run from the repo root:  ``python tests/golden/make_golden.py``
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import problems  # noqa: E402
from oracle import boxqp_ref, envs_ref, ilqr_ref, lqr_ref  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path)} B")


def pack_lqr(pol, val):
    K = np.stack([p[0] for p in pol])
    k = np.stack([p[1][:, 0] for p in pol])
    V = np.stack([v[0] for v in val])
    v = np.stack([w[1][:, 0] for w in val])
    const = np.array([w[2].reshape(()) for w in val])
    return dict(K=K, k=k, V=V, v=v, const=const)


# ---------------------------------------------------------------- README ---
def readme_navlin():
    """README.md:75-90 of the reference -- `tfmpc navlin -b 5.0 -hr 10 -- "0.0 0.0" "8.0 -9.0"`.
    Rows are (state after the step, action, cost), i.e. Trajectory.__getitem__."""
    states = [[2.8645, -3.2225], [4.7018, -5.2895], [5.8795, -6.6145], [6.6331, -7.4623],
              [7.1134, -8.0025], [7.4163, -8.3433], [7.6025, -8.5528], [7.7091, -8.6727],
              [7.7576, -8.7273], [7.7576, -8.7273]]
    actions = [[2.8645, -3.2225], [1.8373, -2.0670], [1.1777, -1.3249], [0.7536, -0.8478],
               [0.4802, -0.5403], [0.3029, -0.3408], [0.1862, -0.2094], [0.1067, -0.1200],
               [0.0485, -0.0545], [0.0000, 0.0000]]
    costs = [92.9486, -47.0048, -104.6422, -128.3791, -138.1544, -142.1795, -143.8354,
             -144.5131, -144.7817, -144.8669]
    save("readme_navlin", source="reference-known-answer README.md:75-90",
         x0=np.zeros(2), goal=np.array([8.0, -9.0]), beta=5.0, T=10,
         next_states=np.array(states), actions=np.array(actions), costs=np.array(costs),
         total=-1045.4086, final_state=np.array([7.757592, -8.727291]))


# ---------------------------------------------------------------- box-QP ---
def boxqp_kats():
    """tests/test_utils_optimization.py:7-15 of the reference: H = 2I, q = -2 goal."""
    cases = [
        ([0.0, 0.0], [-1.0, 0.5], [1.0, 1.0], [0.0, 0.5]),
        ([0.0, 0.0], [0.5, -1.0], [1.0, 1.0], [0.5, 0.0]),
        ([1.0, 1.0], [0.0, 1.5], [2.0, 2.0], [1.0, 1.5]),
        ([1.0, 1.0], [1.5, 0.0], [2.0, 2.0], [1.5, 1.0]),
        ([0.0, 0.0, 0.0], [-1.0, 0.5, -1.0], [1.0, 1.0, 1.0], [0.0, 0.5, 0.0]),
        ([0.0, 0.0, 0.0], [-1.0, 0.5, 0.30], [1.0, 1.0, 1.0], [0.0, 0.5, 0.30]),
    ]
    arrays = {"source": "reference-known-answer tests/test_utils_optimization.py:7-15", "n_cases": len(cases)}
    for i, (goal, low, high, x_star) in enumerate(cases):
        arrays[f"goal{i}"] = np.array(goal)
        arrays[f"low{i}"] = np.array(low)
        arrays[f"high{i}"] = np.array(high)
        arrays[f"x_star{i}"] = np.array(x_star)
    save("boxqp_kats", **arrays)

    # seeded dense box-QPs with the oracle's answer (exercise clamping/refactorisation)
    rng = np.random.default_rng(11)
    arrays = {"source": "oracle-fp64 boxqp_ref.projected_newton_qp", "n_cases": 24}
    for i in range(24):
        m = int(rng.integers(2, 9))
        A = rng.normal(size=(m, m))
        H = A @ A.T + 0.5 * np.eye(m)
        q = 3.0 * rng.normal(size=m)
        low = -rng.uniform(0.1, 1.0, size=m)
        high = rng.uniform(0.1, 1.0, size=m)
        x0 = (low + high) / 2
        x, Hfree, free, clamped = boxqp_ref.projected_newton_qp(H, q, low, high, x0)
        arrays.update({f"H{i}": H, f"q{i}": q, f"low{i}": low, f"high{i}": high, f"x0{i}": x0,
                       f"x{i}": x[:, 0], f"free{i}": free[:, 0]})
    save("boxqp_dense", **arrays)


# ------------------------------------------------------------------- LQR ---
def lqr_cases():
    for name, n, m, T, seeds in (("lqr_cfg1", 3, 2, 10, (0, 1, 2)), ("lqr_cfg3", 16, 8, 50, (1000, 1001, 1002))):
        arrays = {"source": "oracle-fp64 lqr_ref.solve on make_lqr(seed)", "seeds": np.array(seeds), "T": T}
        for i, seed in enumerate(seeds):
            F, f, C, c = problems.make_lqr_instance(seed, n, m)
            x0 = np.array([-1.0, 0.5, 3.6]) if n == 3 else np.random.default_rng(seed).normal(size=n)
            x, u, cs, pol, val = lqr_ref.solve(F, f, C, c, x0, T)
            arrays.update({f"F{i}": F, f"f{i}": f, f"C{i}": C, f"c{i}": c, f"x0{i}": x0,
                           f"states{i}": x, f"actions{i}": u, f"costs{i}": cs})
            for key, val_ in pack_lqr(pol, val).items():
                arrays[f"{key}{i}"] = val_
        save(name, **arrays)

    # navlin: README pair + 8 random pairs, beta = 5
    F, f, C, c, x0, goal = problems.make_navlin_batch(9, 5.0)
    outs = [lqr_ref.solve(F, f, C, c[i], x0[i], 50) for i in range(9)]
    save("lqr_navlin", source="oracle-fp64 lqr_ref.solve on make_navlin_batch(9, beta=5)",
         T=50, beta=5.0, x0=x0, goal=goal,
         states=np.stack([o[0] for o in outs]), actions=np.stack([o[1] for o in outs]),
         costs=np.stack([o[2] for o in outs]))


# ------------------------------------------------------------------ iLQR ---
def ilqr_record(env, x0, T, u_init, **kwargs):
    s = ilqr_ref.ILQRRef(env, **kwargs)
    xs, us, cs = s.start(x0, T, u_init=u_init)
    tm, cm, fm = s.derivatives(xs, us)
    rec = dict(x0=np.asarray(x0, dtype=float).reshape(-1), u_init=np.asarray(u_init)[..., 0], T=T,
               start_states=xs[..., 0], start_costs=cs,
               f=tm.f[..., 0], f_x=tm.f_x, f_u=tm.f_u,
               l=cm.l, l_x=cm.l_x[..., 0], l_u=cm.l_u[..., 0], l_xx=cm.l_xx, l_uu=cm.l_uu,
               l_ux=cm.l_ux, l_xu=cm.l_xu, fl=fm.l, fl_x=fm.l_x[:, 0], fl_xx=fm.l_xx)
    for mu in (0.0, 1.0):
        K, k, J, dV1, dV2 = s.backward(T, us, tm, cm, fm, mu=mu)
        tag = "mu0" if mu == 0.0 else "mu1"
        rec.update({f"K_{tag}": K, f"k_{tag}": k[..., 0], f"J_{tag}": J, f"dV1_{tag}": dV1, f"dV2_{tag}": dV2})
        if mu == 0.0:
            for a, alpha in enumerate((1.0, 0.25)):
                fx, fu, fc, fJ, fr = s.forward(xs, us, K, k, alpha)
                rec.update({f"fwd{a}_alpha": alpha, f"fwd{a}_states": fx[..., 0], f"fwd{a}_actions": fu[..., 0],
                            f"fwd{a}_costs": fc, f"fwd{a}_J": fJ, f"fwd{a}_residual": fr})
    x, u, c, it = s.solve(x0, T, u_init=u_init)
    rec.update(sol_states=x, sol_actions=u, sol_costs=c, sol_iteration=it,
               sol_attempts=len(s.trace),
               sol_alphas=np.array([np.nan if r["alpha"] is None else r["alpha"] for r in s.trace]),
               sol_mus=np.array([r["mu"] for r in s.trace]))
    return rec


def ilqr_cases():
    T = 10
    arrays = {"source": "oracle-fp64 ilqr_ref.ILQRRef (PARITY UNPINNED)", "n_cases": 4}
    i = 0
    for beta in (0.0, 5.0):                       # tests/test_ilqr.py:11-23 of the reference
        for bounds in (None, (-1.0, 1.0)):
            low, high = bounds if bounds else (None, None)
            env = envs_ref.NavigationLQR([[5.5], [-9.0]], beta, low, high)
            u0 = problems.scalar_uniform_actions(T, env.action_space.low, env.action_space.high,
                                                 np.random.default_rng(100 + i))
            rec = ilqr_record(env, [[0.0], [0.0]], T, u0)
            arrays.update({f"beta{i}": beta, f"bounded{i}": bounds is not None})
            arrays.update({f"{k}{i}": v for k, v in rec.items()})
            i += 1
    save("ilqr_navlqr", **arrays)

    cfg = problems.NAV_CONFIG
    env = envs_ref.Navigation(cfg["goal"], cfg["deceleration"]["center"], cfg["deceleration"]["decay"],
                              cfg["low"], cfg["high"])
    arrays = {"source": "oracle-fp64 ilqr_ref.ILQRRef on nav.config.json (PARITY UNPINNED)", "n_cases": 3}
    x0s = ([[0.0], [0.0]], [[2.0], [7.5]], [[9.0], [1.0]])
    for i, x0 in enumerate(x0s):
        u0 = problems.scalar_uniform_actions(20, env.action_space.low, env.action_space.high,
                                             np.random.default_rng(200 + i))
        arrays.update({f"{k}{i}": v for k, v in ilqr_record(env, x0, 20, u0).items()})
    save("ilqr_navigation", **arrays)

    env = envs_ref.HVAC(**problems.HVAC6_CONFIG)
    u0 = problems.scalar_uniform_actions(12, env.action_space.low, env.action_space.high, np.random.default_rng(300))
    save("ilqr_hvac6", source="oracle-fp64 ilqr_ref.ILQRRef on hvac6.config.json (PARITY UNPINNED)",
         **ilqr_record(env, problems.HVAC6_X0, 12, u0))

    env = envs_ref.Reservoir(**problems.RES4_CONFIG)
    u0 = problems.scalar_uniform_actions(12, env.action_space.low, env.action_space.high, np.random.default_rng(400))
    save("ilqr_res4", source="oracle-fp64 ilqr_ref.ILQRRef on res4.config.json (PARITY UNPINNED)",
         **ilqr_record(env, problems.RES4_X0, 12, u0))

    # headline shape through the iLQR path: LQ env n=16, m=8
    # F scaled to spectral radius ~1: with the raw N(0,1) F (rho ~ 4) the OPEN-LOOP start
    # rollout grows like 4^T and delta_x = x - x_hat cancels catastrophically in fp32
    F, f, C, c = problems.make_lqr_instance(1000, 16, 8)
    F = 0.25 * F
    env = envs_ref.LQEnv(F, f, C, c)
    x0 = np.random.default_rng(1000).normal(size=(16, 1))
    u0 = np.random.default_rng(500).normal(size=(12, 8, 1)) * 0.1
    rec = ilqr_record(env, x0, 12, u0)
    save("ilqr_lq16x8", source="oracle-fp64 ilqr_ref.ILQRRef on LQEnv(make_lqr seed 1000, F*0.25) (PARITY UNPINNED)",
         lq_F=F, lq_f=f, lq_C=C, lq_c=c, **rec)
    ilqr_lq_bounded_case()


def ilqr_lq_bounded_case():
    """The headline shape with CONTROL LIMITS (ilqr.py:136-138,364-387 + optimization.py:6-101): the same LQ problem,
    actions boxed to [-1.5, 1.5]."""
    F, f, C, c = problems.make_lqr_instance(1000, 16, 8)
    F = 0.25 * F
    env = envs_ref.LQEnv(F, f, C, c, low=-1.5, high=1.5)
    x0 = np.random.default_rng(1000).normal(size=(16, 1))
    u0 = np.random.default_rng(500).normal(size=(12, 8, 1)) * 0.1
    rec = ilqr_record(env, x0, 12, u0)
    save("ilqr_lq16x8_bounded", source="oracle-fp64 ilqr_ref.ILQRRef on LQEnv(make_lqr seed 1000, F*0.25), actions in "
         "[-1.5, 1.5] (PARITY UNPINNED)", lq_F=F, lq_f=f, lq_C=C, lq_c=c, low=-1.5, high=1.5, **rec)


if __name__ == "__main__":
    readme_navlin()
    boxqp_kats()
    lqr_cases()
    ilqr_cases()
