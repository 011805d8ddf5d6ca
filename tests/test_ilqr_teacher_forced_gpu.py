"""TEACHER-FORCED per-pass parity at workload scale (tests/teacher_forced.py) for the two kernels whose free-running traces are
mostly near-ties:

* ``ilqr_lq_box_mfma_kernel`` on ``bench.py``'s ``control_limited`` workload (tests/workloads.py): >= 128 instances -- 80 in order,
  16 with Cholesky retries, 16 of the 100-iteration family, 16 at the attempt cap.  In round 4 the gated free-running comparison
  covered 17 % of the passes and none of the heavy groups;
* ``ilqr_adjoint_mfma_kernel`` on BASELINE configs[4]'s Reservoir (n = m = 32, T = 100, 12 iterations): 64 instances = four waves
  of sixteen.  Round 4's Reservoir statement was a 16-instance test of iterations 1 and 2.

Every pass the device made (a sample of them on the heavy instances, whose restatement costs seconds per pass) is handed to the fp32
and fp64 restatements together with the device's OWN nominal trajectory, mu, delta: decisions must agree wherever the restatement's
margin is clear, the numbers -- under the device's decisions -- must lie within 5 x the fp32 restatement's own error against fp64.
PARITY UNPINNED for numeric iLQR outputs: the reference holds no numeric iLQR answer -- "vs own restatement"."""

import collections

import numpy as np
import pytest
import torch

import problems
import teacher_forced as tf
import workloads
from oracle import envs_ref, ilqr_ref
from tfmpc import _hip
from tfmpc.solvers.ilqr import iLQR, trace_records


def test_one_pass_chained_is_the_oracle_solve():
    """CPU: `one_pass`, fed its own output, IS ILQRRef.solve (decisions, trajectory, iteration count) on a control-limited LQ problem."""
    F, f, C, c, x0 = problems.make_lqr_batch_fast(3, 6, 3, seed=9)
    T = 12
    for b in range(3):
        env = envs_ref.LQEnv(0.4 * F[b], f[b], C[b], c[b], low=-0.3, high=0.3)
        o = ilqr_ref.ILQRRef(env)
        xs, us, cs, its = o.solve(x0[b][:, None], T, u_init=np.zeros((T, 3, 1)))
        x_hat, u_hat, _ = o.start(x0[b][:, None], T, u_init=np.zeros((T, 3, 1)))
        mu, delta = 0.0, 1.0
        for iteration in range(100):
            r = tf.one_pass(o, x_hat, u_hat, mu, delta, 0, None)["free"]
            if r["converged_g"]:
                break
            g = tf.one_pass(o, x_hat, u_hat, mu, delta, r["level"], r["alpha_index"])["forced"]
            if r["small_step"] or r["accepted"]:
                x_hat, u_hat = g["x"][..., None], g["u"][..., None]
            if r["small_step"]:
                break
            assert r["accepted"]                      # (no rejected pass on these problems: the chain below would need the mu schedule)
            delta = min(1 / o.delta_0, delta / o.delta_0)
            mu = mu * delta * (mu * delta > o.mu_min)
        assert iteration == its and np.array_equal(x_hat[..., 0], xs) and np.array_equal(u_hat[..., 0], us)


def _np(t):
    return t.detach().cpu().numpy()


def _nominals(make_solver, x0, u0, T, k_max, rows, full):
    """N[k] = what the device holds when iteration k starts (k = 0: the start rollout), from launches with max_iterations = k;
    asserts that run k's trace is the prefix of the full run's (the kernels are deterministic and run k extends run k - 1)."""
    start = make_solver(max_iterations=1, atol=1e9).solve_device(x0, T, u_init=u0)        # g_norm < atol at once: the start rollout comes back
    torch.cuda.synchronize()
    assert int(start["iterations"].abs().sum()) == 0
    N = [(_np(start["states"])[..., 0].astype(np.float64), _np(start["actions"])[..., 0].astype(np.float64), _np(start["costs"]).astype(np.float64))]
    full_trace, full_len = _np(full["trace"]), _np(full["trace_len"])
    for k in range(1, k_max + 1):
        out = make_solver(max_iterations=k).solve_device(x0, T, u_init=u0, trace_rows=rows)
        torch.cuda.synchronize()
        tr, ln = _np(out["trace"]), _np(out["trace_len"])
        for b in range(len(ln)):
            assert ln[b] <= full_len[b] and np.array_equal(tr[b, :ln[b]], full_trace[b, :ln[b]]), (k, b)
        N.append((_np(out["states"])[..., 0].astype(np.float64), _np(out["actions"])[..., 0].astype(np.float64), _np(out["costs"]).astype(np.float64)))
    return N


def _run(kind, cfgs, dev, N, final, caps, low=None, high=None, traj_floor=2e-6, label="", second_opinion=True, masks=None, m=None):
    """dev[i]: trace rows of instance i; N[k][j][i]: nominals; caps[i]: passes sampled of instance i.  -> statistics (Counter), details.
    masks[i][p][t] (uint8, control-limited): the device's clamp mask of pass p, step t -- the restatement's numbers are then computed on the
    device's free sets and nothing is excused (tests/teacher_forced.py)."""
    jobs, where = [], []
    for i, rows in enumerate(dev):
        for p in tf.sample_passes(len(rows), caps[i]):
            d = rows[p]
            it = d["iteration"]
            x_hat, u_hat = N[it][0][i][..., None], N[it][1][i][..., None]
            took = d["accepted"] is not None and (d["accepted"] or d["residual"] < 5e-3)
            nxt = None
            if took:
                src = N[it + 1] if it + 1 < len(N) else final
                nxt = (src[0][i], src[1][i], src[2][i])
            dev_k = None
            if kind in ("reservoir", "hvac") and nxt is not None:           # the device's bang-bang step, read off the actions it moved
                du = nxt[1] - N[it][1][i]
                dev_k = (np.where(du > 0, high - N[it][1][i], low - N[it][1][i]) * (du != 0))[..., None]
            dev_free = None
            if masks is not None:
                mk = np.asarray(masks[i][p], dtype=np.uint8)
                assert not np.any(mk == 0xFF) or m == 8, "a step of this pass has no recorded mask"
                dev_free = ((mk[:, None] >> np.arange(m)[None, :]) & 1) == 0                    # [T, m]: bit a set = clamped
            jobs.append((kind, cfgs[i] if isinstance(cfgs, list) else cfgs, x_hat, u_hat, d["mu"], d["delta"], d["level"], d["alpha_index"], dev_k,
                         ("float32", "float64"), dev_free))
            where.append((i, p, d, nxt))
    res = tf.run_passes(jobs)
    stats, details, again = collections.Counter(), [], []
    verdicts = []
    for j, ((i, p, d, nxt), r) in enumerate(zip(where, res)):
        v = tf.compare_pass(d, nxt, r["float32"], r["float64"], traj_floor=traj_floor, forced_sets=masks is not None)
        verdicts.append(v)
        if second_opinion and (v[0] == "mismatch" or v[1] in ("loose", "mismatch", "excused")):
            again.append(j)
    if again:                # second opinion: the fp32 restatement on inputs moved by one ulp measures the rounding noise of THAT pass
        res_p = tf.run_passes([jobs[j][:9] + (("float32p",),) + jobs[j][10:] for j in again])
        for j, rp in zip(again, res_p):
            i, p, d, nxt = where[j]
            verdicts[j] = tf.compare_pass(d, nxt, res[j]["float32"], res[j]["float64"], rp["float32p"], traj_floor=traj_floor, forced_sets=masks is not None)
    for (i, p, d, nxt), r, v in zip(where, res, verdicts):
        stats["passes"] += 1
        stats["decision " + v[0]] += 1
        stats["numbers " + v[1]] += 1
        if r["float32"]["free"] is not None and r["float32"]["free"]["later_failures"] > 0:
            stats["passes with a later box-QP factorisation failure in the fp32 restatement"] += 1
        if r["float32"]["free"] is not None and r["float32"]["free"]["internal_margin"] < 1.0:
            stats["passes with a near-tie inside the backward pass (fp32 restatement)"] += 1
        if r["float32"]["free"] is not None and r["float32"]["free"]["level"] > 0:
            stats["passes with Cholesky retries (fp32 restatement)"] += 1
        if masks is not None and r["float32"]["free"] is not None and "own_free" in r["float32"]["free"]:
            own = r["float32"]["free"]["own_free"]
            mine = ((np.asarray(masks[i][p], dtype=np.uint8)[:, None] >> np.arange(m)[None, :]) & 1) == 0
            if r["float32"]["free"]["level"] == d["level"] and not np.array_equal(own, mine):
                stats["passes whose box-QP free sets differ between device and fp32 restatement (same level)"] += 1
                steps = np.flatnonzero((own != mine).any(axis=1))
                if v[1] != "ok" or len(details) < 4:
                    details.append((i, p, ("free sets", f"{len(steps)} steps differ, first t={int(steps[-1])}: device free "
                                           f"{np.flatnonzero(mine[steps[-1]]).tolist()}, restatement {np.flatnonzero(own[steps[-1]]).tolist()}", "")))
        if v[2]:
            details.append((i, p, v))
    print(f"\n{label}: {dict(stats)}")
    for det in details[:12]:
        print("   ", det)
    return stats, details, where, res, verdicts


def _control_limited(n_order, n_group, cap_light, cap_heavy, every_pass_of=0, log=None, dump=None, replay=None, workers=None, mismatch_share=0.0):
    """The control-limited workload, teacher-forced: `n_order` instances in order + `n_group` of each heavy kind; `cap_*` passes sampled per
    instance (tests/teacher_forced.py:sample_passes), EVERY pass of the first `every_pass_of` instances of the two heavy groups.
    `dump` (GPU side) writes everything the restatement needs -- the device's traces, clamp masks, nominal trajectories, the instances'
    problem data -- to a file and stops; `replay` (no GPU) reads such a file and does the comparison (tools/teacher_forced_heavy.py: the
    every-pass run of the heavy instances is minutes of restatement on CPU cores the GPU box does not have to hold a GPU for)."""
    if replay is not None:
        import pickle
        with open(replay, "rb") as fh:
            saved = pickle.load(fh)
        return _control_limited_compare(log=log, workers=workers, mismatch_share=mismatch_share, **saved)
    w = workloads.control_limited(65536)
    rows = 170
    solver = workloads.solver_of(w)
    full = solver.solve_device(w["x0"], w["T"], u_init=w["u0"], trace_rows=rows)
    torch.cuda.synchronize()
    st, it = _np(full["status"]), _np(full["iterations"])
    retried = np.flatnonzero((st & _hip.ST_NOT_PD) != 0)
    capped = np.flatnonzero((st & _hip.ST_MAX_ATTEMPTS) != 0)
    family = np.flatnonzero((it == 99) & ((st & _hip.ST_MAX_ATTEMPTS) == 0) & ((st & _hip.ST_NOT_PD) != 0))
    light = retried[np.argsort(it[retried], kind="stable")][:n_group]
    groups = collections.OrderedDict([("in order", np.arange(n_order)), ("Cholesky retries, few iterations", light),
                                      ("100-iteration family", family[:n_group]), ("attempt cap", capped[:n_group])])
    assert all(len(g) >= min(n_group, 4) for g in groups.values())
    pick = []
    for g in groups.values():
        pick += [int(b) for b in g if int(b) not in pick]
    heavy = set(int(b) for b in family[:n_group]) | set(int(b) for b in capped[:n_group])
    whole = set(int(b) for b in family[:every_pass_of]) | set(int(b) for b in capped[:every_pass_of])
    # the picked instances as a batch of their own: same bits per instance as in the 65 536 batch (the kernel owns one instance per wave);
    # this launch also records the box-QP clamp masks (tfmpc_ilqr_solve_trace_qp_f32)
    sub = dict(w, F=w["F"][pick], f=w["f"][pick], C=w["C"][pick], c=w["c"][pick], x0=w["x0"][pick].contiguous(), u0=w["u0"][pick].contiguous())
    make = lambda **kw: workloads.solver_of(sub, **kw)
    sub_full = make().solve_device(sub["x0"], sub["T"], u_init=sub["u0"], trace_rows=rows, qp_masks=True)
    torch.cuda.synchronize()
    for key in ("states", "actions", "costs", "iterations", "status", "trace_len"):
        assert torch.equal(sub_full[key], full[key][pick]), key
    assert torch.equal(torch.nan_to_num(sub_full["trace"]), torch.nan_to_num(full["trace"][pick]))
    dev = trace_records(sub_full["trace"], sub_full["trace_len"])
    masks, qp_it = _np(sub_full["clamp_mask"]), _np(sub_full["qp_iterations"])
    m = int(sub["F"].shape[-1] - sub["F"].shape[-2])
    for i, r in enumerate(dev):                         # every pass the device made has a mask at every step, and a QP that iterated
        assert not np.any(qp_it[i, :len(r)] == 0xFF) and int(qp_it[i, :len(r)].min()) >= 1 and int(qp_it[i, :len(r)].max()) <= 100
        assert np.all(masks[i, :len(r)] < (1 << m))
    k_max = int(max(r[-1]["iteration"] for r in dev)) + 1
    N = _nominals(make, sub["x0"], sub["u0"], sub["T"], min(k_max, 100), rows, sub_full)
    final = (_np(sub_full["states"])[..., 0].astype(np.float64), _np(sub_full["actions"])[..., 0].astype(np.float64), _np(sub_full["costs"]).astype(np.float64))
    cfgs = [workloads.instance_cfg(w, b) for b in pick]
    caps = [10 ** 6 if b in whole else (cap_heavy if b in heavy else cap_light) for b in pick]
    # traj_floor: the box-QP stops when an iteration improves its objective by less than 1e-8 of its value (optimization.py:13,27-29), which
    # pins its minimiser to ~1e-4 only; two fp32 programs can end an iterate apart (tests/test_ilqr_lq_trace_gpu.py)
    saved = dict(cfgs=cfgs, dev=dev, N=N, final=final, caps=caps, masks=masks, m=m, pick=pick,
                 groups=collections.OrderedDict((k, [int(b) for b in v]) for k, v in groups.items()))
    if dump is not None:
        import pickle
        with open(dump, "wb") as fh:
            pickle.dump(saved, fh)
        return None
    return _control_limited_compare(log=log, workers=workers, mismatch_share=mismatch_share, **saved)


def _control_limited_compare(cfgs, dev, N, final, caps, masks, m, pick, groups, log=None, workers=None, mismatch_share=0.0):
    if workers:
        tf.run_passes.__defaults__ = (workers,)
    stats, details, where, res, verdicts = _run("lq", cfgs, dev, N, final, caps, traj_floor=5e-4, label="control-limited, teacher-forced on the device's free sets",
                                                masks=masks, m=m)
    pos = {b: i for i, b in enumerate(pick)}
    lines = [f"control-limited, teacher-forced on the device's box-QP free sets: {dict(stats)}"]
    per_group = {}
    for name, members in groups.items():
        idx = set(pos[int(b)] for b in members)
        sel = [v for (i, p, d, nxt), v in zip(where, verdicts) if i in idx]
        c = per_group[name] = collections.Counter([("decision " + v[0]) for v in sel] + [("numbers " + v[1]) for v in sel])
        c["passes"] = len(sel)
        lines.append(f"  {name}: {len(sel)} passes compared of {sum(len(dev[i]) for i in idx)}: {dict(c)}")
    lines += [f"    {det}" for det in details[:40]]
    lines += [f"    {det}" for det in details[40:] if "mismatch" in det[2] or "unposed" in det[2]]       # (every pass that is not ok / loose, wherever it sits)
    print("\n".join(lines[1:]))
    if log:
        with open(log, "w") as fh:
            fh.write("\n".join(lines) + "\n")
    assert stats["decision mismatch"] == 0, details[:5]
    # mismatch_share > 0 (the slow, every-pass sampling only): a share of passes of the HEAVY groups may disagree in numbers -- measured: 4 of 2 059,
    # all one attempt-cap instance (costs ~1e12: fp32 cannot pose it) whose box-QP the device and the restatement leave on different iterates at four
    # time steps -- the QP stops when a step improves its objective by < 1e-8 of its value (optimization.py:27-29), here ~1e4 absolute -- so that the
    # feed-forward k_t, which the device does not export, and with it g_norm differ by 4.6 %; K_t, the candidates and every decision agree.
    order = set(pos[int(b)] for b in groups.get("in order", []))
    bad = [i for (i, p, d, nxt), v in zip(where, verdicts) if v[1] == "mismatch"]
    assert len(bad) <= int(mismatch_share * stats["passes"]) and not (set(bad) & order), (len(bad), details[:5])
    assert stats["numbers excused"] == 0                                    # nothing to excuse: the discrete part came from the device
    # "loose" = between 1 x and 4 x the tolerance (5 x the fp32 restatement's own error against fp64); the instances fp32 cannot pose -- the
    # 100-iteration family and the attempt-cap group, costs of 1e12 .. 1e21 -- carry most of them (measured: 15 of 1 151 passes, 14 of them there)
    assert stats["numbers loose"] <= max(2, stats["passes"] // 50), details[:5]
    assert stats["numbers ok"] + stats["numbers loose"] >= 0.97 * stats["passes"], dict(stats)      # (the rest: "unposed" in fp32)
    # decisions: every pass of the heavy groups is a near-tie by construction (fp32 has lost those problems: costs of 1e12 .. 1e21), so the
    # share of clear-margin decisions is asked of the instances taken in order
    assert per_group["in order"]["decision same"] >= 0.5 * per_group["in order"]["passes"], dict(per_group["in order"])
    return stats


@pytest.mark.gpu
def test_control_limited_workload_teacher_forced_on_the_devices_free_sets():
    """The default run: a tenth of the full sampling below (GPU-suite wall time; the restatement costs seconds per pass on the heavy
    instances) -- 40 instances in order (among them #1 and #38, whose passes 1 and 4 round 5 had to EXCUSE at 476 x and 665 x the tolerance:
    on the device's free sets they are inside it), 4 of each heavy kind."""
    _control_limited(n_order=40, n_group=4, cap_light=12, cap_heavy=6)


@pytest.mark.gpu
@pytest.mark.slow
def test_control_limited_workload_every_pass_teacher_forced():
    """TFMPC_SLOW=1: 80 instances in order, 16 of each heavy kind, EVERY pass of four instances of the 100-iteration family and of four at the
    attempt cap; the statistics go to gpurun_out/ (copied to profiles/ by hand)."""
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    _control_limited(n_order=80, n_group=16, cap_light=24, cap_heavy=12, every_pass_of=4, log="gpurun_out/r06_teacher_forced_control_limited.txt",
                     mismatch_share=0.0025)


@pytest.mark.gpu
def test_reservoir_cfg5_every_pass_teacher_forced():
    w = workloads.cfg5("reservoir", 32768)              # bench.py's own draw; 64 of its instances, spread over the batch
    rows = 40
    pick = [int(b) for b in np.arange(0, 32768, 512)]
    with _hip.option("TFMPC_ILQR_KERNEL", "costate_mfma"):
        solver = iLQR(w["env"], max_iterations=12)
        whole = solver.solve_device(w["x0"], w["T"], u_init=w["u0"], trace_rows=rows)
        torch.cuda.synchronize()
        assert solver.last_kernel.startswith("costate_mfma"), solver.last_kernel
        x0, u0 = w["x0"][pick].contiguous(), w["u0"][pick].contiguous()
        full = solver.solve_device(x0, w["T"], u_init=u0, trace_rows=rows)
        torch.cuda.synchronize()
        # 64 instances as a batch of their own (another wave grouping): the same bits per instance as inside the 32 768 batch
        for key in ("states", "actions", "costs", "iterations", "status", "trace_len"):
            assert torch.equal(full[key], whole[key][pick]), key
        assert torch.equal(torch.nan_to_num(full["trace"]), torch.nan_to_num(whole["trace"][pick]))
        del whole
        dev = trace_records(full["trace"], full["trace_len"])
        k_max = int(max(r[-1]["iteration"] for r in dev)) + 1
        N = _nominals(lambda **kw: iLQR(w["env"], **dict(dict(max_iterations=12), **kw)), x0, u0, w["T"], min(k_max, 12), rows, full)
    final = (_np(full["states"])[..., 0].astype(np.float64), _np(full["actions"])[..., 0].astype(np.float64), _np(full["costs"]).astype(np.float64))
    stats, details, where, res, verdicts = _run("reservoir", w["cfg"], dev, N, final, [40] * 64, low=0.0, high=1.0, label="Reservoir n=32 T=100, teacher-forced")
    assert stats["passes"] == sum(len(r) for r in dev)                      # every pass of every instance
    # the restatement's own selector against the device's, entry by entry, wherever |Q_u,i| is clear of rounding (ilqr.py:140-141)
    decided = agree = entries = 0
    for (i, p, d, nxt), r in zip(where, res):
        f64 = r["float64"]["free"]
        if nxt is None or f64 is None or "k" not in f64:
            continue
        du = nxt[1] - N[d["iteration"]][1][i]
        moved = du != 0
        clear = moved & (np.abs(f64["k"]) > 0) & (f64["selector_margin"] > 1e-3)
        entries += moved.sum()
        decided += clear.sum()
        agree += ((du > 0) == (f64["k"] > 0))[clear].sum()
    print(f"  selector: {agree} of {decided} clear-margin entries agree ({entries} moved entries in all)")
    assert decided > 0.5 * entries and agree == decided
    assert stats["decision mismatch"] == 0, details[:5]
    assert stats["numbers mismatch"] == 0, details[:5]
    assert stats["numbers loose"] <= max(2, stats["passes"] // 100), details[:5]
    assert stats["numbers ok"] + stats["numbers loose"] >= 0.9 * stats["passes"], dict(stats)
