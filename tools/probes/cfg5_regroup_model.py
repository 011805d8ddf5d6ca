"""Offline model of regrouping the 16 columns of a cfg5 wave between iterations (input: tools/probes/cfg5_trace.py).
Cost unit: one rollout time step of a wave.  A wave's sweep costs SWEEP * T; a line-search pass runs until every column
still trying is above J_hat (early stop, granularity 4 steps) or to T if some column's cost stays below (it accepts, or
its last try); HVAC (NA = 2) tries two step sizes per pass and rolls the accepted one out again."""
import sys, numpy as np
T = 100
def wave_cost(acc, fa, NA, sweep_w, reroll):
    """acc[16] accepted index or -1, fa[16][11] first-above step (T+1 = never, -1 = not tried)."""
    cost = sweep_w * T
    tried_any = (fa >= 0).any(axis=1)
    if not tried_any.any(): return cost                      # converged on g_norm / done columns only
    last = np.where(acc >= 0, acc, np.where(tried_any, 10, -1))
    for ai in range(0, 11, NA):
        trying = last >= ai
        if not trying.any(): break
        full = False; longest = 0
        for k in range(NA):
            i = ai + k
            if i > 10: break
            f = fa[trying, i]
            f = f[f >= 0]
            if (f > T).any(): full = True
            elif f.size: longest = max(longest, int(f.max()))
        cost += T if full else min(T, (longest // 4 + 1) * 4) * 1.0
    if reroll and (acc >= 0).any(): cost += T
    return cost
def total(tr, order_fn, NA, sweep_w, reroll):
    B, S = tr.shape[0], tr.shape[1]
    tot = 0.0
    prev = np.zeros(B, dtype=np.int64)
    for s in range(S):
        acc, fa = tr[:, s, 0].astype(np.int64), tr[:, s, 1:].astype(np.int64)
        active = (fa >= 0).any(axis=1)
        if not active.any(): break
        order = order_fn(s, prev, acc, active)
        for w in range(0, len(order), 16):
            idx = order[w:w + 16]
            tot += wave_cost(acc[idx], fa[idx], NA, sweep_w, reroll)
        prev = np.where(active, np.where(acc >= 0, acc, 11), prev)
    return tot
for kind, NA, sweep_w, reroll in (("hvac", 2, 1.0, True), ("reservoir", 1, 1.5, False)):
    tr = np.load(f"gpurun_out/cfg5_trace_{kind}.npy")[:8192]
    B = tr.shape[0]
    acc = tr[:, :, 0]
    print(kind, "accepted-index histogram per sweep (rows = sweep):")
    for s in range(13):
        a = acc[:, s]; tried = (tr[:, s, 1:] >= 0).any(axis=1)
        if not tried.any(): break
        h = np.bincount(a[tried & (a >= 0)], minlength=11)
        print(f"  sweep {s:2d}: tried {tried.sum():6d} none-accepted {(tried & (a < 0)).sum():5d} ", h.tolist())
    fixed = total(tr, lambda s, prev, acc, act: np.arange(B), NA, sweep_w, reroll)
    pred = total(tr, lambda s, prev, acc, act: np.argsort(np.where(act, prev, 99), kind="stable")[:act.sum() + (-act.sum()) % 16], NA, sweep_w, reroll)
    best = total(tr, lambda s, prev, acc, act: np.argsort(np.where(act, np.where(acc >= 0, acc, 11), 99), kind="stable")[:act.sum() + (-act.sum()) % 16], NA, sweep_w, reroll)
    per = lambda tot: tot / (B / 16) / 12               # wave-steps per wave and iteration (12 iterations)
    print(f"{kind}: fixed groups {per(fixed):.0f} wave-steps per iteration, regrouped by last accepted index {per(pred):.0f} "
          f"({fixed / pred:.2f}x), by the true index (bound) {per(best):.0f} ({fixed / best:.2f}x)")
