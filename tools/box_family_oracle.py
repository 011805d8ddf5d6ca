"""What the fp32 / fp64 restatements do on the heavy instances of bench.py's control-limited workload:
instances with Cholesky retries, the family that runs all 100 iterations, and instances at the attempt cap -- first passes of the
device trace (matrix-core box kernel) side by side with the two free-running restatements, and the restatement's decision margin
(tests/trace_oracle.py).  Two phases, because the restatements need minutes of CPU per heavy instance and no GPU:

    python tools/box_family_oracle.py device     # on the GPU box: solve, pick instances, save gpurun_out/box_family_device.npz
    python tools/box_family_oracle.py oracle     # anywhere: restatements on those instances -> profiles/r04_box_family_oracle.json
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tf-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np

NPZ = os.path.join(ROOT, "gpurun_out", "box_family_device.npz")
B, ROWS = 65536, 170


def device():
    import torch
    import workloads
    from tfmpc import _hip
    w = workloads.control_limited(B)
    out = workloads.solver_of(w).solve_device(w["x0"], w["T"], u_init=w["u0"], trace_rows=ROWS)
    torch.cuda.synchronize()
    st, it = out["status"].cpu().numpy(), out["iterations"].cpu().numpy()
    retried = np.flatnonzero((st & _hip.ST_NOT_PD) != 0)
    capped = np.flatnonzero((st & _hip.ST_MAX_ATTEMPTS) != 0)
    family = np.flatnonzero((it == 99) & ((st & _hip.ST_MAX_ATTEMPTS) == 0) & ((st & _hip.ST_NOT_PD) != 0))
    light = retried[np.argsort(it[retried], kind="stable")][:12]
    pick = np.concatenate([light, family[:6], capped[:4]])
    group = np.array(["retries_few_iterations"] * len(light) + ["family_100_iterations"] * len(family[:6]) + ["attempt_cap"] * len(capped[:4]))
    os.makedirs(os.path.dirname(NPZ), exist_ok=True)
    np.savez(NPZ, pick=pick, group=group, trace=out["trace"][pick.tolist()].cpu().numpy(), trace_len=out["trace_len"][pick.tolist()].cpu().numpy(),
             iterations=it[pick], status=st[pick], final_cost=out["costs"][pick.tolist()].sum(dim=1).cpu().numpy(),
             counts=np.array([len(retried), len(family), len(capped)]))
    print("saved", NPZ, "retried / family / capped in the batch:", len(retried), len(family), len(capped))


def oracle():
    import problems
    import trace_oracle
    d = np.load(NPZ)
    pick = d["pick"]
    F, f, C, c, x0 = problems.make_lqr_batch_fast(B, 16, 8, seed=4321)          # tests/workloads.py:control_limited
    F = 0.25 * F
    cfgs = [dict(F=F[b], f=f[b], C=C[b], c=c[b], low=-0.5, high=0.5) for b in pick]
    x0p = x0[pick][..., None].astype(np.float32)
    u0p = np.zeros((len(pick), 50, 8, 1), dtype=np.float32)
    P = len(pick)
    runs = trace_oracle.run_many("lq", cfgs * 2, np.concatenate([x0p] * 2), np.concatenate([u0p] * 2), 50, ["float32"] * P + ["float64"] * P, 100,
                                 workers=max(1, (os.cpu_count() or 2) - 1))
    keys = ("iteration", "mu", "delta", "J_hat", "g_norm", "alpha_index", "alpha", "J", "accepted", "residual")
    def dev_rows(i):
        rows = []
        for r in d["trace"][i, :int(d["trace_len"][i])]:
            searched = r[8] >= 0
            rows.append(dict(iteration=int(r[0]), mu=float(r[1]), g_norm=float(r[4]), alpha_index=int(r[5]) if searched else None,
                             accepted=bool(r[8] > 0) if searched else None, J_hat=float(r[3]), J=float(r[7]) if searched else None))
        return rows
    short = lambda r: None if r is None else dict(iteration=r["iteration"], mu=r["mu"], g_norm=r["g_norm"], alpha_index=r["alpha_index"], accepted=r["accepted"],
                                                  J_hat=r["J_hat"], J=r["J"], cholesky_failures=r.get("cholesky_failures"), margin=r.get("margin"))
    report = {"batch_counts": dict(zip(("instances_with_cholesky_retries", "family_100_iterations", "attempt_cap"), map(int, d["counts"])))}
    for i, b in enumerate(pick):
        r32, r64, dv = runs[i], runs[P + i], dev_rows(i)
        def agree(a, rows):        # passes from the start on which two traces took the same decisions
            n = 0
            for x, y in zip(a, rows):
                if (x["iteration"], x["alpha_index"], x["accepted"]) != (y["iteration"], y["alpha_index"], y["accepted"]):
                    break
                n += 1
            return n
        entry = dict(instance=int(b), device_iterations=int(d["iterations"][i]) + 1, device_status=int(d["status"][i]), device_passes=len(dv),
                     fp32_iterations=None if r32 is None else r32[4] + 1, fp32_passes=None if r32 is None else len(r32[0]),
                     fp64_iterations=None if r64 is None else r64[4] + 1, fp64_passes=None if r64 is None else len(r64[0]),
                     device_final_cost=float(d["final_cost"][i]), fp32_final_cost=None if r32 is None else float(np.sum(r32[3])),
                     fp64_final_cost=None if r64 is None else float(np.sum(r64[3])),
                     leading_passes_with_equal_decisions=dict(device_vs_fp32=agree(dv, r32[0]) if r32 else None, device_vs_fp64=agree(dv, r64[0]) if r64 else None,
                                                              fp32_vs_fp64=agree(r32[0], r64[0]) if r32 and r64 else None),
                     fp32_first_pass_with_margin_below_1=None if r32 is None else next((p for p, r in enumerate(r32[0]) if r["margin"] < 1), None),
                     first_passes=[dict(device=short(dv[p]) if p < len(dv) else None, fp32=short(r32[0][p]) if r32 and p < len(r32[0]) else None,
                                        fp64=short(r64[0][p]) if r64 and p < len(r64[0]) else None) for p in range(3)])
        report.setdefault(str(d["group"][i]), []).append(entry)
        e = entry
        print(d["group"][i], int(b), "iterations dev/fp32/fp64", e["device_iterations"], e["fp32_iterations"], e["fp64_iterations"], "| passes", e["device_passes"],
              e["fp32_passes"], e["fp64_passes"], "| final cost", round(e["device_final_cost"], 2), e["fp32_final_cost"] and round(e["fp32_final_cost"], 2),
              e["fp64_final_cost"] and round(e["fp64_final_cost"], 2), "| equal leading decisions", e["leading_passes_with_equal_decisions"],
              "| first fp32 margin < 1 at pass", e["fp32_first_pass_with_margin_below_1"], flush=True)
    json.dump(report, open(os.path.join(ROOT, "profiles", "r04_box_family_oracle.json"), "w"), indent=1)


if __name__ == "__main__":
    {"device": device, "oracle": oracle}[sys.argv[1]]()
