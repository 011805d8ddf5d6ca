"""Control-limited iLQR (Tassa, Mansard, Todorov 2014) on MI355X -- drop-in for the
reference's ``tfmpc/solvers/ilqr.py`` (``iLQR.__init__`` :24-43, ``start`` :53-82,
``derivatives`` :84-92, ``backward`` :94-172, ``forward`` :174-212, ``solve`` :214-283).

Same class, method and argument names, defaults and return arity as the reference.
Additions: every tensor may carry one leading batch axis ``B`` of independent problem
instances; ``start`` / ``solve`` take ``u_init`` / ``seed`` because the reference's
initial actions come from TensorFlow's RNG (``ilqr.py:70``), which nothing else can
reproduce.  All arithmetic runs in gfx950 HIP kernels through the C ABI
(``include/tfmpc_hip.h``); ``solve`` is ONE kernel launch in which each wavefront runs
its instance's whole iteration loop, so there is no per-iteration host round trip.

Reference quirks kept on purpose: one scalar uniform per
timestep in ``start`` (Q1); the regularisation bump after a Cholesky failure is local
to the backward retry (Q2); ``residual < atol`` accepts a step the line search rejected
(Q3); 11 step sizes (Q4); ``backward`` defaults to ``mu=1.0`` while ``solve`` starts at
0 (Q5).  Deviation: the rejected-attempt loop is capped (``max_attempts``, status bit
``ST_MAX_ATTEMPTS``) where the reference would loop forever.
"""

import ctypes
import logging
import os

import numpy as np
import torch

from tfmpc import _hip
from tfmpc.envs.diffenv import CostApprox, FinalCostApprox, TransitionApprox
from tfmpc.utils import trajectory


TRACE_COLUMNS = ("iteration", "mu", "delta", "J_hat", "g_norm", "alpha_index", "alpha", "J", "accepted", "residual", "level")


def trace_records(trace, trace_len):
    """The device trace of ``solve_device(trace_rows=...)`` as lists of dicts, one list per instance and one dict per
    pass -- the fields the oracle's ``ILQRRef.trace`` holds (oracle/ilqr_ref.py; reference: ilqr.py:238-279).  A pass that
    ended on ``g_norm < atol`` made no line search: ``alpha_index``, ``alpha``, ``J``, ``accepted``, ``residual`` are None."""
    tr = trace.detach().cpu().numpy()
    ln = trace_len.detach().cpu().numpy()
    out = []
    for b in range(tr.shape[0]):
        rows = []
        for r in tr[b, :min(int(ln[b]), tr.shape[1])]:
            searched = r[8] >= 0
            rows.append(dict(iteration=int(r[0]), mu=float(r[1]), delta=float(r[2]), J_hat=float(r[3]), g_norm=float(r[4]),
                             alpha_index=int(r[5]) if searched else None, alpha=float(r[6]) if searched else None,
                             J=float(r[7]) if searched else None, accepted=bool(r[8] > 0) if searched else None,
                             residual=float(r[9]) if searched else None,
                             level=int(r[10])))      # local regularisation bumps before the pass factorised (ilqr.py:305-309)
        out.append(rows)
    return out


def trace_log_lines(records):
    """One instance's trace (``trace_records(...)[b]``) as the lines the reference writes to ``<logdir>/trace.log`` while it
    solves (ilqr.py:229, 244-257, 301-303, 332-333): iteration banner, ``[BACKWARD] J_hat``, ``[SOLVE] g_norm``, the
    ``[FORWARD]`` line of the step size the line search ended on, the convergence messages."""
    lines, last = [], None
    for r in records:
        if r["iteration"] != last:
            lines.append(f"[SOLVE] >>>>>>> Iteration = {r['iteration']} <<<<<<<")
            last = r["iteration"]
        lines.append(f"[BACKWARD] mu = {r['mu']}, delta = {r['delta']}, J_hat = {r['J_hat']:.4f}")
        lines.append(f"[SOLVE] g_norm = {r['g_norm']:.6f}")
        if r["accepted"] is None:
            lines.append("[SOLVE] CONVERGED: g_norm < atol")
            continue
        lines.append(f"[FORWARD] num_iter = {r['alpha_index'] + 1}, alpha = {r['alpha']}, J = {r['J']:.4f}, residual = {r['residual']}, "
                     f"accept = {r['accepted']}")
    return lines


class _Graphed:
    """A block of torch work on fixed shapes, captured once as a hipGraph (``torch.cuda.CUDAGraph``) and replayed: the
    generic-env path's line-search rollouts are T steps of a dozen small element-wise launches each, its derivatives a few
    hundred -- launch-bound from the host.  ``fn(*tensors) -> tuple of tensors``; inputs are copied into static buffers, the
    outputs are the graph's static tensors (valid until the next call).  A block that cannot be captured (a host
    synchronisation or data-dependent control flow inside the user's env functions) runs eagerly from then on."""

    def __init__(self, fn):
        self.fn, self.key, self.graph, self.inputs, self.outputs, self.eager = fn, None, None, None, None, False

    def __call__(self, *args):
        if self.eager or not args[0].is_cuda:
            return self.fn(*args)
        key = tuple((tuple(a.shape), a.dtype) for a in args)
        if key != self.key:
            try:
                self._capture(args, key)
            except RuntimeError as exc:          # what a failed capture raises (an unsupported operation or a synchronisation inside)
                import warnings
                torch.cuda.synchronize()
                self.eager, self.graph = True, None
                warnings.warn(f"tfmpc: this block cannot be captured as a hipGraph and runs eagerly from now on ({exc!s:.200})",
                              RuntimeWarning, stacklevel=2)
                return self.fn(*args)
        for dst, src in zip(self.inputs, args):
            dst.copy_(src)
        self.graph.replay()
        return self.outputs

    def _capture(self, args, key):
        self.inputs = [a.clone() for a in args]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):                  # warm-up: torch.func traces, allocator, lazy initialisation
                self.fn(*self.inputs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self.outputs = self.fn(*self.inputs)
        self.graph, self.key = graph, key


def _f32(a, device):
    if isinstance(a, torch.Tensor):
        return a.detach().to(device=device, dtype=torch.float32)
    return torch.as_tensor(np.asarray(a, dtype=np.float32), device=device)


class iLQR:

    def __init__(self, env, **kwargs):
        # build-only: an env given as PLAIN TORCH FUNCTIONS (tfmpc.envs.torchenv.TorchEnv) is translated to device source and compiled into the fused
        # kernels when that is possible (TorchEnv.to_device_env: two orders of magnitude over the host-driven loop below) -- compile_env="auto"
        # (default): try, and fall back to the host-driven loop when the functions use an operation the translator does not know, branch in Python on
        # the state, or no hipcc is there (the reason is kept in `self.compile_error`); True: raise instead of falling back; False: host-driven loop.
        self.python_env, self.compile_error = None, None
        compile_env = kwargs.get("compile_env", "auto")
        if compile_env == "auto" and not getattr(env, "auto_compile", True):       # (TorchEnv(..., auto_compile=False): this env stays on the host-driven loop)
            compile_env = False
        if compile_env and hasattr(env, "to_device_env") and getattr(env, "kind", 0) is None:
            try:
                device_env = env.to_device_env()
                device_env._library()                       # (compiles, or finds the library in the cache)
                self.python_env, env = env, device_env
            except Exception as exc:                        # noqa: BLE001 -- whatever stops the translation: the host-driven loop serves the env
                if compile_env is True:
                    raise
                self.compile_error = exc
        self.env = env

        # solve
        self.atol = kwargs.get("atol", 5e-3)
        self.max_iterations = kwargs.get("max_iterations", 100)
        # backward
        self.mu_min = kwargs.get("mu_min", 1e-6)
        self.delta_0 = kwargs.get("delta_0", 2.0)
        # forward
        self.c1 = kwargs.get("c1", 0.0)
        self.alpha_min = kwargs.get("alpha_min", 1e-3)
        # build-only: cap on rejected attempts per solve (the reference has none)
        self.max_attempts = kwargs.get("max_attempts", 64)
        # build-only: keep trajectories / gains in HBM at bf16 precision (fp32 arithmetic); wave kernels only
        self.storage_bf16 = bool(kwargs.get("storage_bf16", False))
        # build-only, generic envs (TorchEnv): replay the rollout / derivative blocks of the host-driven loop as hipGraphs
        self.graphs = bool(kwargs.get("graphs", True))
        self._graphed = {}

        self._config = kwargs
        self.last_status = None
        if "logdir" in self._config:
            logging.basicConfig(filename=os.path.join(self._config["logdir"], "trace.log"), level=logging.DEBUG)

    # -- bounds (ilqr.py:45-51) --------------------------------------------------------
    @property
    def device(self):
        return self.env._device()

    def _bound(self, name):
        # (one host-to-device copy per solver, not per use: the rollouts of a generic env are replayed as a hipGraph)
        cache = self.__dict__.setdefault("_bounds", {})
        key = (name, str(self.device))
        if key not in cache:
            cache[key] = torch.as_tensor(np.asarray(getattr(self.env.action_space, name), dtype=np.float32), device=self.device)
        return cache[key]

    @property
    def low(self):
        return self._bound("low")

    @property
    def high(self):
        return self._bound("high")

    # -- helpers --------------------------------------------------------------------------
    def _library(self):
        """The library that serves the env's kernels: the product library, or a DeviceEnv's companion library (envs/deviceenv.py)."""
        hook = getattr(self.env, "_library", None)
        return hook() if hook is not None else _hip.require_gpu()

    def _alphas(self):
        cached = getattr(self, "_alphas_cache", None)                      # (numpy's geomspace is ~20 us: half of a solve_device call's host time)
        if cached is None or cached[0] != self.alpha_min:
            cached = self._alphas_cache = (self.alpha_min, np.geomspace(1.0, self.alpha_min, 11))      # ilqr.py:322
        return cached[1]

    def _c_config(self):
        cfg = _hip.TfmpcIlqrConfig()
        cfg.atol, cfg.max_iterations = float(self.atol), int(self.max_iterations)
        cfg.mu_min, cfg.delta_0, cfg.c1 = float(self.mu_min), float(self.delta_0), float(self.c1)
        alphas = self._alphas()
        cfg.n_alphas = len(alphas)
        for i, a in enumerate(alphas):
            cfg.alphas[i] = float(a)
        cfg.max_attempts = int(self.max_attempts)
        cfg.storage_bf16 = int(self.storage_bf16)
        return cfg

    def _batch_cols(self, a, size, lead):
        """[.., size(,1)] with `lead` leading axes besides the optional batch axis ->
        (contiguous tensor [B, *lead, size], batched?)"""
        t = _f32(a, self.device)
        if t.shape[-1] == 1 and t.dim() >= 2 and t.shape[-2] == size:
            t = t.squeeze(-1)
        if t.shape[-1] != size or t.dim() not in (lead + 1, lead + 2):
            raise ValueError(f"bad shape {tuple(np.shape(a))} for a size-{size} vector with {lead} leading axes")
        batched = t.dim() == lead + 2
        return (t if batched else t.unsqueeze(0)).contiguous(), batched

    def random_actions(self, T, batch_size=None, seed=None):
        """``u_t = lo' + r_t (hi' - lo')`` with ONE scalar uniform per step and infinite
        bounds replaced by -1 / +1 (``ilqr.py:59-70``).  Returns ``[(B,) T, m, 1]``."""
        gen = torch.Generator(device="cpu")
        gen.manual_seed(int(np.random.randint(0, 2 ** 31 - 1)) if seed is None else int(seed))
        low, high = self.low.reshape(-1), self.high.reshape(-1)
        lo = torch.where(torch.isinf(low), -torch.ones_like(low), low)
        hi = torch.where(torch.isinf(high), torch.ones_like(high), high)
        r = torch.rand((batch_size or 1, int(T), 1), generator=gen).to(self.device)
        u = (lo + r * (hi - lo)).unsqueeze(-1)
        return u if batch_size else u[0]

    @property
    def _generic_env(self):
        """True for envs given as Python/torch functions (tfmpc.envs.torchenv.TorchEnv)."""
        return getattr(self.env, "kind", None) is None

    # -- ilqr.py:53-82 -----------------------------------------------------------------------
    def start(self, x0, T, u_init=None, seed=None):
        lib = self._library()
        T = int(T)
        if self._generic_env:
            return self._start_torch(x0, T, u_init, seed)
        n, m = self.env.state_size, self.env.action_size
        x0, batched = self._batch_cols(x0, n, 0)
        B = x0.shape[0]
        if u_init is None:
            u_init = self.random_actions(T, B if batched else None, seed)
        u, ub = self._batch_cols(u_init, m, 1)
        if ub and not batched:
            x0, batched, B = x0.expand(u.shape[0], n).contiguous(), True, u.shape[0]
        u = u.expand(B, T, m).contiguous()
        env, keep = self.env.c_env()
        states = torch.empty((B, T + 1, n), device=self.device)
        costs = torch.empty((B, T + 1), device=self.device)
        rc = lib.tfmpc_ilqr_rollout_f32(ctypes.byref(env), B, T, _hip.ptr(x0), _hip.ptr(u), _hip.ptr(states),
                                        _hip.ptr(costs), _hip.stream())
        _hip.check(rc, "tfmpc_ilqr_rollout_f32")
        states, actions = states.unsqueeze(-1), u.unsqueeze(-1)
        return (states, actions, costs) if batched else (states[0], actions[0], costs[0])

    # -- ilqr.py:84-92 -----------------------------------------------------------------------
    def derivatives(self, states, actions):
        lib = self._library()
        if self._generic_env:
            xs = _f32(states, self.device)
            us = _f32(actions, self.device)
            tdim = xs.dim() - 3
            return (self.env.get_linear_transition(xs.narrow(tdim, 0, xs.shape[tdim] - 1), us),
                    self.env.get_quadratic_cost(xs.narrow(tdim, 0, xs.shape[tdim] - 1), us),
                    self.env.get_quadratic_final_cost(xs.select(tdim, xs.shape[tdim] - 1)))
        n, m = self.env.state_size, self.env.action_size
        x, batched = self._batch_cols(states, n, 1)
        u, _ = self._batch_cols(actions, m, 1)
        B, T = x.shape[0], u.shape[1]
        if x.shape[1] != T + 1:
            raise ValueError("states must have one more step than actions")
        dev = self.device
        f, f_x, f_u = (torch.empty(s, device=dev) for s in ((B, T, n, 1), (B, T, n, n), (B, T, n, m)))
        l, l_x, l_u = (torch.empty(s, device=dev) for s in ((B, T), (B, T, n, 1), (B, T, m, 1)))
        l_xx, l_uu, l_ux, l_xu = (torch.empty(s, device=dev) for s in ((B, T, n, n), (B, T, m, m), (B, T, m, n), (B, T, n, m)))
        fl, fl_x, fl_xx = (torch.empty(s, device=dev) for s in ((B,), (B, n, 1), (B, n, n)))
        env, keep = self.env.c_env()
        outs = (f, f_x, f_u, l, l_x, l_u, l_xx, l_uu, l_ux, l_xu, fl, fl_x, fl_xx)
        rc = lib.tfmpc_ilqr_derivatives_f32(ctypes.byref(env), B, T, _hip.ptr(x), _hip.ptr(u),
                                            *[_hip.ptr(o) for o in outs], _hip.stream())
        _hip.check(rc, "tfmpc_ilqr_derivatives_f32")
        if not batched:
            outs = tuple(o[0] for o in outs)
        return TransitionApprox(*outs[:3]), CostApprox(*outs[3:10]), FinalCostApprox(*outs[10:])

    # -- ilqr.py:94-172 ----------------------------------------------------------------------
    def backward(self, T, actions, transition_model, cost_model, final_cost_model, mu=1.0):
        lib = _hip.require_gpu()
        T = int(T)
        n, m = self.env.state_size, self.env.action_size
        dev = self.device
        u, batched = self._batch_cols(actions, m, 1)
        B = u.shape[0]

        def mat(a, r, c_):      # [(B,) T, r, c] -> contiguous [B, T, r, c]
            t = _f32(a, dev)
            return (t if t.dim() == 4 else t.unsqueeze(0)).reshape(B, T, r, c_).contiguous()

        f_x, f_u = mat(transition_model.f_x, n, n), mat(transition_model.f_u, n, m)
        l = _f32(cost_model.l, dev).reshape(B, T).contiguous()
        l_x, l_u = mat(cost_model.l_x, n, 1), mat(cost_model.l_u, m, 1)
        l_xx, l_uu, l_xu = mat(cost_model.l_xx, n, n), mat(cost_model.l_uu, m, m), mat(cost_model.l_xu, n, m)
        fl = _f32(final_cost_model.l, dev).reshape(B).contiguous()
        fl_x = _f32(final_cost_model.l_x, dev).reshape(B, n).contiguous()
        fl_xx = _f32(final_cost_model.l_xx, dev).reshape(B, n, n).contiguous()
        mu_t = _f32(mu, dev).reshape(-1).contiguous()
        if mu_t.numel() not in (1, B):
            raise ValueError("mu must be a scalar or one value per instance")
        K = torch.empty((B, T, m, n), device=dev)
        k = torch.empty((B, T, m, 1), device=dev)
        J, dV1, dV2 = (torch.empty((B,), device=dev) for _ in range(3))
        status = torch.zeros((B,), dtype=torch.int32, device=dev)
        low, high = self.low.reshape(-1).contiguous(), self.high.reshape(-1).contiguous()
        rc = lib.tfmpc_ilqr_backward_f32(B, n, m, T, _hip.ptr(u), _hip.ptr(f_x), _hip.ptr(f_u), _hip.ptr(l),
                                         _hip.ptr(l_x), _hip.ptr(l_u), _hip.ptr(l_xx), _hip.ptr(l_uu), _hip.ptr(l_xu),
                                         _hip.ptr(fl), _hip.ptr(fl_x), _hip.ptr(fl_xx), _hip.ptr(low), _hip.ptr(high),
                                         int(self.env.action_space.is_bounded()), _hip.ptr(mu_t),
                                         1 if mu_t.numel() == B and B > 1 else 0,
                                         _hip.ptr(K), _hip.ptr(k), _hip.ptr(J), _hip.ptr(dV1), _hip.ptr(dV2),
                                         _hip.ptr(status), _hip.stream())
        _hip.check(rc, "tfmpc_ilqr_backward_f32")
        self.last_status = status
        if not batched:
            if int(status[0]) & _hip.ST_NOT_PD:     # the reference raises tf.errors.InvalidArgumentError here
                raise ArithmeticError("iLQR.backward: Q_uu (regularised) is not positive definite")
            return K[0], k[0], J[0], dV1[0], dV2[0]
        return K, k, J, dV1, dV2

    # -- ilqr.py:174-212 ---------------------------------------------------------------------
    def forward(self, x, u, K, k, alpha=1.0):
        lib = self._library()
        n, m = self.env.state_size, self.env.action_size
        dev = self.device
        if self._generic_env:
            return self._forward_torch(x, u, K, k, alpha)
        xs, batched = self._batch_cols(x, n, 1)
        us, _ = self._batch_cols(u, m, 1)
        B, T = xs.shape[0], us.shape[1]
        Kt = _f32(K, dev)
        Kt = (Kt if Kt.dim() == 4 else Kt.unsqueeze(0)).expand(B, T, m, n).contiguous()
        kt, _ = self._batch_cols(k, m, 1)
        kt = kt.expand(B, T, m).contiguous()
        al = _f32(alpha, dev).reshape(-1).contiguous()
        if al.numel() not in (1, B):
            raise ValueError("alpha must be a scalar or one value per instance")
        states = torch.empty((B, T + 1, n), device=dev)
        actions = torch.empty((B, T, m), device=dev)
        costs = torch.empty((B, T + 1), device=dev)
        J, residual = torch.empty((B,), device=dev), torch.empty((B,), device=dev)
        env, keep = self.env.c_env()
        rc = lib.tfmpc_ilqr_forward_f32(ctypes.byref(env), B, T, _hip.ptr(xs), _hip.ptr(us), _hip.ptr(Kt), _hip.ptr(kt),
                                        _hip.ptr(al), 1 if al.numel() == B and B > 1 else 0, _hip.ptr(states),
                                        _hip.ptr(actions), _hip.ptr(costs), _hip.ptr(J), _hip.ptr(residual), _hip.stream())
        _hip.check(rc, "tfmpc_ilqr_forward_f32")
        states, actions = states.unsqueeze(-1), actions.unsqueeze(-1)
        if not batched:
            return states[0], actions[0], costs[0], J[0], residual[0]
        return states, actions, costs, J, residual

    # -- fused solve, device tensors in/out -----------------------------------------------------
    def solve_device(self, x0, T, u_init=None, seed=None, workspace=None, trace_rows=0, qp_masks=False):
        """ONE kernel launch for B whole iLQR solves.  Returns a dict of device tensors:
        ``states[B,T+1,n,1]``, ``actions[B,T,m,1]``, ``costs[B,T+1]``, ``iterations[B]``
        (the reference's returned loop index) and ``status[B]``.  Never synchronises.

        ``trace_rows > 0``: also the decision trace of every instance -- ``trace[B, trace_rows, TRACE_COLS]`` (one row
        per backward pass + line search, columns ``TRACE_COLUMNS``; rows never written are NaN) and ``trace_len[B]``
        (passes made): what the reference logs per pass of ilqr.py:238-279 (``tfmpc_ilqr_solve_trace_f32``).

        ``qp_masks=True`` (with ``trace_rows``; ``tfmpc_ilqr_solve_trace_qp_f32``): also ``clamp_mask[B, trace_rows, T]`` (uint8: bit a set =
        action a clamped in the last factorised free set of that step's box-QP, i.e. row a of ``K_t`` is zero, ilqr.py:375-385) and
        ``qp_iterations[B, trace_rows, T]`` -- written by the control-limited matrix-core kernel; entries no kernel wrote stay 0xFF."""
        lib = self._library()
        T = int(T)
        n, m = self.env.state_size, self.env.action_size
        dev = self.device
        x0, batched = self._batch_cols(x0, n, 0)
        B = x0.shape[0]
        if u_init is None:
            u_init = self.random_actions(T, B if batched else None, seed)
        u, ub = self._batch_cols(u_init, m, 1)
        if ub and not batched:
            x0, batched, B = x0.expand(u.shape[0], n).contiguous(), True, u.shape[0]
        if self._generic_env:
            return self._solve_host_driven(x0, u.expand(B, T, m).contiguous(), batched, trace_rows=int(trace_rows))
        eb = self.env.env_batch_size()
        if eb is not None:
            if batched and eb != B:
                raise ValueError("x0 batch does not match the env's per-instance parameters")
            if not batched:
                x0, batched, B = x0.expand(eb, n).contiguous(), True, eb
        u = u.expand(B, T, m).contiguous()
        env, keep = self.env.c_env()
        # an env that makes a structural promise to the kernels (TfmpcEnv.coupling_shift) may be refused on the device
        # (TFMPC_ST_ENV_FLAG: nothing computed): its outputs then read as zeros, never as uninitialised memory
        alloc = torch.zeros if env.coupling_shift != 0 else torch.empty
        states = alloc((B, T + 1, n), device=dev)
        actions = alloc((B, T, m), device=dev)
        costs = alloc((B, T + 1), device=dev)
        iterations = torch.zeros((B,), dtype=torch.int32, device=dev)
        status = torch.zeros((B,), dtype=torch.int32, device=dev)
        # (what this env's kernels read: the shape-only size carries every env kind's optional parts -- 0.8 GB of them at the headline batch)
        sized = getattr(lib, "tfmpc_ilqr_workspace_bytes_for", None)
        ws_bytes = int(sized(ctypes.byref(env), B, T)) if sized is not None and not getattr(lib, "path", None) else int(lib.tfmpc_ilqr_workspace_bytes(B, n, m, T))
        if workspace is None or workspace.numel() * workspace.element_size() < ws_bytes:
            workspace = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=dev)
        cfg = self._c_config()
        trace = trace_len = None
        if trace_rows > 0:
            trace = torch.full((B, int(trace_rows), _hip.TRACE_COLS), float("nan"), dtype=torch.float32, device=dev)
            trace_len = torch.zeros((B,), dtype=torch.int32, device=dev)
        clamp = qp_it = None
        if qp_masks:
            if trace is None:
                raise ValueError("qp_masks needs trace_rows > 0 (the masks are indexed by pass like the trace rows)")
            if not hasattr(lib, "tfmpc_ilqr_solve_trace_qp_f32") or getattr(lib, "path", None):
                raise NotImplementedError("qp_masks: served by the main library's control-limited LQ kernel only")
            clamp = torch.full((B, int(trace_rows), T), 0xFF, dtype=torch.uint8, device=dev)
            qp_it = torch.full((B, int(trace_rows), T), 0xFF, dtype=torch.uint8, device=dev)
            rc = lib.tfmpc_ilqr_solve_trace_qp_f32(ctypes.byref(env), ctypes.byref(cfg), B, T, _hip.ptr(x0), _hip.ptr(u),
                                                   _hip.ptr(states), _hip.ptr(actions), _hip.ptr(costs), _hip.ptr(iterations),
                                                   _hip.ptr(status), _hip.ptr(trace), int(trace_rows), _hip.ptr(trace_len),
                                                   _hip.ptr(clamp), _hip.ptr(qp_it),
                                                   _hip.ptr(workspace), workspace.numel() * workspace.element_size(), _hip.stream())
        else:
            rc = lib.tfmpc_ilqr_solve_trace_f32(ctypes.byref(env), ctypes.byref(cfg), B, T, _hip.ptr(x0), _hip.ptr(u),
                                                _hip.ptr(states), _hip.ptr(actions), _hip.ptr(costs), _hip.ptr(iterations),
                                                _hip.ptr(status), _hip.ptr(trace), int(trace_rows), _hip.ptr(trace_len),
                                                _hip.ptr(workspace), workspace.numel() * workspace.element_size(), _hip.stream())
        _hip.check(rc, "tfmpc_ilqr_solve_trace_f32")
        self.last_status = status
        self.last_kernel = lib.tfmpc_ilqr_last_kernel_name().decode()      # which kernel family solved it (a traced solve can take another)
        out = dict(states=states.unsqueeze(-1), actions=actions.unsqueeze(-1), costs=costs, iterations=iterations,
                   status=status, batched=batched, workspace=workspace)
        if trace is not None:
            out.update(trace=trace, trace_len=trace_len)
        if clamp is not None:
            out.update(clamp_mask=clamp, qp_iterations=qp_it)
        return out

    # -- ilqr.py:214-283 ---------------------------------------------------------------------
    def solve(self, x0, T, show_progress=True, u_init=None, seed=None, trace=False, gather=False, total=None):
        """``(Trajectory, iteration)`` as the reference returns them (ilqr.py:281-283).

        ``gather=True`` under an initialised ``torch.distributed`` process group (one process per GPU, ``x0`` / ``u_init`` = THIS rank's
        block of the batch, ``tfmpc.parallel.shard``): the ranks solve their shards with no communication and ONE gather at the end
        (``tfmpc.parallel.gather_results``; with ``total`` = the global batch size it is literally one collective) brings
        ``(Trajectory[B_global], iterations[B_global])`` to rank 0 -- every other rank returns ``(None, None)``; the gathered
        ``status[B_global]`` is left in ``self.last_status``.  Without a process group ``gather`` changes nothing.

        ``trace=True`` also records what
        the reference logs while it solves (ilqr.py:243-279: per pass ``J_hat``, ``g_norm``, the step size the line
        search ended on, its ``J`` and ``residual``, ``mu`` / ``delta``) into ``self.last_trace`` -- a list of dicts
        per instance (``tfmpc.solvers.ilqr.trace_records``); with ``show_progress`` the progress bar's postfix
        (``J``, ``g_norm``, ``residual``, ilqr.py:279) of an unbatched solve is printed per pass to stderr."""
        rows = 0
        if trace:       # an iteration makes at most max_attempts rejected passes; almost all make one or two
            rows = int(self.max_iterations) + int(self.max_attempts) + 1
        out = self.solve_device(x0, T, u_init=u_init, seed=seed, trace_rows=rows)
        promised = hasattr(self.env, "c_env") and int(self.env.c_env()[0].coupling_shift) != 0       # (generic torch envs have no C struct)
        if promised and bool((out["status"] & _hip.ST_ENV_FLAG).any()):
            raise ValueError("the env's TfmpcEnv.coupling_shift promises a chain that its `downstream` matrix is not "
                             "(TFMPC_ST_ENV_FLAG): nothing was computed")
        if trace:
            self.last_trace = trace_records(out["trace"], out["trace_len"])
            if show_progress and not out["batched"]:
                import sys
                for r in self.last_trace[0]:
                    res = "" if r["residual"] is None else f", residual={r['residual']:.4f}"
                    print(f"[iLQR] iteration {r['iteration']}: J={r['J_hat']:.4f}, g_norm={r['g_norm']:.4f}{res}", file=sys.stderr)
        if gather:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                from tfmpc import parallel
                if not out["batched"]:
                    raise ValueError("gather=True shards a BATCH of instances over the ranks: x0 must carry the batch axis")
                res = parallel.gather_results(out["states"], out["actions"], out["costs"], iterations=out["iterations"],
                                              status=out["status"], total=total)
                if res is None:
                    return None, None
                states, actions, costs, iterations, self.last_status = res
                return trajectory.Trajectory(states, actions, costs), iterations.cpu().numpy()
        if out["batched"]:
            return trajectory.Trajectory(out["states"], out["actions"], out["costs"]), out["iterations"].cpu().numpy()
        traj = trajectory.Trajectory(out["states"][0], out["actions"][0], out["costs"][0])
        return traj, int(out["iterations"][0])


    # ===== generic (torch-function) envs: SURVEY.md §8f N2 ============================================
    # The reference's loop (ilqr.py:214-283) driven from the host over a whole batch: derivatives by
    # torch.func, the Riccati backward pass in the HIP kernel, rollouts as batched torch ops.  Every
    # instance carries its own (mu, delta, iteration, converged) exactly like the fused device kernels.
    def _rollout_torch(self, x0, u):
        """x0[R,n], u[R,T,m] -> states[R,T+1,n], costs[R,T+1]"""
        T = u.shape[1]
        xs, cs, x = [x0], [], x0
        for t in range(T):
            nx, c = self.env.step_flat(x, u[:, t])
            xs.append(nx)
            cs.append(c)
            x = nx
        cs.append(self.env.final_cost_flat(x))
        return torch.stack(xs, dim=1), torch.stack(cs, dim=1)

    def _start_torch(self, x0, T, u_init, seed):
        n, m = self.env.state_size, self.env.action_size
        x0, batched = self._batch_cols(x0, n, 0)
        B = x0.shape[0]
        if u_init is None:
            u_init = self.random_actions(T, B if batched else None, seed)
        u, ub = self._batch_cols(u_init, m, 1)
        if ub and not batched:
            x0, batched, B = x0.expand(u.shape[0], n).contiguous(), True, u.shape[0]
        u = u.expand(B, T, m).contiguous()
        states, costs = self._rollout_torch(x0, u)
        states, actions = states.unsqueeze(-1), u.unsqueeze(-1)
        return (states, actions, costs) if batched else (states[0], actions[0], costs[0])

    def _line_search_rollouts(self, xh, uh, K, k, alphas):
        """All step sizes at once: xh[B,T+1,n], uh[B,T,m], K[B,T,m,n], k[B,T,m], alphas[A] ->
        states[B,A,T+1,n], actions[B,A,T,m], costs[B,A,T+1], residual[B,A] (ilqr.py:174-212)."""
        B, T, m = uh.shape
        A, n = alphas.numel(), xh.shape[-1]
        low, high = self.low.reshape(-1), self.high.reshape(-1)
        x = xh[:, None, 0].expand(B, A, n).reshape(B * A, n)
        xs, us, cs = [x], [], []
        resid = torch.zeros(B * A, device=xh.device)
        for t in range(T):
            dx = x.reshape(B, A, n) - xh[:, None, t]
            du = alphas[None, :, None] * k[:, None, t] + torch.einsum("bij,baj->bai", K[:, t], dx)
            ua = torch.minimum(torch.maximum(uh[:, None, t] + du, low), high).reshape(B * A, m)
            nx, c = self.env.step_flat(x, ua)
            resid = torch.maximum(resid, du.abs().amax(dim=-1).reshape(B * A))
            us.append(ua)
            cs.append(c)
            xs.append(nx)
            x = nx
        cs.append(self.env.final_cost_flat(x))
        shape = lambda lst, w: torch.stack(lst, dim=1).reshape(B, A, len(lst), *( (w,) if w else ()))
        return shape(xs, n), shape(us, m), shape(cs, 0), resid.reshape(B, A)

    def _forward_torch(self, x, u, K, k, alpha):
        n, m = self.env.state_size, self.env.action_size
        xs, batched = self._batch_cols(x, n, 1)
        us, _ = self._batch_cols(u, m, 1)
        B, T = xs.shape[0], us.shape[1]
        Kt = _f32(K, self.device)
        Kt = (Kt if Kt.dim() == 4 else Kt.unsqueeze(0)).expand(B, T, m, n)
        kt, _ = self._batch_cols(k, m, 1)
        al = _f32(alpha, self.device).reshape(-1)
        st, ac, co, res = self._line_search_rollouts(xs, us, Kt, kt.expand(B, T, m), al[:1])
        st, ac, co, res = st[:, 0].unsqueeze(-1), ac[:, 0].unsqueeze(-1), co[:, 0], res[:, 0]
        J = co.sum(dim=1)
        return (st, ac, co, J, res) if batched else (st[0], ac[0], co[0], J[0], res[0])

    # A captured graph has baked in every tensor ADDRESS and every Python scalar the env's functions read (their closures),
    # and the bounds copied to the device by `_bound`.  Tensors updated IN PLACE keep their address and are seen by a replay;
    # anything REBOUND (a new goal tensor, another scalar, another `action_space`) is not.  `_graph_signature` is what the
    # captures depend on as far as it can be seen from outside: the three function objects, the cells of their closures
    # (tensors by identity and address, numbers / strings by value), the bounds by value.  `solve` compares it before every
    # solve and drops the captures when it has changed; state reached through globals or attributes of objects in the closure
    # cannot be seen -- call `invalidate_graphs()` after changing such state (or construct the solver with graphs=False).
    def _graph_signature(self):
        def cell_sig(fn):
            sig = [id(fn), id(getattr(fn, "__code__", None))]
            for cell in (getattr(fn, "__closure__", None) or ()):
                try:
                    v = cell.cell_contents
                except ValueError:                       # an empty cell
                    sig.append(None)
                    continue
                if isinstance(v, torch.Tensor):
                    sig.append((id(v), v.data_ptr(), tuple(v.shape), v.dtype))
                elif isinstance(v, (int, float, bool, str, bytes, type(None))):
                    sig.append(v)
                elif isinstance(v, np.ndarray):
                    sig.append((id(v), v.shape, v.tobytes() if v.size <= 4096 else None))
                else:
                    sig.append(id(v))
            return tuple(sig)
        env = self.env
        space = env.action_space
        return (tuple(cell_sig(getattr(env, name, None)) for name in ("_f", "_l", "_lf")), id(space),
                np.asarray(space.low, dtype=np.float32).tobytes(), np.asarray(space.high, dtype=np.float32).tobytes())

    def invalidate_graphs(self):
        """Drop the captured hipGraphs of the generic-env path (and the cached device copies of the bounds): the next solve
        captures again.  Called by ``solve`` / ``solve_device`` themselves when the env's functions, the tensors or scalars in
        their closures, or the action bounds have been REBOUND since the capture; call it yourself after changing state the
        functions reach in any other way (globals, attributes of captured objects)."""
        self._graphed = {}
        self.__dict__.pop("_bounds", None)
        self._graph_sig = None

    def _check_graphs(self):
        if not self.graphs or not self._generic_env:
            return
        sig = self._graph_signature()
        if getattr(self, "_graph_sig", None) != sig:
            self.invalidate_graphs()
            self._graph_sig = sig

    def _rollouts_graphed(self, xh, uh, K, k, alphas):
        if not self.graphs:
            return self._line_search_rollouts(xh, uh, K, k, alphas)
        g = self._graphed.setdefault("rollouts", _Graphed(self._line_search_rollouts))
        return g(xh, uh, K, k, alphas)

    def _derivatives_graphed(self, xh, uh):
        if not self.graphs:
            return self.derivatives(xh.unsqueeze(-1), uh.unsqueeze(-1))

        def flat(x, u):
            tm, cm, fm = self.derivatives(x.unsqueeze(-1), u.unsqueeze(-1))
            return tuple(tm) + tuple(cm) + tuple(fm)
        g = self._graphed.setdefault("derivatives", _Graphed(flat))
        out = g(xh, uh)
        nt, nc = len(TransitionApprox._fields), len(CostApprox._fields)
        return TransitionApprox(*out[:nt]), CostApprox(*out[nt:nt + nc]), FinalCostApprox(*out[nt + nc:])

    def _solve_host_driven(self, x0, u, batched, trace_rows=0):
        self._check_graphs()
        dev = self.device
        B, T, m = u.shape
        trace = trace_len = None
        if trace_rows > 0:                              # the same rows as tfmpc_ilqr_solve_trace_f32 writes (one per pass, TRACE_COLUMNS)
            trace = torch.full((B, trace_rows, _hip.TRACE_COLS), float("nan"), dtype=torch.float32, device=dev)
            trace_len = torch.zeros((B,), dtype=torch.int32, device=dev)
        alphas = torch.as_tensor(self._alphas(), dtype=torch.float32, device=dev)
        A = alphas.numel()
        xh, ch = self._rollout_torch(x0, u)
        uh = u.clone()
        mu = torch.zeros(B, device=dev)
        delta = torch.ones(B, device=dev)
        active = torch.ones(B, dtype=torch.bool, device=dev)
        iterations = torch.zeros(B, dtype=torch.int32, device=dev)
        status = torch.zeros(B, dtype=torch.int32, device=dev)
        attempts = torch.zeros(B, dtype=torch.int32, device=dev)
        d0, mu_min = float(self.delta_0), float(self.mu_min)
        ar = torch.arange(B, device=dev)
        for it in range(int(self.max_iterations)):
            if not bool(active.any()):
                break
            iterations[active] = it
            tm, cm, fm = self._derivatives_graphed(xh, uh)
            J_hat = ch.sum(dim=1)
            pending = active.clone()                       # still inside the `while True` of ilqr.py:238
            converged = torch.zeros_like(active)
            while bool(pending.any()):
                mu_l, delta_l = mu.clone(), delta.clone()   # _backward's local retry (ilqr.py:285-315)
                gave_up = torch.zeros_like(pending)
                level = torch.zeros(B, device=dev)          # local bumps before the pass factorised (trace column `level`)
                for retry in range(41):                     # the fused kernels give up after retry 40 too
                    K, k, _, dV1, dV2 = self.backward(T, uh.unsqueeze(-1), tm, cm, fm, mu=mu_l)
                    failed = ((self.last_status & _hip.ST_NOT_PD) != 0) & pending
                    if not bool(failed.any()):
                        break
                    status[failed] |= _hip.ST_NOT_PD
                    if retry == 40:
                        gave_up = failed                    # their K, k are undefined: never rolled out
                        break
                    delta_l = torch.where(failed, torch.clamp(delta_l * d0, min=d0), delta_l)
                    mu_l = torch.where(failed, torch.clamp(mu_l * delta_l, min=mu_min), mu_l)
                    level = level + failed.float()
                if bool(gave_up.any()):
                    status[gave_up] |= _hip.ST_MAX_ATTEMPTS
                    pending = pending & ~gave_up
                    converged |= gave_up                    # retired with their last nominal trajectory
                    K = torch.where(gave_up.reshape(-1, 1, 1, 1), torch.zeros_like(K), K)
                    k = torch.where(gave_up.reshape(-1, 1, 1, 1), torch.zeros_like(k), k)
                    if not bool(pending.any()):
                        break
                if K.dim() == 3:
                    K, k, dV1, dV2 = K.unsqueeze(0), k.unsqueeze(0), dV1.reshape(1), dV2.reshape(1)
                k = k[..., 0]
                g_norm = (k.abs() / (uh.abs() + 1.0)).amax(dim=2).mean(dim=1)          # ilqr.py:243
                conv_g = pending & (g_norm < self.atol)
                ls = pending & ~conv_g
                xs, us, cs, res = self._rollouts_graphed(xh, uh, K.contiguous(), k.contiguous(), alphas)
                J = cs.sum(dim=2)                                                        # [B, A]
                delta_J = -alphas[None] * (dV1[:, None] + alphas[None] * dV2[:, None])   # :339
                dcost = J_hat[:, None] - J
                z = torch.where(delta_J > 0, dcost / delta_J, torch.sign(dcost))         # :342-346
                ok = z >= self.c1
                accepted = ok.any(dim=1)
                chosen = torch.where(accepted, ok.float().argmax(dim=1), torch.full_like(ar, A - 1))
                small = res[ar, chosen] < self.atol                                      # :253-257
                if trace is not None:                   # mu / delta as handed to this pass, before the schedule below moves them
                    row = (iterations + attempts).long()
                    searched = ls.float()
                    vals = torch.stack([iterations.float(), mu, delta, J_hat, g_norm,
                                        torch.where(ls, chosen.float(), torch.full_like(mu, -1.0)), searched * alphas[chosen],
                                        searched * J[ar, chosen], torch.where(ls, accepted.float(), torch.full_like(mu, -1.0)),
                                        torch.where(ls, res[ar, chosen], torch.full_like(mu, -1.0)), level], dim=1)
                    put = pending & (row < trace_rows)
                    trace[ar[put], row[put]] = vals[put]
                    trace_len = torch.where(pending, (row + 1).int(), trace_len)
                take = ls & (small | accepted)
                xh = torch.where(take[:, None, None], xs[ar, chosen], xh)
                uh = torch.where(take[:, None, None], us[ar, chosen], uh)
                ch = torch.where(take[:, None], cs[ar, chosen], ch)
                acc = ls & accepted & ~small                                             # :259-266
                delta = torch.where(acc, torch.clamp(delta / d0, max=1.0 / d0), delta)
                mu = torch.where(acc, torch.where(mu * delta > mu_min, mu * delta, torch.zeros_like(mu)), mu)
                rej = ls & ~accepted & ~small                                            # :267-270
                delta = torch.where(rej, torch.clamp(delta * d0, min=d0), delta)
                mu = torch.where(rej, torch.clamp(mu * delta, min=mu_min), mu)
                attempts = attempts + rej.int()
                capped = rej & (attempts >= int(self.max_attempts))
                status[capped] |= _hip.ST_MAX_ATTEMPTS
                converged |= conv_g | (ls & small) | capped
                pending = rej & ~capped
            active = active & ~converged
        status[~torch.isfinite(ch[:, -1])] |= _hip.ST_NAN
        self.last_status = status
        out = dict(states=xh.unsqueeze(-1), actions=uh.unsqueeze(-1), costs=ch, iterations=iterations, status=status,
                   batched=batched, workspace=None)
        if trace is not None:
            out.update(trace=trace, trace_len=trace_len)
        return out
