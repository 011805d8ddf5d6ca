"""Any differentiable env at hot-path speed: ``transition`` / ``cost`` / ``final_cost`` written as DEVICE functions.

The reference differentiates whatever ``transition`` / ``cost`` it is handed (``tfmpc/envs/diffenv.py:13-101``:
``GradientTape.batch_jacobian``).  ``tfmpc.envs.torchenv.TorchEnv`` restores that for torch functions, but drives the solve
from the host (two orders of magnitude off the fused kernels).  A ``DeviceEnv`` takes the three functions as C++ device source,
templated on the scalar type::

    template <class S> __device__ void transition(const float *p, const S *x, const S *u, S *x_next);
    template <class S> __device__ S    cost(const float *p, const S *x, const S *u);
    template <class S> __device__ S    final_cost(const float *p, const S *x);

(``p``: the env's parameter floats -- ``params`` below, shared by the batch ``[P]`` or one row per instance ``[B, P]``; on ``S``:
arithmetic, comparisons, ``sqrt exp log sin cos tanh abs pow max min``.)  On first use the source is wrapped into
``csrc/user_env_kernels.hip.in`` and compiled with ``hipcc`` into a companion library (cached by source hash under
``tfmpc/_lib/userenv/``) that instantiates the SAME wave-per-instance kernels the built-in envs run -- start rollout,
linearisation, regularised backward pass with the box-QP, line search, the whole ``iLQR.solve`` loop in one launch -- on an env whose
Jacobians and Hessians come from forward-mode dual numbers evaluated one direction per lane (``csrc/user_env.h``); the
reference's ``unconnected_gradients=ZERO`` semantics.  Every ``DiffEnv`` method and ``tfmpc.solvers.ilqr.iLQR`` work unchanged.
Needs ``hipcc`` at run time (``$ROCM_PATH/bin`` or ``PATH``); without it use ``TorchEnv``."""

import ctypes
import functools
import hashlib
import os
import shutil
import subprocess
import tempfile

import numpy as np

from tfmpc import _hip
from tfmpc.envs.diffenv import Box, DiffEnv

_CSRC = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "csrc"))
_CACHE = os.path.join(os.path.dirname(_hip.lib_path()), "userenv")
_FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-shared"]      # csrc/Makefile's, one TU
_TWINS = {"tfmpc_ilqr_rollout_f32": "tfmpc_userenv_rollout_f32", "tfmpc_ilqr_derivatives_f32": "tfmpc_userenv_derivatives_f32",
          "tfmpc_ilqr_forward_f32": "tfmpc_userenv_forward_f32", "tfmpc_ilqr_solve_trace_f32": "tfmpc_userenv_solve_trace_f32",
          "tfmpc_ilqr_workspace_bytes": "tfmpc_userenv_workspace_bytes"}
_loaded = {}


def hipcc_path():
    for cand in (shutil.which("hipcc"), os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")):
        if cand and os.path.exists(cand):
            return cand
    return None


def translation_unit(source, state_size, action_size, param_count=0, zero_hessian=False):
    """``param_count``: the env's parameter floats per instance, if known (a small parameter vector is then held in registers by the 2 x 2
    lane-group kernel instead of being read through the pointer at every use, csrc/user_env.h)."""
    with open(os.path.join(_CSRC, "user_env_kernels.hip.in")) as fh:
        text = fh.read()
    return (text.replace("@STATE_SIZE@", str(int(state_size))).replace("@ACTION_SIZE@", str(int(action_size)))
            .replace("@PARAM_COUNT@", str(int(param_count))).replace("@ZERO_HESSIAN@", "1" if zero_hessian else "0")
            .replace("@SOURCE@", source))       # (the headers are found through -I: the text, hence the cache key, does not depend on where the tree lives)


def _stamp(text):
    """Hash of everything the compiled library depends on: the translation unit and the device headers it includes."""
    h = hashlib.sha256(text.encode())
    for name in sorted(os.listdir(_CSRC)):
        if name.endswith(".h"):
            with open(os.path.join(_CSRC, name), "rb") as fh:
                h.update(fh.read())
    with open(os.path.join(_CSRC, "..", "..", "include", "tfmpc_hip.h"), "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()[:16]


def _clean_env():
    """The environment hipcc runs in: the caller's minus everything a profiler or tool preload put there.  hipcc execs clang / lld; under
    rocprofv3 the preloaded tool library would initialise the GPU in the child before that exec -- the pattern this pool forbids."""
    drop = ("LD_PRELOAD", "HSA_TOOLS_LIB", "HSA_TOOLS_REPORT_LOAD_FAILURE", "ROCP_", "ROCPROFILER_", "ROCPROF_", "ROCTX_")
    return {k: v for k, v in os.environ.items() if not k.startswith(drop)}


@functools.lru_cache(maxsize=None)
def _toolchain_stamp(hipcc):
    """Flags + compiler version: a toolchain upgrade must not reuse libraries an older compiler produced."""
    h = hashlib.sha256(" ".join(_FLAGS).encode())
    try:
        out = subprocess.run([hipcc, "--version"], capture_output=True, text=True, env=_clean_env(), timeout=60).stdout
        h.update("\n".join(l for l in out.splitlines() if "version" in l.lower()).encode())     # (not InstalledDir: a path)
    except (OSError, subprocess.SubprocessError):
        h.update(b"unknown")
    return h.hexdigest()[:8]


def _cache_roots():
    """Next to the product library when that directory is writable (so prebuilt libraries travel with the tree), else a per-user cache."""
    user = os.environ.get("TFMPC_USERENV_CACHE") or os.path.join(
        os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "tfmpc", "userenv")
    return [_CACHE, user, os.path.join(tempfile.gettempdir(), f"tfmpc-userenv-{os.getuid()}")]


def build(source, state_size, action_size, param_count=0, zero_hessian=False):
    """Compile (or find in the cache) the companion library of a user env; returns its path.  Works without a GPU.
    A process that is being PROFILED should find the library prebuilt (``__graft_entry__.build`` does that for the repo's own sources):
    the compiler is started with the profiler's variables removed, but a profiled run is not the place to compile."""
    text = translation_unit(source, state_size, action_size, param_count, zero_hessian)
    stamp = _stamp(text)
    hipcc = hipcc_path()
    name = "libtfmpc_userenv.so"
    roots = _cache_roots()
    if hipcc is None:
        # nothing can be compiled here: any library built from this very source (whatever toolchain) is what there is
        for root in roots:
            if os.path.isdir(root):
                for folder in sorted(os.listdir(root)):
                    if folder.startswith(stamp) and os.path.exists(os.path.join(root, folder, name)):
                        return os.path.join(root, folder, name)
        raise RuntimeError("tfmpc.envs.deviceenv: no hipcc found ($ROCM_PATH/bin, PATH): a DeviceEnv is compiled when it is first "
                           "used; give the env as torch functions instead (tfmpc.envs.torchenv.TorchEnv)")
    key = f"{stamp}-{_toolchain_stamp(hipcc)}"
    for root in roots:
        if os.path.exists(os.path.join(root, key, name)):
            return os.path.join(root, key, name)
    folder = None
    for root in roots:                           # the first root this process may write to
        try:
            os.makedirs(os.path.join(root, key), exist_ok=True)
            probe = os.path.join(root, key, f".w{os.getpid()}")
            open(probe, "w").close()
            os.remove(probe)
            folder = os.path.join(root, key)
            break
        except OSError:
            continue
    if folder is None:
        raise RuntimeError(f"tfmpc.envs.deviceenv: no writable cache directory among {roots}")
    lib = os.path.join(folder, name)
    src = os.path.join(folder, f"env.{os.getpid()}.hip")
    with open(src, "w") as fh:
        fh.write(text)
    tmp = lib + f".{os.getpid()}.tmp"
    proc = subprocess.run([hipcc, *_FLAGS, "-I", _CSRC, src, "-o", tmp], capture_output=True, text=True, env=_clean_env())
    if proc.returncode != 0:
        errors = "\n".join(l for l in proc.stderr.splitlines() if "warning" not in l)[:4000]      # (the first error is the informative one)
        raise RuntimeError(f"tfmpc.envs.deviceenv: the env's source does not compile:\n{errors}")
    os.replace(src, os.path.join(folder, "env.hip"))
    os.replace(tmp, lib)                       # (atomic: several processes may build the same env at once; each ends with a whole file)
    return lib


class _UserLibrary:
    """The main library with its env-dependent iLQR entry points replaced by the companion library's twins."""

    def __init__(self, path):
        self._main = _hip.require_gpu()
        self._user = ctypes.CDLL(path)
        for name, twin in _TWINS.items():
            fn = getattr(self._user, twin)
            fn.restype, fn.argtypes = _hip._SIGNATURES[name]
            setattr(self, name, fn)
        self.path = path

        self._user.tfmpc_userenv_last_kernel_name.restype = ctypes.c_char_p
        self._user.tfmpc_userenv_last_kernel_name.argtypes = []
        self._user.tfmpc_userenv_set_wave_kernel.argtypes = [ctypes.c_int]
        self._user.tfmpc_userenv_set_wave_kernel.restype = None

    def tfmpc_ilqr_last_kernel_name(self):
        return self._user.tfmpc_userenv_last_kernel_name()

    def force_wave_kernel(self, on):
        """A tiny (2 x 2) user env runs the lane-group kernel; ``True`` keeps the generic wave kernel (A/B timing, tests)."""
        self._user.tfmpc_userenv_set_wave_kernel(int(bool(on)))

    def __getattr__(self, name):               # everything that does not depend on the env (backward pass, box-QP, LQR, options)
        return getattr(self._main, name)


class DeviceEnv(DiffEnv):
    kind = _hip.ENV_USER

    def __init__(self, source, state_size, action_size, params=(), low=None, high=None, zero_cost_hessian=False):
        """``zero_cost_hessian=True``: a PROMISE that every second derivative of ``cost`` and ``final_cost`` is identically zero (piecewise-linear
        costs like the reference's HVAC / Reservoir).  With bounded actions the reference's backward pass then only ever takes its bang-bang
        branch (``ilqr.py:137-141``: ``V_xx`` stays exactly 0), and the companion library is built on the COSTATE form of the
        kernels: one first-order dual evaluation of ``transition`` and ``cost`` per direction and time step instead of the second-order
        linearisation + the dense backward pass, and every step size of a line search rolled out at once, one per lane.
        ``TorchEnv.to_device_env()`` sets it from what its translator proved about the cost; a false promise gives wrong gains."""
        self.source = str(source)
        self._n, self._m = int(state_size), int(action_size)
        self.params = np.asarray(params, dtype=np.float32)
        if self.params.ndim not in (1, 2):
            raise ValueError("params: [P] floats shared by the batch, or [B, P] (one row per instance)")
        self.n_zones = int(self.params.shape[-1])                 # TfmpcEnv.n_zones carries P for this kind (include/tfmpc_hip.h)
        lo = -np.inf if low is None else low
        hi = np.inf if high is None else high
        self.action_space = Box(lo, hi, (self._m, 1))
        self.zero_cost_hessian = bool(zero_cost_hessian) and self.action_space.is_bounded()      # (unbounded: ilqr.py:143 takes the Cholesky controller)
        self._lib = None

    @property
    def state_size(self):
        return self._n

    @property
    def action_size(self):
        return self._m

    def _params(self):
        return [(self.params if self.n_zones else np.zeros((1,), dtype=np.float32), 1)]

    def _library(self):
        if self._lib is None:
            path = build(self.source, self._n, self._m, self.n_zones, self.zero_cost_hessian)
            if path not in _loaded:
                _loaded[path] = _UserLibrary(path)
            self._lib = _loaded[path]
        return self._lib

    def __repr__(self):
        return f"DeviceEnv(n={self._n}, m={self._m}, params={self.n_zones})"
