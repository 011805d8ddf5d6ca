"""ctypes binding of ``libtfmpc_hip.so`` (C ABI: ``include/tfmpc_hip.h``).

torch is imported first on purpose: torch-ROCm ships its own ``libamdhip64.so.7``
and the dynamic loader reuses that already-loaded runtime for our library, so
kernels launched here see torch's allocations and streams.
"""

import ctypes
import os

import torch

# TFMPC_LIB: load another build of the library (A/B timing of an experimental build without touching the product file)
_LIB_PATH = os.environ.get("TFMPC_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib", "libtfmpc_hip.so")
_lib = None

ERRORS = {-1: "bad argument", -2: "shape not supported by any kernel variant",
          -3: "kernel launch failed", -4: "workspace too small"}

ST_SINGULAR, ST_NOT_PD, ST_NAN, ST_QP_MAXITER, ST_MAX_ATTEMPTS, ST_QP_LATER_NOT_PD, ST_ENV_FLAG = 1, 2, 4, 8, 16, 32, 64

_P, _I, _L, _Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_size_t

_SIGNATURES = {
    "tfmpc_version": (ctypes.c_int, []),
    "tfmpc_set_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p]),
    "tfmpc_get_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]),
    "tfmpc_lqr_kernel_name": (ctypes.c_char_p, [_I, _I, _I]),
    "tfmpc_ilqr_last_kernel_name": (ctypes.c_char_p, []),
    "tfmpc_ilqr_last_group_grid": (ctypes.c_int, []),
    "tfmpc_lqr_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "tfmpc_lqr_backward_f32": (_I, [_I, _I, _I, _I, _P, _L, _P, _L, _P, _L, _P, _L,
                                    _P, _P, _P, _P, _P, _P, _P]),
    "tfmpc_lqr_forward_f32": (_I, [_I, _I, _I, _I, _P, _L, _P, _L, _P, _L, _P, _L,
                                   _P, _L, _P, _L, _P, _P, _P, _P, _P]),
    "tfmpc_lqr_solve_f32": (_I, [_I, _I, _I, _I, _P, _L, _P, _L, _P, _L, _P, _L, _P,
                                 _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "tfmpc_lqr_solve_bf16out_f32": (_I, [_I, _I, _I, _I, _P, _L, _P, _L, _P, _L, _P, _L, _P,
                                         _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "tfmpc_lqr_backward_bf16out_f32": (_I, [_I, _I, _I, _I, _P, _L, _P, _L, _P, _L, _P, _L,
                                            _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
}


for _name in ("backward", "forward", "solve"):       # twins for a non-symmetric C (tfmpc_hip.h PRECONDITION)
    _SIGNATURES[f"tfmpc_lqr_{_name}_general_f32"] = _SIGNATURES[f"tfmpc_lqr_{_name}_f32"]

ENV_LQ, ENV_NAVLQR, ENV_NAVIGATION, ENV_HVAC, ENV_RESERVOIR, ENV_USER = range(6)
ENV_MAX_PARAMS = 10
MAX_ALPHAS = 16
TRACE_COLS = 11          # TFMPC_TRACE_COLS
MIN_VERSION = 310        # tfmpc_version() this binding was written against (include/tfmpc_hip.h)


class TfmpcEnv(ctypes.Structure):
    """``struct TfmpcEnv`` of include/tfmpc_hip.h."""
    _fields_ = [("kind", ctypes.c_int32), ("n", ctypes.c_int32), ("m", ctypes.c_int32),
                ("n_zones", ctypes.c_int32), ("bounded", ctypes.c_int32),
                ("any_finite_bound", ctypes.c_int32), ("coupling_shift", ctypes.c_int32), ("reserved2", ctypes.c_int32),
                ("low", ctypes.c_void_p), ("high", ctypes.c_void_p),
                ("p", ctypes.c_void_p * ENV_MAX_PARAMS), ("stride", ctypes.c_int64 * ENV_MAX_PARAMS),
                ("scalar", ctypes.c_float * 4)]


class TfmpcIlqrConfig(ctypes.Structure):
    """``struct TfmpcIlqrConfig`` of include/tfmpc_hip.h."""
    _fields_ = [("atol", ctypes.c_float), ("max_iterations", ctypes.c_int32), ("mu_min", ctypes.c_float),
                ("delta_0", ctypes.c_float), ("c1", ctypes.c_float), ("n_alphas", ctypes.c_int32),
                ("alphas", ctypes.c_float * MAX_ALPHAS), ("max_attempts", ctypes.c_int32),
                ("storage_bf16", ctypes.c_int32)]


_SIGNATURES.update({
    "tfmpc_ilqr_rollout_f32": (_I, [_P, _I, _I, _P, _P, _P, _P, _P]),
    "tfmpc_ilqr_derivatives_f32": (_I, [_P, _I, _I, _P, _P] + [_P] * 13 + [_P]),
    "tfmpc_ilqr_backward_f32": (_I, [_I, _I, _I, _I] + [_P] * 12 + [_P, _P, _I, _P, _L] + [_P] * 6 + [_P]),
    "tfmpc_ilqr_forward_f32": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P]),
    "tfmpc_ilqr_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "tfmpc_ilqr_workspace_bytes_for": (_Z, [_P, _I, _I]),
    "tfmpc_ilqr_solve_f32": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "tfmpc_ilqr_solve_trace_f32": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _Z, _P]),
    "tfmpc_ilqr_solve_trace_qp_f32": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _Z, _P]),
    "tfmpc_boxqp_f32": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
})


def lib_path():
    return _LIB_PATH


def load():
    """Load the shared library (works without a GPU; used by the symbol tests)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"tfmpc: HIP library not built ({_LIB_PATH} missing). Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C tf-mpc_amd/csrc`. "
                "There is no CPU fallback.")
        lib = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                raise RuntimeError(f"tfmpc: {_LIB_PATH} does not export {name}: the library is older than this "
                                   "package (rebuild it: make -C tf-mpc_amd/csrc)") from None
            fn.restype = res
            fn.argtypes = args
        if lib.tfmpc_version() < MIN_VERSION:          # (layouts this package assumes -- TRACE_COLS, TfmpcEnv, status bits)
            raise RuntimeError(f"tfmpc: {_LIB_PATH} is ABI version {lib.tfmpc_version()}, this package needs >= {MIN_VERSION} "
                               "(rebuild it: make -C tf-mpc_amd/csrc)")
        _lib = lib
    return _lib


def require_gpu():
    """The product path runs on the GPU or not at all."""
    lib = load()
    if not torch.cuda.is_available():
        raise RuntimeError("tfmpc: no ROCm GPU visible; the LQR/iLQR solvers only run as HIP kernels "
                           "on gfx950 (no CPU fallback).")
    return lib


def get_option(name):
    """The current override of a kernel-variant option (``None`` = the dispatcher's own choice): what the environment
    variable of that name held at the library's first use, or the last ``set_option``."""
    buf = ctypes.create_string_buffer(64)
    if load().tfmpc_get_option(name.encode(), buf, len(buf)) != 0:
        raise ValueError(f"tfmpc: unknown option {name!r}")
    return buf.value.decode() or None


def set_option(name, value):
    """``tfmpc_set_option``: force a kernel variant (``value=None`` = the dispatcher's own choice).  The library
    reads the environment variables of the same names once, at its first use; afterwards only this call counts.
    Returns the override it replaced (``None`` if there was none), so a caller can put it back."""
    previous = get_option(name)
    rc = load().tfmpc_set_option(name.encode(), None if value is None else str(value).encode())
    if rc != 0:
        raise ValueError(f"tfmpc: unknown option {name!r}")
    return previous


class option:
    """``with _hip.option("TFMPC_LQR_MFMA", "f32"): ...`` -- variant override for the duration of a block; on exit the
    override that was active before the block (e.g. one exported in the environment) is restored."""

    def __init__(self, name, value):
        self.name, self.value, self.previous = name, value, None

    def __enter__(self):
        self.previous = set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.previous)
        return False


def default_device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"tfmpc: {what} failed: {ERRORS.get(rc, rc)}")
