"""TEST INFRASTRUCTURE ONLY -- ctypes loader for ``oracle/c/lqr_oracle.c``."""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblqr_oracle.so")
_lib = None


def build(force=False):
    """Compile the C restatement (gcc, seconds).  ``-march=native`` is used, so
    the library is rebuilt on the machine that runs it."""
    stamp = _SO + ".host"
    host = open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0] if os.path.exists("/proc/cpuinfo") else ""
    stale = force or not os.path.exists(_SO) or not os.path.exists(stamp) or open(stamp).read() != host
    if stale:
        subprocess.check_call(["make", "-s", "-B", "-C", _HERE, "_build/liblqr_oracle.so"])
        with open(stamp, "w") as fh:
            fh.write(host)
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def max_threads():
    return int(_load().lqr_oracle_max_threads())


def lqr_solve(F, f, C, c, x0, T, dtype=np.float32, nthreads=1, want_policy=False, want_value=False):
    """Batch LQR solve on the CPU.  ``F[B|1,n,d] f[B|1,n] C[B|1,d,d] c[B|1,d] x0[B,n]``.
    Returns dict with states[B,T+1,n], actions[B,T,m], costs[B,T+1] (+K,k,V,v,const)."""
    lib = _load()
    dt = np.dtype(dtype)
    fn = lib.lqr_oracle_solve_f32 if dt == np.float32 else lib.lqr_oracle_solve_f64
    x0 = np.ascontiguousarray(x0, dtype=dt)
    B, n = x0.shape
    arrs, strides = [], []
    for a, tail in ((F, 2), (f, 1), (C, 2), (c, 1)):
        a = np.ascontiguousarray(a, dtype=dt)
        if a.ndim == tail:
            a = a[None]
        assert a.shape[0] in (1, B)
        arrs.append(a)
        strides.append(0 if a.shape[0] == 1 and B > 1 else int(np.prod(a.shape[1:])))
    F, f, C, c = arrs
    d = F.shape[-1]
    m = d - n
    out = dict(states=np.empty((B, T + 1, n), dt), actions=np.empty((B, T, m), dt),
               costs=np.empty((B, T + 1), dt))
    if want_policy:
        out.update(K=np.empty((B, T, m, n), dt), k=np.empty((B, T, m), dt))
    if want_value:
        out.update(V=np.empty((B, T, n, n), dt), v=np.empty((B, T, n), dt), const=np.empty((B, T), dt))
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a is not None else None
    L = ctypes.c_long
    rc = fn(ctypes.c_int(B), ctypes.c_int(n), ctypes.c_int(m), ctypes.c_int(T),
            p(F), L(strides[0]), p(f), L(strides[1]), p(C), L(strides[2]), p(c), L(strides[3]),
            p(x0), p(out["states"]), p(out["actions"]), p(out["costs"]),
            p(out.get("K")), p(out.get("k")), p(out.get("V")), p(out.get("v")), p(out.get("const")),
            ctypes.c_int(nthreads))
    out["status"] = int(rc)
    return out
