"""Python env -> ``DeviceEnv`` source, automatically.

The reference differentiates whatever Python ``transition`` / ``cost`` it is handed (``tfmpc/envs/diffenv.py:13-101``); a reference user
has Python methods, not C++ templates.  ``TorchEnv`` (``envs/torchenv.py``) takes such functions as torch code but drives the solve from the
host.  This module TRANSLATES the three torch functions of one instance (``x[n], u[m] -> x'[n]``, ``-> scalar``, ``x[n] -> scalar``) into the
three device templates a ``DeviceEnv`` is made of (``envs/deviceenv.py``), so that the user's env runs inside the same fused kernels as the
built-in ones, its derivatives taken by the dual numbers of ``csrc/user_env.h``:

* the function is traced once to ATen operations with ``torch.fx.experimental.proxy_tensor.make_fx`` on example inputs of the env's shapes
  (shapes are concrete: Python loops over ``range(n)``, ``len``, ``.shape`` all work);
* a second trace on FAKE tensors rejects data-dependent Python control flow (``if x[0] > 0:`` would otherwise be baked in silently) with an
  error that says so -- write it as ``torch.where``;
* the graph is unrolled into straight-line scalar code: every tensor becomes an array of C++ scalar names, every operation one statement per
  element.  Tensor constants the function closes over go into the env's ``params`` vector (``p[k]`` in the source, so the compiled library
  does not depend on their values -- except where a constant is exactly 0: that is taken as structure, e.g. of an adjacency matrix, and
  baked in, so the terms it removes cost nothing); Python numbers become literals.

Supported: elementwise arithmetic and comparisons, ``where`` / ``clamp`` / ``abs`` / ``maximum`` / ``minimum`` / ``relu`` / ``sign``, ``sqrt
exp log log1p expm1 sin cos tanh sigmoid silu softplus elu leaky_relu pow reciprocal rsqrt square neg addcmul lerp``, ``sum mean var std prod amax amin cumsum``, ``linalg.norm`` (1, 2, inf), ``dot mv mm
matmul addmm``, indexing / slicing / ``cat`` / ``stack`` / ``reshape`` / ``transpose`` / ``expand`` / ``unbind`` / ``split``, constant
creation (``zeros ones full arange tensor``), ``detach``, dtype casts, and the in-place forms of the above on views (``out[i] = ...``).
Subgradient conventions are TensorFlow's, as in the built-in envs: ``|y|' = sign(y)`` (0 at 0), a tie of
``max(a, b)`` / ``min(a, b)`` goes to the FIRST argument, ``relu'(0) = 0``, ``clamp`` passes the gradient on the closed interval.
Anything else raises ``UnsupportedOperation`` naming the operation and pointing at the ``TorchEnv`` fallback.
"""

import math

import numpy as np
import torch


class UnsupportedOperation(NotImplementedError):
    pass


class _E:
    """One scalar of the unrolled program: a C++ expression (a variable name, ``p[k]``, ``x[i]`` or a literal) and its kind --
    'S' (the templated scalar: depends on the inputs), 'f' (float: constants and parameters), 'b' (bool), 'i' (int literal)."""
    __slots__ = ("code", "kind", "val", "lin")

    def __init__(self, code, kind, val=None, lin=True):
        # lin (kind 'S' only): the value is PIECEWISE AFFINE in (x, u) -- sums, products with constants, abs / max / min / where of such --
        # so every second derivative the dual numbers would compute for it is exactly zero (what the translator reports as
        # `cost_is_piecewise_linear`: the reference's backward pass then only ever takes its bang-bang branch, ilqr.py:137-141)
        self.code, self.kind, self.val, self.lin = code, kind, val, lin

    def __repr__(self):
        return f"<{self.kind}:{self.code}>"


def _lit(v):
    v = float(v)
    if math.isnan(v):
        return _E("NAN", "f", v)
    if math.isinf(v):
        return _E("INFINITY" if v > 0 else "(-INFINITY)", "f", v)
    v32 = float(np.float32(v))
    s = repr(v32)
    if "e" in s or "E" in s:
        m, e = s.lower().split("e")
        if "." not in m:
            m += ".0"
        s = f"{m}e{int(e)}"
    elif "." not in s:
        s += ".0"
    code = s + "f"
    return _E(f"({code})" if v32 < 0 else code, "f", v32)


def _is_param(e):
    """An element that is read from the parameter vector (its numeric value known at translation time), as opposed to a literal."""
    return e.code.startswith("p[") or e.code.startswith("(-p[")


class ParamTable:
    """The env's parameter floats: every tensor constant of the three functions, each stored once."""

    def __init__(self):
        self.values = []
        self._seen = {}

    def add(self, array):
        a = np.ascontiguousarray(np.asarray(array, dtype=np.float32))
        key = (a.shape, a.tobytes())
        if key not in self._seen:
            self._seen[key] = len(self.values)
            self.values.extend(a.reshape(-1).tolist())
        base = self._seen[key]
        out = np.empty(a.size, dtype=object)
        flat = a.reshape(-1)
        for k in range(a.size):
            # an exact zero of a constant is STRUCTURE (an adjacency matrix, a mask): it is baked in as the literal, so that the products it
            # kills cost no statement and no derivative code; every other entry is read from the parameter vector (`val`: its value, so that
            # arithmetic BETWEEN constants is done here, once, and lands in a derived parameter slot -- see `derived`)
            out[k] = _lit(0.0) if flat[k] == 0.0 else _E(f"p[{base + k}]", "f", float(flat[k]))
        return out.reshape(a.shape)

    def derived(self, value):
        """A constant computed from other constants at translation time (adj / R_wall, dt / capacity, (lo + hi) / 2 ...): one more parameter slot
        instead of a float division in every evaluation of the env on the device."""
        v = float(np.float32(value))
        if v == 0.0 or not math.isfinite(v):
            return _lit(v)
        self.values.append(v)
        return _E(f"p[{len(self.values) - 1}]", "f", v)

    def array(self):
        return np.asarray(self.values, dtype=np.float32)


class _Program:
    def __init__(self):
        self.lines = []
        self.count = 0

    def new(self, kind, expr, lin=True):
        name = f"v{self.count}"
        self.count += 1
        ctype = {"S": "S", "f": "float", "b": "bool"}[kind]
        self.lines.append(f"    const {ctype} {name} = {expr};")
        return _E(name, kind, lin=lin if kind == "S" else True)


def _arr(elems, shape=None):
    a = np.empty(len(elems), dtype=object)
    for k, e in enumerate(elems):
        a[k] = e
    return a.reshape(shape if shape is not None else (len(elems),))


def _full(shape, e):
    a = np.empty(int(np.prod(shape)) if len(shape) else 1, dtype=object)
    for k in range(a.size):
        a[k] = e
    return a.reshape(tuple(shape))


class _Translator:
    """Interprets an ATen-level fx graph over arrays of ``_E``."""

    def __init__(self, gm, params, arg_names):
        self.gm, self.params, self.arg_names = gm, params, arg_names
        self.prog = _Program()
        self.env = {}

    # ---- scalar building blocks ---------------------------------------------------------------------------------------------------
    def _as_e(self, v):
        if isinstance(v, _E):
            return v
        if isinstance(v, bool):
            return _E("true" if v else "false", "b", v)
        if isinstance(v, (int, float, np.integer, np.floating)):
            return _lit(v)
        raise UnsupportedOperation(f"cannot use {type(v).__name__} as a scalar")

    def _num(self, e):
        """A value usable in arithmetic: bools and ints become floats."""
        if e.kind == "b":
            if e.val is not None:
                return _lit(1.0 if e.val else 0.0)
            return self.prog.new("f", f"({e.code} ? 1.0f : 0.0f)")
        if e.kind == "i":
            return _lit(e.val)
        return e

    def _binary(self, a, b, op, fold):
        a, b = self._num(self._as_e(a)), self._num(self._as_e(b))
        if a.val is not None and b.val is not None:
            try:
                value = fold(a.val, b.val)
                return _lit(value) if not (_is_param(a) or _is_param(b)) else self.params.derived(value)
            except (ZeroDivisionError, OverflowError, ValueError):
                pass
        # identities that keep the program (and the derivative code) small
        if op == "*":
            if (a.val == 1.0):
                return b
            if (b.val == 1.0):
                return a
            if a.val == 0.0 or b.val == 0.0:                # (a masked entry: no statement, no derivative code)
                return _lit(0.0)
        if op == "+":
            if a.val == 0.0:
                return b
            if b.val == 0.0:
                return a
        if op == "-" and b.val == 0.0:
            return a
        if op == "/" and b.val == 1.0:
            return a
        kind = "S" if "S" in (a.kind, b.kind) else "f"
        if op in "+-":
            lin = a.lin and b.lin
        elif op == "*":
            lin = (a.lin and b.lin) and not (a.kind == "S" and b.kind == "S")       # a constant times a piecewise-affine value
        else:
            lin = a.lin and b.kind != "S"                                            # ... divided by a constant
        return self.prog.new(kind, f"{a.code} {op} {b.code}", lin)

    def add(self, a, b): return self._binary(a, b, "+", lambda x, y: x + y)
    def sub(self, a, b): return self._binary(a, b, "-", lambda x, y: x - y)
    def mul(self, a, b): return self._binary(a, b, "*", lambda x, y: x * y)
    def div(self, a, b): return self._binary(a, b, "/", lambda x, y: x / y)

    def _unary(self, a, fn, fold=None):
        a = self._num(self._as_e(a))
        if a.val is not None and fold is not None:
            try:
                return _lit(fold(a.val)) if not _is_param(a) else self.params.derived(fold(a.val))
            except (ValueError, OverflowError, ZeroDivisionError):
                pass
        lin = a.lin and (fn == "abs" or a.kind != "S")     # |y| keeps a value piecewise affine; sqrt, exp, ... do not
        if a.kind == "f":                                   # plain float: the C names (no overload resolution between float and dual forms)
            fn = {"sqrt": "sqrtf", "exp": "expf", "log": "logf", "sin": "sinf", "cos": "cosf", "tanh": "tanhf", "abs": "fabsf", "tan": "tanf", "atan": "atanf",
                  "asin": "asinf", "acos": "acosf", "sinh": "sinhf", "cosh": "coshf", "erf": "erff"}.get(fn, fn)
        return self.prog.new(a.kind, f"{fn}({a.code})", lin)

    def atan2(self, y, x):
        y, x = self._num(self._as_e(y)), self._num(self._as_e(x))
        if y.val is not None and x.val is not None and not (_is_param(y) or _is_param(x)):
            return _lit(math.atan2(y.val, x.val))
        kind = "S" if "S" in (y.kind, x.kind) else "f"
        return self.prog.new(kind, f"{'atan2' if kind == 'S' else 'atan2f'}({y.code}, {x.code})", False)

    def neg(self, a):
        a = self._num(self._as_e(a))
        if a.val is not None:                               # (a negated parameter is the parameter with a sign: no slot, no statement -- the ISA negates operands for free)
            return _lit(-a.val) if not _is_param(a) else _E(a.code[2:-1] if a.code.startswith("(-") else f"(-{a.code})", "f", -a.val)
        return self.prog.new(a.kind, f"-{a.code}", a.lin)

    def fmax(self, a, b, fn="max"):
        a, b = self._num(self._as_e(a)), self._num(self._as_e(b))
        if a.val is not None and b.val is not None:
            value = max(a.val, b.val) if fn == "max" else min(a.val, b.val)
            return _lit(value) if not (_is_param(a) or _is_param(b)) else self.params.derived(value)
        if a.kind == "f" and b.kind == "f":
            return self.prog.new("f", f"{'fmaxf' if fn == 'max' else 'fminf'}({a.code}, {b.code})")
        return self.prog.new("S", f"{fn}({a.code}, {b.code})", a.lin and b.lin)        # a tie goes to the first argument (user_env.h)

    def pow(self, a, b):
        a, b = self._num(self._as_e(a)), self._num(self._as_e(b))
        if b.val is not None:
            p = b.val
            if a.val is not None:
                try:
                    return _lit(a.val ** p) if not (_is_param(a) or _is_param(b)) else self.params.derived(a.val ** p)
                except (ValueError, OverflowError, ZeroDivisionError):
                    pass
            if p == 1.0:
                return a
            if p == 2.0:
                return self.mul(a, a)
            if p == 3.0:
                return self.mul(self.mul(a, a), a)
            if p == 0.5:
                return self._unary(a, "sqrt")
            if p == -1.0:
                return self.div(_lit(1.0), a)
            if p == 0.0:
                return _lit(1.0)
        if b.kind != "f":
            raise UnsupportedOperation("pow with an exponent that depends on the state or action (write it as exp(e * log(b)))")
        if a.kind == "f":
            return self.prog.new("f", f"powf({a.code}, {b.code})")
        return self.prog.new("S", f"pow({a.code}, {b.code})", False)

    def compare(self, a, b, op):
        a, b = self._num(self._as_e(a)), self._num(self._as_e(b))
        if a.val is not None and b.val is not None:
            r = {"<": a.val < b.val, ">": a.val > b.val, "<=": a.val <= b.val, ">=": a.val >= b.val, "==": a.val == b.val, "!=": a.val != b.val}[op]
            return _E("true" if r else "false", "b", r)
        return self.prog.new("b", f"({a.code} {op} {b.code})")

    def logical(self, a, b, op):
        a, b = self._as_e(a), self._as_e(b)
        if a.kind != "b" or b.kind != "b":
            raise UnsupportedOperation("bitwise / logical operation on non-boolean tensors")
        return self.prog.new("b", f"({a.code} {op} {b.code})")

    def select(self, c, a, b):
        c = self._as_e(c)
        a, b = self._num(self._as_e(a)), self._num(self._as_e(b))
        if c.kind != "b":
            c = self.compare(c, _lit(0.0), "!=")
        if c.val is not None:
            return a if c.val else b
        kind = "S" if "S" in (a.kind, b.kind) else "f"
        wrap = lambda e: f"S({e.code})" if (kind == "S" and e.kind != "S") else e.code
        return self.prog.new(kind, f"({c.code} ? {wrap(a)} : {wrap(b)})", a.lin and b.lin)

    def stop_gradient(self, a):
        a = self._num(self._as_e(a))
        if a.kind != "S":
            return a
        return self.prog.new("f", f"tfmpc::ad::prim({a.code})")

    # ---- tensor helpers ------------------------------------------------------------------------------------------------------------
    def _t(self, v):
        """Any graph value as an object array of _E."""
        if isinstance(v, np.ndarray) and v.dtype == object:
            return v
        if isinstance(v, _E):
            return _full((), v)
        if isinstance(v, np.ndarray):                     # integer / bool constants
            out = np.empty(v.size, dtype=object)
            flat = v.reshape(-1)
            for k in range(v.size):
                out[k] = _E(str(bool(flat[k])).lower(), "b", bool(flat[k])) if v.dtype == np.bool_ else _E(str(int(flat[k])), "i", int(flat[k]))
            return out.reshape(v.shape)
        if isinstance(v, (int, float, bool, np.integer, np.floating)):
            return _full((), self._as_e(v))
        raise UnsupportedOperation(f"cannot use {type(v).__name__} as a tensor")

    def _map2(self, a, b, fn):
        a, b = self._t(a), self._t(b)
        a, b = np.broadcast_arrays(a, b)
        out = np.empty(a.shape, dtype=object)
        for idx in np.ndindex(*a.shape):
            out[idx] = fn(a[idx], b[idx])
        return out

    def _map1(self, a, fn):
        a = self._t(a)
        out = np.empty(a.shape, dtype=object)
        for idx in np.ndindex(*a.shape):
            out[idx] = fn(a[idx])
        return out

    def _map3(self, a, b, c, fn):
        a, b, c = np.broadcast_arrays(self._t(a), self._t(b), self._t(c))
        out = np.empty(a.shape, dtype=object)
        for idx in np.ndindex(*a.shape):
            out[idx] = fn(a[idx], b[idx], c[idx])
        return out

    def _reduce(self, a, dims, keepdim, fn, init=None):
        a = self._t(a)
        if dims is None or (isinstance(dims, (list, tuple)) and len(dims) == 0):
            dims = list(range(a.ndim))
        if isinstance(dims, int):
            dims = [dims]
        dims = sorted(d % max(a.ndim, 1) for d in dims) if a.ndim else []
        moved = np.moveaxis(a, dims, list(range(len(dims)))) if dims else a
        rest = moved.shape[len(dims):]
        flat = moved.reshape((-1,) + rest) if dims else moved.reshape((1,) + rest)
        out = np.empty(rest, dtype=object)
        for idx in np.ndindex(*rest):
            acc = init
            for k in range(flat.shape[0]):
                e = flat[(k,) + idx]
                acc = e if acc is None else fn(acc, e)
            out[idx] = acc if acc is not None else _lit(0.0)
        if keepdim:
            for d in dims:
                out = np.expand_dims(out, d)
        return out

    def _matmul(self, a, b):
        a, b = self._t(a), self._t(b)
        va, vb = a.ndim == 1, b.ndim == 1
        if va:
            a = a[None, :]
        if vb:
            b = b[:, None]
        if a.ndim != 2 or b.ndim != 2 or a.shape[1] != b.shape[0]:
            raise UnsupportedOperation(f"matmul of shapes {a.shape} x {b.shape} (batched matmul is not translated)")
        out = np.empty((a.shape[0], b.shape[1]), dtype=object)
        for i in range(a.shape[0]):
            for j in range(b.shape[1]):
                acc = None
                for k in range(a.shape[1]):
                    # a structural zero of a constant operand drops out (adjacency matrices): no statement, no derivative code
                    if (a[i, k].val == 0.0) or (b[k, j].val == 0.0):
                        continue
                    prod = self.mul(a[i, k], b[k, j])
                    acc = prod if acc is None else self.add(acc, prod)
                out[i, j] = acc if acc is not None else _lit(0.0)
        if va:
            out = out[0]
        if vb:
            out = out[..., 0]
        return out

    # ---- the interpreter --------------------------------------------------------------------------------------------------------------
    def run(self, inputs):
        aten = torch.ops.aten
        g = self.gm.graph
        ph = iter(inputs)
        result = None
        for node in g.nodes:
            if node.op == "placeholder":
                self.env[node] = next(ph)
            elif node.op == "get_attr":
                t = getattr(self.gm, node.target)
                if not isinstance(t, torch.Tensor):
                    raise UnsupportedOperation(f"constant of type {type(t).__name__}")
                t = t.detach().cpu()
                if t.dtype.is_floating_point:
                    self.env[node] = self.params.add(t.numpy().astype(np.float32)) if t.numel() else np.empty(tuple(t.shape), dtype=object)
                else:
                    self.env[node] = t.numpy().copy()
            elif node.op == "call_function":
                args = torch.fx.node.map_arg(node.args, lambda n: self.env[n])
                kwargs = torch.fx.node.map_arg(node.kwargs, lambda n: self.env[n])
                self.env[node] = self.call(node.target, args, kwargs)
            elif node.op == "output":
                result = torch.fx.node.map_arg(node.args[0], lambda n: self.env[n])
            else:
                raise UnsupportedOperation(f"graph node of kind {node.op}")
        return result

    def call(self, target, args, kw):
        import operator
        aten = torch.ops.aten
        name = str(target)
        if target is operator.getitem:
            return args[0][args[1]]
        packet = getattr(target, "overloadpacket", None)
        op = packet.__name__ if packet is not None else getattr(target, "__name__", name)
        handler = getattr(self, "op_" + op, None)
        if handler is None:
            raise UnsupportedOperation(
                f"torch operation `{name}` is not translated to device code by tfmpc.envs.fxsource (see its docstring for what is); "
                "run this env through tfmpc.envs.torchenv.TorchEnv directly (host-driven solve), or rewrite the step with supported operations")
        return handler(*args, **kw)

    # elementwise --------------------------------------------------------------------------------------------------------------------
    def op_add(self, a, b, alpha=1):
        if alpha != 1:
            b = self._map2(b, alpha, self.mul)
        return self._map2(a, b, self.add)

    def op_sub(self, a, b, alpha=1):
        if alpha != 1:
            b = self._map2(b, alpha, self.mul)
        return self._map2(a, b, self.sub)

    def op_rsub(self, a, b, alpha=1):
        if alpha != 1:
            a = self._map2(a, alpha, self.mul)
        return self._map2(b, a, self.sub)

    def op_mul(self, a, b): return self._map2(a, b, self.mul)
    def op_div(self, a, b, rounding_mode=None):
        if rounding_mode is not None:
            raise UnsupportedOperation("div with a rounding mode")
        return self._map2(a, b, self.div)
    op_true_divide = op_div
    def op_neg(self, a): return self._map1(a, self.neg)
    def op_pow(self, a, b): return self._map2(a, b, self.pow)
    def op_square(self, a): return self._map1(a, lambda e: self.mul(e, e))
    def op_sqrt(self, a): return self._map1(a, lambda e: self._unary(e, "sqrt", math.sqrt))
    def op_rsqrt(self, a): return self._map1(a, lambda e: self.div(_lit(1.0), self._unary(e, "sqrt", math.sqrt)))
    def op_exp(self, a): return self._map1(a, lambda e: self._unary(e, "exp", math.exp))
    def op_log(self, a): return self._map1(a, lambda e: self._unary(e, "log", math.log))
    def op_sin(self, a): return self._map1(a, lambda e: self._unary(e, "sin", math.sin))
    def op_cos(self, a): return self._map1(a, lambda e: self._unary(e, "cos", math.cos))
    def op_tanh(self, a): return self._map1(a, lambda e: self._unary(e, "tanh", math.tanh))
    def op_abs(self, a): return self._map1(a, lambda e: self._unary(e, "abs", abs))
    # (round 6: vehicle / arm models -- csrc/user_env.h holds their dual-number forms)
    def op_tan(self, a): return self._map1(a, lambda e: self._unary(e, "tan", math.tan))
    def op_atan(self, a): return self._map1(a, lambda e: self._unary(e, "atan", math.atan))
    def op_asin(self, a): return self._map1(a, lambda e: self._unary(e, "asin", math.asin))
    def op_acos(self, a): return self._map1(a, lambda e: self._unary(e, "acos", math.acos))
    def op_sinh(self, a): return self._map1(a, lambda e: self._unary(e, "sinh", math.sinh))
    def op_cosh(self, a): return self._map1(a, lambda e: self._unary(e, "cosh", math.cosh))
    def op_erf(self, a): return self._map1(a, lambda e: self._unary(e, "erf", math.erf))
    def op_atan2(self, a, b): return self._map2(a, b, self.atan2)
    op_arctan2 = op_atan2
    op_arctan, op_arcsin, op_arccos = op_atan, op_asin, op_acos
    def op_reciprocal(self, a): return self._map1(a, lambda e: self.div(_lit(1.0), e))
    def op_sigmoid(self, a):
        return self._map1(a, lambda e: self.div(_lit(1.0), self.add(_lit(1.0), self._unary(self.neg(e), "exp", math.exp))))
    def op_relu(self, a): return self._map1(a, lambda e: self.fmax(_lit(0.0), e))          # relu'(0) = 0: the tie goes to the constant
    def op_log1p(self, a): return self._map1(a, lambda e: self._unary(self.add(_lit(1.0), e), "log", math.log))
    def op_expm1(self, a): return self._map1(a, lambda e: self.sub(self._unary(e, "exp", math.exp), _lit(1.0)))
    def op_silu(self, a):
        return self._map1(a, lambda e: self.div(e, self.add(_lit(1.0), self._unary(self.neg(e), "exp", math.exp))))
    def op_leaky_relu(self, a, negative_slope=0.01):
        return self._map1(a, lambda e: self.select(self.compare(e, _lit(0.0), ">"), e, self.mul(e, _lit(negative_slope))))
    def op_elu(self, a, alpha=1.0, scale=1.0, input_scale=1.0):
        def one(e):
            neg = self.mul(_lit(alpha), self.sub(self._unary(self.mul(e, _lit(input_scale)), "exp", math.exp), _lit(1.0)))
            return self.mul(_lit(scale), self.select(self.compare(e, _lit(0.0), ">"), e, neg))
        return self._map1(a, one)
    def op_softplus(self, a, beta=1.0, threshold=20.0):
        def one(e):
            bx = self.mul(e, _lit(beta))
            soft = self.div(self._unary(self.add(_lit(1.0), self._unary(bx, "exp", math.exp)), "log", math.log), _lit(beta))
            return self.select(self.compare(bx, _lit(threshold), ">"), e, soft)          # torch's own switch to the identity above the threshold
        return self._map1(a, one)
    def op_log_sigmoid(self, a): return self._map1(self.op_softplus(self._map1(a, self.neg)), self.neg)
    def op_log_sigmoid_forward(self, a):           # (the ATen form returns (output, buffer))
        out = self.op_log_sigmoid(a)
        return out, out
    def op_addcmul(self, a, t1, t2, value=1): return self.op_add(a, self._map2(self.op_mul(t1, t2), value, self.mul))
    def op_addcdiv(self, a, t1, t2, value=1): return self.op_add(a, self._map2(self.op_div(t1, t2), value, self.mul))
    def op_lerp(self, a, b, w): return self.op_add(a, self.op_mul(w, self.op_sub(b, a)))
    def op_frobenius_norm(self, a, dim=None, keepdim=False): return self.op_linalg_vector_norm(a, 2, dim, keepdim)
    def op_dist(self, a, b, p=2): return self.op_linalg_vector_norm(self.op_sub(a, b), p)
    def op_var(self, a, dim=None, correction=1, keepdim=False, unbiased=None):
        a = self._t(a)
        if unbiased is not None:
            correction = 1 if unbiased else 0
        mean = self.op_mean(a, dim, True)
        sq = self._map1(self.op_sub(a, mean), lambda e: self.mul(e, e))
        total = self._reduce(sq, dim, keepdim, self.add)
        count = a.size // max(total.size, 1)
        return self._map1(total, lambda e: self.mul(e, _lit(1.0 / max(count - correction, 1))))
    def op_std(self, a, dim=None, correction=1, keepdim=False, unbiased=None):
        return self._map1(self.op_var(a, dim, correction, keepdim, unbiased), lambda e: self._unary(e, "sqrt", math.sqrt))
    def op_maximum(self, a, b): return self._map2(a, b, lambda x, y: self.fmax(x, y, "max"))
    def op_minimum(self, a, b): return self._map2(a, b, lambda x, y: self.fmax(x, y, "min"))
    op_fmax, op_fmin = op_maximum, op_minimum
    def op_sign(self, a):
        return self._map1(a, lambda e: self.select(self.compare(e, _lit(0.0), ">"), _lit(1.0), self.select(self.compare(e, _lit(0.0), "<"), _lit(-1.0), _lit(0.0))))
    op_sgn = op_sign

    def op_clamp(self, a, min=None, max=None):
        out = self._t(a)
        if min is not None:
            out = self._map2(out, min, lambda x, lo: self.fmax(x, lo, "max"))       # x >= lo keeps x (and its gradient)
        if max is not None:
            out = self._map2(out, max, lambda x, hi: self.fmax(x, hi, "min"))
        return out
    op_clip = op_clamp
    def op_clamp_min(self, a, min): return self.op_clamp(a, min=min)
    def op_clamp_max(self, a, max): return self.op_clamp(a, max=max)
    def op_hardtanh(self, a, min_val=-1.0, max_val=1.0): return self.op_clamp(a, min_val, max_val)

    def op_gt(self, a, b): return self._map2(a, b, lambda x, y: self.compare(x, y, ">"))
    def op_lt(self, a, b): return self._map2(a, b, lambda x, y: self.compare(x, y, "<"))
    def op_ge(self, a, b): return self._map2(a, b, lambda x, y: self.compare(x, y, ">="))
    def op_le(self, a, b): return self._map2(a, b, lambda x, y: self.compare(x, y, "<="))
    def op_eq(self, a, b): return self._map2(a, b, lambda x, y: self.compare(x, y, "=="))
    def op_ne(self, a, b): return self._map2(a, b, lambda x, y: self.compare(x, y, "!="))
    def op_logical_and(self, a, b): return self._map2(a, b, lambda x, y: self.logical(x, y, "&&"))
    def op_logical_or(self, a, b): return self._map2(a, b, lambda x, y: self.logical(x, y, "||"))
    op_bitwise_and, op_bitwise_or = op_logical_and, op_logical_or
    def op_logical_not(self, a): return self._map1(a, lambda e: self.prog.new("b", f"(!{self._as_e(e).code})"))
    op_bitwise_not = op_logical_not
    def op_where(self, c, a, b): return self._map3(c, a, b, self.select)

    # reductions / linear algebra --------------------------------------------------------------------------------------------------------
    def op_sum(self, a, dim=None, keepdim=False, dtype=None): return self._reduce(a, dim, keepdim, self.add)
    def op_prod(self, a, dim=None, keepdim=False, dtype=None): return self._reduce(a, dim, keepdim, self.mul)
    def op_mean(self, a, dim=None, keepdim=False, dtype=None):
        a = self._t(a)
        total = self._reduce(a, dim, keepdim, self.add)
        count = a.size // max(total.size, 1)
        return self._map1(total, lambda e: self.mul(e, _lit(1.0 / count)))
    def op_amax(self, a, dim=None, keepdim=False): return self._reduce(a, dim, keepdim, lambda x, y: self.fmax(x, y, "max"))
    def op_amin(self, a, dim=None, keepdim=False): return self._reduce(a, dim, keepdim, lambda x, y: self.fmax(x, y, "min"))

    def op_max(self, a, *rest):
        if rest:
            raise UnsupportedOperation("max / min along a dimension with indices (use amax / amin)")
        return self._reduce(a, None, False, lambda x, y: self.fmax(x, y, "max"))

    def op_min(self, a, *rest):
        if rest:
            raise UnsupportedOperation("max / min along a dimension with indices (use amax / amin)")
        return self._reduce(a, None, False, lambda x, y: self.fmax(x, y, "min"))

    def op_linalg_vector_norm(self, a, ord=2, dim=None, keepdim=False, dtype=None):
        if ord in (2, 2.0, None):
            sq = self._map1(a, lambda e: self.mul(e, e))
            return self._map1(self._reduce(sq, dim, keepdim, self.add), lambda e: self._unary(e, "sqrt", math.sqrt))
        if ord in (1, 1.0):
            return self._reduce(self.op_abs(a), dim, keepdim, self.add)
        if ord == float("inf"):
            return self._reduce(self.op_abs(a), dim, keepdim, lambda x, y: self.fmax(x, y, "max"))
        raise UnsupportedOperation(f"vector norm of order {ord}")

    def op_norm(self, a, p=2, dim=None, keepdim=False, dtype=None): return self.op_linalg_vector_norm(a, p, dim, keepdim)
    def op_dot(self, a, b): return self._matmul(a, b)
    op_mv = op_mm = op_matmul = op_dot
    def op_addmm(self, c, a, b, beta=1, alpha=1):
        prod = self._matmul(a, b)
        if alpha != 1:
            prod = self._map2(prod, alpha, self.mul)
        if beta != 1:
            c = self._map2(c, beta, self.mul)
        return self._map2(c, prod, self.add)
    def op_addmv(self, c, a, b, beta=1, alpha=1): return self.op_addmm(c, a, b, beta, alpha)

    def op_cumsum(self, a, dim, dtype=None):
        a = np.moveaxis(self._t(a), dim, 0)
        out = np.empty(a.shape, dtype=object)
        for idx in np.ndindex(*a.shape[1:]):
            acc = None
            for k in range(a.shape[0]):
                acc = a[(k,) + idx] if acc is None else self.add(acc, a[(k,) + idx])
                out[(k,) + idx] = acc
        return np.moveaxis(out, 0, dim)

    # shapes: numpy views of the object arrays, so that in-place writes through a view reach the base like they do in torch ---------
    def op_view(self, a, shape): return self._t(a).reshape(tuple(shape))
    op_reshape = op__unsafe_view = op_view
    def op_flatten(self, a, start_dim=0, end_dim=-1):
        a = self._t(a)
        end = end_dim % a.ndim if a.ndim else 0
        return a.reshape(a.shape[:start_dim] + (-1,) + a.shape[end + 1:]) if a.ndim else a.reshape(1)
    def op_squeeze(self, a, dim=None):
        a = self._t(a)
        if dim is None:
            return a.reshape(tuple(s for s in a.shape if s != 1))
        dims = [dim] if isinstance(dim, int) else list(dim)
        dims = [d % a.ndim for d in dims if a.ndim and a.shape[d % a.ndim] == 1]
        return a.reshape(tuple(s for k, s in enumerate(a.shape) if k not in dims))
    def op_unsqueeze(self, a, dim):
        a = self._t(a)
        return np.expand_dims(a, dim if dim >= 0 else dim + a.ndim + 1)
    def op_expand(self, a, shape, implicit=False):
        a = self._t(a)
        shape = list(shape)
        lead = len(shape) - a.ndim
        full = [a.shape[k - lead] if (s == -1) else s for k, s in enumerate(shape)]
        return np.broadcast_to(a, tuple(full))
    def op_permute(self, a, dims): return np.transpose(self._t(a), tuple(dims))
    def op_transpose(self, a, d0, d1): return np.swapaxes(self._t(a), d0, d1)
    def op_t(self, a):
        a = self._t(a)
        return a.T if a.ndim == 2 else a
    def op_select(self, a, dim, index):
        a = self._t(a)
        idx = [slice(None)] * a.ndim
        idx[dim] = int(index)
        return a[tuple(idx) + (Ellipsis,)]
    def op_slice(self, a, dim=0, start=None, end=None, step=1):
        a = self._t(a)
        idx = [slice(None)] * a.ndim
        if end is not None and end > (1 << 60):
            end = None
        idx[dim] = slice(start, end, step)
        return a[tuple(idx)]
    def op_narrow(self, a, dim, start, length): return self.op_slice(a, dim, start, start + length)
    def op_cat(self, tensors, dim=0):
        parts = [self._t(t) for t in tensors if not (isinstance(t, np.ndarray) and t.ndim == 1 and t.size == 0)]
        return np.concatenate(parts, axis=dim) if parts else np.empty((0,), dtype=object)
    op_concatenate = op_cat
    def op_stack(self, tensors, dim=0): return np.stack([self._t(t) for t in tensors], axis=dim)
    def op_unbind(self, a, dim=0):
        a = self._t(a)
        return [np.take(a, k, axis=dim) for k in range(a.shape[dim])]
    def op_split(self, a, size, dim=0):
        a = self._t(a)
        n = a.shape[dim]
        sizes = [size] * (n // size) + ([n % size] if n % size else []) if isinstance(size, int) else list(size)
        out, at = [], 0
        for s in sizes:
            out.append(self.op_slice(a, dim, at, at + s))
            at += s
        return out
    op_split_with_sizes = op_split
    def op_index(self, a, indices):
        a = self._t(a)
        key = []
        for ix in indices:
            if ix is None:
                key.append(slice(None))
            elif isinstance(ix, np.ndarray) and ix.dtype != object:
                key.append(ix)
            else:
                raise UnsupportedOperation("indexing with an index that depends on the state or action")
        return a[tuple(key)]
    def op_diagonal(self, a, offset=0, dim1=0, dim2=1): return np.diagonal(self._t(a), offset, dim1, dim2)
    def op_diag_embed(self, a, offset=0, dim1=-2, dim2=-1):
        a = self._t(a)
        if a.ndim != 1 or offset != 0:
            raise UnsupportedOperation("diag_embed of a batch or with an offset")
        out = _full((a.size, a.size), _lit(0.0))
        for k in range(a.size):
            out[k, k] = a[k]
        return out

    def _same(self, a, *_, **__): return self._t(a)
    op_clone = op_alias = op_contiguous = op_lift_fresh_copy = op_lift_fresh = op_view_as_real = op__to_copy = op_to = _same

    def op__to_copy(self, a, dtype=None, **kw):
        a = self._t(a)
        if dtype is not None and dtype.is_floating_point:
            return self._map1(a, self._num)
        if dtype is not None and dtype == torch.bool:
            return self._map1(a, lambda e: e if e.kind == "b" else self.compare(e, _lit(0.0), "!="))
        if dtype is not None and not dtype.is_floating_point:
            if all(e.kind in ("i", "b") for e in a.reshape(-1)):
                return a
            raise UnsupportedOperation(f"cast of a computed value to {dtype}")
        return a
    def op_detach(self, a): return self._map1(a, self.stop_gradient)

    # constants -------------------------------------------------------------------------------------------------------------------------
    def op_scalar_tensor(self, v, **kw): return _full((), self._as_e(v))
    def op_full(self, size, v, **kw): return _full(tuple(size), self._as_e(v))
    def op_zeros(self, size, **kw): return _full(tuple(size), _lit(0.0))
    def op_ones(self, size, **kw): return _full(tuple(size), _lit(1.0))
    def op_empty(self, size, **kw): return _full(tuple(size), _lit(0.0))
    def op_zeros_like(self, a, **kw): return _full(self._t(a).shape, _lit(0.0))
    def op_ones_like(self, a, **kw): return _full(self._t(a).shape, _lit(1.0))
    def op_full_like(self, a, v, **kw): return _full(self._t(a).shape, self._as_e(v))
    op_empty_like = op_zeros_like
    def op_new_zeros(self, a, size, **kw): return _full(tuple(size), _lit(0.0))
    def op_new_ones(self, a, size, **kw): return _full(tuple(size), _lit(1.0))
    def op_new_full(self, a, size, v, **kw): return _full(tuple(size), self._as_e(v))
    op_new_empty = op_new_zeros
    def op_arange(self, *a, **kw):
        nums = [v for v in a if isinstance(v, (int, float))]
        vals = np.arange(*nums)
        return vals.astype(np.int64) if all(isinstance(v, int) for v in nums) and not (kw.get("dtype") and kw["dtype"].is_floating_point) else _arr([_lit(v) for v in vals])
    def op_eye(self, n, m=None, **kw):
        m = n if m is None else m
        out = _full((n, m), _lit(0.0))
        for k in range(min(n, m)):
            out[k, k] = _lit(1.0)
        return out

    # in-place forms: the destination is a numpy view of its base, as the torch tensor is a view of its storage ---------------------------
    def _write(self, dst, value):
        dst = self._t(dst)
        src = np.broadcast_to(self._t(value), dst.shape)
        if not dst.flags.writeable:
            raise UnsupportedOperation("in-place write into an expanded tensor")
        dst[...] = src
        return dst
    def op_copy_(self, dst, src, non_blocking=False): return self._write(dst, src)
    def op_fill_(self, dst, v): return self._write(dst, _full((), self._as_e(v) if not isinstance(v, np.ndarray) else v.reshape(-1)[0]))
    def op_zero_(self, dst): return self._write(dst, _full((), _lit(0.0)))
    def op_add_(self, dst, b, alpha=1): return self._write(dst, self.op_add(dst, b, alpha))
    def op_sub_(self, dst, b, alpha=1): return self._write(dst, self.op_sub(dst, b, alpha))
    def op_mul_(self, dst, b): return self._write(dst, self.op_mul(dst, b))
    def op_div_(self, dst, b): return self._write(dst, self.op_div(dst, b))
    def op_neg_(self, dst): return self._write(dst, self.op_neg(dst))
    def op_clamp_(self, dst, min=None, max=None): return self._write(dst, self.op_clamp(dst, min, max))
    def op_index_put_(self, dst, indices, values, accumulate=False):
        dst = self._t(dst)
        key = []
        for ix in indices:
            if ix is None:
                key.append(slice(None))
            elif isinstance(ix, np.ndarray) and ix.dtype != object:
                key.append(ix)
            else:
                raise UnsupportedOperation("index_put with an index that depends on the state or action")
        key = tuple(key)
        vals = self._t(values)
        if accumulate:
            vals = self._map2(dst[key], vals, self.add)
        dst[key] = np.broadcast_to(vals, dst[key].shape)
        return dst
    def op_index_put(self, a, indices, values, accumulate=False): return self.op_index_put_(self._t(a).copy(), indices, values, accumulate)


def _trace(fn, example):
    from torch.fx.experimental.proxy_tensor import make_fx
    name = getattr(fn, "__name__", "function")
    try:
        # data-dependent Python control flow cannot be seen by a trace: on fake tensors it raises instead of being baked in
        make_fx(fn, tracing_mode="fake", _allow_non_fake_inputs=True)(*example)
    except Exception as exc:       # noqa: BLE001 -- (the exception types of the tracer are not a stable API)
        text = f"{type(exc).__name__}: {exc}"
        if "data-dependent" in text or "DataDependent" in text or "Could not guard" in text or "_local_scalar_dense" in text:
            raise UnsupportedOperation(
                f"`{name}` branches in Python on a value that depends on the state or action (`if x[0] > 0:`, `float(x[0])`, `.item()`): a trace "
                "would keep only the branch the example input took.  Write the choice with torch.where / torch.clamp / torch.maximum, or run the "
                "env through TorchEnv directly") from exc
        # (anything else the fake trace trips over -- e.g. an op without a fake kernel -- is left to the real trace below to report)
    return make_fx(fn)(*example)


def translate(transition_fn, cost_fn, final_cost_fn, state_size, action_size, device="cpu"):
    """-> ``(source, params)``: the three device templates of a ``DeviceEnv`` and its parameter vector (float32 ``[P]``)."""
    source, params, _ = translate_ex(transition_fn, cost_fn, final_cost_fn, state_size, action_size, device)
    return source, params


def translate_ex(transition_fn, cost_fn, final_cost_fn, state_size, action_size, device="cpu"):
    """``translate`` + what the translator learned on the way: ``info["cost_is_piecewise_linear"]`` -- stage and final cost are piecewise
    affine in (x, u) (every operation on the way from the inputs to the cost keeps second derivatives at exactly zero), as the reference's
    HVAC and Reservoir costs are."""
    n, m = int(state_size), int(action_size)
    gen = torch.Generator().manual_seed(0)
    x = (0.25 + torch.rand(n, generator=gen)).to(device)
    u = (0.25 + torch.rand(m, generator=gen)).to(device)
    params = ParamTable()
    xe = _arr([_E(f"x[{i}]", "S") for i in range(n)])
    ue = _arr([_E(f"u[{a}]", "S") for a in range(m)])
    pieces = []
    cost_lin = True
    for name, fn, example, inputs in (("transition", transition_fn, (x, u), (xe, ue)), ("cost", cost_fn, (x, u), (xe, ue)),
                                      ("final_cost", final_cost_fn, (x,), (xe,))):
        with torch.no_grad():
            gm = _trace(fn, example)
        tr = _Translator(gm, params, None)
        out = tr.run([a.copy() for a in inputs])
        if isinstance(out, (list, tuple)):
            if len(out) != 1:
                raise UnsupportedOperation(f"`{name}` must return one tensor, got {len(out)} values")
            out = out[0]
        out = tr._t(out)
        if name == "transition":
            flat = out.reshape(-1)
            if flat.size != n:
                raise ValueError(f"transition returns {tuple(out.shape)}, expected {n} values")
            stores = [f"    x_next[{i}] = S({tr._num(flat[i]).code});" for i in range(n)]      # (_num may still add a statement: stores come last)
            pieces.append("template <class S> __device__ void transition(const float *p, const S *x, const S *u, S *x_next)\n{\n"
                          + "\n".join(tr.prog.lines + stores) + "\n}\n")
        else:
            if out.size != 1:
                raise ValueError(f"{name} returns {tuple(out.shape)}, expected a scalar")
            e = tr._num(out.reshape(-1)[0])
            cost_lin = cost_lin and (e.kind != "S" or e.lin)
            sig = ("template <class S> __device__ S cost(const float *p, const S *x, const S *u)" if name == "cost"
                   else "template <class S> __device__ S final_cost(const float *p, const S *x)")
            pieces.append(sig + "\n{\n" + "\n".join(tr.prog.lines + [f"    return S({e.code});"]) + "\n}\n")
    return "\n".join(pieces), params.array(), {"cost_is_piecewise_linear": bool(cost_lin)}
