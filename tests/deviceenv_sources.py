"""Two of the reference's envs written as DeviceEnv source (tfmpc/envs/deviceenv.py): the C++ a user would write for an env of their own.
Used by tests/test_deviceenv_gpu.py (held against the built-in kernels and the restatement) and by bench.py's DeviceEnv line."""

import numpy as np

# tfmpc/envs/navigation/__init__.py:34-74 with two deceleration zones: p = goal[2], centers[2][2], decay[2]
NAVIGATION = """
template <class S> __device__ void transition(const float *p, const S *x, const S *u, S *x_next)
{
    S lam = S(1.0f);
    for (int z = 0; z < 2; ++z) {                                                   // :61-74
        const S dx = x[0] - p[2 + 2 * z], dy = x[1] - p[3 + 2 * z];
        const S r = sqrt(dx * dx + dy * dy);
        lam = lam * (2.0f / (1.0f + exp(-p[6 + z] * r)) - 1.0f);
    }
    x_next[0] = x[0] + lam * u[0];                                                  // :36-43
    x_next[1] = x[1] + lam * u[1];
}
template <class S> __device__ S cost(const float *p, const S *x, const S *u)       // :47-50
{
    const S dx = x[0] - p[0], dy = x[1] - p[1];
    return dx * dx + dy * dy;
}
template <class S> __device__ S final_cost(const float *p, const S *x)             // :52-55
{
    const S dx = x[0] - p[0], dy = x[1] - p[1];
    return dx * dx + dy * dy;
}
"""


def navigation_params(cfg):
    return np.concatenate([np.ravel(cfg["goal"]), np.ravel(cfg["deceleration"]["center"]), np.ravel(cfg["deceleration"]["decay"])]).astype(np.float32)


# tfmpc/envs/reservoir/__init__.py:47-105 (cec=True) for n reservoirs: p = cap, lower, upper, LP, HP, SP, rain (n each; penalties positive), D[n][n]
def reservoir_source(n):
    return """
constexpr int n = %d;
template <class S> __device__ void transition(const float *p, const S *x, const S *u, S *x_next)
{
    const float *cap = p, *rain = p + 6 * n, *D = p + 7 * n;
    S out[n];
    for (int i = 0; i < n; ++i) out[i] = u[i] * x[i];                                // :89-91 outflow
    for (int i = 0; i < n; ++i) {
        S inflow = S(0.0f);
        for (int k = 0; k < n; ++k) inflow = inflow + D[k * n + i] * out[k];         // :93-95 D^T outflow
        x_next[i] = x[i] + rain[i] + inflow - 0.5f * sin(x[i] / cap[i]) * x[i] - out[i];      // :56-60, :85-87
    }
}
template <class S> __device__ S cost(const float *p, const S *x, const S *u)        // :63-79
{
    const float *lo = p + n, *hi = p + 2 * n, *LP = p + 3 * n, *HP = p + 4 * n, *SP = p + 5 * n;
    S c = S(0.0f);
    for (int i = 0; i < n; ++i)
        c = c + LP[i] * max(0.0f, lo[i] - x[i]) + HP[i] * max(0.0f, x[i] - hi[i]) + SP[i] * abs((lo[i] + hi[i]) / 2.0f - x[i]);
    return c;
}
template <class S> __device__ S final_cost(const float *p, const S *x) { return cost<S>(p, x, x); }      // :81-83
""" % n


def reservoir_params(cfg):
    g = lambda k: np.ravel(np.asarray(cfg[k], dtype=np.float32))
    rain = g("rain_shape") * g("rain_scale")
    return np.concatenate([g("max_res_cap"), g("lower_bound"), g("upper_bound"), -g("low_penalty"), -g("high_penalty"),
                           -g("set_point_penalty"), rain, g("downstream")]).astype(np.float32)


# tfmpc/solvers/lqr.py:36-57 as an env (the LQ env of the iLQR API line): p = F[n][n+m], f[n], C[n+m][n+m], c[n+m] -- per instance
def lq_source(n, m):
    return """
constexpr int n = %d, m = %d, d = n + m;
template <class S> __device__ void transition(const float *p, const S *x, const S *u, S *x_next)      // :36-39
{
    const float *F = p, *f = p + n * d;
    for (int i = 0; i < n; ++i) {
        S s = S(f[i]);
        for (int j = 0; j < n; ++j) s = s + F[i * d + j] * x[j];
        for (int j = 0; j < m; ++j) s = s + F[i * d + n + j] * u[j];
        x_next[i] = s;
    }
}
template <class S> __device__ S cost(const float *p, const S *x, const S *u)                           // :41-47
{
    const float *C = p + n * d + n, *c = C + d * d;
    S total = S(0.0f);
    for (int r = 0; r < d; ++r) {
        S cz = S(0.0f);
        for (int j = 0; j < n; ++j) cz = cz + C[r * d + j] * x[j];
        for (int j = 0; j < m; ++j) cz = cz + C[r * d + n + j] * u[j];
        const S zr = r < n ? x[r] : u[r - n];
        total = total + zr * (0.5f * cz + c[r]);
    }
    return total;
}
template <class S> __device__ S final_cost(const float *p, const S *x)                                 // :49-57
{
    const float *C = p + n * d + n, *c = C + d * d;
    S total = S(0.0f);
    for (int r = 0; r < n; ++r) {
        S cz = S(0.0f);
        for (int j = 0; j < n; ++j) cz = cz + C[r * d + j] * x[j];
        total = total + x[r] * (0.5f * cz + c[r]);
    }
    return total;
}
""" % (n, m)


def lq_params(F, f, C, c):
    B = F.shape[0]
    return np.concatenate([F.reshape(B, -1), f.reshape(B, -1), C.reshape(B, -1), c.reshape(B, -1)], axis=1).astype(np.float32)


# an env of the user's own (not one of the reference's): tests/test_deviceenv_gpu.py, tools/probes/r6_pendulum_rate.py
PENDULUM = """
// a damped pendulum with a torque limit: x = [angle, angular velocity], u = [torque]; p = dt, g / l, damping, target angle, q_angle, q_velocity, r
template <class S> __device__ void transition(const float *p, const S *x, const S *u, S *x_next)
{
    x_next[0] = x[0] + p[0] * x[1];
    x_next[1] = x[1] + p[0] * (u[0] - p[1] * sin(x[0]) - p[2] * x[1]);
}
template <class S> __device__ S cost(const float *p, const S *x, const S *u)
{
    const S e = x[0] - p[3];
    return p[4] * e * e + p[5] * x[1] * x[1] + p[6] * u[0] * u[0];
}
template <class S> __device__ S final_cost(const float *p, const S *x)
{
    const S e = x[0] - p[3];
    return 10.0f * (p[4] * e * e + p[5] * x[1] * x[1]);
}
"""
