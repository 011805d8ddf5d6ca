// envs.h -- device-side models of the reference's differentiable environments.
//
// The reference gets every Jacobian/Hessian from TensorFlow autodiff
// (tfmpc/envs/diffenv.py:13-101).  Here each env carries its closed-form
// derivatives (each confirmed against the reference's own
// tests/test_env_*.py and, in tests/, against an independent torch-autodiff
// restatement).  All functions are wave-cooperative: the 64 lanes of the wave that
// owns the instance call them together; x, u and every output live in that wave's
// LDS.  Dense outputs (f_x[n][n], f_u[n][m], l_xx ...) use leading dimension
// odd_ld(cols) like the rest of the wave kernels.
//
//   kind            reference file                              parameters p[i]
//   ENV_LQ          tfmpc/solvers/lqr.py:36-57                  F[n][n+m], f[n], C[d][d], c[d]
//   ENV_NAVLQR      tfmpc/envs/lqr/navigation/__init__.py:30-47 goal[n]; scalar[0] = beta
//   ENV_NAVIGATION  tfmpc/envs/navigation/__init__.py:34-74     goal[2], center[Z][2], decay[Z]
//   ENV_HVAC        tfmpc/envs/hvac/__init__.py:69-149          temp_outside, temp_hall, lower, upper,
//                                                               k_out = adj_out/R_out, k_hall = adj_hall/R_hall,
//                                                               capacity, air_max, G[n][n] = (adj|adj^T)/R_wall
//   ENV_RESERVOIR   tfmpc/envs/reservoir/__init__.py:47-105     max_res_cap, lower, upper, low_penalty,
//                                                               high_penalty, set_point_penalty,
//                                                               rain = shape*scale, downstream[n][n]
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/tfmpc_hip.h"
#include "trig.h"
#include "wave_ops.h"

namespace tfmpc {

// Per-wave LDS copy of one instance's env parameters.
struct EnvLds {
    int n, m, zones;
    float beta;
    const float *p[TFMPC_ENV_MAX_PARAMS];
    const float *low, *high;   // [m]
    const float *aux;          // per-instance constants derived once per kernel (env_aux_len floats)
};

__host__ __device__ inline int env_param_len(int kind, int i, int n, int m, int zones)
{
    const int d = n + m;
    switch (kind) {
    case TFMPC_ENV_LQ: { const int len[4] = {n * d, n, d * d, d}; return i < 4 ? len[i] : 0; }
    case TFMPC_ENV_NAVLQR: return i == 0 ? n : 0;
    case TFMPC_ENV_NAVIGATION: { const int len[3] = {n, zones * n, zones}; return i < 3 ? len[i] : 0; }
    case TFMPC_ENV_HVAC: return i < 8 ? n : (i == 8 ? n * n : 0);
    case TFMPC_ENV_RESERVOIR: return i < 7 ? n : (i == 7 ? n * n : 0);
    case TFMPC_ENV_USER: return i == 0 ? zones : 0;          // p0: the user env's parameter floats, TfmpcEnv::n_zones of them (user_env.h)
    }
    return 0;
}

// The n x n matrix parameter of HVAC (adjacency G, p[8]) and Reservoir (downstream D, p[7]) is kept
// in LDS with an ODD leading dimension ldn = n | 1: rows walked across lanes then hit every bank.
__host__ __device__ inline int env_matrix_param(int kind)
{
    return kind == TFMPC_ENV_HVAC ? 8 : (kind == TFMPC_ENV_RESERVOIR ? 7 : -1);
}
// Constants derived once per kernel so that no per-timestep code recomputes them:
//   HVAC  dtc[n] = TIME_DELTA / cap, gsum[n] = sum_k G[i][k]
__host__ __device__ inline int env_aux_len(int kind, int n) { return kind == TFMPC_ENV_HVAC ? 2 * n : 0; }

__host__ __device__ inline size_t env_lds_floats(int kind, int n, int m, int zones)
{
    size_t s = 2 * (size_t)m + env_aux_len(kind, n);
    for (int i = 0; i < TFMPC_ENV_MAX_PARAMS; ++i)
        s += (i == env_matrix_param(kind)) ? (size_t)n * (n | 1) : (size_t)env_param_len(kind, i, n, m, zones);
    return s;
}

// Copies instance b's parameters (and the action bounds) into LDS at `base`;
// returns the first free float after them.
__device__ inline float *env_load(EnvLds &e, const TfmpcEnv &g, int b, float *base)
{
    const int lane = lane_id();
    e.n = g.n; e.m = g.m; e.zones = g.n_zones; e.beta = g.scalar[0];
    const int n = g.n, ldn = n | 1, mat = env_matrix_param(g.kind);
    float *p = base;
    for (int i = 0; i < TFMPC_ENV_MAX_PARAMS; ++i) {
        const int len = env_param_len(g.kind, i, g.n, g.m, g.n_zones);
        e.p[i] = p;
        if (len > 0) {
            const float *src = g.p[i] + (size_t)b * g.stride[i];
            if (i == mat) {
                for (int j = lane; j < len; j += kWave) p[(j / n) * ldn + j % n] = src[j];
                p += n * ldn;
            } else {
                for (int j = lane; j < len; j += kWave) p[j] = src[j];
                p += len;
            }
        }
    }
    float *lo = p, *hi = p + g.m;
    for (int j = lane; j < g.m; j += kWave) { lo[j] = g.low[j]; hi[j] = g.high[j]; }
    e.low = lo; e.high = hi;
    p += 2 * g.m;
    float *aux = p;
    e.aux = aux;
    if (g.kind == TFMPC_ENV_HVAC) {
        wsync();
        const float *cap = e.p[6], *G = e.p[8];
        float *dtc = aux, *gsum = aux + n;
        for (int i = lane; i < n; i += kWave) {
            dtc[i] = 1.0f / cap[i];                  // TIME_DELTA / cap, TIME_DELTA = 1 (hvac :13)
            float gs = 0.0f;
            for (int k = 0; k < n; ++k) gs += G[i * ldn + k];
            gsum[i] = gs;
        }
    }
    return p + env_aux_len(g.kind, n);
}

__device__ __forceinline__ float signf(float y) { return (y > 0.0f) ? 1.0f : ((y < 0.0f) ? -1.0f : 0.0f); }

template <int KIND> struct Env;

// ----------------------------------------------------------------------- LQ ---
template <> struct Env<TFMPC_ENV_LQ> {
    // true iff every second derivative of cost and final cost is identically zero
    static constexpr bool kPiecewiseLinearCost = false;
    // x' = F [x;u] + f                                               lqr.py:36-39
    static __device__ void transition(const EnvLds &e, const float *x, const float *u, float *xn)
    {
        const int n = e.n, m = e.m, d = n + m;
        const float *F = e.p[0], *f = e.p[1];
        for (int i = lane_id(); i < n; i += kWave) {
            float s = f[i];
            for (int j = 0; j < n; ++j) s = fmaf(F[i * d + j], x[j], s);
            for (int j = 0; j < m; ++j) s = fmaf(F[i * d + n + j], u[j], s);
            xn[i] = s;
        }
    }
    // 1/2 z^T C z + c^T z                                            lqr.py:41-47
    static __device__ float cost(const EnvLds &e, const float *x, const float *u)
    {
        const int n = e.n, m = e.m, d = n + m;
        const float *C = e.p[2], *c = e.p[3];
        float part = 0.0f;
        for (int r = lane_id(); r < d; r += kWave) {
            float cz = 0.0f;
            for (int j = 0; j < n; ++j) cz = fmaf(C[r * d + j], x[j], cz);
            for (int j = 0; j < m; ++j) cz = fmaf(C[r * d + n + j], u[j], cz);
            const float zr = r < n ? x[r] : u[r - n];
            part += zr * (0.5f * cz + c[r]);
        }
        return wave_sum(part);
    }
    // 1/2 x^T C_xx x + c_x^T x                                       lqr.py:49-57
    static __device__ float final_cost(const EnvLds &e, const float *x)
    {
        const int n = e.n, d = n + e.m;
        const float *C = e.p[2], *c = e.p[3];
        float part = 0.0f;
        for (int r = lane_id(); r < n; r += kWave) {
            float cz = 0.0f;
            for (int j = 0; j < n; ++j) cz = fmaf(C[r * d + j], x[j], cz);
            part += x[r] * (0.5f * cz + c[r]);
        }
        return wave_sum(part);
    }
    static __device__ float linearize(const EnvLds &e, const float *x, const float *u, float *fx, float *fu,
                                      float *lx, float *lu, float *lxx, float *luu, float *lux)
    {
        const int n = e.n, m = e.m, d = n + m, ldn = odd_ld(n), ldm = odd_ld(m);
        const float *F = e.p[0], *C = e.p[2], *c = e.p[3];
        const int lane = lane_id();
        for (int idx = lane; idx < n * d; idx += kWave) {
            const int i = idx / d, j = idx - i * d;
            if (j < n) fx[i * ldn + j] = F[idx]; else fu[i * ldm + (j - n)] = F[idx];
        }
        // gradient / Hessian of 1/2 z^T C z + c^T z use the symmetric part of C
        for (int r = lane; r < d; r += kWave) {
            float g = c[r];
            for (int j = 0; j < d; ++j) {
                const float zj = j < n ? x[j] : u[j - n];
                g = fmaf(0.5f * (C[r * d + j] + C[j * d + r]), zj, g);
            }
            if (r < n) lx[r] = g; else lu[r - n] = g;
        }
        for (int idx = lane; idx < d * d; idx += kWave) {
            const int r = idx / d, j = idx - r * d;
            const float h = 0.5f * (C[r * d + j] + C[j * d + r]);
            if (r < n && j < n) lxx[r * ldn + j] = h;
            else if (r >= n && j >= n) luu[(r - n) * ldm + (j - n)] = h;
            else if (r >= n) lux[(r - n) * ldn + j] = h;
        }
        return cost(e, x, u);
    }
    static __device__ float final_quad(const EnvLds &e, const float *x, float *lx, float *lxx)
    {
        const int n = e.n, d = n + e.m, ldn = odd_ld(n);
        const float *C = e.p[2], *c = e.p[3];
        for (int r = lane_id(); r < n; r += kWave) {
            float g = c[r];
            for (int j = 0; j < n; ++j) g = fmaf(0.5f * (C[r * d + j] + C[j * d + r]), x[j], g);
            lx[r] = g;
        }
        for (int idx = lane_id(); idx < n * n; idx += kWave) {
            const int r = idx / n, j = idx - r * n;
            lxx[r * ldn + j] = 0.5f * (C[r * d + j] + C[j * d + r]);
        }
        return final_cost(e, x);
    }
};

// ------------------------------------------------------------------- NAVLQR ---
template <> struct Env<TFMPC_ENV_NAVLQR> {
    // true iff every second derivative of cost and final cost is identically zero
    static constexpr bool kPiecewiseLinearCost = false;
    static __device__ void transition(const EnvLds &e, const float *x, const float *u, float *xn)
    {
        for (int i = lane_id(); i < e.n; i += kWave) xn[i] = x[i] + u[i];            // :32
    }
    static __device__ float cost(const EnvLds &e, const float *x, const float *u)
    {
        const float *g = e.p[0];
        float c1 = 0.0f, c2 = 0.0f;
        for (int i = lane_id(); i < e.n; i += kWave) {
            const float dx = x[i] - g[i];
            c1 = fmaf(dx, dx, c1);
            c2 = fmaf(u[i], u[i], c2);
        }
        return wave_sum(c1) + e.beta * wave_sum(c2);                                 // :39-41
    }
    static __device__ float final_cost(const EnvLds &e, const float *x)
    {
        const float *g = e.p[0];
        float c1 = 0.0f;
        for (int i = lane_id(); i < e.n; i += kWave) {
            const float dx = x[i] - g[i];
            c1 = fmaf(dx, dx, c1);
        }
        return wave_sum(c1);                                                          // :47
    }
    static __device__ float linearize(const EnvLds &e, const float *x, const float *u, float *fx, float *fu,
                                      float *lx, float *lu, float *lxx, float *luu, float *lux)
    {
        const int n = e.n, ldn = odd_ld(n);
        const float *g = e.p[0];
        for (int idx = lane_id(); idx < n * n; idx += kWave) {
            const int i = idx / n, j = idx - i * n;
            const float id = (i == j) ? 1.0f : 0.0f;
            fx[i * ldn + j] = id;
            fu[i * ldn + j] = id;
            lxx[i * ldn + j] = 2.0f * id;
            luu[i * ldn + j] = 2.0f * e.beta * id;
            lux[i * ldn + j] = 0.0f;
        }
        for (int i = lane_id(); i < n; i += kWave) {
            lx[i] = 2.0f * (x[i] - g[i]);
            lu[i] = 2.0f * e.beta * u[i];
        }
        return cost(e, x, u);
    }
    static __device__ float final_quad(const EnvLds &e, const float *x, float *lx, float *lxx)
    {
        const int n = e.n, ldn = odd_ld(n);
        const float *g = e.p[0];
        for (int idx = lane_id(); idx < n * n; idx += kWave) {
            const int i = idx / n, j = idx - i * n;
            lxx[i * ldn + j] = (i == j) ? 2.0f : 0.0f;
        }
        for (int i = lane_id(); i < n; i += kWave) lx[i] = 2.0f * (x[i] - g[i]);
        return final_cost(e, x);
    }
};

// --------------------------------------------------------------- NAVIGATION ---
// The deceleration factor of one zone (envs/navigation/__init__.py:61-74), shared by EVERY kernel that evaluates the
// env (wave kernels here, lane kernels in ilqr_lane.hip) so that all of them round identically: hardware square
// root and exponential (1 ulp each; exp(a) = exp2(a log2 e)) and reciprocals by v_rcp_f32 + one Newton step -- the
// sequential rollouts of the lane kernels are bound by the length of this instruction stream (libm's expf / sqrtf and
// IEEE divisions were ~100 of a rollout step's ~170 instructions).
__device__ __forceinline__ float env_rcp(float x)
{
    const float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ float nav_zone_lambda(float r2, float decay, float *r_out, float *ex_out)
{
    const float r = __builtin_amdgcn_sqrtf(r2);
    const float ex = __builtin_amdgcn_exp2f(-decay * r * 1.44269504088896340736f);
    if (r_out) { *r_out = r; *ex_out = ex; }
    return 2.0f * env_rcp(1.0f + ex) - 1.0f;
}
// h_z = d lambda_z / d r = 2 d e^{-d r} / (1 + e^{-d r})^2
__device__ __forceinline__ float nav_zone_slope(float decay, float ex) { return 2.0f * decay * ex * env_rcp((1.0f + ex) * (1.0f + ex)); }

template <> struct Env<TFMPC_ENV_NAVIGATION> {
    // true iff every second derivative of cost and final cost is identically zero
    static constexpr bool kPiecewiseLinearCost = false;
    // lambda = prod_z (2 / (1 + exp(-decay_z |x - c_z|)) - 1); also d lambda / d x.   :61-74
    static __device__ float deceleration(const EnvLds &e, const float *x, float *grad /* [n] regs or null */)
    {
        const int n = e.n, Z = e.zones;
        const float *center = e.p[1], *decay = e.p[2];
        float lam = 1.0f;
        for (int z = 0; z < Z; ++z) {
            float r2 = 0.0f;
            for (int i = 0; i < n; ++i) { const float dlt = x[i] - center[z * n + i]; r2 = fmaf(dlt, dlt, r2); }
            lam *= nav_zone_lambda(r2, decay[z], nullptr, nullptr);
        }
        if (grad) {
            for (int i = 0; i < n; ++i) grad[i] = 0.0f;
            for (int z = 0; z < Z; ++z) {
                float r2 = 0.0f;
                for (int i = 0; i < n; ++i) { const float dlt = x[i] - center[z * n + i]; r2 = fmaf(dlt, dlt, r2); }
                float r, ex;
                nav_zone_lambda(r2, decay[z], &r, &ex);
                const float h = nav_zone_slope(decay[z], ex);
                float others = 1.0f;
                for (int y = 0; y < Z; ++y) {
                    if (y == z) continue;
                    float q2 = 0.0f;
                    for (int i = 0; i < n; ++i) { const float dlt = x[i] - center[y * n + i]; q2 = fmaf(dlt, dlt, q2); }
                    others *= nav_zone_lambda(q2, decay[y], nullptr, nullptr);
                }
                for (int i = 0; i < n; ++i) grad[i] += h * (x[i] - center[z * n + i]) * env_rcp(r) * others;
            }
        }
        return lam;
    }
    static __device__ void transition(const EnvLds &e, const float *x, const float *u, float *xn)
    {
        const float lam = deceleration(e, x, nullptr);
        for (int i = lane_id(); i < e.n; i += kWave) xn[i] = fmaf(lam, u[i], x[i]);     // :36-43
    }
    static __device__ float cost(const EnvLds &e, const float *x, const float *)
    {
        return Env<TFMPC_ENV_NAVLQR>::final_cost(e, x);                                  // :50-53
    }
    static __device__ float final_cost(const EnvLds &e, const float *x)
    {
        return Env<TFMPC_ENV_NAVLQR>::final_cost(e, x);                                  // :56-59
    }
    static __device__ float linearize(const EnvLds &e, const float *x, const float *u, float *fx, float *fu,
                                      float *lx, float *lu, float *lxx, float *luu, float *lux)
    {
        const int n = e.n, ldn = odd_ld(n);
        const float *g = e.p[0];
        float grad[8];
        const float lam = deceleration(e, x, grad);
        for (int idx = lane_id(); idx < n * n; idx += kWave) {
            const int i = idx / n, j = idx - i * n;
            const float id = (i == j) ? 1.0f : 0.0f;
            fx[i * ldn + j] = id + u[i] * grad[j];          // I + u grad(lambda)^T
            fu[i * ldn + j] = lam * id;
            lxx[i * ldn + j] = 2.0f * id;
            luu[i * ldn + j] = 0.0f;
            lux[i * ldn + j] = 0.0f;
        }
        for (int i = lane_id(); i < n; i += kWave) {
            lx[i] = 2.0f * (x[i] - g[i]);
            lu[i] = 0.0f;
        }
        return cost(e, x, u);
    }
    static __device__ float final_quad(const EnvLds &e, const float *x, float *lx, float *lxx)
    {
        return Env<TFMPC_ENV_NAVLQR>::final_quad(e, x, lx, lxx);
    }
};

// --------------------------------------------------------------------- HVAC ---
template <> struct Env<TFMPC_ENV_HVAC> {
    // true iff every second derivative of cost and final cost is identically zero
    static constexpr bool kPiecewiseLinearCost = true;
    static constexpr float CAP_AIR = 1.006f, COST_AIR = 1.0f, TEMP_AIR = 40.0f, TIME_DELTA = 1.0f;   // :10-13
    static constexpr float PENALTY = 20000.0f, SET_POINT_PENALTY = 10.0f;                              // :14-15

    static __device__ void transition(const EnvLds &e, const float *x, const float *u, float *xn)
    {
        const int n = e.n, ldg = odd_ld(n);
        const float *t_out = e.p[0], *t_hall = e.p[1], *k_out = e.p[4], *k_hall = e.p[5], *cap = e.p[6],
                    *air_max = e.p[7], *G = e.p[8];
        // conduction between rooms, sum_j -G[i][j] (x_i - x_j) (:131-139): each row is split over
        // `parts` adjacent lanes (all 64 lanes busy at n = 32) and every lane keeps 4 independent
        // partial sums so the LDS reads pipeline; the column walk is rotated by the row index so the
        // lanes of a half-wave hit different banks (leading dimension n = 32 would be a 32-way conflict).
        const int parts = (n <= 32) ? 2 : 1;      // the register-resident kernels (ilqr_adjoint.hip) sum in this order too
        const int lane = lane_id();
        for (int base = 0; base < n; base += kWave / parts) {
            const int i = base + lane / parts, part = lane % parts;
            float between = 0.0f;
            if (i < n) {
                const int per = (n + parts - 1) / parts;
                const int j0 = part * per, j1 = (j0 + per < n) ? j0 + per : n;
                float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
                const float xi = x[i];
                int jj = j0;
                for (; jj + 4 <= j1; jj += 4) {
                    int c0 = jj + i, c1 = jj + 1 + i, c2 = jj + 2 + i, c3 = jj + 3 + i;
                    c0 -= (c0 >= n) ? n : 0; c1 -= (c1 >= n) ? n : 0; c2 -= (c2 >= n) ? n : 0; c3 -= (c3 >= n) ? n : 0;
                    s0 = fmaf(-G[i * ldg + c0], xi - x[c0], s0);
                    s1 = fmaf(-G[i * ldg + c1], xi - x[c1], s1);
                    s2 = fmaf(-G[i * ldg + c2], xi - x[c2], s2);
                    s3 = fmaf(-G[i * ldg + c3], xi - x[c3], s3);
                }
                for (; jj < j1; ++jj) {
                    int c0 = jj + i;
                    c0 -= (c0 >= n) ? n : 0;
                    s0 = fmaf(-G[i * ldg + c0], xi - x[c0], s0);
                }
                between = (s0 + s1) + (s2 + s3);
            }
            if (parts >= 2) between += quad_xor1(between);
            if (parts >= 4) between += quad_xor2(between);
            if (i < n && part == 0) {
                const float air = u[i] * air_max[i];                                          // :72
                const float heating = air * CAP_AIR * (TEMP_AIR - x[i]);                      // :74
                const float outside = k_out[i] * (t_out[i] - x[i]);                           // :143-144
                const float hall = k_hall[i] * (t_hall[i] - x[i]);                            // :148-149
                xn[i] = x[i] + TIME_DELTA / cap[i] * (heating + between + outside + hall);    // :80-88
            }
        }
    }
    static __device__ float penalties(const EnvLds &e, const float *x, int i)
    {
        const float lo = e.p[2][i], hi = e.p[3][i];
        const float oob = PENALTY * (fmaxf(0.0f, lo - x[i]) + fmaxf(0.0f, x[i] - hi));   // :97-100
        const float sp = SET_POINT_PENALTY * fabsf((lo + hi) / 2 - x[i]);                 // :101-105
        return oob + sp;
    }
    static __device__ float cost(const EnvLds &e, const float *x, const float *u)
    {
        float part = 0.0f;
        for (int i = lane_id(); i < e.n; i += kWave) part += COST_AIR * (u[i] * e.p[7][i]) + penalties(e, x, i);
        return wave_sum(part);                                                            // :107-110
    }
    static __device__ float final_cost(const EnvLds &e, const float *x)
    {
        float part = 0.0f;
        for (int i = lane_id(); i < e.n; i += kWave) part += penalties(e, x, i);
        return wave_sum(part);                                                            // :112-129
    }
    static __device__ void cost_grad_x(const EnvLds &e, const float *x, float *lx)
    {
        for (int i = lane_id(); i < e.n; i += kWave) {
            const float lo = e.p[2][i], hi = e.p[3][i];
            lx[i] = PENALTY * (-(lo > x[i] ? 1.0f : 0.0f) + (x[i] > hi ? 1.0f : 0.0f))
                    - SET_POINT_PENALTY * signf((lo + hi) / 2 - x[i]);
        }
    }
    // d x'_i / d x_i and d x'_i / d u_i (the only state-dependent entries of f_x, f_u)      :69-89
    static __device__ __forceinline__ float fx_diag(const EnvLds &e, const float *u, int i)
    {
        const int n = e.n;
        const float *k_out = e.p[4], *k_hall = e.p[5], *air_max = e.p[7], *G = e.p[8];
        const float dtc = e.aux[i], gsum = e.aux[n + i];
        return 1.0f + dtc * (G[i * odd_ld(n) + i] - u[i] * air_max[i] * CAP_AIR - gsum - k_out[i] - k_hall[i]);
    }
    static __device__ __forceinline__ float fu_diag(const EnvLds &e, const float *x, int i)
    {
        return e.aux[i] * e.p[7][i] * CAP_AIR * (TEMP_AIR - x[i]);
    }
    // Q_x[i] = l_x[i] + sum_k f_x[k][i] V_x[k] and Q_u[a] = l_u[a] + sum_k f_u[k][a] V_x[k] with the
    // entries of f_x, f_u formed on the fly, in the summation order of the dense path (k ascending;
    // f_u is diagonal, its exact-zero terms are skipped)
    static __device__ float adjoint_qx(const EnvLds &e, const float *x, const float *u, const float *Vx, float lx_i, int i)
    {
        const int n = e.n, ldn = odd_ld(n);
        const float *dtc = e.aux, *G = e.p[8];
        float acc = fmaf(fx_diag(e, u, i), Vx[i], lx_i);            // diagonal term first (ilqr_core.h backward_pass)
        for (int kk = 0; kk < n; ++kk) if (kk != i) acc = fmaf(dtc[kk] * G[kk * ldn + i], Vx[kk], acc);
        return acc;
    }
    static __device__ float adjoint_qu(const EnvLds &e, const float *x, const float *u, const float *Vx, int a)
    {
        return fmaf(fu_diag(e, x, a), Vx[a], COST_AIR * e.p[7][a]);
    }
    static __device__ __forceinline__ float cost_grad_x_i(const EnvLds &e, const float *x, int i)
    {
        const float lo = e.p[2][i], hi = e.p[3][i];
        return PENALTY * (-(lo > x[i] ? 1.0f : 0.0f) + (x[i] > hi ? 1.0f : 0.0f)) - SET_POINT_PENALTY * signf((lo + hi) / 2 - x[i]);
    }
    // f_x, f_u and l_x, l_u only (first-order model; what the adjoint backward pass needs)
    static __device__ float linearize1(const EnvLds &e, const float *x, const float *u, float *fx, float *fu,
                                       float *lx, float *lu)
    {
        const int n = e.n, ldn = odd_ld(n);
        const float *air_max = e.p[7], *G = e.p[8];
        for (int idx = lane_id(); idx < n * n; idx += kWave) {
            const int i = idx / n, j = idx - i * n;
            float v = e.aux[i] * G[i * ldn + j];                          // dtc_i * G[i][j]
            float d = 0.0f;
            if (i == j) { v = fx_diag(e, u, i); d = fu_diag(e, x, i); }
            fx[i * ldn + j] = v;
            fu[i * ldn + j] = d;
        }
        cost_grad_x(e, x, lx);
        for (int i = lane_id(); i < n; i += kWave) lu[i] = COST_AIR * air_max[i];
        return cost(e, x, u);
    }
    static __device__ float linearize(const EnvLds &e, const float *x, const float *u, float *fx, float *fu,
                                      float *lx, float *lu, float *lxx, float *luu, float *lux)
    {
        const int n = e.n, ldn = odd_ld(n);
        for (int idx = lane_id(); idx < n * n; idx += kWave) {
            const int o = (idx / n) * ldn + idx % n;
            lxx[o] = 0.0f; luu[o] = 0.0f; lux[o] = 0.0f;
        }
        return linearize1(e, x, u, fx, fu, lx, lu);
    }
    static __device__ float final_grad(const EnvLds &e, const float *x, float *lx)
    {
        cost_grad_x(e, x, lx);
        return final_cost(e, x);
    }
    static __device__ float final_quad(const EnvLds &e, const float *x, float *lx, float *lxx)
    {
        const int n = e.n, ldn = odd_ld(n);
        for (int idx = lane_id(); idx < n * n; idx += kWave) lxx[(idx / n) * ldn + idx % n] = 0.0f;
        cost_grad_x(e, x, lx);
        return final_cost(e, x);
    }
};

// ---------------------------------------------------------------- RESERVOIR ---
template <> struct Env<TFMPC_ENV_RESERVOIR> {
    // true iff every second derivative of cost and final cost is identically zero
    static constexpr bool kPiecewiseLinearCost = true;
    static __device__ void transition(const EnvLds &e, const float *x, const float *u, float *xn)
    {
        const int n = e.n, ldd = odd_ld(n);
        const float *cap = e.p[0], *rain = e.p[6], *D = e.p[7];
        // inflow_i = sum_j D[j][i] u_j x_j (:91): rows split over `parts` lanes, 4 partial sums per lane
        const int parts = (n <= 32) ? 2 : 1;      // the register-resident kernels (ilqr_adjoint.hip) sum in this order too
        const int lane = lane_id();
        for (int base = 0; base < n; base += kWave / parts) {
            const int i = base + lane / parts, part = lane % parts;
            float inflow = 0.0f;
            if (i < n) {
                const int per = (n + parts - 1) / parts;
                const int j0 = part * per, j1 = (j0 + per < n) ? j0 + per : n;
                float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
                int j = j0;
                for (; j + 4 <= j1; j += 4) {
                    s0 = fmaf(D[j * ldd + i], u[j] * x[j], s0);
                    s1 = fmaf(D[(j + 1) * ldd + i], u[j + 1] * x[j + 1], s1);
                    s2 = fmaf(D[(j + 2) * ldd + i], u[j + 2] * x[j + 2], s2);
                    s3 = fmaf(D[(j + 3) * ldd + i], u[j + 3] * x[j + 3], s3);
                }
                for (; j < j1; ++j) s0 = fmaf(D[j * ldd + i], u[j] * x[j], s0);
                inflow = (s0 + s1) + (s2 + s3);
            }
            if (parts >= 2) inflow += quad_xor1(inflow);
            if (parts >= 4) inflow += quad_xor2(inflow);
            if (i < n && part == 0) {
                const float vaporated = 0.5f * sin_f32(x[i] * (1.0f / cap[i])) * x[i];                    // :87
                xn[i] = x[i] + rain[i] + inflow - vaporated - u[i] * x[i];                    // :56-60
            }
        }
    }
    static __device__ float cost(const EnvLds &e, const float *x, const float *)
    {
        float part = 0.0f;
        for (int i = lane_id(); i < e.n; i += kWave) {
            const float lo = e.p[1][i], hi = e.p[2][i];
            const float c1 = -e.p[3][i] * fmaxf(0.0f, lo - x[i]);                         // :70
            const float c2 = -e.p[4][i] * fmaxf(0.0f, x[i] - hi);                         // :71
            const float c3 = -e.p[5][i] * fabsf((lo + hi) / 2.0f - x[i]);                 // :72
            part += c1 + c2 + c3;
        }
        return wave_sum(part);
    }
    static __device__ float final_cost(const EnvLds &e, const float *x) { return cost(e, x, nullptr); }   // :81-83
    static __device__ void cost_grad_x(const EnvLds &e, const float *x, float *lx)
    {
        for (int i = lane_id(); i < e.n; i += kWave) {
            const float lo = e.p[1][i], hi = e.p[2][i];
            const float LP = -e.p[3][i], HP = -e.p[4][i], SP = -e.p[5][i];
            lx[i] = -LP * (lo > x[i] ? 1.0f : 0.0f) + HP * (x[i] > hi ? 1.0f : 0.0f) - SP * signf((lo + hi) / 2.0f - x[i]);
        }
    }
    // Q_x[j] = l_x[j] + sum_k f_x[k][j] V_x[k], Q_u[a] = sum_k f_u[k][a] V_x[k] with
    // f_x[k][j] = D[j][k] u_j (+ evaporation/outflow term on the diagonal), f_u[k][a] = D[a][k] x_a
    // (- x_a on the diagonal) formed on the fly from row j (a) of D, in the dense path's order
    static __device__ float adjoint_qx(const EnvLds &e, const float *x, const float *u, const float *Vx, float lx_j, int j)
    {
        const int n = e.n, ldn = odd_ld(n);
        const float *Drow = e.p[7] + j * ldn;
        const float uj = u[j];
        const float r = x[j] * (1.0f / e.p[0][j]);
        float sr, cr;
            sincos_f32(r, sr, cr);
            const float diag_extra = 1.0f - 0.5f * (cr * r + sr) - uj;
        float acc = fmaf(Drow[j] * uj + diag_extra, Vx[j], lx_j);   // diagonal term first (ilqr_core.h backward_pass)
        for (int kk = 0; kk < n; ++kk) if (kk != j) acc = fmaf(Drow[kk] * uj, Vx[kk], acc);
        return acc;
    }
    static __device__ float adjoint_qu(const EnvLds &e, const float *x, const float *u, const float *Vx, int a_)
    {
        const int n = e.n, ldn = odd_ld(n);
        const float *Drow = e.p[7] + a_ * ldn;
        const float xa = x[a_];
        float acc = fmaf(Drow[a_] * xa - xa, Vx[a_], 0.0f);         // diagonal term first
        for (int kk = 0; kk < n; ++kk) if (kk != a_) acc = fmaf(Drow[kk] * xa, Vx[kk], acc);
        return acc;
    }
    static __device__ __forceinline__ float cost_grad_x_i(const EnvLds &e, const float *x, int i)
    {
        const float lo = e.p[1][i], hi = e.p[2][i];
        const float LP = -e.p[3][i], HP = -e.p[4][i], SP = -e.p[5][i];
        return -LP * (lo > x[i] ? 1.0f : 0.0f) + HP * (x[i] > hi ? 1.0f : 0.0f) - SP * signf((lo + hi) / 2.0f - x[i]);
    }
    static __device__ float linearize1(const EnvLds &e, const float *x, const float *u, float *fx, float *fu,
                                       float *lx, float *lu)
    {
        const int n = e.n, ldn = odd_ld(n);
        const float *cap = e.p[0], *D = e.p[7];
        for (int idx = lane_id(); idx < n * n; idx += kWave) {
            const int i = idx / n, j = idx - i * n;
            float a = D[j * ldn + i] * u[j];        // D^T diag(u)
            float b = D[j * ldn + i] * x[j];        // D^T diag(x)
            if (i == j) {
                const float r = x[i] * (1.0f / cap[i]);      // every kernel: x times the rounded reciprocal (trig.h)
                float sr, cr;
                sincos_f32(r, sr, cr);
                a += 1.0f - 0.5f * (cr * r + sr) - u[i];
                b -= x[i];
            }
            fx[i * ldn + j] = a;
            fu[i * ldn + j] = b;
        }
        cost_grad_x(e, x, lx);
        for (int i = lane_id(); i < n; i += kWave) lu[i] = 0.0f;
        return cost(e, x, u);
    }
    static __device__ float linearize(const EnvLds &e, const float *x, const float *u, float *fx, float *fu,
                                      float *lx, float *lu, float *lxx, float *luu, float *lux)
    {
        const int n = e.n, ldn = odd_ld(n);
        for (int idx = lane_id(); idx < n * n; idx += kWave) {
            const int o = (idx / n) * ldn + idx % n;
            lxx[o] = 0.0f; luu[o] = 0.0f; lux[o] = 0.0f;
        }
        return linearize1(e, x, u, fx, fu, lx, lu);
    }
    static __device__ float final_grad(const EnvLds &e, const float *x, float *lx)
    {
        cost_grad_x(e, x, lx);
        return final_cost(e, x);
    }
    static __device__ float final_quad(const EnvLds &e, const float *x, float *lx, float *lxx)
    {
        const int n = e.n, ldn = odd_ld(n);
        for (int idx = lane_id(); idx < n * n; idx += kWave) lxx[(idx / n) * ldn + idx % n] = 0.0f;
        cost_grad_x(e, x, lx);
        return final_cost(e, x);
    }
};

}  // namespace tfmpc
