// trig.h -- sine / cosine of fp32 arguments, shared by every kernel that evaluates the Reservoir env
// (evaporation 0.5 sin(x / cap) x, tfmpc/envs/reservoir/__init__.py:85-89, and its derivative), so that all of them
// round identically: the value depends on the argument alone, never on which kernel or lane evaluates it.
//
// |r| <= pi/2 -- every physical state, r = level / capacity -- takes minimax polynomials through r^9 (sine) and r^10 (cosine)
// in Horner form: approximation error < 5e-9, six instructions (rounds 1 - 3: Taylor through r^13 / r^14, eight).  Any other argument (and NaN) takes the general
// path: reduction in fp64 (k = rint(r 2/pi), y = r - k pi/2 with a two-part pi/2: exact to fp32 rounding for
// |r| < 2^30; beyond that -- a reservoir a billion times over capacity -- the result is some value in [-1, 1]), the
// Cephes single-precision minimax kernels on [-pi/4, pi/4] and the quadrant fix-up, all branch-free.  The general path
// sits behind ONE wave-uniform branch per call ("does any lane of the wave need it"); lanes are then selected by their
// own argument.  libm's sinf carries a Payne-Hanek slow path whose registers and branches every caller pays for; the
// MI355X runs fp64 FMAs at half the fp32 rate, which makes seven fp64 instructions the cheaper general reduction.
#pragma once

#include <hip/hip_runtime.h>

namespace tfmpc {

__device__ __forceinline__ void sincos_general(float r, float &s, float &c)
{
    const double rd = (double)r;
    const double kd = __builtin_rint(rd * 0.63661977236758134308);
    double yd = __builtin_fma(-kd, 1.57079632679489655800e+00, rd);
    yd = __builtin_fma(-kd, 6.12323399573676603587e-17, yd);
    float y = (float)yd;
    y = fminf(fmaxf(y, -0.7853982f), 0.7853982f);              // only bites when the reduction has lost the argument
    const int k = (int)kd;                                     // saturating conversion
    const float z = y * y;
    const float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    const float S = fmaf(y * z, ps, y);
    const float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    const float C = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
    const float s0 = (k & 1) ? C : S, c0 = (k & 1) ? S : C;
    s = (k & 2) ? -s0 : s0;
    c = ((k + 1) & 2) ? -c0 : c0;
    if (!(r == r)) { s = r; c = r; }                           // NaN in, NaN out
}

__device__ __forceinline__ bool trig_small(float r) { return fabsf(r) <= 1.5707963f; }       // false for NaN

// Round 4: MINIMAX polynomials on [0, pi/2] instead of the Taylor ones through r^13 / r^14 -- two multiply-adds fewer each at the same
// accuracy (sine through r^9: |error| < 4.7e-9 before rounding; cosine through r^10: < 4e-10; evaluated in fp32 over 2.5 M arguments in
// [0, pi/2] the sine is within 1.85 ulp (Taylor: 1.96) and the cosine within 7.5e-8 absolute (the same)).  Coefficients: weighted
// minimax fit (Lawson iteration) of (sin r - r) / r^3 and (cos r - 1 + r^2 / 2) / r^4 in z = r^2.  Every kernel that evaluates the
// Reservoir env shares these functions, so they still round identically everywhere.
__device__ __forceinline__ float sin_small(float r)
{
    const float z = r * r;
    float p = fmaf(z, 2.60005346e-06f, -1.98066145e-04f);
    p = fmaf(p, z, 8.33301728e-03f);
    p = fmaf(p, z, -1.66666571e-01f);
    return fmaf(r * z, p, r);
}
__device__ __forceinline__ float cos_small(float r)
{
    const float z = r * r;
    float p = fmaf(z, -2.61938020e-07f, 2.47693035e-05f);
    p = fmaf(p, z, -1.38885691e-03f);
    p = fmaf(p, z, 4.16666558e-02f);
    return fmaf(z * z, p, fmaf(-0.5f, z, 1.0f));
}

// "do all N arguments of this lane take the short path?" as ONE comparison of their largest magnitude (v_max3_f32 with |.| operand
// modifiers: N / 2 instructions instead of N compares and N - 1 ands).  A NaN argument is not seen by the maximum -- and need not be:
// the short polynomials return NaN for NaN, as the general path does.
template <int N>
__device__ __forceinline__ bool trig_all_small(const float (&r)[N])
{
    float m = fabsf(r[0]);
#pragma unroll
    for (int e = 1; e < N; ++e) m = fmaxf(m, fabsf(r[e]));
    return m <= 1.5707963f;
}

// s[e] = sin(r[e]) (and c[e] = cos(r[e])) for the N arguments of a lane
template <int N>
__device__ __forceinline__ void sin_vec(const float (&r)[N], float (&s)[N])
{
#pragma unroll
    for (int e = 0; e < N; ++e) s[e] = sin_small(r[e]);
    if (__any(!trig_all_small(r))) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
            float sg, cg;
            sincos_general(r[e], sg, cg);
            if (!trig_small(r[e])) s[e] = sg;
        }
    }
}
template <int N>
__device__ __forceinline__ void sincos_vec(const float (&r)[N], float (&s)[N], float (&c)[N])
{
#pragma unroll
    for (int e = 0; e < N; ++e) { s[e] = sin_small(r[e]); c[e] = cos_small(r[e]); }
    if (__any(!trig_all_small(r))) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
            float sg, cg;
            sincos_general(r[e], sg, cg);
            if (!trig_small(r[e])) { s[e] = sg; c[e] = cg; }
        }
    }
}

__device__ __forceinline__ void sincos_f32(float r, float &s, float &c)
{
    const float r1[1] = {r};
    float s1[1], c1[1];
    sincos_vec<1>(r1, s1, c1);
    s = s1[0]; c = c1[0];
}
__device__ __forceinline__ float sin_f32(float r)
{
    const float r1[1] = {r};
    float s1[1];
    sin_vec<1>(r1, s1);
    return s1[0];
}

}  // namespace tfmpc
