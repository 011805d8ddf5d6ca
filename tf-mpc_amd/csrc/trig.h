// trig.h -- sine / cosine of one fp32 argument, shared by every kernel that evaluates the Reservoir env
// (evaporation 0.5 sin(x / cap) x, tfmpc/envs/reservoir/__init__.py:85-89, and its derivative), so that all of them
// round identically.
//
// Branch-free: the argument is reduced in fp64 (k = rint(r 2/pi), y = r - k pi/2 with a two-part pi/2: exact to fp32
// rounding for |r| < 2^30; beyond that -- a reservoir a billion times over capacity -- the result is some value in
// [-1, 1]), then the Cephes single-precision minimax kernels on [-pi/4, pi/4] (peak relative error 1.2e-7) and the
// quadrant fix-up.  libm's sinf carries a Payne-Hanek slow path whose registers and branches every caller pays for; the
// MI355X runs fp64 FMAs at half the fp32 rate, which makes the seven fp64 instructions here the cheaper reduction.
#pragma once

#include <hip/hip_runtime.h>

namespace tfmpc {

__device__ __forceinline__ void sincos_f32(float r, float &s, float &c)
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

__device__ __forceinline__ float sin_f32(float r)
{
    float s, c;
    sincos_f32(r, s, c);
    return s;
}

}  // namespace tfmpc
