// mfma_bf16x3.h -- fp32-accurate 16x16x16 products on the bf16 matrix cores of gfx950, shared by
// the LQR sweep (lqr_mfma16x8.hip) and the iLQR-on-LQ-env sweep (ilqr_lq_mfma.hip).
#pragma once

#include <hip/hip_runtime.h>

namespace tfmpc {
namespace bf3 {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// ---- fp32 products on the bf16 matrix cores ("bf16x3") ------------------------------------------
// x = h + m + l with h, m, l bf16 (24 mantissa bits in all).  A 16x16x16 product X Y is evaluated
// as Xh Yh + Xh Ym + Xm Yh + Xm Ym + Xh Yl + Xl Yh (the dropped terms are below 2^-24 relative),
// accumulated in fp32 by three v_mfma_f32_16x16x32_bf16, each carrying TWO of the terms in its
// K = 32: 3 x 16 cycles instead of 4 x 32 for the f32 MFMA, and -- unlike the f32 MFMA, which runs
// on the vector FMA lanes -- on the matrix pipe proper, so it overlaps other waves' VALU work.
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;

__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    const bf16x2 p = {(__bf16)a, (__bf16)b};                 // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, p);
}

struct Split3 { unsigned h01, h23, m01, m23, l01, l23; };

__device__ __forceinline__ Split3 split3(f32x4 x)
{
    Split3 s;
    s.h01 = pack_bf16(x[0], x[1]);
    s.h23 = pack_bf16(x[2], x[3]);
    const float r0 = x[0] - __uint_as_float(s.h01 << 16), r1 = x[1] - __uint_as_float(s.h01 & 0xffff0000u);
    const float r2 = x[2] - __uint_as_float(s.h23 << 16), r3 = x[3] - __uint_as_float(s.h23 & 0xffff0000u);
    s.m01 = pack_bf16(r0, r1);
    s.m23 = pack_bf16(r2, r3);
    s.l01 = pack_bf16(r0 - __uint_as_float(s.m01 << 16), r1 - __uint_as_float(s.m01 & 0xffff0000u));
    s.l23 = pack_bf16(r2 - __uint_as_float(s.m23 << 16), r3 - __uint_as_float(s.m23 & 0xffff0000u));
    return s;
}

// fragments (8 bf16 per lane: k-slots 0..3 = first term rows r, 4..7 = second term rows r)
//   resident operand Y (F~):   [h|h], [m|m], [h|l]
//   per-step operand X (V, W): [h|m] (used against [h|h] and [m|m]) and [l|h] (against [h|l]);
//   both are 4-register windows of ONE 6-register block [l01 l23 h01 h23 m01 m23], so building
//   them costs no register moves.
struct ConstFrag { u32x4 hh, mm, hl; };
using u32x6 = __attribute__((ext_vector_type(6))) unsigned;
struct VarFrag {
    u32x6 r;                                               // l01 l23 h01 h23 m01 m23
    __device__ __forceinline__ u32x4 hm() const { return __builtin_shufflevector(r, r, 2, 3, 4, 5); }
    __device__ __forceinline__ u32x4 lh() const { return __builtin_shufflevector(r, r, 0, 1, 2, 3); }
};

__device__ __forceinline__ ConstFrag const_frag(f32x4 x)
{
    const Split3 s = split3(x);
    return ConstFrag{u32x4{s.h01, s.h23, s.h01, s.h23}, u32x4{s.m01, s.m23, s.m01, s.m23}, u32x4{s.h01, s.h23, s.l01, s.l23}};
}
__device__ __forceinline__ VarFrag var_frag(f32x4 x)
{
    const Split3 s = split3(x);
    return VarFrag{u32x6{s.l01, s.l23, s.h01, s.h23, s.m01, s.m23}};
}
__device__ __forceinline__ f32x4 mfma_bf(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// acc += X (as A operand) x Y (as B operand):  Xh Yh + Xm Yh | Xh Ym + Xm Ym | Xl Yh + Xh Yl
__device__ __forceinline__ f32x4 mm_var_const(const VarFrag &x, const ConstFrag &y, f32x4 acc)
{
    acc = mfma_bf(x.hm(), y.hh, acc);
    acc = mfma_bf(x.hm(), y.mm, acc);
    return mfma_bf(x.lh(), y.hl, acc);
}
// acc += Y (as A operand) x X (as B operand)
__device__ __forceinline__ f32x4 mm_const_var(const ConstFrag &y, const VarFrag &x, f32x4 acc)
{
    acc = mfma_bf(y.hh, x.hm(), acc);
    acc = mfma_bf(y.mm, x.hm(), acc);
    return mfma_bf(y.hl, x.lh(), acc);
}

}  // namespace bf3
}  // namespace tfmpc
