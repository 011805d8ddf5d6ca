// ilqr_adjoint_mfma.hip -- iLQR.solve (tfmpc/solvers/ilqr.py:214-355) for HVAC (tfmpc/envs/hvac/__init__.py) and
// Reservoir (tfmpc/envs/reservoir/__init__.py) when the env is SHARED by the batch (every parameter has batch
// stride 0, which is what the reference's one-env-object API gives): SIXTEEN instances per wavefront, the coupling
// matrix products on the matrix cores.
//
// On these envs the backward pass is the costate recursion (bang-bang branch, ilqr.py:140-141) and every
// rollout is open loop, so all the dense work of an iteration is  y = M z  with a matrix that is the same for every
// instance and every timestep: HVAC's wall-conduction matrix G, Reservoir's `downstream` D.  With one instance per
// wave (ilqr_adjoint.hip) that is a mat-vec per wave and step, ~100 vector instructions and an LDS exchange of x.
// Here the 16 states of a wave form the matrix X (n x 16) and  Y = M X  is 16 `v_mfma_f32_16x16x4_f32` at n = 32
// (exact fp32 multiply-adds), with M resident in 16 registers as the A operands and X ALREADY in operand layout:
//   * lane (j, q) = (lane & 15, lane >> 4) holds rows 16 b + 4 q + r (b < NT, r < 4) of column j = instance j -- the
//     accumulator layout of the 16x16 tile b -- as registers e = 4 b + r;
//   * enumerating the contraction as k-step (b, r) = "row 16 b + 4 q + r from lane quarter q", register e of X is the
//     B operand of k-step e as it stands, and the A operand of output tile a is M[16 a + (lane & 15)][16 b + 4 q + r]:
//     no LDS, no cross-lane traffic between a step's result and the next step's operand;
//   * everything else of a step is element-wise on the 4 NT values a lane owns (costs, clipping, the bilinear / sine
//     terms), with per-row parameters in registers; the only cross-lane work is the column sum of the stage cost and
//     the column max of the gradient norm (two lane exchanges each).
//   * n <= 8 / n <= 4: a column carries two / four instances (rows 0-7 | 8-15, or one per lane quarter) against the
//     block-diagonal operand diag(M, .., M); the column reductions then span two lane quarters / one.
// The 16 (32, 64) instances run the reference state machine in lockstep with masked stores (as ilqr_adjoint_group_kernel does
// for its 2 / 4 groups): one costate sweep, then line-search rounds in which every instance still searching rolls out
// ITS next step size(s) WITHOUT storing them, then the step size each instance settled on is rolled out once more with
// stores; the nominal and candidate trajectories ping-pong between the output arrays and the workspace per instance (no
// copy on acceptance); of the gains only one selector bit per action goes through the workspace (k_t = bound - u_t).
//
// Arithmetic: Reservoir keeps the operation order of the wave kernels for every element-wise expression, the cost
// reduction tree and the J / value bookkeeping, so on 0/1 `downstream` matrices (every reference config: a row sum is
// then a single exact term) its results are BIT-identical to theirs -- that is how it is tested.  HVAC folds the linear
// part of the room balance into the matrix (conduction - row sums - outside / hall conductances) and is compared within
// fp32 tolerance.  Dense couplings: tolerance as well (the MFMA sums four products per accumulate).
#include <hip/hip_runtime.h>

#include <stdint.h>

#include <type_traits>

#include "../../include/tfmpc_hip.h"
#include "ilqr_adjoint.h"
#include "mfma_bf16x3.h"
#include "options.h"
#include "trig.h"
#include "wave_ops.h"

namespace tfmpc {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

constexpr int kCols = 16;                 // instances per wave

// The element-wise work of a step is written on PAIRS of adjacent rows (registers e, e + 1 of a lane: consecutive
// registers of a 16-byte trajectory load, an LDS parameter read and an MFMA accumulator alike), so that adds, multiplies
// and fused multiply-adds become v_pk_{add,mul,fma}_f32 -- two rows per instruction, each rounded exactly like the
// scalar instruction -- without the register shuffles the auto-vectoriser's own pairings needed (it paired the two
// step-size chains of the HVAC search: 67 v_mov per 373 vector instructions of a step).  max / abs / compare / select
// have no packed fp32 form and stay per row.
__device__ __forceinline__ f32x2 pr(const float *a, int e) { return f32x2{a[e], a[e + 1]}; }
__device__ __forceinline__ void unpr(float *a, int e, f32x2 v) { a[e] = v.x; a[e + 1] = v.y; }
__device__ __forceinline__ f32x2 splat(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 max2(f32x2 a, f32x2 b) { return f32x2{fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
__device__ __forceinline__ f32x2 abs2(f32x2 a) { return f32x2{fabsf(a.x), fabsf(a.y)}; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ float sgnf_(float y) { return (y > 0.0f) ? 1.0f : ((y < 0.0f) ? -1.0f : 0.0f); }

// Ordering fence between cross-lane producer / consumer phases of ONE wave (LDS and global memory): the waits of a
// workgroup-scope fence without the s_barrier.  wave_ops.h's wsync() is __syncthreads(), which is the same thing while a
// workgroup is one wave but a real barrier in the multi-wave groups below, whose waves are in different phases.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---- LDS-DMA input ring (round 3) ---------------------------------------------------------------------------------
// The inputs of a time step (the nominal actions and the selector byte of a rollout; the nominal state, action and stage
// cost of the sweep) used to be prefetched into a REGISTER ring.  At 250 registers the compiler rotates such a ring
// through copies at the loop's back edge, and the copies wait for the loads just issued (s_waitcnt vmcnt(0) once per
// trip): the prefetch distance was gone, an HBM round trip was exposed every two steps, and round 2's counters show the
// HVAC waves parked in s_waitcnt for 52 % of their cycles.  Now the loads are LDS-DMA (global_load_lds: HBM -> LDS
// without passing registers), issued kRingDepth - 1 steps ahead into a ring of LDS slots, and a step begins with
// s_waitcnt vmcnt(N), N = the loads of the YOUNGER steps, which stay in flight.  Both the issue and the wait are inline
// assembly the compiler knows nothing about: it cannot merge, copy or hoist them, and it inserts no vmcnt(0) of its
// own in front of LDS reads (which it does, once per loop trip, for the __builtin_amdgcn_global_load_lds form).  Its
// own waits stay correct: an extra outstanding load can only make a compiler-placed vmcnt(N) wait longer.  M0 (the
// LDS base of the DMA) is set in front of each issue (round 6; rounds 3 - 5 saved and restored it around each).
#define TFMPC_LDS __attribute__((address_space(3)))
__device__ __forceinline__ unsigned lds_addr_of(const void *p) { return (unsigned)(uintptr_t)(const TFMPC_LDS void *)p; }
#ifdef TFMPC_AB_M0_SAVE                 // A/B builds: round 3 - 5's form, M0 saved and restored around every issue (two more scalar slots per DMA)
#define TFMPC_DMA_ASM(BODY) unsigned keep; asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t" BODY "\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds) : "memory")
#else
// M0 is overwritten, not saved and restored (round 6): nothing else in these kernels lives in it -- gfx950's ds_* instructions do not read M0, and the
// compiler, for which M0 is a reserved register it does not allocate (it rejects it on a clobber list), emits no other use; tools/check_ring_waits.py
// holds the device assembly to that (every mention of m0 is one of these writes).  A lone wave pays ~4 cycles for each of the two copies --
// 4 DMAs a sweep step, 2 a rollout step.
#define TFMPC_DMA_ASM(BODY) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t" BODY : : "v"(g), "s"(lds) : "memory")
#endif
#ifdef TFMPC_AB_M0_SAVE
#define TFMPC_DMA_G "%1"
#else
#define TFMPC_DMA_G "%0"
#endif
__device__ __forceinline__ void dma16(const void *g, unsigned lds)          // 16 bytes per lane -> lds + 16 lane
{
    TFMPC_DMA_ASM("global_load_lds_dwordx4 " TFMPC_DMA_G ", off");
}
__device__ __forceinline__ void dma4(const void *g, unsigned lds)           // 4 bytes per lane -> lds + 4 lane
{
    TFMPC_DMA_ASM("global_load_lds_dword " TFMPC_DMA_G ", off");
}
__device__ __forceinline__ void dma1(const void *g, unsigned lds)           // 1 byte per lane -> the low byte of lds + 4 lane
{
    TFMPC_DMA_ASM("global_load_lds_ubyte " TFMPC_DMA_G ", off");
}
// three 16-byte pieces per lane, 1 KB apart in memory AND in LDS (the instruction offset moves both addresses): one M0 set-up
__device__ __forceinline__ void dma16x3(const void *g, unsigned lds)
{
    TFMPC_DMA_ASM("global_load_lds_dwordx4 " TFMPC_DMA_G ", off\n\tglobal_load_lds_dwordx4 " TFMPC_DMA_G ", off offset:1024\n\tglobal_load_lds_dwordx4 " TFMPC_DMA_G ", off offset:2048");
}
// bit E of `bits` ? a : b in two instructions that touch no condition register (compare + select is four issue slots with its hazard no-ops)
template <int E>
__device__ __forceinline__ float select_bit(unsigned bits, float a, float b)
{
    int m;
    float r;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(bits), "n"(E));      // the bit, sign-extended: 0 | ~0
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));  // (m & a) | (~m & b)
    return r;
}
template <int NV, int E = 0>
__device__ __forceinline__ void select_bits(unsigned bits, const float (&a)[NV], const float (&b)[NV], float (&o)[NV])
{
    if constexpr (E < NV) {
        o[E] = select_bit<E>(bits, a[E], b[E]);
        select_bits<NV, E + 1>(bits, a, b, o);
    }
}
template <int N>
__device__ __forceinline__ void wait_vmem() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// a slot's LDS reads have delivered before the slot is refilled
__device__ __forceinline__ void lds_reads_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Exchange across the lane quarters that hold the other rows of a column, on gfx950's row swaps:
// v_permlane16_swap a, b swaps row 1 of a with row 0 of b and row 3 of a with row 2 of b (rows of 16 lanes);
// v_permlane32_swap a, b swaps the upper half of a with the lower half of b.  With a == b == v going in,
// a' (+) b' is v (+) v[lane ^ 16] resp. v (+) v[lane ^ 32] in every lane.  (The s_nop covers the VALU-write ->
// permlane-read hazard the compiler cannot see inside the asm.)
__device__ __forceinline__ void swap16(float &a, float &b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap32(float &a, float &b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
// sum over the rows of one tile that belong to this lane's instance, balanced tree in row order (= the row tree of
// wave_sum / group_sum).  PK instances share a tile (PK = 2: 8 rows = 2 lane quarters each, PK = 4: one quarter each).
template <int PK>
__device__ __forceinline__ float tile_sum(float s4)
{
    if (PK == 4) return s4;
    float a = s4, b = s4;
    swap16(a, b);
    a += b;
    if (PK == 2) return a;
    b = a;
    swap32(a, b);
    return a + b;
}
template <int NT, int PK>
__device__ __forceinline__ float col_sum(const float (&v)[4 * NT])
{
    float t = tile_sum<PK>((v[0] + v[1]) + (v[2] + v[3]));
    if (NT == 2) t += tile_sum<PK>((v[4] + v[5]) + (v[6] + v[7]));
    return t;
}
// v_max_f32 as the instruction it is.  fmaxf() of a value the compiler cannot prove canonical (a loop-carried running maximum, the result of a lane
// swap) is preceded by a quieting v_max x, x -- one more issue slot per operand on every step of kernels whose time is their instruction count.  The
// instruction itself returns the same bits for every input but a signalling NaN (which no arithmetic here produces).  max_abs: max(a, |b|).
__device__ __forceinline__ float max_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float max_abs(float a, float b) { float r; asm("v_max_f32 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <int NT, int PK>
__device__ __forceinline__ float col_max(const float (&v)[4 * NT])         // non-negative inputs
{
    float a = v[0];
#pragma unroll
    for (int e = 1; e < 4 * NT; ++e) a = fmaxf(a, v[e]);
    if (PK == 4) return a;
    float b = a;
    swap16(a, b);
    a = max_raw(a, b);
    if (PK == 2) return a;
    b = a;
    swap32(a, b);
    return max_raw(a, b);
}

// The trajectory pointers are picked per column (output array or workspace), which hides their address space from the
// compiler: say "global" explicitly, or every access becomes a flat_* instruction (which also waits on the LDS counter).
#define TFMPC_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ TFMPC_GLOBAL T *as_global(T *p) { return (TFMPC_GLOBAL T *)p; }
template <class T>
__device__ __forceinline__ T gld(const T *p) { return *as_global(p); }
template <class T>
__device__ __forceinline__ void gst(T *p, T v) { *as_global(p) = v; }

template <int NT>
__device__ __forceinline__ int row_of(int e, int q) { return 16 * (e >> 2) + 4 * q + (e & 3); }

// per-row parameter vector p[0..n) -> this lane's 4 NT rows (rows >= n: dflt)
template <int NT>
__device__ __forceinline__ void load_rows(const float *p, int n, int q, float dflt, float (&o)[4 * NT])
{
#pragma unroll
    for (int e = 0; e < 4 * NT; ++e) {
        const int r = row_of<NT>(e, q);
        o[e] = (r < n) ? p[r] : dflt;
    }
}

// one instance-time vector v[0..n) of a trajectory <-> this lane's rows, in pieces of VW = 4, 2 or 1 floats: VW divides
// n (a piece is then inside or outside the vector as a whole) and the arrays are 4 VW-byte aligned
template <int VW> struct VecOf { using type = float; };
template <> struct VecOf<2> { using type = __attribute__((ext_vector_type(2))) float; };
template <> struct VecOf<4> { using type = f32x4; };
template <int NT, int VW>
__device__ __forceinline__ void ldv(const float *p, int n, int q, float (&o)[4 * NT])
{
    using V = typename VecOf<VW>::type;
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int c = 0; c < 4; c += VW) {
            const int r0 = 16 * b + 4 * q + c;
            if constexpr (VW == 1) {
                o[4 * b + c] = (r0 < n) ? gld(p + r0) : 0.0f;
            } else {
                V t = {};
                if (r0 < n) t = gld(reinterpret_cast<const V *>(p + r0));
#pragma unroll
                for (int i = 0; i < VW; ++i) o[4 * b + c + i] = t[i];
            }
        }
}
template <int NT, int VW>
__device__ __forceinline__ void stv(float *p, int n, int q, bool keep, const float (&v)[4 * NT])
{
    using V = typename VecOf<VW>::type;
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int c = 0; c < 4; c += VW) {
            const int r0 = 16 * b + 4 * q + c;
            if constexpr (VW == 1) {
                if (keep && r0 < n) gst(p + r0, v[4 * b + c]);
            } else {
                V t;
#pragma unroll
                for (int i = 0; i < VW; ++i) t[i] = v[4 * b + c + i];
                if (keep && r0 < n) gst(reinterpret_cast<V *>(p + r0), t);
            }
        }
}

// ---- 16-bit trajectory containers (cfg.storage_bf16): the same accessors on bf16 storage.  A value is rounded to nearest
// even when it is STORED and widened exactly when it is read; arithmetic stays fp32 (the definition of the storage mode,
// TfmpcIlqrConfig::storage_bf16: what the wave kernel emulates in fp32 containers is a real format here, half the bytes).
using bf16_t = unsigned short;
__device__ __forceinline__ float widen(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ bf16_t narrow(float v)
{
    unsigned u = __float_as_uint(v);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;          // round to nearest even on the fp32 bits (ilqr_core.h stq)
    return (bf16_t)u;
}
template <int VW> struct PackOf { using type = bf16_t; };
template <> struct PackOf<2> { using type = unsigned; };
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
template <> struct PackOf<4> { using type = u32x2; };
template <int NT, int VW>
__device__ __forceinline__ void ldv(const bf16_t *p, int n, int q, float (&o)[4 * NT])
{
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int c = 0; c < 4; c += VW) {
            const int r0 = 16 * b + 4 * q + c;
            if constexpr (VW == 1) {
                o[4 * b + c] = (r0 < n) ? widen(gld(p + r0)) : 0.0f;
            } else if constexpr (VW == 2) {
                const unsigned t = (r0 < n) ? gld(reinterpret_cast<const unsigned *>(p + r0)) : 0u;
                o[4 * b + c] = __uint_as_float(t << 16);
                o[4 * b + c + 1] = __uint_as_float(t & 0xFFFF0000u);
            } else {
                u32x2 t = {0u, 0u};
                if (r0 < n) t = gld(reinterpret_cast<const u32x2 *>(p + r0));
                o[4 * b + c] = __uint_as_float(t.x << 16);
                o[4 * b + c + 1] = __uint_as_float(t.x & 0xFFFF0000u);
                o[4 * b + c + 2] = __uint_as_float(t.y << 16);
                o[4 * b + c + 3] = __uint_as_float(t.y & 0xFFFF0000u);
            }
        }
}
template <int NT, int VW>
__device__ __forceinline__ void stv(bf16_t *p, int n, int q, bool keep, const float (&v)[4 * NT])
{
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int c = 0; c < 4; c += VW) {
            const int r0 = 16 * b + 4 * q + c;
            if constexpr (VW == 1) {
                if (keep && r0 < n) gst(p + r0, narrow(v[4 * b + c]));
            } else if constexpr (VW == 2) {
                const unsigned t = (unsigned)narrow(v[4 * b + c]) | ((unsigned)narrow(v[4 * b + c + 1]) << 16);
                if (keep && r0 < n) gst(reinterpret_cast<unsigned *>(p + r0), t);
            } else {
                u32x2 t;
                t.x = (unsigned)narrow(v[4 * b + c]) | ((unsigned)narrow(v[4 * b + c + 1]) << 16);
                t.y = (unsigned)narrow(v[4 * b + c + 2]) | ((unsigned)narrow(v[4 * b + c + 3]) << 16);
                if (keep && r0 < n) gst(reinterpret_cast<u32x2 *>(p + r0), t);
            }
        }
}
// ---- the solver's OWN trajectory buffers are WAVE-major: [time step][column][tile][lane quarter] 16-byte pieces (a lane's four
// rows of one tile; `lane` below is the piece index column * 4 NT + quarter, + 4 per tile).  A time step of the wave is one
// contiguous 2 KB (two tiles; half of that in 16-bit containers) instead of sixteen 64-byte segments 12.8 KB apart, and
// the 128 bytes of ONE column are contiguous, so that a store masked by column writes whole 64-byte sectors.  Instance-major arrays (the ABI's inputs and outputs) are touched once each: the start
// rollout reads u_init, the end of the kernel copies the nominal trajectory out.  Measured (tools/probes/traj_layout_probe.hip,
// same access sequence without the arithmetic): scattered 64-byte stores sustain 3.5 TB/s, contiguous ones 6.1 TB/s --
// a storing pass of the instance-major layout took 0.37 ms, longer than its arithmetic (0.15 ms).
constexpr int kTileElems = 4 * kWave;            // one tile of one time step: 64 lanes x 4 rows
template <int NT>
__device__ __forceinline__ void ldw(const float *p, int t, int lane, float (&o)[4 * NT])
{
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const f32x4 v = gld(reinterpret_cast<const f32x4 *>(p + ((size_t)t * NT * kTileElems) + 4 * (lane + 4 * b)));
        o[4 * b] = v[0]; o[4 * b + 1] = v[1]; o[4 * b + 2] = v[2]; o[4 * b + 3] = v[3];
    }
}
template <int NT>
__device__ __forceinline__ void stw(float *p, int t, int lane, bool keep, const float (&v)[4 * NT])
{
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const f32x4 w = {v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]};
        if (keep) gst(reinterpret_cast<f32x4 *>(p + ((size_t)t * NT * kTileElems) + 4 * (lane + 4 * b)), w);
    }
}
template <int NT>
__device__ __forceinline__ void ldw(const bf16_t *p, int t, int lane, float (&o)[4 * NT])
{
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const u32x2 w = gld(reinterpret_cast<const u32x2 *>(p + ((size_t)t * NT * kTileElems) + 4 * (lane + 4 * b)));
        o[4 * b] = __uint_as_float(w.x << 16);
        o[4 * b + 1] = __uint_as_float(w.x & 0xFFFF0000u);
        o[4 * b + 2] = __uint_as_float(w.y << 16);
        o[4 * b + 3] = __uint_as_float(w.y & 0xFFFF0000u);
    }
}
template <int NT>
__device__ __forceinline__ void stw(bf16_t *p, int t, int lane, bool keep, const float (&v)[4 * NT])
{
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        u32x2 w;
        w.x = (unsigned)narrow(v[4 * b]) | ((unsigned)narrow(v[4 * b + 1]) << 16);
        w.y = (unsigned)narrow(v[4 * b + 2]) | ((unsigned)narrow(v[4 * b + 3]) << 16);
        if (keep) gst(reinterpret_cast<u32x2 *>(p + ((size_t)t * NT * kTileElems) + 4 * (lane + 4 * b)), w);
    }
}
constexpr int kCostLd = 64;                      // stage costs: [time step][instance of the wave] (up to 16 columns x 4)
__device__ __forceinline__ float ldc(const float *p) { return gld(p); }
__device__ __forceinline__ float ldc(const bf16_t *p) { return widen(gld(p)); }
__device__ __forceinline__ void stc(float *p, float v) { gst(p, v); }
__device__ __forceinline__ void stc(bf16_t *p, float v) { gst(p, narrow(v)); }

// ---- the matrix operand of  Y = M Z ---------------------------------------------------------------------------------
// One tile (n <= 16): M in 4 registers per lane, 4 v_mfma_f32_16x16x4_f32 (exact fp32 multiply-adds), as before.
// Two tiles (n <= 32), round 2: the f32 MFMA costs 32 cycles, runs on the vector FMA lanes and overlaps with nothing
// (tools/probes/coexec_probe.hip): 16 of them were 512 of a step's ~1 300 issue cycles.  Now M and Z are split into bf16
// parts, x = h + m + l EXACTLY (8 + 8 + 8 mantissa bits, mfma_bf16x3.h), and the product is evaluated on the bf16 matrix
// cores, K = 32 = all rows of Z in ONE v_mfma_f32_16x16x32_bf16 (16 cycles) per pair of parts: k-slot s of lane quarter
// q is row 4 q + s (s < 4) or 16 + 4 q + s - 4, which is the order a lane holds its eight rows in -- Z needs no data
// movement, only the split (22 vector instructions per four rows).
//   * M exactly representable in bf16 (Reservoir's 0/1 `downstream`): M Z = M Zh + M Zm + M Zl, three MFMAs per output
//     tile, and with a single nonzero per row (every reference config) the sum (Zh + Zm) + Zl is Z bit for bit -- the
//     results stay BIT-identical to the wave kernels'.  6 MFMAs x 16 cycles + the split instead of 16 x 32.
//   * general M (HVAC's conduction matrix; dense Reservoir couplings): Mh Zh + Mh Zm + Mm Zh + Mm Zm + Mh Zl + Ml Zh, the
//     dropped terms are below 2^-24 relative (the same bound as one fp32 rounding of the product): 12 MFMAs x 16 cycles.
using bf3::u32x4;
template <int NT, bool SPARE_REGISTERS> struct MatOp;
template <bool S> struct MatOp<1, S> { float a[1][1][4]; int shift, leak_mask, pk; };      // shift / leak_mask: see MatOp<2, true>; pk: instances per column
// all three parts of M resident (HVAC: 24 registers)
template <> struct MatOp<2, false> { u32x4 h[2], m[2], l[2]; };
// only the leading part resident; the other two parts live in LDS and are read when M is not bf16-exact (Reservoir: 8 registers).
// `shift` (round 4): M is a SHIFT -- M[R][C] = 1 exactly where C == R + shift (shift = -1 | +1, rows and columns < n), 0 elsewhere:
// the transpose of / the off-diagonal part of the `downstream` matrix of a chain of reservoirs, which is what every config
// the reference holds is (/root/reference/tests/conftest.py:70-75 `linear_topology`, tfmpc/envs/reservoir/res4.config.json:13-18).
// Then M Z is a move of Z by one row: no product at all (mat_apply below).  shift == 0: any other matrix, the products above.
template <> struct MatOp<2, true> { u32x4 h[2]; const u32x4 *rest; bool exact; int shift; int leak_mask; };

struct ZParts { u32x4 h, m, l; };
__device__ __forceinline__ ZParts split_rows(const float (&z)[8])
{
    const bf3::Split3 s0 = bf3::split3(f32x4{z[0], z[1], z[2], z[3]}), s1 = bf3::split3(f32x4{z[4], z[5], z[6], z[7]});
    return ZParts{u32x4{s0.h01, s0.h23, s1.h01, s1.h23}, u32x4{s0.m01, s0.m23, s1.m01, s1.m23}, u32x4{s0.l01, s0.l23, s1.l01, s1.l23}};
}
// One tile, PK instances per column (rows 4 q + r of a lane; an instance owns 16 / PK consecutive rows = 4 / PK lane quarters):
// the shift of MatOp<2, true> inside every instance's rows.  Replaces four DEPENDENT v_mfma_f32_16x16x4_f32 (~130 cycles of
// latency on the critical path of a step of the small-env kernels, which are bound by one wave's instruction latency).
// Round 5: no branch on the direction -- both moves are selects under a wave-uniform mask (one fetch across the lane quarters, from the side the
// direction names): the small-env kernels are bound by ONE wave's instruction count, branches and register copies included (~4 cycles each).
__device__ __forceinline__ void shift_apply1(int shift, int leak_mask, int pk, const float (&z)[4], float (&acc)[4])
{
    const int QS = 4 / pk;                           // lane quarters per instance (wave-uniform)
    const int lane = lane_id(), q = lane >> 4;
    const bool down = shift < 0;                     // wave-uniform: o[R] = z[R - 1], else o[R] = z[R + 1]
    float t = 0.0f;
    if (QS > 1) {
        // (the two candidates as opaque register values: a select between two ELEMENTS of `z` is rewritten into an indexed load, and with that the
        // array -- a lane's state -- moves to scratch memory)
        float z_first = z[0], z_last = z[3];
        asm volatile("" : "+v"(z_first), "+v"(z_last));
        t = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((down ? lane - 16 : lane + 16) & 63) << 2,
                                                                   __builtin_bit_cast(int, down ? z_last : z_first)));
        t = (q % QS == (down ? 0 : QS - 1)) ? 0.0f : t;
    }
    float o[4] = {down ? t : z[1], down ? z[0] : z[2], down ? z[1] : z[3], down ? z[2] : t};
    if (leak_mask) {                                 // (only ever set for shift < 0; wave-uniform through bit 8)
        asm volatile("" ::: "memory");               // stays a branch: if-converted it is eight selects on every step of every env that has no such row
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = ((leak_mask >> e) & 1) ? 0.0f : o[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += o[e];
}
template <bool S>
__device__ __forceinline__ void mat_apply(const MatOp<1, S> &A, const float (&z)[4], float (&acc)[4])
{
    if (A.shift != 0) { shift_apply1(A.shift, A.leak_mask, A.pk, z, acc); return; }      // wave-uniform
    f32x4 c = {acc[0], acc[1], acc[2], acc[3]};
#pragma unroll
    for (int r = 0; r < 4; ++r) c = __builtin_amdgcn_mfma_f32_16x16x4f32(A.a[0][0][r], z[r], c, 0, 0, 0);
    acc[0] = c[0]; acc[1] = c[1]; acc[2] = c[2]; acc[3] = c[3];
}
__device__ __forceinline__ void mat_apply(const MatOp<2, false> &A, const float (&z)[8], float (&acc)[8])
{
    const ZParts Z = split_rows(z);
    f32x4 c[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) c[a] = f32x4{acc[4 * a], acc[4 * a + 1], acc[4 * a + 2], acc[4 * a + 3]};
#pragma unroll
    for (int a = 0; a < 2; ++a) c[a] = bf3::mfma_bf(A.h[a], Z.h, c[a]);      // the two tiles' chains interleaved
#pragma unroll
    for (int a = 0; a < 2; ++a) c[a] = bf3::mfma_bf(A.h[a], Z.m, c[a]);
#pragma unroll
    for (int a = 0; a < 2; ++a) c[a] = bf3::mfma_bf(A.m[a], Z.h, c[a]);
#pragma unroll
    for (int a = 0; a < 2; ++a) c[a] = bf3::mfma_bf(A.m[a], Z.m, c[a]);
#pragma unroll
    for (int a = 0; a < 2; ++a) c[a] = bf3::mfma_bf(A.h[a], Z.l, c[a]);
#pragma unroll
    for (int a = 0; a < 2; ++a) c[a] = bf3::mfma_bf(A.l[a], Z.h, c[a]);
#pragma unroll
    for (int a = 0; a < 2; ++a) { acc[4 * a] = c[a][0]; acc[4 * a + 1] = c[a][1]; acc[4 * a + 2] = c[a][2]; acc[4 * a + 3] = c[a][3]; }
}
// acc += M z for a shift matrix: row R of the result is z[R + shift].  A lane holds rows 16 b + 4 q + r in z[4 b + r], so three of
// the four rows of a tile move between the lane's own registers and the fourth comes from the neighbouring lane quarter
// (ds_bpermute: a rotation of the wave by 16 lanes; quarter 0 / 3 wraps into the other tile, the matrix edge gets 0).  The value
// is z itself -- what the matrix-core product (0 + Zh) + Zm + Zl returned bit for bit -- so results do not change; it replaces
// the 3-part split of eight rows (44 vector instructions) and six v_mfma_f32_16x16x32_bf16 per chain-step by 2 ds_bpermute
// + 2 selects (+ 8 adds that keep `acc + z`'s treatment of -0).  `leak_mask`: for n < 32 the row R == n of a forward shift
// would receive z[n - 1]; it is a padding row and must stay 0.
__device__ __forceinline__ void shift_apply(int shift, int leak_mask, const float (&z)[8], float (&acc)[8])
{
    const int lane = lane_id(), q = lane >> 4;
    float o[8];
    if (shift < 0) {                                                          // wave-uniform: o[R] = z[R - 1]
        const int src = ((lane - 16) & 63) << 2;
        const float t0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, z[3])));
        const float t1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, z[7])));
        o[0] = (q == 0) ? 0.0f : t0;  o[1] = z[0]; o[2] = z[1]; o[3] = z[2];
        o[4] = (q == 0) ? t0 : t1;    o[5] = z[4]; o[6] = z[5]; o[7] = z[6];
        if (leak_mask) {                                                      // wave-uniform (n < 32)
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = ((leak_mask >> e) & 1) ? 0.0f : o[e];
        }
    } else {                                                                  // o[R] = z[R + 1]
        const int src = ((lane + 16) & 63) << 2;
        const float t0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, z[0])));
        const float t1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, z[4])));
        o[0] = z[1]; o[1] = z[2]; o[2] = z[3]; o[3] = (q == 3) ? t1 : t0;
        o[4] = z[5]; o[5] = z[6]; o[6] = z[7]; o[7] = (q == 3) ? 0.0f : t1;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += o[e];
}
__device__ __forceinline__ void mat_apply(const MatOp<2, true> &A, const float (&z)[8], float (&acc)[8])
{
    if (A.shift != 0) {                                                        // wave-uniform
        shift_apply(A.shift, A.leak_mask, z, acc);
        return;
    }
    const ZParts Z = split_rows(z);
    f32x4 c[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) c[a] = f32x4{acc[4 * a], acc[4 * a + 1], acc[4 * a + 2], acc[4 * a + 3]};
#pragma unroll
    for (int a = 0; a < 2; ++a) c[a] = bf3::mfma_bf(A.h[a], Z.h, c[a]);      // (0 + Zh) + Zm + Zl: exact for a single term
#pragma unroll
    for (int a = 0; a < 2; ++a) c[a] = bf3::mfma_bf(A.h[a], Z.m, c[a]);
#pragma unroll
    for (int a = 0; a < 2; ++a) c[a] = bf3::mfma_bf(A.h[a], Z.l, c[a]);
    if (!A.exact) {                                                            // wave-uniform
        const int lane = lane_id();
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const u32x4 Am = A.rest[(2 * a) * kWave + lane], Al = A.rest[(2 * a + 1) * kWave + lane];
            c[a] = bf3::mfma_bf(Am, Z.h, c[a]);
            c[a] = bf3::mfma_bf(Am, Z.m, c[a]);
            c[a] = bf3::mfma_bf(Al, Z.h, c[a]);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) { acc[4 * a] = c[a][0]; acc[4 * a + 1] = c[a][1]; acc[4 * a + 2] = c[a][2]; acc[4 * a + 3] = c[a][3]; }
}

// TFMPC_COSTATE_COUPLING=dense (AdjointSolveArgs::dense_coupling): keep the products also for a shift matrix
__device__ __forceinline__ void force_dense(MatOp<2, false> &, bool) {}
__device__ __forceinline__ void force_dense(MatOp<2, true> &A, bool dense) { if (dense) A.shift = 0; }
template <bool S> __device__ __forceinline__ void force_dense(MatOp<1, S> &A, bool dense) { A.shift = __builtin_amdgcn_readfirstlane(dense ? 0 : A.shift); }     // (wave-uniform: a scalar register)
// "this operand is never a row move" as a compile-time fact (HVAC: its matrix carries -(row sum + outside + hall conductances) on the diagonal; were it a
// 0/1 shift all the same, the products return the moved values bit for bit): the branch around the matrix instructions and the move's code are gone
__device__ __forceinline__ void never_a_shift(MatOp<2, false> &) {}
__device__ __forceinline__ void never_a_shift(MatOp<2, true> &A) { A.shift = 0; A.leak_mask = 0; }
template <bool S> __device__ __forceinline__ void never_a_shift(MatOp<1, S> &A) { A.shift = 0; A.leak_mask = 0; }

// A copy of a lane-dependent index the optimiser cannot see through: what is computed from it inside a loop stays
// inside (hoisted out, the 16 operand addresses of each phase would stay live across the whole solve).
__device__ __forceinline__ int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ void opaque_f(float &v) { asm volatile("" : "+v"(v)); }

// A operands of  Y = M Z  for a matrix given element-wise: el(R, C) = M[R][C] (0 outside n x n).  PK > 1 (one tile,
// n <= 16 / PK): the tile carries PK instances in its rows, the operand is diag(M, .., M).
template <int PK, bool S, class F>
__device__ __forceinline__ void load_operand(int n, int i, int q, F el, MatOp<1, S> &A, u32x4 *)
{
    constexpr int kSub = 16 / PK;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int R = i, C = 4 * q + r;
        if (PK == 1) {
            A.a[0][0][r] = (R < n && C < n) ? el(R, C) : 0.0f;
        } else {
            const int Rl = R % kSub, Cl = C % kSub;
            A.a[0][0][r] = (R / kSub == C / kSub && Rl < n && Cl < n) ? el(Rl, Cl) : 0.0f;
        }
    }
    // is the (block-diagonal) operand a shift inside every instance's rows?  (see MatOp<2, true>)
    bool down = true, up = true;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int R = i, C = 4 * q + r, Rl = R % kSub, Cl = C % kSub;
        const bool in = R / kSub == C / kSub && Rl < n && Cl < n;
        down = down && A.a[0][0][r] == ((in && Cl == Rl - 1) ? 1.0f : 0.0f);
        up = up && A.a[0][0][r] == ((in && Cl == Rl + 1) ? 1.0f : 0.0f);
    }
    A.shift = __all(down) ? -1 : (__all(up) ? 1 : 0);
    A.pk = PK;
    A.leak_mask = 0;
    if (A.shift < 0 && n < kSub) {
        int mask = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) mask |= ((4 * q + e) % kSub == n) ? (1 << e) : 0;
        A.leak_mask = __any(mask != 0) ? (mask | 0x100) : 0;
    }
}
// two tiles: lane (i, q) holds, for output tile a, M[16 a + i][k-slots of quarter q] split into its three bf16 parts
template <class F>
__device__ __forceinline__ void operand_parts(int n, int i, int q, int a, F el, u32x4 &h, u32x4 &m, u32x4 &l)
{
    f32x4 v[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int R = 16 * a + i, C = 16 * b + 4 * q + r;
            v[b][r] = (R < n && C < n) ? el(R, C) : 0.0f;
        }
    const bf3::Split3 s0 = bf3::split3(v[0]), s1 = bf3::split3(v[1]);
    h = u32x4{s0.h01, s0.h23, s1.h01, s1.h23};
    m = u32x4{s0.m01, s0.m23, s1.m01, s1.m23};
    l = u32x4{s0.l01, s0.l23, s1.l01, s1.l23};
}
template <int PK, class F>
__device__ __forceinline__ void load_operand(int n, int i, int q, F el, MatOp<2, false> &A, u32x4 *)
{
    static_assert(PK == 1, "instances are packed into ONE tile");
#pragma unroll
    for (int a = 0; a < 2; ++a) operand_parts(n, i, q, a, el, A.h[a], A.m[a], A.l[a]);
}
template <int PK, class F>
__device__ __forceinline__ void load_operand(int n, int i, int q, F el, MatOp<2, true> &A, u32x4 *rest)
{
    static_assert(PK == 1, "instances are packed into ONE tile");
    const int lane = lane_id();
    bool exact = true;
    wave_sync();                                 // the previous phase's reads of `rest` are done
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        u32x4 m, l;
        operand_parts(n, i, q, a, el, A.h[a], m, l);
        rest[(2 * a) * kWave + lane] = m;
        rest[(2 * a + 1) * kWave + lane] = l;
        exact = exact && ((m[0] | m[1] | m[2] | m[3] | l[0] | l[1] | l[2] | l[3]) & 0x7FFF7FFFu) == 0u;     // +-0 parts only
    }
    wave_sync();
    A.rest = rest;
    A.exact = __all(exact);
    // is M a shift?  Every lane tests the 16 entries it holds of each output tile against both candidates.
    bool down = true, up = true;                 // M[R][C] == (C == R - 1) / (C == R + 1) on the n x n block
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int R = 16 * a + i, C = 16 * b + 4 * q + r;
                const float v = (R < n && C < n) ? el(R, C) : 0.0f;
                down = down && v == ((R < n && C < n && C == R - 1) ? 1.0f : 0.0f);
                up = up && v == ((R < n && C < n && C == R + 1) ? 1.0f : 0.0f);
            }
    A.shift = __all(down) ? -1 : (__all(up) ? 1 : 0);
    A.leak_mask = 0;
    if (A.shift < 0 && n < 32) {
        int mask = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) mask |= (16 * (e >> 2) + 4 * q + (e & 3) == n) ? (1 << e) : 0;
        A.leak_mask = __any(mask != 0) ? (mask | 0x100) : 0;          // bit 8: "some lane has a leak row" keeps the test wave-uniform
    }
}

// Per-row parameter vectors that a step uses ONCE live in LDS ([slot][32] floats, rows >= n padded), not in registers:
// a lane reads its rows 16 b + 4 q .. + 3 as one ds_read_b128 per tile (a broadcast within the lane quarter).  `qo` is
// an opaque copy of q made once per step, so that the reads stay inside the time loops instead of being hoisted back
// into registers.
constexpr int kRowSlots = 8, kRowLd = 32;
template <int NT>
__device__ __forceinline__ void lds_rows(const float *lds, int slot, int qo, float (&o)[4 * NT])
{
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const float4 t = *reinterpret_cast<const float4 *>(lds + slot * kRowLd + 16 * b + 4 * qo);
        o[4 * b] = t.x; o[4 * b + 1] = t.y; o[4 * b + 2] = t.z; o[4 * b + 3] = t.w;
    }
}
constexpr int kSlotALow = 6, kSlotAHigh = 7;          // action bounds (both envs)

// ROWREGS (round 5; the multi-wave one-tile groups): the env's constant rows are read from LDS ONCE (cache_rows) and stay in registers -- a lone wave pays
// ~4 cycles for every instruction it issues, LDS reads and their waits included, and one tile leaves the registers for it (two tiles do not)
template <int KIND, int NT, bool LEAN = false, bool ROWREGS = false> struct EnvM;

// ---------------------------------------------------------------------------------- HVAC ----
// x' = x + rcap (heating + [A x + c0])   with  A = G - diag(gsum + k_out + k_hall),  c0 = k_out t_out + k_hall t_hall
// (hvac/__init__.py:69-89, :131-149);  Q_x = l_x + V_x - u am CAP w + A^T w,  w = rcap V_x;  Q_u = COST am + am CAP (TEMP - x) w
template <int NT, bool LEAN, bool ROWREGS> struct EnvM<TFMPC_ENV_HVAC, NT, LEAN, ROWREGS> {
    static constexpr int NV = 4 * NT;
    static constexpr float CAP_AIR = 1.006f, COST_AIR = 1.0f, TEMP_AIR = 40.0f, TIME_DELTA = 1.0f;
    static constexpr float PENALTY = 20000.0f, SET_POINT_PENALTY = 10.0f;
    static constexpr int kC0 = 0, kDiag = 1;            // LDS slots
    // step sizes per line-search pass: two independent chains give a wave something to issue while the other chain waits
    // (cfg5 HVAC: 20.1 -> 16.6 ms in round 1; one step size per pass with stored candidates, re-tried in round 2: 18.0 vs 14.1 ms)
#ifdef TFMPC_SEARCH_ALPHAS            // A/B builds: step sizes per line-search pass of a one-wave group
    static constexpr int kSearchAlphas = TFMPC_SEARCH_ALPHAS;
#else
    // round 3 (LDS ring: no registers for the prefetch; lane-partial cost sums): THREE chains per pass at two tiles
    // 13.6 - 14.0 -> 13.0 - 13.1 ms (four: 15.1; one, storing its candidate: 19.0 - 19.8), same box
    static constexpr int kSearchAlphas = NT == 2 ? 3 : 2;
#endif
    float lo[NV], hi[NV], am[NV], rcap[NV];
    float c_rows[1][NV];                               // ROWREGS: the LDS rows, in registers
    const float *lds;
    __device__ __forceinline__ void cache_rows(int qo)
    {
        if constexpr (ROWREGS) lds_rows<NT>(lds, kC0, qo, c_rows[0]);         // (kC0 == 0, the only row)
    }
    template <int SLOT>
    __device__ __forceinline__ void row(int qo, float (&o)[NV]) const
    {
        if constexpr (ROWREGS) {
#pragma unroll
            for (int e = 0; e < NV; ++e) o[e] = c_rows[SLOT][e];
        } else {
            lds_rows<NT>(lds, SLOT, qo, o);
        }
    }

    // A[R][C] = G[R][C] - (R == C) (gsum_R + k_out_R + k_hall_R); the operands of a phase are loaded when it starts
    // (they come from L2) so that the two sets are never live together
    // (round 5: the diagonal's gsum_R + k_out_R + k_hall_R comes from an LDS row that `load` fills once per kernel -- summed per phase it was a
    // chain of n DEPENDENT loads, load -> wait -> add, in front of every rollout and sweep: ~20 k cycles at n = 32)
    __device__ __forceinline__ float el(const TfmpcEnv &g, int R, int C) const
    {
        const float *G = g.p[8];
        float v = G[R * g.n + C];
        if (R == C) v -= lds[kDiag * kRowLd + R];
        return v;
    }
    using Operand = MatOp<NT, LEAN>;                    // LEAN (multi-wave groups): non-leading parts in LDS
    template <int PK>
    __device__ __forceinline__ void load_forward(const TfmpcEnv &g, int i, int q, Operand &A, u32x4 *rest) const
    {
        load_operand<PK>(g.n, i, q, [&](int R, int C) { return el(g, R, C); }, A, rest);
        never_a_shift(A);
    }
    template <int PK>
    __device__ __forceinline__ void load_backward(const TfmpcEnv &g, int i, int q, Operand &A, u32x4 *rest) const
    {
        load_operand<PK>(g.n, i, q, [&](int R, int C) { return el(g, C, R); }, A, rest);
        never_a_shift(A);
    }
    __device__ void load(const TfmpcEnv &g, int lane, int q, float *lds_)
    {
        const int n = g.n;
        const float *pt_out = g.p[0], *pt_hall = g.p[1], *plo = g.p[2], *phi = g.p[3], *pk_out = g.p[4], *pk_hall = g.p[5],
                    *pcap = g.p[6], *pam = g.p[7];
        lds = lds_;
        if (lane < kRowLd) {
            lds_[kC0 * kRowLd + lane] = (lane < n) ? pk_out[lane] * pt_out[lane] + pk_hall[lane] * pt_hall[lane] : 0.0f;
            float gs = 0.0f;                             // sum_k G[R][k] in the order of k, then + k_out_R + k_hall_R: the diagonal of A is G[R][R] minus this
            if (lane < n)
                for (int k = 0; k < n; ++k) gs += g.p[8][lane * n + k];
            lds_[kDiag * kRowLd + lane] = (lane < n) ? gs + pk_out[lane] + pk_hall[lane] : 0.0f;
        }
        load_rows<NT>(plo, n, q, 0.0f, lo);
        load_rows<NT>(phi, n, q, 0.0f, hi);
        load_rows<NT>(pam, n, q, 0.0f, am);
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const int r = row_of<NT>(e, q);
            rcap[e] = (r < n) ? TIME_DELTA / pcap[r] : 0.0f;
        }
    }
    // Called when a phase (one rollout, one sweep) starts: what the optimiser derives from the parameters (midpoints,
    // products of constants ...) is then derived per phase instead of once per kernel, where the derived values of ALL
    // phases stayed live through every loop (~100 registers).  No instruction is emitted.
    __device__ __forceinline__ void fence()
    {
#pragma unroll
        for (int e = 0; e < NV; ++e) { opaque_f(lo[e]); opaque_f(hi[e]); opaque_f(am[e]); opaque_f(rcap[e]); }
        if constexpr (ROWREGS) {
#pragma unroll
            for (int e = 0; e < NV; ++e) opaque_f(c_rows[0][e]);
        }
    }
    // every stage / final cost is >= 0 for actions in [0, 1] (then partial cost sums never decrease)
    __device__ __forceinline__ bool costs_nonnegative() const
    {
        bool ok = true;
#pragma unroll
        for (int e = 0; e < NV; ++e) ok = ok && am[e] >= 0.0f;
        return __all(ok);
    }
    __device__ __forceinline__ f32x2 penalties(f32x2 x, int e) const
    {
        const f32x2 LO = pr(lo, e), HI = pr(hi, e);
        const f32x2 mid = (LO + HI) / 2;
        const f32x2 oob = PENALTY * (max2(splat(0.0f), LO - x) + max2(splat(0.0f), x - HI));  // hvac :97-100
        const f32x2 sp = SET_POINT_PENALTY * abs2(mid - x);                                   // :101-105
        return oob + sp;
    }
    __device__ __forceinline__ void stage_costs(const float (&x)[NV], const float (&u)[NV], int, float (&c)[NV]) const
    {
#pragma unroll
        for (int e = 0; e < NV; e += 2) unpr(c, e, COST_AIR * (pr(u, e) * pr(am, e)) + penalties(pr(x, e), e));    // :91-110
    }
    __device__ __forceinline__ void final_costs(const float (&x)[NV], int, float (&c)[NV]) const
    {
#pragma unroll
        for (int e = 0; e < NV; e += 2) unpr(c, e, penalties(pr(x, e), e));               // :112-129
    }
    // PENALTY (above - below) - SET_POINT_PENALTY sgn(mid - x), hvac :97-105 differentiated, without comparisons: the violation
    // v = x - clamp(x, lo, hi) is 0 inside the band and carries the side's sign outside, and c sgn(v) = med3(v 2^126, -c, c) for
    // c > 0 (the product saturates to +-inf or stays an exact 0; a violation below 1e-33 degrees would fall short of c).  The same
    // values as the compare-and-select form, in 11 instead of 20 instructions per row pair.
    __device__ __forceinline__ f32x2 grad_x(f32x2 x, int e) const
    {
        static_assert(PENALTY > 0.0f && SET_POINT_PENALTY > 0.0f, "c sgn(v) as a median needs c > 0");
        constexpr float kBig = 0x1p+126f;
        const f32x2 LO = pr(lo, e), HI = pr(hi, e);
        const f32x2 mid = (LO + HI) / 2;
        const f32x2 v = (x - f32x2{__builtin_amdgcn_fmed3f(x.x, LO.x, HI.x), __builtin_amdgcn_fmed3f(x.y, LO.y, HI.y)}) * kBig;
        const f32x2 w = (x - mid) * kBig;
        return f32x2{__builtin_amdgcn_fmed3f(v.x, -PENALTY, PENALTY), __builtin_amdgcn_fmed3f(v.y, -PENALTY, PENALTY)} +
               f32x2{__builtin_amdgcn_fmed3f(w.x, -SET_POINT_PENALTY, SET_POINT_PENALTY), __builtin_amdgcn_fmed3f(w.y, -SET_POINT_PENALTY, SET_POINT_PENALTY)};
    }
    __device__ __forceinline__ void grads(const float (&x)[NV], int, float (&gx)[NV]) const
    {
#pragma unroll
        for (int e = 0; e < NV; e += 2) unpr(gx, e, grad_x(pr(x, e), e));
    }
    // x' = x + rcap (air CAP (TEMP - x) + [A x + c0]) with the two outer operations fused (one rounding each instead of
    // two: 4 packed instructions per row pair instead of 6; round 3 -- this env was never bit-tied to the wave kernels)
    // The product A x + c0 needs the state only: the one-tile kernels issue it when a step STARTS (state_product), so that its four dependent
    // matrix instructions (40 cycles each, ~10 % of a small-env step) run under the ring's wait and the control's arithmetic instead of in front of
    // the final multiply-adds (round 5).  Two tiles keep the one-piece `step`: their schedule and register budget are cfg5's.
    static constexpr bool kProductOfStateOnly = true;
    __device__ __forceinline__ void state_product(const Operand &A, const float (&x)[NV], int qo, float (&acc)[NV]) const
    {
        row<kC0>(qo, acc);
#ifndef TFMPC_PROBE_NO_MATRIX_PRODUCT      // probe builds (tools/probes/cfg5_phases.py): what the bf16x3 product costs a step
        mat_apply(A, x, acc);
#endif
    }
    __device__ __forceinline__ void step_with(const float (&acc)[NV], const float (&x)[NV], const float (&u)[NV], float (&xn)[NV]) const
    {
#pragma unroll
        for (int e = 0; e < NV; e += 2) {
            const f32x2 X = pr(x, e);
            const f32x2 air = pr(u, e) * pr(am, e);                                       // :72
            const f32x2 h = air * (TEMP_AIR - X);                                         // :74 (x CAP_AIR in the next line)
            unpr(xn, e, fma2(pr(rcap, e), fma2(h, splat(CAP_AIR), pr(acc, e)), X));       // :80-88
        }
    }
    __device__ __forceinline__ void step(const Operand &A, const float (&x)[NV], const float (&u)[NV], int qo,
                                         float (&xn)[NV]) const
    {
        float acc[NV];
        state_product(A, x, qo, acc);
        step_with(acc, x, u, xn);
    }
    // A line-search rollout only needs J = sum of all stage costs: the costs of a lane's rows are accumulated straight
    // into two running pairs (no per-row cost values, no per-step column sum); the column sum is taken when the pass
    // ends or tests its early exit.  max(0, lo - x) + max(0, x - hi) = max(lo - x, x - hi, 0) needs lo <= hi: `ordered`.
    static constexpr bool kFusedSearchCost = true;
    __device__ __forceinline__ bool bounds_ordered() const
    {
        bool ok = true;
#pragma unroll
        for (int e = 0; e < NV; ++e) ok = ok && lo[e] <= hi[e];
        return __all(ok);
    }
    __device__ __forceinline__ void cost_accumulate(const float (&x)[NV], const float (&u)[NV], int, f32x2 (&jacc)[2]) const
    {
#pragma unroll
        for (int e = 0; e < NV; e += 2) {
            const f32x2 X = pr(x, e), LO = pr(lo, e), HI = pr(hi, e);
            const f32x2 a = LO - X, b = X - HI, d = (LO + HI) / 2 - X;
            const f32x2 oob = {fmaxf(fmaxf(a.x, b.x), 0.0f), fmaxf(fmaxf(a.y, b.y), 0.0f)};
            f32x2 acc = jacc[(e >> 1) & 1];
            acc = fma2(splat(PENALTY), oob, acc);                                         // :97-100
            acc = f32x2{fmaf(SET_POINT_PENALTY, fabsf(d.x), acc.x), fmaf(SET_POINT_PENALTY, fabsf(d.y), acc.y)};   // :101-105
            jacc[(e >> 1) & 1] = fma2(pr(u, e), pr(am, e), acc);                          // :93 (COST_AIR == 1)
        }
    }
    __device__ __forceinline__ void adjoint(const Operand &A, const float (&xh)[NV], const float (&uh)[NV],
                                            const float (&vx)[NV], int, float (&Qx)[NV], float (&Qu)[NV]) const
    {
        float w[NV];
#pragma unroll
        for (int e = 0; e < NV; e += 2) {
            const f32x2 W = pr(rcap, e) * pr(vx, e);
            unpr(w, e, W);
            unpr(Qx, e, fma2(-(pr(uh, e) * pr(am, e) * CAP_AIR), W, grad_x(pr(xh, e), e) + pr(vx, e)));
        }
        mat_apply(A, w, Qx);
#pragma unroll
        for (int e = 0; e < NV; e += 2) {
            const f32x2 d = pr(rcap, e) * pr(am, e) * CAP_AIR * (TEMP_AIR - pr(xh, e));
            unpr(Qu, e, fma2(d, pr(vx, e), COST_AIR * pr(am, e)));
        }
    }
};

// ----------------------------------------------------------------------------- RESERVOIR ----
// element-wise expressions in the order of ilqr_adjoint.hip / envs.h (reservoir/__init__.py:47-105)
template <int NT, bool LEAN, bool ROWREGS> struct EnvM<TFMPC_ENV_RESERVOIR, NT, LEAN, ROWREGS> {
    static constexpr int NV = 4 * NT;
    static constexpr int kRain = 0, kDii = 1, kLP = 2, kHP = 3, kSP = 4;      // LDS slots
    // see EnvM<HVAC>.  Two tiles, round 2: with a column's rows contiguous in the buffers the single-step-size search that
    // writes its candidates (53 GB of HBM traffic per cfg5 launch, 3.75 G vector instructions) and the two-step-size
    // search that writes nothing and rolls the accepted size out again (41 GB, 4.66 G instructions, 23 spilled registers)
    // take the same time on a box with fast memory (13.8 vs 13.4 ms); the one that moves fewer bytes is kept -- cfg5
    // Reservoir varied by 20 % between boxes while it was the heavier on HBM.
#ifdef TFMPC_SEARCH_ALPHAS            // A/B builds: step sizes per line-search pass of a one-wave group
    static constexpr int kSearchAlphas = TFMPC_SEARCH_ALPHAS;
#else
    static constexpr int kSearchAlphas = 2;
#endif
    float rcap[NV], lo[NV], hi[NV];                  // 1 / max_res_cap: x / cap is x times the rounded reciprocal everywhere
    float c_rows[ROWREGS ? 5 : 1][NV];               // ROWREGS: the five LDS rows, in registers
    const float *lds;
    __device__ __forceinline__ void cache_rows(int qo)
    {
        if constexpr (ROWREGS) {
#pragma unroll
            for (int sl = 0; sl < 5; ++sl) lds_rows<NT>(lds, sl, qo, c_rows[sl]);
        }
    }
    template <int SLOT>
    __device__ __forceinline__ void row(int qo, float (&o)[NV]) const
    {
        if constexpr (ROWREGS) {
#pragma unroll
            for (int e = 0; e < NV; ++e) o[e] = c_rows[SLOT][e];
        } else {
            lds_rows<NT>(lds, SLOT, qo, o);
        }
    }

    // forward: D^T (inflow = D^T (u x)); backward: D without its diagonal.  Loaded when a phase starts (from L2), so
    // the two operand sets are never live together.
    using Operand = MatOp<NT, true>;
    template <int PK>
    __device__ __forceinline__ void load_forward(const TfmpcEnv &g, int i, int q, Operand &A, u32x4 *rest) const
    {
        const float *D = g.p[7];
        const int n = g.n;
        load_operand<PK>(n, i, q, [&](int R, int C) { return D[C * n + R]; }, A, rest);
    }
    template <int PK>
    __device__ __forceinline__ void load_backward(const TfmpcEnv &g, int i, int q, Operand &A, u32x4 *rest) const
    {
        const float *D = g.p[7];
        const int n = g.n;
        load_operand<PK>(n, i, q, [&](int R, int C) { return (R == C) ? 0.0f : D[R * n + C]; }, A, rest);
    }
    __device__ void load(const TfmpcEnv &g, int lane, int q, float *lds_)
    {
        const int n = g.n;
        const float *pcap = g.p[0], *plo = g.p[1], *phi = g.p[2], *plp = g.p[3], *php = g.p[4], *psp = g.p[5], *prain = g.p[6],
                    *D = g.p[7];
        lds = lds_;
        if (lane < kRowLd) {
            const bool st = lane < n;
            lds_[kRain * kRowLd + lane] = st ? prain[lane] : 0.0f;
            lds_[kDii * kRowLd + lane] = st ? D[lane * n + lane] : 0.0f;
            lds_[kLP * kRowLd + lane] = st ? -plp[lane] : 0.0f;
            lds_[kHP * kRowLd + lane] = st ? -php[lane] : 0.0f;
            lds_[kSP * kRowLd + lane] = st ? -psp[lane] : 0.0f;
        }
        load_rows<NT>(pcap, n, q, 1.0f, rcap);
#pragma unroll
        for (int e = 0; e < NV; ++e) rcap[e] = 1.0f / rcap[e];
        load_rows<NT>(plo, n, q, 0.0f, lo);
        load_rows<NT>(phi, n, q, 0.0f, hi);
    }
    __device__ __forceinline__ void fence()             // see EnvM<HVAC>::fence
    {
#pragma unroll
        for (int e = 0; e < NV; ++e) { opaque_f(lo[e]); opaque_f(hi[e]); opaque_f(rcap[e]); }
        if constexpr (ROWREGS) {
#pragma unroll
            for (int sl = 0; sl < 5; ++sl)
#pragma unroll
                for (int e = 0; e < NV; ++e) opaque_f(c_rows[sl][e]);
        }
    }
    __device__ __forceinline__ bool costs_nonnegative() const          // see EnvM<HVAC>
    {
        float LP[NV], HP[NV], SP[NV];
        bool ok = true;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {                 // all rows, whatever this lane's quarter
            lds_rows<NT>(lds, kLP, q4, LP);
            lds_rows<NT>(lds, kHP, q4, HP);
            lds_rows<NT>(lds, kSP, q4, SP);
#pragma unroll
            for (int e = 0; e < NV; ++e) ok = ok && LP[e] >= 0.0f && HP[e] >= 0.0f && SP[e] >= 0.0f;
        }
        return __all(ok);
    }
    __device__ __forceinline__ void stage_costs(const float (&x)[NV], const float (&)[NV], int qo, float (&c)[NV]) const
    {
        float LP[NV], HP[NV], SP[NV];                                                     // reservoir :63-79
        row<kLP>(qo, LP);
        row<kHP>(qo, HP);
        row<kSP>(qo, SP);
#pragma unroll
        for (int e = 0; e < NV; e += 2) {
            const f32x2 X = pr(x, e), LO = pr(lo, e), HI = pr(hi, e);
            const f32x2 mid = (LO + HI) / 2.0f;
            const f32x2 c1 = pr(LP, e) * max2(splat(0.0f), LO - X);
            const f32x2 c2 = pr(HP, e) * max2(splat(0.0f), X - HI);
            const f32x2 c3 = pr(SP, e) * abs2(mid - X);
            unpr(c, e, c1 + c2 + c3);
        }
    }
    __device__ __forceinline__ void final_costs(const float (&x)[NV], int qo, float (&c)[NV]) const { stage_costs(x, x, qo, c); }   // :81-83
    static constexpr bool kFusedSearchCost = false;       // the per-row costs keep the wave kernels' expression (and bits)
    static constexpr bool kProductOfStateOnly = false;    // (the coupling product is of u x: nothing to start early)
    __device__ __forceinline__ bool bounds_ordered() const { return true; }
    __device__ __forceinline__ void cost_accumulate(const float (&)[NV], const float (&)[NV], int, f32x2 (&)[2]) const {}
    __device__ __forceinline__ void grads(const float (&x)[NV], int qo, float (&gx)[NV]) const
    {
        float LP[NV], HP[NV], SP[NV];
        row<kLP>(qo, LP);
        row<kHP>(qo, HP);
        row<kSP>(qo, SP);
#pragma unroll
        for (int e = 0; e < NV; e += 2) {
            const f32x2 X = pr(x, e), LO = pr(lo, e), HI = pr(hi, e);
            const f32x2 mid = (LO + HI) / 2.0f, d = mid - X;
            const f32x2 below = {LO.x > X.x ? 1.0f : 0.0f, LO.y > X.y ? 1.0f : 0.0f}, above = {X.x > HI.x ? 1.0f : 0.0f, X.y > HI.y ? 1.0f : 0.0f};
            unpr(gx, e, -pr(LP, e) * below + pr(HP, e) * above - pr(SP, e) * f32x2{sgnf_(d.x), sgnf_(d.y)});
        }
    }
    template <class OP>
    __device__ __forceinline__ void step(const OP &A, const float (&x)[NV], const float (&u)[NV], int qo,
                                         float (&xn)[NV]) const
    {
        float z[NV], inflow[NV], rain[NV];
#pragma unroll
        for (int e = 0; e < NV; e += 2) { unpr(z, e, pr(u, e) * pr(x, e)); inflow[e] = 0.0f; inflow[e + 1] = 0.0f; }
        mat_apply(A, z, inflow);
        row<kRain>(qo, rain);
        float r[NV], sr[NV];
#pragma unroll
        for (int e = 0; e < NV; e += 2) unpr(r, e, pr(x, e) * pr(rcap, e));
        sin_vec<NV>(r, sr);
#pragma unroll
        for (int e = 0; e < NV; e += 2) {
            const f32x2 xi = pr(x, e);
            const f32x2 vaporated = 0.5f * pr(sr, e) * xi;                                // :87
            unpr(xn, e, xi + pr(rain, e) + pr(inflow, e) - vaporated - pr(u, e) * xi);    // :56-60
        }
    }
    // `step` and `adjoint_coefficients` of the same (x, u) in one piece (the stored rollouts of the forms with kSweepCoefficients): ONE evaluation of
    // sine and cosine serves the evaporation and the diagonal coefficient -- the same sin_small / cos_small / general path per argument, so both
    // results carry the bits the separate calls give.
    template <class OP>
    __device__ __forceinline__ void step_with_coefficients(const OP &A, const float (&x)[NV], const float (&u)[NV], int qo, float (&xn)[NV],
                                                           float (&cA)[NV], float (&gx)[NV]) const
    {
        float z[NV], inflow[NV], rain[NV], Dii[NV];
#pragma unroll
        for (int e = 0; e < NV; e += 2) { unpr(z, e, pr(u, e) * pr(x, e)); inflow[e] = 0.0f; inflow[e + 1] = 0.0f; }
        mat_apply(A, z, inflow);
        row<kRain>(qo, rain);
        row<kDii>(qo, Dii);
        grads(x, qo, gx);
        float r[NV], sr[NV], cr[NV];
#pragma unroll
        for (int e = 0; e < NV; e += 2) unpr(r, e, pr(x, e) * pr(rcap, e));
        sincos_vec<NV>(r, sr, cr);
#pragma unroll
        for (int e = 0; e < NV; e += 2) {
            const f32x2 xi = pr(x, e), uj = pr(u, e);
            const f32x2 vaporated = 0.5f * pr(sr, e) * xi;                                // :87
            unpr(xn, e, xi + pr(rain, e) + pr(inflow, e) - vaporated - uj * xi);          // :56-60
            const f32x2 diag_extra = 1.0f - 0.5f * (pr(cr, e) * pr(r, e) + pr(sr, e)) - uj;
            unpr(cA, e, pr(Dii, e) * uj + diag_extra);
        }
    }
    // `adjoint` in two parts.  What does not depend on V_x -- Q_x = u Y + cA V_x + l_x and Q_u = x Y + cB V_x with Y = the coupling product,
    // cA = D_ii u + (1 - (r cos r + sin r) / 2 - u), cB = D_ii x - x: the trigonometry and the cost gradient, ~250 of a sweep step's ~320
    // instructions -- can be evaluated for every step of the horizon at once; what is left of the recursion is four multiply-adds per
    // row pair (and cB, two instructions a pair: not worth a fourth piece of memory traffic).  The multi-wave one-tile groups do exactly that
    // (`kSweepCoefficients` in the kernel); every other form calls `adjoint`.  Same expressions in the same order either way: the same bits.
    __device__ __forceinline__ void adjoint_coefficients(const float (&xh)[NV], const float (&uh)[NV], int qo, float (&cA)[NV],
                                                         float (&gx)[NV]) const
    {
        float Dii[NV];
        row<kDii>(qo, Dii);
        grads(xh, qo, gx);
        float r[NV], sr[NV], cr[NV];
#pragma unroll
        for (int e = 0; e < NV; e += 2) unpr(r, e, pr(xh, e) * pr(rcap, e));
        sincos_vec<NV>(r, sr, cr);
#pragma unroll
        for (int e = 0; e < NV; e += 2) {
            const f32x2 uj = pr(uh, e), D = pr(Dii, e);
            const f32x2 diag_extra = 1.0f - 0.5f * (pr(cr, e) * pr(r, e) + pr(sr, e)) - uj;
            unpr(cA, e, D * uj + diag_extra);
        }
    }
    template <class OP>
    __device__ __forceinline__ void adjoint_apply(const OP &A, const float (&xh)[NV], const float (&uh)[NV], const float (&vx)[NV], int qo,
                                                  const float (&cA)[NV], const float (&gx)[NV], float (&Qx)[NV], float (&Qu)[NV]) const
    {
        float Y[NV], Dii[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) Y[e] = 0.0f;
        mat_apply(A, vx, Y);                                                          // sum_{k != i} D[i][k] V_x[k]
        row<kDii>(qo, Dii);
#pragma unroll
        for (int e = 0; e < NV; e += 2) {
            const f32x2 V = pr(vx, e), Yp = pr(Y, e), xa = pr(xh, e);
            unpr(Qx, e, fma2(pr(uh, e), Yp, fma2(pr(cA, e), V, pr(gx, e))));
            unpr(Qu, e, fma2(xa, Yp, fma2(pr(Dii, e) * xa - xa, V, splat(0.0f))));
        }
    }
    template <class OP>
    __device__ __forceinline__ void adjoint(const OP &A, const float (&xh)[NV], const float (&uh)[NV],
                                            const float (&vx)[NV], int qo, float (&Qx)[NV], float (&Qu)[NV]) const
    {
        float cA[NV], gx[NV];
        adjoint_coefficients(xh, uh, qo, cA, gx);
        adjoint_apply(A, xh, uh, vx, qo, cA, gx, Qx, Qu);
    }
};

// Round 5: Reservoir whose `downstream` matrix the CALLER states to be a chain (TfmpcEnv::coupling_shift = +1: reservoir i drains into
// i + 1, -1: into i - 1 -- every config the reference holds, tests/conftest.py:70-75, res4.config.json:13-18): the coupling products are
// row moves at COMPILE time (shift_apply), so the instantiation carries no operand fragments, no LDS slab for their low-order parts (14.9
// instead of 18.9 KB) and no product code: 11.2 - 11.3 against 11.6 ms on a cfg5 batch (tools/probes/r5_cfg5_chain_ab.sh).  What it was built
// for -- a register budget for THREE waves per SIMD with a two-deep ring (10.9 KB of LDS) -- does not fit: at 168 registers the compiler
// spills 130 (two step sizes per pass: 26.4 ms) or 113 (one: 16.8 ms, and no longer the same bits); profiles/r05_cfg5_budget.md.
// Internal template tag only (genv.kind stays TFMPC_ENV_RESERVOIR); the kernel checks the promise
// against the matrix before it computes anything (TFMPC_ST_ENV_FLAG).  Same arithmetic as the run-time shift path: the same bits.
constexpr int kEnvReservoirChain = 100;
struct MatShift { int shift, leak_mask; };
__device__ __forceinline__ void mat_apply(const MatShift &A, const float (&z)[8], float (&acc)[8]) { shift_apply(A.shift, A.leak_mask, z, acc); }
__device__ __forceinline__ void force_dense(MatShift &, bool) {}
template <int NT, bool LEAN, bool ROWREGS> struct EnvM<kEnvReservoirChain, NT, LEAN, ROWREGS> : EnvM<TFMPC_ENV_RESERVOIR, NT, LEAN, ROWREGS> {
    static_assert(NT == 2, "the one-tile forms keep the run-time test");
    using Operand = MatShift;
    static constexpr bool kChain = true;
#ifndef TFMPC_CHAIN_ALPHAS
#define TFMPC_CHAIN_ALPHAS 2
#endif
    static constexpr int kSearchAlphas = TFMPC_CHAIN_ALPHAS;      // step sizes per line-search pass (see the note at the register budget)
    __device__ __forceinline__ static int leak(int shift, int n, int q)
    {
        if (!(shift < 0 && n < 32)) return 0;
        int mask = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) mask |= (16 * (e >> 2) + 4 * q + (e & 3) == n) ? (1 << e) : 0;
        return __any(mask != 0) ? (mask | 0x100) : 0;
    }
    // forward: D^T (row R receives z[R - dir]); backward: D without its diagonal (row R receives V_x[R + dir])
    template <int PK>
    __device__ __forceinline__ void load_forward(const TfmpcEnv &g, int, int q, Operand &A, u32x4 *) const
    {
        A.shift = -g.coupling_shift;
        A.leak_mask = leak(A.shift, g.n, q);
    }
    template <int PK>
    __device__ __forceinline__ void load_backward(const TfmpcEnv &g, int, int q, Operand &A, u32x4 *) const
    {
        A.shift = g.coupling_shift;
        A.leak_mask = leak(A.shift, g.n, q);
    }
    // is `downstream` the chain the caller promised?  (every lane looks at 16 entries of each 16 x 16 tile of the 32 x 32 block)
    __device__ __forceinline__ static bool promise_holds(const TfmpcEnv &g, int lane)
    {
        const float *D = g.p[7];
        const int n = g.n, i = lane & 15, q = lane >> 4, dir = g.coupling_shift;
        bool ok = dir == 1 || dir == -1;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int R = 16 * a + i, C = 16 * b + 4 * q + r;
                    if (R < n && C < n) ok = ok && D[R * n + C] == ((C == R + dir) ? 1.0f : 0.0f);      // (a zero diagonal too)
                }
        return __all(ok);
    }
};
// Round 6: the same promise in ONE tile (the reference's own res4 config, 64 instances per wave: n = 16 / PK, so that an instance's rows fill
// its lane quarters and no padding row can receive a neighbour's value -- no leak mask; reservoir i drains into i + 1, coupling_shift = +1, which is what
// every config the reference holds is).  The run-time form (MatOp<1>::shift, shift_apply1) pays for its generality on every step of a kernel whose time
// IS one wave's instruction count: a wave-uniform branch around the matrix instructions, the matrix path's seven hazard no-ops in the block both paths
// join, eight selects (direction, leak) -- ~17 of a step's ~150 - 200 issue slots.  Here the direction is a constant of the phase (forward = down,
// backward = up), PK = 4 moves registers only, and the products do not exist.  Same values moved, same `acc + moved` adds: the same bits.
struct MatShift1 { int dir; };
template <int PK>
__device__ __forceinline__ void shift_apply1_const(int dir, const float (&z)[4], float (&acc)[4])
{
    constexpr int QS = 4 / PK;                       // lane quarters per instance
    const bool down = dir < 0;                       // (a compile-time constant wherever this is inlined: load_forward / load_backward set literals)
    float t = 0.0f;
    if constexpr (QS > 1) {
        const int lane = lane_id(), q = lane >> 4;
        float z_first = z[0], z_last = z[3];
        asm volatile("" : "+v"(z_first), "+v"(z_last));       // (see shift_apply1)
        t = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((down ? lane - 16 : lane + 16) & 63) << 2,
                                                                   __builtin_bit_cast(int, down ? z_last : z_first)));
        t = (q % QS == (down ? 0 : QS - 1)) ? 0.0f : t;
    }
    if (down) { acc[0] += t; acc[1] += z[0]; acc[2] += z[1]; acc[3] += z[2]; }
    else { acc[0] += z[1]; acc[1] += z[2]; acc[2] += z[3]; acc[3] += t; }
}
template <int PK> struct MatShift1Of : MatShift1 {};
template <int PK>
__device__ __forceinline__ void mat_apply(const MatShift1Of<PK> &A, const float (&z)[4], float (&acc)[4]) { shift_apply1_const<PK>(A.dir, z, acc); }
template <int PK> __device__ __forceinline__ void force_dense(MatShift1Of<PK> &, bool) {}
template <bool LEAN, bool ROWREGS, int PK> struct EnvChain1 : EnvM<TFMPC_ENV_RESERVOIR, 1, LEAN, ROWREGS> {
    using Operand = MatShift1Of<PK>;
    static constexpr bool kChain = true;
    template <int PK_>
    __device__ __forceinline__ void load_forward(const TfmpcEnv &, int, int, Operand &A, u32x4 *) const { A.dir = -1; }     // D^T: row R receives z[R - 1]
    template <int PK_>
    __device__ __forceinline__ void load_backward(const TfmpcEnv &, int, int, Operand &A, u32x4 *) const { A.dir = 1; }     // D without its diagonal: V_x[R + 1]
    // is `downstream` the chain i -> i + 1 on exactly n = 16 / PK reservoirs?  (every lane tests all entries: n <= 16, at most 256 loads once per kernel... n = 4: 16)
    __device__ __forceinline__ static bool promise_holds(const TfmpcEnv &g, int)
    {
        const float *D = g.p[7];
        const int n = g.n;
        bool ok = g.coupling_shift == 1 && n == 16 / PK;
        if (ok)
            for (int R = 0; R < n; ++R)
                for (int C = 0; C < n; ++C) ok = ok && D[R * n + C] == ((C == R + 1) ? 1.0f : 0.0f);      // (a zero diagonal too)
        return __all(ok);
    }
};
constexpr int kEnvReservoirChain1 = 101;             // internal tag: EnvChain1 (the PK is the kernel's)
template <int KIND, int NT, bool LEAN> struct EnvTraits { static constexpr bool kChain = false; };
template <bool LEAN> struct EnvTraits<kEnvReservoirChain1, 1, LEAN> { static constexpr bool kChain = true; };
template <int NT, bool LEAN> struct EnvTraits<kEnvReservoirChain, NT, LEAN> { static constexpr bool kChain = true; };

#ifdef TFMPC_CFG5_TRACE
// probe builds only (tools/probes/cfg5_trace.py): per instance and sweep, the accepted step-size index and, for every
// step size tried, the first time step at which the partial cost was above J_hat.  [B][16][12] ints.
__device__ int *g_cfg5_trace = nullptr;
#endif

#ifdef TFMPC_PHASE_PROBE
__device__ unsigned long long *g_cfg5_phases = nullptr;
#define TFMPC_PHASE_BEGIN() const unsigned long long phase_t0_ = __builtin_amdgcn_s_memtime()
#define TFMPC_PHASE_END(i) do { __builtin_amdgcn_s_waitcnt(0); phase_acc[i] += __builtin_amdgcn_s_memtime() - phase_t0_; ++phase_cnt[i]; } while (0)
#else
#define TFMPC_PHASE_BEGIN() do {} while (0)
#define TFMPC_PHASE_END(i) do {} while (0)
#endif

// Multi-wave groups (round 3): every line-search chain leaves the states it passes through at the segment boundaries of the
// horizon (kMaxGroupWaves - 1 of them) in the workspace, one tile set each, fp32; the stored rollout of the accepted step
// size then runs as one segment per wave, each from its checkpoint (see the kernel).
constexpr int kMaxGroupWaves = 8;
__host__ __device__ constexpr size_t adjoint_mfma_checkpoint_floats(int NT) { return (size_t)kMaxGroupWaves * (kMaxGroupWaves - 1) * NT * kTileElems; }
// the sweep coefficients of a one-tile group (Reservoir, `kSweepCoefficients` in the kernel): [time step][cA | l_x | 1 / (|u| + 1)][lane]
// 16-byte pieces behind the checkpoint tiles
constexpr int kCoefPieces = 3;
// (the slab used to sit in EVERY group's slice of every one-tile shape -- + 315 MB at B = 65 536, n = m = 4 that no kernel touched.
// Only the four- / eight-wave forms read it, and those exist for at most kCoefGroups groups: it now sits once behind the groups' slices.)
constexpr int kCoefGroups = 512;
__host__ __device__ constexpr size_t adjoint_mfma_coefficient_floats(int NT, int T) { return NT == 1 ? (size_t)T * kCoefPieces * kTileElems : 0; }
__host__ __device__ constexpr size_t adjoint_mfma_coefficient_bytes(int NT, int T) { return (adjoint_mfma_coefficient_floats(NT, T) * sizeof(float) + 255) & ~(size_t)255; }
// bytes of one wave's slice of the wave-major workspace (sized for fp32 containers; the 16-bit ones use half of each buffer)
__host__ __device__ constexpr size_t adjoint_mfma_wave_bytes(int NT, int T)
{
    // ... + one time step's worth of "nowhere" (kTrashFloats): the stored rollout issues EVERY store on every step -- lanes
    // of columns that do not keep the candidate write there -- so that the number of vector-memory operations of a step
    // is a constant the LDS-DMA ring's s_waitcnt vmcnt(N) can count (see `rollout`)
    return (2 * ((size_t)(T + 1) * NT * kTileElems + (size_t)T * NT * kTileElems + (size_t)(T + 1) * kCostLd) * sizeof(float) +
            (size_t)T * kWave + 255 + ((size_t)NT * kTileElems + kCostLd) * sizeof(float) +
            adjoint_mfma_checkpoint_floats(NT) * sizeof(float)) & ~(size_t)255;
}

// ---- round 3: groups of NW waves, one step-size chain per wave ----------------------------------------------------------
// The second step size of a line-search pass used to ride in the same wave: two chains per lane.  An in-order wave does
// not overlap them -- tools/probes/matvec_bf16x3_probe.hip: the split + 12-MFMA block of a chain-step takes 448 cycles
// for one chain and 880 for two in one wave -- so the second chain only pays off through the shared loads and the
// clip's shared half.  With NW = 2 the sixteen columns of a group are shared by a WORKGROUP of two waves: wave w rolls out
// step size NW p + w of pass p (one chain per lane), stops on its own once all of ITS columns are above J_hat, and the
// two meet at a barrier per pass to exchange J through LDS; the state machine runs replicated in both waves on the same
// inputs.  The costate sweep and the start rollout run on wave 0 (their result reaches the other waves through HBM / the
// L1 of the same CU behind the barrier).  NW = 4, 8 (fp32 containers): four / eight step sizes per pass, and the STORED
// rollout of the accepted step size as one segment of the horizon per wave, each from the state the accepted chain left
// at that segment's start (`ckpt`; see `rollout`) -- the same bits as the one-piece rollout.
// Where it pays: launches that leave SIMDs idle (<= 1 024 groups: small batches, the packed tiny envs, one instance) -- the
// form that brings the launch to about two waves per SIMD wins (ilqr_adjoint_mfma_launch lists the timings).  With a group per SIMD or more the one-wave
// form is the faster one: the second wave's registers and LDS ring cost residency (cfg5: 13.4 -> 19 ms).  The verdict
// of round 2 had asked for an eight-columns-per-wave form to reach four waves per SIMD at cfg5: this IS that split (half
// the element-wise state per wave) without wasting half of every matrix instruction, and the measurement says no:
// at 128 registers the costate sweep spills ~120-330 registers (31.8 / 41.6 ms), at the register count it wants
// two groups no longer fit a SIMD.
// issue priority of this wave among the waves of its SIMD (s_setprio takes an immediate)
__device__ __forceinline__ void set_priority(int p)
{
    if (p >= 3) __builtin_amdgcn_s_setprio(3);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}
template <int NW>
__device__ __forceinline__ void group_sync()
{
    if (NW == 1) wave_sync();
    else __syncthreads();
}

template <int KIND, int NT, int VW, int PK, bool BF16 = false, int NW = 1>
#ifndef TFMPC_GROUP_EU
#define TFMPC_GROUP_EU 4               // A/B builds: the register target (waves per SIMD) of the multi-wave one-tile forms
#endif
#ifndef TFMPC_TWO_TILE_EU
#define TFMPC_TWO_TILE_EU(KIND_) 2     // the register target of the one-wave two-tile forms (cfg5); A/B builds override (round 5: THREE was tried on the
                                       // Reservoir chain form -- 168 registers + 113 .. 130 spilled: 16.8 .. 26.4 ms against 11.7, profiles/r05_cfg5_budget.md)
#endif
#define TFMPC_AM_EU (NW > 1 && NT == 1 ? (NW > 2 ? TFMPC_GROUP_EU : 4) : (NT == 1 && PK == 1 ? 3 : (NT == 2 && NW == 1 ? TFMPC_TWO_TILE_EU(KIND) : 2)))
__global__ __launch_bounds__(kWave * NW) __attribute__((amdgpu_waves_per_eu(TFMPC_AM_EU, TFMPC_AM_EU))) void ilqr_adjoint_mfma_kernel(TfmpcEnv genv, TfmpcIlqrConfig cfg, AdjointSolveArgs a)
{
    constexpr int NV = 4 * NT;
    using TT = typename std::conditional<BF16, bf16_t, float>::type;      // element type of the trajectories in HBM
    static_assert(PK == 1 || NT == 1, "instances are packed into ONE tile");
    static_assert(NW == 1 || NW == 2 || NW == 4 || NW == 8, "waves per sixteen-column group");
    static_assert(NW <= kMaxGroupWaves, "checkpoint tiles");
    const int lane = lane_id(), j = lane & 15, q = lane >> 4, n = genv.n, m = n, T = a.T;
    // this wave within its group.  The LOGICAL wave 0 -- which runs the start rollout and the costate recursion alone -- is a different hardware wave in
    // every other round of 256 groups: two groups that share a CU then do not put their lone phases on the same SIMD (A/B: TFMPC_NO_WAVE_ROTATION)
#ifdef TFMPC_NO_WAVE_ROTATION
    const int wave_rot = 0;
#else
    const int wave_rot = NW > 1 ? ((int)(blockIdx.x >> 8) * (NW / 2)) & (NW - 1) : 0;
#endif
    const int wv = NW == 1 ? 0 : (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) + wave_rot) & (NW - 1);
    // PK instances per column (n <= 16 / PK): lane quarter q belongs to sub-instance q / (4 / PK) and holds its rows
    // 4 ql .. 4 ql + 3, ql = q mod (4 / PK).  Everything below indexes rows with ql; only the MFMA operands know q.
    const int ql = q & (4 / PK - 1);
    // position of this lane's 16-byte piece inside a wave-major tile: COLUMN-major, so that the 64 bytes of one column
    // (instance; 128 with two tiles) are contiguous -- a store masked by column (a line-search pass that only some columns still try) then
    // writes whole 64-byte sectors instead of 16-byte pieces 256 bytes apart
    const int wl = j * NT * 4 + q;                      // [column][tile][lane quarter]: + 4 per tile (ldw / stw)
    const int b_raw = (blockIdx.x * kCols + j) * PK + q / (4 / PK);
    const bool live = b_raw < a.B;                       // the last group may carry empty columns: they compute on the
    const size_t b = live ? b_raw : a.B - 1;             // last instance's data and store nothing
    __shared__ __attribute__((aligned(16))) float rows[kRowSlots * kRowLd];
    constexpr bool kChain = EnvTraits<KIND, NT, (NW > 1)>::kChain;       // Reservoir chain: no operand at all (see EnvM<kEnvReservoirChain>)
#ifdef TFMPC_NO_ROW_REGISTERS          // A/B builds
    constexpr bool kRowRegs = false;
#else
    constexpr bool kRowRegs = NT == 1 && NW >= 4 && !BF16;              // the env's constant rows and the action bounds in registers (see EnvM)
#endif
    using Env = typename std::conditional<KIND == kEnvReservoirChain1, EnvChain1<(NW > 1), kRowRegs, PK>, EnvM<KIND == kEnvReservoirChain1 ? TFMPC_ENV_RESERVOIR : KIND, NT, (NW > 1), kRowRegs>>::type;
    __shared__ u32x4 op_rest_all[NW][NT == 2 && !kChain ? 4 * kWave : 1];  // MatOp<2, true>: the non-leading parts of the operand, per wave
    __shared__ float x_sweep[NW > 1 ? 4 : 1][kWave];           // wave 0 -> the others: J_hat, dV1, g_norm, max |k|
    __shared__ float x_pass[2][NW][2][kWave];                  // [pass parity][wave][J | cut short][lane]
    // Round 6, groups of eight waves: a chain GIVES UP once it cannot matter.  The search takes, per column, the FIRST step size (in the
    // reference's order, ilqr.py:323-353) whose rollout is accepted; wave w rolls out position w of the pass.  So when every wave below w has finished
    // and every trying column was accepted by one of them, position w decides nothing -- and on a chip where the chains of a pass share SIMDs in pairs
    // (a res4 pass: positions 0 - 4 take 17 - 83 k cycles, 5 - 7 next to a partner 103 - 110 k, and the group waits for the slowest) the pass is over
    // when the positions that CAN matter are (three of res4's first five iterations accept by position 2 in every column of a group).  Each wave posts,
    // when its rollout ends, which trying columns it accepts (`give_acc`, a lane mask) and then its bit in `give_posted`; a chain polls that word every
    // fourth step (one LDS read, consumed at the NEXT poll) and stops -- reporting "cut short", as the cost-based early exit does -- once the waves
    // 0 .. p have ALL posted for some p below it and their masks cover the trying columns (it does not wait for the waves between p and itself: they
    // give up on the same evidence).  No wave ever WAITS on another here (a poll that fails changes nothing), and what a given-up chain would have
    // reported is read by no column: the results are the same bits.  The chains of the earlier positions are given issue priority (s_setprio) over
    // their SIMD partners (measured neutral; kept for the ordering it suggests).  Slots alternate with the pass parity; a wave clears its bit of the
    // OTHER slot when it posts (nobody reads that slot between the two barriers around this pass).
#ifdef TFMPC_AB_NO_GIVE_UP             // A/B builds
    constexpr bool kGiveUp = false;
#else
    constexpr bool kGiveUp = NW == 8 && !BF16;       // (four-wave groups: one chain per SIMD and group -- measured no gain on hvac6, and the poll is ~4 issue slots a step)
#endif
    __shared__ unsigned long long give_acc[2][kGiveUp ? NW : 1];
    __shared__ unsigned give_posted[2];
    if (kGiveUp && threadIdx.x < 2) give_posted[threadIdx.x] = 0u;      // (visible to the group behind the barrier below)
    u32x4 *const op_rest = op_rest_all[wv];
    // the LDS-DMA input ring of this wave (fp32 containers): [slot][x tiles | u tiles][lane] 16-byte pieces, the stage
    // cost and the selector byte of a slot.  Depth 3 with two tiles keeps 8 groups per CU inside the 160 KB.
    constexpr bool kLdsRing = !BF16;
    // Round 5, Reservoir in one tile, groups of four / eight waves: the costate sweep in two parts (EnvM::adjoint_coefficients).  A group's iteration
    // is a chain of dependent 100-step phases on a chip with idle SIMDs, and the sweep was the one phase still running on ONE wave: ~320
    // instructions a step at one wave's issue rate (1 500 cycles; 44 % of a res4 solve, profiles/r03_small_env_phase_split.txt).  What does not
    // depend on V_x -- the trigonometry, l_x, the diagonal coefficient that carries them, 1 / (|u| + 1) of the stationarity measure -- is a function of
    // the nominal trajectory alone, and every nominal trajectory comes out of a STORED rollout (the start rollout; the accepted step size rolled out
    // as one segment of the horizon per wave): that rollout, which has x_t, u_t and the sine of the evaporation in registers, writes the three
    // 16-byte pieces per lane next to the trajectory, and wave 0 runs the recursion proper on them (the LDS-DMA ring carries them with x_hat, u_hat).
    // Same expressions, same order: same bits.  (First built as a pass of its own in front of the sweep, all waves on alternate time steps: 27 k cycles
    // per iteration, bound by the memory system -- 256 groups wrote and re-read at once; in the stored rollout it is ~90 instruction slots a step.)
#ifdef TFMPC_NO_SWEEP_COEFFICIENTS     // A/B builds
    constexpr bool kSweepCoefficients = false;
#else
    constexpr bool kSweepCoefficients = (KIND == TFMPC_ENV_RESERVOIR || KIND == kEnvReservoirChain1) && NT == 1 && NW >= 4 && kLdsRing;
#endif
#ifndef TFMPC_CHAIN_RING
#define TFMPC_CHAIN_RING 3
#endif
    constexpr int kRingDepth = NT == 2 ? (kChain ? TFMPC_CHAIN_RING : 3) : 4;
    __shared__ f32x4 ring_v_all[NW][kLdsRing ? kRingDepth : 1][2 * NT][kWave];
    __shared__ float ring_c_all[NW][kLdsRing ? kRingDepth : 1][kWave];
    __shared__ unsigned ring_k_all[NW][kLdsRing ? kRingDepth : 1][kWave];    // (a sub-dword LDS-DMA still strides the lanes by 4 bytes)
    __shared__ f32x4 ring_e[kSweepCoefficients ? kRingDepth * kCoefPieces : 1][kWave];      // wave 0's ring of the sweep coefficients: [slot][piece][lane]
    auto ring_next = [](int slot_) {                    // (scalar arithmetic; a power-of-two depth wraps with a mask)
        if constexpr ((kRingDepth & (kRingDepth - 1)) == 0) return (slot_ + 1) & (kRingDepth - 1);
        else return slot_ + 1 == kRingDepth ? 0 : slot_ + 1;
    };
    f32x4 (*const ring_v)[2 * NT][kWave] = ring_v_all[wv];
    float (*const ring_c)[kWave] = ring_c_all[wv];
    unsigned (*const ring_k)[kWave] = ring_k_all[wv];
    Env env;
    env.load(genv, lane, ql, rows);                      // (every wave of the group writes the same values)
    if constexpr (kChain) {
        // the caller's promise (TfmpcEnv::coupling_shift) against the matrix itself, before anything is computed on it
        if (!Env::promise_holds(genv, lane)) {                  // (wave-uniform)
            if (wv == 0 && ql == 0 && live) { a.iterations[b] = 0; a.status[b] = TFMPC_ST_ENV_FLAG; }
            return;
        }
    }
    if (lane < kRowLd) {
        rows[kSlotALow * kRowLd + lane] = (lane < m) ? genv.low[lane] : 0.0f;
        rows[kSlotAHigh * kRowLd + lane] = (lane < m) ? genv.high[lane] : 0.0f;
    }
    group_sync<NW>();
    env.cache_rows(ql);
    // kRowRegs: both matrix operands stay resident too (one tile: four registers each).  Loaded per phase -- as the two-tile forms must, for their
    // registers -- a phase starts with the operand's round trip to L2 (HVAC: a dependent chain of n loads for the diagonal): 2 - 5 k cycles, three
    // phases per iteration of a group whose whole iteration is ~300 k.
    typename Env::Operand A_fwd, A_bwd;
    if constexpr (kRowRegs) {
        env.template load_forward<PK>(genv, j, q, A_fwd, op_rest);
        force_dense(A_fwd, a.dense_coupling != 0);
        env.template load_backward<PK>(genv, j, q, A_bwd, op_rest);
        force_dense(A_bwd, a.dense_coupling != 0);
    }
    float alow_c[NV], ahigh_c[NV];                       // kRowRegs: the action bounds of this lane's rows, read once
    if constexpr (kRowRegs) {
        lds_rows<NT>(rows, kSlotALow, ql, alow_c);
        lds_rows<NT>(rows, kSlotAHigh, ql, ahigh_c);
    }
    auto bounds = [&](int qo, float (&alow)[NV], float (&ahigh)[NV]) {
        if constexpr (kRowRegs) {
#pragma unroll
            for (int e = 0; e < NV; ++e) { alow[e] = alow_c[e]; ahigh[e] = ahigh_c[e]; }
        } else {
            lds_rows<NT>(rows, kSlotALow, qo, alow);
            lds_rows<NT>(rows, kSlotAHigh, qo, ahigh);
        }
    };
#ifdef TFMPC_STAGGER
    // A/B builds: every other group starts late, so that the store-heavy phase of one half of the chip (the stored rollout
    // of a full cfg5 batch asks for ~6 TB/s when all groups run it at once) meets the line-search passes of the other
    if (blockIdx.x & 1)
        for (int i = 0; i < TFMPC_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif
    const bool early_exit = cfg.c1 == 0.0f && env.costs_nonnegative();     // see `rollout`
    const bool fused_cost = Env::kFusedSearchCost && env.bounds_ordered();

    // trajectories of this group, wave-major (see ldw): two buffers of states / actions / stage costs in the group's slice
    // of the workspace, the nominal one is [flip]; at the end the nominal trajectory is copied (16-bit containers:
    // widened) into the instance-major output arrays.  Then the selector bytes [t][column][lane quarter]: 4 NT bits each.
    const size_t kXs = (size_t)(T + 1) * NT * kTileElems, kUs = (size_t)T * NT * kTileElems, kCs = (size_t)(T + 1) * kCostLd;
    unsigned char *const wave_ws = static_cast<unsigned char *>(a.wave_ws) + (size_t)blockIdx.x * adjoint_mfma_wave_bytes(NT, T);
    TT *xbuf[2], *ubuf[2], *cbuf[2];
    const int ccol = j * PK + q / (4 / PK);             // this lane's instance within the group
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        xbuf[i] = reinterpret_cast<TT *>(wave_ws + (size_t)i * (kXs + kUs + kCs) * sizeof(float));
        ubuf[i] = xbuf[i] + kXs;
        cbuf[i] = ubuf[i] + kUs + ccol;
    }
    unsigned char *const ksel = wave_ws + 2 * (kXs + kUs + kCs) * sizeof(float) + (j * 4 + q);     // [t][column][lane quarter]
    // the trash slot behind the selector bytes (16-byte aligned): one tile set + one cost row
    TT *const trash = reinterpret_cast<TT *>(wave_ws + ((2 * (kXs + kUs + kCs) * sizeof(float) + (size_t)T * kWave + 15) & ~(size_t)15));
    // the checkpoint tiles behind the trash slot: [chain of the pass][segment boundary - 1][tile][lane] 16-byte pieces, fp32
    float *const ckpt = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(trash) + ((size_t)NT * kTileElems + kCostLd) * sizeof(float));
    // the sweep coefficients of this group, [t][piece][lane] 16-byte pieces: behind the slices of ALL groups (launches of <= kCoefGroups groups only;
    // the launcher keeps the forms that use them to such launches)
    float *const coef = reinterpret_cast<float *>(static_cast<unsigned char *>(a.wave_ws) + (size_t)gridDim.x * adjoint_mfma_wave_bytes(NT, T) +
                                                   (size_t)blockIdx.x * adjoint_mfma_coefficient_bytes(NT, T));
    const int t_seg = NW > 1 ? (T + NW - 1) / NW : T;    // segment w of a stored rollout: steps [w t_seg, min((w + 1) t_seg, T))
    const float *const x0p = a.x0 + b * n;
    int flip = 0;
    // x_0 resident as well (check the code object for scratch_ instructions after touching this: see shift_apply1's note on indexed loads)
    constexpr bool kResidentX0 = kRowRegs;
    float x0_c[NV];
    if constexpr (kResidentX0) ldv<NT, VW>(x0p, n, ql, x0_c);

    // One rollout of every column from x0.  SEARCH: u_t = clip(u_hat_t + alpha k_t) (ilqr.py:193-197), else the
    // injected start actions (:53-82).  The bang-bang gain is k_t = bound - u_hat_t (:140-141), so the sweep leaves ONE
    // BIT per action (which bound) and the rollout rebuilds k_t from u_hat_t with the sweep's own expression.
    // The inputs of step t + kAhead are requested before step t is computed (a register ring, the time loop unrolled by kAhead).
    // (two tiles: the ring is 8 registers per step ahead; two-wave groups have 128 registers and four waves per SIMD to hide a load behind)
    constexpr int kAheadRoll = NW > 1 ? (NT == 2 ? 1 : 2) : (NT == 2 ? 2 : 4), kAhead = NW > 1 && NT == 2 ? 1 : 2;        // rollouts / costate sweep
    // STORE: the trajectory is written (rows of columns with `keep`).  The line search only needs J: its rollouts store
    // nothing, and the one step size a column settles on is rolled out again with STORE (same arithmetic, same bits)
    // -- every speculative rollout writing its 25 KB per instance made the solve HBM-write-bound.
    // NA = 2 (one-wave groups): TWO step sizes in one pass -- the inputs u_hat_t and the selector byte are loaded once.
    // A line-search rollout that stores nothing exists to answer "is J(alpha) <= J_hat?" (ilqr.py:339-353 with c1 = 0:
    // z >= 0 <=> J_hat - J >= 0 on either branch of :342-346).  Stage costs are >= 0 on both envs, so the partial sum only
    // grows: once it is above `reject_above` (= J_hat) in every column that is still trying (`trying`), the answer is
    // "no" whatever follows, and the pass stops -- the first, too long step sizes of a Reservoir search blow the cost
    // up within 5 .. 40 of the 100 steps.
    // Round 3: such a rollout does not form the stage cost of a step at all.  Every lane adds the costs of ITS rows into two
    // running pairs (`jacc`; HVAC: fused multiply-adds straight from the penalty terms, EnvM::cost_accumulate), and the
    // sum over a column's lanes is taken when the pass ends or tests its early exit (once per kAheadRoll steps) -- the
    // per-step column reduction (two lane exchanges, ~20 instructions and their hazard no-ops per chain-step) is gone.  J(alpha) is
    // then summed in a different order than the stored pass and the wave kernels sum theirs (a decision can differ only
    // where |J - J_hat| is at rounding level, ~1e-7 relative); the trajectories a pass stores are computed as before.
#ifdef TFMPC_CFG5_TRACE
    int trace_fa[4] = {-1, -1, -1, -1};
#endif
    auto rollout = [&](auto search, auto store, auto n_alpha, const float (&alpha)[decltype(n_alpha)::value], auto uh,
                       bool keep, TT *xs, TT *us, TT *cs, float (&J_out)[decltype(n_alpha)::value],
                       bool trying = false, float reject_above = 0.0f, bool may_stop = false, bool *stopped_out = nullptr,
                       float *ck_out = nullptr, int t_lo = 0, int t_hi_ = -1, const float *ck_in = nullptr, int give_slot = -1) {
        // ck_out (line-search chains of a multi-wave group): the state at every segment boundary goes to the workspace, columns
        // that are trying only.  [t_lo, t_hi) + ck_in (stored rollout of a multi-wave group): ONE segment of the horizon, from
        // the accepted chain's checkpoint -- the state the one-piece rollout reaches there, bit for bit.
        const int t_hi = t_hi_ < 0 ? T : t_hi_;
        constexpr bool SEARCH = decltype(search)::value, STORE = decltype(store)::value;
        constexpr int NA = decltype(n_alpha)::value;
        constexpr bool DEFER = SEARCH && !STORE;            // only J matters: lane-partial cost sums
        static_assert(!STORE || NA == 1, "only a single rollout is stored");
        typename Env::Operand A;
        env.fence();
        if constexpr (kRowRegs) {
            A = A_fwd;
        } else {
            env.template load_forward<PK>(genv, opaque(j), opaque(q), A, op_rest);
            force_dense(A, a.dense_coupling != 0);
        }
        constexpr bool RING = SEARCH && kLdsRing;           // wave-major fp32 inputs: the LDS-DMA ring; else a register ring
        constexpr int kSlots = RING ? 1 : kAheadRoll, kLoads = NT + 1;      // kLoads: DMA instructions per step
        float x[NA][NV], ur[kSlots][NV], J[NA];
        f32x2 jacc[NA][2];
        unsigned kb[kSlots];
        if constexpr (kResidentX0) {                         // (one round trip to L2 less per rollout)
#pragma unroll
            for (int e = 0; e < NV; ++e) x[0][e] = x0_c[e];
        } else {
            ldv<NT, VW>(x0p, n, ql, x[0]);
        }
        float xck[kRowRegs ? NV : 1];
        if constexpr (kRowRegs) {
            if (STORE && t_lo > 0) ldw<NT>(ck_in, 0, wl, xck);   // (requested here; used in `adopt_start`, behind the ring's first DMAs)
        }
        int ck_next = t_seg, ck_slot = 0;                   // ck_out: the next boundary and its tile
        // The start state of this rollout is USED before the time loop: the compiler then waits for its loads here.  Left to the first use inside
        // the loop, the wait it places there is an s_waitcnt vmcnt(0) on every trip -- it cannot count the ring's loads, which it does not see --
        // and drains the whole ring on every second step (found in the multi-wave search loops, round 5: tools/probes/vmwaits.py).  The forms with
        // registers to spare (kRowRegs) do it BEHIND the ring's first DMAs, so that the two round trips overlap.
        auto adopt_start = [&]() {
            if (STORE && t_lo > 0) {
                if constexpr (kRowRegs) {
#pragma unroll
                    for (int e = 0; e < NV; ++e) x[0][e] = keep ? xck[e] : x[0][e];   // (lanes without a checkpoint compute on finite values)
                } else {
                    float xc_[NV];
                    ldw<NT>(ck_in, 0, wl, xc_);
#pragma unroll
                    for (int e = 0; e < NV; ++e) x[0][e] = keep ? xc_[e] : x[0][e];
                }
            }
            if (STORE && t_lo == 0) stw<NT>(xs, 0, wl, keep, x[0]);
#pragma unroll
            for (int k = 0; k < NA; ++k) {
                J[k] = 0.0f;
                jacc[k][0] = jacc[k][1] = splat(0.0f);
#pragma unroll
                for (int e = 0; e < NV; ++e) x[k][e] = x[0][e];
            }
#pragma unroll
            for (int k = 0; k < NA; ++k)
#pragma unroll
                for (int e = 0; e < NV; ++e) opaque_f(x[k][e]);
        };
        if constexpr (!kRowRegs) adopt_start();
        // the column's partial cost of a deferred pass: this lane's pairs, then the lanes of the column
        // (the first sum is hidden from the vectoriser, which pairs the two up through four register copies)
        auto partial = [&](int k) { float s0 = jacc[k][0].x + jacc[k][0].y; opaque_f(s0); return tile_sum<PK>(s0 + (jacc[k][1].x + jacc[k][1].y)); };
        auto request = [&](int t, float (&u_)[NV], unsigned &k_) {
            if constexpr (SEARCH) {
                ldw<NT>(uh, t, wl, u_);
                k_ = gld(ksel + (size_t)t * kWave);
            } else {
                ldv<NT, VW>(uh + (size_t)t * m, m, ql, u_);                 // the injected start actions: instance-major
            }
        };
        // LDS-DMA ring: the actions of step t (this lane's 16-byte pieces, as ldw addresses them) and the selector byte
        auto issue = [&](int slot, int t) {
            if constexpr (RING) {
                lds_reads_done();
                const float *src = reinterpret_cast<const float *>(uh) + (size_t)t * NT * kTileElems + 4 * wl;
#pragma unroll
                for (int b = 0; b < NT; ++b) dma16(src + 16 * b, lds_addr_of(&ring_v[slot][b][0]));
                dma1(ksel + (size_t)t * kWave, lds_addr_of(&ring_k[slot][0]));
            }
        };
        // Either ring is refilled UNCONDITIONALLY (the time index clamped to the horizon; the last steps re-read step
        // T - 1): the register ring because behind a condition the compiler merges "old value or loaded value" through
        // copies, the LDS ring because its waits count the loads of the younger steps.
        int slot = 0, pslot = kRingDepth - 1;               // LDS ring: the slot of the current step, of the step before it
        if constexpr (RING) {
#pragma unroll
            for (int d = 0; d < kRingDepth - 1; ++d)
                if (t_hi > t_lo) issue(d, t_lo + d < t_hi ? t_lo + d : t_hi - 1);
        } else {
#pragma unroll
            for (int d = 0; d < kAheadRoll; ++d) {
                kb[d] = 0;
#pragma unroll
                for (int e = 0; e < NV; ++e) ur[d][e] = 0.0f;
                if (T > 0) request(d < T ? d : T - 1, ur[d], kb[d]);
            }
        }
        if constexpr (kRowRegs) adopt_start();
        bool stopped = false;
#ifdef TFMPC_CFG5_TRACE
        trace_fa[0] = trace_fa[1] = trace_fa[2] = trace_fa[3] = -1;
#endif
        // (give-up, see `kGiveUp`) the trying columns as a lane mask, and the bits of the waves below this one
        unsigned long long trying_mask = 0;
        if constexpr (kGiveUp && DEFER) trying_mask = __builtin_amdgcn_ballot_w64(trying);
        unsigned give_seen = 0u;                             // (nothing posted yet)
        int give_run = 0;                                    // waves 0 .. give_run - 1 have posted and are in give_cov
        unsigned long long give_cov = 0;
        // The time loop, once per form of the deferred stage cost: `fused_cost` (HVAC with ordered comfort bounds) is a run-time fact, and tested inside
        // the step it was two taken branches and three scalar instructions on every step of every chain (round 6: hoisted -- the loop exists twice
        // where an env has a fused form at all, and the test runs once per rollout).
        auto time_loop = [&](auto fused_c) {
            for (int t0 = t_lo; t0 < t_hi; t0 += kAheadRoll) {
                if constexpr (kGiveUp && DEFER) {
                    if (give_slot >= 0 && wv > 0 && early_exit && may_stop && (t0 & 3) == 0) {                  // (wave-uniform)
                        // (atomic accessors on the __shared__ objects themselves: ds_* instructions, in order within a wave; a volatile access through a cast
                        // pointer would be a FLAT access -- another queue than the ds_or that posts the bit)
                        // The word tested is the one READ AT THE POLL BEFORE (`give_seen`): the read issued now is used four steps on, so no step waits for the
                        // LDS round trip (tested synchronously the poll cost ~300 cycles -- a tenth of a pass on the chains that matter); a give-up is decided
                        // at most one poll late.  The masks are read when the bits are complete, hence after them.
                        const unsigned posted = __builtin_amdgcn_readfirstlane(give_seen);
                        give_seen = __hip_atomic_load(&give_posted[give_slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        // the waves 0 .. L - 1 have all posted (L: the run of set bits from bit 0, capped at this wave): their masks, each read once
                        int run = __builtin_ctz(~posted);
                        run = run < wv ? run : wv;
                        if (give_run < run) {
                            for (; give_run < run; ++give_run) give_cov |= __hip_atomic_load(&give_acc[give_slot][give_run], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            // (wave-uniform: into scalar registers.  readfirstlane returns a SIGNED int -- widened without the cast it smears bit 31 over the upper half)
                            give_cov = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(give_cov >> 32)) << 32) |
                                       (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)give_cov);
                        }
                        const bool decided = give_run > 0 && (trying_mask & ~give_cov) == 0;      // every trying column is accepted by a position below this one
    #ifdef TFMPC_AB_GIVE_UP_DRY            // probe builds: the whole protocol without the stop
                        if (decided) asm volatile("s_nop 0");
    #else
                        if (decided) { stopped = true; break; }
    #endif
                    }
                }
                // (tested every FOURTH step: the column sums behind `partial` are two lane swaps and their hazard no-ops per chain -- ~10 issue slots a
                // step of hvac6's chains when tested every second step; a chain that is over stops at most two steps later, which changes nothing
                // but the partial sum a rejected try reports)
#ifdef TFMPC_AB_EXIT_EVERY_TRIP         // A/B builds: every trip of the loop, as before
                constexpr bool kExitEveryTrip = true;
#else
                constexpr bool kExitEveryTrip = kAheadRoll > 2;
#endif
                if (SEARCH && early_exit && may_stop && (kExitEveryTrip || (t0 & 3) == (t_lo & 3))) {
                    bool open = false;                       // a trying column whose partial cost may still end at or below J_hat
    #pragma unroll
                    for (int k = 0; k < NA; ++k) open = open || (trying && !((DEFER ? partial(k) : J[k]) > reject_above));
                    if (__builtin_amdgcn_ballot_w64(open) == 0) { stopped = true; break; }      // (__any materialises an integer per lane first)
                }
    #pragma unroll
                for (int d = 0; d < kAheadRoll; ++d) {
                    const int t = t0 + d;
                    __builtin_amdgcn_sched_barrier(0);          // the unrolled steps are not interleaved (registers)
                    if (t < t_hi) {
                        const int qo = opaque(ql);
                        float u[NA][NV];
                        const int dd = RING ? 0 : d;                // the register ring's slot of this unrolled step
                        constexpr bool kEarlyProduct = NT == 1 && NW > 1 && Env::kProductOfStateOnly;      // (see EnvM<HVAC>::state_product; one-wave groups share their SIMD with two others and have no registers to park it in)
                        float pre[NA][NV];
                        if constexpr (kEarlyProduct) {
    #pragma unroll
                            for (int k = 0; k < NA; ++k) env.state_product(A, x[k], qo, pre[k]);
                        }
                        if (NW > 1 && DEFER && ck_out && t == ck_next) {    // (wave-uniform; uncounted stores only make the ring's waits longer)
    #pragma unroll
                            for (int k = 0; k < NA; ++k) stw<NT>(ck_out + ((size_t)k * (kMaxGroupWaves - 1) + ck_slot) * NT * kTileElems, 0, wl, trying, x[k]);
                            ck_next += t_seg;
                            ++ck_slot;
                        }
                        if constexpr (RING) {
                            // step t + depth - 1 goes into the slot the previous step has just read; then the loads of THIS
                            // step have landed once at most the (depth - 1) younger steps' are outstanding
                            const int ahead = t + kRingDepth - 1;
                            issue(pslot, ahead < t_hi ? ahead : t_hi - 1);
                            // (a stored rollout also issues kStores stores per step, unconditionally: they are younger than the
                            // loads waited for and stay in flight too -- counted without them, every step waited for the previous
                            // step's stores to be acknowledged by the memory system, ~2 us: the stored rollout, one chain, took as
                            // long as a search pass with two)
                            // (the loads of THIS step were issued at the start of step t - (depth - 1), BEFORE that step's stores: the
                            // stores of the last depth - 1 steps are younger and stay in flight; that many exist from step depth - 1 on)
    #ifdef TFMPC_PROBE_NO_RING_READS        // probe builds (timing only, wrong results): what the ring's wait and its LDS reads cost a step
    #pragma unroll
                            for (int e = 0; e < NV; ++e) { ur[0][e] = 0.25f; opaque_f(ur[0][e]); }
                            kb[0] = 5u + (unsigned)t;
    #else
                            if (STORE && t - t_lo >= kRingDepth - 1) wait_vmem<(kRingDepth - 1) * (kLoads + 2 * NT + 1 + (kSweepCoefficients ? kCoefPieces : 0))>();
                            else wait_vmem<(kRingDepth - 1) * kLoads>();
    #pragma unroll
                            for (int b = 0; b < NT; ++b) {
                                const f32x4 v = ring_v[slot][b][lane];
                                ur[0][4 * b] = v[0]; ur[0][4 * b + 1] = v[1]; ur[0][4 * b + 2] = v[2]; ur[0][4 * b + 3] = v[3];
                            }
                            kb[0] = ring_k[slot][lane] & 0xFFu;
    #endif
                            pslot = slot, slot = ring_next(slot);
                        }
                        if (SEARCH) {
                            float alow[NV], ahigh[NV];
                            bounds(qo, alow, ahigh);
                            float bnd[NV];
                            select_bits<NV>(kb[dd], alow, ahigh, bnd);              // the bound the sweep's selector bit names
    #pragma unroll
                            for (int e = 0; e < NV; e += 2) {
                                const f32x2 uh_e = pr(ur[dd], e);
                                const f32x2 bound = pr(bnd, e);
                                const f32x2 kt = bound - uh_e;                                                   // :140-141
    #pragma unroll
                                for (int k = 0; k < NA; ++k) {
                                    const f32x2 un = uh_e + alpha[k] * kt;                                       // :193-194
                                    u[k][e] = __builtin_amdgcn_fmed3f(un.x, alow[e], ahigh[e]);                  // :196-197 (low <= high)
                                    u[k][e + 1] = __builtin_amdgcn_fmed3f(un.y, alow[e + 1], ahigh[e + 1]);
                                }
                            }
                        } else {
    #pragma unroll
                            for (int e = 0; e < NV; ++e) u[0][e] = ur[dd][e];
                        }
                        if constexpr (!RING) {
                            // the refill is issued AFTER the slot's last use (a scheduling fence on either side)
                            __builtin_amdgcn_sched_barrier(0);
                            request(t + kAheadRoll < T ? t + kAheadRoll : T - 1, ur[dd], kb[dd]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
    #pragma unroll
                        for (int k = 0; k < NA; ++k) {
                            float xn[NV];
                            if constexpr (DEFER) {
                                if constexpr (decltype(fused_c)::value) {            // (see `time_loop`)
                                    env.cost_accumulate(x[k], u[k], qo, jacc[k]);
                                } else {
                                    float cp[NV];
                                    env.stage_costs(x[k], u[k], qo, cp);
    #pragma unroll
                                    for (int e = 0; e < NV; e += 2) jacc[k][(e >> 1) & 1] += pr(cp, e);
                                }
                                if constexpr (kEarlyProduct) env.step_with(pre[k], x[k], u[k], xn);
                                else env.step(A, x[k], u[k], qo, xn);
    #ifdef TFMPC_CFG5_TRACE
                                if (trace_fa[k] < 0 && partial(k) > reject_above) trace_fa[k] = t;
    #endif
                            } else {
                                float cp[NV];
                                env.stage_costs(x[k], u[k], qo, cp);
                                const float c = col_sum<NT, PK>(cp);
                                float cA[NV], gx[NV];                               // (kSweepCoefficients: the next sweep's coefficients of step t)
                                if constexpr (kSweepCoefficients) {                 // (Reservoir: no early product)
                                    if constexpr (STORE) env.step_with_coefficients(A, x[k], u[k], qo, xn, cA, gx);
                                    else env.step(A, x[k], u[k], qo, xn);
                                } else {
                                    if constexpr (kEarlyProduct) env.step_with(pre[k], x[k], u[k], xn);
                                    else env.step(A, x[k], u[k], qo, xn);
                                }
                                J[k] += c;
                                if (STORE) {
                                    if constexpr (RING) {
                                        // every store is issued on every step (columns that do not keep the candidate write to
                                        // the trash slot): kStores per step, which the ring's wait counts
                                        stw<NT>(keep ? us : trash, keep ? t : 0, wl, true, u[k]);
                                        stw<NT>(keep ? xs : trash, keep ? t + 1 : 0, wl, true, xn);
                                        stc(keep ? cs + (size_t)t * kCostLd : trash + NT * kTileElems + ccol, c);
                                    } else {
                                        stw<NT>(us, t, wl, keep, u[k]);
                                        stw<NT>(xs, t + 1, wl, keep, xn);
                                        if (keep && ql == 0) stc(cs + (size_t)t * kCostLd, c);
                                    }
                                    if constexpr (kSweepCoefficients) {             // (see kSweepCoefficients)
                                        f32x4 rd;
    #pragma unroll
                                        for (int e = 0; e < NV; ++e) rd[e] = __builtin_amdgcn_rcpf(fabsf(u[k][e]) + 1.0f);      // (see the sweep: |k| / (|u| + 1))
                                        f32x4 *const dst = reinterpret_cast<f32x4 *>(keep ? coef + (size_t)t * kCoefPieces * kTileElems : reinterpret_cast<float *>(trash)) + lane;
                                        if (RING || keep) {                         // (the LDS-DMA ring counts its stores: issued on every step, as above)
                                            gst(dst, f32x4{cA[0], cA[1], cA[2], cA[3]});
                                            gst(keep ? dst + kWave : dst, f32x4{gx[0], gx[1], gx[2], gx[3]});
                                            gst(keep ? dst + 2 * kWave : dst, rd);
                                        }
                                    }
                                }
                            }
    #pragma unroll
                            for (int e = 0; e < NV; ++e) x[k][e] = xn[e];
                        }
                    }
                }
            }
        };
        if constexpr (DEFER && Env::kFusedSearchCost) {
            if (fused_cost) time_loop(std::true_type{});
            else time_loop(std::false_type{});
        } else {
            time_loop(std::false_type{});
        }
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            float cp[NV];
            env.final_costs(x[k], opaque(ql), cp);
            if constexpr (DEFER) {
                // stopped: already above J_hat in every column that asked
                float fl = (cp[0] + cp[1]) + (cp[2] + cp[3]);
                if constexpr (NT == 2) fl += (cp[4] + cp[5]) + (cp[6] + cp[7]);
                const float own = (jacc[k][0].x + jacc[k][0].y) + (jacc[k][1].x + jacc[k][1].y);
                J_out[k] = tile_sum<PK>(stopped ? own : own + fl);
            } else {
                const float fc = col_sum<NT, PK>(cp);
                if (STORE && keep && ql == 0 && t_hi == T) stc(cs + (size_t)T * kCostLd, fc);
                J_out[k] = stopped ? J[k] : J[k] + fc;
            }
        }
        if (stopped_out) *stopped_out = stopped;
        if (STORE) wave_sync();             // costs are written by lane quarter 0 and read by all four in the next sweep
    };
    using one_t = std::integral_constant<int, 1>;

#ifdef TFMPC_PHASE_PROBE
    unsigned long long phase_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, phase_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long kernel_t0 = __builtin_amdgcn_s_memtime();
#endif
    if (wv == 0) {
        TFMPC_PHASE_BEGIN();
        const float a0[1] = {0.0f};
        float J[1];
        // EVERY column stores its start trajectory, the empty ones of the last group too (they run the last instance's
        // data): the sweeps below read every column's nominal buffer, and with several instances per matrix-core column
        // (n <= 8) a NaN left in the workspace by an earlier use would reach the live rows of the same column through
        // the block-diagonal operand's zeros (0 x NaN).  Found by tools/probes/fuzz_costate.py / nan_workspace.py.
        rollout(std::false_type{}, std::true_type{}, one_t{}, a0, a.u_init + b * T * m, true, xbuf[0], ubuf[0], cbuf[0], J);
        TFMPC_PHASE_END(0);
    }

    float mu = 0.0f, delta = 1.0f;
    int status = 0, attempts = 0, iteration = 0, last_index = 0;     // last_index: position of the step size accepted last
    int parity = 0;                                                  // of the exchange buffer x_pass
    bool done = !live || cfg.max_iterations <= 0;
    while (__any(!done)) {
        TT *const xhat = xbuf[flip], *const uhat = ubuf[flip], *const chat = cbuf[flip];
        TT *const xc = xbuf[flip ^ 1], *const uc = ubuf[flip ^ 1], *const cc = cbuf[flip ^ 1];
        // ---- backward (ilqr.py:94-172 on the bang-bang branch): costate recursion, all columns (wave 0) ----
        float rJ = 0.0f, dV1 = 0.0f, g_norm = 0.0f, kmax = 0.0f;
        if (wv == 0) {
            TFMPC_PHASE_BEGIN();
            typename Env::Operand A;
            env.fence();
            if constexpr (kRowRegs) {
                A = A_bwd;
            } else {
                env.template load_backward<PK>(genv, opaque(j), opaque(q), A, op_rest);
                force_dense(A, a.dense_coupling != 0);
            }
            constexpr bool RING = kLdsRing;
            constexpr int kSlots = RING ? 1 : kAhead, kLoads = 2 * NT + 1 + (kSweepCoefficients ? kCoefPieces : 0);
            float vx[NV], xT[NV], p1[NV], ka[NV], xr[kSlots][NV], ur[kSlots][NV], lr[kSlots];
            ldw<NT>(xhat, T, wl, xT);
            rJ = ldc(chat + (size_t)T * kCostLd);                                    // the stage costs of the nominal trajectory are the l_t
            // V_x = l_x^f.  (kRowRegs: x_T and l_T are requested above and first USED behind the ring's first DMAs -- the round trips overlap;
            // used before the time loop either way, see `rollout`.)
            auto adopt_final = [&]() {
                env.grads(xT, opaque(ql), vx);
                opaque_f(rJ);
            };
            if constexpr (!kRowRegs) env.grads(xT, opaque(ql), vx);
#pragma unroll
            for (int e = 0; e < NV; ++e) { p1[e] = 0.0f; ka[e] = 0.0f; }
            float gsum = 0.0f;
            auto request = [&](int t, float (&x_)[NV], float (&u_)[NV], float &l_) {
                ldw<NT>(xhat, t, wl, x_);
                ldw<NT>(uhat, t, wl, u_);
                l_ = ldc(chat + (size_t)t * kCostLd);
            };
            auto issue = [&](int slot, int t) {                // LDS-DMA ring: x_hat_t, u_hat_t, l_t (see `rollout`)
                if constexpr (RING) {
                    lds_reads_done();
                    const float *sx = reinterpret_cast<const float *>(xhat) + (size_t)t * NT * kTileElems + 4 * wl;
                    const float *su = reinterpret_cast<const float *>(uhat) + (size_t)t * NT * kTileElems + 4 * wl;
#pragma unroll
                    for (int b = 0; b < NT; ++b) dma16(sx + 16 * b, lds_addr_of(&ring_v[slot][b][0]));
#pragma unroll
                    for (int b = 0; b < NT; ++b) dma16(su + 16 * b, lds_addr_of(&ring_v[slot][NT + b][0]));
                    dma4(reinterpret_cast<const float *>(chat) + (size_t)t * kCostLd, lds_addr_of(&ring_c[slot][0]));
                    if constexpr (kSweepCoefficients)
                        dma16x3(coef + (size_t)t * kCoefPieces * kTileElems + 4 * lane, lds_addr_of(&ring_e[slot * kCoefPieces][0]));
                }
            };
            int slot = 0, pslot = kRingDepth - 1;
            if constexpr (RING) {
#pragma unroll
                for (int d = 0; d < kRingDepth - 1; ++d)
                    if (T > 0) issue(d, T - 1 - d >= 0 ? T - 1 - d : 0);
            } else {
#pragma unroll
                for (int d = 0; d < kAhead; ++d) {
                    lr[d] = 0.0f;
#pragma unroll
                    for (int e = 0; e < NV; ++e) { xr[d][e] = 0.0f; ur[d][e] = 0.0f; }
                    if (T > 0) request(T - 1 - d >= 0 ? T - 1 - d : 0, xr[d], ur[d], lr[d]);      // (unconditional refills: see `rollout`)
                }
            }
            if constexpr (kRowRegs) adopt_final();
            for (int t0 = T - 1; t0 >= 0; t0 -= kAhead) {
#pragma unroll
                for (int d = 0; d < kAhead; ++d) {
                    const int t = t0 - d;
                    __builtin_amdgcn_sched_barrier(0);      // the unrolled steps are not interleaved (registers)
                    if (t >= 0) {
                        const int qo = opaque(ql);
                        const int dd = RING ? 0 : d;
                        f32x4 ce[kCoefPieces];
                        if constexpr (RING) {
                            const int ahead = t - (kRingDepth - 1);
                            issue(pslot, ahead >= 0 ? ahead : 0);
                            // (the selector store of a step is younger than the loads issued at that step's start: the stores
                            // of the last depth - 1 steps stay in flight -- uncounted, every wait also asked for one load of the NEXT step)
                            if (RING && t <= T - kRingDepth) wait_vmem<(kRingDepth - 1) * (kLoads + 1)>();
                            else wait_vmem<(kRingDepth - 1) * kLoads>();
#pragma unroll
                            for (int b = 0; b < NT; ++b) {
                                const f32x4 vx_ = ring_v[slot][b][lane], vu_ = ring_v[slot][NT + b][lane];
                                xr[0][4 * b] = vx_[0]; xr[0][4 * b + 1] = vx_[1]; xr[0][4 * b + 2] = vx_[2]; xr[0][4 * b + 3] = vx_[3];
                                ur[0][4 * b] = vu_[0]; ur[0][4 * b + 1] = vu_[1]; ur[0][4 * b + 2] = vu_[2]; ur[0][4 * b + 3] = vu_[3];
                            }
                            lr[0] = ring_c[slot][lane];
                            if constexpr (kSweepCoefficients) {
#pragma unroll
                                for (int k = 0; k < kCoefPieces; ++k) ce[k] = ring_e[slot * kCoefPieces + k][lane];
                            }
                            pslot = slot, slot = ring_next(slot);
                        }
                        float xh[NV], uh[NV];
#pragma unroll
                        for (int e = 0; e < NV; ++e) { xh[e] = xr[dd][e]; uh[e] = ur[dd][e]; }
                        const float l = lr[dd];
                        if constexpr (!RING) {
                            __builtin_amdgcn_sched_barrier(0);          // (see `rollout`)
                            request(t - kAhead >= 0 ? t - kAhead : 0, xr[dd], ur[dd], lr[dd]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        float Qx[NV], Qu[NV], gm[NV], alow[NV], ahigh[NV];
                        if constexpr (kSweepCoefficients) {
                            const float cA[NV] = {ce[0][0], ce[0][1], ce[0][2], ce[0][3]}, gx[NV] = {ce[1][0], ce[1][1], ce[1][2], ce[1][3]};
                            env.adjoint_apply(A, xh, uh, vx, qo, cA, gx, Qx, Qu);
                        } else {
                            env.adjoint(A, xh, uh, vx, qo, Qx, Qu);
                        }
                        bounds(qo, alow, ahigh);
                        unsigned sel = 0;
#pragma unroll
                        for (int e = 0; e < NV; e += 2) {
                            const bool lowb0 = Qu[e] >= 0.0f, lowb1 = Qu[e + 1] >= 0.0f;
                            const f32x2 kt = f32x2{lowb0 ? alow[e] : ahigh[e], lowb1 ? alow[e + 1] : ahigh[e + 1]} - pr(uh, e);   // :140-141
                            sel |= (lowb0 ? (1u << e) : 0u) | (lowb1 ? (2u << e) : 0u);
                            unpr(p1, e, fma2(kt, pr(Qu, e), pr(p1, e)));
                            const f32x2 akt = abs2(kt);
                            ka[e] = max_abs(ka[e], kt.x);                                           // (max_raw's note: the running maximum of |k_t|)
                            ka[e + 1] = max_abs(ka[e + 1], kt.y);
                            const f32x2 den = abs2(pr(uh, e)) + 1.0f;
                            // |k| / (|u| + 1) (:243-245) with the hardware reciprocal (1 ulp) instead of an IEEE division
                            // (8 per step were ~130 instructions): g_norm is only ever compared with atol
                            if constexpr (kSweepCoefficients) {
                                gm[e] = akt.x * ce[2][e];
                                gm[e + 1] = akt.y * ce[2][e + 1];
                            } else {
                                gm[e] = akt.x * __builtin_amdgcn_rcpf(den.x);
                                gm[e + 1] = akt.y * __builtin_amdgcn_rcpf(den.y);
                            }
                            vx[e] = Qx[e];                                                          // V_x <- Q_x
                            vx[e + 1] = Qx[e + 1];
                        }
                        if constexpr (RING) {       // issued on every step (the waits count it): a finished column writes to the trash slot
                            gst(done ? reinterpret_cast<unsigned char *>(trash) + lane : ksel + (size_t)t * kWave, (unsigned char)sel);
                        } else {
                            if (!done) gst(ksel + (size_t)t * kWave, (unsigned char)sel);
                        }
                        rJ += l;
                        gsum += col_max<NT, PK>(gm);
                    }
                }
            }
            kmax = col_max<NT, PK>(ka);
            dV1 = col_sum<NT, PK>(p1);
            g_norm = T > 0 ? gsum / (float)T : 0.0f;
            TFMPC_PHASE_END(1);
            if constexpr (NW > 1) { x_sweep[0][lane] = rJ; x_sweep[1][lane] = dV1; x_sweep[2][lane] = g_norm; x_sweep[3][lane] = kmax; }
        }
        if constexpr (NW > 1) {
            __syncthreads();                 // the sweep's selector bytes (HBM) and its four results (LDS) are visible to the group
            if (wv != 0) { rJ = x_sweep[0][lane]; dV1 = x_sweep[1][lane]; g_norm = x_sweep[2][lane]; kmax = x_sweep[3][lane]; }
        }
        const bool converged_g = !done && g_norm < cfg.atol;                   // :243-248
        // ---- line search rounds (ilqr.py:317-355): every searching column tries its next step size(s) ------
        const bool searching = !done && !converged_g;
        bool accept = false;
        float residual = 0.0f, alpha_last = 0.0f;          // alpha_last: the step size of this column's last rollout
        float J_last = 0.0f;                               // ... and its cost, position: the decision trace
        int index_last = -1, chain_last = 0;               // chain_last: ... and which chain of its pass that was (the checkpoints)
        const bool active = !done;
        const float mu_pass = mu, delta_pass = delta;
        const int row_pass = iteration + attempts;
        // step sizes per wave and pass: a four-wave group of two-tile waves rolls out two per wave -- eight per pass, as an eight-wave
        // group does, on the 2 048 wave slots that 512 groups leave (n = 32, B = 8 192: HVAC 5.59 -> 4.89 ms, Reservoir 4.99 -> 4.78; with one
        // tile the second chain costs what the saved passes bring: hvac6 2.46 -> 2.43, n = 16 HVAC 2.26 -> 2.40)
#ifndef TFMPC_GROUP4_ALPHAS
// (round 6, after the instruction trims -- tools/probes/r6_g4_alphas.py, one against two, ms: hvac6 1.851 / 1.809, Reservoir n = 16 2.19 / 2.09; but
// res4 at 512 groups 1.98 / 2.05, Reservoir n = 8 2.37 / 2.89, HVAC n = 12 1.83 / 1.85: two for HVAC with two instances per column and for one-tile
// Reservoir with one)
#define TFMPC_GROUP4_ALPHAS ((NT == 2 || (KIND == TFMPC_ENV_HVAC && PK == 2) || (KIND == TFMPC_ENV_RESERVOIR && PK == 1)) ? 2 : 1)
#endif
        constexpr int NA = NW == 1 ? Env::kSearchAlphas : (NW == 4 ? TFMPC_GROUP4_ALPHAS : 1);
        static_assert(NA * NW <= kMaxGroupWaves || NW == 1, "checkpoint tiles of a pass");
        static_assert(NW == 1 || NA <= 2, "the exchange buffer holds two values per wave");
        constexpr int NAP = NA * NW;                                          // ... per group and pass
        // One step size per pass in a one-wave group: the pass WRITES the candidate of every column that is trying it (a column's buffer is
        // then final the moment the column accepts: no second rollout), and the early stop keeps the passes that are
        // rejected anyway -- most of a Reservoir search -- from writing much.  `complete`: this column's last try ran to
        // the end of the horizon.  (Several step sizes per pass: nothing is written, the accepted one is rolled out again.)
        constexpr bool kStoreWhileSearching = NAP == 1;     // (no env of the current tree searches one step size per pass)
        // ... but not the passes that are all but certain to be rejected: the step size a column accepts rarely moves to
        // an EARLIER position from one iteration to the next (Reservoir, cfg5: never below index 3 of 11; tools/probes/
        // cfg5_trace.py), so the passes before the SMALLEST index any column of the wave accepted last time only answer
        // "J(alpha) <= J_hat?" and write nothing.  A column that accepts in
        // such a pass after all is rolled out again below (`complete` stays false): the results do not depend on the guess.
        int store_from = 0;
        if (kStoreWhileSearching) {
            store_from = cfg.n_alphas;
            for (int v = 0; v < cfg.n_alphas; ++v)
                if (__any(searching && last_index == v)) { store_from = v; break; }
        }
        bool complete = false;
        // multi-wave groups on fp32 containers: the stored rollout runs as one segment of the horizon per wave (see `rollout`)
#ifdef TFMPC_NO_ROLLOUT_SEGMENTS       // A/B builds
        constexpr bool kSegments = false;
#else
        constexpr bool kSegments = kLdsRing && NW > 1;
#endif
        for (int ai = 0; ai < cfg.n_alphas && __any(searching && !accept); ai += NAP) {
            float al[NA], J[NA], Jall[NAP];
            const int mine = ai + wv * NA;                                     // this wave's first step size of the pass
#pragma unroll
            for (int k = 0; k < NA; ++k) al[k] = cfg.alphas[mine + k < cfg.n_alphas ? mine + k : ai];
            const bool trying_now = searching && !accept;
            bool stopped = false;
            const bool storing = kStoreWhileSearching && ai >= store_from;         // wave-uniform
            if (mine < cfg.n_alphas) {                                         // (else: nothing left for this wave in the last pass)
                TFMPC_PHASE_BEGIN();
                if (storing)
                    rollout(std::true_type{}, std::integral_constant<bool, kStoreWhileSearching>{}, std::integral_constant<int, NA>{}, al,
                            uhat, trying_now, xc, uc, cc, J, trying_now, rJ, true, &stopped);
                else {
#ifndef TFMPC_AB_NO_PRIORITY
                    // (see `kGiveUp`) SIMD partners are waves w and w + NW / 2 (of this group, or of the group that shares the CU): the earlier position first
                    if constexpr (kGiveUp) set_priority(NW == 8 ? 3 - (wv >> 1) : 3 - wv);
#endif
                    rollout(std::true_type{}, std::false_type{}, std::integral_constant<int, NA>{}, al,
                            uhat, false, xc, uc, cc, J, trying_now, rJ, true, &stopped,
                            kSegments ? ckpt + (size_t)wv * NA * (kMaxGroupWaves - 1) * NT * kTileElems : nullptr, 0, -1, nullptr,
                            kGiveUp ? parity : -1);
                    if constexpr (kGiveUp) set_priority(0);
                }
                TFMPC_PHASE_END(2);
            } else {
#pragma unroll
                for (int k = 0; k < NA; ++k) J[k] = 0.0f;
            }
            if constexpr (kGiveUp) {                                           // (see `kGiveUp`; every wave posts, also one without a step size left)
                bool mine_accepts = false;                                     // c1 == 0 (early_exit): z >= c1 <=> J_hat - J >= 0 on either branch of :342-346
#pragma unroll
                for (int k = 0; k < NA; ++k) mine_accepts = mine_accepts || (mine + k < cfg.n_alphas && (rJ - J[k]) >= 0.0f);
                const unsigned long long m = __builtin_amdgcn_ballot_w64(trying_now && !stopped && mine_accepts);
                if (lane == 0) {
                    __hip_atomic_store(&give_acc[parity][wv], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_or(&give_posted[parity], 1u << wv, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_and(&give_posted[parity ^ 1], ~(1u << wv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            if (NW > 1) {                                                      // the group's J(alpha) of this pass, in step-size order
#pragma unroll
                for (int k = 0; k < NA; ++k) x_pass[parity][wv][k][lane] = J[k];
                {
                    TFMPC_PHASE_BEGIN();
                    __syncthreads();
                    TFMPC_PHASE_END(5);      // (probe builds: how long wave 0 waits for the group's slowest chain)
                }
#pragma unroll
                for (int w = 0; w < NW; ++w)
#pragma unroll
                    for (int k = 0; k < NA; ++k) Jall[w * NA + k] = x_pass[parity][w][k][lane];
                parity ^= 1;
            } else {
#pragma unroll
                for (int k = 0; k < NA; ++k) Jall[k] = J[k];
            }
            if (trying_now) complete = storing && !stopped;
#pragma unroll
            for (int k = 0; k < NAP; ++k) {                                    // in the reference's order
                const bool trying = searching && !accept && ai + k < cfg.n_alphas;
                const float alpha = cfg.alphas[ai + k < cfg.n_alphas ? ai + k : ai];
                // residual = max |alpha k_t| (:206, before clipping) = alpha max |k_t|: rounding is monotone and alpha >= 0
                const float res = alpha * kmax;
                const float delta_J = -alpha * (dV1 + alpha * 0.0f);           // :339 (dV2 == 0 here)
                const float dcost = rJ - Jall[k];
                const float z = (delta_J > 0.0f) ? dcost / delta_J : sgnf_(dcost); // :342-346
                if (trying) {
                    residual = res;
                    alpha_last = alpha;
                    J_last = Jall[k];
                    index_last = ai + k;
                    chain_last = k;
                    if (z >= cfg.c1) { accept = true; last_index = ai + k; }   // :351-353
#ifdef TFMPC_CFG5_TRACE
                    if (NW == 1 && g_cfg5_trace && q == 0 && live && iteration + attempts < 16 && ai + k < 11) {
                        int *tr = g_cfg5_trace + ((size_t)b * 16 + iteration + attempts) * 12;
                        tr[1 + ai + k] = trace_fa[k % 4] < 0 ? T + 1 : trace_fa[k % 4];
                        if (accept) tr[0] = ai + k;
                    }
#endif
                }
            }
        }
        const bool small_step = searching && residual < cfg.atol;              // :253-257
        const bool take = searching && (small_step || accept);                 // (:253 takes the last rollout even if rejected)
        if (__any(take && !complete)) {     // (also: the last step size tried was cut short and :253 takes it all the same)
            const float al[1] = {alpha_last};
            float J[1];
            // (a column taken WITHOUT having accepted -- :253 on a rejected, possibly cut-short last try -- has no checkpoints)
            if (kSegments && !__any(take && !accept)) {
                TFMPC_PHASE_BEGIN();
                const int lo = wv * t_seg, hi = lo + t_seg < T ? lo + t_seg : T;
                const int kacc = take ? chain_last : 0;                       // the accepted chain of its pass
                if (lo < hi)
                    rollout(std::true_type{}, std::true_type{}, one_t{}, al, uhat, take && !complete, xc, uc, cc, J, false, 0.0f, false, nullptr,
                            nullptr, lo, hi, ckpt + ((size_t)kacc * (kMaxGroupWaves - 1) + (wv > 0 ? wv - 1 : 0)) * NT * kTileElems);
                TFMPC_PHASE_END(3);
                {
                    TFMPC_PHASE_BEGIN();
                    group_sync<NW>();              // every segment is visible to wave 0's sweep
                    TFMPC_PHASE_END(6);
                }
            } else if (wv == 0) {
                TFMPC_PHASE_BEGIN();
                rollout(std::true_type{}, std::true_type{}, one_t{}, al, uhat, take && !complete, xc, uc, cc, J);
                TFMPC_PHASE_END(3);
            }
        }
        if (a.trace.rows && active && live && wv == 0 && ql == 0)             // one lane per instance (tfmpc_ilqr_solve_trace_f32)
            trace_write(a.trace, b, row_pass, iteration, mu_pass, delta_pass, rJ, g_norm, index_last, alpha_last, J_last,
                        searching ? (accept ? 1 : 0) : -1, searching ? residual : -1.0f);
        if (take) flip ^= 1;                                                   // the candidate becomes the nominal
        if (converged_g || small_step) done = true;                            // converged
        else if (searching && accept) {                                        // :259-266
            delta = fminf(1.0f / cfg.delta_0, delta / cfg.delta_0);
            mu = (mu * delta > cfg.mu_min) ? mu * delta : 0.0f;
            if (++iteration >= cfg.max_iterations) done = true;
        } else if (searching) {                                                // :267-270
            delta = fmaxf(cfg.delta_0, delta * cfg.delta_0);
            mu = fmaxf(cfg.mu_min, mu * delta);
            if (++attempts >= cfg.max_attempts || !(mu < 1e30f)) { status |= TFMPC_ST_MAX_ATTEMPTS; done = true; }
        }
    }
    // the nominal trajectory of every column goes out to the instance-major output arrays (16-bit containers: widened);
    // the waves of a group take alternate time steps
    group_sync<NW>();                                      // wave 0's last stored rollout is visible to the group
    {
        const TT *xsrc = xbuf[flip], *usrc = ubuf[flip], *csrc = cbuf[flip];
        float *xdst = a.states + b * (T + 1) * n, *udst = a.actions + b * T * m, *cdst = a.costs + b * (T + 1);
        for (int t = wv; t <= T; t += NW) {
            float v[NV];
            ldw<NT>(xsrc, t, wl, v);
            stv<NT, VW>(xdst + (size_t)t * n, n, ql, live, v);
            if (t < T) {
                ldw<NT>(usrc, t, wl, v);
                stv<NT, VW>(udst + (size_t)t * m, m, ql, live, v);
            }
            if (ql == 0) { const float c = ldc(csrc + (size_t)t * kCostLd); if (live) gst(cdst + t, c); }
        }
    }
#ifdef TFMPC_PHASE_PROBE
    if (g_cfg5_phases && lane == 0) {                      // [group][wave of the group (8 slots)][16]
        unsigned long long *out = g_cfg5_phases + ((size_t)blockIdx.x * kMaxGroupWaves + wv) * 16;
        for (int i = 0; i < 7; ++i) { out[i == 5 ? 7 : i] = phase_acc[i]; out[8 + i] = phase_cnt[i]; }      // (4: the sweep's coefficient phase, 5 -> slot 7 / 6: barrier waits; slot 5: the kernel)
        out[5] = __builtin_amdgcn_s_memtime() - kernel_t0;
    }
#endif
    if (iteration >= cfg.max_iterations) iteration = cfg.max_iterations - 1;
    if (wv == 0 && ql == 0 && live) {
        const float cT = ldc(cbuf[flip] + (size_t)T * kCostLd);
        if (!(cT == cT)) status |= TFMPC_ST_NAN;
        a.iterations[b] = iteration;
        a.status[b] = status;
    }
}

}  // namespace

// ---- host side.  The instantiations are spread over NINE translation units (Makefile: this file compiled with
// -DTFMPC_AM_PART=0..7 = (HVAC | Reservoir) x (two tiles | one tile with 1, 2, 4 instances per column), 18 kernels each, and parts 8, 9 = the
// Reservoir-chain forms (two tiles; one tile of four instances per column), built in parallel); part 0 also carries the host functions.  Without the macro (tools/probes) everything is one unit.
#ifndef TFMPC_AM_PART
#define TFMPC_AM_PART -1
#endif

#if TFMPC_AM_PART <= 0
#ifdef TFMPC_PHASE_PROBE
// probe builds only (tools/probes/cfg5_phases.py): cycles a group spends per phase, [group][8] 64-bit counters:
// 0 start rollout, 1 sweep, 2 search passes, 3 stored rollout, 4 copy-out, 5 whole kernel, 6 search rollouts, 7 sweeps
extern "C" int tfmpc_debug_cfg5_phases(unsigned long long *device_buffer)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cfg5_phases), &device_buffer, sizeof(device_buffer));
}
#endif

#ifdef TFMPC_CFG5_TRACE
extern "C" int tfmpc_debug_cfg5_trace(int *device_buffer)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cfg5_trace), &device_buffer, sizeof(device_buffer));
}
#endif

size_t ilqr_adjoint_mfma_workspace_bytes(int B, int n, int m, int T)
{
    if (B <= 0 || n != m || n < 1 || n > 32 || T < 0) return 0;
    const int pk = n <= 4 ? 4 : (n <= 8 ? 2 : 1);                   // as in ilqr_adjoint_mfma_launch
    const size_t waves = ((size_t)B + kCols * pk - 1) / (kCols * pk);
    const int nt = n > 16 ? 2 : 1;
    return waves * adjoint_mfma_wave_bytes(nt, T) + (waves < (size_t)kCoefGroups ? waves : (size_t)kCoefGroups) * adjoint_mfma_coefficient_bytes(nt, T);
}

bool ilqr_adjoint_mfma_supported(const TfmpcEnv &env, const TfmpcIlqrConfig &cfg)
{
    TfmpcIlqrConfig fp32 = cfg;
    fp32.storage_bf16 = 0;                     // the shape test of the register-resident kernels; 16-bit containers are built in here
    if (!ilqr_adjoint_supported(env, fp32)) return false;
    for (int i = 0; i < TFMPC_ENV_MAX_PARAMS; ++i)
        if (env.p[i] && env.stride[i] != 0) return false;          // the coupling matrix must be the batch's, not the instance's
    return true;
}

#endif      // TFMPC_AM_PART <= 0

#define TFMPC_LAUNCH_PAIR(KIND, NT_, PK_, VW_, BF_)                                                                            \
    if (nw == 2) hipLaunchKernelGGL((ilqr_adjoint_mfma_kernel<KIND, NT_, VW_, PK_, BF_, 2>), grid, block, 0, stream, env, cfg, a); \
    else if (!BF_ && nw == 4) hipLaunchKernelGGL((ilqr_adjoint_mfma_kernel<KIND, NT_, VW_, PK_, false, 4>), grid, block, 0, stream, env, cfg, a); \
    else if (!BF_ && nw == 8) hipLaunchKernelGGL((ilqr_adjoint_mfma_kernel<KIND, NT_, VW_, PK_, false, 8>), grid, block, 0, stream, env, cfg, a); \
    else
#define TFMPC_LAUNCH_AM4(KIND, NT_, PK_, VW_, BF_)                                                                             \
    do {                                                                                                                       \
        TFMPC_LAUNCH_PAIR(KIND, NT_, PK_, VW_, BF_)                                                                            \
        hipLaunchKernelGGL((ilqr_adjoint_mfma_kernel<KIND, NT_, VW_, PK_, BF_, 1>), grid, block, 0, stream, env, cfg, a);      \
    } while (0)
#define TFMPC_LAUNCH_AM3(KIND, NT_, PK_, VW_)                                                                                  \
    do {                                                                                                                       \
        if (cfg.storage_bf16) TFMPC_LAUNCH_AM4(KIND, NT_, PK_, VW_, true);                                                     \
        else TFMPC_LAUNCH_AM4(KIND, NT_, PK_, VW_, false);                                                                     \
    } while (0)
#define TFMPC_LAUNCH_AM2(KIND, NT_, PK_)                                                                                        \
    do {                                                                                                                       \
        if (vw == 4) TFMPC_LAUNCH_AM3(KIND, NT_, PK_, 4);                                                                      \
        else if (vw == 2) TFMPC_LAUNCH_AM3(KIND, NT_, PK_, 2);                                                                 \
        else TFMPC_LAUNCH_AM3(KIND, NT_, PK_, 1);                                                                              \
    } while (0)
#define TFMPC_AM_PART_FN(P, KIND, NT_, PK_)                                                                                     \
    int ilqr_adjoint_mfma_launch_part##P(const TfmpcEnv &env, const TfmpcIlqrConfig &cfg, const AdjointSolveArgs &a, hipStream_t stream, \
                                         int vw, int nw, dim3 grid, dim3 block)                                                 \
    {                                                                                                                          \
        TFMPC_LAUNCH_AM2(KIND, NT_, PK_);                                                                                      \
        return hipGetLastError() == hipSuccess ? TFMPC_OK : TFMPC_ERR_LAUNCH;                                                  \
    }
#define TFMPC_AM_PART_DECL(P)                                                                                                  \
    int ilqr_adjoint_mfma_launch_part##P(const TfmpcEnv &, const TfmpcIlqrConfig &, const AdjointSolveArgs &, hipStream_t, int, int, dim3, dim3);
TFMPC_AM_PART_DECL(0) TFMPC_AM_PART_DECL(1) TFMPC_AM_PART_DECL(2) TFMPC_AM_PART_DECL(3)
TFMPC_AM_PART_DECL(4) TFMPC_AM_PART_DECL(5) TFMPC_AM_PART_DECL(6) TFMPC_AM_PART_DECL(7)
TFMPC_AM_PART_DECL(8) TFMPC_AM_PART_DECL(9)
#if TFMPC_AM_PART < 0 || TFMPC_AM_PART == 8
// the Reservoir chain (EnvM<kEnvReservoirChain>): one-wave groups, fp32 containers, two tiles -- the form a full cfg5 batch takes
int ilqr_adjoint_mfma_launch_part8(const TfmpcEnv &env, const TfmpcIlqrConfig &cfg, const AdjointSolveArgs &a, hipStream_t stream,
                                   int vw, int, dim3 grid, dim3 block)
{
    if (vw == 4) hipLaunchKernelGGL((ilqr_adjoint_mfma_kernel<kEnvReservoirChain, 2, 4, 1, false, 1>), grid, block, 0, stream, env, cfg, a);
    else if (vw == 2) hipLaunchKernelGGL((ilqr_adjoint_mfma_kernel<kEnvReservoirChain, 2, 2, 1, false, 1>), grid, block, 0, stream, env, cfg, a);
    else hipLaunchKernelGGL((ilqr_adjoint_mfma_kernel<kEnvReservoirChain, 2, 1, 1, false, 1>), grid, block, 0, stream, env, cfg, a);
    return hipGetLastError() == hipSuccess ? TFMPC_OK : TFMPC_ERR_LAUNCH;
}
#endif
#if TFMPC_AM_PART < 0 || TFMPC_AM_PART == 9
// the Reservoir chain in ONE tile, four instances per column (EnvChain1; res4.config.json): 16-byte pieces, fp32 containers, every group size
int ilqr_adjoint_mfma_launch_part9(const TfmpcEnv &env, const TfmpcIlqrConfig &cfg, const AdjointSolveArgs &a, hipStream_t stream,
                                   int, int nw, dim3 grid, dim3 block)
{
    if (nw == 8) hipLaunchKernelGGL((ilqr_adjoint_mfma_kernel<kEnvReservoirChain1, 1, 4, 4, false, 8>), grid, block, 0, stream, env, cfg, a);
    else if (nw == 4) hipLaunchKernelGGL((ilqr_adjoint_mfma_kernel<kEnvReservoirChain1, 1, 4, 4, false, 4>), grid, block, 0, stream, env, cfg, a);
    else if (nw == 2) hipLaunchKernelGGL((ilqr_adjoint_mfma_kernel<kEnvReservoirChain1, 1, 4, 4, false, 2>), grid, block, 0, stream, env, cfg, a);
    else hipLaunchKernelGGL((ilqr_adjoint_mfma_kernel<kEnvReservoirChain1, 1, 4, 4, false, 1>), grid, block, 0, stream, env, cfg, a);
    return hipGetLastError() == hipSuccess ? TFMPC_OK : TFMPC_ERR_LAUNCH;
}
#endif
#if TFMPC_AM_PART < 0 || TFMPC_AM_PART == 0
TFMPC_AM_PART_FN(0, TFMPC_ENV_HVAC, 2, 1)
#endif
#if TFMPC_AM_PART < 0 || TFMPC_AM_PART == 1
TFMPC_AM_PART_FN(1, TFMPC_ENV_HVAC, 1, 1)
#endif
#if TFMPC_AM_PART < 0 || TFMPC_AM_PART == 2
TFMPC_AM_PART_FN(2, TFMPC_ENV_HVAC, 1, 2)
#endif
#if TFMPC_AM_PART < 0 || TFMPC_AM_PART == 3
TFMPC_AM_PART_FN(3, TFMPC_ENV_HVAC, 1, 4)
#endif
#if TFMPC_AM_PART < 0 || TFMPC_AM_PART == 4
TFMPC_AM_PART_FN(4, TFMPC_ENV_RESERVOIR, 2, 1)
#endif
#if TFMPC_AM_PART < 0 || TFMPC_AM_PART == 5
TFMPC_AM_PART_FN(5, TFMPC_ENV_RESERVOIR, 1, 1)
#endif
#if TFMPC_AM_PART < 0 || TFMPC_AM_PART == 6
TFMPC_AM_PART_FN(6, TFMPC_ENV_RESERVOIR, 1, 2)
#endif
#if TFMPC_AM_PART < 0 || TFMPC_AM_PART == 7
TFMPC_AM_PART_FN(7, TFMPC_ENV_RESERVOIR, 1, 4)
#endif
#undef TFMPC_AM_PART_FN
#undef TFMPC_AM_PART_DECL
#undef TFMPC_LAUNCH_AM2
#undef TFMPC_LAUNCH_AM3
#undef TFMPC_LAUNCH_AM4
#undef TFMPC_LAUNCH_PAIR

#if TFMPC_AM_PART <= 0
int ilqr_adjoint_mfma_launch(const TfmpcEnv &env, const TfmpcIlqrConfig &cfg, const AdjointSolveArgs &a, hipStream_t stream)
{
    auto aligned = [&](unsigned mask) {
        const void *ps[] = {a.x0, a.u_init, a.states, a.actions};
        for (const void *p : ps)
            if (reinterpret_cast<uintptr_t>(p) & mask) return false;
        return true;
    };
    if (!a.wave_ws || (reinterpret_cast<uintptr_t>(a.wave_ws) & 15u)) return TFMPC_ERR_WORKSPACE;
    const int vw = (env.n % 4 == 0 && aligned(15u)) ? 4 : ((env.n % 2 == 0 && aligned(7u)) ? 2 : 1);
    // instances per column: one up to n = 32 (two tiles) or 16, two for n <= 8, four for n <= 4
    const int pk = env.n <= 4 ? 4 : (env.n <= 8 ? 2 : 1);
    // Waves per sixteen-column group (see the kernel).  One step size per wave instead of several chains in one wave's
    // instruction stream, and the stored rollout as one segment of the horizon per wave: a group's iteration is a chain of
    // dependent 100-step phases, and wherever the chip has idle SIMDs more waves per group shorten it.  The form that
    // brings the launch to ~2 waves per SIMD (2 048 waves) wins (MI355X, 12 iterations, T = 100, tools/probes/group_waves_sweep.py;
    // ms by waves 1 / 2 / 4 / 8): hvac6 B = 16 384 (512 groups) 3.98 / 3.51 / 2.46 / 3.10; res4 B = 16 384 (256 groups) 3.68 /
    // 2.69 / 2.03 / 1.89, B = 65 536 (1 024 groups) 4.03 / 3.84 / 5.15 / 7.30; n = 32 Reservoir B = 256 6.50 / 4.86 / 3.46 / 3.05,
    // B = 8 192 (512 groups) 6.92 / 6.43 / 4.99 / 6.31; two tiles (n > 16) beyond 512 groups: one wave (B = 16 384: 8.48 / 9.77 HVAC,
    // 7.53 / 7.84 Reservoir -- a two-tile wave needs the registers of a whole SIMD half).  TFMPC_COSTATE_WAVES=1|2|4|8 forces a form.
    const int groups = (a.B + kCols * pk - 1) / (kCols * pk);
    const int forced = option_int(kOptCostateWaves, 0);
    int nw = (forced == 1 || forced == 2 || forced == 4 || forced == 8) ? forced
             : (groups <= 256 ? 8 : (groups <= 512 ? 4 : (groups <= 1024 && env.n <= 16 ? 2 : 1)));
    if (cfg.storage_bf16 && nw > 2) nw = 2;                          // (16-bit containers: the one- and two-wave forms)
    if (groups > kCoefGroups && nw > 2) nw = 2;                      // (a forced four- / eight-wave form on more groups than the coefficient slab is sized for)
    const dim3 block(kWave * nw), grid((a.B + kCols * pk - 1) / (kCols * pk));
    int part = (env.kind == TFMPC_ENV_HVAC ? 0 : 4) + (env.n > 16 ? 0 : (pk == 1 ? 1 : (pk == 2 ? 2 : 3)));
    // Reservoir with a promised chain topology, in the form a large batch takes (one wave per group, fp32 containers, two tiles): the
    // compile-time shift instantiation.  TFMPC_COSTATE_COUPLING=dense|runtime keeps the general kernel (A/B timing, bit-identity tests).
    if (part == 4 && nw == 1 && !cfg.storage_bf16 && (env.coupling_shift == 1 || env.coupling_shift == -1) && !a.dense_coupling &&
        !option_is(kOptCostateCoupling, "runtime"))
        part = 8;
    // ... and in the form the reference's own res4 config takes (one tile, four instances per column, n = 4 exactly, i -> i + 1): EnvChain1
    if (part == 7 && vw == 4 && env.n == 4 && !cfg.storage_bf16 && env.coupling_shift == 1 && !a.dense_coupling && !option_is(kOptCostateCoupling, "runtime"))
        part = 9;
    switch (part) {
    case 0: return ilqr_adjoint_mfma_launch_part0(env, cfg, a, stream, vw, nw, grid, block);
    case 1: return ilqr_adjoint_mfma_launch_part1(env, cfg, a, stream, vw, nw, grid, block);
    case 2: return ilqr_adjoint_mfma_launch_part2(env, cfg, a, stream, vw, nw, grid, block);
    case 3: return ilqr_adjoint_mfma_launch_part3(env, cfg, a, stream, vw, nw, grid, block);
    case 4: return ilqr_adjoint_mfma_launch_part4(env, cfg, a, stream, vw, nw, grid, block);
    case 5: return ilqr_adjoint_mfma_launch_part5(env, cfg, a, stream, vw, nw, grid, block);
    case 6: return ilqr_adjoint_mfma_launch_part6(env, cfg, a, stream, vw, nw, grid, block);
    case 8: return ilqr_adjoint_mfma_launch_part8(env, cfg, a, stream, vw, nw, grid, block);
    case 9: return ilqr_adjoint_mfma_launch_part9(env, cfg, a, stream, vw, nw, grid, block);
    default: return ilqr_adjoint_mfma_launch_part7(env, cfg, a, stream, vw, nw, grid, block);
    }
}
#endif      // TFMPC_AM_PART <= 0

}  // namespace tfmpc
