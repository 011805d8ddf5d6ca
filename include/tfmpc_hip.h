/* tfmpc_hip.h -- C ABI of the MI355X (gfx950) batched LQR / iLQR hot path.
 *
 * The reference (thiagopbueno/tf-mpc v0.7.0) is pure Python on TensorFlow and has
 * no FFI layer: its boundary is the Python class API of tfmpc/solvers/lqr.py and
 * tfmpc/solvers/ilqr.py.  Each entry point below names the reference method it
 * replaces (file:line in the reference tree); the Python host package
 * tf-mpc_amd/tfmpc binds them with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - All pointers are DEVICE pointers (hipMalloc / torch-ROCm storage), fp32,
 *    row-major, batch-major: the leading axis B indexes independent problem
 *    instances.  The library never allocates, frees or retains them.
 *  - A "batch stride" argument is the element distance between instances of that
 *    operand; 0 shares one copy between all instances.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls
 *    only enqueue work; they never synchronise.
 *  - Return value: 0 on success, <0 on argument / launch error (TFMPC_ERR_*).
 *    Numerical trouble is reported per instance in `status[B]` (bit flags
 *    TFMPC_ST_*), never by aborting.
 *  - Scratch is passed in explicitly and sized by the *_workspace_bytes query.
 */
#ifndef TFMPC_HIP_H
#define TFMPC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFMPC_OK 0
#define TFMPC_ERR_ARG (-1)         /* bad size / null pointer                          */
#define TFMPC_ERR_UNSUPPORTED (-2) /* shape exceeds what one wave's LDS tile can hold  */
#define TFMPC_ERR_LAUNCH (-3)      /* hipLaunchKernel reported an error                */
#define TFMPC_ERR_WORKSPACE (-4)   /* workspace too small                              */

#define TFMPC_ST_SINGULAR 0x1     /* LQR: zero pivot while inverting Q_uu (lqr.py:84)            */
#define TFMPC_ST_NOT_PD 0x2       /* iLQR: Cholesky of Q_uu_reg failed at least once (ilqr.py:305)*/
#define TFMPC_ST_NAN 0x4          /* a non-finite value reached an output                        */
#define TFMPC_ST_QP_MAXITER 0x8   /* box-QP hit its 100-iteration cap (optimization.py:13)       */
#define TFMPC_ST_MAX_ATTEMPTS 0x10 /* iLQR: backward/line-search attempt cap reached             */

/* Library / device information ------------------------------------------------ */
int tfmpc_version(void);
/* Name of the kernel variant the dispatcher would pick for an LQR shape
 * ("generic_wave", "mfma_16x8", ...).  Host-only; never touches the GPU. */
const char *tfmpc_lqr_kernel_name(int n, int m, int T);

/* ---------------------------------------------------------------- LQR --------
 * Problem (tfmpc/solvers/lqr.py:18-57): x' = F [x;u] + f, stage cost
 * 1/2 z^T C z + c^T z, final cost 1/2 x^T C_xx x + c_x^T x, with
 * F[n][n+m], f[n], C[n+m][n+m], c[n+m].
 */

/* Bytes of scratch tfmpc_lqr_solve_f32 needs when K / k are not requested as
 * outputs (the gains are kept for the forward pass). */
size_t tfmpc_lqr_workspace_bytes(int B, int n, int m, int T);

/* LQR.backward (lqr.py:59-129): Riccati recursion from V=C_xx, v=c_x.
 * Outputs (each may be NULL except K, k): K[B][T][m][n], k[B][T][m],
 * V[B][T][n][n], v[B][T][n], cst[B][T] -- entry t is time t, the terminal entry
 * is dropped as in lqr.py:126-127.  status[B] may be NULL. */
int tfmpc_lqr_backward_f32(int B, int n, int m, int T,
                           const float *F, long strideF, const float *f, long stride_f,
                           const float *C, long strideC, const float *c, long stride_c,
                           float *K, float *k, float *V, float *v, float *cst,
                           int32_t *status, void *stream);

/* LQR.forward (lqr.py:131-161): u_t = K_t x_t + k_t, rollout and costs.
 * K[B][T][m][n] (batch stride strideK, 0 = shared policy), k likewise.
 * Outputs states[B][T+1][n], actions[B][T][m], costs[B][T+1]. */
int tfmpc_lqr_forward_f32(int B, int n, int m, int T,
                          const float *F, long strideF, const float *f, long stride_f,
                          const float *C, long strideC, const float *c, long stride_c,
                          const float *K, long strideK, const float *k, long stride_k,
                          const float *x0,
                          float *states, float *actions, float *costs, void *stream);

/* LQR.solve (lqr.py:163-166): backward + forward fused in one launch.
 * K, k, V, v, cst are optional outputs (NULL to skip).  If K or k is NULL,
 * `workspace` must hold tfmpc_lqr_workspace_bytes(...) bytes. */
int tfmpc_lqr_solve_f32(int B, int n, int m, int T,
                        const float *F, long strideF, const float *f, long stride_f,
                        const float *C, long strideC, const float *c, long stride_c,
                        const float *x0,
                        float *states, float *actions, float *costs,
                        float *K, float *k, float *V, float *v, float *cst,
                        int32_t *status, void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TFMPC_HIP_H */
