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
#define TFMPC_ST_ENV_FLAG 0x40     /* TfmpcEnv::coupling_shift promised a chain that `downstream` is not: nothing was computed   */
#define TFMPC_ST_QP_LATER_NOT_PD 0x20 /* tfmpc_boxqp_f32: a factorisation AFTER the first failed; the loop broke out there as
                                         optimization.py:47-51 does: x and free are those at the break.  (Inside iLQR the QP starts at the box
                                         centre, ilqr.py:369, where every coordinate is free: the first factorisation is of the whole H, and a
                                         positive definite H has no indefinite principal block -- the case needs a boundary start or rounding.) */

/* Library / device information ------------------------------------------------ */
int tfmpc_version(void);
/* Kernel-variant overrides for A/B timing and tests.  The environment variables TFMPC_LQR_KERNEL
 * (generic | lane | block), TFMPC_LQR_MFMA (f32 | bf16x3), TFMPC_ILQR_KERNEL (wave | lane | lane1 |
 * lean | lean1 | costate_mfma), TFMPC_LQR_WAVES (4 | 5: the register budget -- waves per SIMD -- of the headline LQR kernel's
 * instantiation; default: chosen from the batch size; same bits), TFMPC_COSTATE_WAVES (1 | 2 | 4 | 8: waves per sixteen-instance group of the HVAC /
 * Reservoir kernel; default: the form that brings the launch to about two waves per SIMD) and TFMPC_ILQR_RETRY (bracket: the control-limited LQ kernel looks for the regularisation level
 * of a failed factorisation around the level of its previous pass instead of probing 0, 1, 2, ... as ilqr.py:285-315
 * does -- another regularisation path on ~0.5 % of the instances; unsorted: that kernel launches its blocks in instance order instead of
 * starting the instances whose first backward pass probes most levels first -- same results, for A/B timing; levels: a batch whose sample shows no
 * such instance keeps instance order instead of starting its largest start costs first, as before round 6 -- same results; sorted: the whole batch is
 * probed whatever the sample says) and TFMPC_COSTATE_COUPLING
 * (dense: the 16-per-wave Reservoir kernel multiplies by its `downstream` matrix also when that matrix is a shift -- a chain of
 * reservoirs, every config the reference holds -- instead of moving rows; same bits, for A/B timing and tests), TFMPC_BOX_HELPERS (off | number of
 * helper teams, default 8: a control-limited batch of more than 4 096 instances without heavy ones in the launcher's sample lends five helper
 * blocks to each of its longest-running instances, which roll out the step sizes of a line search side by side -- same bits) and
 * TFMPC_BOX_HELP_AFTER (passes before an instance may claim a team, default 8), TFMPC_ILQR_LQ_REUSE (0: the matrix-core iLQR kernel of the
 * unbounded LQ env runs the full backward pass in every iteration, as ilqr.py:94-172 does; default: from the second pass on it keeps K_t and
 * Q_uu(t)^-1 -- which for a time-invariant LQ env at mu = 0 do not depend on the trajectory, so the recomputed ones would be the same bits -- and
 * runs only the vector recursion for k_t, V_x: results agree to fp32 rounding), TFMPC_GROUP_STORED (1 .. 4, default 4: how many step sizes of a
 * line search of the 2 x 2 lane-group kernel keep their candidate trajectory in the workspace; a pass that adopts another one rolls it out once
 * more -- same bits, for tests and A/B timing), TFMPC_BOX_SPECULATE (off | n, default 0: after n rejected passes in a row the helper team of a
 * control-limited instance runs the backward passes of its next passes beside its own -- same bits) are read ONCE per process, at the first use of the library; afterwards only
 * this call changes them (value NULL or "" = back to the dispatcher's own choice).  Returns TFMPC_ERR_ARG
 * for an unknown name.  Process-wide; not meant to be flipped while other threads launch. */
int tfmpc_set_option(const char *name, const char *value);
/* The current override of `name` as a string in buf[len] ("" = none), so that a caller can restore what it replaces. */
int tfmpc_get_option(const char *name, char *buf, int len);
/* Name of the kernel variant the dispatcher would pick for an LQR shape
 * ("generic_wave", "mfma_16x8", ...).  Host-only; never touches the GPU. */
const char *tfmpc_lqr_kernel_name(int n, int m, int T);
/* Name of the kernel family the calling thread's LAST tfmpc_ilqr_solve_f32 / tfmpc_ilqr_solve_trace_f32 was dispatched to ("" before
 * the first).  A traced solve records its rows in the kernel that solves it; on HVAC / Reservoir and on the per-lane kernel of the
 * tiny 2-D envs that can be another kernel than the untraced solve takes (the LQ env's matrix-core kernels record their own trace
 * since round 4), which is what this call lets a caller report next to the trace (the CLI's -v writes it into trace.log). */
const char *tfmpc_ilqr_last_kernel_name(void);
/* Workgroups of the calling thread's last PERSISTENT lane-group launch (the 2-D envs of BASELINE configs[3] beyond 4 096 instances: the grid
 * is what the device holds at once for that launch's LDS size, and the groups pull instances from a queue); 0 before the first.  A
 * diagnostic: the grid must follow the horizon and the device of each launch, not those of the process's first launch. */
int tfmpc_ilqr_last_group_grid(void);

/* ---------------------------------------------------------------- LQR --------
 * Problem (tfmpc/solvers/lqr.py:18-57): x' = F [x;u] + f, stage cost
 * 1/2 z^T C z + c^T z, final cost 1/2 x^T C_xx x + c_x^T x, with
 * F[n][n+m], f[n], C[n+m][n+m], c[n+m].
 */

/* PRECONDITION of tfmpc_lqr_backward_f32 / _forward_f32 / _solve_f32: C is symmetric (every problem the
 * reference builds is: make_lqr draws make_spd_matrix, make_lqr_linear_navigation a diagonal,
 * tfmpc/envs/__init__.py:9-30).  The fast kernels use it: Q_xu is read as Q_ux^T, Q_uu is eliminated from its
 * upper triangle without pivoting (so Q_uu must be positive definite: C_uu > 0, C >= 0; a non-positive pivot
 * is reported as TFMPC_ST_NOT_PD / TFMPC_ST_SINGULAR) and V is kept exactly symmetric.  For a C that is NOT
 * symmetric the reference's recursion (lqr.py:74-105: Q_ux and Q_xu separately, tf.linalg.inv, four-term
 * update, no symmetrisation) gives a different answer than any symmetrised solve; call the *_general_f32
 * twins below, which restate it term by term on the wave-per-instance kernel (same arguments, any C).
 * The Python class checks C once at construction and picks the entry points accordingly. */

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

/* The policy (K_t, k_t) and the value function (V_t, v_t, const_t) of lqr.py:107-129 in 16-BIT containers (SURVEY.md 8f
 * N4: they are 9 x the bytes of the trajectory -- 81.8 KB per solve at n = 16, m = 8, T = 50, 40.9 KB like this).
 * K16, k16, V16, v16, cst16: bf16 arrays (uint16_t) in the layouts of K, k, V, v, cst, any may be NULL; every value is
 * the fp32 result rounded to nearest even, i.e. exactly what rounding tfmpc_lqr_solve_f32's outputs would give.  The
 * trajectory (states, actions, costs: fp32) is rolled out with the fp32 gains and is identical to
 * tfmpc_lqr_solve_f32's.  `workspace`: tfmpc_lqr_workspace_bytes (the fp32 gains live there).  Shapes the lane /
 * workgroup kernels serve go to the wave kernel; TFMPC_ERR_UNSUPPORTED beyond its LDS limit (n = m ~ 48). */
int tfmpc_lqr_solve_bf16out_f32(int B, int n, int m, int T,
                                const float *F, long strideF, const float *f, long stride_f,
                                const float *C, long strideC, const float *c, long stride_c,
                                const float *x0,
                                float *states, float *actions, float *costs,
                                uint16_t *K16, uint16_t *k16, uint16_t *V16, uint16_t *v16, uint16_t *cst16,
                                int32_t *status, void *workspace, size_t workspace_bytes, void *stream);
/* LQR.backward alone, 16-bit outputs (same contract; `workspace` as above). */
int tfmpc_lqr_backward_bf16out_f32(int B, int n, int m, int T,
                                   const float *F, long strideF, const float *f, long stride_f,
                                   const float *C, long strideC, const float *c, long stride_c,
                                   uint16_t *K16, uint16_t *k16, uint16_t *V16, uint16_t *v16, uint16_t *cst16,
                                   int32_t *status, void *workspace, size_t workspace_bytes, void *stream);


/* The same three calls for a C that is not symmetric (see PRECONDITION above). */
int tfmpc_lqr_backward_general_f32(int B, int n, int m, int T,
                                   const float *F, long strideF, const float *f, long stride_f,
                                   const float *C, long strideC, const float *c, long stride_c,
                                   float *K, float *k, float *V, float *v, float *cst,
                                   int32_t *status, void *stream);
int tfmpc_lqr_forward_general_f32(int B, int n, int m, int T,
                                  const float *F, long strideF, const float *f, long stride_f,
                                  const float *C, long strideC, const float *c, long stride_c,
                                  const float *K, long strideK, const float *k, long stride_k,
                                  const float *x0,
                                  float *states, float *actions, float *costs, void *stream);
int tfmpc_lqr_solve_general_f32(int B, int n, int m, int T,
                                const float *F, long strideF, const float *f, long stride_f,
                                const float *C, long strideC, const float *c, long stride_c,
                                const float *x0,
                                float *states, float *actions, float *costs,
                                float *K, float *k, float *V, float *v, float *cst,
                                int32_t *status, void *workspace, size_t workspace_bytes, void *stream);


/* --------------------------------------------------------------- iLQR --------
 * Control-limited iLQR (tfmpc/solvers/ilqr.py) over the reference's differentiable
 * environments (tfmpc/envs).  An environment is described by a kind tag plus
 * parameter arrays; Jacobians / Hessians are closed forms evaluated on the device
 * (the reference obtains them by autodiff, tfmpc/envs/diffenv.py:13-101).
 */
#define TFMPC_ENV_LQ 0          /* (F, f, C, c) of tfmpc/solvers/lqr.py:36-57 as an env: n != m allowed */
#define TFMPC_ENV_NAVLQR 1      /* tfmpc/envs/lqr/navigation/__init__.py:8-47                           */
#define TFMPC_ENV_NAVIGATION 2  /* tfmpc/envs/navigation/__init__.py:9-74  (cec=True)                   */
#define TFMPC_ENV_HVAC 3        /* tfmpc/envs/hvac/__init__.py:8-149                                    */
#define TFMPC_ENV_RESERVOIR 4   /* tfmpc/envs/reservoir/__init__.py:9-105 (cec=True)                    */
#define TFMPC_ENV_USER 5        /* ANY differentiable env (tfmpc/envs/diffenv.py:13-101 differentiates whatever it is handed): transition /
                                   cost / final_cost given as device functions, derivatives by forward-mode dual numbers in the kernel.  Not
                                   served by this library's entry points (TFMPC_ERR_ARG): tfmpc.envs.deviceenv compiles csrc/user_env_kernels.hip.in
                                   around the user's source into a companion library that exports tfmpc_userenv_* twins of the env-dependent
                                   entry points (rollout, derivatives, forward, solve_trace; same signatures).  p0 = the env's parameter
                                   floats, n_zones of them per instance. */
#define TFMPC_ENV_MAX_PARAMS 10
#define TFMPC_MAX_ALPHAS 16

/* Parameter arrays p[i] per kind (device pointers, fp32; stride[i] = element
 * distance between instances, 0 = shared by the whole batch):
 *   LQ         p0 F[n][n+m]  p1 f[n]  p2 C[n+m][n+m]  p3 c[n+m]
 *   NAVLQR     p0 goal[n]; scalar[0] = beta
 *   NAVIGATION p0 goal[n]  p1 center[Z][n]  p2 decay[Z]           (n_zones = Z, n <= 8)
 *   HVAC       p0 temp_outside  p1 temp_hall  p2 temp_lower_bound  p3 temp_upper_bound
 *              p4 adj_outside/R_outside  p5 adj_hall/R_hall  p6 capacity  p7 air_max   (all [n])
 *              p8 G[n][n] = (adj | adj^T) / R_wall
 *   RESERVOIR  p0 max_res_cap  p1 lower_bound  p2 upper_bound  p3 low_penalty  p4 high_penalty
 *              p5 set_point_penalty  p6 rain_shape*rain_scale   (all [n])  p7 downstream[n][n]
 * low[m] / high[m]: action bounds (+-inf allowed, shared by the batch); `bounded`
 * is gym's Box.is_bounded(): every bound finite (ilqr.py:136). */
typedef struct TfmpcEnv {
    int32_t kind, n, m, n_zones, bounded;
    int32_t any_finite_bound;   /* 1 if any entry of low/high is finite (forward clips, ilqr.py:197) */
    int32_t coupling_shift;     /* RESERVOIR only, a PROMISE by the caller: +1 = `downstream` is the chain i -> i + 1 (D[i][i+1] = 1, every other
                                   entry 0), -1 = the chain i -> i - 1, 0 = no statement (any matrix).  Every config the reference holds
                                   is a chain (tests/conftest.py:70-75, reservoir/res4.config.json:13-18); with the promise a large batch runs an
                                   instantiation without coupling products.  Checked on the device against p7 before anything is computed:
                                   a broken promise leaves TFMPC_ST_ENV_FLAG in status[b] and the outputs untouched.  (Row moves
                                   equal the products for FINITE states; a product spreads one row's Inf / NaN into every row, a row move
                                   hands it to the neighbour only: a DIVERGING instance's status can differ from TFMPC_COSTATE_COUPLING=dense.) */
    int32_t reserved2;
    const float *low, *high;
    const float *p[TFMPC_ENV_MAX_PARAMS];
    int64_t stride[TFMPC_ENV_MAX_PARAMS];
    float scalar[4];
} TfmpcEnv;

/* Solver hyper-parameters (ilqr.py:27-37) + the line-search step sizes
 * np.geomspace(1, alpha_min, 11) (ilqr.py:322) computed by the host. */
typedef struct TfmpcIlqrConfig {
    float atol;             /* 5e-3 */
    int32_t max_iterations; /* 100  */
    float mu_min;           /* 1e-6 */
    float delta_0;          /* 2.0  */
    float c1;               /* 0.0  */
    int32_t n_alphas;       /* 11   */
    float alphas[TFMPC_MAX_ALPHAS];
    int32_t max_attempts;   /* cap on rejected backward/line-search attempts per solve (the
                               reference loops without bound, ilqr.py:238-270) */
    int32_t storage_bf16;   /* 1: trajectories (and gains) kept in HBM between passes are rounded to bf16 when
                               stored, arithmetic stays fp32 -- BASELINE configs[4] "fp32 vs bf16".  HVAC /
                               Reservoir envs shared by the batch: REAL 16-bit containers in the workspace (half
                               the bytes); other envs: the wave kernel emulates the format in fp32 containers.
                               Outputs are fp32 arrays holding bf16-representable values either way. */
} TfmpcIlqrConfig;

/* iLQR.start (ilqr.py:53-82) with the random actions injected: roll the env from
 * x0[B][n] under actions[B][T][m]; writes states[B][T+1][n], costs[B][T+1]. */
int tfmpc_ilqr_rollout_f32(const TfmpcEnv *env, int B, int T, const float *x0, const float *actions,
                           float *states, float *costs, void *stream);

/* iLQR.derivatives (ilqr.py:84-92) = DiffEnv.get_linear_transition / get_quadratic_cost /
 * get_quadratic_final_cost (diffenv.py:13-101) at every (x_t, u_t), t < T, and at x_T.
 * states[B][T+1][n], actions[B][T][m].  Outputs (any may be NULL): f[B][T][n],
 * f_x[B][T][n][n], f_u[B][T][n][m], l[B][T], l_x[B][T][n], l_u[B][T][m], l_xx[B][T][n][n],
 * l_uu[B][T][m][m], l_ux[B][T][m][n], l_xu[B][T][n][m]; final fl[B], fl_x[B][n], fl_xx[B][n][n]. */
int tfmpc_ilqr_derivatives_f32(const TfmpcEnv *env, int B, int T, const float *states, const float *actions,
                               float *f, float *f_x, float *f_u, float *l, float *l_x, float *l_u,
                               float *l_xx, float *l_uu, float *l_ux, float *l_xu,
                               float *fl, float *fl_x, float *fl_xx, void *stream);

/* iLQR.backward (ilqr.py:94-172) on materialised models (layouts as above).  mu[B]
 * (mu_stride 1) or one shared value (mu_stride 0).  Controller per step: Cholesky-type
 * solve (:357-362), box-QP (:364-387 + tfmpc/utils/optimization.py:6-101) when `bounded`
 * and V_xx != 0, bang-bang otherwise (:140-141).  Outputs K[B][T][m][n], k[B][T][m], J[B],
 * dV1[B], dV2[B]; status[B] gets TFMPC_ST_NOT_PD where the reference would raise
 * InvalidArgumentError (outputs of that instance are then undefined). */
int tfmpc_ilqr_backward_f32(int B, int n, int m, int T, const float *actions,
                            const float *f_x, const float *f_u, const float *l, const float *l_x,
                            const float *l_u, const float *l_xx, const float *l_uu, const float *l_xu,
                            const float *fl, const float *fl_x, const float *fl_xx,
                            const float *low, const float *high, int bounded,
                            const float *mu, long mu_stride,
                            float *K, float *k, float *J, float *dV1, float *dV2, int32_t *status, void *stream);

/* iLQR.forward (ilqr.py:174-212): closed-loop rollout around (x[B][T+1][n], u[B][T][m])
 * with gains K, k and step alpha[B] (alpha_stride 1) or shared (0).  Outputs states,
 * actions, costs[B][T+1], J[B], residual[B]. */
int tfmpc_ilqr_forward_f32(const TfmpcEnv *env, int B, int T, const float *x, const float *u,
                           const float *K, const float *k, const float *alpha, long alpha_stride,
                           float *states, float *actions, float *costs, float *J, float *residual,
                           void *stream);

/* Scratch of tfmpc_ilqr_solve_f32, a function of the shape alone (the env kind is not known here): per instance the
 * gains K, k and one candidate trajectory; for n = m = 2 a line-search block per wavefront; for n = m <= 32 the two
 * wave-major trajectory buffers of the 16-instances-per-wave HVAC / Reservoir kernel (+ for n <= 16 the coefficients
 * of its two-part costate sweep: 3 KB per group and time step).  The control-limited LQ kernel keeps its helper teams'
 * board (TFMPC_BOX_HELPERS: ~60 KB per team) in the candidate-trajectory part: no extra bytes.  The workspace handed to
 * tfmpc_ilqr_solve_f32 must be 256-byte aligned (TFMPC_ERR_WORKSPACE otherwise). */
size_t tfmpc_ilqr_workspace_bytes(int B, int n, int m, int T);
/* The same for ONE env (round 6): only the parts the kernels of that env kind read -- never more than tfmpc_ilqr_workspace_bytes of its shape, and
 * hundreds of MB less at large batches (an HVAC / Reservoir batch does not carry the LQ kernels' slab, an LQ batch not the costate kernel's buffers,
 * neither the 2-D envs' scratch).  tfmpc_ilqr_solve*_f32 accept either size.  0 for a null env or a bad shape. */
size_t tfmpc_ilqr_workspace_bytes_for(const TfmpcEnv *env, int B, int T);

/* iLQR.solve (ilqr.py:214-283) in ONE launch: each wave runs its instance's whole
 * iteration loop (linearise on the fly, regularised backward pass with the local
 * Cholesky-failure retry of :285-315, 11-point line search :317-355, mu/delta schedule
 * :259-270, both convergence tests :243-257).  u_init[B][T][m] are the start actions
 * (ilqr.py:70 draws them from TF's RNG; the host injects them).  Outputs: the final
 * nominal trajectory states[B][T+1][n], actions[B][T][m], costs[B][T+1];
 * iterations[B] = the reference's returned loop index; status[B]. */
int tfmpc_ilqr_solve_f32(const TfmpcEnv *env, const TfmpcIlqrConfig *cfg, int B, int T,
                         const float *x0, const float *u_init,
                         float *states, float *actions, float *costs,
                         int32_t *iterations, int32_t *status,
                         void *workspace, size_t workspace_bytes, void *stream);

/* tfmpc_ilqr_solve_f32 with the DECISION TRACE of the solve (what the reference prints per pass through the body of
 * ilqr.py:238-270 -- `[BACKWARD] J_hat`, `[SOLVE] g_norm`, `[FORWARD] J`, the progress bar's J / g_norm / residual,
 * ilqr.py:243-279 -- and what its regularisation schedule holds): one row of TFMPC_TRACE_COLS floats per backward
 * pass + line search of an instance, in the order they happen, trace[B][trace_rows][TFMPC_TRACE_COLS]; rows beyond
 * trace_rows are dropped.  trace_len[B] = the number of passes the instance made (may exceed trace_rows).
 * Columns: see TFMPC_TR_*.  trace == NULL: identical to tfmpc_ilqr_solve_f32.  With a trace the solve runs on a
 * kernel that records one (the HVAC / Reservoir shared-env kernel, the 2-D lane-group kernel, else the generic
 * wave kernel): same algorithm, possibly another kernel than the untraced call would pick. */
#define TFMPC_TRACE_COLS 11
#define TFMPC_TR_ITERATION 0  /* the reference's loop index `iteration` (ilqr.py:227) of this pass            */
#define TFMPC_TR_MU 1         /* mu handed to _backward (ilqr.py:240), before a local Cholesky-failure bump   */
#define TFMPC_TR_DELTA 2      /* delta at that point                                                          */
#define TFMPC_TR_J_HAT 3      /* cost of the nominal trajectory as the backward pass sums it (ilqr.py:164)    */
#define TFMPC_TR_G_NORM 4     /* ilqr.py:243                                                                  */
#define TFMPC_TR_ALPHA_INDEX 5 /* position (0-based) in the step-size list of the LAST rollout of the line search;
                                 -1 when the pass ended on g_norm < atol before any rollout                  */
#define TFMPC_TR_ALPHA 6      /* that step size                                                               */
#define TFMPC_TR_J 7          /* its total cost J (ilqr.py:205-210)                                           */
#define TFMPC_TR_ACCEPTED 8   /* 1 = z >= c1 (ilqr.py:351-353), 0 = all step sizes rejected, -1 = no line search */
#define TFMPC_TR_RESIDUAL 9   /* ilqr.py:206 of that rollout (-1 without line search)                         */
#define TFMPC_TR_LEVEL 10     /* local regularisation bumps _backward needed before the pass factorised
                                 (ilqr.py:305-309; 0 = at TFMPC_TR_MU itself): the level whose gains the pass used */
int tfmpc_ilqr_solve_trace_f32(const TfmpcEnv *env, const TfmpcIlqrConfig *cfg, int B, int T,
                               const float *x0, const float *u_init,
                               float *states, float *actions, float *costs,
                               int32_t *iterations, int32_t *status,
                               float *trace, int trace_rows, int32_t *trace_len,
                               void *workspace, size_t workspace_bytes, void *stream);

/* tfmpc_ilqr_solve_trace_f32 that ALSO exports what the box-QP of every backward step ended on (version >= 300; test and
 * diagnosis instrument: ilqr.py:364-385 computes K_t from the free set optimization.py:35-72 returns, so two fp32 programs that
 * end a QP on different free sets differ DISCRETELY -- other rows of K_t are zero -- and no tolerance on numbers can compare them):
 * clamp_mask[B][trace_rows][T], bit a of entry (b, pass, t) set = action a was CLAMPED in the last factorised free set of that
 * step's QP (row a of K_t is zero); qp_iterations[B][trace_rows][T] = iterations of the projected-Newton loop
 * (optimization.py:24; 0 = the step ran no QP).  Both NULL: identical to tfmpc_ilqr_solve_trace_f32; both need `trace`.
 * Written by the control-limited matrix-core kernel (LQ env, n <= 16, m <= 8, bounded actions); other kernels leave the two
 * buffers untouched -- pre-fill them with 0xFF to tell. */
int tfmpc_ilqr_solve_trace_qp_f32(const TfmpcEnv *env, const TfmpcIlqrConfig *cfg, int B, int T,
                                  const float *x0, const float *u_init,
                                  float *states, float *actions, float *costs,
                                  int32_t *iterations, int32_t *status,
                                  float *trace, int trace_rows, int32_t *trace_len,
                                  uint8_t *clamp_mask, uint8_t *qp_iterations,
                                  void *workspace, size_t workspace_bytes, void *stream);

/* projected_newton_qp (tfmpc/utils/optimization.py:6-101): B independent box QPs
 * min 1/2 x^T H x + q^T x, low <= x <= high.  H[B][m][m], q/low/high/x0[B][m];
 * outputs x[B][m], free[B][m] (1.0 / 0.0), status[B]. */
int tfmpc_boxqp_f32(int B, int m, const float *H, const float *q, const float *low, const float *high,
                    const float *x0, float *x, float *free_mask, int32_t *status, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TFMPC_HIP_H */
