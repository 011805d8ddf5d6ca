/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of tfmpc/solvers/lqr.py
 * (backward :59-129, forward :131-161, solve :163-166) for a batch of
 * independent instances.  It is the checker for the HIP path at sizes the numpy
 * restatement is too slow for, and the timed "port" CPU baseline of bench.py.
 * It is never linked into or called from the product library.
 *
 * Operation order follows the reference: Q = C + (F^T V) F, q = c + (F^T V) f +
 * F^T v, explicit general inverse of Q_uu (Gauss-Jordan with partial pivoting,
 * standing in for tf.linalg.inv's LU, lqr.py:84), K = -inv Q_ux, k = -inv q_u,
 * the four-term V / v updates (:97-105) and the const recursion (:113-121).
 * The reference recomputes W == Q, w == q (:107-116); that duplicate work is
 * NOT repeated here, which makes this baseline faster than the reference's
 * own arithmetic would be.
 *
 * Layout (row-major, batch-major): F[B][n][d] f[B][n] C[B][d][d] c[B][d]
 * x0[B][n]; states[B][T+1][n] actions[B][T][m] costs[B][T+1];
 * optional K[B][T][m][n] k[B][T][m] V[B][T][n][n] v[B][T][n] cst[B][T].
 * A zero batch stride shares that operand between instances.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include <alloca.h>

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)

#define REAL float
#define NAME(x) CAT(x, f32)
#include "lqr_oracle_impl.h"
#undef REAL
#undef NAME

#define REAL double
#define NAME(x) CAT(x, f64)
#include "lqr_oracle_impl.h"
#undef REAL
#undef NAME

int lqr_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
