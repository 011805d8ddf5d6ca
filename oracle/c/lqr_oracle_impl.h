/* TEST INFRASTRUCTURE ONLY -- body of oracle/c/lqr_oracle.c, included once per
 * precision with REAL and NAME(x) defined.  See lqr_oracle.c for the contract. */
static int NAME(invert)(int m, const REAL *A, REAL *Ainv, REAL *work)
{
    /* work: m x 2m augmented [A | I] */
    int w = 2 * m;
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) {
            work[i * w + j] = A[i * m + j];
            work[i * w + m + j] = (i == j) ? (REAL)1 : (REAL)0;
        }
    for (int p = 0; p < m; ++p) {
        int piv = p;
        REAL best = (REAL)fabs((double)work[p * w + p]);
        for (int i = p + 1; i < m; ++i) {
            REAL a = (REAL)fabs((double)work[i * w + p]);
            if (a > best) { best = a; piv = i; }
        }
        if (best == (REAL)0) return -1;
        if (piv != p)
            for (int j = 0; j < w; ++j) {
                REAL t = work[p * w + j]; work[p * w + j] = work[piv * w + j];
                work[piv * w + j] = t;
            }
        REAL inv = (REAL)1 / work[p * w + p];
        for (int j = 0; j < w; ++j) work[p * w + j] *= inv;
        for (int i = 0; i < m; ++i) {
            if (i == p) continue;
            REAL fct = work[i * w + p];
            if (fct == (REAL)0) continue;
            for (int j = 0; j < w; ++j) work[i * w + j] -= fct * work[p * w + j];
        }
    }
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) Ainv[i * m + j] = work[i * w + m + j];
    return 0;
}

static int NAME(solve_one)(int n, int m, int T, const REAL *F, const REAL *f,
                               const REAL *C, const REAL *c, const REAL *x0, REAL *states,
                               REAL *actions, REAL *costs, REAL *Kout, REAL *kout,
                               REAL *Vout, REAL *vout, REAL *cout, REAL *ws)
{
    const int d = n + m;
    REAL *V = ws;            ws += n * n;
    REAL *v = ws;            ws += n;
    REAL *FtV = ws;          ws += d * n;
    REAL *Q = ws;            ws += d * d;
    REAL *q = ws;            ws += d;
    REAL *inv = ws;          ws += m * m;
    REAL *aug = ws;          ws += 2 * m * m;
    REAL *KtQuu = ws;        ws += n * m;
    REAL *Vn = ws;           ws += n * n;
    REAL *vn = ws;           ws += n;
    REAL *Vf = ws;           ws += n;
    REAL *z = ws;            ws += d;
    REAL *Kall = ws;         ws += (size_t)T * m * n;
    REAL *kall = ws;         ws += (size_t)T * m;
    REAL cst = 0;
    int status = 0;
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) V[i * n + j] = C[i * d + j];
        v[i] = c[i];
    }
    for (int t = T - 1; t >= 0; --t) {
        REAL *K = Kall + (size_t)t * m * n, *k = kall + (size_t)t * m;
        for (int a = 0; a < d; ++a)                       /* F^T V            :74 */
            for (int j = 0; j < n; ++j) {
                REAL s = 0;
                for (int i = 0; i < n; ++i) s += F[i * d + a] * V[i * n + j];
                FtV[a * n + j] = s;
            }
        for (int a = 0; a < d; ++a) {                     /* Q, q             :75-78 */
            for (int b = 0; b < d; ++b) {
                REAL s = 0;
                for (int j = 0; j < n; ++j) s += FtV[a * n + j] * F[j * d + b];
                Q[a * d + b] = C[a * d + b] + s;
            }
            REAL s1 = 0, s2 = 0;
            for (int j = 0; j < n; ++j) { s1 += FtV[a * n + j] * f[j]; s2 += F[j * d + a] * v[j]; }
            q[a] = c[a] + s1 + s2;
        }
        {                                                 /* inv(Q_uu)        :84 */
            REAL *Quu = (REAL *)alloca(sizeof(REAL) * m * m);
            for (int a = 0; a < m; ++a)
                for (int b = 0; b < m; ++b) Quu[a * m + b] = Q[(n + a) * d + n + b];
            if (NAME(invert)(m, Quu, inv, aug) != 0) status |= 1;
        }
        for (int a = 0; a < m; ++a) {                     /* K, k             :86-87 */
            for (int j = 0; j < n; ++j) {
                REAL s = 0;
                for (int b = 0; b < m; ++b) s += inv[a * m + b] * Q[(n + b) * d + j];
                K[a * n + j] = -s;
            }
            REAL s = 0;
            for (int b = 0; b < m; ++b) s += inv[a * m + b] * q[n + b];
            k[a] = -s;
        }
        for (int i = 0; i < n; ++i)                       /* K^T Q_uu         :95 */
            for (int b = 0; b < m; ++b) {
                REAL s = 0;
                for (int a = 0; a < m; ++a) s += K[a * n + i] * Q[(n + a) * d + n + b];
                KtQuu[i * m + b] = s;
            }
        for (int i = 0; i < n; ++i) {                     /* V, v             :97-105 */
            for (int j = 0; j < n; ++j) {
                REAL s1 = 0, s2 = 0, s3 = 0;
                for (int a = 0; a < m; ++a) {
                    s1 += Q[i * d + n + a] * K[a * n + j];
                    s2 += K[a * n + i] * Q[(n + a) * d + j];
                    s3 += KtQuu[i * m + a] * K[a * n + j];
                }
                Vn[i * n + j] = Q[i * d + j] + s1 + s2 + s3;
            }
            REAL s1 = 0, s2 = 0, s3 = 0;
            for (int a = 0; a < m; ++a) {
                s1 += Q[i * d + n + a] * k[a];
                s2 += K[a * n + i] * q[n + a];
                s3 += KtQuu[i * m + a] * k[a];
            }
            vn[i] = q[i] + s1 + s2 + s3;
        }
        {                                                 /* const            :113-121 */
            REAL c1 = 0, c2 = 0, c3a = 0, c3b = 0;
            for (int a = 0; a < m; ++a) {
                REAL s = 0;
                for (int b = 0; b < m; ++b) s += Q[(n + a) * d + n + b] * k[b];
                c1 += k[a] * s;
                c2 += k[a] * q[n + a];
            }
            for (int i = 0; i < n; ++i) {
                REAL s = 0;
                for (int j = 0; j < n; ++j) s += V[i * n + j] * f[j];
                Vf[i] = s;
            }
            for (int i = 0; i < n; ++i) { c3a += f[i] * Vf[i]; c3b += f[i] * v[i]; }
            cst += ((REAL)0.5 * c1 + c2 + ((REAL)0.5 * c3a + c3b));
        }
        memcpy(V, Vn, sizeof(REAL) * n * n);
        memcpy(v, vn, sizeof(REAL) * n);
        if (Kout) memcpy(Kout + (size_t)t * m * n, K, sizeof(REAL) * m * n);
        if (kout) memcpy(kout + (size_t)t * m, k, sizeof(REAL) * m);
        if (Vout) memcpy(Vout + (size_t)t * n * n, V, sizeof(REAL) * n * n);
        if (vout) memcpy(vout + (size_t)t * n, v, sizeof(REAL) * n);
        if (cout) cout[t] = cst;
    }
    /* forward :131-161 */
    for (int i = 0; i < n; ++i) { z[i] = x0[i]; states[i] = x0[i]; }
    for (int t = 0; t < T; ++t) {
        const REAL *K = Kall + (size_t)t * m * n, *k = kall + (size_t)t * m;
        for (int a = 0; a < m; ++a) {
            REAL s = 0;
            for (int j = 0; j < n; ++j) s += K[a * n + j] * z[j];
            z[n + a] = s + k[a];
            actions[(size_t)t * m + a] = z[n + a];
        }
        REAL quad = 0, lin = 0;
        for (int a = 0; a < d; ++a) {
            REAL s = 0;
            for (int b = 0; b < d; ++b) s += C[a * d + b] * z[b];
            quad += z[a] * s;
            lin += z[a] * c[a];
        }
        costs[t] = (REAL)0.5 * quad + lin;
        REAL *xn = states + (size_t)(t + 1) * n;
        for (int i = 0; i < n; ++i) {
            REAL s = 0;
            for (int b = 0; b < d; ++b) s += F[i * d + b] * z[b];
            xn[i] = s + f[i];
        }
        for (int i = 0; i < n; ++i) z[i] = xn[i];
    }
    {
        REAL quad = 0, lin = 0;
        for (int i = 0; i < n; ++i) {
            REAL s = 0;
            for (int j = 0; j < n; ++j) s += C[i * d + j] * z[j];
            quad += z[i] * s;
            lin += z[i] * c[i];
        }
        costs[T] = (REAL)0.5 * quad + lin;
    }
    return status;
}

int NAME(lqr_oracle_solve)(int B, int n, int m, int T, const REAL *F, long sF,
                               const REAL *f, long sf, const REAL *C, long sC, const REAL *c,
                               long sc, const REAL *x0, REAL *states, REAL *actions,
                               REAL *costs, REAL *K, REAL *k, REAL *V, REAL *v, REAL *cst,
                               int nthreads)
{
    const int d = n + m;
    size_t ws_elems = (size_t)n * n * 2 + n * 3 + d * n + d * d + d * 2 + 3 * m * m +
                      n * m + (size_t)T * m * (n + 1) + 64;
    int bad = 0;
    if (nthreads < 1) nthreads = 1;
    #pragma omp parallel num_threads(nthreads) reduction(|:bad)
    {
        REAL *ws = (REAL *)malloc(sizeof(REAL) * ws_elems);
        #pragma omp for schedule(static)
        for (int b = 0; b < B; ++b) {
            bad |= NAME(solve_one)(n, m, T, F + b * sF, f + b * sf, C + b * sC,
                                      c + b * sc, x0 + (size_t)b * n,
                                      states + (size_t)b * (T + 1) * n,
                                      actions + (size_t)b * T * m, costs + (size_t)b * (T + 1),
                                      K ? K + (size_t)b * T * m * n : 0,
                                      k ? k + (size_t)b * T * m : 0,
                                      V ? V + (size_t)b * T * n * n : 0,
                                      v ? v + (size_t)b * T * n : 0,
                                      cst ? cst + (size_t)b * T : 0, ws);
        }
        free(ws);
    }
    return bad;
}

