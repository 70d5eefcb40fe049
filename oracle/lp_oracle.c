/*
 * lp_oracle.c — CPU restatement of haskell-md2/SimplexMethod's dense-LP hot path.
 * TEST INFRASTRUCTURE ONLY (see lp_oracle.h for who may call it and for the
 * parity-pinning status of every function).
 *
 * Build: gcc -O2 -ffp-contract=off -mfma (see oracle/Makefile).  Every fused
 * multiply-add below is an explicit fma(); nothing else is contracted, so the
 * HIP kernels (built with -ffp-contract=off and the same explicit fma()s) can
 * be compared bit for bit.
 *
 * Eigen 3.4.0 (pinned at /root/reference/CMakeLists.txt:15) is absent from
 * /root/reference; where the reference calls into it the published algorithm
 * is restated here and the call site is cited.
 */
#include "lp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* small helpers                                                             */
/* ------------------------------------------------------------------------- */

#define AT(M, ld, i, j) ((M)[(size_t)(j) * (size_t)(ld) + (size_t)(i)]) /* column-major */

static void* xmalloc(size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (!p) abort();
    return p;
}

/* ------------------------------------------------------------------------- */
/* chain selection — SimplexSolover.h:153-161, :164-172, :181-192            */
/* ------------------------------------------------------------------------- */

int orc_chain_select(const double* v, const unsigned char* mask, int len, int want_max, double eps,
                     double* best_out) {
    int sel = -1;
    if (want_max) {
        double best = -INFINITY; /* :153 */
        for (int j = 0; j < len; ++j) {
            if (mask && !mask[j]) continue;
            if (v[j] > best + eps) { /* :157 */
                best = v[j];
                sel = j;
            }
        }
        if (best_out) *best_out = best;
    } else {
        double best = INFINITY; /* :164 / :181 */
        for (int j = 0; j < len; ++j) {
            if (mask && !mask[j]) continue;
            if (v[j] < best - eps) { /* :168 / :187 */
                best = v[j];
                sel = j;
            }
        }
        if (best_out) *best_out = best;
    }
    return sel;
}

/* ------------------------------------------------------------------------- */
/* Eigen::FullPivLU restated — call sites SimplexSolover.h:124-128           */
/* ------------------------------------------------------------------------- */

typedef struct {
    int n;
    double* lu;    /* n x n column-major, L (unit) below, U on/above diagonal */
    int* rowt;     /* row transpositions, step k swapped rows k <-> rowt[k]   */
    int* colt;     /* column transpositions                                    */
    int nonzero_pivots;
    double maxpivot;
} fullpivlu_t;

/* FullPivLU::compute: at step k the entry of largest absolute value of the
 * trailing block (first maximum in column-major visiting order, strict >) is
 * swapped to (k,k); multipliers are stored below the diagonal; right-looking
 * rank-1 update of the Schur complement; stops at an exactly zero pivot.      */
static void fullpivlu_compute(fullpivlu_t* f) {
    const int n = f->n;
    double* a = f->lu;
    f->nonzero_pivots = n;
    f->maxpivot = 0.0;
    for (int k = 0; k < n; ++k) {
        int prow = k, pcol = k;
        double biggest = -1.0;
        for (int j = k; j < n; ++j) {
            const double* col = &AT(a, n, 0, j);
            for (int i = k; i < n; ++i) {
                double s = fabs(col[i]);
                if (s > biggest) {
                    biggest = s;
                    prow = i;
                    pcol = j;
                }
            }
        }
        if (biggest == 0.0) {
            f->nonzero_pivots = k;
            for (int i = k; i < n; ++i) {
                f->rowt[i] = i;
                f->colt[i] = i;
            }
            break;
        }
        if (biggest > f->maxpivot) f->maxpivot = biggest;
        f->rowt[k] = prow;
        f->colt[k] = pcol;
        if (prow != k) {
            for (int j = 0; j < n; ++j) {
                double t = AT(a, n, k, j);
                AT(a, n, k, j) = AT(a, n, prow, j);
                AT(a, n, prow, j) = t;
            }
        }
        if (pcol != k) {
            double* c0 = &AT(a, n, 0, k);
            double* c1 = &AT(a, n, 0, pcol);
            for (int i = 0; i < n; ++i) {
                double t = c0[i];
                c0[i] = c1[i];
                c1[i] = t;
            }
        }
        if (k < n - 1) {
            double piv = AT(a, n, k, k);
            double* lk = &AT(a, n, 0, k);
            for (int i = k + 1; i < n; ++i) lk[i] /= piv;
            for (int j = k + 1; j < n; ++j) {
                double* cj = &AT(a, n, 0, j);
                double ukj = cj[k];
                for (int i = k + 1; i < n; ++i) cj[i] -= lk[i] * ukj;
            }
        }
    }
}

/* FullPivLU::rank / isInvertible with the default threshold
 * epsilon * diagonalSize (what SimplexSolover.h:125 relies on).               */
static int fullpivlu_invertible(const fullpivlu_t* f) {
    const int n = f->n;
    const double thr = fabs(f->maxpivot) * (DBL_EPSILON * (double)n);
    int rank = 0;
    for (int i = 0; i < f->nonzero_pivots; ++i)
        if (fabs(AT(f->lu, n, i, i)) > thr) ++rank;
    return rank == n;
}

/* FullPivLU::inverse = solve(Identity): c = P*I; unit-lower solve; upper solve;
 * result = Q*c (column-oriented substitutions, as Eigen's col-major
 * triangular solver does).  out is n x n column-major.                         */
static void fullpivlu_inverse(const fullpivlu_t* f, double* out) {
    const int n = f->n;
    const double* a = f->lu;
    int* perm = (int*)xmalloc(sizeof(int) * (size_t)n); /* row i of P*I is row perm[i] of I */
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int k = 0; k < n; ++k) {
        int t = perm[k];
        perm[k] = perm[f->rowt[k]];
        perm[f->rowt[k]] = t;
    }
    double* cvec = (double*)xmalloc(sizeof(double) * (size_t)n);
    double* tmp = (double*)xmalloc(sizeof(double) * (size_t)n * (size_t)n);
    for (int j = 0; j < n; ++j) {
        for (int i = 0; i < n; ++i) cvec[i] = (perm[i] == j) ? 1.0 : 0.0;
        for (int k = 0; k < n; ++k) { /* L y = c, unit diagonal */
            double yk = cvec[k];
            if (yk != 0.0) {
                const double* lk = &AT(a, n, 0, k);
                for (int i = k + 1; i < n; ++i) cvec[i] -= yk * lk[i];
            }
        }
        for (int k = n - 1; k >= 0; --k) { /* U z = y */
            const double* uk = &AT(a, n, 0, k);
            cvec[k] /= uk[k];
            double zk = cvec[k];
            if (zk != 0.0)
                for (int i = 0; i < k; ++i) cvec[i] -= zk * uk[i];
        }
        memcpy(&AT(tmp, n, 0, j), cvec, sizeof(double) * (size_t)n);
    }
    /* undo the column transpositions: x = Q * z (apply in reverse order to rows) */
    for (int j = 0; j < n; ++j) {
        double* col = &AT(tmp, n, 0, j);
        for (int k = n - 1; k >= 0; --k) {
            int q = f->colt[k];
            if (q != k) {
                double t = col[k];
                col[k] = col[q];
                col[q] = t;
            }
        }
    }
    memcpy(out, tmp, sizeof(double) * (size_t)n * (size_t)n);
    free(tmp);
    free(cvec);
    free(perm);
}

/* ------------------------------------------------------------------------- */
/* reference-shaped revised simplex — SimplexSolover.h:97-209, :408-451      */
/* ------------------------------------------------------------------------- */

typedef struct {
    int m, n;
    const double *A, *b, *c;
    double* B;     /* m x m */
    double* Binv;  /* m x m */
    double* x;     /* n */
    double* xB;    /* m */
    fullpivlu_t lu;
} refsolver_t;

/* computeBFS, SimplexSolover.h:117-133 (+ basisMatrix :110-115).              */
static int ref_compute_bfs(refsolver_t* s, const int* N) {
    const int m = s->m, n = s->n;
    for (int t = 0; t < m; ++t) /* :113 */
        memcpy(&AT(s->B, m, 0, t), &AT(s->A, m, 0, N[t]), sizeof(double) * (size_t)m);
    memcpy(s->lu.lu, s->B, sizeof(double) * (size_t)m * (size_t)m);
    fullpivlu_compute(&s->lu);                          /* :124 */
    if (!fullpivlu_invertible(&s->lu)) return ORC_SINGULAR; /* :125-126 */
    fullpivlu_inverse(&s->lu, s->Binv);                 /* :128 */
    for (int i = 0; i < m; ++i) s->xB[i] = 0.0;         /* :129  xB = Binv * b */
    for (int j = 0; j < m; ++j) {
        const double bj = s->b[j];
        const double* col = &AT(s->Binv, m, 0, j);
        for (int i = 0; i < m; ++i) s->xB[i] += col[i] * bj;
    }
    for (int j = 0; j < n; ++j) s->x[j] = 0.0;          /* :131 */
    for (int t = 0; t < m; ++t) s->x[N[t]] = s->xB[t];  /* :132 */
    return ORC_OPTIMAL;
}

int orc_simplex_reference(const double* A, int m, int n, const double* b, const double* c,
                          const int* basis_in, int maximize, int n_orig, double eps, int max_iter,
                          int dense_eta_product, double* x_out, int* basis_out, double* obj_out,
                          int* iters_out, int* trace_enter, int* trace_leave, int trace_cap) {
    /* Canonical's ctor checks, Canonical.cpp:27-46, and SetOriginalVariablesCount :156-163 */
    if (m <= 0 || n < m || !A || !b || !c || !basis_in) return ORC_BAD_ARG;
    if (n_orig <= 0 || n_orig > n) return ORC_BAD_ARG;
    for (int t = 0; t < m; ++t)
        if (basis_in[t] < 0 || basis_in[t] >= n) return ORC_BAD_ARG;

    refsolver_t s;
    s.m = m; s.n = n; s.A = A; s.b = b; s.c = c;
    s.B = (double*)xmalloc(sizeof(double) * (size_t)m * m);
    s.Binv = (double*)xmalloc(sizeof(double) * (size_t)m * m);
    s.x = (double*)xmalloc(sizeof(double) * (size_t)n);
    s.xB = (double*)xmalloc(sizeof(double) * (size_t)m);
    s.lu.n = m;
    s.lu.lu = (double*)xmalloc(sizeof(double) * (size_t)m * m);
    s.lu.rowt = (int*)xmalloc(sizeof(int) * (size_t)m);
    s.lu.colt = (int*)xmalloc(sizeof(int) * (size_t)m);
    int* N = (int*)xmalloc(sizeof(int) * (size_t)m);
    double* yT = (double*)xmalloc(sizeof(double) * (size_t)m);
    double* u = (double*)xmalloc(sizeof(double) * (size_t)m);
    double* xB = (double*)xmalloc(sizeof(double) * (size_t)m);
    double* F = dense_eta_product ? (double*)xmalloc(sizeof(double) * (size_t)m * m) : NULL;
    double* FB = dense_eta_product ? (double*)xmalloc(sizeof(double) * (size_t)m * m) : NULL;
    unsigned char* inN = (unsigned char*)xmalloc((size_t)n);
    memcpy(N, basis_in, sizeof(int) * (size_t)m); /* :419 */

    int status = ref_compute_bfs(&s, N); /* :423 */
    int iteration = 0;
    int done = (status != ORC_OPTIMAL);
    while (!done && iteration < max_iter) { /* :429 */
        /* ---- simplexIter :135-209 ---- */
        for (int j = 0; j < m; ++j) { /* :144-146  yT = cB * Binv */
            const double* col = &AT(s.Binv, m, 0, j);
            double acc = 0.0;
            for (int t = 0; t < m; ++t) acc += c[N[t]] * col[t];
            yT[j] = acc;
        }
        memset(inN, 0, (size_t)n); /* complement :97-108 */
        for (int t = 0; t < m; ++t) inN[N[t]] = 1;
        int enter = -1;
        double best = maximize ? -INFINITY : INFINITY; /* :153 / :164 */
        for (int j = 0; j < n; ++j) {
            if (inN[j]) continue;
            const double* aj = &AT(A, m, 0, j);
            double dot = 0.0;
            for (int i = 0; i < m; ++i) dot += yT[i] * aj[i];
            double d = c[j] - dot; /* :156 / :167 */
            if (maximize) {
                if (d > best + eps) { best = d; enter = j; } /* :157-160 */
            } else {
                if (d < best - eps) { best = d; enter = j; } /* :168-171 */
            }
        }
        int optimal = maximize ? (best <= eps) : (best >= -eps); /* :162 / :173 */
        if (optimal) {
            status = ref_compute_bfs(&s, N); /* :433 */
            break;                           /* :435-439 */
        }
        { /* :176  u = Binv * A.col(enter) ; :177 xB = Binv * b */
            const double* ae = &AT(A, m, 0, enter);
            for (int i = 0; i < m; ++i) { u[i] = 0.0; xB[i] = 0.0; }
            for (int j = 0; j < m; ++j) {
                const double* col = &AT(s.Binv, m, 0, j);
                const double aej = ae[j], bj = b[j];
                for (int i = 0; i < m; ++i) {
                    u[i] += col[i] * aej;
                    xB[i] += col[i] * bj;
                }
            }
        }
        int any_pos = 0; /* :179 */
        for (int i = 0; i < m; ++i) if (!(u[i] <= eps)) any_pos = 1;
        if (!any_pos) { status = ORC_UNBOUNDED; break; }
        double theta = INFINITY; /* :181 */
        int leave_pos = -1;
        for (int i = 0; i < m; ++i) { /* :184-192 */
            if (u[i] > eps) {
                double r = xB[i] / u[i];
                if (r < theta - eps) { theta = r; leave_pos = i; }
            }
        }
        if (leave_pos == -1) { status = ORC_UNBOUNDED; break; } /* :194 */
        if (iteration < trace_cap) {
            if (trace_enter) trace_enter[iteration] = enter;
            if (trace_leave) trace_leave[iteration] = leave_pos;
        }
        N[leave_pos] = enter; /* :196 */
        if (dense_eta_product) { /* :198-206, result discarded by :446 */
            for (size_t k = 0; k < (size_t)m * m; ++k) F[k] = 0.0;
            for (int i = 0; i < m; ++i) AT(F, m, i, i) = 1.0;
            for (int i = 0; i < m; ++i)
                if (i != leave_pos) AT(F, m, i, leave_pos) = -u[i] / u[leave_pos];
            AT(F, m, leave_pos, leave_pos) = 1.0 / u[leave_pos];
            for (int j = 0; j < m; ++j) {
                double* out = &AT(FB, m, 0, j);
                for (int i = 0; i < m; ++i) out[i] = 0.0;
                for (int k = 0; k < m; ++k) {
                    const double bkj = AT(s.Binv, m, k, j);
                    const double* fk = &AT(F, m, 0, k);
                    for (int i = 0; i < m; ++i) out[i] += fk[i] * bkj;
                }
            }
            memcpy(s.Binv, FB, sizeof(double) * (size_t)m * m);
        }
        status = ref_compute_bfs(&s, N); /* :446 */
        if (status != ORC_OPTIMAL) break;
        ++iteration; /* :447 */
        if (iteration >= max_iter) { status = ORC_ITER_LIMIT; break; } /* :450 */
    }
    if (!done && max_iter <= 0) status = ORC_ITER_LIMIT;

    if (status == ORC_OPTIMAL) {
        for (int j = 0; j < n_orig; ++j) x_out[j] = s.x[j]; /* :435-438 */
        if (obj_out) { /* Canonical::Evaluate, Canonical.cpp:86, on the full vertex */
            double z = 0.0;
            for (int j = 0; j < n; ++j) z += c[j] * s.x[j];
            *obj_out = z;
        }
    }
    if (basis_out) memcpy(basis_out, N, sizeof(int) * (size_t)m);
    if (iters_out) *iters_out = iteration;

    free(inN); free(FB); free(F); free(xB); free(u); free(yT); free(N);
    free(s.lu.colt); free(s.lu.rowt); free(s.lu.lu);
    free(s.xB); free(s.x); free(s.Binv); free(s.B);
    return status;
}

/* ------------------------------------------------------------------------- */
/* tableau simplex — same rules, the form the GPU executes                   */
/* ------------------------------------------------------------------------- */

/* One Gauss-Jordan pivot on row r, column e of the (rows x ld) row-major
 * tableau T: the elementwise form of Binv = F*Binv, SimplexSolover.h:198-206
 * (F(i,r) = -u_i/u_r, F(r,r) = 1/u_r).  Column e is set to the exact unit
 * vector.                                                                       */
static void tableau_pivot(double* T, int rows, int cols, int ld, int r, int e) {
    const double ur = T[(size_t)r * ld + e];
    const double inv = 1.0 / ur; /* :204 */
    double* Tr = T + (size_t)r * ld;
    for (int i = 0; i < rows; ++i) {
        if (i == r) continue;
        double* Ti = T + (size_t)i * ld;
        const double l = -Ti[e] / ur; /* :201 */
        for (int j = 0; j < cols; ++j) Ti[j] = fma(l, Tr[j], Ti[j]);
        Ti[e] = 0.0;
    }
    for (int j = 0; j < cols; ++j) Tr[j] = Tr[j] * inv;
    Tr[e] = 1.0;
}

/* The pivot loop of solveWithBasis (:429-450) on a tableau T ((m+1) x (n+1), row stride ld, rows by
 * basis position, row m = reduced costs, column n = xB) whose basic columns are unit vectors.
 * Columns j >= n_enter never enter the basis (phase II of the two-phase flow bars the artificial
 * columns this way); n_enter = n is the plain solve.                                             */
static int tableau_loop(double* T, int m, int n, int ld, int* N, int n_enter, int maximize, double eps,
                        int max_iter, int* iteration_io, int* trace_enter, int* trace_leave, int trace_cap,
                        unsigned char* nonbasic, unsigned char* rowmask, double* ratio) {
    const int rows = m + 1, cols = n + 1;
    int status = ORC_OPTIMAL;
    int iteration = *iteration_io;
    if (max_iter <= 0) status = ORC_ITER_LIMIT;
    while (status == ORC_OPTIMAL) {
        memset(nonbasic, 1, (size_t)n); /* complement :97-108 */
        for (int t = 0; t < m; ++t) nonbasic[N[t]] = 0;
        for (int j = n_enter; j < n; ++j) nonbasic[j] = 0;
        double best;
        const double* d = T + (size_t)m * ld;
        int enter = orc_chain_select(d, nonbasic, n, maximize, eps, &best); /* :152-174 */
        int optimal = maximize ? (best <= eps) : (best >= -eps);
        if (optimal) break;
        int any_pos = 0; /* :179 */
        for (int i = 0; i < m; ++i) {
            double ui = T[(size_t)i * ld + enter];
            if (!(ui <= eps)) any_pos = 1;
            rowmask[i] = (ui > eps);
            ratio[i] = rowmask[i] ? T[(size_t)i * ld + n] / ui : 0.0; /* :186 */
        }
        if (!any_pos) { status = ORC_UNBOUNDED; break; }
        int leave_pos = orc_chain_select(ratio, rowmask, m, 0, eps, NULL); /* :181-192 */
        if (leave_pos < 0) { status = ORC_UNBOUNDED; break; }
        if (iteration < trace_cap) {
            if (trace_enter) trace_enter[iteration] = enter;
            if (trace_leave) trace_leave[iteration] = leave_pos;
        }
        N[leave_pos] = enter; /* :196 */
        tableau_pivot(T, rows, cols, ld, leave_pos, enter);
        ++iteration;
        if (iteration >= max_iter) { status = ORC_ITER_LIMIT; break; } /* :450 */
    }
    *iteration_io = iteration;
    return status;
}

int orc_simplex_tableau(const double* A, int m, int n, const double* b, const double* c,
                        const int* basis_in, int maximize, int n_orig, double eps, int max_iter,
                        double* x_out, int* basis_out, double* obj_out, int* iters_out,
                        int* trace_enter, int* trace_leave, int trace_cap, double* tableau_out) {
    if (m <= 0 || n < m || !A || !b || !c || !basis_in) return ORC_BAD_ARG;
    if (n_orig <= 0 || n_orig > n) return ORC_BAD_ARG;
    for (int t = 0; t < m; ++t)
        if (basis_in[t] < 0 || basis_in[t] >= n) return ORC_BAD_ARG;

    const int rows = m + 1, cols = n + 1, ld = cols;
    double* T = (double*)xmalloc(sizeof(double) * (size_t)rows * ld);
    int* N = (int*)xmalloc(sizeof(int) * (size_t)m);
    unsigned char* nonbasic = (unsigned char*)xmalloc((size_t)n);
    unsigned char* rowmask = (unsigned char*)xmalloc((size_t)m);
    double* ratio = (double*)xmalloc(sizeof(double) * (size_t)m);
    memcpy(N, basis_in, sizeof(int) * (size_t)m);

    for (int i = 0; i < m; ++i) {
        for (int j = 0; j < n; ++j) T[(size_t)i * ld + j] = AT(A, m, i, j);
        T[(size_t)i * ld + n] = b[i];
    }
    for (int j = 0; j < n; ++j) T[(size_t)m * ld + j] = c[j];
    T[(size_t)m * ld + n] = 0.0;

    int status = ORC_OPTIMAL;

    /* computeBFS at :423 in tableau form.  If the basis columns are exactly the
     * unit vectors e_t in order (Symmetrical::ToCanonical's slack basis,
     * Symmetrical.cpp:169-188) T = [A|b] already; otherwise m Gauss-Jordan pivots
     * with partial (row) pivoting bring column N(t) to a unit vector, then rows
     * are put in basis-position order.                                           */
    int identity = 1;
    for (int t = 0; t < m && identity; ++t)
        for (int i = 0; i < m; ++i)
            if (AT(A, m, i, N[t]) != ((i == t) ? 1.0 : 0.0)) { identity = 0; break; }
    for (int t = 0; t < m && identity; ++t)
        if (c[N[t]] != 0.0) identity = 0; /* reduced-cost row needs elimination */
    if (!identity) {
        int* rowpos = (int*)xmalloc(sizeof(int) * (size_t)m);
        unsigned char* used = (unsigned char*)xmalloc((size_t)m);
        memset(used, 0, (size_t)m);
        double minp = INFINITY, maxp = 0.0;
        for (int t = 0; t < m; ++t) {
            const int q = N[t];
            int p = -1;
            double big = -1.0;
            for (int i = 0; i < m; ++i) {
                if (used[i]) continue;
                double a = fabs(T[(size_t)i * ld + q]);
                if (a > big) { big = a; p = i; }
            }
            if (!(big > 0.0)) { status = ORC_SINGULAR; break; }
            if (big < minp) minp = big;
            if (big > maxp) maxp = big;
            tableau_pivot(T, rows, cols, ld, p, q);
            used[p] = 1;
            rowpos[t] = p;
        }
        if (status == ORC_OPTIMAL && minp <= DBL_EPSILON * (double)m * maxp) status = ORC_SINGULAR;
        if (status == ORC_OPTIMAL) {
            double* T2 = (double*)xmalloc(sizeof(double) * (size_t)rows * ld);
            for (int t = 0; t < m; ++t)
                memcpy(T2 + (size_t)t * ld, T + (size_t)rowpos[t] * ld, sizeof(double) * (size_t)ld);
            memcpy(T2 + (size_t)m * ld, T + (size_t)m * ld, sizeof(double) * (size_t)ld);
            free(T);
            T = T2;
        }
        free(used);
        free(rowpos);
    }

    int iteration = 0;
    if (status == ORC_OPTIMAL)
        status = tableau_loop(T, m, n, ld, N, n, maximize, eps, max_iter, &iteration, trace_enter, trace_leave,
                              trace_cap, nonbasic, rowmask, ratio);

    if (status == ORC_OPTIMAL) {
        double* x = (double*)xmalloc(sizeof(double) * (size_t)n);
        for (int j = 0; j < n; ++j) x[j] = 0.0;
        for (int t = 0; t < m; ++t) x[N[t]] = T[(size_t)t * ld + n];
        for (int j = 0; j < n_orig; ++j) x_out[j] = x[j];
        if (obj_out) {
            double z = 0.0;
            for (int j = 0; j < n; ++j) z += c[j] * x[j];
            *obj_out = z;
        }
        free(x);
    }
    if (basis_out) memcpy(basis_out, N, sizeof(int) * (size_t)m);
    if (iters_out) *iters_out = iteration;
    if (tableau_out) memcpy(tableau_out, T, sizeof(double) * (size_t)rows * ld);
    free(ratio); free(rowmask); free(nonbasic); free(N); free(T);
    return status;
}

/* ------------------------------------------------------------------------- */
/* two-phase simplex — the flow the reference sketches in its */
/* unreachable code (SimplexSolover.h:61-95 make_b_nonneg/createAuxiliaryProblem, */
/* :331-381 replaceArtificialColumns, :383-406 twoPhaseSimplex), made consistent */
/* and built from the tableau simplex above.  Build-defined, parity unpinned.   */
/* ------------------------------------------------------------------------- */
int orc_two_phase(const double* A, int m, int n, const double* b, const double* c, int maximize,
                  int n_orig, double eps, int max_iter, double* x_out, int* basis_out,
                  double* obj_out, int* iters_out /* 3: phase I, drive-out, phase II */) {
    if (m <= 0 || n < m || !A || !b || !c || !x_out) return ORC_BAD_ARG;
    if (n_orig <= 0 || n_orig > n) return ORC_BAD_ARG;
    const int na = n + m;
    double* A1 = (double*)xmalloc(sizeof(double) * (size_t)m * na); /* [A' | I], column-major */
    double* b1 = (double*)xmalloc(sizeof(double) * (size_t)m);
    double* c1 = (double*)xmalloc(sizeof(double) * (size_t)na);
    double* xa = (double*)xmalloc(sizeof(double) * (size_t)na);
    int* N = (int*)xmalloc(sizeof(int) * (size_t)m);
    double* T = (double*)xmalloc(sizeof(double) * (size_t)(m + 1) * (na + 1));
    int it[3] = {0, 0, 0};
    /* make_b_nonneg, :61-68: rows with b < -EPS change sign */
    for (int i = 0; i < m; ++i) {
        const int flip = b[i] < -eps;
        b1[i] = flip ? -b[i] : b[i];
        for (int j = 0; j < n; ++j) A1[(size_t)j * m + i] = flip ? -AT(A, m, i, j) : AT(A, m, i, j);
        for (int j = 0; j < m; ++j) A1[(size_t)(n + j) * m + i] = (i == j) ? 1.0 : 0.0;
    }
    /* createAuxiliaryProblem, :70-95: minimise the sum of the artificials from their basis */
    for (int j = 0; j < na; ++j) c1[j] = (j < n) ? 0.0 : 1.0;
    for (int t = 0; t < m; ++t) N[t] = n + t;
    int status = orc_simplex_tableau(A1, m, na, b1, c1, N, 0, na, eps, max_iter, xa, N, NULL, &it[0],
                                     NULL, NULL, 0, T);
    if (status == ORC_OPTIMAL) {
        double sum = 0.0; /* :347-350 */
        for (int i = 0; i < m; ++i) sum += xa[n + i];
        if (sum > eps) status = ORC_INFEASIBLE; /* :352-353 */
    }
    if (status == ORC_OPTIMAL) {
        /* replaceArtificialColumns, :331-381: an artificial still basic (at level 0) leaves for the
         * first non-basic original column with |T[pos][cand]| > EPS; none = dependent rows        */
        const int ld = na + 1;
        unsigned char* basic = (unsigned char*)xmalloc((size_t)na);
        for (int pos = 0; pos < m && status == ORC_OPTIMAL; ++pos) {
            if (N[pos] < n) continue;
            memset(basic, 0, (size_t)na);
            for (int t = 0; t < m; ++t) basic[N[t]] = 1;
            int cand = -1;
            for (int j = 0; j < n; ++j)
                if (!basic[j] && fabs(T[(size_t)pos * ld + j]) > eps) { cand = j; break; }
            if (cand < 0) { status = ORC_SINGULAR; break; } /* :372-380 */
            tableau_pivot(T, m + 1, na + 1, ld, pos, cand);
            N[pos] = cand;
            ++it[1];
        }
        free(basic);
    }
    if (status == ORC_OPTIMAL) {
        /* Phase II (:383-404) CONTINUES on the phase-I tableau instead of re-inverting the basis from
         * [A' | b'] (what the reference's sketch does through computeBFS): the reduced-cost row is set
         * to the original costs (0 for the artificial columns and the right-hand side) and priced out
         * over the current basis with the crash step's arithmetic — m Gauss-Jordan pivots whose pivot
         * elements are the 1s of the basic unit columns, so they only touch row m; the artificial
         * columns stay in the tableau but never enter again.                                        */
        const int ld = na + 1;
        for (int j = 0; j <= na; ++j) T[(size_t)m * ld + j] = (j < n) ? c[j] : 0.0;
        for (int t = 0; t < m; ++t) tableau_pivot(T, m + 1, na + 1, ld, t, N[t]);
        unsigned char* nonbasic = (unsigned char*)xmalloc((size_t)na);
        unsigned char* rowmask = (unsigned char*)xmalloc((size_t)m);
        double* ratio = (double*)xmalloc(sizeof(double) * (size_t)m);
        status = tableau_loop(T, m, na, ld, N, n, maximize, eps, max_iter, &it[2], NULL, NULL, 0, nonbasic,
                              rowmask, ratio);
        free(ratio); free(rowmask); free(nonbasic);
        if (status == ORC_OPTIMAL) {
            for (int j = 0; j < na; ++j) xa[j] = 0.0;
            for (int t = 0; t < m; ++t) xa[N[t]] = T[(size_t)t * ld + na];
            for (int j = 0; j < n_orig; ++j) x_out[j] = xa[j];
            if (obj_out) { /* Canonical::Evaluate, Canonical.cpp:86 */
                double z = 0.0;
                for (int j = 0; j < n; ++j) z += c[j] * xa[j];
                *obj_out = z;
            }
        }
    }
    if (basis_out) memcpy(basis_out, N, sizeof(int) * (size_t)m);
    if (iters_out) memcpy(iters_out, it, sizeof(it));
    free(T); free(N); free(xa); free(c1); free(b1); free(A1);
    return status;
}

/* ------------------------------------------------------------------------- */
/* per-basis solve — Canonical.cpp:165-197 (ColPivHouseholderQR restated)    */
/* ------------------------------------------------------------------------- */

int orc_basic_solution(const double* A, int m, int n, const double* b, const int* basis,
                       double* x_out) {
    if (m <= 0 || n < m) return ORC_BAD_ARG;
    for (int t = 0; t < m; ++t)
        if (basis[t] < 0 || basis[t] >= n) return ORC_BAD_ARG;
    double* Q = (double*)xmalloc(sizeof(double) * (size_t)m * m); /* working copy of B */
    double* rhs = (double*)xmalloc(sizeof(double) * (size_t)m);
    double* norms = (double*)xmalloc(sizeof(double) * (size_t)m);
    int* perm = (int*)xmalloc(sizeof(int) * (size_t)m);
    for (int t = 0; t < m; ++t) /* Canonical.cpp:184-187 */
        memcpy(&AT(Q, m, 0, t), &AT(A, m, 0, basis[t]), sizeof(double) * (size_t)m);
    memcpy(rhs, b, sizeof(double) * (size_t)m);
    for (int j = 0; j < m; ++j) {
        perm[j] = j;
        double s = 0.0;
        for (int i = 0; i < m; ++i) s += AT(Q, m, i, j) * AT(Q, m, i, j);
        norms[j] = s;
    }
    int rank = m;
    double maxpivot = 0.0;
    for (int k = 0; k < m; ++k) {
        /* column of largest remaining squared norm (first maximum) */
        int best = k;
        double bn = -1.0;
        for (int j = k; j < m; ++j) {
            double s = 0.0;
            for (int i = k; i < m; ++i) s += AT(Q, m, i, j) * AT(Q, m, i, j);
            norms[j] = s;
            if (s > bn) { bn = s; best = j; }
        }
        if (best != k) {
            for (int i = 0; i < m; ++i) {
                double t = AT(Q, m, i, k);
                AT(Q, m, i, k) = AT(Q, m, i, best);
                AT(Q, m, i, best) = t;
            }
            int tp = perm[k]; perm[k] = perm[best]; perm[best] = tp;
        }
        /* Householder vector for column k (Eigen's makeHouseholder convention:
         * no reflection when the tail is exactly zero)                          */
        double tail = 0.0;
        for (int i = k + 1; i < m; ++i) tail += AT(Q, m, i, k) * AT(Q, m, i, k);
        double c0 = AT(Q, m, k, k);
        double beta, tau;
        if (tail == 0.0) {
            tau = 0.0;
            beta = c0;
        } else {
            beta = sqrt(c0 * c0 + tail);
            if (c0 >= 0.0) beta = -beta;
            for (int i = k + 1; i < m; ++i) AT(Q, m, i, k) /= (c0 - beta);
            tau = (beta - c0) / beta;
        }
        AT(Q, m, k, k) = beta;
        if (fabs(beta) > maxpivot) maxpivot = fabs(beta);
        if (tau != 0.0) {
            for (int j = k + 1; j < m; ++j) { /* apply H = I - tau v v^T to the trailing columns */
                double w = AT(Q, m, k, j);
                for (int i = k + 1; i < m; ++i) w += AT(Q, m, i, k) * AT(Q, m, i, j);
                w *= tau;
                AT(Q, m, k, j) -= w;
                for (int i = k + 1; i < m; ++i) AT(Q, m, i, j) -= w * AT(Q, m, i, k);
            }
            double w = rhs[k]; /* and to the right-hand side */
            for (int i = k + 1; i < m; ++i) w += AT(Q, m, i, k) * rhs[i];
            w *= tau;
            rhs[k] -= w;
            for (int i = k + 1; i < m; ++i) rhs[i] -= w * AT(Q, m, i, k);
        }
    }
    /* ColPivHouseholderQR::rank with the default threshold eps*size: the solve
     * uses only the leading `rank` pivots (truncated solution for singular B —
     * SURVEY.md §8 row E2)                                                      */
    {
        const double thr = maxpivot * DBL_EPSILON * (double)m;
        rank = 0;
        for (int k = 0; k < m; ++k)
            if (fabs(AT(Q, m, k, k)) > thr) ++rank;
    }
    double* y = (double*)xmalloc(sizeof(double) * (size_t)m);
    for (int i = 0; i < m; ++i) y[i] = 0.0;
    for (int k = rank - 1; k >= 0; --k) {
        double s = rhs[k];
        for (int j = k + 1; j < rank; ++j) s -= AT(Q, m, k, j) * y[j];
        y[k] = s / AT(Q, m, k, k);
    }
    for (int j = 0; j < n; ++j) x_out[j] = 0.0; /* Canonical.cpp:181 */
    for (int k = 0; k < m; ++k) x_out[basis[perm[k]]] = y[k]; /* :191-194 */
    free(y); free(perm); free(norms); free(rhs); free(Q);
    return ORC_OPTIMAL;
}

int orc_is_feasible_basis(const double* A, int m, int n, const double* b, const int* basis) {
    double* x = (double*)xmalloc(sizeof(double) * (size_t)n);
    int ok = (orc_basic_solution(A, m, n, b, basis, x) == ORC_OPTIMAL);
    for (int i = 0; ok && i < n; ++i)
        if (x[i] < -1e-9) ok = 0; /* Canonical.cpp:171 */
    free(x);
    return ok;
}

double orc_evaluate(const double* c, const double* x, int n) { /* Canonical.cpp:86 */
    double z = 0.0;
    for (int j = 0; j < n; ++j) z += c[j] * x[j];
    return z;
}

/* ------------------------------------------------------------------------- */
/* enumeration — build-defined semantics               */
/* ------------------------------------------------------------------------- */

uint64_t orc_binom(int n, int k) {
    if (k < 0 || k > n) return 0;
    if (k > n - k) k = n - k;
    unsigned __int128 r = 1;
    for (int i = 1; i <= k; ++i) {
        r = r * (unsigned)(n - k + i) / (unsigned)i;
        if (r > (unsigned __int128)UINT64_MAX) return 0;
    }
    return (uint64_t)r;
}

void orc_unrank(int n, int m, uint64_t rank, int* subset) {
    int a = 0;
    for (int t = 0; t < m; ++t) {
        for (int j = a;; ++j) {
            uint64_t cnt = orc_binom(n - 1 - j, m - 1 - t);
            if (rank < cnt) {
                subset[t] = j;
                a = j + 1;
                break;
            }
            rank -= cnt;
        }
    }
}

uint64_t orc_rank(int n, int m, const int* subset) {
    uint64_t r = 0;
    int a = 0;
    for (int t = 0; t < m; ++t) {
        for (int j = a; j < subset[t]; ++j) r += orc_binom(n - 1 - j, m - 1 - t);
        a = subset[t] + 1;
    }
    return r;
}

int orc_next_subset(int n, int m, int* subset) {
    int t = m - 1;
    while (t >= 0 && subset[t] == n - m + t) --t;
    if (t < 0) return 0;
    ++subset[t];
    for (int s = t + 1; s < m; ++s) subset[s] = subset[s - 1] + 1;
    return 1;
}

/* Per-subset solve of [A[:,S] | b] (m x (m+1), row-major W), columns in ascending order.
 * Exact operation order (the HIP kernels replay it, also across shared prefixes of
 * consecutive subsets — everything up to step m-3 depends only on S_0..S_{m-3}):
 *
 *   Gauss-Jordan with partial (row) pivoting on the first g = max(m-2, 0) columns:
 *   for t = 0..g-1:
 *     p   = first row, among rows not yet used as a pivot, of largest |W[i][t]|
 *     piv = W[p][t];  piv == 0 -> singular
 *     inv = 1/piv
 *     every row i != p (used or not): l = -(W[i][t]*inv);
 *                                      W[i][c] = fma(l, W[p][c], W[i][c])  for c > t
 *     row p:                           W[p][c] = W[p][c]*inv               for c > t
 *
 *   2x2 block on the last two columns a = m-2, b = m-1 and the two unused rows r1 < r2
 *   (elimination inside the block, then back-substitution into the used rows):
 *     p = (|W[r2][a]| > |W[r1][a]|) ? r2 : r1 ; q = the other;  piv1 = W[p][a]; inv1 = 1/piv1
 *     l   = -(W[q][a]*inv1);  wqb = fma(l, W[p][b], W[q][b]);  rq = fma(l, rhs[p], rhs[q])
 *     piv2 = wqb; inv2 = 1/piv2
 *     x_b = rq*inv2;          x_a = fma(-W[p][b], x_b, rhs[p]) * inv1
 *     used row i (pivot row of step t): x(S_t) = fma(-W[i][b], x_b, fma(-W[i][a], x_a, rhs[i]))
 *   (m == 1: the single pivot piv = W[0][0]; x = rhs[0] * (1/piv).)
 *
 *   singular iff some pivot is 0 or min|piv| <= DBL_EPSILON * m * max|piv| over all m pivots
 *   feasible iff every x >= -1e-9 (a NaN is infeasible)
 *   z = fma(c[S_t], x(S_t), z) for t ascending, from z = 0
 */
int orc_enum_subset(const double* A, int m, int n, const double* b, const double* c,
                    const int* subset, double* xB_out, double* z_out) {
    (void)n;
    const int ld = m + 1;
    double Wst[17 * 18];
    double xst[64];
    int rowst[64];
    unsigned char usedst[64];
    double* W = (m <= 17) ? Wst : (double*)xmalloc(sizeof(double) * (size_t)m * ld);
    double* x = (m <= 64) ? xst : (double*)xmalloc(sizeof(double) * (size_t)m);
    int* rowpos = (m <= 64) ? rowst : (int*)xmalloc(sizeof(int) * (size_t)m);
    unsigned char* used = (m <= 64) ? usedst : (unsigned char*)xmalloc((size_t)m);
    for (int i = 0; i < m; ++i) {
        for (int t = 0; t < m; ++t) W[i * ld + t] = AT(A, m, i, subset[t]);
        W[i * ld + m] = b[i];
        used[i] = 0;
    }
    int status = ORC_SUBSET_FEASIBLE;
    double minp = INFINITY, maxp = 0.0;
    const int g = m >= 2 ? m - 2 : 0;
    for (int t = 0; t < g; ++t) {
        int p = -1;
        double big = -1.0;
        for (int i = 0; i < m; ++i) {
            if (used[i]) continue;
            double a = fabs(W[i * ld + t]);
            if (a > big) { big = a; p = i; }
        }
        if (!(big > 0.0)) { status = ORC_SUBSET_SINGULAR; break; }
        if (big < minp) minp = big;
        if (big > maxp) maxp = big;
        const double inv = 1.0 / W[p * ld + t];
        for (int i = 0; i < m; ++i) {
            if (i == p) continue;
            const double l = -(W[i * ld + t] * inv);
            for (int cc = t + 1; cc <= m; ++cc)
                W[i * ld + cc] = fma(l, W[p * ld + cc], W[i * ld + cc]);
        }
        for (int cc = t + 1; cc <= m; ++cc) W[p * ld + cc] = W[p * ld + cc] * inv;
        used[p] = 1;
        rowpos[t] = p;
    }
    if (status == ORC_SUBSET_FEASIBLE) {
        if (m == 1) {
            const double piv = W[0];
            const double big = fabs(piv);
            if (!(big > 0.0)) status = ORC_SUBSET_SINGULAR;
            minp = maxp = big;
            x[0] = W[1] * (1.0 / piv);
        } else {
            int r1 = -1, r2 = -1;
            for (int i = 0; i < m; ++i)
                if (!used[i]) { if (r1 < 0) r1 = i; else r2 = i; }
            const int ca = m - 2, cb = m - 1;
            const int p = (fabs(W[r2 * ld + ca]) > fabs(W[r1 * ld + ca])) ? r2 : r1;
            const int q = (p == r1) ? r2 : r1;
            const double piv1 = W[p * ld + ca];
            const double big1 = fabs(piv1);
            if (!(big1 > 0.0)) status = ORC_SUBSET_SINGULAR;
            const double inv1 = 1.0 / piv1;
            const double l = -(W[q * ld + ca] * inv1);
            const double wqb = fma(l, W[p * ld + cb], W[q * ld + cb]);
            const double rq = fma(l, W[p * ld + m], W[q * ld + m]);
            const double big2 = fabs(wqb);
            if (!(big2 > 0.0)) status = ORC_SUBSET_SINGULAR;
            const double inv2 = 1.0 / wqb;
            const double xb = rq * inv2;
            const double xa = fma(-W[p * ld + cb], xb, W[p * ld + m]) * inv1;
            if (big1 < minp) minp = big1;
            if (big1 > maxp) maxp = big1;
            if (big2 < minp) minp = big2;
            if (big2 > maxp) maxp = big2;
            for (int t = 0; t < g; ++t) {
                const int i = rowpos[t];
                x[t] = fma(-W[i * ld + cb], xb, fma(-W[i * ld + ca], xa, W[i * ld + m]));
            }
            x[ca] = xa;
            x[cb] = xb;
        }
    }
    if (status == ORC_SUBSET_FEASIBLE && minp <= DBL_EPSILON * (double)m * maxp)
        status = ORC_SUBSET_SINGULAR;
    if (status == ORC_SUBSET_FEASIBLE) {
        double z = 0.0;
        for (int t = 0; t < m; ++t) {
            const double xv = x[t];
            if (xB_out) xB_out[t] = xv;
            if (!(xv >= -1e-9)) status = ORC_SUBSET_INFEASIBLE; /* Canonical.cpp:171; NaN counts as infeasible */
            z = fma(c[subset[t]], xv, z);
        }
        if (z_out) *z_out = z;
    }
    if (W != Wst) free(W);
    if (x != xst) free(x);
    if (rowpos != rowst) free(rowpos);
    if (used != usedst) free(used);
    return status;
}

int orc_enum_range(const double* A, int m, int n, const double* b, const double* c, int maximize,
                   uint64_t begin, uint64_t end, double* zbest_out, uint64_t counts[3]) {
    int subset[64];
    double xB[64];
    uint64_t cnt[3] = {0, 0, 0};
    double zbest = maximize ? -INFINITY : INFINITY;
    if (begin < end) {
        orc_unrank(n, m, begin, subset);
        for (uint64_t k = begin; k < end; ++k) {
            double z;
            int st = orc_enum_subset(A, m, n, b, c, subset, xB, &z);
            ++cnt[st];
            if (st == ORC_SUBSET_FEASIBLE) {
                if (maximize ? (z > zbest) : (z < zbest)) zbest = z;
            }
            if (!orc_next_subset(n, m, subset)) break;
        }
    }
    if (counts) { counts[0] = cnt[0]; counts[1] = cnt[1]; counts[2] = cnt[2]; }
    if (zbest_out) *zbest_out = zbest;
    return cnt[0] ? ORC_OPTIMAL : ORC_INFEASIBLE;
}

uint64_t orc_enum_first_within(const double* A, int m, int n, const double* b, const double* c,
                               int maximize, uint64_t begin, uint64_t end, double zstar,
                               double tol) {
    int subset[64];
    double xB[64];
    if (begin >= end) return UINT64_MAX;
    orc_unrank(n, m, begin, subset);
    for (uint64_t k = begin; k < end; ++k) {
        double z;
        int st = orc_enum_subset(A, m, n, b, c, subset, xB, &z);
        if (st == ORC_SUBSET_FEASIBLE) {
            if (maximize ? (z >= zstar - tol) : (z <= zstar + tol)) return k;
        }
        if (!orc_next_subset(n, m, subset)) break;
    }
    return UINT64_MAX;
}

int orc_enum_solve(const double* A, int m, int n, const double* b, const double* c, int maximize,
                   int n_orig, double* x_out, int* basis_out, uint64_t* rank_out, double* obj_out,
                   uint64_t counts[3]) {
    if (m <= 0 || n < m || n > 64 || n_orig <= 0 || n_orig > n) return ORC_BAD_ARG;
    const uint64_t total = orc_binom(n, m);
    if (total == 0) return ORC_BAD_ARG;
    double zstar;
    int st = orc_enum_range(A, m, n, b, c, maximize, 0, total, &zstar, counts);
    if (st != ORC_OPTIMAL) return st;
    uint64_t k = orc_enum_first_within(A, m, n, b, c, maximize, 0, total, zstar, 1e-9);
    int subset[64];
    double xB[64], z;
    orc_unrank(n, m, k, subset);
    orc_enum_subset(A, m, n, b, c, subset, xB, &z);
    double x[64];
    for (int j = 0; j < n; ++j) x[j] = 0.0;
    for (int t = 0; t < m; ++t) x[subset[t]] = xB[t];
    for (int j = 0; j < n_orig; ++j) x_out[j] = x[j];
    if (basis_out) memcpy(basis_out, subset, sizeof(int) * (size_t)m);
    if (rank_out) *rank_out = k;
    if (obj_out) *obj_out = z;
    return ORC_OPTIMAL;
}

/* ------------------------------------------------------------------------- */
/* synthetic LPs — SURVEY.md §8(d)                                           */
/* ------------------------------------------------------------------------- */

static uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static double u01(uint64_t key, uint64_t k) { /* counter-based: value k of stream `key` */
    uint64_t z = mix64(key + (k + 1) * 0x9E3779B97F4A7C15ULL);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

void orc_gen_lp(uint64_t seed, int m, int n, double* A, double* b, double* c, int* basis) {
    const int no = n - m;
    const uint64_t key = mix64(seed + 0x5851F42D4C957F2DULL);
    uint64_t k = 0;
    for (int j = 0; j < no; ++j)
        for (int i = 0; i < m; ++i) AT(A, m, i, j) = u01(key, k++);
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < m; ++i) AT(A, m, i, no + j) = (i == j) ? 1.0 : 0.0;
    for (int i = 0; i < m; ++i) b[i] = (1.0 + u01(key, k++)) * ((double)no * 0.5);
    for (int j = 0; j < no; ++j) c[j] = u01(key, k++);
    for (int j = no; j < n; ++j) c[j] = 0.0;
    if (basis)
        for (int i = 0; i < m; ++i) basis[i] = no + i;
}
