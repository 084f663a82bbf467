/*
 * pm_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C, single-threaded (optionally OpenMP for the timed CPU baseline) restatement of the
 * fp32 kernels on PySCo's per-step particle-mesh path.  Each function cites the reference
 * file:line it follows (paths relative to /root/reference/pysco/).  Arithmetic is fp32 with the
 * reference's operation order; build with -ffp-contract=off so no FMA is formed (the reference
 * executed as interpreted Python forms none either), which makes the deposit / gather / stencil
 * functions bit-identical to the golden vectors in tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* The arithmetic type: float as the reference (mesh.py:2467, fourier.py:21); -DPMO_DOUBLE builds the SAME formulas in
 * double -- the "fp64 CPU path" BASELINE.md section 6 holds the fp64 device mode against (libpm_oracle_f64.so,
 * oracle/pysco_oracle_f64.py).  Literals such as 0.5f, 0.75f, 8.0f are exact in both. */
#ifdef PMO_DOUBLE
typedef double REAL;
#define R_FABS fabs
#define R_POW pow
#define R_FLOOR floor
#else
typedef float REAL;
#define R_FABS fabsf
#define R_POW powf
#define R_FLOOR floorf
#endif

#define IDX3(i, j, k, n) ((((size_t)(i)) * (n) + (size_t)(j)) * (n) + (size_t)(k))

/* Python negative-index wrap used by the reference: a[i-1] with i==0 -> a[n-1];
 * a[i-(n-1)] is a[i+1] wrapped (mesh.py:2533-2538). */
static inline int wrap_m1(int i, int n) { return i == 0 ? n - 1 : i - 1; }
static inline int wrap_p1(int i, int n) { return i == n - 1 ? 0 : i + 1; }
static inline int wrapn(int i, int n) {
    i %= n;
    return i < 0 ? i + n : i;
}

void pmo_set_num_threads(int t) {
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#else
    (void)t;
#endif
}

int pmo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---------------------------------------------------------------- deposit -------------- */

/* mesh.py:2240-2278 NGP */
void pmo_ngp(const REAL *pos, int64_t np_, int n, REAL *rho) {
    const REAL nf = (REAL)n;
    memset(rho, 0, sizeof(REAL) * (size_t)n * n * n);
    for (int64_t p = 0; p < np_; ++p) {
        int i = (int)(int16_t)(pos[3 * p + 0] * nf);
        int j = (int)(int16_t)(pos[3 * p + 1] * nf);
        int k = (int)(int16_t)(pos[3 * p + 2] * nf);
        rho[IDX3(i, j, k, n)] += 1.0f;
    }
}

/* mesh.py:2284-2358 CIC */
void pmo_cic(const REAL *pos, int64_t np_, int n, REAL *rho) {
    const REAL nf = (REAL)n, one = 1.0f, half = 0.5f;
    memset(rho, 0, sizeof(REAL) * (size_t)n * n * n);
    for (int64_t p = 0; p < np_; ++p) {
        REAL xp = pos[3 * p + 0] * nf, yp = pos[3 * p + 1] * nf, zp = pos[3 * p + 2] * nf;
        int i = (int)(int16_t)xp, j = (int)(int16_t)yp, k = (int)(int16_t)zp;
        REAL dx = xp - half - (REAL)i, dy = yp - half - (REAL)j, dz = zp - half - (REAL)k;
        int sx = (dx > 0) - (dx < 0), sy = (dy > 0) - (dy < 0), sz = (dz > 0) - (dz < 0);
        dx = R_FABS(dx);
        dy = R_FABS(dy);
        dz = R_FABS(dz);
        REAL wx = one - dx, wy = one - dy, wz = one - dz;
        int i2 = wrapn(i + sx, n), j2 = wrapn(j + sy, n), k2 = wrapn(k + sz, n);
        rho[IDX3(i, j, k, n)] += wx * wy * wz;
        rho[IDX3(i, j, k2, n)] += wx * wy * dz;
        rho[IDX3(i, j2, k, n)] += wx * dy * wz;
        rho[IDX3(i, j2, k2, n)] += wx * dy * dz;
        rho[IDX3(i2, j, k, n)] += dx * wy * wz;
        rho[IDX3(i2, j, k2, n)] += dx * wy * dz;
        rho[IDX3(i2, j2, k, n)] += dx * dy * wz;
        rho[IDX3(i2, j2, k2, n)] += dx * dy * dz;
    }
}

/* TSC 1-D weights, mesh.py:2506-2520: w[0]=m1, w[1]=centre, w[2]=p1 */
static inline void tsc_w(REAL xp, int *i, REAL w[3]) {
    const REAL half = 0.5f, tq = 0.75f;
    *i = (int)(int16_t)xp;
    REAL d = xp - half - (REAL)(*i);
    w[1] = tq - d * d;
    REAL a = half - d, b = half + d;
    w[0] = half * (a * a);
    w[2] = half * (b * b);
}

/* mesh.py:2363-2462 TSC_seq (== TSC mesh.py:2468-2595 with a fixed summation order).  The OpenMP build
 * (cpu_baseline timing only) is the reference's parallel TSC: prange over particles + atomic adds
 * (mesh.py:2468-2595, numpy_atomic.atomic_add), whose summation order is not fixed. */
void pmo_tsc(const REAL *pos, int64_t np_, int n, REAL *rho) {
    const REAL nf = (REAL)n;
    memset(rho, 0, sizeof(REAL) * (size_t)n * n * n);
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < np_; ++p) {
        int i, j, k;
        REAL wx[3], wy[3], wz[3];
        tsc_w(pos[3 * p + 0] * nf, &i, wx);
        tsc_w(pos[3 * p + 1] * nf, &j, wy);
        tsc_w(pos[3 * p + 2] * nf, &k, wz);
        int ii[3] = {wrap_m1(i, n), i, wrap_p1(i, n)};
        int jj[3] = {wrap_m1(j, n), j, wrap_p1(j, n)};
        int kk[3] = {wrap_m1(k, n), k, wrap_p1(k, n)};
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                REAL wxy = wx[a] * wy[b];
                for (int c = 0; c < 3; ++c) {
                    const REAL w = wxy * wz[c];
#pragma omp atomic
                    rho[IDX3(ii[a], jj[b], kk[c], n)] += w;
                }
            }
    }
    return;
#endif
    for (int64_t p = 0; p < np_; ++p) {
        int i, j, k;
        REAL wx[3], wy[3], wz[3];
        tsc_w(pos[3 * p + 0] * nf, &i, wx);
        tsc_w(pos[3 * p + 1] * nf, &j, wy);
        tsc_w(pos[3 * p + 2] * nf, &k, wz);
        int ii[3] = {wrap_m1(i, n), i, wrap_p1(i, n)};
        int jj[3] = {wrap_m1(j, n), j, wrap_p1(j, n)};
        int kk[3] = {wrap_m1(k, n), k, wrap_p1(k, n)};
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                REAL wxy = wx[a] * wy[b];
                for (int c = 0; c < 3; ++c) rho[IDX3(ii[a], jj[b], kk[c], n)] += wxy * wz[c];
            }
    }
}

/* ---------------------------------------------------------------- gather --------------- */

/* mesh.py:2639-2675 invNGP_vec */
void pmo_inv_ngp_vec(const REAL *grid, int n, const REAL *pos, int64_t np_, REAL *out) {
    const REAL nf = (REAL)n;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < np_; ++p) {
        int i = (int)(int16_t)(pos[3 * p + 0] * nf);
        int j = (int)(int16_t)(pos[3 * p + 1] * nf);
        int k = (int)(int16_t)(pos[3 * p + 2] * nf);
        const REAL *g = grid + 3 * IDX3(i, j, k, n);
        out[3 * p + 0] = g[0];
        out[3 * p + 1] = g[1];
        out[3 * p + 2] = g[2];
    }
}

/* mesh.py:2752-2822 invCIC_vec */
void pmo_inv_cic_vec(const REAL *grid, int n, const REAL *pos, int64_t np_, REAL *out) {
    const REAL nf = (REAL)n, one = 1.0f, half = 0.5f;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < np_; ++p) {
        REAL xp = pos[3 * p + 0] * nf, yp = pos[3 * p + 1] * nf, zp = pos[3 * p + 2] * nf;
        int i = (int)(int16_t)xp, j = (int)(int16_t)yp, k = (int)(int16_t)zp;
        REAL dx = xp - half - (REAL)i, dy = yp - half - (REAL)j, dz = zp - half - (REAL)k;
        int sx = (dx > 0) - (dx < 0), sy = (dy > 0) - (dy < 0), sz = (dz > 0) - (dz < 0);
        dx = R_FABS(dx);
        dy = R_FABS(dy);
        dz = R_FABS(dz);
        REAL wx = one - dx, wy = one - dy, wz = one - dz;
        int i2 = wrapn(i + sx, n), j2 = wrapn(j + sy, n), k2 = wrapn(k + sz, n);
        REAL w[8] = {wx * wy * wz, wx * wy * dz, wx * dy * wz, wx * dy * dz,
                      dx * wy * wz, dx * wy * dz, dx * dy * wz, dx * dy * dz};
        size_t c[8] = {IDX3(i, j, k, n),  IDX3(i, j, k2, n),  IDX3(i, j2, k, n),  IDX3(i, j2, k2, n),
                       IDX3(i2, j, k, n), IDX3(i2, j, k2, n), IDX3(i2, j2, k, n), IDX3(i2, j2, k2, n)};
        for (int m = 0; m < 3; ++m) {
            REAL s = w[0] * grid[3 * c[0] + m];
            for (int q = 1; q < 8; ++q) s = s + w[q] * grid[3 * c[q] + m];
            out[3 * p + m] = s;
        }
    }
}

/* mesh.py:2960-3088 invTSC_vec; ncomp=1 gives the scalar invTSC (mesh.py:2827-2955) */
void pmo_inv_tsc(const REAL *grid, int n, int ncomp, const REAL *pos, int64_t np_, REAL *out) {
    const REAL nf = (REAL)n;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < np_; ++p) {
        int i, j, k;
        REAL wx[3], wy[3], wz[3];
        tsc_w(pos[3 * p + 0] * nf, &i, wx);
        tsc_w(pos[3 * p + 1] * nf, &j, wy);
        tsc_w(pos[3 * p + 2] * nf, &k, wz);
        int ii[3] = {wrap_m1(i, n), i, wrap_p1(i, n)};
        int jj[3] = {wrap_m1(j, n), j, wrap_p1(j, n)};
        int kk[3] = {wrap_m1(k, n), k, wrap_p1(k, n)};
        REAL w[27];
        size_t c[27];
        int q = 0;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                REAL wxy = wx[a] * wy[b];
                for (int cc = 0; cc < 3; ++cc, ++q) {
                    w[q] = wxy * wz[cc];
                    c[q] = IDX3(ii[a], jj[b], kk[cc], n);
                }
            }
        for (int m = 0; m < ncomp; ++m) {
            REAL s = w[0] * grid[ncomp * c[0] + m];
            for (q = 1; q < 27; ++q) s = s + w[q] * grid[ncomp * c[q] + m];
            out[ncomp * p + m] = s;
        }
    }
}

/* ---------------------------------------------------------------- gradient ------------- */

/* mesh.py:639-850 derivative2/3/5/7 -> +grad(a), AoS [n,n,n,3].
 * mesh.py:860-2069 derivativeK_fR_n1/n2: + f * grad(b^2) (n=1) or f * grad(b^3) (n=2);
 * pass b=NULL for the plain gradient. */
static inline REAL fr_pow(REAL b, int fr_n) { return fr_n == 1 ? b * b : b * b * b; }

void pmo_derivative(const REAL *a, const REAL *b, REAL f, int fr_n, int n, int order, REAL *out) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            for (int k = 0; k < n; ++k) {
                int ax[3] = {i, j, k};
                REAL r[3];
                for (int m = 0; m < 3; ++m) {
                    size_t id[7]; /* offsets -3..+3 along axis m */
                    for (int o = -3; o <= 3; ++o) {
                        int t[3] = {ax[0], ax[1], ax[2]};
                        t[m] = wrapn(ax[m] + o, n);
                        id[o + 3] = IDX3(t[0], t[1], t[2], n);
                    }
#define A(o) a[id[(o) + 3]]
#define B(o) fr_pow(b[id[(o) + 3]], fr_n)
                    REAL v;
                    if (order == 2) { /* mesh.py:639, :860 */
                        REAL invh = (REAL)n;
                        if (!b)
                            v = invh * (-A(0) + A(1));
                        else
                            v = invh * (-A(0) + A(1) + f * (-B(0) + B(1)));
                    } else if (order == 3) { /* mesh.py:681, :924 */
                        REAL inv2h = (REAL)(0.5 * n);
                        if (!b)
                            v = inv2h * (-A(-1) + A(1));
                        else
                            v = inv2h * (-A(-1) + A(1) + f * (-B(-1) + B(1)));
                    } else if (order == 5) { /* mesh.py:725, :1001 */
                        REAL inv12h = (REAL)(n / 12.0);
                        if (!b)
                            v = inv12h * (8.0f * (-A(-1) + A(1)) + A(-2) - A(2));
                        else
                            v = inv12h * (8.0f * (-A(-1) + A(1) + f * (-B(-1) + B(1))) + A(-2) - A(2) +
                                          f * (B(-2) - B(2)));
                    } else { /* mesh.py:782, :1097 */
                        REAL inv60h = (REAL)(n / 60.0);
                        if (!b)
                            v = inv60h * (45.0f * (-A(-1) + A(1)) + 9.0f * (A(-2) - A(2)) - A(-3) + A(3));
                        else
                            v = inv60h * (45.0f * (-A(-1) + A(1) + f * (-B(-1) + B(1))) +
                                          9.0f * (+A(-2) - A(2) + f * (B(-2) - B(2))) - A(-3) + A(3) +
                                          f * (-B(-3) + B(3)));
                    }
#undef A
#undef B
                    r[m] = v;
                }
                REAL *o3 = out + 3 * IDX3(i, j, k, n);
                o3[0] = r[0];
                o3[1] = r[1];
                o3[2] = r[2];
            }
}

/* ---------------------------------------------------------------- element-wise --------- */

/* utils.py:264-298 add_vector_scalar_inplace: y += a*x with the a==+-1 special cases */
void pmo_axpy(REAL *y, const REAL *x, REAL a, int64_t n) {
    if (a == 1.0f) {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) y[i] += x[i];
    } else if (a == -1.0f) {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) y[i] -= x[i];
    } else {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) y[i] += a * x[i];
    }
}

/* utils.py:410-430 prod_vector_scalar_inplace */
void pmo_scale(REAL *y, REAL a, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) y[i] *= a;
}

/* utils.py:644-718 linear_operator(_inplace): out = f1*x + f2 */
void pmo_linear_operator(const REAL *x, REAL f1, REAL f2, REAL *out, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) out[i] = f1 * x[i] + f2;
}

/* utils.py:721-755 linear_operator_vectors_inplace: x = f1*x + f2*y */
void pmo_linear_operator_vectors(REAL *x, REAL f1, const REAL *y, REAL f2, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) x[i] = f1 * x[i] + f2 * y[i];
}

/* utils.py:220-240 max_abs */
REAL pmo_max_abs(const REAL *x, int64_t n) {
    REAL m = 0.0f;
#pragma omp parallel for schedule(static) reduction(max : m)
    for (int64_t i = 0; i < n; ++i) {
        REAL v = R_FABS(x[i]);
        if (v > m) m = v;
    }
    return m;
}

/* utils.py:1120-1149 periodic_wrap; eps is a Python double: -(0.5**25)*(1+1e-6) */
void pmo_periodic_wrap(REAL *x, int64_t n) {
    double eps = -(1.0 / 33554432.0);
    eps += 1e-6 * eps;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        REAL t = x[i];
        if (t < 0.0f) {
            if ((double)t > eps)
                x[i] = 0.0f;
            else
                x[i] = t + 1.0f;
        } else if (t >= 1.0f) {
            x[i] = t - 1.0f;
        }
    }
}

/* ---------------------------------------------------------------- morton ---------------- */

/* morton.py:42-79 */
static inline int64_t interleave21(int64_t x) {
    x &= 0x1FFFFF;
    x = (x | x << 32) & 0x1F00000000FFFFLL;
    x = (x | x << 16) & 0x1F0000FF0000FFLL;
    x = (x | x << 8) & 0x100F00F00F00F00FLL;
    x = (x | x << 4) & 0x10C30C30C30C30C3LL;
    x = (x | x << 2) & 0x1249249249249249LL;
    return x;
}

/* morton.py:82-137 key / positions_to_keys: floor(x * 2**21) evaluated as fp32 * int -> fp32 */
void pmo_morton_keys(const REAL *pos, int64_t np_, int64_t *keys) {
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < np_; ++p) {
        int64_t xx = interleave21((int64_t)R_FLOOR(pos[3 * p + 0] * 2097152.0f));
        int64_t yy = interleave21((int64_t)R_FLOOR(pos[3 * p + 1] * 2097152.0f));
        int64_t zz = interleave21((int64_t)R_FLOOR(pos[3 * p + 2] * 2097152.0f));
        keys[p] = xx << 2 | yy << 1 | zz;
    }
}

/* ---------------------------------------------------------------- k-space --------------- */

/* fourier.py:460-595: x[i,j,k] *= G.  The caller passes per-axis tables built in numpy with the
 * reference's fp32 expressions (k2[i] = kx**2, s[i] = sinc(kx*h) or sin(pi_h*kx)**2):
 *   kind 0 (inverse_laplacian, :460-491):        G = c0 / (k2x + k2y + k2z)
 *   kind 1 (_compensated, :502-544):             G = (sx*sy*sz)**(-2p) * (c0 / (k2x+k2y+k2z))
 *   kind 2 (_7pt, :555-595):                     G = c0 / (fx + fy + fz)
 * Element [0,0,0] is set to 0 afterwards in all three. */
void pmo_greens(REAL *xc, int n, int kind, int p, const REAL *k2, const REAL *s, double c0) {
    const int nh = n / 2 + 1;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            REAL *row = xc + 2 * (((size_t)i * n + j) * nh);
            if (kind == 2) {
                REAL fxy = s[i] + s[j];
                for (int k = 0; k < nh; ++k) {
                    REAL g = (REAL)c0 / (fxy + s[k]);
                    row[2 * k] *= g;
                    row[2 * k + 1] *= g;
                }
            } else {
                REAL k2xy = k2[i] + k2[j];
                REAL wxy = (kind == 1) ? s[i] * s[j] : 1.0f;
                for (int k = 0; k < nh; ++k) {
                    REAL invk2 = (REAL)c0 / (k2xy + k2[k]);
                    REAL g = invk2;
                    if (kind == 1) {
                        REAL w = wxy * s[k];
                        /* w_xyz**minus_twop: fp32 ** python int -> fp32 pow */
                        g = R_POW(w, (REAL)(-2 * p)) * invk2;
                    }
                    row[2 * k] *= g;
                    row[2 * k + 1] *= g;
                }
            }
        }
    xc[0] = 0.0f;
    xc[1] = 0.0f;
}

/* fourier.py:22-100 fourier_grid_to_Pk: per-i private histograms, summed over i afterwards.
 * out arrays have length n (bins 0..n-1); the Python wrapper slices [1:kmax_orszag] and divides.
 * sincs[i] = sinc(k_i/n) in fp32 from numpy.  Side effect: density_k[0,0,0] = 0. */
void pmo_pk(REAL *dk, int n, int p, const REAL *sincs, REAL *ksum, REAL *pksum, REAL *nmodes) {
    const int nh = n / 2 + 1, middle = n / 2;
    REAL *karr = (REAL *)calloc((size_t)n * n, sizeof(REAL));
    REAL *parr = (REAL *)calloc((size_t)n * n, sizeof(REAL));
    REAL *narr = (REAL *)calloc((size_t)n * n, sizeof(REAL));
    for (int i = 0; i < n; ++i) {
        REAL kx = (i >= middle) ? (REAL)(i - n) : (REAL)i;
        REAL kx2 = kx * kx;
        REAL wx = sincs[i];
        for (int j = 0; j < n; ++j) {
            REAL ky = (j >= middle) ? (REAL)(j - n) : (REAL)j;
            REAL wxy = wx * sincs[j];
            REAL kx2ky2 = kx2 + ky * ky;
            for (int k = 0; k < nh; ++k) {
                REAL *c = dk + 2 * (((size_t)i * n + j) * nh + k);
                if (i == 0 && j == 0 && k == 0) {
                    c[0] = 0.0f;
                    c[1] = 0.0f;
                    continue;
                }
                REAL kz = (REAL)k;
                REAL wxyz = wxy * sincs[k];
                REAL comp = R_POW(wxyz, (REAL)(-p));
                REAL re = c[0] * comp, im = c[1] * comp;
                REAL delta2 = re * re + im * im;
                /* math.sqrt on an fp32 -> double */
                double knorm = sqrt((double)(kx2ky2 + kz * kz));
                int kidx = (int)(knorm + 0.5);
                narr[(size_t)i * n + kidx] += 1.0f;
                karr[(size_t)i * n + kidx] += (REAL)knorm;
                parr[(size_t)i * n + kidx] += delta2;
            }
        }
    }
    /* np.sum(axis=0) on fp32: pairwise only along contiguous axes; axis-0 reduction is a plain
     * row-by-row accumulation */
    for (int b = 0; b < n; ++b) {
        REAL sk = 0, sp = 0, sn = 0;
        for (int i = 0; i < n; ++i) {
            sk += karr[(size_t)i * n + b];
            sp += parr[(size_t)i * n + b];
            sn += narr[(size_t)i * n + b];
        }
        ksum[b] = sk;
        pksum[b] = sp;
        nmodes[b] = sn;
    }
    free(karr);
    free(parr);
    free(narr);
}
