/*
 * pm_oracle_mg.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement of the grid solvers on PySCo's per-step path: the 7-point Laplacian multigrid
 * pieces (laplacian.py, mesh.py transfer operators), the f(R) n=1 cubic operator (cubic.py) and the
 * QUMOND source term (mond.py).  Citations are relative to /root/reference/pysco/.
 *
 * Typing follows what Numba infers for the reference (fp32 arrays and np.float32 scalars stay
 * fp32; Python-float / int literals mixed with them promote to fp64; solution_cubic_equation is
 * explicitly fp64).  The linear-Laplacian functions reproduce the reference's per-cell operation
 * order exactly and are bit-identical to the golden vectors; the cubic/MOND ones agree to a few
 * ulp (the interpreter evaluates `x**2` through powf and keeps mixed scalars in fp32).
 */
#define _USE_MATH_DEFINES
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IDX3(i, j, k, n) ((((size_t)(i)) * (n) + (size_t)(j)) * (n) + (size_t)(k))
static inline int wr(int i, int n) { return (i + n) & (n - 1); } /* n is a power of two */

/* ------------------------------------------------------------------ transfer operators ------- */

/* mesh.py:14-60 restriction (scale = 0.125) / :63-110 minus_restriction (scale = -0.125) */
void pmo_restriction(const float *x, int nf, float scale, float *out) {
    const int nc = nf / 2;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nc; ++i)
        for (int j = 0; j < nc; ++j)
            for (int k = 0; k < nc; ++k) {
                int ii = 2 * i, jj = 2 * j, kk = 2 * k;
                out[IDX3(i, j, k, nc)] =
                    scale * (x[IDX3(ii, jj, kk, nf)] + x[IDX3(ii, jj, kk + 1, nf)] + x[IDX3(ii, jj + 1, kk, nf)] +
                             x[IDX3(ii, jj + 1, kk + 1, nf)] + x[IDX3(ii + 1, jj, kk, nf)] +
                             x[IDX3(ii + 1, jj, kk + 1, nf)] + x[IDX3(ii + 1, jj + 1, kk, nf)] +
                             x[IDX3(ii + 1, jj + 1, kk + 1, nf)]);
            }
}

/* mesh.py:206-330 prolongation (add=0) / :334-455 add_prolongation (add=1): tri-linear weights
 * 27/64, 9/64, 3/64, 1/64; xc is [nc]^3, y is [2nc]^3 */
void pmo_prolongation(float *y, const float *xc, int nc, int add) {
    const int nf = 2 * nc;
    const float f0 = (float)(27.0 / 64), f1 = (float)(9.0 / 64), f2 = (float)(3.0 / 64), f3 = (float)(1.0 / 64);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nc; ++i)
        for (int j = 0; j < nc; ++j)
            for (int k = 0; k < nc; ++k) {
                const float c = xc[IDX3(i, j, k, nc)];
                const float tmp0 = f0 * c;
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 2; ++b)
                        for (int d = 0; d < 2; ++d) {
                            const int ia = wr(i + (a ? 1 : -1), nc), jb = wr(j + (b ? 1 : -1), nc),
                                      kd = wr(k + (d ? 1 : -1), nc);
                            const float X = xc[IDX3(ia, j, k, nc)], Y = xc[IDX3(i, jb, k, nc)], Z = xc[IDX3(i, j, kd, nc)];
                            const float XY = xc[IDX3(ia, jb, k, nc)], XZ = xc[IDX3(ia, j, kd, nc)],
                                        YZ = xc[IDX3(i, jb, kd, nc)];
                            const float XYZ = xc[IDX3(ia, jb, kd, nc)];
                            const float v = tmp0 + f1 * (X + Y + Z) + f2 * (XY + XZ + YZ) + f3 * XYZ;
                            float *dst = &y[IDX3(2 * i + a, 2 * j + b, 2 * k + d, nf)];
                            if (add)
                                *dst += v;
                            else
                                *dst = v;
                        }
            }
}

/* ------------------------------------------------------------------ 7-point Laplacian -------- */

/* laplacian.py:12-59 operator (mode 0) and :63-121 residual = b - Ax (mode 1) */
void pmo_lap_operator(const float *x, const float *b, int n, int mode, float *out) {
    const float invh2 = (float)((double)n * n), six = 6.0f;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            for (int k = 0; k < n; ++k) {
                const int im = wr(i - 1, n), ip = wr(i + 1, n), jm = wr(j - 1, n), jp = wr(j + 1, n), km = wr(k - 1, n),
                          kp = wr(k + 1, n);
                const float s = x[IDX3(im, j, k, n)] + x[IDX3(i, jm, k, n)] + x[IDX3(i, j, km, n)] -
                                six * x[IDX3(i, j, k, n)] + x[IDX3(i, j, kp, n)] + x[IDX3(i, jp, k, n)] +
                                x[IDX3(ip, j, k, n)];
                out[IDX3(i, j, k, n)] = mode ? (-(s)*invh2 + b[IDX3(i, j, k, n)]) : s * invh2;
            }
}

/* laplacian.py:125-226 restrict_residual: (1/8) * sum over the 2x2x2 children of (b - Ax), with the
 * interior faces cancelled analytically; summation order as written in the reference */
void pmo_lap_restrict_residual(const float *x, const float *b, int n, float *out) {
    const int nc = n / 2;
    const float invh2 = (float)((double)n * n), inveight = 0.125f, three = 3.0f;
#define X(a, bb, c) x[IDX3(wr(a, n), wr(bb, n), wr(c, n), n)]
#define B(a, bb, c) b[IDX3(a, bb, c, n)]
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nc; ++i)
        for (int j = 0; j < nc; ++j)
            for (int k = 0; k < nc; ++k) {
                const int ii = 2 * i, jj = 2 * j, kk = 2 * k;
                const int m1 = ii - 1, p1 = ii + 1, p2 = ii + 2;
                const int jm1 = jj - 1, jp1 = jj + 1, jp2 = jj + 2;
                const int km1 = kk - 1, kp1 = kk + 1, kp2 = kk + 2;
                float s = +X(m1, jj, kk) + X(m1, jj, kp1) + X(m1, jp1, kk) + X(m1, jp1, kp1) + X(ii, jm1, kk) +
                          X(ii, jm1, kp1) + X(ii, jj, km1) + X(ii, jj, kp2) + X(ii, jp1, km1) + X(ii, jp1, kp2) +
                          X(ii, jp2, kk) + X(ii, jp2, kp1) + X(p1, jm1, kk) + X(p1, jm1, kp1) + X(p1, jj, km1) +
                          X(p1, jj, kp2) + X(p1, jp1, km1) + X(p1, jp1, kp2) + X(p1, jp2, kk) + X(p1, jp2, kp1) +
                          X(p2, jj, kk) + X(p2, jj, kp1) + X(p2, jp1, kk) + X(p2, jp1, kp1) -
                          three * (X(ii, jj, kk) + X(ii, jj, kp1) + X(ii, jp1, kk) + X(ii, jp1, kp1) + X(p1, jj, kk) +
                                   X(p1, jj, kp1) + X(p1, jp1, kk) + X(p1, jp1, kp1));
                out[IDX3(i, j, k, nc)] =
                    inveight * (-(s)*invh2 + B(ii, jj, kk) + B(ii, jj, kp1) + B(ii, jp1, kk) + B(ii, jp1, kp1) +
                                B(p1, jj, kk) + B(p1, jj, kp1) + B(p1, jp1, kk) + B(p1, jp1, kp1));
            }
#undef X
#undef B
}

/* laplacian.py:327-381 residual_error: sqrt(sum (b - Ax)^2), fp32 sequential accumulation */
float pmo_lap_residual_error(const float *x, const float *b, int n) {
    const float invh2 = (float)((double)n * n), six = 6.0f;
    float result = 0.0f;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            for (int k = 0; k < n; ++k) {
                const int im = wr(i - 1, n), ip = wr(i + 1, n), jm = wr(j - 1, n), jp = wr(j + 1, n), km = wr(k - 1, n),
                          kp = wr(k + 1, n);
                const float s = +x[IDX3(im, j, k, n)] + x[IDX3(i, j, km, n)] + x[IDX3(i, j, kp, n)] +
                                x[IDX3(i, jm, k, n)] - six * x[IDX3(i, j, k, n)] + x[IDX3(i, jp, k, n)] +
                                x[IDX3(ip, j, k, n)];
                const float r = -(s)*invh2 + b[IDX3(i, j, k, n)];
                result += r * r;
            }
    return sqrtf(result);
}

/* sqrt(sum (f*a - b)^2) over flat arrays: the tail of laplacian.truncation_error :502-533 (f=1) and
 * cubic.truncation_error :1021-1062 (f=4) */
float pmo_diff_norm(const float *a, const float *b, float f, int64_t count) {
    float result = 0.0f;
    for (int64_t i = 0; i < count; ++i) {
        const float d = f * a[i] - b[i];
        result += d * d;
    }
    return sqrtf(result);
}

/* laplacian.py:765-797 initialise_potential: -h^2/6 * b */
void pmo_lap_initialise_potential(const float *b, int n, float *out) {
    const float h = (float)(1.0 / n);
    const float c = (float)(-(double)(h * h) / 6.0);
    const int64_t count = (int64_t)n * n * n;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < count; ++i) out[i] = c * b[i];
}

/* laplacian.py:844-1022 gauss_seidel: red (i+j+k odd) then black sweep, over-relaxation f_relax.
 * Each of the 8 cell parities sums its 6 neighbours and the -h2*b term in its own order in the
 * reference; the tables below list that order (0..5 = xm,xp,ym,yp,zm,zp; 6 = -h2*b). */
static const int GS_ORDER[8][7] = {
    /* parity (pi,pj,pk) -> index pi*4+pj*2+pk */
    /* 000 black */ {0, 2, 4, 6, 5, 3, 1},
    /* 001 red   */ {0, 2, 5, 4, 6, 3, 1},
    /* 010 red   */ {0, 3, 4, 1, 2, 5, 6},
    /* 011 black */ {0, 5, 3, 2, 4, 6, 1},
    /* 100 red   */ {2, 4, 1, 0, 6, 5, 3},
    /* 101 black */ {2, 5, 1, 0, 4, 6, 3},
    /* 110 black */ {4, 3, 1, 0, 2, 6, 5},
    /* 111 red   */ {5, 3, 1, 0, 2, 4, 6},
};

void pmo_lap_gauss_seidel(float *x, const float *b, int n, float f_relax) {
    const float h2 = (float)(1.0 / ((double)n * n)), invsix = (float)(1.0 / 6);
    for (int colour = 1; colour >= 0; --colour) { /* red = odd parity first */
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                for (int k = 0; k < n; ++k) {
                    if (((i + j + k) & 1) != colour) continue;
                    const size_t c = IDX3(i, j, k, n);
                    float t[7];
                    t[0] = x[IDX3(wr(i - 1, n), j, k, n)];
                    t[1] = x[IDX3(wr(i + 1, n), j, k, n)];
                    t[2] = x[IDX3(i, wr(j - 1, n), k, n)];
                    t[3] = x[IDX3(i, wr(j + 1, n), k, n)];
                    t[4] = x[IDX3(i, j, wr(k - 1, n), n)];
                    t[5] = x[IDX3(i, j, wr(k + 1, n), n)];
                    t[6] = -(h2 * b[c]);
                    const int *o = GS_ORDER[(i & 1) * 4 + (j & 1) * 2 + (k & 1)];
                    float s = t[o[0]];
                    for (int q = 1; q < 7; ++q) s = s + t[o[q]];
                    x[c] += f_relax * (s * invsix - x[c]);
                }
    }
}

/* ------------------------------------------------------------------ f(R) n = 1 (cubic.py) and n = 2 (quartic.py) ---- */

/* quartic.py:157-210 solution_quartic_equation (fp64) */
static inline double quartic_root(float pf, float qf) {
    const double pp = (double)pf, qq = (double)qf;
    if (pp == 0.0) return pow(-qq, 0.25);
    const double d0 = 12.0 * qq, d1 = 27.0 * (pp * pp);
    const double r = d0 / d1;
    const double sqrt_term = 1.0 - 4.0 * d0 * (r * r);
    if (sqrt_term < 0.0) return pow(-qq, 0.25);
    const double Q = pow(0.5 * d1 * (1.0 + sqrt(sqrt_term)), 1.0 / 3.0);
    const double Q_d0oQ = Q + d0 / Q;
    if (Q_d0oQ > 0.0) {
        const double S = 0.5 * sqrt(Q_d0oQ * (1.0 / 3.0));
        if (pp > 0.0) return -S + 0.5 * sqrt(-4.0 * (S * S) + pp / S);
        return S + 0.5 * sqrt(-4.0 * (S * S) - pp / S);
    }
    return pow(-qq, 0.25);
}

float pmo_quartic_solution(float p, float q) { return (float)quartic_root(p, q); }


/* cubic.py:162-207 solution_cubic_equation (fp64) */
static inline double cubic_root(float pf, float d1f) {
    const double inv3 = 1.0 / 3;
    const double d1 = (double)d1f, p = (double)pf;
    double d = d1 * d1 + 108.0 * (p * p * p);
    if (d > 0.0) {
        d = d1 + sqrt(d);
        if (d == 0.0) return -inv3 * pow(d1, inv3);
        const double C = pow(0.5 * d, inv3);
        return -inv3 * (C - 3.0 * p / C);
    } else if (d < 0.0) {
        const double d0 = -3.0 * p;
        d = d1 / (2.0 * pow(d0, 1.5));
        if (fabs(d) < 1.0) {
            const double theta = acos(d);
            return (double)(float)(-2.0 * inv3 * sqrt(d0) * cos(inv3 * (theta + 2.0 * M_PI)));
        }
        return -inv3 * pow(d1, inv3);
    }
    return -inv3 * pow(d1, inv3);
}

float pmo_cubic_solution(float p, float d1) { return (float)cubic_root(p, d1); }

static inline float fr_pw(float a, int fr_n) { return fr_n == 1 ? a * a : a * a * a; } /* u^2 (n=1), u^3 (n=2) */

static inline float cubic_p(const float *x, const float *b, int n, int i, int j, int k, float h2, float invsix,
                            int fr_n) {
    const int im = wr(i - 1, n), ip = wr(i + 1, n), jm = wr(j - 1, n), jp = wr(j + 1, n), km = wr(k - 1, n),
              kp = wr(k + 1, n);
    const float a0 = x[IDX3(im, j, k, n)], a1 = x[IDX3(i, jm, k, n)], a2 = x[IDX3(i, j, km, n)],
                a3 = x[IDX3(i, j, kp, n)], a4 = x[IDX3(i, jp, k, n)], a5 = x[IDX3(ip, j, k, n)];
    return h2 * b[IDX3(i, j, k, n)] - invsix * (fr_pw(a0, fr_n) + fr_pw(a1, fr_n) + fr_pw(a2, fr_n) + fr_pw(a3, fr_n) +
                                                 fr_pw(a4, fr_n) + fr_pw(a5, fr_n));
}

static inline float fr_lead(float xt, int fr_n) { return fr_n == 1 ? xt * xt * xt : (xt * xt) * (xt * xt); }

/* cubic.py:23-86 / quartic.py:20-82 operator (rhs == NULL): u^(n+2) + p u + q h^2 ;
 * cubic.py:90-158 / quartic.py:86-152 residual_with_rhs: -(...) + rhs */
void pmo_cubic_operator(const float *x, const float *b, float q, const float *rhs, int n, int fr_n, float *out) {
    const float h2 = (float)(1.0 / ((double)n * n)), invsix = (float)(1.0 / 6);
    const float qh2 = q * h2;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            for (int k = 0; k < n; ++k) {
                const float p = cubic_p(x, b, n, i, j, k, h2, invsix, fr_n);
                const float xt = x[IDX3(i, j, k, n)];
                if (!rhs)
                    out[IDX3(i, j, k, n)] = fr_lead(xt, fr_n) + p * xt + qh2;
                else
                    out[IDX3(i, j, k, n)] = -(fr_lead(xt, fr_n)) - p * xt - qh2 + rhs[IDX3(i, j, k, n)];
            }
}

/* cubic.py:844-900 residual_error: sqrt(sum (u^3 + p u + q h^2)^2) */
float pmo_cubic_residual_error(const float *x, const float *b, float q, int n, int fr_n) {
    const float h2 = (float)(1.0 / ((double)n * n)), invsix = (float)(1.0 / 6);
    const float qh2 = q * h2;
    float result = 0.0f;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            for (int k = 0; k < n; ++k) {
                const float p = cubic_p(x, b, n, i, j, k, h2, invsix, fr_n);
                const float xt = x[IDX3(i, j, k, n)];
                const float r = fr_lead(xt, fr_n) + p * xt + qh2;
                result += r * r;
            }
    return sqrtf(result);
}

/* cubic.py:217-265 initialise_potential: all fp64 under Numba's typing (3*h2, 27.0*h2*q) */
void pmo_cubic_initialise_potential(const float *b, float q, int n, float *out) {
    const float h2 = (float)(1.0 / ((double)n * n));
    const double threeh2 = 3.0 * (double)h2;
    const double d1 = 27.0 * (double)h2 * (double)q;
    const double inv3 = 1.0 / 3;
    const int64_t count = (int64_t)n * n * n;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < count; ++i) {
        const double d0 = -threeh2 * (double)b[i];
        const double C = pow(0.5 * (d1 + sqrt(d1 * d1 - 4.0 * (d0 * d0 * d0))), inv3);
        out[i] = (float)(-inv3 * (C + d0 / C));
    }
}

/* quartic.py:214-266 initialise_potential (fp64 throughout) */
void pmo_quartic_initialise_potential(const float *b, float q, int n, float *out) {
    const double h2 = 1.0 / ((double)n * n);
    const double d0 = 12.0 * h2 * (double)q;
    const int64_t count = (int64_t)n * n * n;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < count; ++i) {
        const double p = h2 * (double)b[i];
        const double d1 = 27.0 * (p * p);
        const double Q = pow(0.5 * (d1 + sqrt(d1 * d1 - 4.0 * (d0 * d0 * d0))), 1.0 / 3.0);
        const double S = 0.5 * sqrt((Q + d0 / Q) * (1.0 / 3.0));
        out[i] = (float)(-S + 0.5 * sqrt(-4.0 * (S * S) + p / S));
    }
}

/* cubic.py:269-442 / quartic.py:270-444 gauss_seidel (rhs == NULL) and cubic.py:446-668 /
 * quartic.py:448-632 gauss_seidel_with_rhs */
void pmo_cubic_gauss_seidel(float *x, const float *b, float q, const float *rhs, int n, int fr_n, float f_relax) {
    const float h2 = (float)(1.0 / ((double)n * n)), invsix = (float)(1.0 / 6);
    const float qh2 = q * h2; /* quartic.py:306 */
    const float d1_q = rhs ? 27.0f * h2 * q : (float)(27.0 * (double)h2 * (double)q);
    for (int colour = 1; colour >= 0; --colour) {
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                for (int k = 0; k < n; ++k) {
                    if (((i + j + k) & 1) != colour) continue;
                    const size_t c = IDX3(i, j, k, n);
                    const float p = cubic_p(x, b, n, i, j, k, h2, invsix, fr_n);
                    float sol;
                    if (fr_n == 1) {
                        const float d1 = rhs ? d1_q - 27.0f * rhs[c] : d1_q;
                        sol = (float)cubic_root(p, d1);
                    } else {
                        const float qq = rhs ? qh2 - rhs[c] : qh2;
                        sol = (float)quartic_root(p, qq);
                    }
                    x[c] += f_relax * (sol - x[c]);
                }
    }
}

/* ------------------------------------------------------------------ QUMOND (mond.py) --------- */

/* mond.py:16-162 interpolating functions nu(y); kind: 0 simple, 1 n, 2 beta, 3 gamma, 4 delta */
static inline float mond_nu(float y, int kind, float alpha) {
    switch (kind) {
        case 0: return 0.5f + sqrtf(0.25f + 1.0f / y);
        case 1: {
            const int n = (int)alpha;
            const float inv_n = (float)(1.0 / n);
            return powf(0.5f + sqrtf(0.25f + powf(y, (float)(-n))), inv_n);
        }
        case 2: {
            const float e = expf(-y);
            float nu = alpha * e;
            const float om = 1.0f - e;
            if (om > 0) nu += powf(om, -0.5f);
            return nu;
        }
        case 3: {
            const float half_gamma = (float)(0.5 * alpha), inv_gamma = (float)(1.0 / alpha),
                        minus_inv_gamma = (float)(-1.0 / alpha);
            const float e = expf(-powf(y, half_gamma));
            return powf(1.0f - e, minus_inv_gamma) + (1.0f - inv_gamma) * e;
        }
        default: {
            const float half_delta = (float)(0.5 * alpha), minus_inv_delta = (float)(-1.0 / alpha);
            return powf(1.0f - expf(-powf(y, half_delta)), minus_inv_delta);
        }
    }
}

float pmo_mond_nu(float y, int kind, float alpha) { return mond_nu(y, kind, alpha); }

/* mond.py:171-316 rhs_simple (and :322-932 its four siblings, identical but for nu):
 * out = div[ nu(|grad phi|/g0) grad phi ] on the half-cell points A (-h/2) and B (+h/2) */
void pmo_mond_rhs(const float *pot, float *out, int n, float g0, int kind, float alpha) {
    const float inv_g0 = (float)(1.0 / (double)g0);
    const float invh = (float)n, inv4h = (float)(0.25 * n);
#define P(a, b, c) pot[IDX3(wr(a, n), wr(b, n), wr(c, n), n)]
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            for (int k = 0; k < n; ++k) {
                const int im1 = i - 1, ip1 = i + 1, jm1 = j - 1, jp1 = j + 1, km1 = k - 1, kp1 = k + 1;
                const float p000 = P(i, j, k);
                const float Axx = invh * (p000 - P(im1, j, k));
                const float Axy = inv4h * (P(i, jp1, k) - P(i, jm1, k) + P(im1, jp1, k) - P(im1, jm1, k));
                const float Axz = inv4h * (P(i, j, kp1) - P(i, j, km1) + P(im1, j, kp1) - P(im1, j, km1));
                const float fAx = sqrtf(Axx * Axx + Axy * Axy + Axz * Axz);
                const float Bxx = invh * (-p000 + P(ip1, j, k));
                const float Bxy = inv4h * (P(ip1, jp1, k) - P(ip1, jm1, k) + P(i, jp1, k) - P(i, jm1, k));
                const float Bxz = inv4h * (P(ip1, j, kp1) - P(ip1, j, km1) + P(i, j, kp1) - P(i, j, km1));
                const float fBx = sqrtf(Bxx * Bxx + Bxy * Bxy + Bxz * Bxz);
                const float Ayy = invh * (p000 - P(i, jm1, k));
                const float Ayx = inv4h * (P(ip1, j, k) - P(im1, j, k) + P(ip1, jm1, k) - P(im1, jm1, k));
                const float Ayz = inv4h * (P(i, j, kp1) - P(i, j, km1) + P(i, jm1, kp1) - P(i, jm1, km1));
                const float fAy = sqrtf(Ayx * Ayx + Ayy * Ayy + Ayz * Ayz);
                const float Byy = invh * (-p000 + P(i, jp1, k));
                const float Byx = inv4h * (P(ip1, jp1, k) - P(im1, jp1, k) + P(ip1, j, k) - P(im1, j, k));
                const float Byz = inv4h * (P(i, jp1, kp1) - P(i, jp1, km1) + P(i, j, kp1) - P(i, j, km1));
                const float fBy = sqrtf(Byx * Byx + Byy * Byy + Byz * Byz);
                const float Azz = invh * (p000 - P(i, j, km1));
                const float Azx = inv4h * (P(ip1, j, k) - P(im1, j, k) + P(ip1, j, km1) - P(im1, j, km1));
                const float Azy = inv4h * (P(i, jp1, k) - P(i, jm1, k) + P(i, jp1, km1) - P(i, jm1, km1));
                const float fAz = sqrtf(Azx * Azx + Azy * Azy + Azz * Azz);
                const float Bzz = invh * (-p000 + P(i, j, kp1));
                const float Bzx = inv4h * (P(ip1, j, kp1) - P(im1, j, kp1) + P(ip1, j, k) - P(im1, j, k));
                const float Bzy = inv4h * (P(i, jp1, kp1) - P(i, jm1, kp1) + P(i, jp1, k) - P(i, jm1, k));
                const float fBz = sqrtf(Bzx * Bzx + Bzy * Bzy + Bzz * Bzz);
                const float nAx = mond_nu(fAx * inv_g0, kind, alpha), nAy = mond_nu(fAy * inv_g0, kind, alpha),
                            nAz = mond_nu(fAz * inv_g0, kind, alpha), nBx = mond_nu(fBx * inv_g0, kind, alpha),
                            nBy = mond_nu(fBy * inv_g0, kind, alpha), nBz = mond_nu(fBz * inv_g0, kind, alpha);
                out[IDX3(i, j, k, n)] = invh * (nBx * Bxx - nAx * Axx + nBy * Byy - nAy * Ayy + nBz * Bzz - nAz * Azz);
            }
#undef P
}
