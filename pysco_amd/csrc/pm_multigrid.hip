// Grid solvers: 7-point Laplacian multigrid pieces (laplacian.py, mesh.py transfer operators), the
// f(R) n=1 cubic operator with its per-cell fp64 root (cubic.py) and the QUMOND source stencil
// (mond.py).  Every function takes the level's grid size explicitly: the host walks the V-cycle.
//
// Arithmetic: fp32 with the reference's operation order (built with -ffp-contract=off), so the
// linear-Laplacian kernels are bit-identical to the CPU oracle; the norms accumulate in fp64.
#include "pm_common.h"

#define BLK 256

__device__ __forceinline__ int64_t cidx(int i, int j, int k, int log2n) {
    return ((((int64_t)i << log2n) + j) << log2n) + k;
}

static inline int ilog2(int n) { return __builtin_ctz((unsigned)n); }

// Which planes of a level array a kernel works on.  Single GPU: the periodic cube {0, n, n - 1}.  x-slab
// (multi-GPU): the array holds one ghost plane on each side of the ni owned ones and x never wraps
// {1, ni, INT_MAX}; the host fills the ghosts (ring exchange) before every kernel that reads neighbours.
struct XRange {
    int i0, ni, xmask;
};
static inline XRange xr_cube(int n) { return XRange{0, n, n - 1}; }
static inline XRange xr_slab(int nx) { return XRange{1, nx, 0x7fffffff}; }

#define PM_CHECK_GRID(n) PM_REQUIRE((n) >= 2 && (n) <= 16384 && (((n) & ((n)-1)) == 0), "grid size must be a power of two")

// ------------------------------------------------------------------------------------------------
// transfer operators

// mesh.py:14-60 restriction (scale 1/8), :63-110 minus_restriction (scale -1/8)
__global__ __launch_bounds__(BLK) void restriction_kernel(const float *__restrict__ x, int log2nf, float scale,
                                                          float *__restrict__ out, XRange xf, XRange xc) {
    const int log2nc = log2nf - 1, nc = 1 << log2nc;
    const int64_t total = (int64_t)xc.ni << (2 * log2nc);
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLK) {
        const int k = (int)(t & (nc - 1)), j = (int)((t >> log2nc) & (nc - 1)), ic = (int)(t >> (2 * log2nc));
        const int ii = xf.i0 + 2 * ic, jj = 2 * j, kk = 2 * k;
        const float2 a = *reinterpret_cast<const float2 *>(x + cidx(ii, jj, kk, log2nf));
        const float2 b = *reinterpret_cast<const float2 *>(x + cidx(ii, jj + 1, kk, log2nf));
        const float2 c = *reinterpret_cast<const float2 *>(x + cidx(ii + 1, jj, kk, log2nf));
        const float2 d = *reinterpret_cast<const float2 *>(x + cidx(ii + 1, jj + 1, kk, log2nf));
        out[cidx(xc.i0 + ic, j, k, log2nc)] = scale * (a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y);
    }
}

static int restriction_launch(pm_ctx *c, const float *d_x, int nf, float scale, float *d_out, XRange xf, XRange xc) {
    const int64_t total = (int64_t)xc.ni * (nf / 2) * (nf / 2);
    hipLaunchKernelGGL(restriction_kernel, dim3(pm_stream_blocks(total, BLK)), dim3(BLK), 0, c->stream, d_x, ilog2(nf),
                       scale, d_out, xf, xc);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

extern "C" int pm_restriction(pm_ctx *c, const float *d_x, int nf, float scale, float *d_out) {
    PM_REQUIRE(c && d_x && d_out, "NULL argument");
    PM_CHECK_GRID(nf);
    return restriction_launch(c, d_x, nf, scale, d_out, xr_cube(nf), xr_cube(nf / 2));
}

extern "C" int pm_restriction_slab(pm_ctx *c, const float *d_x, int nf, int nxf, float scale, float *d_out) {
    PM_REQUIRE(c && d_x && d_out && nxf >= 2 && (nxf % 2) == 0, "bad argument (even number of owned fine planes)");
    PM_CHECK_GRID(nf);
    return restriction_launch(c, d_x, nf, scale, d_out, xr_slab(nxf), xr_slab(nxf / 2));
}

// mesh.py:206-330 prolongation / :334-455 add_prolongation: one thread per FINE cell
// (the loop bodies of the scalar level kernels are functions of (first item, stride): the grid-stride kernels run
// them over the whole grid, lap_vcycle_coarse_kernel runs them on one workgroup's LDS arrays -- same expressions)
__device__ __forceinline__ void prolongation_body(float *__restrict__ y, const float *__restrict__ xc, int log2nc, int add,
                                                  XRange xf, XRange xcr, int64_t first, int64_t stride) {
    const int log2nf = log2nc + 1, nf = 1 << log2nf, nc = 1 << log2nc, mc = nc - 1;
    const float f0 = (float)(27.0 / 64), f1 = (float)(9.0 / 64), f2 = (float)(3.0 / 64), f3 = (float)(1.0 / 64);
    const int64_t total = (int64_t)xf.ni << (2 * log2nf);
    for (int64_t t = first; t < total; t += stride) {
        const int kf = (int)(t & (nf - 1)), jf = (int)((t >> log2nf) & (nf - 1)), fi = (int)(t >> (2 * log2nf));
        const int i = xcr.i0 + (fi >> 1), j = jf >> 1, k = kf >> 1;
        const int ia = (i + ((fi & 1) ? 1 : -1)) & xcr.xmask, jb = (j + ((jf & 1) ? 1 : -1)) & mc,
                  kd = (k + ((kf & 1) ? 1 : -1)) & mc;
        const float tmp0 = f0 * xc[cidx(i, j, k, log2nc)];
        const float X = xc[cidx(ia, j, k, log2nc)], Y = xc[cidx(i, jb, k, log2nc)], Z = xc[cidx(i, j, kd, log2nc)];
        const float XY = xc[cidx(ia, jb, k, log2nc)], XZ = xc[cidx(ia, j, kd, log2nc)], YZ = xc[cidx(i, jb, kd, log2nc)];
        const float XYZ = xc[cidx(ia, jb, kd, log2nc)];
        const float v = tmp0 + f1 * (X + Y + Z) + f2 * (XY + XZ + YZ) + f3 * XYZ;
        float *dst = y + cidx(xf.i0 + fi, jf, kf, log2nf);
        *dst = add ? *dst + v : v;
    }
}

__global__ __launch_bounds__(BLK) void prolongation_kernel(float *__restrict__ y, const float *__restrict__ xc,
                                                           int log2nc, int add, XRange xf, XRange xcr) {
    prolongation_body(y, xc, log2nc, add, xf, xcr, (int64_t)blockIdx.x * BLK + threadIdx.x, (int64_t)gridDim.x * BLK);
}

// four fine k-cells per thread (nc >= 8): the four coarse rows are read as one aligned float2 + two scalars each,
// the fine row is one float4 read-modify-write; same expression per fine cell
__global__ __launch_bounds__(BLK) void prolongation_rows_kernel(float *y, const float *ysrc, const float *__restrict__ xc,
                                                                int log2nc, int add, XRange xf, XRange xcr) {
    const int log2nf = log2nc + 1, nf = 1 << log2nf, nc = 1 << log2nc, mc = nc - 1, lq = log2nf - 2;
    const float f0 = (float)(27.0 / 64), f1 = (float)(9.0 / 64), f2 = (float)(3.0 / 64), f3 = (float)(1.0 / 64);
    const int64_t total = ((int64_t)xf.ni << log2nf) << lq;
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLK) {
        const int q = (int)(t & ((1 << lq) - 1)), jf = (int)((t >> lq) & (nf - 1)), fi = (int)(t >> (lq + log2nf));
        const int i = xcr.i0 + (fi >> 1), j = jf >> 1;
        const int ia = (i + ((fi & 1) ? 1 : -1)) & xcr.xmask, jb = (j + ((jf & 1) ? 1 : -1)) & mc;
        auto ld = [&](int a, int bb, float *d) {  // d[0..3] = xc[a][bb][2q-1 .. 2q+2]
            const float *r = xc + ((((int64_t)a << log2nc) + bb) << log2nc);
            const float2 v = *reinterpret_cast<const float2 *>(r + 2 * q);
            d[0] = r[(2 * q - 1) & mc], d[1] = v.x, d[2] = v.y, d[3] = r[(2 * q + 2) & mc];
        };
        float c00[4], cX[4], cY[4], cXY[4];
        ld(i, j, c00), ld(ia, j, cX), ld(i, jb, cY), ld(ia, jb, cXY);
        const int64_t at = cidx(xf.i0 + fi, jf, 4 * q, log2nf);
        float *dst = y + at;  // (add: y = ysrc + prolongation; ysrc == y for the in-place form)
        float4 o = add ? *reinterpret_cast<const float4 *>(ysrc + at) : make_float4(0.f, 0.f, 0.f, 0.f);
        float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) {
            const int k = 1 + (c2 >> 1), kd = k + ((c2 & 1) ? 1 : -1);  // indices into the d[] arrays
            const float tmp0 = f0 * c00[k];
            const float v = tmp0 + f1 * (cX[k] + cY[k] + c00[kd]) + f2 * (cXY[k] + cX[kd] + cY[kd]) + f3 * cXY[kd];
            ov[c2] = add ? ov[c2] + v : v;
        }
        *reinterpret_cast<float4 *>(dst) = make_float4(ov[0], ov[1], ov[2], ov[3]);
    }
}

// eight fine k-cells per thread (nc >= 16): a coarse row is one aligned float4 + two scalars for twice the cells, the
// fine row two float4 read-modify-writes; same expression per fine cell (0.39 -> 0.3 ms at 512^3)
__global__ __launch_bounds__(BLK) void prolongation_rows8_kernel(float *y, const float *ysrc, const float *__restrict__ xc,
                                                                 int log2nc, int add, XRange xf, XRange xcr) {
    const int log2nf = log2nc + 1, nf = 1 << log2nf, nc = 1 << log2nc, mc = nc - 1, lq = log2nf - 3;
    const float f0 = (float)(27.0 / 64), f1 = (float)(9.0 / 64), f2 = (float)(3.0 / 64), f3 = (float)(1.0 / 64);
    const int64_t total = ((int64_t)xf.ni << log2nf) << lq;
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLK) {
        const int q = (int)(t & ((1 << lq) - 1)), jf = (int)((t >> lq) & (nf - 1)), fi = (int)(t >> (lq + log2nf));
        const int i = xcr.i0 + (fi >> 1), j = jf >> 1;
        const int ia = (i + ((fi & 1) ? 1 : -1)) & xcr.xmask, jb = (j + ((jf & 1) ? 1 : -1)) & mc;
        auto ld = [&](int a, int bb, float *d) {  // d[0..5] = xc[a][bb][4q-1 .. 4q+4]
            const float *r = xc + ((((int64_t)a << log2nc) + bb) << log2nc);
            const float4 v = *reinterpret_cast<const float4 *>(r + 4 * q);
            d[0] = r[(4 * q - 1) & mc], d[1] = v.x, d[2] = v.y, d[3] = v.z, d[4] = v.w, d[5] = r[(4 * q + 4) & mc];
        };
        float c00[6], cX[6], cY[6], cXY[6];
        ld(i, j, c00), ld(ia, j, cX), ld(i, jb, cY), ld(ia, jb, cXY);
        const int64_t at = cidx(xf.i0 + fi, jf, 8 * q, log2nf);
        float ov[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (add) {
            const float4 o0 = *reinterpret_cast<const float4 *>(ysrc + at), o1 = *reinterpret_cast<const float4 *>(ysrc + at + 4);
            ov[0] = o0.x, ov[1] = o0.y, ov[2] = o0.z, ov[3] = o0.w, ov[4] = o1.x, ov[5] = o1.y, ov[6] = o1.z, ov[7] = o1.w;
        }
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) {
            const int k = 1 + (c2 >> 1), kd = k + ((c2 & 1) ? 1 : -1);  // indices into the d[] arrays
            const float tmp0 = f0 * c00[k];
            const float v = tmp0 + f1 * (cX[k] + cY[k] + c00[kd]) + f2 * (cXY[k] + cX[kd] + cY[kd]) + f3 * cXY[kd];
            ov[c2] = add ? ov[c2] + v : v;
        }
        *reinterpret_cast<float4 *>(y + at) = make_float4(ov[0], ov[1], ov[2], ov[3]);
        *reinterpret_cast<float4 *>(y + at + 4) = make_float4(ov[4], ov[5], ov[6], ov[7]);
    }
}

static int prolongation_launch(pm_ctx *c, float *d_y, const float *d_xc, int nc, int add, XRange xf, XRange xcr,
                               const float *d_ysrc = nullptr) {
    const int64_t total = (int64_t)xf.ni * (2 * nc) * (2 * nc);
    if (!d_ysrc) d_ysrc = d_y;
    if (nc >= 16 && pm_aligned16(d_y) && pm_aligned16(d_xc) && pm_aligned16(d_ysrc)) {
        hipLaunchKernelGGL(prolongation_rows8_kernel, dim3(pm_stream_blocks(total / 8, BLK)), dim3(BLK), 0, c->stream, d_y,
                           d_ysrc, d_xc, ilog2(nc), add, xf, xcr);
        PM_LAUNCH_CHECK();
        return PM_OK;
    }
    if (nc >= 8 && pm_aligned16(d_y) && pm_aligned16(d_xc) && pm_aligned16(d_ysrc)) {
        hipLaunchKernelGGL(prolongation_rows_kernel, dim3(pm_stream_blocks(total / 4, BLK)), dim3(BLK), 0, c->stream, d_y,
                           d_ysrc, d_xc, ilog2(nc), add, xf, xcr);
        PM_LAUNCH_CHECK();
        return PM_OK;
    }
    if (d_ysrc != d_y) {
        pm_set_error("out-of-place prolongation needs nc >= 8 and 16-byte aligned arrays");
        return PM_EINVAL;
    }
    hipLaunchKernelGGL(prolongation_kernel, dim3(pm_stream_blocks(total, BLK)), dim3(BLK), 0, c->stream, d_y, d_xc,
                       ilog2(nc), add, xf, xcr);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

extern "C" int pm_prolongation(pm_ctx *c, float *d_y, const float *d_xc, int nc, int add) {
    PM_REQUIRE(c && d_y && d_xc, "NULL argument");
    PM_CHECK_GRID(nc);
    return prolongation_launch(c, d_y, d_xc, nc, add, xr_cube(2 * nc), xr_cube(nc));
}

extern "C" int pm_prolongation_slab(pm_ctx *c, float *d_y, const float *d_xc, int nc, int nxc, int add) {
    PM_REQUIRE(c && d_y && d_xc && nxc >= 1, "bad argument");
    PM_CHECK_GRID(nc);
    return prolongation_launch(c, d_y, d_xc, nc, add, xr_slab(2 * nxc), xr_slab(nxc));
}

// ------------------------------------------------------------------------------------------------
// 7-point Laplacian

// laplacian.py:12-59 operator (mode 0) / :63-121 residual b - Ax (mode 1)
__global__ __launch_bounds__(BLK) void lap_operator_kernel(const float *__restrict__ x, const float *__restrict__ b,
                                                           int log2n, int mode, float invh2,
                                                           float *__restrict__ out, XRange xr) {
    const int n = 1 << log2n, m = n - 1;
    const int64_t total = (int64_t)xr.ni << (2 * log2n);
    for (int64_t t0 = (int64_t)blockIdx.x * BLK + threadIdx.x; t0 < total; t0 += (int64_t)gridDim.x * BLK) {
        const int k = (int)(t0 & m), j = (int)((t0 >> log2n) & m), i = xr.i0 + (int)(t0 >> (2 * log2n));
        const int64_t t = cidx(i, j, k, log2n);
        const float s = x[cidx((i - 1) & xr.xmask, j, k, log2n)] + x[cidx(i, (j - 1) & m, k, log2n)] +
                        x[cidx(i, j, (k - 1) & m, log2n)] - 6.0f * x[t] + x[cidx(i, j, (k + 1) & m, log2n)] +
                        x[cidx(i, (j + 1) & m, k, log2n)] + x[cidx((i + 1) & xr.xmask, j, k, log2n)];
        out[t] = mode ? (-(s)*invh2 + b[t]) : s * invh2;
    }
}

static bool lap_operator_march(pm_ctx *c, const float *d_x, const float *d_b, int n, int mode, float *d_out, XRange xr);

static int lap_operator_launch(pm_ctx *c, const float *d_x, const float *d_b, int n, int mode, float *d_out, XRange xr) {
    const int64_t total = (int64_t)xr.ni * n * n;
    if (lap_operator_march(c, d_x, d_b, n, mode, d_out, xr)) {  // (from 512^3: see lap_residual_error_march_kernel)
        PM_LAUNCH_CHECK();
        return PM_OK;
    }
    hipLaunchKernelGGL(lap_operator_kernel, dim3(pm_stream_blocks(total, BLK)), dim3(BLK), 0, c->stream, d_x, d_b,
                       ilog2(n), mode, (float)((double)n * n), d_out, xr);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

extern "C" int pm_lap_operator(pm_ctx *c, const float *d_x, const float *d_b, int n, int mode, float *d_out) {
    PM_REQUIRE(c && d_x && d_out && (mode == 0 || d_b), "NULL argument");
    PM_CHECK_GRID(n);
    return lap_operator_launch(c, d_x, d_b, n, mode, d_out, xr_cube(n));
}

extern "C" int pm_lap_operator_slab(pm_ctx *c, const float *d_x, const float *d_b, int n, int nx, int mode,
                                    float *d_out) {
    PM_REQUIRE(c && d_x && d_out && (mode == 0 || d_b) && nx >= 1, "bad argument");
    PM_CHECK_GRID(n);
    return lap_operator_launch(c, d_x, d_b, n, mode, d_out, xr_slab(nx));
}

// laplacian.py:125-226 restrict_residual: fused residual + 1/8 restriction, reference summation order
__device__ __forceinline__ void lap_restrict_residual_body(const float *__restrict__ x, const float *__restrict__ b,
                                                           int log2n, float invh2, float *__restrict__ out, XRange xf,
                                                           XRange xc, int64_t first, int64_t stride) {
    const int n = 1 << log2n, m = n - 1, log2nc = log2n - 1, nc = n >> 1;
    const int64_t total = (int64_t)xc.ni << (2 * log2nc);
#define X(a, bb, cc) x[cidx((a)&xf.xmask, (bb)&m, (cc)&m, log2n)]
#define B(a, bb, cc) b[cidx(a, bb, cc, log2n)]
    for (int64_t t = first; t < total; t += stride) {
        const int k = (int)(t & (nc - 1)), j = (int)((t >> log2nc) & (nc - 1)), i = (int)(t >> (2 * log2nc));
        const int ii = xf.i0 + 2 * i, jj = 2 * j, kk = 2 * k;
        const int m1 = ii - 1, p1 = ii + 1, p2 = ii + 2, jm1 = jj - 1, jp1 = jj + 1, jp2 = jj + 2, km1 = kk - 1,
                  kp1 = kk + 1, kp2 = kk + 2;
        const float s = +X(m1, jj, kk) + X(m1, jj, kp1) + X(m1, jp1, kk) + X(m1, jp1, kp1) + X(ii, jm1, kk) +
                        X(ii, jm1, kp1) + X(ii, jj, km1) + X(ii, jj, kp2) + X(ii, jp1, km1) + X(ii, jp1, kp2) +
                        X(ii, jp2, kk) + X(ii, jp2, kp1) + X(p1, jm1, kk) + X(p1, jm1, kp1) + X(p1, jj, km1) +
                        X(p1, jj, kp2) + X(p1, jp1, km1) + X(p1, jp1, kp2) + X(p1, jp2, kk) + X(p1, jp2, kp1) +
                        X(p2, jj, kk) + X(p2, jj, kp1) + X(p2, jp1, kk) + X(p2, jp1, kp1) -
                        3.0f * (X(ii, jj, kk) + X(ii, jj, kp1) + X(ii, jp1, kk) + X(ii, jp1, kp1) + X(p1, jj, kk) +
                                X(p1, jj, kp1) + X(p1, jp1, kk) + X(p1, jp1, kp1));
        out[cidx(xc.i0 + i, j, k, log2nc)] =
            0.125f * (-(s)*invh2 + B(ii, jj, kk) + B(ii, jj, kp1) + B(ii, jp1, kk) + B(ii, jp1, kp1) + B(p1, jj, kk) +
                      B(p1, jj, kp1) + B(p1, jp1, kk) + B(p1, jp1, kp1));
    }
#undef X
#undef B
}

__global__ __launch_bounds__(BLK) void lap_restrict_residual_kernel(const float *__restrict__ x,
                                                                    const float *__restrict__ b, int log2n,
                                                                    float invh2, float *__restrict__ out, XRange xf,
                                                                    XRange xc) {
    lap_restrict_residual_body(x, b, log2n, invh2, out, xf, xc, (int64_t)blockIdx.x * BLK + threadIdx.x,
                               (int64_t)gridDim.x * BLK);
}

// The same with 16-byte accesses (n >= 16): a thread computes two adjacent coarse cells (four fine k) from twelve
// float4 of x (+ eight halo scalars) and four of b, instead of 40 scalar loads per coarse cell.  Same expression.
__global__ __launch_bounds__(BLK) void lap_restrict_residual_rows_kernel(const float *__restrict__ x,
                                                                         const float *__restrict__ b, int log2n,
                                                                         float invh2, float *__restrict__ out, XRange xf,
                                                                         XRange xc) {
    const int n = 1 << log2n, m = n - 1, log2nc = log2n - 1, nc = n >> 1, lq = log2nc - 1;
    const int64_t total = ((int64_t)xc.ni << log2nc) << lq;
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLK) {
        const int q = (int)(t & ((1 << lq) - 1)), j = (int)((t >> lq) & (nc - 1)), i = (int)(t >> (lq + log2nc));
        const int ii = xf.i0 + 2 * i, jj = 2 * j, k0 = 4 * q;
        auto row = [&](const float *g, int a, int bb) -> const float * {
            return g + ((((int64_t)(a & xf.xmask) << log2n) + (bb & m)) << log2n);
        };
        auto ld4 = [&](const float *r, float *d) {  // d[1..4] = r[k0 .. k0+3]
            const float4 v = *reinterpret_cast<const float4 *>(r + k0);
            d[1] = v.x, d[2] = v.y, d[3] = v.z, d[4] = v.w;
        };
        auto ld6 = [&](const float *r, float *d) {  // d[0..5] = r[k0-1 .. k0+4]
            ld4(r, d);
            d[0] = r[(k0 - 1) & m], d[5] = r[(k0 + 4) & m];
        };
        float A[2][6], I[4][6], P[4][6], Q[2][6], Bi[2][6], Bp[2][6];
        ld4(row(x, ii - 1, jj), A[0]), ld4(row(x, ii - 1, jj + 1), A[1]);
        ld4(row(x, ii, jj - 1), I[0]), ld6(row(x, ii, jj), I[1]), ld6(row(x, ii, jj + 1), I[2]), ld4(row(x, ii, jj + 2), I[3]);
        ld4(row(x, ii + 1, jj - 1), P[0]), ld6(row(x, ii + 1, jj), P[1]), ld6(row(x, ii + 1, jj + 1), P[2]);
        ld4(row(x, ii + 1, jj + 2), P[3]);
        ld4(row(x, ii + 2, jj), Q[0]), ld4(row(x, ii + 2, jj + 1), Q[1]);
        ld4(row(b, ii, jj), Bi[0]), ld4(row(b, ii, jj + 1), Bi[1]), ld4(row(b, ii + 1, jj), Bp[0]);
        ld4(row(b, ii + 1, jj + 1), Bp[1]);
        float r2[2];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const int kk = 1 + 2 * c2, kp1 = kk + 1, km1 = kk - 1, kp2 = kk + 2;  // indices into the d[] arrays
            const float s = +A[0][kk] + A[0][kp1] + A[1][kk] + A[1][kp1] + I[0][kk] + I[0][kp1] + I[1][km1] + I[1][kp2] +
                            I[2][km1] + I[2][kp2] + I[3][kk] + I[3][kp1] + P[0][kk] + P[0][kp1] + P[1][km1] + P[1][kp2] +
                            P[2][km1] + P[2][kp2] + P[3][kk] + P[3][kp1] + Q[0][kk] + Q[0][kp1] + Q[1][kk] + Q[1][kp1] -
                            3.0f * (I[1][kk] + I[1][kp1] + I[2][kk] + I[2][kp1] + P[1][kk] + P[1][kp1] + P[2][kk] +
                                    P[2][kp1]);
            r2[c2] = 0.125f * (-(s)*invh2 + Bi[0][kk] + Bi[0][kp1] + Bi[1][kk] + Bi[1][kp1] + Bp[0][kk] + Bp[0][kp1] +
                               Bp[1][kk] + Bp[1][kp1]);
        }
        *reinterpret_cast<float2 *>(out + cidx(xc.i0 + i, j, 2 * q, log2nc)) = make_float2(r2[0], r2[1]);
    }
}

static int lap_restrict_residual_launch(pm_ctx *c, const float *d_x, const float *d_b, int n, float *d_out, XRange xf,
                                        XRange xc) {
    const int64_t total = (int64_t)xc.ni * (n / 2) * (n / 2);
    if (n >= 16 && pm_aligned16(d_x) && pm_aligned16(d_b) && pm_aligned16(d_out)) {
        hipLaunchKernelGGL(lap_restrict_residual_rows_kernel, dim3(pm_stream_blocks(total / 2, BLK)), dim3(BLK), 0,
                           c->stream, d_x, d_b, ilog2(n), (float)((double)n * n), d_out, xf, xc);
        PM_LAUNCH_CHECK();
        return PM_OK;
    }
    hipLaunchKernelGGL(lap_restrict_residual_kernel, dim3(pm_stream_blocks(total, BLK)), dim3(BLK), 0, c->stream, d_x,
                       d_b, ilog2(n), (float)((double)n * n), d_out, xf, xc);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

extern "C" int pm_lap_restrict_residual(pm_ctx *c, const float *d_x, const float *d_b, int n, float *d_out) {
    PM_REQUIRE(c && d_x && d_b && d_out, "NULL argument");
    PM_CHECK_GRID(n);
    return lap_restrict_residual_launch(c, d_x, d_b, n, d_out, xr_cube(n), xr_cube(n / 2));
}

extern "C" int pm_lap_restrict_residual_slab(pm_ctx *c, const float *d_x, const float *d_b, int n, int nx,
                                             float *d_out) {
    PM_REQUIRE(c && d_x && d_b && d_out && nx >= 2 && (nx % 2) == 0, "bad argument (even number of owned planes)");
    PM_CHECK_GRID(n);
    return lap_restrict_residual_launch(c, d_x, d_b, n, d_out, xr_slab(nx), xr_slab(nx / 2));
}

// block-level fp64 sum -> one atomicAdd(double) per workgroup
__device__ __forceinline__ void block_sum_to(double v, double *result) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __shared__ double ws[BLK / 64];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < BLK / 64; ++w) v += ws[w];
        atomicAdd(result, v);
    }
}

static int read_norm(pm_ctx *c, float *h_out) {
    double s = 0;
    if (int rb_rc = pm_read_back(c, &s, c->d_hist, sizeof(double))) return rb_rc;
    *h_out = (float)sqrt(s);
    return PM_OK;
}

// laplacian.py:327-381 residual_error = sqrt(sum (b - Ax)^2)
__global__ __launch_bounds__(BLK) void lap_residual_error_kernel(const float *__restrict__ x,
                                                                 const float *__restrict__ b, int log2n, float invh2,
                                                                 double *__restrict__ result, XRange xr) {
    const int n = 1 << log2n, m = n - 1;
    const int64_t total = (int64_t)xr.ni << (2 * log2n);
    double acc = 0;
    for (int64_t t0 = (int64_t)blockIdx.x * BLK + threadIdx.x; t0 < total; t0 += (int64_t)gridDim.x * BLK) {
        const int k = (int)(t0 & m), j = (int)((t0 >> log2n) & m), i = xr.i0 + (int)(t0 >> (2 * log2n));
        const int64_t t = cidx(i, j, k, log2n);
        const float s = +x[cidx((i - 1) & xr.xmask, j, k, log2n)] + x[cidx(i, j, (k - 1) & m, log2n)] +
                        x[cidx(i, j, (k + 1) & m, log2n)] + x[cidx(i, (j - 1) & m, k, log2n)] - 6.0f * x[t] +
                        x[cidx(i, (j + 1) & m, k, log2n)] + x[cidx((i + 1) & xr.xmask, j, k, log2n)];
        const float r = -(s)*invh2 + b[t];
        acc += (double)(r * r);
    }
    block_sum_to(acc, result);
}

// eight k-cells per thread with 16-byte accesses (n >= 16); same expression per cell, squares summed in fp64
__global__ __launch_bounds__(BLK) void lap_residual_error_rows_kernel(const float *__restrict__ x,
                                                                      const float *__restrict__ b, int log2n, float invh2,
                                                                      double *__restrict__ result, XRange xr) {
    const int n = 1 << log2n, m = n - 1, lq = log2n - 3;
    const int64_t total = ((int64_t)xr.ni << log2n) << lq;
    double acc = 0;
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLK) {
        const int k0 = (int)(t & ((1 << lq) - 1)) << 3, j = (int)((t >> lq) & m), i = xr.i0 + (int)(t >> (lq + log2n));
        auto row = [&](const float *g, int a, int bb) -> const float * {
            return g + ((((int64_t)(a & xr.xmask) << log2n) + (bb & m)) << log2n);
        };
        auto ld8 = [&](const float *p, float *d) {
            const float4 q0 = *reinterpret_cast<const float4 *>(p + k0), q1 = *reinterpret_cast<const float4 *>(p + k0 + 4);
            d[0] = q0.x, d[1] = q0.y, d[2] = q0.z, d[3] = q0.w, d[4] = q1.x, d[5] = q1.y, d[6] = q1.z, d[7] = q1.w;
        };
        const float *mine = row(x, i, j);
        float v[10], axm[8], axp[8], aym[8], ayp[8], ab[8];
        ld8(mine, v + 1);
        ld8(row(x, i - 1, j), axm), ld8(row(x, i + 1, j), axp), ld8(row(x, i, j - 1), aym), ld8(row(x, i, j + 1), ayp);
        ld8(row(b, i, j), ab);
        v[0] = mine[(k0 - 1) & m], v[9] = mine[(k0 + 8) & m];
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) {
            const float s = +axm[c2] + v[c2] + v[c2 + 2] + aym[c2] - 6.0f * v[c2 + 1] + ayp[c2] + axp[c2];
            const float r = -(s)*invh2 + ab[c2];
            acc += (double)(r * r);
        }
    }
    block_sum_to(acc, result);
}

// The same for n >= 512, marching: a workgroup owns 16 rows (j) x 128 cells (k) and walks RM_LX planes along x; a thread
// keeps its eight cells of planes i - 1, i, i + 1 in registers (every plane of x is loaded once, as "i + 1"), the rows
// j - 1 / j + 1 and the cells k - 1 / k + 8 come from the other threads through LDS (the two rows and the cells just
// outside the workgroup's patch: loaded by its edge threads).  The row kernel above asks the L2 for six octets per
// octet of cells (its neighbours' rows again and again: 11 TB/s of L2 reads at 512^3, 0.29 ms); this one for a little
// over two.  Same expression per cell; the squares are summed in fp64 (in another order: the last bits of the fp64 sum).
constexpr int RM_TJ = 16, RM_TK = 128, RM_LX = 32;
__global__ __launch_bounds__(BLK) void lap_residual_error_march_kernel(const float *__restrict__ x, const float *__restrict__ b,
                                                                       int log2n, float invh2, double *__restrict__ result,
                                                                       XRange xr) {
    __shared__ __attribute__((aligned(16))) float sh[2][RM_TJ + 2][RM_TK + 8];  // [..][row jj + 1][4 + k]: cells k0 - 1 .. k0 + 128
    const int n = 1 << log2n, m = n - 1, tid = threadIdx.x;
    const int ncol_k = n / RM_TK, ncol_j = n / RM_TJ;
    const int col = blockIdx.x % (ncol_k * ncol_j), seg = blockIdx.x / (ncol_k * ncol_j);
    const int kb = (col % ncol_k) * RM_TK, j0 = (col / ncol_k) * RM_TJ;
    const int jj = tid >> 4, kq = tid & 15, j = j0 + jj, k0 = kb + 8 * kq;
    const int ia = xr.i0 + seg * RM_LX, ie = min(ia + RM_LX, xr.i0 + xr.ni);
    auto row = [&](const float *g, int a, int bb) -> const float * {
        return g + ((((int64_t)(a & xr.xmask) << log2n) + (bb & m)) << log2n);
    };
    auto ld8 = [&](const float *p, float *d) {
        const float4 q0 = *reinterpret_cast<const float4 *>(p), q1 = *reinterpret_cast<const float4 *>(p + 4);
        d[0] = q0.x, d[1] = q0.y, d[2] = q0.z, d[3] = q0.w, d[4] = q1.x, d[5] = q1.y, d[6] = q1.z, d[7] = q1.w;
    };
    // what a thread brings besides its own octet: a row outside the patch (jj = 0: j0 - 1, jj = 15: j0 + 16), a cell
    // outside it (kq = 0: kb - 1, kq = 15: kb + 128)
    const bool edge_j = jj == 0 || jj == RM_TJ - 1, edge_k = kq == 0 || kq == 15;
    const int jrow = jj == 0 ? j0 - 1 : j0 + RM_TJ, lrow = jj == 0 ? 0 : RM_TJ + 1;
    const int kcell = kq == 0 ? kb - 1 : kb + RM_TK, lcell = kq == 0 ? 3 : 4 + RM_TK;
    float prev[8], cur[8], next[8], erow[8], bb8[8], ecell = 0.f;
    if (ia >= ie) {  // (never with the launch below; keeps the barriers of the loop uniform)
        block_sum_to(0.0, result);
        return;
    }
    ld8(row(x, ia - 1, j) + k0, prev);
    ld8(row(x, ia, j) + k0, cur);
    double acc = 0;
    for (int i = ia; i < ie; ++i) {
        float *pl = &sh[i & 1][0][0];
        constexpr int RS = RM_TK + 8;
        ld8(row(x, i + 1, j) + k0, next);
        ld8(row(b, i, j) + k0, bb8);
        if (edge_j) ld8(row(x, i, jrow) + k0, erow);
        if (edge_k) ecell = row(x, i, j)[kcell & m];
        *reinterpret_cast<float4 *>(pl + (jj + 1) * RS + 4 + 8 * kq) = make_float4(cur[0], cur[1], cur[2], cur[3]);
        *reinterpret_cast<float4 *>(pl + (jj + 1) * RS + 8 + 8 * kq) = make_float4(cur[4], cur[5], cur[6], cur[7]);
        if (edge_j) {
            *reinterpret_cast<float4 *>(pl + lrow * RS + 4 + 8 * kq) = make_float4(erow[0], erow[1], erow[2], erow[3]);
            *reinterpret_cast<float4 *>(pl + lrow * RS + 8 + 8 * kq) = make_float4(erow[4], erow[5], erow[6], erow[7]);
        }
        if (edge_k) pl[(jj + 1) * RS + lcell] = ecell;
        __syncthreads();  // (one barrier per plane: the next plane goes to the other buffer)
        const float *me = pl + (jj + 1) * RS + 4 + 8 * kq, *ym = me - RS, *yp = me + RS;
        const float4 a0 = *reinterpret_cast<const float4 *>(ym), a1 = *reinterpret_cast<const float4 *>(ym + 4);
        const float4 c0 = *reinterpret_cast<const float4 *>(yp), c1 = *reinterpret_cast<const float4 *>(yp + 4);
        const float aym[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const float ayp[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        float v[10];
        v[0] = me[-1], v[9] = me[8];
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) v[c2 + 1] = cur[c2];
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) {
            const float s = +prev[c2] + v[c2] + v[c2 + 2] + aym[c2] - 6.0f * v[c2 + 1] + ayp[c2] + next[c2];
            const float r = -(s)*invh2 + bb8[c2];
            acc += (double)(r * r);
        }
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) prev[c2] = cur[c2], cur[c2] = next[c2];
    }
    block_sum_to(acc, result);
}

// laplacian.operator / residual (lap_operator_kernel) through the same march: same expression per cell
template <int MODE>
__global__ __launch_bounds__(BLK) void lap_operator_march_kernel(const float *__restrict__ x, const float *__restrict__ b,
                                                                 int log2n, float invh2, float *__restrict__ out, XRange xr) {
    __shared__ __attribute__((aligned(16))) float sh[2][RM_TJ + 2][RM_TK + 8];
    const int n = 1 << log2n, m = n - 1, tid = threadIdx.x;
    const int ncol_k = n / RM_TK, ncol_j = n / RM_TJ;
    const int col = blockIdx.x % (ncol_k * ncol_j), seg = blockIdx.x / (ncol_k * ncol_j);
    const int kb = (col % ncol_k) * RM_TK, j0 = (col / ncol_k) * RM_TJ;
    const int jj = tid >> 4, kq = tid & 15, j = j0 + jj, k0 = kb + 8 * kq;
    const int ia = xr.i0 + seg * RM_LX, ie = min(ia + RM_LX, xr.i0 + xr.ni);
    auto row = [&](const float *g, int a, int bb) -> const float * {
        return g + ((((int64_t)(a & xr.xmask) << log2n) + (bb & m)) << log2n);
    };
    auto ld8 = [&](const float *p, float *d) {
        const float4 q0 = *reinterpret_cast<const float4 *>(p), q1 = *reinterpret_cast<const float4 *>(p + 4);
        d[0] = q0.x, d[1] = q0.y, d[2] = q0.z, d[3] = q0.w, d[4] = q1.x, d[5] = q1.y, d[6] = q1.z, d[7] = q1.w;
    };
    const bool edge_j = jj == 0 || jj == RM_TJ - 1, edge_k = kq == 0 || kq == 15;
    const int jrow = jj == 0 ? j0 - 1 : j0 + RM_TJ, lrow = jj == 0 ? 0 : RM_TJ + 1;
    const int kcell = kq == 0 ? kb - 1 : kb + RM_TK, lcell = kq == 0 ? 3 : 4 + RM_TK;
    float prev[8], cur[8], next[8], erow[8], bb8[8], ecell = 0.f;
    if (ia < ie) {
        ld8(row(x, ia - 1, j) + k0, prev);
        ld8(row(x, ia, j) + k0, cur);
    }
    for (int i = ia; i < ie; ++i) {
        float *pl = &sh[i & 1][0][0];
        constexpr int RS = RM_TK + 8;
        ld8(row(x, i + 1, j) + k0, next);
        if (MODE) ld8(b + cidx(i, j, k0, log2n), bb8);
        if (edge_j) ld8(row(x, i, jrow) + k0, erow);
        if (edge_k) ecell = row(x, i, j)[kcell & m];
        *reinterpret_cast<float4 *>(pl + (jj + 1) * RS + 4 + 8 * kq) = make_float4(cur[0], cur[1], cur[2], cur[3]);
        *reinterpret_cast<float4 *>(pl + (jj + 1) * RS + 8 + 8 * kq) = make_float4(cur[4], cur[5], cur[6], cur[7]);
        if (edge_j) {
            *reinterpret_cast<float4 *>(pl + lrow * RS + 4 + 8 * kq) = make_float4(erow[0], erow[1], erow[2], erow[3]);
            *reinterpret_cast<float4 *>(pl + lrow * RS + 8 + 8 * kq) = make_float4(erow[4], erow[5], erow[6], erow[7]);
        }
        if (edge_k) pl[(jj + 1) * RS + lcell] = ecell;
        __syncthreads();
        const float *me = pl + (jj + 1) * RS + 4 + 8 * kq, *ym = me - RS, *yp = me + RS;
        const float4 a0 = *reinterpret_cast<const float4 *>(ym), a1 = *reinterpret_cast<const float4 *>(ym + 4);
        const float4 c0 = *reinterpret_cast<const float4 *>(yp), c1 = *reinterpret_cast<const float4 *>(yp + 4);
        const float aym[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const float ayp[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        float v[10], res[8];
        v[0] = me[-1], v[9] = me[8];
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) v[c2 + 1] = cur[c2];
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) {
            const float s = prev[c2] + aym[c2] + v[c2] - 6.0f * v[c2 + 1] + v[c2 + 2] + ayp[c2] + next[c2];
            res[c2] = MODE ? (-(s)*invh2 + bb8[c2]) : s * invh2;
        }
        float4 *o4 = reinterpret_cast<float4 *>(out + cidx(i, j, k0, log2n));
        o4[0] = make_float4(res[0], res[1], res[2], res[3]);
        o4[1] = make_float4(res[4], res[5], res[6], res[7]);
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) prev[c2] = cur[c2], cur[c2] = next[c2];
    }
}

static bool lap_operator_march(pm_ctx *c, const float *d_x, const float *d_b, int n, int mode, float *d_out, XRange xr) {
    if (!(n >= 512 && c->mg_fused_sweep && pm_aligned16(d_x) && pm_aligned16(d_b) && pm_aligned16(d_out))) return false;
    const unsigned blocks = (unsigned)((n / RM_TJ) * (n / RM_TK) * ((xr.ni + RM_LX - 1) / RM_LX));
    const float invh2 = (float)((double)n * n);
    if (mode)
        hipLaunchKernelGGL(lap_operator_march_kernel<1>, dim3(blocks), dim3(BLK), 0, c->stream, d_x, d_b, ilog2(n), invh2, d_out, xr);
    else
        hipLaunchKernelGGL(lap_operator_march_kernel<0>, dim3(blocks), dim3(BLK), 0, c->stream, d_x, d_b, ilog2(n), invh2, d_out, xr);
    return true;
}

static int lap_residual_sumsq(pm_ctx *c, const float *d_x, const float *d_b, int n, XRange xr, double *h_sum) {
    PM_HIP(hipMemsetAsync(c->d_hist, 0, sizeof(double), c->stream));
    const int64_t total = (int64_t)xr.ni * n * n;
    int blocks = pm_stream_blocks(total, BLK * 4);
    // (from 512^3: at 256^3 the patches x segments are one workgroup per CU and the row kernel is as fast)
    if (n >= 512 && c->mg_fused_sweep && pm_aligned16(d_x) && pm_aligned16(d_b))
        hipLaunchKernelGGL(lap_residual_error_march_kernel,
                           dim3((unsigned)((n / RM_TJ) * (n / RM_TK) * ((xr.ni + RM_LX - 1) / RM_LX))), dim3(BLK), 0, c->stream, d_x,
                           d_b, ilog2(n), (float)((double)n * n), c->d_hist, xr);
    else if (n >= 16 && pm_aligned16(d_x) && pm_aligned16(d_b))
        // (2048 workgroups at most: each ends with one same-address fp64 atomic, ~12 ns apiece)
        hipLaunchKernelGGL(lap_residual_error_rows_kernel, dim3(blocks > 2048 ? 2048 : blocks), dim3(BLK), 0, c->stream, d_x, d_b, ilog2(n),
                           (float)((double)n * n), c->d_hist, xr);
    else
        hipLaunchKernelGGL(lap_residual_error_kernel, dim3(blocks), dim3(BLK), 0, c->stream, d_x, d_b, ilog2(n),
                           (float)((double)n * n), c->d_hist, xr);
    PM_LAUNCH_CHECK();
    if (int rb_rc = pm_read_back(c, h_sum, c->d_hist, sizeof(double))) return rb_rc;
    return PM_OK;
}

extern "C" int pm_lap_residual_error(pm_ctx *c, const float *d_x, const float *d_b, int n, float *h_out) {
    PM_REQUIRE(c && d_x && d_b && h_out, "NULL argument");
    PM_CHECK_GRID(n);
    double s = 0;
    int rc = lap_residual_sumsq(c, d_x, d_b, n, xr_cube(n), &s);
    *h_out = (float)sqrt(s);
    return rc;
}

// slab: the local sum of squares (fp64); the caller adds the ranks' sums and takes the root
extern "C" int pm_lap_residual_sumsq_slab(pm_ctx *c, const float *d_x, const float *d_b, int n, int nx, double *h_sum) {
    PM_REQUIRE(c && d_x && d_b && h_sum && nx >= 1, "bad argument");
    PM_CHECK_GRID(n);
    return lap_residual_sumsq(c, d_x, d_b, n, xr_slab(nx), h_sum);
}

// sqrt(sum (f*a - b)^2): tail of laplacian.truncation_error :529-532 (f=1), cubic.truncation_error :1057-1061 (f=4)
__global__ __launch_bounds__(BLK) void diff_norm_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                        float f, int64_t count, double *__restrict__ result) {
    double acc = 0;
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < count; t += (int64_t)gridDim.x * BLK) {
        const float d = f * a[t] - b[t];
        acc += (double)(d * d);
    }
    block_sum_to(acc, result);
}

extern "C" int pm_diff_norm(pm_ctx *c, const float *d_a, const float *d_b, float f, int64_t count, float *h_out) {
    PM_REQUIRE(c && d_a && d_b && h_out && count > 0, "bad argument");
    PM_HIP(hipMemsetAsync(c->d_hist, 0, sizeof(double), c->stream));
    hipLaunchKernelGGL(diff_norm_kernel, dim3(pm_stream_blocks(count, BLK * 4)), dim3(BLK), 0, c->stream, d_a, d_b, f,
                       count, c->d_hist);
    PM_LAUNCH_CHECK();
    return read_norm(c, h_out);
}

// the same without the root, in fp64 (slab ranks add their sums)
extern "C" int pm_diff_sumsq(pm_ctx *c, const float *d_a, const float *d_b, float f, int64_t count, double *h_sum) {
    PM_REQUIRE(c && d_a && d_b && h_sum && count > 0, "bad argument");
    PM_HIP(hipMemsetAsync(c->d_hist, 0, sizeof(double), c->stream));
    hipLaunchKernelGGL(diff_norm_kernel, dim3(pm_stream_blocks(count, BLK * 4)), dim3(BLK), 0, c->stream, d_a, d_b, f,
                       count, c->d_hist);
    PM_LAUNCH_CHECK();
    if (int rb_rc = pm_read_back(c, h_sum, c->d_hist, sizeof(double))) return rb_rc;
    return PM_OK;
}

// laplacian.py:844-1022 gauss_seidel.  One launch per colour (the reference warns that its two
// colour loops must never fuse, laplacian.py:870-871).  Thread = one 2x2x2 block; it updates the
// four cells of the launch's colour, each with the reference's own summation order for that
// parity class (dirs: 0 xm, 1 xp, 2 ym, 3 yp, 4 zm, 5 zp, 6 -h2*b).
template <int O0, int O1, int O2, int O3, int O4, int O5, int O6>
__device__ __forceinline__ void gs_cell(float *__restrict__ x, const float *__restrict__ b, int i, int j, int k, int m,
                                        int log2n, float h2, float invsix, float f_relax) {
    const int64_t c = cidx(i, j, k, log2n);
    float t[7];
    t[0] = x[cidx((i - 1) & m, j, k, log2n)];
    t[1] = x[cidx((i + 1) & m, j, k, log2n)];
    t[2] = x[cidx(i, (j - 1) & m, k, log2n)];
    t[3] = x[cidx(i, (j + 1) & m, k, log2n)];
    t[4] = x[cidx(i, j, (k - 1) & m, log2n)];
    t[5] = x[cidx(i, j, (k + 1) & m, log2n)];
    t[6] = -(h2 * b[c]);
    const float s = (((((t[O0] + t[O1]) + t[O2]) + t[O3]) + t[O4]) + t[O5]) + t[O6];
    const float xc = x[c];
    x[c] = xc + f_relax * (s * invsix - xc);
}

template <int COLOUR>
__device__ __forceinline__ void lap_gauss_seidel_body(float *__restrict__ x, const float *__restrict__ b, int log2n, float h2,
                                                      float invsix, float f_relax, int64_t first, int64_t stride) {
    const int n = 1 << log2n, m = n - 1, log2nc = log2n - 1, nc = n >> 1;
    const int64_t total = (int64_t)nc * nc * nc;
    for (int64_t t = first; t < total; t += stride) {
        const int k = 2 * (int)(t & (nc - 1)), j = 2 * (int)((t >> log2nc) & (nc - 1)), i = 2 * (int)(t >> (2 * log2nc));
        if (COLOUR == 1) {  // red: i+j+k odd
            gs_cell<5, 3, 1, 0, 2, 4, 6>(x, b, i + 1, j + 1, k + 1, m, log2n, h2, invsix, f_relax);  // 111
            gs_cell<2, 4, 1, 0, 6, 5, 3>(x, b, i + 1, j, k, m, log2n, h2, invsix, f_relax);          // 100
            gs_cell<0, 3, 4, 1, 2, 5, 6>(x, b, i, j + 1, k, m, log2n, h2, invsix, f_relax);          // 010
            gs_cell<0, 2, 5, 4, 6, 3, 1>(x, b, i, j, k + 1, m, log2n, h2, invsix, f_relax);          // 001
        } else {  // black: even
            gs_cell<4, 3, 1, 0, 2, 6, 5>(x, b, i + 1, j + 1, k, m, log2n, h2, invsix, f_relax);      // 110
            gs_cell<2, 5, 1, 0, 4, 6, 3>(x, b, i + 1, j, k + 1, m, log2n, h2, invsix, f_relax);      // 101
            gs_cell<0, 5, 3, 2, 4, 6, 1>(x, b, i, j + 1, k + 1, m, log2n, h2, invsix, f_relax);      // 011
            gs_cell<0, 2, 4, 6, 5, 3, 1>(x, b, i, j, k, m, log2n, h2, invsix, f_relax);              // 000
        }
    }
}

template <int COLOUR>
__global__ __launch_bounds__(BLK) void lap_gauss_seidel_kernel(float *__restrict__ x, const float *__restrict__ b,
                                                               int log2n, float h2, float invsix, float f_relax) {
    lap_gauss_seidel_body<COLOUR>(x, b, log2n, h2, invsix, f_relax, (int64_t)blockIdx.x * BLK + threadIdx.x,
                                  (int64_t)gridDim.x * BLK);
}

// Row form of the same half-sweep for n >= 16: a thread owns 8 consecutive k-cells of one (i, j) row, four
// of which have the launch's colour and all four the same parity class (so one summation order per thread).
// Everything is 16-byte accesses -- the row itself, its four neighbour rows, the rhs -- against 28 scalar
// stride-2 loads per 2x2x2 block above, which left the kernel L1-bound at a quarter of the HBM rate.  The
// four cells of the other colour are written back unchanged (nobody writes them during this launch).
template <int O0, int O1, int O2, int O3, int O4, int O5, int O6>
__device__ __forceinline__ float gs_update(float xm, float xp, float ym, float yp, float zm, float zp, float mh2b,
                                           float xc, float invsix, float f_relax) {
    const float t[7] = {xm, xp, ym, yp, zm, zp, mh2b};
    const float s = (((((t[O0] + t[O1]) + t[O2]) + t[O3]) + t[O4]) + t[O5]) + t[O6];
    return xc + f_relax * (s * invsix - xc);
}

template <int COLOUR>
__global__ __launch_bounds__(BLK) void lap_gauss_seidel_rows_kernel(float *__restrict__ x, const float *__restrict__ b,
                                                                    int log2n, float h2, float invsix, float f_relax,
                                                                    XRange xr, int gpar) {
    // gpar: parity of (global plane index - local plane index), 0 on a single GPU
    const int n = 1 << log2n, m = n - 1, lseg = log2n - 3;
    const int64_t total = ((int64_t)xr.ni << log2n) << lseg;
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLK) {
        const int k0 = (int)(t & ((1 << lseg) - 1)) << 3;
        const int64_t row0 = t >> lseg;
        const int j = (int)(row0 & m), il = xr.i0 + (int)(row0 >> log2n), i = il + gpar;  // i: parity of the global plane
        const int64_t row = ((int64_t)il << log2n) + j;
        auto rowp = [&](int ii, int jj) -> const float * {
            return x + ((((int64_t)(ii & xr.xmask) << log2n) + (jj & m)) << log2n);
        };
        float *mine = x + (row << log2n);
        float v[10], xm[8], xp[8], ym[8], yp[8], bb[8];
        auto ld8 = [&](const float *p, float *d) {
            const float4 q0 = *reinterpret_cast<const float4 *>(p + k0), q1 = *reinterpret_cast<const float4 *>(p + k0 + 4);
            d[0] = q0.x, d[1] = q0.y, d[2] = q0.z, d[3] = q0.w, d[4] = q1.x, d[5] = q1.y, d[6] = q1.z, d[7] = q1.w;
        };
        ld8(mine, v + 1);
        v[0] = mine[(k0 - 1) & m], v[9] = mine[(k0 + 8) & m];
        ld8(rowp(il - 1, j), xm), ld8(rowp(il + 1, j), xp), ld8(rowp(il, j - 1), ym), ld8(rowp(il, j + 1), yp);
        ld8(b + (row << log2n), bb);
        const int o = (COLOUR + i + j) & 1;  // k parity of this row's cells of the launch's colour
        const int cls = ((i & 1) << 2) | ((j & 1) << 1) | o;
#define GS_ROW(O0, O1, O2, O3, O4, O5, O6)                                                                         \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                \
        const int l = 2 * q + o;                                                                                   \
        v[l + 1] = gs_update<O0, O1, O2, O3, O4, O5, O6>(xm[l], xp[l], ym[l], yp[l], v[l], v[l + 2], -(h2 * bb[l]),  \
                                                         v[l + 1], invsix, f_relax);                              \
    }
        // same orders as gs_cell above, by (i & 1, j & 1, k & 1)
        switch (cls) {
            case 7: GS_ROW(5, 3, 1, 0, 2, 4, 6) break;  // 111
            case 4: GS_ROW(2, 4, 1, 0, 6, 5, 3) break;  // 100
            case 2: GS_ROW(0, 3, 4, 1, 2, 5, 6) break;  // 010
            case 1: GS_ROW(0, 2, 5, 4, 6, 3, 1) break;  // 001
            case 6: GS_ROW(4, 3, 1, 0, 2, 6, 5) break;  // 110
            case 5: GS_ROW(2, 5, 1, 0, 4, 6, 3) break;  // 101
            case 3: GS_ROW(0, 5, 3, 2, 4, 6, 1) break;  // 011
            default: GS_ROW(0, 2, 4, 6, 5, 3, 1) break;  // 000
        }
#undef GS_ROW
        *reinterpret_cast<float4 *>(mine + k0) = make_float4(v[1], v[2], v[3], v[4]);
        *reinterpret_cast<float4 *>(mine + k0 + 4) = make_float4(v[5], v[6], v[7], v[8]);
    }
}

extern "C" int pm_lap_gauss_seidel(pm_ctx *c, float *d_x, const float *d_b, int n, float f_relax) {
    PM_REQUIRE(c && d_x && d_b, "NULL argument");
    PM_CHECK_GRID(n);
    const float h2 = (float)(1.0 / ((double)n * n)), invsix = (float)(1.0 / 6);
    if (n >= 16 && pm_aligned16(d_x) && pm_aligned16(d_b)) {
        const int64_t threads = (int64_t)n * n * (n / 8);
        const int blocks = pm_stream_blocks(threads, BLK);
        hipLaunchKernelGGL((lap_gauss_seidel_rows_kernel<1>), dim3(blocks), dim3(BLK), 0, c->stream, d_x, d_b, ilog2(n), h2,
                           invsix, f_relax, xr_cube(n), 0);
        hipLaunchKernelGGL((lap_gauss_seidel_rows_kernel<0>), dim3(blocks), dim3(BLK), 0, c->stream, d_x, d_b, ilog2(n), h2,
                           invsix, f_relax, xr_cube(n), 0);
        PM_LAUNCH_CHECK();
        return PM_OK;
    }
    const int64_t total = (int64_t)(n / 2) * (n / 2) * (n / 2);
    const int blocks = pm_stream_blocks(total, BLK);
    hipLaunchKernelGGL((lap_gauss_seidel_kernel<1>), dim3(blocks), dim3(BLK), 0, c->stream, d_x, d_b, ilog2(n), h2,
                       invsix, f_relax);
    hipLaunchKernelGGL((lap_gauss_seidel_kernel<0>), dim3(blocks), dim3(BLK), 0, c->stream, d_x, d_b, ilog2(n), h2,
                       invsix, f_relax);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

static int mg_arena(pm_ctx *c, size_t bytes, float **out) {
    if (c->mg_buf_bytes < bytes) {
        if (c->mg_buf) PM_HIP(hipFree(c->mg_buf));
        c->mg_buf = nullptr, c->mg_buf_bytes = 0;
        PM_HIP(hipMalloc(&c->mg_buf, bytes));
        c->mg_buf_bytes = bytes;
    }
    *out = (float *)c->mg_buf;
    return PM_OK;
}

// ------------------------------------------------------------------------------------------------
// Red + black sweep in ONE pass over the grid (n >= 256), out of place: x_in, b read once, x_out written once --
// 12 B per cell plus the halos, where the two half-sweep launches above move 24 (each reads x and b and writes x
// whole).  A workgroup owns a column of 16 x 64 cells (j, k) and marches 32 planes along x with five planes of x in
// LDS (each with a 2-cell rim, rows of 72 floats = 18 aligned float4): at step i it turns the red cells of plane
// i + 1 to their new values in LDS (they only read black cells: the old ones of planes i, i + 1, i + 2), then computes
// the black cells of plane i from the new reds of planes i - 1, i, i + 1 and stores plane i.  The next plane and the
// right-hand sides travel one step ahead in registers.  Every cell is updated by gs_update with the summation order
// of its parity class: the same bits as the two launches.  (Round 1's version of this -- no register prefetch,
// whole-x columns, three barriers per plane -- was latency-bound and slower than the half-sweeps.)
constexpr int GF_TJ = 16, GF_TK = 64, GF_ROW = GF_TK + 8, GF_ROWS = GF_TJ + 4, GF_PLANE = GF_ROW * GF_ROWS;
constexpr int GF_SLOTS = 5, GF_Q4 = GF_ROWS * (GF_ROW / 4);  // 360 float4 per plane
constexpr int GF_RING = 2 * (GF_TK + 2) + 2 * GF_TJ;         // 164 cells around the own region

__device__ __forceinline__ float gs_update_cls(int cls, float xm, float xp, float ym, float yp, float zm, float zp,
                                               float mh2b, float xc, float invsix, float f_relax) {
    switch (cls) {  // (i & 1, j & 1, k & 1): the orders of gs_cell / GS_ROW
        case 7: return gs_update<5, 3, 1, 0, 2, 4, 6>(xm, xp, ym, yp, zm, zp, mh2b, xc, invsix, f_relax);
        case 4: return gs_update<2, 4, 1, 0, 6, 5, 3>(xm, xp, ym, yp, zm, zp, mh2b, xc, invsix, f_relax);
        case 2: return gs_update<0, 3, 4, 1, 2, 5, 6>(xm, xp, ym, yp, zm, zp, mh2b, xc, invsix, f_relax);
        case 1: return gs_update<0, 2, 5, 4, 6, 3, 1>(xm, xp, ym, yp, zm, zp, mh2b, xc, invsix, f_relax);
        case 6: return gs_update<4, 3, 1, 0, 2, 6, 5>(xm, xp, ym, yp, zm, zp, mh2b, xc, invsix, f_relax);
        case 5: return gs_update<2, 5, 1, 0, 4, 6, 3>(xm, xp, ym, yp, zm, zp, mh2b, xc, invsix, f_relax);
        case 3: return gs_update<0, 5, 3, 2, 4, 6, 1>(xm, xp, ym, yp, zm, zp, mh2b, xc, invsix, f_relax);
        default: return gs_update<0, 2, 4, 6, 5, 3, 1>(xm, xp, ym, yp, zm, zp, mh2b, xc, invsix, f_relax);
    }
}

// The cell update of the linear sweep: gs_update with the summation order of the cell's parity class.
struct LapCell {
    static constexpr bool HAS_RHS = false;
    float h2, invsix, f_relax;
    __device__ __forceinline__ float operator()(int cls, float xm, float xp, float ym, float yp, float zm, float zp, float bval,
                                                float, float xc) const {
        return gs_update_cls(cls, xm, xp, ym, yp, zm, zp, -(h2 * bval), xc, invsix, f_relax);
    }
};

// GF_LX planes per workgroup: 32, or 64 from 512^3 up (three planes of lead-in per segment instead of three per 32:
// 0.42 -> 0.40 ms per sweep at 512^3; at 256^3 64 planes leave 256 workgroups for 256 CUs and the V-cycle loses 4 %).
// CELL: the update of one cell from its six neighbours, its density term b (and right-hand side, CELL::HAS_RHS) and
// its own value -- LapCell here, FrCell (the f(R) scalaron's cubic / quartic root) further down.
template <int GF_LX, class CELL>
__global__ __launch_bounds__(BLK) void gs_fused_kernel(const float *__restrict__ xin, const float *__restrict__ b,
                                                       const float *__restrict__ rhs, float *__restrict__ xout, int log2n,
                                                       CELL cell) {
    __shared__ __attribute__((aligned(16))) float pl[GF_SLOTS][GF_PLANE];  // 28.8 KB
    const int n = 1 << log2n, m = n - 1, tid = threadIdx.x;
    const int ncol_k = n / GF_TK, ncol_j = n / GF_TJ;
    const int col = blockIdx.x % (ncol_k * ncol_j), seg = blockIdx.x / (ncol_k * ncol_j);
    const int k0 = (col % ncol_k) * GF_TK, j0 = (col / ncol_k) * GF_TJ, i0 = seg * GF_LX;
    auto slot = [&](int p) -> float * { return pl[((p % GF_SLOTS) + GF_SLOTS) % GF_SLOTS]; };  // p relative to i0, >= -2

    // global -> registers -> LDS of one raw plane (relative index p): float4 number f of the plane, 2 per thread
    const int f0 = tid, f1 = tid + BLK;
    auto q_src = [&](int p, int f) -> const float4 * {
        const int rr = f / (GF_ROW / 4), c4 = f - rr * (GF_ROW / 4);
        return reinterpret_cast<const float4 *>(xin + cidx((i0 + p) & m, (j0 - 2 + rr) & m, (k0 - 4 + 4 * c4) & m, log2n));
    };
    auto q_put = [&](int p, int f, float4 v) {
        const int rr = f / (GF_ROW / 4), c4 = f - rr * (GF_ROW / 4);
        *reinterpret_cast<float4 *>(slot(p) + rr * GF_ROW + 4 * c4) = v;
    };
    // the thread's own quad: row rq (0..15), k 4 kq .. 4 kq + 3 of the own region
    const int rq = tid >> 4, kq = tid & 15;
    const int jq = j0 + rq, lrow = (rq + 2) * GF_ROW, lcol = 4 + 4 * kq;
    auto b_quad = [&](int p) -> float4 {
        return *reinterpret_cast<const float4 *>(b + cidx((i0 + p) & m, jq, k0 + 4 * kq, log2n));
    };
    auto r_quad = [&](int p) -> float4 {
        if constexpr (CELL::HAS_RHS) return *reinterpret_cast<const float4 *>(rhs + cidx((i0 + p) & m, jq, k0 + 4 * kq, log2n));
        return make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // the thread's cell of the ring around the own region (tid < GF_RING): LDS row / column and global j, k
    int ring_r = 0, ring_c = 0;
    if (tid < GF_TK + 2) ring_r = 1, ring_c = 3 + tid;
    else if (tid < 2 * (GF_TK + 2)) ring_r = GF_TJ + 2, ring_c = 3 + tid - (GF_TK + 2);
    else if (tid < 2 * (GF_TK + 2) + GF_TJ) ring_r = 2 + tid - 2 * (GF_TK + 2), ring_c = 3;
    else ring_r = 2 + tid - 2 * (GF_TK + 2) - GF_TJ, ring_c = GF_TK + 4;
    const int ring_j = (j0 - 2 + ring_r) & m, ring_k = (k0 - 4 + ring_c) & m;
    auto b_ring = [&](int p) -> float {
        return tid < GF_RING ? b[cidx((i0 + p) & m, ring_j, ring_k, log2n)] : 0.f;
    };
    auto r_ring = [&](int p) -> float {
        if constexpr (CELL::HAS_RHS) return tid < GF_RING ? rhs[cidx((i0 + p) & m, ring_j, ring_k, log2n)] : 0.f;
        return 0.f;
    };
    // new value of the cell at LDS (row offset ro, column c) of plane p (relative), parities from the global indices
    auto update = [&](int p, int ro, int c, int gj, int gk, float bval, float rval) -> float {
        const float *vm = slot(p - 1), *v0 = slot(p), *vp = slot(p + 1);
        const int o = ro + c;
        const int cls = ((((i0 + p) & 1) << 2) | ((gj & 1) << 1) | (gk & 1));
        return cell(cls, vm[o], vp[o], v0[o - GF_ROW], v0[o + GF_ROW], v0[o - 1], v0[o + 1], bval, rval, v0[o]);
    };
    // red cells (i + j + k odd) of plane p, in place in LDS: the two of the own quad + the ring cell if it is red
    auto red_phase = [&](int p, float4 bq, float br, float4 rq4, float rring) {
        const int e0 = ((i0 + p) + jq + 1) & 1;  // k offsets e0, e0 + 2 of the quad are red
        const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
        const float rb[4] = {rq4.x, rq4.y, rq4.z, rq4.w};
        float *v0 = slot(p);
        const float r0 = update(p, lrow, lcol + e0, jq, e0, bb[e0], rb[e0]);
        const float r1 = update(p, lrow, lcol + e0 + 2, jq, e0, bb[e0 + 2], rb[e0 + 2]);
        float rr = 0.f;
        const bool ring_red = tid < GF_RING && (((i0 + p) + ring_j + ring_k) & 1);
        if (ring_red) rr = update(p, ring_r * GF_ROW, ring_c, ring_j, ring_k, br, rring);
        v0[lrow + lcol + e0] = r0, v0[lrow + lcol + e0 + 2] = r1;
        if (ring_red) v0[ring_r * GF_ROW + ring_c] = rr;
    };

    // prologue: planes -2, -1, 0 -> R(-1); plane 1 -> R(0)
    float4 pf0, pf1 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = -2; p <= 0; ++p) {
        q_put(p, f0, *q_src(p, f0));
        if (f1 < GF_Q4) q_put(p, f1, *q_src(p, f1));
    }
    float4 bq_m1 = b_quad(-1), bq_0 = b_quad(0), rq_m1 = r_quad(-1), rq_0 = r_quad(0);
    float br_m1 = b_ring(-1), br_0 = b_ring(0), rr_m1 = r_ring(-1), rr_0 = r_ring(0);
    pf0 = *q_src(1, f0);
    if (f1 < GF_Q4) pf1 = *q_src(1, f1);
    __syncthreads();
    red_phase(-1, bq_m1, br_m1, rq_m1, rr_m1);
    q_put(1, f0, pf0);
    if (f1 < GF_Q4) q_put(1, f1, pf1);
    pf0 = *q_src(2, f0);
    if (f1 < GF_Q4) pf1 = *q_src(2, f1);
    float4 bq_next = b_quad(1);  // b of plane i + 1 (red phase of step i), then of plane i (black phase of step i + 1)
    float br_next = b_ring(1);
    float4 rq_next = r_quad(1);
    float rr_next = r_ring(1);
    __syncthreads();
    red_phase(0, bq_0, br_0, rq_0, rr_0);
    float4 bq_cur = bq_0, rq_cur = rq_0;  // b (and rhs) of plane i for the black phase

#pragma unroll 1
    for (int i = 0; i < GF_LX; ++i) {
        // a. the prefetched plane i + 2 goes to LDS; plane i + 3 and the next right-hand sides start their way
        q_put(i + 2, f0, pf0);
        if (f1 < GF_Q4) q_put(i + 2, f1, pf1);
        if (i + 1 < GF_LX) {
            pf0 = *q_src(i + 3, f0);
            if (f1 < GF_Q4) pf1 = *q_src(i + 3, f1);
        }
        const float4 bq_r = bq_next, rq_r = rq_next;
        const float br_r = br_next, rr_r = rr_next;
        if (i + 1 < GF_LX) bq_next = b_quad(i + 2), br_next = b_ring(i + 2), rq_next = r_quad(i + 2), rr_next = r_ring(i + 2);
        __syncthreads();
        // c. red cells of plane i + 1
        red_phase(i + 1, bq_r, br_r, rq_r, rr_r);
        __syncthreads();
        // e. black cells of plane i (own quad), then the whole quad of plane i leaves
        {
            const int e0 = ((i0 + i) + jq + 1) & 1;  // reds at e0, e0 + 2; blacks at 1 - e0, 3 - e0
            const float bb[4] = {bq_cur.x, bq_cur.y, bq_cur.z, bq_cur.w};
            const float rb[4] = {rq_cur.x, rq_cur.y, rq_cur.z, rq_cur.w};
            const float *v0 = slot(i);
            float q[4];
            q[e0] = v0[lrow + lcol + e0], q[e0 + 2] = v0[lrow + lcol + e0 + 2];
            q[1 - e0] = update(i, lrow, lcol + 1 - e0, jq, 1 - e0, bb[1 - e0], rb[1 - e0]);
            q[3 - e0] = update(i, lrow, lcol + 3 - e0, jq, 1 - e0, bb[3 - e0], rb[3 - e0]);
            *reinterpret_cast<float4 *>(xout + cidx((i0 + i) & m, jq, k0 + 4 * kq, log2n)) = make_float4(q[0], q[1], q[2], q[3]);
        }
        bq_cur = bq_r, rq_cur = rq_r;
    }
}

// n_sweeps red-black sweeps starting from `cur`: the single-pass kernel ping-ponging between `cur` and `other` where it
// applies (n >= 256), the two half-sweep launches in place otherwise.  *where = the array that holds the result.
static bool lap_fused_applies(const pm_ctx *c, int n, const float *x, const float *b, const float *tmp) {
    return c->mg_fused_sweep && tmp && n >= 256 && pm_aligned16(x) && pm_aligned16(b) && pm_aligned16(tmp);
}

static int lap_smooth_pp(pm_ctx *c, float *cur, float *other, const float *b, int n, int n_sweeps, float f_relax,
                         float **where) {
    int rc;
    *where = cur;
    if (!lap_fused_applies(c, n, cur, b, other)) {
        for (int s = 0; s < n_sweeps; ++s)
            if ((rc = pm_lap_gauss_seidel(c, cur, b, n, f_relax))) return rc;
        return PM_OK;
    }
    const LapCell cell{(float)(1.0 / ((double)n * n)), (float)(1.0 / 6), f_relax};
    const int lx = n >= 512 ? 64 : 32;
    const unsigned blocks = (unsigned)((n / GF_TJ) * (n / GF_TK) * (n / lx));
    float *src = cur, *dst = other;
    for (int s = 0; s < n_sweeps; ++s) {
        if (lx == 64)
            hipLaunchKernelGGL((gs_fused_kernel<64, LapCell>), dim3(blocks), dim3(BLK), 0, c->stream, src, b, (const float *)nullptr,
                               dst, ilog2(n), cell);
        else
            hipLaunchKernelGGL((gs_fused_kernel<32, LapCell>), dim3(blocks), dim3(BLK), 0, c->stream, src, b, (const float *)nullptr,
                               dst, ilog2(n), cell);
        PM_LAUNCH_CHECK();
        float *t = src;
        src = dst, dst = t;
    }
    *where = src;
    return PM_OK;
}

// ... with the result in x (an odd count of single-pass sweeps ends with a copy back)
static int lap_smooth(pm_ctx *c, float *x, const float *b, int n, int n_sweeps, float f_relax, float *tmp) {
    float *where = x;
    int rc = lap_smooth_pp(c, x, tmp, b, n, n_sweeps, f_relax, &where);
    if (rc) return rc;
    if (where != x) PM_HIP(hipMemcpyAsync(x, where, (size_t)n * n * n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    return PM_OK;
}

extern "C" int pm_set_fused_sweep(pm_ctx *c, int on) {
    PM_REQUIRE(c != nullptr, "ctx is NULL");
    c->mg_fused_sweep = on ? 1 : 0;
    return PM_OK;
}

// laplacian.smoothing (laplacian.py:1026) on device arrays: n_smoothing sweeps with omega = 1.25, in place
extern "C" int pm_lap_smoothing(pm_ctx *c, float *d_x, const float *d_b, int n, int n_smoothing) {
    PM_REQUIRE(c && d_x && d_b && n_smoothing >= 0, "bad argument");
    PM_CHECK_GRID(n);
    float *tmp = nullptr;
    if (n >= 256) {
        int rc = mg_arena(c, ((size_t)n * n * n * 9 / 7 + 1024) * sizeof(float), &tmp);
        if (rc) return rc;
    }
    return lap_smooth(c, d_x, d_b, n, n_smoothing, 1.25f, tmp);
}

// ------------------------------------------------------------------------------------------------
// One whole linear V-cycle (multigrid.py:474-517) issued from C: the recursion over levels with its scratch grids
// taken from one arena of the context, every level operation the kernel the stand-alone entry points launch (same
// bits).  Issued one by one from Python the ~20 calls per level cost more than the kernels below 128^3.

// ---- every level from 16^3 down in ONE workgroup ------------------------------------------------------------
// Below 32^3 a level is nine dependent launches of 2.5-3 us each whatever issues them (stream or hipGraph): 22 launches
// for the levels 16^3, 8^3 and the 4^3 smoothing.  Here one workgroup keeps the solution and right-hand side of those
// levels in LDS (37 KB) and runs the same sequence -- the loop bodies of the scalar level kernels over its own threads,
// a barrier where a launch boundary was -- so every cell goes through the same expression in the same order of
// half-sweeps: the same bits as the launches (test_fused_coarse_levels_are_the_launches).
constexpr int CV_BLK = 512, CV_CELLS = 4096 + 512 + 64;

__global__ __launch_bounds__(CV_BLK) void lap_vcycle_coarse_kernel(float *__restrict__ gx, const float *__restrict__ gb,
                                                                   int log2n0, int npre, int npost, float f_relax) {
    __shared__ float sx[CV_CELLS], sb[CV_CELLS];
    const int tid = threadIdx.x;
    const int n0 = 1 << log2n0, cells0 = n0 * n0 * n0;
    const int off[3] = {0, cells0, cells0 + (cells0 >> 3)};
    for (int c = tid; c < cells0; c += CV_BLK) sx[c] = gx[c], sb[c] = gb[c];
    __syncthreads();
    const float invsix = (float)(1.0 / 6);
    auto smooth = [&](float *x, const float *b, int log2n, int sweeps) {  // pm_lap_gauss_seidel, `sweeps` times
        const int n = 1 << log2n;
        const float h2 = (float)(1.0 / ((double)n * n));
        for (int s = 0; s < sweeps; ++s) {
            lap_gauss_seidel_body<1>(x, b, log2n, h2, invsix, f_relax, tid, CV_BLK);
            __syncthreads();
            lap_gauss_seidel_body<0>(x, b, log2n, h2, invsix, f_relax, tid, CV_BLK);
            __syncthreads();
        }
    };
    const int D = log2n0 - 2;  // levels of 8^3 cells and more (16^3: two, 8^3: one); the one below is only smoothed
    for (int l = 0; l < D; ++l) {
        const int log2n = log2n0 - l, n = 1 << log2n, nc = n >> 1;
        float *x = sx + off[l], *xc = sx + off[l + 1];
        const float *b = sb + off[l];
        float *bc = sb + off[l + 1];
        smooth(x, b, log2n, npre);
        lap_restrict_residual_body(x, b, log2n, (float)((double)n * n), bc, XRange{0, n, n - 1}, XRange{0, nc, nc - 1}, tid,
                                   CV_BLK);
        __syncthreads();
        // laplacian.initialise_potential (laplacian.py:765) as lap_vcycle_level does it: pm_linear_operator(f1, 0)
        const float h = 1.0f / (float)nc;
        const float minus_h2_over_six = (float)(-(double)(h * h) / 6.0);
        for (int cc = tid; cc < nc * nc * nc; cc += CV_BLK) xc[cc] = minus_h2_over_six * bc[cc] + 0.0f;
        __syncthreads();
    }
    smooth(sx + off[D], sb + off[D], log2n0 - D, npre);
    for (int l = D - 1; l >= 0; --l) {
        const int log2n = log2n0 - l, n = 1 << log2n, nc = n >> 1;
        prolongation_body(sx + off[l], sx + off[l + 1], log2n - 1, 1, XRange{0, n, n - 1}, XRange{0, nc, nc - 1}, tid, CV_BLK);
        __syncthreads();
        smooth(sx + off[l], sb + off[l], log2n, npost);
    }
    for (int c = tid; c < cells0; c += CV_BLK) gx[c] = sx[c];
}

// on (default): pm_lap_vcycle runs the levels from 16^3 down in one launch; off: level by level
extern "C" int pm_set_fused_coarse(pm_ctx *c, int on) {
    PM_REQUIRE(c, "NULL argument");
    c->mg_fused_coarse = on ? 1 : 0;
    return PM_OK;
}

static int lap_vcycle_level(pm_ctx *c, float *x, const float *b, int n, int npre, int npost, float *scratch) {
    const float f_relax = 1.25f;  // laplacian.py:1051
    int rc;
    if (c->mg_fused_coarse && n <= 16) {
        hipLaunchKernelGGL(lap_vcycle_coarse_kernel, dim3(1), dim3(CV_BLK), 0, c->stream, x, b, ilog2(n), npre, npost, f_relax);
        PM_LAUNCH_CHECK();
        return PM_OK;
    }
    // (the level's ping-pong grid for the single-pass sweep sits first in its share of the arena)
    float *tmp = n >= 256 ? scratch : nullptr;
    if (tmp) scratch += (int64_t)n * n * n;
    float *cur = x;  // the array that holds the level's solution at the moment
    if ((rc = lap_smooth_pp(c, x, tmp, b, n, npre, f_relax, &cur))) return rc;
    const int nc = n / 2;
    const int64_t cells = (int64_t)nc * nc * nc;
    float *res_c = scratch, *x_corr_c = scratch + cells;
    if ((rc = pm_lap_restrict_residual(c, cur, b, n, res_c))) return rc;
    // laplacian.initialise_potential (laplacian.py:765): -h^2/6 * b with h = 1/nc in fp32, the quotient in fp64
    const float h = 1.0f / (float)nc;
    const float minus_h2_over_six = (float)(-(double)(h * h) / 6.0);
    if ((rc = pm_linear_operator(c, x_corr_c, res_c, minus_h2_over_six, 0.0f, cells))) return rc;
    if (n <= 8) {  // nlevel >= ncoarse - 3 (multigrid.py:510): the coarsest correction is only smoothed
        for (int s = 0; s < npre; ++s)
            if ((rc = pm_lap_gauss_seidel(c, x_corr_c, res_c, nc, f_relax))) return rc;
    } else if ((rc = lap_vcycle_level(c, x_corr_c, res_c, nc, npre, npost, scratch + 2 * cells))) {
        return rc;
    }
    // the corrected solution goes where an odd / even number of ping-pong sweeps will leave the result in x
    float *target = x;
    if (lap_fused_applies(c, n, x, b, tmp) && (npost & 1)) target = tmp;
    if ((rc = prolongation_launch(c, target, x_corr_c, nc, 1, xr_cube(n), xr_cube(nc), cur))) return rc;
    float *where = target;
    if ((rc = lap_smooth_pp(c, target, target == x ? tmp : x, b, n, npost, f_relax, &where))) return rc;
    if (where != x) PM_HIP(hipMemcpyAsync(x, where, (size_t)n * n * n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    return PM_OK;
}

// Runs `body` (a whole V-cycle: nothing but kernel launches, copies and memsets on c->stream) -- as a replay of its
// captured hipGraph when the very same call was seen before (same arrays, sizes, sweeps, arena), else capturing it
// on the way.  Any failure of the capture machinery switches it off for the context and the launches go out plainly.
template <class F>
static int vcycle_graphed(pm_ctx *c, int kind, const void *x, const void *b, float q, int n, int npre, int npost, F body) {
    if (!c->vc_graph_mode) return body();
    const void *arena = c->mg_buf;
    for (auto &g : c->vc_graphs)
        if (g.exec && g.x == x && g.b == b && g.arena == arena && g.kind == kind && g.n == n && g.npre == npre &&
            g.npost == npost && g.q == q) {
            g.last_use = ++c->vc_clock, ++c->vc_hits;
            if (hipGraphLaunch(g.exec, c->stream) == hipSuccess) return PM_OK;
            (void)hipGetLastError();
            c->vc_graph_mode = 0;
            return body();
        }
    ++c->vc_misses;
    if (c->vc_misses > 32 && c->vc_misses > 2 * c->vc_hits) {  // the caller's arrays keep changing: capturing costs more than it saves
        c->vc_graph_mode = 0;
        return body();
    }
    auto give_up = [&]() {
        (void)hipGetLastError();
        c->vc_graph_mode = 0;
        return body();
    };
    if (!c->capture_stream && hipStreamCreateWithFlags(&c->capture_stream, hipStreamNonBlocking) != hipSuccess) return give_up();
    hipStream_t user = c->stream;
    if (hipStreamBeginCapture(c->capture_stream, hipStreamCaptureModeRelaxed) != hipSuccess) return give_up();
    c->stream = c->capture_stream;
    const int rc = body();
    c->stream = user;
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(c->capture_stream, &graph);
    if (rc) {
        if (graph) hipGraphDestroy(graph);
        return rc;
    }
    if (e != hipSuccess || !graph) return give_up();
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (ei != hipSuccess || !exec) return give_up();
    pm_ctx::VcGraph *slot = &c->vc_graphs[0];
    for (auto &g : c->vc_graphs)
        if (!g.exec || (slot->exec && g.last_use < slot->last_use)) slot = &g;
    if (slot->exec) hipGraphExecDestroy(slot->exec);
    *slot = pm_ctx::VcGraph{x, b, arena, kind, n, npre, npost, q, exec, ++c->vc_clock};
    if (hipGraphLaunch(exec, c->stream) == hipSuccess) return PM_OK;
    slot->exec = nullptr;
    hipGraphExecDestroy(exec);
    return give_up();
}

// mode 1 (default): V-cycles are captured and replayed as hipGraphs; 0: plain launches.  Either way the same kernels
// with the same arguments in the same order.
extern "C" int pm_set_vcycle_graphs(pm_ctx *c, int mode) {
    PM_REQUIRE(c && (mode == 0 || mode == 1), "bad argument");
    c->vc_graph_mode = mode;
    return PM_OK;
}

extern "C" int pm_vcycle_graph_stats(pm_ctx *c, unsigned *replayed, unsigned *captured) {
    PM_REQUIRE(c && replayed && captured, "NULL argument");
    *replayed = c->vc_hits, *captured = c->vc_misses;
    return PM_OK;
}

extern "C" int pm_lap_vcycle(pm_ctx *c, float *d_x, const float *d_b, int n, int npre, int npost) {
    PM_REQUIRE(c && d_x && d_b && npre >= 0 && npost >= 0, "bad argument");
    PM_CHECK_GRID(n);
    PM_REQUIRE(n >= 8, "pm_lap_vcycle needs n >= 8 (ncoarse >= 3)");
    // two grids per coarse level: 2 (n/2)^3 (1 + 1/8 + ...) < 2 n^3 / 7 floats, and a ping-pong grid per level of 256^3
    // and more: n^3 (1 + 1/8 + ...) < 8 n^3 / 7
    float *scratch = nullptr;
    int rc = mg_arena(c, ((size_t)n * n * n * (n >= 256 ? 11 : 2) / 7 + 1024) * sizeof(float), &scratch);
    if (rc) return rc;
    return vcycle_graphed(c, (c->mg_fused_sweep ? 1 : 0) | (c->mg_fused_coarse ? 2 : 0), d_x, d_b, 0.f, n, npre, npost,
                          [&]() { return lap_vcycle_level(c, d_x, d_b, n, npre, npost, scratch); });
}

// One colour half-sweep on a slab [nx + 2, n, n] (owned planes 1..nx, ghosts filled by the caller, who
// exchanges them again between the red (colour 1) and the black (colour 0) half).  x_lo = global index of
// the first owned plane.  Same arithmetic per cell as the single-GPU sweep.
extern "C" int pm_lap_gauss_seidel_slab(pm_ctx *c, float *d_x, const float *d_b, int n, int nx, int x_lo, int colour,
                                        float f_relax) {
    PM_REQUIRE(c && d_x && d_b && nx >= 1 && (colour == 0 || colour == 1), "bad argument");
    PM_CHECK_GRID(n);
    PM_REQUIRE(n >= 16 && pm_aligned16(d_x) && pm_aligned16(d_b), "slab smoother needs n >= 16 and 16-byte aligned arrays");
    const float h2 = (float)(1.0 / ((double)n * n)), invsix = (float)(1.0 / 6);
    const int64_t threads = (int64_t)nx * n * (n / 8);
    const int blocks = pm_stream_blocks(threads, BLK);
    const int gpar = (x_lo - 1) & 1;  // global = local - 1 + x_lo
    if (colour == 1)
        hipLaunchKernelGGL((lap_gauss_seidel_rows_kernel<1>), dim3(blocks), dim3(BLK), 0, c->stream, d_x, d_b, ilog2(n), h2,
                           invsix, f_relax, xr_slab(nx), gpar);
    else
        hipLaunchKernelGGL((lap_gauss_seidel_rows_kernel<0>), dim3(blocks), dim3(BLK), 0, c->stream, d_x, d_b, ilog2(n), h2,
                           invsix, f_relax, xr_slab(nx), gpar);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

// ------------------------------------------------------------------------------------------------
// f(R) n = 1 (cubic.py) and n = 2 (quartic.py): same stencil with u^2 / u^3 neighbour powers and the
// depressed cubic / quartic root; kernels are templated on FRN = 1, 2.

// quartic.py:157-210 solution_quartic_equation: fp64, same branch structure.  Out of line: quartic_root below calls it
// for the cells its fast path declines.
__device__ __attribute__((noinline)) double quartic_root_closed_form(float pf, float qf) {
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

// The same root for ~80 instructions (cubic_root's recipe, further down): the reference's formula in its regular
// branch (p != 0, the square-root term >= 0, Q + d0 / Q > 0) evaluated in fp32 with the hardware's 1-ulp sqrt / rcp /
// log / exp, one fp32 Newton step of u^4 + p u + q (the formula's last line subtracts terms of like size), one more
// whose residual is fp64 and whose slope and quotient are fp32; the last step is the check (more than 3e-5 of u, a NaN
// anywhere: the closed form in fp64, as every other branch).  numpy emulation over 10^7 (p, q): the fp64 formula's float
// but for rounding-boundary cases, no NaN of the reference turned into a number.
__device__ __forceinline__ float quartic_root(float pf, float qf) {
    if (pf != 0.0f) {
        const float d0 = 12.0f * qf, d1 = 27.0f * (pf * pf);
        const float r = d0 * __builtin_amdgcn_rcpf(d1);
        const float st = 1.0f - 4.0f * d0 * (r * r);
        if (st >= 0.0f) {
            const float h = 0.5f * d1 * (1.0f + __builtin_amdgcn_sqrtf(st));
            const float Q = __builtin_amdgcn_exp2f(0.33333334f * __builtin_amdgcn_logf(h));
            const float QQ = Q + d0 * __builtin_amdgcn_rcpf(Q);
            if (QQ > 0.0f) {
                const float S = 0.5f * __builtin_amdgcn_sqrtf(QQ * 0.33333334f);
                const float pS = pf * __builtin_amdgcn_rcpf(S);
                const float x0 = pf > 0.0f ? -S + 0.5f * __builtin_amdgcn_sqrtf(-4.0f * (S * S) + pS)
                                           : S + 0.5f * __builtin_amdgcn_sqrtf(-4.0f * (S * S) - pS);
                const float x02 = x0 * x0;
                const float f0 = __builtin_fmaf(x02, x02, __builtin_fmaf(pf, x0, qf));
                const float x1 = x0 - f0 * __builtin_amdgcn_rcpf(__builtin_fmaf(4.0f * x02, x0, pf));
                const double x = (double)x1, x2 = x * x;
                const double f = fma(x2, x2, fma((double)pf, x, (double)qf));
                const float x12 = x1 * x1;
                const float delta = (float)f * __builtin_amdgcn_rcpf(__builtin_fmaf(4.0f * x12, x1, pf));
                if (fabsf(delta) <= 3.0e-5f * fabsf(x1)) return x1 - delta;
            }
        }
    }
    return (float)quartic_root_closed_form(pf, qf);
}

template <int FRN>
__device__ __forceinline__ float fr_pw(float a) {
    return FRN == 1 ? a * a : a * a * a;
}
template <int FRN>
__device__ __forceinline__ float fr_lead(float x) {
    return FRN == 1 ? x * x * x : (x * x) * (x * x);
}

// cubic.py:162-207 solution_cubic_equation, the reference's formulas and branch structure in fp64 (NaNs included).
// Out of line: cubic_root below calls it for the few cells its fast paths decline.
__device__ __attribute__((noinline)) double cubic_root_closed_form(float pf, float d1f) {
    const double inv3 = 1.0 / 3;
    const double d1 = (double)d1f, p = (double)pf;
    double d = d1 * d1 + 108.0 * (p * p * p);
    if (d > 0.0) {
        d = d1 + sqrt(d);
        if (d == 0.0) return -inv3 * pow(d1, inv3);
        // x^(1/3) of a positive x is cbrt (a third of pow's cost); a negative x keeps pow and its NaN, as the reference
        const double hd = 0.5 * d;
        const double C = hd > 0.0 ? cbrt(hd) : pow(hd, inv3);
        return -inv3 * (C - 3.0 * p / C);
    } else if (d < 0.0) {
        const double d0 = -3.0 * p;  // > 0 here (d < 0 needs p < 0)
        const double sd0 = sqrt(d0);
        d = d1 / (2.0 * (d0 * sd0));  // d0^1.5 without pow
        if (fabs(d) < 1.0) {
            const double theta = acos(d);
            return (double)(float)(-2.0 * inv3 * sd0 * cos(inv3 * (theta + 2.0 * 3.14159265358979323846)));
        }
        return -inv3 * pow(d1, inv3);
    }
    return -inv3 * pow(d1, inv3);
}

// The root the reference picks (cubic.py:162-207), rounded to fp32, for ~70 instructions instead of the ~200 of the
// closed form with libm's sqrt / cbrt / acos / cos (measured: the 512^3 sweep was bound by them, not by HBM):
//   * a start good to ~1e-6 from hardware-rate approximations (v_sqrt, v_rcp, v_log, v_exp: 1 ulp each) --
//     one real root (d > 0) where the reference's cube C^3 = (d1 + sqrt(d)) / 2 is positive (d1 >= 0 or p > 0; the
//     scalaron's regime has d1 = 27 h^2 q - 27 rhs of order 10^2): the reference's own formula -(C - 3p / C) / 3 (for
//     d1 < 0 with C^3 = 54 p^3 / (sqrt(d) - d1), which does not cancel), then one fp32 Newton step (the formula cancels
//     where |d1| << p^1.5);
//     three real roots (p < 0, d1^2 < 0.766 * 108 |p|^3, i.e. |cos 3 phi| < 7/8: away from the double root): the
//     reference takes the largest one, t = s y with s = sqrt(-p / 3) and y in [1.28, 2] the largest root of
//     y^3 - 3 y + c = 0, c = d1 / (27 s^3); y is analytic in w = sqrt(2 - c) and a degree-5 polynomial in w is good to
//     2e-7 on the whole range;
//   * one Newton step of t^3 + p t + d1 / 27 whose residual is evaluated in fp64 (it is a difference of terms 10^6 times
//     its size) and whose slope and quotient are fp32 (the correction is ~1e-6 of t, so its own 1e-7 relative error is
//     1e-13 of the root), the sum rounded once: the fp64 closed form's float, but for rounding-boundary cases (numpy
//     emulation over 10^7 (p, d1): no mismatch in either regime);
//   * the step doubles as the check: where it is more than 3e-5 of t (bad start, overflow, NaN) -- and in every branch
//     not listed above, the reference's NaN ones included -- the closed form in fp64 (cubic_root_closed_form).
__device__ __forceinline__ float cubic_root(float pf, float d1f) {
    const double p = (double)pf, d1 = (double)d1f;
    const double d = d1 * d1 + 108.0 * (p * p * p);
    float ts = 0.f;
    bool have = false;
    if (d > 0.0) {
        if (d1f >= 0.0f || pf > 0.0f) {
            const float sf = __builtin_amdgcn_sqrtf((float)d);
            const float hf = d1f >= 0.0f ? 0.5f * (d1f + sf) : 54.0f * (pf * pf * pf) * __builtin_amdgcn_rcpf(sf - d1f);
            if (hf > 1.0e-30f && hf < 1.0e30f) {
                const float Cf = __builtin_amdgcn_exp2f(0.33333334f * __builtin_amdgcn_logf(hf));
                const float t0 = -0.33333334f * (Cf - 3.0f * pf * __builtin_amdgcn_rcpf(Cf));
                const float f0 = __builtin_fmaf(__builtin_fmaf(t0, t0, pf), t0, d1f * 0.037037037f);
                ts = t0 - f0 * __builtin_amdgcn_rcpf(__builtin_fmaf(3.0f * t0, t0, pf));
                have = true;
            }
        }
    } else if (pf < 0.0f && d1f * d1f < 0.766f * (-108.0f * (pf * pf * pf))) {
        const float s2 = -0.33333334f * pf, s = __builtin_amdgcn_sqrtf(s2);
        const float c = (d1f * 0.037037037f) * __builtin_amdgcn_rcpf(s2 * s);
        const float w = __builtin_amdgcn_sqrtf(2.0f - c);
        float y = 0.00017914111958816648f;
        y = __builtin_fmaf(y, w, -0.0019134014146402478f);
        y = __builtin_fmaf(y, w, 0.010815435089170933f);
        y = __builtin_fmaf(y, w, -0.05377647280693054f);
        y = __builtin_fmaf(y, w, 0.5766810774803162f);
        y = __builtin_fmaf(y, w, 1.0001031160354614f);
        ts = s * y;
        have = true;
    }
    if (have) {
        const double t = (double)ts;
        const double f = fma(fma(t, t, p), t, d1 * (1.0 / 27));
        const float delta = (float)f * __builtin_amdgcn_rcpf(__builtin_fmaf(3.0f * ts, ts, pf));
        if (fabsf(delta) <= 3.0e-5f * fabsf(ts)) return ts - delta;
    }
    return (float)cubic_root_closed_form(pf, d1f);
}

template <int FRN>
__device__ __forceinline__ float cubic_p(const float *__restrict__ x, const float *__restrict__ b, int i, int j, int k,
                                         int m, int log2n, float h2, float invsix, int xmask) {
    const float a0 = x[cidx((i - 1) & xmask, j, k, log2n)], a1 = x[cidx(i, (j - 1) & m, k, log2n)],
                a2 = x[cidx(i, j, (k - 1) & m, log2n)], a3 = x[cidx(i, j, (k + 1) & m, log2n)],
                a4 = x[cidx(i, (j + 1) & m, k, log2n)], a5 = x[cidx((i + 1) & xmask, j, k, log2n)];
    return h2 * b[cidx(i, j, k, log2n)] - invsix * (fr_pw<FRN>(a0) + fr_pw<FRN>(a1) + fr_pw<FRN>(a2) + fr_pw<FRN>(a3) +
                                                    fr_pw<FRN>(a4) + fr_pw<FRN>(a5));
}

// cubic.py:23-86 operator / :90-158 residual_with_rhs (mode 1) / :844-900 residual_error (mode 2)
template <int MODE, int FRN>
__global__ __launch_bounds__(BLK) void cubic_operator_kernel(const float *__restrict__ x, const float *__restrict__ b,
                                                             float qh2, const float *__restrict__ rhs, int log2n,
                                                             float h2, float invsix, float *__restrict__ out,
                                                             double *__restrict__ result, XRange xr) {
    const int n = 1 << log2n, m = n - 1;
    const int64_t total = (int64_t)xr.ni << (2 * log2n);
    double acc = 0;
    for (int64_t t0 = (int64_t)blockIdx.x * BLK + threadIdx.x; t0 < total; t0 += (int64_t)gridDim.x * BLK) {
        const int k = (int)(t0 & m), j = (int)((t0 >> log2n) & m), i = xr.i0 + (int)(t0 >> (2 * log2n));
        const int64_t t = cidx(i, j, k, log2n);
        const float p = cubic_p<FRN>(x, b, i, j, k, m, log2n, h2, invsix, xr.xmask);
        const float xt = x[t];
        if (MODE == 0) out[t] = fr_lead<FRN>(xt) + p * xt + qh2;
        if (MODE == 1) out[t] = -(fr_lead<FRN>(xt)) - p * xt - qh2 + rhs[t];
        if (MODE == 2) {
            const float r = fr_lead<FRN>(xt) + p * xt + qh2;
            acc += (double)(r * r);
        }
    }
    if (MODE == 2) block_sum_to(acc, result);
}

// The same from 512^3, marching (lap_residual_error_march_kernel's scheme: 16 x 128 patch, RM_LX planes, x loaded once per
// plane, j +- 1 and k +- 1 through a double-buffered LDS plane): the cell-per-thread kernel above asks the L2 for seven
// scalars of x per cell.  Same expression per cell, in the same order.
template <int MODE, int FRN>
__global__ __launch_bounds__(BLK) void cubic_operator_march_kernel(const float *__restrict__ x, const float *__restrict__ b,
                                                                   float qh2, const float *__restrict__ rhs, int log2n,
                                                                   float h2, float invsix, float *__restrict__ out,
                                                                   double *__restrict__ result, XRange xr) {
    __shared__ __attribute__((aligned(16))) float sh[2][RM_TJ + 2][RM_TK + 8];
    const int n = 1 << log2n, m = n - 1, tid = threadIdx.x;
    const int ncol_k = n / RM_TK, ncol_j = n / RM_TJ;
    const int col = blockIdx.x % (ncol_k * ncol_j), seg = blockIdx.x / (ncol_k * ncol_j);
    const int kb = (col % ncol_k) * RM_TK, j0 = (col / ncol_k) * RM_TJ;
    const int jj = tid >> 4, kq = tid & 15, j = j0 + jj, k0 = kb + 8 * kq;
    const int ia = xr.i0 + seg * RM_LX, ie = min(ia + RM_LX, xr.i0 + xr.ni);
    auto row = [&](const float *g, int a, int bb) -> const float * {
        return g + ((((int64_t)(a & xr.xmask) << log2n) + (bb & m)) << log2n);
    };
    auto ld8 = [&](const float *p, float *d) {
        const float4 q0 = *reinterpret_cast<const float4 *>(p), q1 = *reinterpret_cast<const float4 *>(p + 4);
        d[0] = q0.x, d[1] = q0.y, d[2] = q0.z, d[3] = q0.w, d[4] = q1.x, d[5] = q1.y, d[6] = q1.z, d[7] = q1.w;
    };
    const bool edge_j = jj == 0 || jj == RM_TJ - 1, edge_k = kq == 0 || kq == 15;
    const int jrow = jj == 0 ? j0 - 1 : j0 + RM_TJ, lrow = jj == 0 ? 0 : RM_TJ + 1;
    const int kcell = kq == 0 ? kb - 1 : kb + RM_TK, lcell = kq == 0 ? 3 : 4 + RM_TK;
    float prev[8], cur[8], next[8], erow[8], bb8[8], rr8[8], ecell = 0.f;
    double acc = 0;
    if (ia < ie) {
        ld8(row(x, ia - 1, j) + k0, prev);
        ld8(row(x, ia, j) + k0, cur);
    }
    for (int i = ia; i < ie; ++i) {
        float *pl = &sh[i & 1][0][0];
        constexpr int RS = RM_TK + 8;
        ld8(row(x, i + 1, j) + k0, next);
        ld8(row(b, i, j) + k0, bb8);
        if (MODE == 1) ld8(rhs + cidx(i, j, k0, log2n), rr8);
        if (edge_j) ld8(row(x, i, jrow) + k0, erow);
        if (edge_k) ecell = row(x, i, j)[kcell & m];
        *reinterpret_cast<float4 *>(pl + (jj + 1) * RS + 4 + 8 * kq) = make_float4(cur[0], cur[1], cur[2], cur[3]);
        *reinterpret_cast<float4 *>(pl + (jj + 1) * RS + 8 + 8 * kq) = make_float4(cur[4], cur[5], cur[6], cur[7]);
        if (edge_j) {
            *reinterpret_cast<float4 *>(pl + lrow * RS + 4 + 8 * kq) = make_float4(erow[0], erow[1], erow[2], erow[3]);
            *reinterpret_cast<float4 *>(pl + lrow * RS + 8 + 8 * kq) = make_float4(erow[4], erow[5], erow[6], erow[7]);
        }
        if (edge_k) pl[(jj + 1) * RS + lcell] = ecell;
        __syncthreads();
        const float *me = pl + (jj + 1) * RS + 4 + 8 * kq, *ym = me - RS, *yp = me + RS;
        const float4 a0 = *reinterpret_cast<const float4 *>(ym), a1 = *reinterpret_cast<const float4 *>(ym + 4);
        const float4 c0 = *reinterpret_cast<const float4 *>(yp), c1 = *reinterpret_cast<const float4 *>(yp + 4);
        const float aym[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const float ayp[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        float v[10], res[8];
        v[0] = me[-1], v[9] = me[8];
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) v[c2 + 1] = cur[c2];
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) {
            const float xt = v[c2 + 1];
            const float p = h2 * bb8[c2] - invsix * (fr_pw<FRN>(prev[c2]) + fr_pw<FRN>(aym[c2]) + fr_pw<FRN>(v[c2]) +
                                                     fr_pw<FRN>(v[c2 + 2]) + fr_pw<FRN>(ayp[c2]) + fr_pw<FRN>(next[c2]));
            if (MODE == 0) res[c2] = fr_lead<FRN>(xt) + p * xt + qh2;
            if (MODE == 1) res[c2] = -(fr_lead<FRN>(xt)) - p * xt - qh2 + rr8[c2];
            if (MODE == 2) {
                const float r = fr_lead<FRN>(xt) + p * xt + qh2;
                acc += (double)(r * r);
            }
        }
        if (MODE != 2) {
            float4 *o4 = reinterpret_cast<float4 *>(out + cidx(i, j, k0, log2n));
            o4[0] = make_float4(res[0], res[1], res[2], res[3]);
            o4[1] = make_float4(res[4], res[5], res[6], res[7]);
        }
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) prev[c2] = cur[c2], cur[c2] = next[c2];
    }
    if (MODE == 2) block_sum_to(acc, result);
}

// The FAS V-cycle's way down in one pass over the fine grids (n >= 512, cube): the marching operator above, whose values
// never reach memory -- the restriction (mesh.py:14-110: scale * the eight fine cells summed in the order (i, j), (i, j + 1),
// (i + 1, j), (i + 1, j + 1), k then k + 1 within each) is taken on the fly, of the operator / residual (scale -1/8
// without a right-hand side, +1/8 with one), of x and of b alike.  A thread holds eight k-cells of one row of one
// plane: the k pairs are its own, the row j + 1 belongs to the lane 16 further on (one shuffle per value), the plane
// i + 1 is the next trip of its own march.  Same operator expression, same order of the sums: the bits of
// fr_operator + three pm_restriction launches, which write the operator's n^3 floats and read x, b and them again
// (0.73 -> ms at 512^3).
template <bool WITH_RHS, int FRN>
__global__ __launch_bounds__(BLK) void fr_restrict_march_kernel(const float *__restrict__ x, const float *__restrict__ b,
                                                                float qh2, const float *__restrict__ rhs, int log2n, float h2,
                                                                float invsix, float res_scale, float *__restrict__ res_c,
                                                                float *__restrict__ x_c, float *__restrict__ b_c) {
    __shared__ __attribute__((aligned(16))) float sh[2][RM_TJ + 2][RM_TK + 8];
    static_assert(RM_LX % 2 == 0 && RM_TJ % 2 == 0, "plane and row pairs stay inside a workgroup's segment");
    const int n = 1 << log2n, m = n - 1, tid = threadIdx.x, log2nc = log2n - 1;
    const int ncol_k = n / RM_TK, ncol_j = n / RM_TJ;
    const int col = blockIdx.x % (ncol_k * ncol_j), seg = blockIdx.x / (ncol_k * ncol_j);
    const int kb = (col % ncol_k) * RM_TK, j0 = (col / ncol_k) * RM_TJ;
    const int jj = tid >> 4, kq = tid & 15, j = j0 + jj, k0 = kb + 8 * kq;
    const int ia = seg * RM_LX, ie = min(ia + RM_LX, n);
    auto row = [&](const float *g, int a, int bb) -> const float * {
        return g + ((((int64_t)(a & m) << log2n) + (bb & m)) << log2n);
    };
    auto ld8 = [&](const float *p, float *d) {
        const float4 q0 = *reinterpret_cast<const float4 *>(p), q1 = *reinterpret_cast<const float4 *>(p + 4);
        d[0] = q0.x, d[1] = q0.y, d[2] = q0.z, d[3] = q0.w, d[4] = q1.x, d[5] = q1.y, d[6] = q1.z, d[7] = q1.w;
    };
    const bool edge_j = jj == 0 || jj == RM_TJ - 1, edge_k = kq == 0 || kq == 15;
    const int jrow = jj == 0 ? j0 - 1 : j0 + RM_TJ, lrow = jj == 0 ? 0 : RM_TJ + 1;
    const int kcell = kq == 0 ? kb - 1 : kb + RM_TK, lcell = kq == 0 ? 3 : 4 + RM_TK;
    float prev[8], cur[8], next[8], erow[8], bb8[8], rr8[8], ecell = 0.f;
    float sr[4] = {0.f, 0.f, 0.f, 0.f}, sx[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};  // the coarse cells under way
    if (ia < ie) {
        ld8(row(x, ia - 1, j) + k0, prev);
        ld8(row(x, ia, j) + k0, cur);
    }
    for (int i = ia; i < ie; ++i) {
        float *pl = &sh[i & 1][0][0];
        constexpr int RS = RM_TK + 8;
        ld8(row(x, i + 1, j) + k0, next);
        ld8(row(b, i, j) + k0, bb8);
        if (WITH_RHS) ld8(rhs + cidx(i, j, k0, log2n), rr8);
        if (edge_j) ld8(row(x, i, jrow) + k0, erow);
        if (edge_k) ecell = row(x, i, j)[kcell & m];
        *reinterpret_cast<float4 *>(pl + (jj + 1) * RS + 4 + 8 * kq) = make_float4(cur[0], cur[1], cur[2], cur[3]);
        *reinterpret_cast<float4 *>(pl + (jj + 1) * RS + 8 + 8 * kq) = make_float4(cur[4], cur[5], cur[6], cur[7]);
        if (edge_j) {
            *reinterpret_cast<float4 *>(pl + lrow * RS + 4 + 8 * kq) = make_float4(erow[0], erow[1], erow[2], erow[3]);
            *reinterpret_cast<float4 *>(pl + lrow * RS + 8 + 8 * kq) = make_float4(erow[4], erow[5], erow[6], erow[7]);
        }
        if (edge_k) pl[(jj + 1) * RS + lcell] = ecell;
        __syncthreads();
        const float *me = pl + (jj + 1) * RS + 4 + 8 * kq, *ym = me - RS, *yp = me + RS;
        const float4 a0 = *reinterpret_cast<const float4 *>(ym), a1 = *reinterpret_cast<const float4 *>(ym + 4);
        const float4 c0 = *reinterpret_cast<const float4 *>(yp), c1 = *reinterpret_cast<const float4 *>(yp + 4);
        const float aym[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const float ayp[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        float v[10], res[8];
        v[0] = me[-1], v[9] = me[8];
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) v[c2 + 1] = cur[c2];
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) {
            const float xt = v[c2 + 1];
            const float p = h2 * bb8[c2] - invsix * (fr_pw<FRN>(prev[c2]) + fr_pw<FRN>(aym[c2]) + fr_pw<FRN>(v[c2]) +
                                                     fr_pw<FRN>(v[c2 + 2]) + fr_pw<FRN>(ayp[c2]) + fr_pw<FRN>(next[c2]));
            if (!WITH_RHS) res[c2] = fr_lead<FRN>(xt) + p * xt + qh2;
            else res[c2] = -(fr_lead<FRN>(xt)) - p * xt - qh2 + rr8[c2];
        }
        // the eight sums in the restriction's order: (i even: own row, then row j + 1), (i odd: the same, and out they go)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float r0 = res[2 * q], r1 = res[2 * q + 1], x0 = cur[2 * q], x1 = cur[2 * q + 1], b0 = bb8[2 * q], b1 = bb8[2 * q + 1];
            const float pr0 = __shfl_xor(r0, 16, 64), pr1 = __shfl_xor(r1, 16, 64);
            const float px0 = __shfl_xor(x0, 16, 64), px1 = __shfl_xor(x1, 16, 64);
            const float pb0 = __shfl_xor(b0, 16, 64), pb1 = __shfl_xor(b1, 16, 64);
            if (!(i & 1)) {
                sr[q] = r0 + r1 + pr0 + pr1, sx[q] = x0 + x1 + px0 + px1, sb[q] = b0 + b1 + pb0 + pb1;
            } else {
                sr[q] = sr[q] + r0 + r1 + pr0 + pr1, sx[q] = sx[q] + x0 + x1 + px0 + px1, sb[q] = sb[q] + b0 + b1 + pb0 + pb1;
            }
        }
        if ((i & 1) && !(jj & 1)) {
            const int64_t at = cidx(i >> 1, j >> 1, k0 >> 1, log2nc);
            *reinterpret_cast<float4 *>(res_c + at) = make_float4(res_scale * sr[0], res_scale * sr[1], res_scale * sr[2], res_scale * sr[3]);
            *reinterpret_cast<float4 *>(x_c + at) = make_float4(0.125f * sx[0], 0.125f * sx[1], 0.125f * sx[2], 0.125f * sx[3]);
            *reinterpret_cast<float4 *>(b_c + at) = make_float4(0.125f * sb[0], 0.125f * sb[1], 0.125f * sb[2], 0.125f * sb[3]);
        }
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) prev[c2] = cur[c2], cur[c2] = next[c2];
    }
}

static bool fr_march_applies(const pm_ctx *c, int n, const float *x, const float *b, const float *o1, const float *o2) {
    return n >= 512 && c->fr_rows_sweep && pm_aligned16(x) && pm_aligned16(b) && pm_aligned16(o1) && pm_aligned16(o2);
}
static unsigned fr_march_blocks(int n, XRange xr) {
    return (unsigned)((n / RM_TJ) * (n / RM_TK) * ((xr.ni + RM_LX - 1) / RM_LX));
}

template <int FRN>
static int fr_operator(pm_ctx *c, const float *d_x, const float *d_b, float q, const float *d_rhs, int n, float *d_out,
                       XRange xr) {
    PM_REQUIRE(c && d_x && d_b && d_out, "NULL argument");
    PM_CHECK_GRID(n);
    const int64_t total = (int64_t)xr.ni * n * n;
    const float h2 = (float)(1.0 / ((double)n * n)), invsix = (float)(1.0 / 6);
    const int blocks = pm_stream_blocks(total, BLK);
    if (fr_march_applies(c, n, d_x, d_b, d_rhs, d_out)) {
        if (d_rhs)
            hipLaunchKernelGGL((cubic_operator_march_kernel<1, FRN>), dim3(fr_march_blocks(n, xr)), dim3(BLK), 0, c->stream, d_x,
                               d_b, q * h2, d_rhs, ilog2(n), h2, invsix, d_out, (double *)nullptr, xr);
        else
            hipLaunchKernelGGL((cubic_operator_march_kernel<0, FRN>), dim3(fr_march_blocks(n, xr)), dim3(BLK), 0, c->stream, d_x,
                               d_b, q * h2, d_rhs, ilog2(n), h2, invsix, d_out, (double *)nullptr, xr);
        PM_LAUNCH_CHECK();
        return PM_OK;
    }
    if (d_rhs)
        hipLaunchKernelGGL((cubic_operator_kernel<1, FRN>), dim3(blocks), dim3(BLK), 0, c->stream, d_x, d_b, q * h2,
                           d_rhs, ilog2(n), h2, invsix, d_out, (double *)nullptr, xr);
    else
        hipLaunchKernelGGL((cubic_operator_kernel<0, FRN>), dim3(blocks), dim3(BLK), 0, c->stream, d_x, d_b, q * h2,
                           d_rhs, ilog2(n), h2, invsix, d_out, (double *)nullptr, xr);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

template <int FRN>
static int fr_residual_sumsq(pm_ctx *c, const float *d_x, const float *d_b, float q, int n, XRange xr, double *h_sum) {
    PM_REQUIRE(c && d_x && d_b && h_sum, "NULL argument");
    PM_CHECK_GRID(n);
    PM_HIP(hipMemsetAsync(c->d_hist, 0, sizeof(double), c->stream));
    const int64_t total = (int64_t)xr.ni * n * n;
    const float h2 = (float)(1.0 / ((double)n * n)), invsix = (float)(1.0 / 6);
    if (fr_march_applies(c, n, d_x, d_b, nullptr, nullptr))
        hipLaunchKernelGGL((cubic_operator_march_kernel<2, FRN>), dim3(fr_march_blocks(n, xr)), dim3(BLK), 0, c->stream, d_x,
                           d_b, q * h2, (const float *)nullptr, ilog2(n), h2, invsix, (float *)nullptr, c->d_hist, xr);
    else
        hipLaunchKernelGGL((cubic_operator_kernel<2, FRN>), dim3(pm_stream_blocks(total, BLK * 4)), dim3(BLK), 0, c->stream,
                           d_x, d_b, q * h2, (const float *)nullptr, ilog2(n), h2, invsix, (float *)nullptr, c->d_hist, xr);
    PM_LAUNCH_CHECK();
    if (int rb_rc = pm_read_back(c, h_sum, c->d_hist, sizeof(double))) return rb_rc;
    return PM_OK;
}

template <int FRN>
static int fr_residual_error(pm_ctx *c, const float *d_x, const float *d_b, float q, int n, float *h_out) {
    PM_REQUIRE(h_out, "NULL argument");
    double s = 0;
    int rc = fr_residual_sumsq<FRN>(c, d_x, d_b, q, n, xr_cube(n), &s);
    *h_out = (float)sqrt(s);
    return rc;
}

extern "C" int pm_cubic_operator(pm_ctx *c, const float *d_x, const float *d_b, float q, const float *d_rhs, int n,
                                 float *d_out) {
    return fr_operator<1>(c, d_x, d_b, q, d_rhs, n, d_out, xr_cube(n));
}
extern "C" int pm_quartic_operator(pm_ctx *c, const float *d_x, const float *d_b, float q, const float *d_rhs, int n,
                                   float *d_out) {
    return fr_operator<2>(c, d_x, d_b, q, d_rhs, n, d_out, xr_cube(n));
}
extern "C" int pm_cubic_residual_error(pm_ctx *c, const float *d_x, const float *d_b, float q, int n, float *h_out) {
    return fr_residual_error<1>(c, d_x, d_b, q, n, h_out);
}
extern "C" int pm_quartic_residual_error(pm_ctx *c, const float *d_x, const float *d_b, float q, int n, float *h_out) {
    return fr_residual_error<2>(c, d_x, d_b, q, n, h_out);
}

// slab forms (level arrays [nx + 2, n, n], ghosts of d_x filled by the caller); fr_n = 1 cubic, 2 quartic
extern "C" int pm_fr_operator_slab(pm_ctx *c, int fr_n, const float *d_x, const float *d_b, float q, const float *d_rhs,
                                   int n, int nx, float *d_out) {
    PM_REQUIRE(nx >= 1 && (fr_n == 1 || fr_n == 2), "bad argument");
    return fr_n == 1 ? fr_operator<1>(c, d_x, d_b, q, d_rhs, n, d_out, xr_slab(nx))
                     : fr_operator<2>(c, d_x, d_b, q, d_rhs, n, d_out, xr_slab(nx));
}
extern "C" int pm_fr_residual_sumsq_slab(pm_ctx *c, int fr_n, const float *d_x, const float *d_b, float q, int n, int nx,
                                         double *h_sum) {
    PM_REQUIRE(nx >= 1 && (fr_n == 1 || fr_n == 2), "bad argument");
    return fr_n == 1 ? fr_residual_sumsq<1>(c, d_x, d_b, q, n, xr_slab(nx), h_sum)
                     : fr_residual_sumsq<2>(c, d_x, d_b, q, n, xr_slab(nx), h_sum);
}

// cubic.py:217-265 initialise_potential (fp64 per cell under Numba's typing)
__global__ __launch_bounds__(BLK) void cubic_init_kernel(const float *__restrict__ b, double threeh2, double d1,
                                                         int64_t count, float *__restrict__ out) {
    const double inv3 = 1.0 / 3;
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < count; t += (int64_t)gridDim.x * BLK) {
        const double d0 = -threeh2 * (double)b[t];
        const double C = pow(0.5 * (d1 + sqrt(d1 * d1 - 4.0 * (d0 * d0 * d0))), inv3);
        out[t] = (float)(-inv3 * (C + d0 / C));
    }
}

extern "C" int pm_cubic_initialise_potential(pm_ctx *c, const float *d_b, float q, int n, float *d_out) {
    PM_REQUIRE(c && d_b && d_out, "NULL argument");
    PM_CHECK_GRID(n);
    const int64_t total = (int64_t)n * n * n;
    const float h2 = (float)(1.0 / ((double)n * n));
    hipLaunchKernelGGL(cubic_init_kernel, dim3(pm_stream_blocks(total, BLK)), dim3(BLK), 0, c->stream, d_b,
                       3.0 * (double)h2, 27.0 * (double)h2 * (double)q, total, d_out);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

// the same on `count` cells of a level with n cells per side (slab level arrays)
extern "C" int pm_cubic_initialise_potential_flat(pm_ctx *c, const float *d_b, float q, int n, int64_t count,
                                                  float *d_out) {
    PM_REQUIRE(c && d_b && d_out && count > 0, "bad argument");
    PM_CHECK_GRID(n);
    const float h2 = (float)(1.0 / ((double)n * n));
    hipLaunchKernelGGL(cubic_init_kernel, dim3(pm_stream_blocks(count, BLK)), dim3(BLK), 0, c->stream, d_b,
                       3.0 * (double)h2, 27.0 * (double)h2 * (double)q, count, d_out);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

// quartic.py:214-266 initialise_potential (fp64 throughout)
__global__ __launch_bounds__(BLK) void quartic_init_kernel(const float *__restrict__ b, double h2, double d0,
                                                           int64_t count, float *__restrict__ out) {
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < count; t += (int64_t)gridDim.x * BLK) {
        const double p = h2 * (double)b[t];
        const double d1 = 27.0 * (p * p);
        const double Q = pow(0.5 * (d1 + sqrt(d1 * d1 - 4.0 * (d0 * d0 * d0))), 1.0 / 3.0);
        const double S = 0.5 * sqrt((Q + d0 / Q) * (1.0 / 3.0));
        out[t] = (float)(-S + 0.5 * sqrt(-4.0 * (S * S) + p / S));
    }
}

extern "C" int pm_quartic_initialise_potential(pm_ctx *c, const float *d_b, float q, int n, float *d_out) {
    PM_REQUIRE(c && d_b && d_out, "NULL argument");
    PM_CHECK_GRID(n);
    const int64_t total = (int64_t)n * n * n;
    const double h2 = 1.0 / ((double)n * n);
    hipLaunchKernelGGL(quartic_init_kernel, dim3(pm_stream_blocks(total, BLK)), dim3(BLK), 0, c->stream, d_b, h2,
                       12.0 * h2 * (double)q, total, d_out);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

extern "C" int pm_quartic_initialise_potential_flat(pm_ctx *c, const float *d_b, float q, int n, int64_t count,
                                                    float *d_out) {
    PM_REQUIRE(c && d_b && d_out && count > 0, "bad argument");
    PM_CHECK_GRID(n);
    const double h2 = 1.0 / ((double)n * n);
    hipLaunchKernelGGL(quartic_init_kernel, dim3(pm_stream_blocks(count, BLK)), dim3(BLK), 0, c->stream, d_b, h2,
                       12.0 * h2 * (double)q, count, d_out);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

// cubic.py:269-442 / quartic.py:270-444 gauss_seidel and cubic.py:446-668 / quartic.py:448-632
// gauss_seidel_with_rhs: one launch per colour, one thread per cell of that colour.  `c_q` is 27 h^2 q
// for the cubic (d1 = c_q - 27 rhs) and q h^2 for the quartic (qq = c_q - rhs).
template <int COLOUR, bool WITH_RHS, int FRN>
__global__ __launch_bounds__(BLK) void cubic_gauss_seidel_kernel(float *__restrict__ x, const float *__restrict__ b,
                                                                 float d1_q, const float *__restrict__ rhs, int log2n,
                                                                 float h2, float invsix, float f_relax, XRange xr,
                                                                 int gpar) {
    const int n = 1 << log2n, m = n - 1, nh = n >> 1, log2nh = log2n - 1;
    const int64_t total = ((int64_t)xr.ni << log2n) << log2nh;
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLK) {
        const int kh = (int)(t & (nh - 1)), j = (int)((t >> log2nh) & m), i = xr.i0 + (int)(t >> (log2nh + log2n));
        const int k = 2 * kh + ((i + gpar + j + COLOUR) & 1);  // gpar: parity of (global - local) plane index
        const int64_t c = cidx(i, j, k, log2n);
        const float p = cubic_p<FRN>(x, b, i, j, k, m, log2n, h2, invsix, xr.xmask);
        float sol;
        if (FRN == 1) {
            const float d1 = WITH_RHS ? d1_q - 27.0f * rhs[c] : d1_q;
            sol = (float)cubic_root(p, d1);
        } else {
            const float qq = WITH_RHS ? d1_q - rhs[c] : d1_q;
            sol = (float)quartic_root(p, qq);
        }
        const float xc = x[c];
        x[c] = xc + f_relax * (sol - xc);
    }
}

// colours: 3 = red then black (the whole sweep), 1 = red only, 0 = black only (slabs: ghosts refreshed between)
// Row form of the same half-sweep for n >= 16, as lap_gauss_seidel_rows_kernel: a thread owns 8 consecutive k-cells of
// one (i, j) row and updates the four of the launch's colour; the row, its four neighbour rows, the density term (and
// the right-hand side) are 16-byte accesses, where the cell-per-thread kernel above touches every other float of
// every line it loads (measured at 512^3: 1.19 ms of its 1.53 ms sweep are those accesses, 0.34 the fp64 root).
// Same expression per cell: cubic_p's sum in its order, the same root, the same relaxation.
template <int COLOUR, bool WITH_RHS, int FRN>
__global__ __launch_bounds__(BLK) void cubic_gauss_seidel_rows_kernel(float *__restrict__ x, const float *__restrict__ b,
                                                                      float d1_q, const float *__restrict__ rhs, int log2n,
                                                                      float h2, float invsix, float f_relax, XRange xr,
                                                                      int gpar) {
    const int n = 1 << log2n, m = n - 1, lseg = log2n - 3;
    const int64_t total = ((int64_t)xr.ni << log2n) << lseg;
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLK) {
        const int k0 = (int)(t & ((1 << lseg) - 1)) << 3;
        const int64_t row0 = t >> lseg;
        const int j = (int)(row0 & m), il = xr.i0 + (int)(row0 >> log2n), i = il + gpar;  // i: parity of the global plane
        const int64_t row = ((int64_t)il << log2n) + j;
        auto rowp = [&](int ii, int jj) -> const float * {
            return x + ((((int64_t)(ii & xr.xmask) << log2n) + (jj & m)) << log2n);
        };
        float *mine = x + (row << log2n);
        float v[10], xm[8], xp[8], ym[8], yp[8], bb[8], rr[8];
        auto ld8 = [&](const float *p, float *d) {
            const float4 q0 = *reinterpret_cast<const float4 *>(p + k0), q1 = *reinterpret_cast<const float4 *>(p + k0 + 4);
            d[0] = q0.x, d[1] = q0.y, d[2] = q0.z, d[3] = q0.w, d[4] = q1.x, d[5] = q1.y, d[6] = q1.z, d[7] = q1.w;
        };
        ld8(mine, v + 1);
        v[0] = mine[(k0 - 1) & m], v[9] = mine[(k0 + 8) & m];
        ld8(rowp(il - 1, j), xm), ld8(rowp(il + 1, j), xp), ld8(rowp(il, j - 1), ym), ld8(rowp(il, j + 1), yp);
        ld8(b + (row << log2n), bb);
        if (WITH_RHS) ld8(rhs + (row << log2n), rr);
        const int o = (COLOUR + i + j) & 1;  // k parity of this row's cells of the launch's colour
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int l = 2 * q + o;
            // cubic_p: h2 b - (1/6) (x- , y-, z-, z+, y+, x+ powers summed in that order)
            const float p = h2 * bb[l] - invsix * (fr_pw<FRN>(xm[l]) + fr_pw<FRN>(ym[l]) + fr_pw<FRN>(v[l]) + fr_pw<FRN>(v[l + 2]) +
                                                   fr_pw<FRN>(yp[l]) + fr_pw<FRN>(xp[l]));
            float sol;
            if (FRN == 1) {
                const float d1 = WITH_RHS ? d1_q - 27.0f * rr[l] : d1_q;
                sol = (float)cubic_root(p, d1);
            } else {
                const float qq = WITH_RHS ? d1_q - rr[l] : d1_q;
                sol = (float)quartic_root(p, qq);
            }
            const float xc = v[l + 1];
            v[l + 1] = xc + f_relax * (sol - xc);
        }
        *reinterpret_cast<float4 *>(mine + k0) = make_float4(v[1], v[2], v[3], v[4]);
        *reinterpret_cast<float4 *>(mine + k0 + 4) = make_float4(v[5], v[6], v[7], v[8]);
    }
}

template <int COLOUR, bool WITH_RHS, int FRN>
static void fr_gauss_seidel_launch(pm_ctx *c, bool rows, float *d_x, const float *d_b, float c_q, const float *d_rhs, int l2,
                                   float h2, float invsix, float f_relax, XRange xr, int gpar) {
    const int n = 1 << l2;
    if (rows) {
        const int64_t threads = (int64_t)xr.ni * n * (n / 8);
        hipLaunchKernelGGL((cubic_gauss_seidel_rows_kernel<COLOUR, WITH_RHS, FRN>), dim3(pm_stream_blocks(threads, BLK)), dim3(BLK),
                           0, c->stream, d_x, d_b, c_q, d_rhs, l2, h2, invsix, f_relax, xr, gpar);
    } else {
        const int64_t total = (int64_t)xr.ni * n * (n / 2);
        hipLaunchKernelGGL((cubic_gauss_seidel_kernel<COLOUR, WITH_RHS, FRN>), dim3(pm_stream_blocks(total, BLK)), dim3(BLK), 0,
                           c->stream, d_x, d_b, c_q, d_rhs, l2, h2, invsix, f_relax, xr, gpar);
    }
}

// c_q of the f(R) sweeps: 27 h^2 q for the cubic (d1 = c_q - 27 rhs), q h^2 for the quartic (qq = c_q - rhs)
template <int FRN>
static float fr_c_q(float q, float h2, bool with_rhs) {
    if (FRN == 2) return q * h2;                 // quartic.py:306, 488
    if (with_rhs) return 27.0f * h2 * q;         // cubic.py:487 (all fp32)
    return (float)(27.0 * (double)h2 * (double)q);  // cubic.py:305 (int * f4 -> f8, then f4)
}

template <int FRN>
static int fr_gauss_seidel(pm_ctx *c, float *d_x, const float *d_b, float q, const float *d_rhs, int n, float f_relax,
                           XRange xr, int gpar, int colours) {
    PM_REQUIRE(c && d_x && d_b, "NULL argument");
    PM_CHECK_GRID(n);
    const int64_t total = (int64_t)xr.ni * n * (n / 2);
    const int blocks = pm_stream_blocks(total, BLK);
    const float h2 = (float)(1.0 / ((double)n * n)), invsix = (float)(1.0 / 6);
    const int l2 = ilog2(n);
    const float c_q = fr_c_q<FRN>(q, h2, d_rhs != nullptr);
    const bool red = colours == 3 || colours == 1, black = colours == 3 || colours == 0;
    const bool rows = c->fr_rows_sweep && n >= 16 && pm_aligned16(d_x) && pm_aligned16(d_b) && (!d_rhs || pm_aligned16(d_rhs));
    (void)total, (void)blocks;
    if (d_rhs) {
        if (red) fr_gauss_seidel_launch<1, true, FRN>(c, rows, d_x, d_b, c_q, d_rhs, l2, h2, invsix, f_relax, xr, gpar);
        if (black) fr_gauss_seidel_launch<0, true, FRN>(c, rows, d_x, d_b, c_q, d_rhs, l2, h2, invsix, f_relax, xr, gpar);
    } else {
        if (red) fr_gauss_seidel_launch<1, false, FRN>(c, rows, d_x, d_b, c_q, d_rhs, l2, h2, invsix, f_relax, xr, gpar);
        if (black) fr_gauss_seidel_launch<0, false, FRN>(c, rows, d_x, d_b, c_q, d_rhs, l2, h2, invsix, f_relax, xr, gpar);
    }
    PM_LAUNCH_CHECK();
    return PM_OK;
}

// The cell update of the f(R) sweeps for the single-pass kernel (gs_fused_kernel): cubic_p's sum in its order, the root,
// the relaxation -- the expression of the two kernels above.
template <bool WITH_RHS, int FRN>
struct FrCell {
    static constexpr bool HAS_RHS = WITH_RHS;
    float h2, invsix, f_relax, c_q;
    __device__ __forceinline__ float operator()(int, float xm, float xp, float ym, float yp, float zm, float zp, float bval,
                                                float rval, float xc) const {
        const float p = h2 * bval - invsix * (fr_pw<FRN>(xm) + fr_pw<FRN>(ym) + fr_pw<FRN>(zm) + fr_pw<FRN>(zp) + fr_pw<FRN>(yp) +
                                              fr_pw<FRN>(xp));
        float sol;
        if (FRN == 1) {
            const float d1 = WITH_RHS ? c_q - 27.0f * rval : c_q;
            sol = (float)cubic_root(p, d1);
        } else {
            const float qq = WITH_RHS ? c_q - rval : c_q;
            sol = (float)quartic_root(p, qq);
        }
        return xc + f_relax * (sol - xc);
    }
};

// n_sweeps red + black sweeps of the scalaron starting from `cur`: the single-pass kernel ping-ponging between `cur` and
// `other` where it applies (n >= 256, as lap_smooth_pp), the two half-sweep launches in place otherwise.  *where = the
// array that holds the result.
template <int FRN>
static int fr_smooth_pp(pm_ctx *c, float *cur, float *other, const float *b, float q, const float *rhs, int n, int n_sweeps,
                        float f_relax, float **where) {
    int rc;
    *where = cur;
    if (!lap_fused_applies(c, n, cur, b, other) || (rhs && !pm_aligned16(rhs))) {
        for (int s = 0; s < n_sweeps; ++s)
            if ((rc = fr_gauss_seidel<FRN>(c, cur, b, q, rhs, n, f_relax, xr_cube(n), 0, 3))) return rc;
        return PM_OK;
    }
    const float h2 = (float)(1.0 / ((double)n * n)), invsix = (float)(1.0 / 6);
    const float c_q = fr_c_q<FRN>(q, h2, rhs != nullptr);
    const int lx = n >= 512 ? 64 : 32;
    const unsigned blocks = (unsigned)((n / GF_TJ) * (n / GF_TK) * (n / lx));
    float *src = cur, *dst = other;
    for (int s = 0; s < n_sweeps; ++s) {
#define FR_FUSED(LX, WITH)                                                                                                   \
    hipLaunchKernelGGL((gs_fused_kernel<LX, FrCell<WITH, FRN>>), dim3(blocks), dim3(BLK), 0, c->stream, src, b, rhs, dst, ilog2(n), \
                       FrCell<WITH, FRN>{h2, invsix, f_relax, c_q})
        if (lx == 64) {
            if (rhs) FR_FUSED(64, true);
            else FR_FUSED(64, false);
        } else {
            if (rhs) FR_FUSED(32, true);
            else FR_FUSED(32, false);
        }
#undef FR_FUSED
        PM_LAUNCH_CHECK();
        float *t = src;
        src = dst, dst = t;
    }
    *where = src;
    return PM_OK;
}

// cubic.py:1065 smoothing / :1100 smoothing_with_rhs (quartic.py:1066, 1101) on device arrays: n_smoothing sweeps with
// omega = 1.25, result in d_x
extern "C" int pm_fr_smoothing(pm_ctx *c, int fr_n, float *d_x, const float *d_b, float q, const float *d_rhs, int n,
                               int n_smoothing) {
    PM_REQUIRE(c && d_x && d_b && n_smoothing >= 0 && (fr_n == 1 || fr_n == 2), "bad argument");
    PM_CHECK_GRID(n);
    float *tmp = nullptr, *where = d_x;
    int rc;
    if (n >= 256 && (rc = mg_arena(c, ((size_t)n * n * n + 1024) * sizeof(float), &tmp))) return rc;
    rc = fr_n == 1 ? fr_smooth_pp<1>(c, d_x, tmp, d_b, q, d_rhs, n, n_smoothing, 1.25f, &where)
                   : fr_smooth_pp<2>(c, d_x, tmp, d_b, q, d_rhs, n, n_smoothing, 1.25f, &where);
    if (rc) return rc;
    if (where != d_x)
        PM_HIP(hipMemcpyAsync(d_x, where, (size_t)n * n * n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    return PM_OK;
}

// 1 (default): pm_fr_vcycle restricts operator, solution and density term in one marching pass from 512^3 up; 0: operator
// kernel + three restrictions (comparison)
extern "C" int pm_set_fr_fused_restrict(pm_ctx *c, int on) {
    PM_REQUIRE(c, "NULL argument");
    c->fr_fused_restrict = on ? 1 : 0;
    return PM_OK;
}

// 1 (default): the f(R) smoothers use the row kernels for n >= 16; 0: one cell per thread (comparison)
extern "C" int pm_set_fr_rows_sweep(pm_ctx *c, int on) {
    PM_REQUIRE(c, "NULL argument");
    c->fr_rows_sweep = on ? 1 : 0;
    return PM_OK;
}

extern "C" int pm_fr_gauss_seidel_slab(pm_ctx *c, int fr_n, float *d_x, const float *d_b, float q, const float *d_rhs,
                                       int n, int nx, int x_lo, int colour, float f_relax) {
    PM_REQUIRE(nx >= 1 && (fr_n == 1 || fr_n == 2) && (colour == 0 || colour == 1), "bad argument");
    const int gpar = (x_lo - 1) & 1;
    return fr_n == 1 ? fr_gauss_seidel<1>(c, d_x, d_b, q, d_rhs, n, f_relax, xr_slab(nx), gpar, colour)
                     : fr_gauss_seidel<2>(c, d_x, d_b, q, d_rhs, n, f_relax, xr_slab(nx), gpar, colour);
}

extern "C" int pm_cubic_gauss_seidel(pm_ctx *c, float *d_x, const float *d_b, float q, const float *d_rhs, int n,
                                     float f_relax) {
    return fr_gauss_seidel<1>(c, d_x, d_b, q, d_rhs, n, f_relax, xr_cube(n), 0, 3);
}
extern "C" int pm_quartic_gauss_seidel(pm_ctx *c, float *d_x, const float *d_b, float q, const float *d_rhs, int n,
                                       float f_relax) {
    return fr_gauss_seidel<2>(c, d_x, d_b, q, d_rhs, n, f_relax, xr_cube(n), 0, 3);
}

// ------------------------------------------------------------------------------------------------
// One whole Full-Approximation-Scheme V-cycle for the f(R) scalaron (multigrid.py:521-579) issued from C, like
// pm_lap_vcycle: per level -- Npre smoothing sweeps, restricted residual (rhs-less at the top: minus_restriction of the
// operator; below: restriction of rhs - operator), restricted solution and density term, coarse right-hand side
// 4 res + L(x_c), coarse cycle (only smoothed below 8^3 cells), x += prolongation(x_corr - x_c), Npost sweeps.  Every
// operation is the kernel the per-operator entry points launch (same bits); ~45 Python -> C calls per level otherwise.

template <int FRN>
static int fr_vcycle_level(pm_ctx *c, float *x, const float *b, float q, const float *rhs, int n, int npre, int npost,
                           float *scratch) {
    const float f_relax = 1.25f;  // cubic.py:1092 / quartic.py
    int rc;
    const int nc = n / 2;
    const int64_t cells = (int64_t)n * n * n, ccells = (int64_t)nc * nc * nc;
    // scratch of this level: one fine grid (the single-pass sweeps' ping-pong partner, then operator / residual), five
    // coarse ones
    float *fine = scratch, *res_c = fine + cells, *x_c = res_c + ccells, *x_corr_c = x_c + ccells, *b_c = x_corr_c + ccells,
          *L_c = b_c + ccells, *next = L_c + ccells;
    const bool pingpong = lap_fused_applies(c, n, x, b, fine) && (!rhs || pm_aligned16(rhs));
    float *cur = x;  // the array that holds the level's solution at the moment; the other of (x, fine) is free
    if ((rc = fr_smooth_pp<FRN>(c, x, fine, b, q, rhs, n, npre, f_relax, &cur))) return rc;
    float *op = cur == x ? fine : x;
    // multigrid.restrict_residual (:282): no rhs -> minus_restriction(operator); rhs -> restriction(rhs - operator)
    if (c->fr_fused_restrict && fr_march_applies(c, n, cur, b, rhs, res_c) && pm_aligned16(x_c) && pm_aligned16(b_c)) {
        const float h2 = (float)(1.0 / ((double)n * n)), invsix = (float)(1.0 / 6);
        const unsigned blocks = fr_march_blocks(n, xr_cube(n));
        if (rhs)
            hipLaunchKernelGGL((fr_restrict_march_kernel<true, FRN>), dim3(blocks), dim3(BLK), 0, c->stream, cur, b, q * h2, rhs,
                               ilog2(n), h2, invsix, 0.125f, res_c, x_c, b_c);
        else
            hipLaunchKernelGGL((fr_restrict_march_kernel<false, FRN>), dim3(blocks), dim3(BLK), 0, c->stream, cur, b, q * h2, rhs,
                               ilog2(n), h2, invsix, -0.125f, res_c, x_c, b_c);
        PM_LAUNCH_CHECK();
    } else {
        if ((rc = fr_operator<FRN>(c, cur, b, q, rhs, n, op, xr_cube(n)))) return rc;
        if ((rc = pm_restriction(c, op, n, rhs ? 0.125f : -0.125f, res_c))) return rc;
        if ((rc = pm_restriction(c, cur, n, 0.125f, x_c))) return rc;
        if ((rc = pm_restriction(c, b, n, 0.125f, b_c))) return rc;
    }
    PM_HIP(hipMemcpyAsync(x_corr_c, x_c, (size_t)ccells * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    if ((rc = fr_operator<FRN>(c, x_c, b_c, q, nullptr, nc, L_c, xr_cube(nc)))) return rc;
    // res_c = 4 res_c + L_c  (utils.linear_operator_vectors_inplace with multigrid.normalisation_residual = 4)
    if ((rc = pm_linear_operator_vectors(c, res_c, 4.0f, L_c, 1.0f, ccells))) return rc;
    if (n <= 8) {
        for (int s = 0; s < npre; ++s)
            if ((rc = fr_gauss_seidel<FRN>(c, x_corr_c, b_c, q, res_c, nc, f_relax, xr_cube(nc), 0, 3))) return rc;
    } else if ((rc = fr_vcycle_level<FRN>(c, x_corr_c, b_c, q, res_c, nc, npre, npost, next))) {
        return rc;
    }
    if ((rc = pm_axpy(c, x_corr_c, x_c, -1.0f, ccells))) return rc;
    // the corrected solution goes where an odd / even number of ping-pong sweeps will leave the result in x
    float *target = pingpong && (npost & 1) ? fine : x;
    if ((rc = prolongation_launch(c, target, x_corr_c, nc, 1, xr_cube(n), xr_cube(nc), cur))) return rc;
    float *where = target;
    if ((rc = fr_smooth_pp<FRN>(c, target, target == x ? fine : x, b, q, rhs, n, npost, f_relax, &where))) return rc;
    if (where != x) PM_HIP(hipMemcpyAsync(x, where, (size_t)cells * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    return PM_OK;
}

extern "C" int pm_fr_vcycle(pm_ctx *c, int fr_n, float *d_x, const float *d_b, float q, int n, int npre, int npost) {
    PM_REQUIRE(c && d_x && d_b && npre >= 0 && npost >= 0 && (fr_n == 1 || fr_n == 2), "bad argument");
    PM_CHECK_GRID(n);
    PM_REQUIRE(n >= 8, "pm_fr_vcycle needs n >= 8 (ncoarse >= 3)");
    // per level n^3 + 5 (n/2)^3 floats: (1 + 5/8) (1 + 1/8 + ...) n^3 < 13 n^3 / 7
    float *scratch = nullptr;
    int rc = mg_arena(c, ((size_t)n * n * n * 13 / 7 + 4096) * sizeof(float), &scratch);
    if (rc) return rc;
    return vcycle_graphed(c, 16 * fr_n + (c->mg_fused_sweep ? 1 : 0) + (c->fr_rows_sweep ? 2 : 0) + (c->fr_fused_restrict ? 4 : 0), d_x, d_b, q, n, npre, npost, [&]() {
        return fr_n == 1 ? fr_vcycle_level<1>(c, d_x, d_b, q, nullptr, n, npre, npost, scratch)
                         : fr_vcycle_level<2>(c, d_x, d_b, q, nullptr, n, npre, npost, scratch);
    });
}

// ------------------------------------------------------------------------------------------------
// QUMOND (mond.py)

// mond.py:16-162; kind: 0 simple, 1 n, 2 beta, 3 gamma, 4 delta
template <int KIND>
__device__ __forceinline__ float mond_nu(float y, float alpha) {
    if (KIND == 0) return 0.5f + sqrtf(0.25f + 1.0f / y);
    if (KIND == 1) {
        const int n = (int)alpha;
        return powf(0.5f + sqrtf(0.25f + powf(y, (float)(-n))), (float)(1.0 / n));
    }
    if (KIND == 2) {
        const float e = expf(-y);
        float nu = alpha * e;
        const float om = 1.0f - e;
        if (om > 0) nu += powf(om, -0.5f);
        return nu;
    }
    if (KIND == 3) {
        const float e = expf(-powf(y, (float)(0.5 * alpha)));
        return powf(1.0f - e, (float)(-1.0 / alpha)) + (1.0f - (float)(1.0 / alpha)) * e;
    }
    return powf(1.0f - expf(-powf(y, (float)(0.5 * alpha))), (float)(-1.0 / alpha));
}

// nu of y = |grad| / g0 from the SQUARED gradient norm s.  FAST (the simple function only, the default): nu = 1/2 +
// sqrt(1/4 + g0 rsq(s)) with the hardware's 1-ulp v_rsq_f32 / v_sqrt_f32 -- 1.5e-7 relative, the size of the six
// roundings of the exact chain sqrt, * 1/g0, 1/y, + 1/4, sqrt, + 1/2 -- instead of two IEEE square roots and an IEEE
// division (12 + 12 + 10 instructions, six times per cell: the 512^3 kernel was bound by them at 0.10 of the HBM
// roofline).  s = 0 gives inf either way.  pm_set_mond_fast_nu(ctx, 0) restores the exact chain (comparison).
template <int KIND, bool FAST>
__device__ __forceinline__ float mond_nu_sq(float s, float inv_g0, float g0, float alpha) {
    if (FAST && KIND == 0) return 0.5f + __builtin_amdgcn_sqrtf(0.25f + g0 * __builtin_amdgcn_rsqf(s));
    return mond_nu<KIND>(sqrtf(s) * inv_g0, alpha);
}

// mond.py:171-316 rhs_simple and its four siblings
template <int KIND, bool FAST>
__global__ __launch_bounds__(BLK) void mond_rhs_kernel(const float *__restrict__ pot, float *__restrict__ out,
                                                       int log2n, float inv_g0, float g0, float alpha, XRange xr) {
    const int n = 1 << log2n, m = n - 1;
    const float invh = (float)n, inv4h = (float)(0.25 * n);
    const int64_t total = (int64_t)xr.ni << (2 * log2n);
#define P(a, b, c) pot[cidx((a)&xr.xmask, (b)&m, (c)&m, log2n)]
    for (int64_t t0 = (int64_t)blockIdx.x * BLK + threadIdx.x; t0 < total; t0 += (int64_t)gridDim.x * BLK) {
        const int k = (int)(t0 & m), j = (int)((t0 >> log2n) & m), i = xr.i0 + (int)(t0 >> (2 * log2n));
        const int64_t t = cidx(i, j, k, log2n);
        const int im1 = i - 1, ip1 = i + 1, jm1 = j - 1, jp1 = j + 1, km1 = k - 1, kp1 = k + 1;
        const float p000 = P(i, j, k);
        const float Axx = invh * (p000 - P(im1, j, k));
        const float Axy = inv4h * (P(i, jp1, k) - P(i, jm1, k) + P(im1, jp1, k) - P(im1, jm1, k));
        const float Axz = inv4h * (P(i, j, kp1) - P(i, j, km1) + P(im1, j, kp1) - P(im1, j, km1));
        const float sAx = Axx * Axx + Axy * Axy + Axz * Axz;
        const float Bxx = invh * (-p000 + P(ip1, j, k));
        const float Bxy = inv4h * (P(ip1, jp1, k) - P(ip1, jm1, k) + P(i, jp1, k) - P(i, jm1, k));
        const float Bxz = inv4h * (P(ip1, j, kp1) - P(ip1, j, km1) + P(i, j, kp1) - P(i, j, km1));
        const float sBx = Bxx * Bxx + Bxy * Bxy + Bxz * Bxz;
        const float Ayy = invh * (p000 - P(i, jm1, k));
        const float Ayx = inv4h * (P(ip1, j, k) - P(im1, j, k) + P(ip1, jm1, k) - P(im1, jm1, k));
        const float Ayz = inv4h * (P(i, j, kp1) - P(i, j, km1) + P(i, jm1, kp1) - P(i, jm1, km1));
        const float sAy = Ayx * Ayx + Ayy * Ayy + Ayz * Ayz;
        const float Byy = invh * (-p000 + P(i, jp1, k));
        const float Byx = inv4h * (P(ip1, jp1, k) - P(im1, jp1, k) + P(ip1, j, k) - P(im1, j, k));
        const float Byz = inv4h * (P(i, jp1, kp1) - P(i, jp1, km1) + P(i, j, kp1) - P(i, j, km1));
        const float sBy = Byx * Byx + Byy * Byy + Byz * Byz;
        const float Azz = invh * (p000 - P(i, j, km1));
        const float Azx = inv4h * (P(ip1, j, k) - P(im1, j, k) + P(ip1, j, km1) - P(im1, j, km1));
        const float Azy = inv4h * (P(i, jp1, k) - P(i, jm1, k) + P(i, jp1, km1) - P(i, jm1, km1));
        const float sAz = Azx * Azx + Azy * Azy + Azz * Azz;
        const float Bzz = invh * (-p000 + P(i, j, kp1));
        const float Bzx = inv4h * (P(ip1, j, kp1) - P(im1, j, kp1) + P(ip1, j, k) - P(im1, j, k));
        const float Bzy = inv4h * (P(i, jp1, kp1) - P(i, jm1, kp1) + P(i, jp1, k) - P(i, jm1, k));
        const float sBz = Bzx * Bzx + Bzy * Bzy + Bzz * Bzz;
        const float nAx = mond_nu_sq<KIND, FAST>(sAx, inv_g0, g0, alpha), nAy = mond_nu_sq<KIND, FAST>(sAy, inv_g0, g0, alpha),
                    nAz = mond_nu_sq<KIND, FAST>(sAz, inv_g0, g0, alpha), nBx = mond_nu_sq<KIND, FAST>(sBx, inv_g0, g0, alpha),
                    nBy = mond_nu_sq<KIND, FAST>(sBy, inv_g0, g0, alpha), nBz = mond_nu_sq<KIND, FAST>(sBz, inv_g0, g0, alpha);
        out[t] = invh * (nBx * Bxx - nAx * Axx + nBy * Byy - nAy * Ayy + nBz * Bzz - nAz * Azz);
    }
#undef P
}

// Four consecutive k-cells per thread (n >= 16): the nine rows (i-1..i+1, j-1..j+1) as one aligned float4 each, the
// five rows of the "plus" stencil with their k-1 / k+4 neighbours -- 19 loads for four cells where the cell-per-thread
// kernel issues 76 -- and the same expressions per cell (1.49 -> ms at 512^3).
template <int KIND, bool FAST>
__global__ __launch_bounds__(BLK) void mond_rhs_rows_kernel(const float *__restrict__ pot, float *__restrict__ out,
                                                            int log2n, float inv_g0, float g0, float alpha, XRange xr) {
    const int n = 1 << log2n, m = n - 1, lq = log2n - 2;
    const float invh = (float)n, inv4h = (float)(0.25 * n);
    const int64_t total = ((int64_t)xr.ni << log2n) << lq;
    for (int64_t t0 = (int64_t)blockIdx.x * BLK + threadIdx.x; t0 < total; t0 += (int64_t)gridDim.x * BLK) {
        const int k0 = (int)(t0 & ((1 << lq) - 1)) << 2, j = (int)((t0 >> lq) & m), i = xr.i0 + (int)(t0 >> (lq + log2n));
        float Pl[3][3][6];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float *r = pot + ((((int64_t)((i + a - 1) & xr.xmask) << log2n) + ((j + b - 1) & m)) << log2n);
                const float4 v = *reinterpret_cast<const float4 *>(r + k0);
                Pl[a][b][1] = v.x, Pl[a][b][2] = v.y, Pl[a][b][3] = v.z, Pl[a][b][4] = v.w;
                const bool plus = a == 1 || b == 1;  // (the corner rows are only read at k itself)
                Pl[a][b][0] = plus ? r[(k0 - 1) & m] : 0.f;
                Pl[a][b][5] = plus ? r[(k0 + 4) & m] : 0.f;
            }
        float res[4];
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) {
            const int kk = c2 + 1;
        const float p000 = Pl[1][1][kk];
        const float Axx = invh * (p000 - Pl[0][1][kk]);
        const float Axy = inv4h * (Pl[1][2][kk] - Pl[1][0][kk] + Pl[0][2][kk] - Pl[0][0][kk]);
        const float Axz = inv4h * (Pl[1][1][kk + 1] - Pl[1][1][kk - 1] + Pl[0][1][kk + 1] - Pl[0][1][kk - 1]);
        const float sAx = Axx * Axx + Axy * Axy + Axz * Axz;
        const float Bxx = invh * (-p000 + Pl[2][1][kk]);
        const float Bxy = inv4h * (Pl[2][2][kk] - Pl[2][0][kk] + Pl[1][2][kk] - Pl[1][0][kk]);
        const float Bxz = inv4h * (Pl[2][1][kk + 1] - Pl[2][1][kk - 1] + Pl[1][1][kk + 1] - Pl[1][1][kk - 1]);
        const float sBx = Bxx * Bxx + Bxy * Bxy + Bxz * Bxz;
        const float Ayy = invh * (p000 - Pl[1][0][kk]);
        const float Ayx = inv4h * (Pl[2][1][kk] - Pl[0][1][kk] + Pl[2][0][kk] - Pl[0][0][kk]);
        const float Ayz = inv4h * (Pl[1][1][kk + 1] - Pl[1][1][kk - 1] + Pl[1][0][kk + 1] - Pl[1][0][kk - 1]);
        const float sAy = Ayx * Ayx + Ayy * Ayy + Ayz * Ayz;
        const float Byy = invh * (-p000 + Pl[1][2][kk]);
        const float Byx = inv4h * (Pl[2][2][kk] - Pl[0][2][kk] + Pl[2][1][kk] - Pl[0][1][kk]);
        const float Byz = inv4h * (Pl[1][2][kk + 1] - Pl[1][2][kk - 1] + Pl[1][1][kk + 1] - Pl[1][1][kk - 1]);
        const float sBy = Byx * Byx + Byy * Byy + Byz * Byz;
        const float Azz = invh * (p000 - Pl[1][1][kk - 1]);
        const float Azx = inv4h * (Pl[2][1][kk] - Pl[0][1][kk] + Pl[2][1][kk - 1] - Pl[0][1][kk - 1]);
        const float Azy = inv4h * (Pl[1][2][kk] - Pl[1][0][kk] + Pl[1][2][kk - 1] - Pl[1][0][kk - 1]);
        const float sAz = Azx * Azx + Azy * Azy + Azz * Azz;
        const float Bzz = invh * (-p000 + Pl[1][1][kk + 1]);
        const float Bzx = inv4h * (Pl[2][1][kk + 1] - Pl[0][1][kk + 1] + Pl[2][1][kk] - Pl[0][1][kk]);
        const float Bzy = inv4h * (Pl[1][2][kk + 1] - Pl[1][0][kk + 1] + Pl[1][2][kk] - Pl[1][0][kk]);
        const float sBz = Bzx * Bzx + Bzy * Bzy + Bzz * Bzz;
        const float nAx = mond_nu_sq<KIND, FAST>(sAx, inv_g0, g0, alpha), nAy = mond_nu_sq<KIND, FAST>(sAy, inv_g0, g0, alpha),
                    nAz = mond_nu_sq<KIND, FAST>(sAz, inv_g0, g0, alpha), nBx = mond_nu_sq<KIND, FAST>(sBx, inv_g0, g0, alpha),
                    nBy = mond_nu_sq<KIND, FAST>(sBy, inv_g0, g0, alpha), nBz = mond_nu_sq<KIND, FAST>(sBz, inv_g0, g0, alpha);
        res[c2] = invh * (nBx * Bxx - nAx * Axx + nBy * Byy - nAy * Ayy + nBz * Bzz - nAz * Azz);
        }
        *reinterpret_cast<float4 *>(out + cidx(i, j, k0, log2n)) = make_float4(res[0], res[1], res[2], res[3]);
    }
}

template <int KIND, bool FAST>
static void mond_rhs_launch_kind(pm_ctx *c, bool rows, const float *d_pot, float *d_out, int l2, float inv_g0, float g0,
                                 float alpha, XRange xr) {
    const int n = 1 << l2;
    const int64_t total = (int64_t)xr.ni * n * n;
    if (rows)
        hipLaunchKernelGGL((mond_rhs_rows_kernel<KIND, FAST>), dim3(pm_stream_blocks(total / 4, BLK)), dim3(BLK), 0, c->stream,
                           d_pot, d_out, l2, inv_g0, g0, alpha, xr);
    else
        hipLaunchKernelGGL((mond_rhs_kernel<KIND, FAST>), dim3(pm_stream_blocks(total, BLK)), dim3(BLK), 0, c->stream, d_pot,
                           d_out, l2, inv_g0, g0, alpha, xr);
}

static int mond_rhs_launch(pm_ctx *c, const float *d_pot, float *d_out, int n, float g0, int kind, float alpha,
                           XRange xr) {
    const float inv_g0 = (float)(1.0 / (double)g0);
    const int l2 = ilog2(n);
    // (the row form shares the switch of the f(R) smoothers' row kernels: pm_set_fr_rows_sweep)
    const bool rows = c->fr_rows_sweep && n >= 16 && pm_aligned16(d_pot) && pm_aligned16(d_out);
    switch (kind) {
        case 0:
            if (c->mond_fast_nu) mond_rhs_launch_kind<0, true>(c, rows, d_pot, d_out, l2, inv_g0, g0, alpha, xr);
            else mond_rhs_launch_kind<0, false>(c, rows, d_pot, d_out, l2, inv_g0, g0, alpha, xr);
            break;
        case 1: mond_rhs_launch_kind<1, false>(c, rows, d_pot, d_out, l2, inv_g0, g0, alpha, xr); break;
        case 2: mond_rhs_launch_kind<2, false>(c, rows, d_pot, d_out, l2, inv_g0, g0, alpha, xr); break;
        case 3: mond_rhs_launch_kind<3, false>(c, rows, d_pot, d_out, l2, inv_g0, g0, alpha, xr); break;
        case 4: mond_rhs_launch_kind<4, false>(c, rows, d_pot, d_out, l2, inv_g0, g0, alpha, xr); break;
        default:
            pm_set_error("MOND function %d, should be 0..4 (simple, n, beta, gamma, delta)", kind);
            return PM_EINVAL;
    }
    PM_LAUNCH_CHECK();
    return PM_OK;
}

// 1 (default): the simple interpolating function through the hardware's rsq / sqrt (mond_nu_sq); 0: the exact chain
extern "C" int pm_set_mond_fast_nu(pm_ctx *c, int on) {
    PM_REQUIRE(c, "NULL argument");
    c->mond_fast_nu = on ? 1 : 0;
    return PM_OK;
}

extern "C" int pm_mond_rhs(pm_ctx *c, const float *d_pot, float *d_out, int n, float g0, int kind, float alpha) {
    PM_REQUIRE(c && d_pot && d_out && d_pot != d_out, "bad argument");
    PM_CHECK_GRID(n);
    return mond_rhs_launch(c, d_pot, d_out, n, g0, kind, alpha, xr_cube(n));
}

// slab form: d_pot / d_out are [nx + 2, n, n] with the ghosts of d_pot filled by the caller
extern "C" int pm_mond_rhs_slab(pm_ctx *c, const float *d_pot, float *d_out, int n, int nx, float g0, int kind,
                                float alpha) {
    PM_REQUIRE(c && d_pot && d_out && d_pot != d_out && nx >= 1, "bad argument");
    PM_CHECK_GRID(n);
    return mond_rhs_launch(c, d_pot, d_out, n, g0, kind, alpha, xr_slab(nx));
}

// ------------------------------------------------------------------------------------------------
// The stopping loops of multigrid.linear (multigrid.py:57-83) and multigrid.FAS (:129-138) on this side of the C
// ABI: V-cycle, residual norm, and the reference's two exits -- the norm is under the tolerance, or it no longer
// halves (residual_err / residual_error_tmp < 2) -- with one small read-back per cycle and nothing of the host
// language in between (round 2 ran the loop in Python: a few tens of microseconds of idle device per cycle, which
// is most of a 128^3 step).  Same kernels, same order, same comparisons in float: the same number of cycles.
// *h_cycles: V-cycles taken; *h_residual: the last residual norm.  max_cycles <= 0: no bound (the reference's loop).

extern "C" int pm_lap_solve(pm_ctx *c, float *d_x, const float *d_b, int n, int npre, int npost, float tolerance,
                            int max_cycles, int *h_cycles, float *h_residual) {
    PM_REQUIRE(c && d_x && d_b && h_cycles && h_residual, "NULL argument");
    float residual_err = 1e30f, tmp = 0.f;
    int cycles = 0;
    while (residual_err > tolerance) {
        int rc = pm_lap_vcycle(c, d_x, d_b, n, npre, npost);
        if (rc) return rc;
        ++cycles;
        rc = pm_lap_residual_error(c, d_x, d_b, n, &tmp);
        if (rc) return rc;
        if (tmp < tolerance || residual_err / tmp < 2.0f) break;
        residual_err = tmp;
        if (max_cycles > 0 && cycles >= max_cycles) break;
    }
    *h_cycles = cycles, *h_residual = tmp;
    return PM_OK;
}

extern "C" int pm_fr_solve(pm_ctx *c, int fr_n, float *d_x, const float *d_b, float q, int n, int npre, int npost,
                           float tolerance, int max_cycles, int *h_cycles, float *h_residual) {
    PM_REQUIRE(c && d_x && d_b && h_cycles && h_residual, "NULL argument");
    PM_REQUIRE(fr_n == 1 || fr_n == 2, "Only f(R) with n = 1 and 2");
    float residual_err = 1e30f, tmp = 0.f;
    int cycles = 0;
    while (residual_err > tolerance) {
        int rc = pm_fr_vcycle(c, fr_n, d_x, d_b, q, n, npre, npost);
        if (rc) return rc;
        ++cycles;
        rc = fr_n == 1 ? pm_cubic_residual_error(c, d_x, d_b, q, n, &tmp) : pm_quartic_residual_error(c, d_x, d_b, q, n, &tmp);
        if (rc) return rc;
        if (tmp < tolerance || residual_err / tmp < 2.0f) break;
        residual_err = tmp;
        if (max_cycles > 0 && cycles >= max_cycles) break;
    }
    *h_cycles = cycles, *h_residual = tmp;
    return PM_OK;
}
