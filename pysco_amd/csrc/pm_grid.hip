// Streaming grid / vector kernels: utils.py element-wise ops and reductions, mesh.py gradients,
// fourier.py Green's functions and the P(k) estimator.  All HBM-bound: 16 B per lane, grid-stride.
// Built with -ffp-contract=off so a*x+y keeps the reference's two roundings (bit-exact vs oracle).
#include "pm_common.h"

#define BLK 256

// ------------------------------------------------------------------------------------------------
// element-wise (utils.py:264-298, 410-430, 644-755, 1120-1149)

enum { OP_AXPY = 0, OP_ADD, OP_SUB, OP_SCALE, OP_LINOP, OP_LINOPV, OP_WRAP };

template <int OP>
__device__ __forceinline__ float ew_apply(float y, float x, float a, float b) {
    if (OP == OP_AXPY) return y + a * x;          // y += a*x
    if (OP == OP_ADD) return y + x;               // a == 1 special case (utils.py:289)
    if (OP == OP_SUB) return y - x;               // a == -1 special case (utils.py:292)
    if (OP == OP_SCALE) return y * a;             // y *= a
    if (OP == OP_LINOP) return a * x + b;         // out = f1*x + f2
    if (OP == OP_LINOPV) return a * y + b * x;    // x = f1*x + f2*y   (y arg = mutable, x arg = other)
    if (OP == OP_WRAP) return pm_wrap01(y);       // periodic_wrap
    return y;
}

// out[i] = f(out_or_y[i], x[i]).  `y` is read only for ops that use it.
template <int OP, bool READ_Y, bool READ_X>
__global__ __launch_bounds__(BLK) void ew_kernel(float *__restrict__ out, const float *__restrict__ x, float a,
                                                 float b, int64_t count, int vec_ok) {
    const int64_t tid = (int64_t)blockIdx.x * BLK + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * BLK;
    if (vec_ok) {
        const int64_t n4 = count >> 2;
        float4 *o4 = reinterpret_cast<float4 *>(out);
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
        for (int64_t i = tid; i < n4; i += nthreads) {
            float4 yv = READ_Y ? o4[i] : make_float4(0, 0, 0, 0);
            float4 xv = READ_X ? x4[i] : make_float4(0, 0, 0, 0);
            float4 r;
            r.x = ew_apply<OP>(yv.x, xv.x, a, b);
            r.y = ew_apply<OP>(yv.y, xv.y, a, b);
            r.z = ew_apply<OP>(yv.z, xv.z, a, b);
            r.w = ew_apply<OP>(yv.w, xv.w, a, b);
            o4[i] = r;
        }
        for (int64_t i = (n4 << 2) + tid; i < count; i += nthreads)
            out[i] = ew_apply<OP>(READ_Y ? out[i] : 0.f, READ_X ? x[i] : 0.f, a, b);
    } else {
        for (int64_t i = tid; i < count; i += nthreads)
            out[i] = ew_apply<OP>(READ_Y ? out[i] : 0.f, READ_X ? x[i] : 0.f, a, b);
    }
}

template <int OP, bool RY, bool RX>
static int ew_launch(pm_ctx *c, float *out, const float *x, float a, float b, int64_t count) {
    if (count <= 0) return PM_OK;
    int vec_ok = pm_aligned16(out) && (!RX || pm_aligned16(x));
    int blocks = pm_stream_blocks(count, BLK * 4 * 4);
    hipLaunchKernelGGL((ew_kernel<OP, RY, RX>), dim3(blocks), dim3(BLK), 0, c->stream, out, x, a, b, count, vec_ok);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

extern "C" int pm_axpy(pm_ctx *c, float *d_y, const float *d_x, float a, int64_t count) {
    PM_REQUIRE(c && d_y && d_x && count >= 0, "bad argument");
    if (a == 1.0f) return ew_launch<OP_ADD, true, true>(c, d_y, d_x, a, 0.f, count);
    if (a == -1.0f) return ew_launch<OP_SUB, true, true>(c, d_y, d_x, a, 0.f, count);
    return ew_launch<OP_AXPY, true, true>(c, d_y, d_x, a, 0.f, count);
}

extern "C" int pm_scale(pm_ctx *c, float *d_y, float a, int64_t count) {
    PM_REQUIRE(c && d_y && count >= 0, "bad argument");
    return ew_launch<OP_SCALE, true, false>(c, d_y, nullptr, a, 0.f, count);
}

// The memory traffic of the fused kick + drift of np particles -- two [np, 3] arrays updated in place, a third one
// read, 16-byte nontemporal accesses -- with the values left as they are (x * 1): what
// main.tune_placement times to choose which allocations hold position, velocity and acceleration.  How fast the
// update runs depends on the COMBINATION of arrays, not only on each one (placement).
typedef float probe_v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(BLK) void probe_update_kernel(probe_v4f *__restrict__ a, probe_v4f *__restrict__ b,
                                                           const probe_v4f *__restrict__ c4, int64_t quads, float one) {
    for (int64_t q = (int64_t)blockIdx.x * BLK + threadIdx.x; q < quads; q += (int64_t)gridDim.x * BLK) {
        const probe_v4f x = __builtin_nontemporal_load(a + q), y = __builtin_nontemporal_load(b + q),
                        z = __builtin_nontemporal_load(c4 + q);
        const bool never = z.x != z.x && z.y == 12345.0f && z.z == 54321.0f;  // (keeps the third stream's loads alive)
        // (`one` is 1 at run time only: with a literal the compiler sees stores of the values just loaded and drops
        // them -- the first version of this probe timed three reads and no write)
        __builtin_nontemporal_store(never ? z : x * one, a + q);
        __builtin_nontemporal_store(y * one, b + q);
    }
}

extern "C" int pm_probe_update(pm_ctx *c, float *d_a, float *d_b, const float *d_c, int64_t np) {
    PM_REQUIRE(c && d_a && d_b && d_c && np >= 0 && d_a != d_b, "bad argument");
    PM_REQUIRE(pm_aligned16(d_a) && pm_aligned16(d_b) && pm_aligned16(d_c) && (np % 4) == 0,
               "pm_probe_update wants 16-byte aligned arrays of a multiple of four particles");
    if (np == 0) return PM_OK;
    const int64_t quads = 3 * np / 4;
    hipLaunchKernelGGL(probe_update_kernel, dim3(pm_stream_blocks(quads, BLK)), dim3(BLK), 0, c->stream, (probe_v4f *)d_a,
                       (probe_v4f *)d_b, (const probe_v4f *)d_c, quads, 1.0f);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

extern "C" int pm_linear_operator(pm_ctx *c, float *d_out, const float *d_x, float f1, float f2, int64_t count) {
    PM_REQUIRE(c && d_out && d_x && count >= 0, "bad argument");
    return ew_launch<OP_LINOP, false, true>(c, d_out, d_x, f1, f2, count);
}

extern "C" int pm_linear_operator_vectors(pm_ctx *c, float *d_x, float f1, const float *d_y, float f2,
                                          int64_t count) {
    PM_REQUIRE(c && d_x && d_y && count >= 0, "bad argument");
    return ew_launch<OP_LINOPV, true, true>(c, d_x, d_y, f1, f2, count);
}

extern "C" int pm_periodic_wrap(pm_ctx *c, float *d_pos, int64_t count) {
    if (c) c->lists_pos = nullptr;  // positions change: prepared tile lists (pm_tile_lists) are void
    PM_REQUIRE(c && d_pos && count >= 0, "bad argument");
    return ew_launch<OP_WRAP, true, false>(c, d_pos, nullptr, 0.f, 0.f, count);
}

// leapfrog first half (integration.py:250-258): v += -half_dt*a; x += dt*v; wrap(x).  One pass over
// the three particle arrays instead of the reference's three (axpy, axpy, wrap).
// DETECT (slab decomposition): particles whose new x leaves [x_lo, x_hi) are appended to a list
// (unordered; the host sorts the few entries) so that migration never scans the arrays again.
template <bool DETECT>
__global__ __launch_bounds__(BLK) void kick_drift_wrap_kernel(float *__restrict__ pos, float *__restrict__ vel,
                                                              const float *__restrict__ acc, float mhalf_dt, float dt,
                                                              int64_t count, int vec_ok, float x_lo, float x_hi,
                                                              int64_t *__restrict__ leave_idx, int cap,
                                                              int *__restrict__ leave_count) {
    const int64_t tid = (int64_t)blockIdx.x * BLK + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * BLK;
    auto one = [&](float &x, float &v, float a, int64_t flat) {
        pm_kick_drift_wrap_one(x, v, a, mhalf_dt, dt);
        if (DETECT && (flat % 3) == 0 && (x < x_lo || x >= x_hi)) {
            const int slot = atomicAdd(leave_count, 1);
            if (slot < cap) leave_idx[slot] = flat / 3;
        }
    };
    int64_t start = 0;
    if (vec_ok) {
        const int64_t n4 = count >> 2;
        float4 *p4 = reinterpret_cast<float4 *>(pos);
        float4 *v4 = reinterpret_cast<float4 *>(vel);
        const float4 *a4 = reinterpret_cast<const float4 *>(acc);
        for (int64_t i = tid; i < n4; i += nthreads) {
            float4 x = p4[i], v = v4[i], a = a4[i];
            one(x.x, v.x, a.x, 4 * i);
            one(x.y, v.y, a.y, 4 * i + 1);
            one(x.z, v.z, a.z, 4 * i + 2);
            one(x.w, v.w, a.w, 4 * i + 3);
            v4[i] = v;
            p4[i] = x;
        }
        start = n4 << 2;
    }
    for (int64_t i = start + tid; i < count; i += nthreads) {
        float x = pos[i], v = vel[i];
        one(x, v, acc[i], i);
        vel[i] = v;
        pos[i] = x;
    }
}

extern "C" int pm_kick_drift_wrap(pm_ctx *c, float *d_pos, float *d_vel, const float *d_acc, float half_dt, float dt,
                                  int64_t np) {
    if (c) c->lists_pos = nullptr;  // positions change: prepared tile lists (pm_tile_lists) are void
    PM_REQUIRE(c && d_pos && d_vel && d_acc && np >= 0, "bad argument");
    if (np == 0) return PM_OK;
    // (the reference's a == +-1 special cases, utils.py:289-295, are arithmetically identical to y + a*x)
    const int64_t count = 3 * np;
    int vec_ok = pm_aligned16(d_pos) && pm_aligned16(d_vel) && pm_aligned16(d_acc);
    int blocks = pm_stream_blocks(count, BLK * 4 * 2);
    hipLaunchKernelGGL((kick_drift_wrap_kernel<false>), dim3(blocks), dim3(BLK), 0, c->stream, d_pos, d_vel, d_acc,
                       -half_dt, dt, count, vec_ok, 0.f, 1.f, (int64_t *)nullptr, 0, (int *)nullptr);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

extern "C" int pm_kick_drift_wrap_slab(pm_ctx *c, float *d_pos, float *d_vel, const float *d_acc, float half_dt,
                                       float dt, int64_t np, float x_lo, float x_hi, int64_t *d_leave_idx, int cap,
                                       int *h_count) {
    if (c) c->lists_pos = nullptr;  // positions change: prepared tile lists (pm_tile_lists) are void
    PM_REQUIRE(c && d_leave_idx && h_count && cap > 0 && np >= 0 && (np == 0 || (d_pos && d_vel && d_acc)),
               "bad argument");
    int *d_count = reinterpret_cast<int *>(c->d_scratch + 16);
    PM_HIP(hipMemsetAsync(d_count, 0, sizeof(int), c->stream));
    if (np > 0) {
        const int64_t count = 3 * np;
        int vec_ok = pm_aligned16(d_pos) && pm_aligned16(d_vel) && pm_aligned16(d_acc);
        int blocks = pm_stream_blocks(count, BLK * 4 * 2);
        hipLaunchKernelGGL((kick_drift_wrap_kernel<true>), dim3(blocks), dim3(BLK), 0, c->stream, d_pos, d_vel, d_acc,
                           -half_dt, dt, count, vec_ok, x_lo, x_hi, d_leave_idx, cap, d_count);
        PM_LAUNCH_CHECK();
    }
    if (int rb_rc = pm_read_back(c, h_count, d_count, sizeof(int))) return rb_rc;
    return PM_OK;
}

// ------------------------------------------------------------------------------------------------
// max_abs (utils.py:220): the bit pattern of |x| orders like an unsigned int, and NaN (0x7fc00000)
// sorts above +inf, so an integer max reproduces np.max(np.abs(x)) including NaN propagation.

__global__ __launch_bounds__(BLK) void max_abs_kernel(const float *__restrict__ x, int64_t count, int vec_ok,
                                                      unsigned *__restrict__ result) {
    const int64_t tid = (int64_t)blockIdx.x * BLK + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * BLK;
    unsigned m = 0;
    int64_t start = 0;
    if (vec_ok) {
        const int64_t n4 = count >> 2;
        const uint4 *x4 = reinterpret_cast<const uint4 *>(x);
        for (int64_t i = tid; i < n4; i += nthreads) {
            uint4 v = x4[i];
            m = max(m, max(max(v.x & 0x7fffffffu, v.y & 0x7fffffffu), max(v.z & 0x7fffffffu, v.w & 0x7fffffffu)));
        }
        start = n4 << 2;
    }
    for (int64_t i = start + tid; i < count; i += nthreads) m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_down((int)m, off, 64));
    __shared__ unsigned wmax[BLK / 64];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < BLK / 64; ++w) m = max(m, wmax[w]);
        atomicMax(result, m);
    }
}

extern "C" int pm_max_abs(pm_ctx *c, const float *d_x, int64_t count, float *h_out) {
    PM_REQUIRE(c && d_x && h_out && count > 0, "bad argument");
    PM_HIP(hipMemsetAsync(c->d_scratch, 0, sizeof(unsigned), c->stream));
    int blocks = pm_stream_blocks(count, BLK * 4 * 8);
    hipLaunchKernelGGL(max_abs_kernel, dim3(blocks), dim3(BLK), 0, c->stream, d_x, count, (int)pm_aligned16(d_x),
                       c->d_scratch);
    PM_LAUNCH_CHECK();
    unsigned bits = 0;
    if (int rb_rc = pm_read_back(c, &bits, c->d_scratch, sizeof(unsigned))) return rb_rc;
    memcpy(h_out, &bits, sizeof(float));
    return PM_OK;
}

// ------------------------------------------------------------------------------------------------
// gradients (mesh.py:639-850 and the f(R) fused variants :860-2069)
//
// Thread (tx) owns 4 consecutive k-cells of one (i,j) row: the x/y neighbours are aligned float4
// loads, the z neighbours come from the three float4 at k-4, k, k+4 (periodic), and the AoS
// [n,n,n,3] output is three contiguous float4 stores (48 B) per thread.  Consecutive threads walk k
// then j, so every wave reads and writes whole 1-3 KiB segments.

// Slab form: phi (and u) hold nxp planes starting at global plane x0p, the output holds nxf planes
// starting at x0f; plane lookups go through the periodic origin (single GPU: nxp = nxf = n, 0, 0).
template <int ORDER, int FRN>
__global__ __launch_bounds__(BLK) void gradient_kernel(const float *__restrict__ phi, const float *__restrict__ u,
                                                       float f, float scale, int n, int log2n, int nxp, int x0p,
                                                       int nxf, int x0f, float *__restrict__ out) {
    constexpr int R = (ORDER == 2) ? 1 : (ORDER == 3 ? 1 : (ORDER == 5 ? 2 : 3));  // stencil reach
    const int nq = n >> 2;                                                        // float4 per row
    const int64_t total = (int64_t)nxf * n * nq;
    const int mask = n - 1;
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLK) {
        const int kq = (int)(t % nq);
        const int64_t ij = t / nq;  // local output plane * n + j
        const int j = (int)(ij & mask);
        const int i = x0f + (int)(ij >> log2n);  // global plane index (may exceed n: wrapped below)
        const int k0 = kq << 2;
        auto row = [&](const float *g, int ii, int jj) -> const float * {
            int li = (ii - x0p) & mask;
            li = li < nxp ? li : nxp - 1;
            return g + (((int64_t)li << log2n) + (jj & mask)) * n;
        };
        // gather lines: x[-R..R], y[-R..R] as float4 at k0; z as 12 floats k0-4..k0+7
        float ax[7][4], ay[7][4], az[12];
        float bx[7][4], by[7][4], bz[12];
#pragma unroll
        for (int o = -R; o <= R; ++o) {
            if (ORDER == 2 && o < 0) continue;
            float4 vx = *reinterpret_cast<const float4 *>(row(phi, i + o, j) + k0);
            float4 vy = *reinterpret_cast<const float4 *>(row(phi, i, j + o) + k0);
            ax[o + 3][0] = vx.x, ax[o + 3][1] = vx.y, ax[o + 3][2] = vx.z, ax[o + 3][3] = vx.w;
            ay[o + 3][0] = vy.x, ay[o + 3][1] = vy.y, ay[o + 3][2] = vy.z, ay[o + 3][3] = vy.w;
            if (FRN > 0) {
                float4 wx = *reinterpret_cast<const float4 *>(row(u, i + o, j) + k0);
                float4 wy = *reinterpret_cast<const float4 *>(row(u, i, j + o) + k0);
                bx[o + 3][0] = wx.x, bx[o + 3][1] = wx.y, bx[o + 3][2] = wx.z, bx[o + 3][3] = wx.w;
                by[o + 3][0] = wy.x, by[o + 3][1] = wy.y, by[o + 3][2] = wy.z, by[o + 3][3] = wy.w;
            }
        }
        {
            const float *r = row(phi, i, j);
            float4 zl = *reinterpret_cast<const float4 *>(r + ((k0 - 4) & mask));
            float4 zr = *reinterpret_cast<const float4 *>(r + ((k0 + 4) & mask));
            az[0] = zl.x, az[1] = zl.y, az[2] = zl.z, az[3] = zl.w;
            az[4] = ax[3][0], az[5] = ax[3][1], az[6] = ax[3][2], az[7] = ax[3][3];
            az[8] = zr.x, az[9] = zr.y, az[10] = zr.z, az[11] = zr.w;
            if (FRN > 0) {
                const float *ru = row(u, i, j);
                float4 wl = *reinterpret_cast<const float4 *>(ru + ((k0 - 4) & mask));
                float4 wr = *reinterpret_cast<const float4 *>(ru + ((k0 + 4) & mask));
                bz[0] = wl.x, bz[1] = wl.y, bz[2] = wl.z, bz[3] = wl.w;
                bz[4] = bx[3][0], bz[5] = bx[3][1], bz[6] = bx[3][2], bz[7] = bx[3][3];
                bz[8] = wr.x, bz[9] = wr.y, bz[10] = wr.z, bz[11] = wr.w;
            }
        }
        float res[12];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float sa[7], sb[7];
#pragma unroll
            for (int o = 0; o < 7; ++o) sa[o] = ax[o][c], sb[o] = (FRN > 0) ? bx[o][c] : 0.f;
            res[3 * c + 0] = Stencil<ORDER, FRN>::eval(sa, sb, f, scale);
#pragma unroll
            for (int o = 0; o < 7; ++o) sa[o] = ay[o][c], sb[o] = (FRN > 0) ? by[o][c] : 0.f;
            res[3 * c + 1] = Stencil<ORDER, FRN>::eval(sa, sb, f, scale);
#pragma unroll
            for (int o = 0; o < 7; ++o) sa[o] = az[4 + c + o - 3], sb[o] = (FRN > 0) ? bz[4 + c + o - 3] : 0.f;
            res[3 * c + 2] = Stencil<ORDER, FRN>::eval(sa, sb, f, scale);
        }
        float4 *o4 = reinterpret_cast<float4 *>(out + 3 * ((ij << log2n) + k0));
        o4[0] = make_float4(res[0], res[1], res[2], res[3]);
        o4[1] = make_float4(res[4], res[5], res[6], res[7]);
        o4[2] = make_float4(res[8], res[9], res[10], res[11]);
    }
}

// Marching form of the plain gradient (single-GPU layout): a thread owns one column of 4 k-cells
// and walks SEG consecutive x-planes with the 2R+1 planes of its x-stencil in registers, so phi
// comes from HBM once (plus 2R/SEG halo) instead of being re-fetched by 2R+1 different workgroups;
// the y/z neighbours are L1/L2 hits inside the plane.  Same Stencil::eval, same bits.
template <int ORDER, int FRN>
__global__ __launch_bounds__(BLK) void gradient_march_kernel(const float *__restrict__ phi, const float *__restrict__ u,
                                                             float f, float scale, int n, int log2n, int seg,
                                                             float *__restrict__ out) {
    constexpr int R = (ORDER == 2) ? 1 : (ORDER == 3 ? 1 : (ORDER == 5 ? 2 : 3));
    // a workgroup owns a patch of 8 rows x 32 float4 (128 k-cells), so most y/z neighbours are
    // its own threads' loads (L1 hits) rather than lines another XCD's L2 has to fetch again
    const int nq = n >> 2, mask = n - 1;
    const int pk = nq >> 5, pj = n >> 3;  // patches along k and j (n >= 128) -- see launcher for small n
    const int patch = (int)(blockIdx.x % (pk * pj));
    const int i0 = (int)(blockIdx.x / (pk * pj)) * seg;
    const int kq = (patch % pk) * 32 + (threadIdx.x & 31), j = (patch / pk) * 8 + (threadIdx.x >> 5), k0 = kq << 2;
    auto ld = [&](const float *g, int ii, int jj, int kk) -> float4 {
        return *reinterpret_cast<const float4 *>(g + ((((int64_t)(ii & mask) << log2n) + (jj & mask)) << log2n) + (kk & mask));
    };
    auto put = [](float *d, float4 v) { d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w; };
    float4 w[2 * R + 1], wu[FRN > 0 ? 2 * R + 1 : 1];
#pragma unroll
    for (int o = 0; o < 2 * R; ++o) {  // planes i0-R .. i0+R-1 (slot 0 filled below)
        w[o + 1] = ld(phi, i0 - R + o, j, k0);
        if (FRN > 0) wu[o + 1] = ld(u, i0 - R + o, j, k0);
    }
#pragma unroll 1
    for (int i = i0; i < i0 + seg; ++i) {
#pragma unroll
        for (int o = 0; o < 2 * R; ++o) {
            w[o] = w[o + 1];
            if (FRN > 0) wu[o] = wu[o + 1];
        }
        w[2 * R] = ld(phi, i + R, j, k0);
        if (FRN > 0) wu[2 * R] = ld(u, i + R, j, k0);
        float ax[7][4], ay[7][4], az[12];
        float bx[7][4], by[7][4], bz[12];
#pragma unroll
        for (int o = -R; o <= R; ++o) {
            put(ax[o + 3], w[o + R]);
            if (FRN > 0) put(bx[o + 3], wu[o + R]);
            if (ORDER == 2 && o < 0) continue;
            put(ay[o + 3], (o == 0) ? w[R] : ld(phi, i, j + o, k0));
            if (FRN > 0) put(by[o + 3], (o == 0) ? wu[R] : ld(u, i, j + o, k0));
        }
        put(az, ld(phi, i, j, k0 - 4)), put(az + 8, ld(phi, i, j, k0 + 4));
        az[4] = ax[3][0], az[5] = ax[3][1], az[6] = ax[3][2], az[7] = ax[3][3];
        if (FRN > 0) {
            put(bz, ld(u, i, j, k0 - 4)), put(bz + 8, ld(u, i, j, k0 + 4));
            bz[4] = bx[3][0], bz[5] = bx[3][1], bz[6] = bx[3][2], bz[7] = bx[3][3];
        }
        float res[12];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float sa[7], sb[7];
#pragma unroll
            for (int o = 0; o < 7; ++o) sa[o] = ax[o][c], sb[o] = (FRN > 0) ? bx[o][c] : 0.f;
            res[3 * c + 0] = Stencil<ORDER, FRN>::eval(sa, sb, f, scale);
#pragma unroll
            for (int o = 0; o < 7; ++o) sa[o] = ay[o][c], sb[o] = (FRN > 0) ? by[o][c] : 0.f;
            res[3 * c + 1] = Stencil<ORDER, FRN>::eval(sa, sb, f, scale);
#pragma unroll
            for (int o = 0; o < 7; ++o) sa[o] = az[4 + c + o - 3], sb[o] = (FRN > 0) ? bz[4 + c + o - 3] : 0.f;
            res[3 * c + 2] = Stencil<ORDER, FRN>::eval(sa, sb, f, scale);
        }
        // (the force grid is read once, by the gather's tile fills: nontemporal)
        probe_v4f *o4 = reinterpret_cast<probe_v4f *>(out + 3 * (((((int64_t)i << log2n) + j) << log2n) + k0));
        __builtin_nontemporal_store(probe_v4f{res[0], res[1], res[2], res[3]}, o4);
        __builtin_nontemporal_store(probe_v4f{res[4], res[5], res[6], res[7]}, o4 + 1);
        __builtin_nontemporal_store(probe_v4f{res[8], res[9], res[10], res[11]}, o4 + 2);
    }
}

template <int ORDER, int FRN = 0>
static int gradient_march_launch(pm_ctx *c, float scale, const float *phi, float *out, const float *u = nullptr, float f = 0.f) {
    const int n = c->n, seg = 32;
    const int64_t blocks = (int64_t)((n >> 2) >> 5) * (n >> 3) * (n / seg);
    hipLaunchKernelGGL((gradient_march_kernel<ORDER, FRN>), dim3((unsigned)blocks), dim3(BLK), 0, c->stream, phi, u, f, scale, n,
                       c->log2n, seg, out);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

template <int FRN>
static int gradient_dispatch(pm_ctx *c, int order, float f, const float *phi, const float *u, float *out, int nxp,
                             int x0p, int nxf, int x0f) {
    const int n = c->n;
    const int64_t total = (int64_t)nxf * n * (n >> 2);
    const int blocks = pm_stream_blocks(total, BLK);
    float scale;
    switch (order) {
        case 2:
            scale = (float)n;  // invh (mesh.py:656)
            hipLaunchKernelGGL((gradient_kernel<2, FRN>), dim3(blocks), dim3(BLK), 0, c->stream, phi, u, f, scale, n,
                               c->log2n, nxp, x0p, nxf, x0f, out);
            break;
        case 3:
            scale = (float)(0.5 * n);  // inv2h (mesh.py:698)
            hipLaunchKernelGGL((gradient_kernel<3, FRN>), dim3(blocks), dim3(BLK), 0, c->stream, phi, u, f, scale, n,
                               c->log2n, nxp, x0p, nxf, x0f, out);
            break;
        case 5:
            scale = (float)(n / 12.0);  // inv12h (mesh.py:743)
            hipLaunchKernelGGL((gradient_kernel<5, FRN>), dim3(blocks), dim3(BLK), 0, c->stream, phi, u, f, scale, n,
                               c->log2n, nxp, x0p, nxf, x0f, out);
            break;
        case 7:
            scale = (float)(n / 60.0);  // inv60h (mesh.py:801)
            hipLaunchKernelGGL((gradient_kernel<7, FRN>), dim3(blocks), dim3(BLK), 0, c->stream, phi, u, f, scale, n,
                               c->log2n, nxp, x0p, nxf, x0f, out);
            break;
        default:
            pm_set_error("gradient order %d, should be 2, 3, 5 or 7", order);
            return PM_EINVAL;
    }
    PM_LAUNCH_CHECK();
    return PM_OK;
}

extern "C" int pm_gradient(pm_ctx *c, int order, const float *d_phi, float *d_force) {
    PM_REQUIRE(c && d_phi && d_force, "NULL argument");
    PM_REQUIRE(pm_aligned16(d_phi) && pm_aligned16(d_force), "grids must be 16-byte aligned");
    PM_REQUIRE(c->n >= 8, "gradient kernels need n >= 8");
    if (c->n >= 128) {
        const int n = c->n;
        switch (order) {
            case 2: return gradient_march_launch<2>(c, (float)n, d_phi, d_force);
            case 3: return gradient_march_launch<3>(c, (float)(0.5 * n), d_phi, d_force);
            case 5: return gradient_march_launch<5>(c, (float)(n / 12.0), d_phi, d_force);
            case 7: return gradient_march_launch<7>(c, (float)(n / 60.0), d_phi, d_force);
        }
    }
    return gradient_dispatch<0>(c, order, 0.f, d_phi, nullptr, d_force, c->n, 0, c->n, 0);
}

extern "C" int pm_gradient_slab(pm_ctx *c, int order, const float *d_phi, int nx_phi, int x0_phi, float *d_force,
                                int nx_force, int x0_force) {
    PM_REQUIRE(c && d_phi && d_force, "NULL argument");
    PM_REQUIRE(pm_aligned16(d_phi) && pm_aligned16(d_force), "grids must be 16-byte aligned");
    PM_REQUIRE(c->n >= 8 && nx_phi >= 1 && nx_force >= 1, "bad slab geometry");
    const int n = c->n;
    return gradient_dispatch<0>(c, order, 0.f, d_phi, nullptr, d_force, nx_phi, ((x0_phi % n) + n) % n, nx_force,
                                ((x0_force % n) + n) % n);
}

// f(R) force on a slab: d_phi and d_u both hold nx_phi planes from global plane x0_phi (reach + 1 ghosts)
extern "C" int pm_gradient_fr_slab(pm_ctx *c, int order, int fr_n, float f, const float *d_phi, const float *d_u,
                                   int nx_phi, int x0_phi, float *d_force, int nx_force, int x0_force) {
    PM_REQUIRE(c && d_phi && d_u && d_force, "NULL argument");
    PM_REQUIRE(pm_aligned16(d_phi) && pm_aligned16(d_u) && pm_aligned16(d_force), "grids must be 16-byte aligned");
    PM_REQUIRE(c->n >= 8 && nx_phi >= 1 && nx_force >= 1, "bad slab geometry");
    const int n = c->n, x0p = ((x0_phi % n) + n) % n, x0f = ((x0_force % n) + n) % n;
    if (fr_n == 1) return gradient_dispatch<1>(c, order, f, d_phi, d_u, d_force, nx_phi, x0p, nx_force, x0f);
    if (fr_n == 2) return gradient_dispatch<2>(c, order, f, d_phi, d_u, d_force, nx_phi, x0p, nx_force, x0f);
    pm_set_error("Only f(R) with n = 1 and 2, currently fR_n=%d", fr_n);
    return PM_ENOSUP;
}

extern "C" int pm_gradient_fr(pm_ctx *c, int order, int fr_n, float f, const float *d_phi, const float *d_u,
                              float *d_force) {
    PM_REQUIRE(c && d_phi && d_u && d_force, "NULL argument");
    PM_REQUIRE(pm_aligned16(d_phi) && pm_aligned16(d_u) && pm_aligned16(d_force), "grids must be 16-byte aligned");
    PM_REQUIRE(c->n >= 8, "gradient kernels need n >= 8");
    if (c->n >= 128 && c->fr_march_gradient && (fr_n == 1 || fr_n == 2)) {  // marching form, as pm_gradient (same bits)
        const int n = c->n;
#define PM_FR_MARCH(O, SC)                                                                  \
    return fr_n == 1 ? gradient_march_launch<O, 1>(c, (float)(SC), d_phi, d_force, d_u, f) \
                     : gradient_march_launch<O, 2>(c, (float)(SC), d_phi, d_force, d_u, f)
        switch (order) {
            case 2: PM_FR_MARCH(2, n);
            case 3: PM_FR_MARCH(3, 0.5 * n);
            case 5: PM_FR_MARCH(5, n / 12.0);
            case 7: PM_FR_MARCH(7, n / 60.0);
        }
#undef PM_FR_MARCH
    }
    if (fr_n == 1) return gradient_dispatch<1>(c, order, f, d_phi, d_u, d_force, c->n, 0, c->n, 0);
    if (fr_n == 2) return gradient_dispatch<2>(c, order, f, d_phi, d_u, d_force, c->n, 0, c->n, 0);
    pm_set_error("Only f(R) with n = 1 and 2, currently fR_n=%d", fr_n);
    return PM_ENOSUP;
}

// off: pm_gradient_fr always takes the row kernel (tests compare the two)
extern "C" int pm_set_fr_march_gradient(pm_ctx *c, int on) {
    PM_REQUIRE(c, "NULL argument");
    c->fr_march_gradient = on ? 1 : 0;
    return PM_OK;
}

// ------------------------------------------------------------------------------------------------
// k-space Green's functions (fourier.py:460-595): one thread per complex element, per-axis tables.

// The array holds x[i, j0:j0+nj, k] (nj == n, j0 == 0 for the single-GPU layout; a y-range in the
// transposed slab layout).
template <int KIND>
__global__ __launch_bounds__(BLK) void greens_kernel(float2 *__restrict__ x, int n, int j0, int nj, int p,
                                                     const float *__restrict__ k2, const float *__restrict__ s,
                                                     float c0) {
    const int nh = (n >> 1) + 1;
    const int64_t total = (int64_t)n * nj * nh;
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLK) {
        const int k = (int)(t % nh);
        const int64_t ij = t / nh;
        const int j = j0 + (int)(ij % nj);
        const int i = (int)(ij / nj);
        float g;
        if (KIND == PM_GREEN_7PT) {
            g = c0 / ((s[i] + s[j]) + s[k]);
        } else {
            const float invk2 = c0 / ((k2[i] + k2[j]) + k2[k]);
            if (KIND == PM_GREEN_COMPENSATED) {
                const float w = (s[i] * s[j]) * s[k];
                g = powf(w, (float)(-2 * p)) * invk2;
            } else {
                g = invk2;
            }
        }
        float2 v = x[t];
        if (i == 0 && j == 0 && k == 0) {
            v = make_float2(0.f, 0.f);  // x[0,0,0] = 0 (fourier.py:491, 544, 595)
        } else {
            v.x *= g;
            v.y *= g;
        }
        x[t] = v;
    }
}

extern "C" int pm_greens(pm_ctx *c, int kind, int p, void *d_xk, const float *h_k2, const float *h_s, float c0) {
    return pm_greens_slab(c, kind, p, d_xk, h_k2, h_s, c0, 0, c ? c->n : 0);
}

extern "C" int pm_greens_slab(pm_ctx *c, int kind, int p, void *d_xk, const float *h_k2, const float *h_s, float c0,
                              int j0, int nj) {
    PM_REQUIRE(c && d_xk && h_k2 && h_s, "NULL argument");
    const int n = c->n;
    PM_REQUIRE(j0 >= 0 && nj >= 1 && j0 + nj <= n, "bad y range");
    // tables are tiny (n floats): staged through pageable memcpy on the stream
    PM_HIP(hipMemcpyAsync(c->d_tab, h_k2, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    PM_HIP(hipMemcpyAsync(c->d_tab + n, h_s, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    const int64_t total = (int64_t)n * nj * (n / 2 + 1);
    const int blocks = pm_stream_blocks(total, BLK * 2);
    float2 *x = reinterpret_cast<float2 *>(d_xk);
    switch (kind) {
        case PM_GREEN_PLAIN:
            hipLaunchKernelGGL((greens_kernel<PM_GREEN_PLAIN>), dim3(blocks), dim3(BLK), 0, c->stream, x, n, j0, nj, p,
                               c->d_tab, c->d_tab + n, c0);
            break;
        case PM_GREEN_COMPENSATED:
            hipLaunchKernelGGL((greens_kernel<PM_GREEN_COMPENSATED>), dim3(blocks), dim3(BLK), 0, c->stream, x, n, j0,
                               nj, p, c->d_tab, c->d_tab + n, c0);
            break;
        case PM_GREEN_7PT:
            hipLaunchKernelGGL((greens_kernel<PM_GREEN_7PT>), dim3(blocks), dim3(BLK), 0, c->stream, x, n, j0, nj, p,
                               c->d_tab, c->d_tab + n, c0);
            break;
        default:
            pm_set_error("unknown Green's function kind %d", kind);
            return PM_EINVAL;
    }
    PM_LAUNCH_CHECK();
    return PM_OK;
}

// ------------------------------------------------------------------------------------------------
// Force in k-space (fourier.py:606-719): out_m = -i k_m / (2 pi k^2) [W^(-2p)] x, the `full_fft` solver's
// gradient of the inverse Laplacian.  Planar output (three half-spectra), each inverse-transformed on its own.
// (0 - i c)(a + i b) = c b - i c a exactly, then each component times k_m: the reference's rounding.

template <bool COMP>
__global__ __launch_bounds__(BLK) void grad_greens_kernel(const float2 *__restrict__ x, int n, int p,
                                                          const float *__restrict__ kf, const float *__restrict__ s,
                                                          float2 *__restrict__ out) {
    const int nh = (n >> 1) + 1;
    const int64_t total = (int64_t)n * n * nh;
    const float invtwopi = (float)(0.5 / 3.14159265358979323846);
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLK) {
        const int k = (int)(t % nh);
        const int64_t ij = t / nh;
        const int j = (int)(ij % n), i = (int)(ij / n);
        const float kx = kf[i], ky = kf[j], kz = (float)k;
        float c = invtwopi / ((kx * kx + ky * ky) + kz * kz);
        if (COMP) {
            const float w = (s[i] * s[j]) * s[k];
            c = powf(w, (float)(-2 * p)) * c;
        }
        const float2 v = x[t];
        float re = c * v.y, im = -(c * v.x);
        if (t == 0) re = 0.f, im = 0.f;  // result[0, 0, 0, :] = 0 (fourier.py:656, 718)
        out[t] = make_float2(re * kx, im * kx);
        out[total + t] = make_float2(re * ky, im * ky);
        out[2 * total + t] = make_float2(re * kz, im * kz);
    }
}

extern "C" int pm_grad_inverse_laplacian(pm_ctx *c, int p, const void *d_xk, void *d_out3, const float *h_kfreq,
                                         const float *h_sinc) {
    PM_REQUIRE(c && d_xk && d_out3 && h_kfreq && h_sinc && p >= 0, "bad argument");
    const int n = c->n;
    PM_HIP(hipMemcpyAsync(c->d_tab, h_kfreq, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    PM_HIP(hipMemcpyAsync(c->d_tab + n, h_sinc, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    const int64_t total = (int64_t)n * n * (n / 2 + 1);
    const int blocks = pm_stream_blocks(total, BLK * 2);
    const float2 *x = reinterpret_cast<const float2 *>(d_xk);
    float2 *out = reinterpret_cast<float2 *>(d_out3);
    if (p > 0)
        hipLaunchKernelGGL((grad_greens_kernel<true>), dim3(blocks), dim3(BLK), 0, c->stream, x, n, p, c->d_tab, c->d_tab + n, out);
    else
        hipLaunchKernelGGL((grad_greens_kernel<false>), dim3(blocks), dim3(BLK), 0, c->stream, x, n, p, c->d_tab, c->d_tab + n, out);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

// ------------------------------------------------------------------------------------------------
// k-space operators of the initial conditions (fourier.py:730-905): gradient i 2 pi k_m x (three planar half-spectra)
// and -(k_a1 k_b1 + k_a2 k_b2) 4 pi^2 x (hessian: second pair off; sum_of_hessian: both), products in the
// reference's order.

__global__ __launch_bounds__(BLK) void kspace_gradient_kernel(const float2 *__restrict__ x, int n,
                                                              const float *__restrict__ kf, float2 *__restrict__ out) {
    const int nh = (n >> 1) + 1;
    const int64_t total = (int64_t)n * n * nh;
    const float twopi = (float)(2.0 * 3.14159265358979323846);
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLK) {
        const int k = (int)(t % nh);
        const int64_t ij = t / nh;
        const float kx = kf[(int)(ij / n)], ky = kf[(int)(ij % n)], kz = (float)k;
        const float2 v = x[t];
        const float re = -(twopi * v.y), im = twopi * v.x;  // (0 + i 2 pi)(a + i b)
        out[t] = make_float2(re * kx, im * kx);
        out[total + t] = make_float2(re * ky, im * ky);
        out[2 * total + t] = make_float2(re * kz, im * kz);
    }
}

__global__ __launch_bounds__(BLK) void kspace_hessian_kernel(const float2 *__restrict__ x, int n,
                                                             const float *__restrict__ kf, int a1, int b1, int a2, int b2,
                                                             float2 *__restrict__ out) {
    const int nh = (n >> 1) + 1;
    const int64_t total = (int64_t)n * n * nh;
    const float fourpi2 = (float)(4.0 * 3.14159265358979323846 * 3.14159265358979323846);
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLK) {
        const int k = (int)(t % nh);
        const int64_t ij = t / nh;
        const float kk[3] = {kf[(int)(ij / n)], kf[(int)(ij % n)], (float)k};
        float c;
        if (a2 < 0)
            c = (-kk[a1] * kk[b1]) * fourpi2;  // hessian: -kn * km * fourpi2
        else if (a2 >= 4)
            c = -(kk[a1] * kk[b1] - kk[a2 - 4] * kk[b2 - 4]) * fourpi2;  // diff_of_hessian (fourier.py:907)
        else
            c = -(kk[a1] * kk[b1] + kk[a2] * kk[b2]) * fourpi2;  // sum_of_hessian
        const float2 v = x[t];
        out[t] = make_float2(c * v.x, c * v.y);
    }
}

extern "C" int pm_kspace_gradient(pm_ctx *c, const void *d_xk, void *d_out3, const float *h_kfreq) {
    PM_REQUIRE(c && d_xk && d_out3 && h_kfreq, "NULL argument");
    const int n = c->n;
    PM_HIP(hipMemcpyAsync(c->d_tab, h_kfreq, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    const int64_t total = (int64_t)n * n * (n / 2 + 1);
    hipLaunchKernelGGL(kspace_gradient_kernel, dim3(pm_stream_blocks(total, BLK * 2)), dim3(BLK), 0, c->stream,
                       reinterpret_cast<const float2 *>(d_xk), n, c->d_tab, reinterpret_cast<float2 *>(d_out3));
    PM_LAUNCH_CHECK();
    return PM_OK;
}

extern "C" int pm_kspace_hessian(pm_ctx *c, const void *d_xk, void *d_out, const float *h_kfreq, int a1, int b1, int a2,
                                 int b2) {
    PM_REQUIRE(c && d_xk && d_out && h_kfreq, "NULL argument");
    PM_REQUIRE(a1 >= 0 && a1 < 3 && b1 >= 0 && b1 < 3 && a2 < 7 && a2 != 3 && b2 < 7 && b2 != 3 && (a2 < 0) == (b2 < 0) &&
                   (a2 >= 4) == (b2 >= 4),
               "axes are 0, 1, 2 (second pair -1, -1: off; 4 + axis, 4 + axis: subtracted)");
    const int n = c->n;
    PM_HIP(hipMemcpyAsync(c->d_tab, h_kfreq, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    const int64_t total = (int64_t)n * n * (n / 2 + 1);
    hipLaunchKernelGGL(kspace_hessian_kernel, dim3(pm_stream_blocks(total, BLK * 2)), dim3(BLK), 0, c->stream,
                       reinterpret_cast<const float2 *>(d_xk), n, c->d_tab, a1, b1, a2, b2, reinterpret_cast<float2 *>(d_out));
    PM_LAUNCH_CHECK();
    return PM_OK;
}

// The same on the padded grid of the de-aliased initial conditions: m = 3 n / 2 (h_kfreq has m entries; the table of
// the context holds 3 n floats)
extern "C" int pm_kspace_hessian_padded(pm_ctx *c, const void *d_xk, void *d_out, const float *h_kfreq, int a1, int b1,
                                        int a2, int b2) {
    PM_REQUIRE(c && d_xk && d_out && h_kfreq, "NULL argument");
    PM_REQUIRE(a1 >= 0 && a1 < 3 && b1 >= 0 && b1 < 3 && a2 < 7 && a2 != 3 && b2 < 7 && b2 != 3 && (a2 < 0) == (b2 < 0) &&
                   (a2 >= 4) == (b2 >= 4),
               "axes are 0, 1, 2 (second pair -1, -1: off; 4 + axis, 4 + axis: subtracted)");
    const int m = c->n * 3 / 2;
    PM_HIP(hipMemcpyAsync(c->d_tab, h_kfreq, m * sizeof(float), hipMemcpyHostToDevice, c->stream));
    const int64_t total = (int64_t)m * m * (m / 2 + 1);
    hipLaunchKernelGGL(kspace_hessian_kernel, dim3(pm_stream_blocks(total, BLK * 2)), dim3(BLK), 0, c->stream,
                       reinterpret_cast<const float2 *>(d_xk), m, c->d_tab, a1, b1, a2, b2, reinterpret_cast<float2 *>(d_out));
    PM_LAUNCH_CHECK();
    return PM_OK;
}

// initial_conditions.pad (:1859-1894, up = 1: [n, n, n/2+1] -> [m, m, m/2+1], m = 3 n / 2) and trim (:1897-1928,
// up = 0, the way back): zeros, then the four corner blocks of the x and y axes (indices below n/2, and the n/2 - 1
// highest ones) times the kz below n/2 -- the Nyquist planes of the small grid are dropped, both ways.
__global__ __launch_bounds__(BLK) void spectrum_blocks_kernel(const float2 *__restrict__ in, int n_in, float2 *__restrict__ out,
                                                              int n_out) {
    const int mid = min(n_in, n_out) >> 1, nh_in = (n_in >> 1) + 1, nh_out = (n_out >> 1) + 1;
    const int64_t total = (int64_t)n_out * n_out * nh_out;
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLK) {
        const int k = (int)(t % nh_out);
        const int64_t ij = t / nh_out;
        const int i = (int)(ij / n_out), j = (int)(ij % n_out);
        // low block: the same index; high block: the same distance from the end
        const int si = i < mid ? i : (i > n_out - mid ? i - n_out + n_in : -1);
        const int sj = j < mid ? j : (j > n_out - mid ? j - n_out + n_in : -1);
        float2 v = make_float2(0.f, 0.f);
        if (si >= 0 && sj >= 0 && k < mid) v = in[((int64_t)si * n_in + sj) * nh_in + k];
        out[t] = v;
    }
}

extern "C" int pm_dealias_resize(pm_ctx *c, const void *d_in, void *d_out, int up) {
    PM_REQUIRE(c && d_in && d_out && d_in != d_out, "bad argument");
    const int n = c->n, m = n * 3 / 2, n_in = up ? n : m, n_out = up ? m : n;
    const int64_t total = (int64_t)n_out * n_out * (n_out / 2 + 1);
    hipLaunchKernelGGL(spectrum_blocks_kernel, dim3(pm_stream_blocks(total, BLK * 2)), dim3(BLK), 0, c->stream,
                       reinterpret_cast<const float2 *>(d_in), n_in, reinterpret_cast<float2 *>(d_out), n_out);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

// y = (accumulate ? y : 0) + f * a * b (utils.prod_vector_vector_inplace :500, add_vector_vector_inplace :330)
__global__ __launch_bounds__(BLK) void addcmul_kernel(float *__restrict__ y, float f, const float *__restrict__ a,
                                                      const float *__restrict__ b, int64_t count, int accumulate) {
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < count; t += (int64_t)gridDim.x * BLK) {
        const float p = f * a[t] * b[t];
        y[t] = accumulate ? y[t] + p : p;
    }
}

// y += f * a * b * c (utils.add_vector_vector_vector_inplace, utils.py:368)
__global__ __launch_bounds__(BLK) void addcmul3_kernel(float *__restrict__ y, float f, const float *__restrict__ a,
                                                       const float *__restrict__ b, const float *__restrict__ c3,
                                                       int64_t count) {
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < count; t += (int64_t)gridDim.x * BLK)
        y[t] = y[t] + f * a[t] * b[t] * c3[t];
}

extern "C" int pm_addcmul3(pm_ctx *c, float *d_y, float f, const float *d_a, const float *d_b, const float *d_c,
                           int64_t count) {
    PM_REQUIRE(c && d_y && d_a && d_b && d_c && count >= 0, "bad argument");
    if (count == 0) return PM_OK;
    hipLaunchKernelGGL(addcmul3_kernel, dim3(pm_stream_blocks(count, BLK)), dim3(BLK), 0, c->stream, d_y, f, d_a, d_b, d_c,
                       count);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

extern "C" int pm_addcmul(pm_ctx *c, float *d_y, float f, const float *d_a, const float *d_b, int64_t count, int accumulate) {
    PM_REQUIRE(c && d_y && d_a && d_b && count >= 0, "bad argument");
    if (count == 0) return PM_OK;
    hipLaunchKernelGGL(addcmul_kernel, dim3(pm_stream_blocks(count, BLK)), dim3(BLK), 0, c->stream, d_y, f, d_a, d_b, count,
                       accumulate);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

// Particles from a displacement field psi [n, n, n, 3] (initial_conditions.py:1688-1856).  mode 0 / 1: first order
// from cell centres / edges: x = q - D psi, v = -(D fH) psi; mode 2: add an n-th order term: x += D psi, v += (D fH) psi.
__global__ __launch_bounds__(BLK) void lpt_displace_kernel(const float *__restrict__ psi, int n, float dplus, float dfh,
                                                           int mode, float *__restrict__ pos, float *__restrict__ vel) {
    const int64_t total = (int64_t)n * n * n;
    const float h = 1.0f / (float)n, half_h = 0.5f / (float)n;
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLK) {
        const int k = (int)(t % n), j = (int)((t / n) % n), i = (int)(t / ((int64_t)n * n));
        const int q[3] = {i, j, k};
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const float p = psi[3 * t + m];
            if (mode == 2) {
                pos[3 * t + m] += dplus * p;
                vel[3 * t + m] += dfh * p;
            } else {
                const float x0 = mode == 0 ? half_h + (float)q[m] * h : (float)q[m] * h;
                pos[3 * t + m] = x0 + dplus * (-p);
                vel[3 * t + m] = dfh * (-p);
            }
        }
    }
}

extern "C" int pm_lpt_displace(pm_ctx *c, const float *d_psi, float dplus, float dfh, int mode, float *d_pos, float *d_vel) {
    PM_REQUIRE(c && d_psi && d_pos && d_vel && mode >= 0 && mode <= 2, "bad argument");
    const int64_t total = (int64_t)c->n * c->n * c->n;
    hipLaunchKernelGGL(lpt_displace_kernel, dim3(pm_stream_blocks(total, BLK)), dim3(BLK), 0, c->stream, d_psi, c->n, dplus, dfh,
                       mode, d_pos, d_vel);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

// three planar arrays of m elements -> one [m, 3] array (the force grid's layout)
__global__ __launch_bounds__(BLK) void interleave3_kernel(const float *__restrict__ a, int64_t m, float *__restrict__ out) {
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < m; t += (int64_t)gridDim.x * BLK) {
        out[3 * t] = a[t], out[3 * t + 1] = a[m + t], out[3 * t + 2] = a[2 * m + t];
    }
}

extern "C" int pm_interleave3(pm_ctx *c, const float *d_planar, int64_t m, float *d_out) {
    PM_REQUIRE(c && d_planar && d_out && m > 0, "bad argument");
    hipLaunchKernelGGL(interleave3_kernel, dim3(pm_stream_blocks(m, BLK)), dim3(BLK), 0, c->stream, d_planar, m, d_out);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

// ------------------------------------------------------------------------------------------------
// P(k) estimator (fourier.py:22-100).  Per-block LDS histograms in fp32 (the reference keeps one
// private fp32 histogram per i-plane), flushed to fp64 global bins.

__global__ __launch_bounds__(BLK) void pk_kernel(float2 *__restrict__ x, int n, int log2n, int p,
                                                 const float *__restrict__ sincs, double *__restrict__ hist) {
    extern __shared__ float lds[];  // 3*n floats: ksum, pksum, nmodes
    for (int b = threadIdx.x; b < 3 * n; b += BLK) lds[b] = 0.f;
    __syncthreads();
    const int nh = (n >> 1) + 1, middle = n >> 1, mask = n - 1;
    const int64_t total = (int64_t)n * n * nh;
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLK) {
        const int k = (int)(t % nh);
        const int64_t ij = t / nh;
        const int j = (int)(ij & mask);
        const int i = (int)(ij >> log2n);
        if (t == 0) {
            x[0] = make_float2(0.f, 0.f);  // side effect at fourier.py:80-82
            continue;
        }
        const float kx = (i >= middle) ? (float)(i - n) : (float)i;
        const float ky = (j >= middle) ? (float)(j - n) : (float)j;
        const float kz = (float)k;
        const float w = (sincs[i] * sincs[j]) * sincs[k];
        const float comp = powf(w, (float)(-p));
        const float2 v = x[t];
        const float re = v.x * comp, im = v.y * comp;
        const float delta2 = re * re + im * im;
        const double knorm = sqrt((double)((kx * kx + ky * ky) + kz * kz));
        const int kidx = (int)(knorm + 0.5);
        atomicAdd(&lds[kidx], (float)knorm);
        atomicAdd(&lds[n + kidx], delta2);
        atomicAdd(&lds[2 * n + kidx], 1.0f);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < 3 * n; b += BLK) {
        float v = lds[b];
        if (v != 0.f) atomicAdd(&hist[b], (double)v);
    }
}

extern "C" int pm_pk(pm_ctx *c, int p, void *d_xk, const float *h_sinc, double *h_ksum, double *h_pksum,
                     double *h_nmodes) {
    PM_REQUIRE(c && d_xk && h_sinc && h_ksum && h_pksum && h_nmodes, "NULL argument");
    const int n = c->n;
    PM_REQUIRE((size_t)3 * n * sizeof(float) <= 64 * 1024, "n too large for the LDS histogram");
    PM_HIP(hipMemcpyAsync(c->d_tab + 2 * n, h_sinc, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    PM_HIP(hipMemsetAsync(c->d_hist, 0, 3 * (size_t)n * sizeof(double), c->stream));
    const int64_t total = (int64_t)n * n * (n / 2 + 1);
    int blocks = pm_stream_blocks(total, BLK * 16);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(pk_kernel, dim3(blocks), dim3(BLK), 3 * n * sizeof(float), c->stream,
                       reinterpret_cast<float2 *>(d_xk), n, c->log2n, p, c->d_tab + 2 * n, c->d_hist);
    PM_LAUNCH_CHECK();
    PM_HIP(hipMemcpyAsync(h_ksum, c->d_hist, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PM_HIP(hipMemcpyAsync(h_pksum, c->d_hist + n, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (int rb_rc = pm_read_back(c, h_nmodes, c->d_hist + 2 * n, n * sizeof(double))) return rb_rc;
    return PM_OK;
}
