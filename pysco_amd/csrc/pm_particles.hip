// Particle <-> mesh kernels: mass deposit (mesh.py:2240-2595), inverse interpolation
// (mesh.py:2600-3088), Morton keys (morton.py:42-137) and the particle re-ordering
// (utils.py:1019-1116).
//
// Deposit, MI355X design: global fp32 atomics run at ~1.3 TB/s chip-wide and only when a wave
// instruction touches contiguous bytes, so scattering 27 atomics per particle is ~20x off the HBM
// roofline.  Instead a workgroup takes a contiguous chunk of the (Morton-sorted) particle stream,
// accumulates it into an LDS tile of 16x16x16 cells + a one-cell halo with ds_add_f32, and flushes
// the tile with one global atomic per touched cell (rows of 18 contiguous floats).  A chunk that
// spans several tiles is handled by re-anchoring the LDS tile (a few passes); particles that fall
// outside the first PM_MAX_PASSES tiles -- unsorted input -- go straight to global atomics, so
// the result never depends on the order, only the speed does.
#include "pm_common.h"

#include <rocprim/rocprim.hpp>

#define BLK 256

// ------------------------------------------------------------------------------------------------
// 1-D assignment weights, written exactly as the reference does (fp32, same operation order)

__device__ __forceinline__ int cell_of(float xp, int n) {
    int i = (int)xp;  // np.int16(x_part): truncation (mesh.py:2506)
    return min(max(i, 0), n - 1);  // memory safety for out-of-range / NaN input (reference: UB)
}

// TSC (mesh.py:2510-2520): w[0] at i-1, w[1] at i, w[2] at i+1
__device__ __forceinline__ void tsc_weights(float xp, int i, float w[3]) {
    const float half = 0.5f;
    const float d = xp - half - (float)i;
    w[1] = 0.75f - d * d;
    const float a = half - d, b = half + d;
    w[0] = half * (a * a);
    w[2] = half * (b * b);
}

// CIC (mesh.py:2312-2330): weight (1-|d|) at i, |d| at i+sign(d)
__device__ __forceinline__ void cic_weights(float xp, int i, float &w0, float &w1, int &sgn) {
    const float d = xp - 0.5f - (float)i;
    sgn = (d > 0.f) - (d < 0.f);
    w1 = fabsf(d);
    w0 = 1.0f - w1;
}

// ------------------------------------------------------------------------------------------------
// deposit, order-independent fallback: one particle per thread, global atomics

template <int SCHEME>
__device__ __forceinline__ void deposit_global(float x, float y, float z, int n, int log2n, float *rho) {
    const float nf = (float)n;
    const int mask = n - 1;
    const float xp = x * nf, yp = y * nf, zp = z * nf;
    const int i = cell_of(xp, n), j = cell_of(yp, n), k = cell_of(zp, n);
    auto at = [&](int ii, int jj, int kk) -> float * {
        return rho + ((((int64_t)(ii & mask) << log2n) + (jj & mask)) << log2n) + (kk & mask);
    };
    if (SCHEME == PM_NGP) {
        unsafeAtomicAdd(at(i, j, k), 1.0f);
    } else if (SCHEME == PM_CIC) {
        float wx[2], wy[2], wz[2];
        int sx, sy, sz;
        cic_weights(xp, i, wx[0], wx[1], sx);
        cic_weights(yp, j, wy[0], wy[1], sy);
        cic_weights(zp, k, wz[0], wz[1], sz);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    unsafeAtomicAdd(at(i + a * sx, j + b * sy, k + c * sz), wx[a] * wy[b] * wz[c]);
    } else {
        float wx[3], wy[3], wz[3];
        tsc_weights(xp, i, wx);
        tsc_weights(yp, j, wy);
        tsc_weights(zp, k, wz);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float wxy = wx[a] * wy[b];
#pragma unroll
                for (int c = 0; c < 3; ++c) unsafeAtomicAdd(at(i + a - 1, j + b - 1, k + c - 1), wxy * wz[c]);
            }
    }
}

template <int SCHEME>
__global__ __launch_bounds__(BLK) void deposit_atomic_kernel(const float *__restrict__ pos, int64_t np, int n,
                                                             int log2n, float *__restrict__ rho) {
    for (int64_t p = (int64_t)blockIdx.x * BLK + threadIdx.x; p < np; p += (int64_t)gridDim.x * BLK)
        deposit_global<SCHEME>(pos[3 * p], pos[3 * p + 1], pos[3 * p + 2], n, log2n, rho);
}

// ------------------------------------------------------------------------------------------------
// deposit, LDS-tiled (see the header comment)

constexpr int PM_T = 16;                       // tile edge in cells
constexpr int PM_TH = PM_T + 2;                // + halo
constexpr int PM_TILE = PM_TH * PM_TH * PM_TH; // 5832 floats = 23.3 KB
constexpr int PM_PPT = 8;                      // particles per thread
constexpr int PM_CHUNK = BLK * PM_PPT;         // particles per workgroup
constexpr int PM_MAX_PASSES = 6;

template <int SCHEME>
__device__ __forceinline__ void deposit_lds(float xp, float yp, float zp, int i, int j, int k, int ti, int tj, int tk,
                                            float *tile) {
    // local coordinates inside the haloed tile: cell (ti*T-1) is local 0
    const int li = i - ti * PM_T + 1, lj = j - tj * PM_T + 1, lk = k - tk * PM_T + 1;
    auto at = [&](int a, int b, int c) -> float * { return tile + ((li + a) * PM_TH + (lj + b)) * PM_TH + (lk + c); };
    if (SCHEME == PM_NGP) {
        atomicAdd(at(0, 0, 0), 1.0f);
    } else if (SCHEME == PM_CIC) {
        float wx[2], wy[2], wz[2];
        int sx, sy, sz;
        cic_weights(xp, i, wx[0], wx[1], sx);
        cic_weights(yp, j, wy[0], wy[1], sy);
        cic_weights(zp, k, wz[0], wz[1], sz);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) atomicAdd(at(a * sx, b * sy, c * sz), wx[a] * wy[b] * wz[c]);
    } else {
        float wx[3], wy[3], wz[3];
        tsc_weights(xp, i, wx);
        tsc_weights(yp, j, wy);
        tsc_weights(zp, k, wz);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float wxy = wx[a] * wy[b];
#pragma unroll
                for (int c = 0; c < 3; ++c) atomicAdd(at(a - 1, b - 1, c - 1), wxy * wz[c]);
            }
    }
}

template <int SCHEME>
__global__ __launch_bounds__(BLK) void deposit_tiled_kernel(const float *__restrict__ pos, int64_t np, int n,
                                                            int log2n, float *__restrict__ rho) {
    __shared__ float tile[PM_TILE];
    __shared__ int s_first;
    __shared__ int s_anchor;
    const int tid = threadIdx.x;
    const int mask = n - 1;
    const float nf = (float)n;
    const int tshift = log2n - 4;  // tiles per side = n / 16
    for (int c = tid; c < PM_TILE; c += BLK) tile[c] = 0.f;

    const int64_t base = (int64_t)blockIdx.x * PM_CHUNK;
    // Particles are re-read from L2 on every pass rather than held in registers: 27 weights per
    // particle already fill the VGPR budget, and a sorted chunk needs only 1-3 passes.
    unsigned todo = 0;
#pragma unroll
    for (int r = 0; r < PM_PPT; ++r)
        if (base + r * BLK + tid < np) todo |= 1u << r;

    for (int pass = 0; pass < PM_MAX_PASSES; ++pass) {
        if (tid == 0) s_first = 0x7fffffff;
        __syncthreads();  // also orders the tile zeroing / previous flush before this pass's adds
        if (todo) {
            const int r = __ffs((int)todo) - 1;
            atomicMin(&s_first, r * BLK + tid);
        }
        __syncthreads();
        const int first = s_first;
        if (first == 0x7fffffff) break;  // uniform: nothing left in the whole workgroup
        if (tid == 0) {
            const float *q = pos + 3 * (base + first);
            const int i = cell_of(q[0] * nf, n), j = cell_of(q[1] * nf, n), k = cell_of(q[2] * nf, n);
            s_anchor = ((((i >> 4) << tshift) + (j >> 4)) << tshift) + (k >> 4);
        }
        __syncthreads();
        const int anchor = s_anchor;
        const int tk = anchor & ((1 << tshift) - 1);
        const int tj = (anchor >> tshift) & ((1 << tshift) - 1);
        const int ti = anchor >> (2 * tshift);
#pragma unroll 1
        for (int r = 0; r < PM_PPT; ++r) {
            if (todo >> r & 1u) {
                const float *q = pos + 3 * (base + r * BLK + tid);
                const float xp = q[0] * nf, yp = q[1] * nf, zp = q[2] * nf;
                const int i = cell_of(xp, n), j = cell_of(yp, n), k = cell_of(zp, n);
                if ((i >> 4) == ti && (j >> 4) == tj && (k >> 4) == tk) {
                    deposit_lds<SCHEME>(xp, yp, zp, i, j, k, ti, tj, tk, tile);
                    todo &= ~(1u << r);
                }
            }
        }
        __syncthreads();
        // flush: one global atomic per touched cell; rows of 18 contiguous floats along k
        const int gi0 = ti * PM_T - 1, gj0 = tj * PM_T - 1, gk0 = tk * PM_T - 1;
        for (int c = tid; c < PM_TILE; c += BLK) {
            const float v = tile[c];
            if (v != 0.f) {
                const int lk = c % PM_TH, lij = c / PM_TH;
                const int lj = lij % PM_TH, li = lij / PM_TH;
                const int64_t g =
                    ((((int64_t)((gi0 + li) & mask) << log2n) + ((gj0 + lj) & mask)) << log2n) + ((gk0 + lk) & mask);
                unsafeAtomicAdd(rho + g, v);
                tile[c] = 0.f;
            }
        }
    }
    // stragglers (unsorted input): straight to global atomics
#pragma unroll 1
    for (int r = 0; r < PM_PPT; ++r)
        if (todo >> r & 1u) {
            const float *q = pos + 3 * (base + r * BLK + tid);
            deposit_global<SCHEME>(q[0], q[1], q[2], n, log2n, rho);
        }
}

template <int SCHEME>
static int deposit_launch(pm_ctx *c, const float *pos, int64_t np, float *rho, int sorted_hint) {
    const int n = c->n;
    if (sorted_hint && n >= 16) {
        const int64_t blocks = (np + PM_CHUNK - 1) / PM_CHUNK;
        hipLaunchKernelGGL((deposit_tiled_kernel<SCHEME>), dim3((unsigned)blocks), dim3(BLK), 0, c->stream, pos, np, n,
                           c->log2n, rho);
    } else {
        const int blocks = pm_stream_blocks(np, BLK);
        hipLaunchKernelGGL((deposit_atomic_kernel<SCHEME>), dim3(blocks), dim3(BLK), 0, c->stream, pos, np, n,
                           c->log2n, rho);
    }
    PM_LAUNCH_CHECK();
    return PM_OK;
}

extern "C" int pm_deposit(pm_ctx *c, int scheme, const float *d_pos, int64_t np, float *d_rho, int sorted_hint) {
    PM_REQUIRE(c && d_rho && np >= 0 && (d_pos || np == 0), "bad argument");
    PM_REQUIRE((np + PM_CHUNK - 1) / PM_CHUNK < 2147483647LL, "too many particles for one launch");
    const size_t cells = (size_t)c->n * c->n * c->n;
    PM_HIP(hipMemsetAsync(d_rho, 0, cells * sizeof(float), c->stream));  // utils.initialise_zero
    if (np == 0) return PM_OK;
    switch (scheme) {
        case PM_NGP: return deposit_launch<PM_NGP>(c, d_pos, np, d_rho, sorted_hint);
        case PM_CIC: return deposit_launch<PM_CIC>(c, d_pos, np, d_rho, sorted_hint);
        case PM_TSC: return deposit_launch<PM_TSC>(c, d_pos, np, d_rho, sorted_hint);
    }
    pm_set_error("mass scheme %d, should be NGP(1), CIC(2) or TSC(3)", scheme);
    return PM_EINVAL;
}

// ------------------------------------------------------------------------------------------------
// inverse interpolation (gather).  One particle per thread; the 27 (8, 1) cell values come through
// L1/L2, which the Morton-sorted stream keeps warm.  The sum keeps the reference's left-to-right
// order (mesh.py:3058-3086) so the result is bit-identical to the CPU oracle.

template <int SCHEME, int NCOMP>
__global__ __launch_bounds__(BLK) void interp_kernel(const float *__restrict__ grid, const float *__restrict__ pos,
                                                     int64_t np, int n, int log2n, float *__restrict__ out) {
    const float nf = (float)n;
    const int mask = n - 1;
    for (int64_t p = (int64_t)blockIdx.x * BLK + threadIdx.x; p < np; p += (int64_t)gridDim.x * BLK) {
        const float xp = pos[3 * p] * nf, yp = pos[3 * p + 1] * nf, zp = pos[3 * p + 2] * nf;
        const int i = cell_of(xp, n), j = cell_of(yp, n), k = cell_of(zp, n);
        auto at = [&](int ii, int jj, int kk) -> const float * {
            return grid + NCOMP * (((((int64_t)(ii & mask) << log2n) + (jj & mask)) << log2n) + (kk & mask));
        };
        float acc[NCOMP];
        if (SCHEME == PM_NGP) {
            const float *g = at(i, j, k);
#pragma unroll
            for (int m = 0; m < NCOMP; ++m) acc[m] = g[m];
        } else if (SCHEME == PM_CIC) {
            float wx[2], wy[2], wz[2];
            int sx, sy, sz;
            cic_weights(xp, i, wx[0], wx[1], sx);
            cic_weights(yp, j, wy[0], wy[1], sy);
            cic_weights(zp, k, wz[0], wz[1], sz);
            bool first = true;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const float w = wx[a] * wy[b] * wz[c];
                        const float *g = at(i + a * sx, j + b * sy, k + c * sz);
#pragma unroll
                        for (int m = 0; m < NCOMP; ++m) acc[m] = first ? w * g[m] : acc[m] + w * g[m];
                        first = false;
                    }
        } else {
            float wx[3], wy[3], wz[3];
            tsc_weights(xp, i, wx);
            tsc_weights(yp, j, wy);
            tsc_weights(zp, k, wz);
            bool first = true;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const float wxy = wx[a] * wy[b];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float w = wxy * wz[c];
                        const float *g = at(i + a - 1, j + b - 1, k + c - 1);
#pragma unroll
                        for (int m = 0; m < NCOMP; ++m) acc[m] = first ? w * g[m] : acc[m] + w * g[m];
                        first = false;
                    }
                }
        }
#pragma unroll
        for (int m = 0; m < NCOMP; ++m) out[NCOMP * p + m] = acc[m];
    }
}

template <int NCOMP>
static int interp_launch(pm_ctx *c, int scheme, const float *grid, const float *pos, int64_t np, float *out) {
    const int blocks = pm_stream_blocks(np, BLK);
    switch (scheme) {
        case PM_NGP:
            hipLaunchKernelGGL((interp_kernel<PM_NGP, NCOMP>), dim3(blocks), dim3(BLK), 0, c->stream, grid, pos, np,
                               c->n, c->log2n, out);
            break;
        case PM_CIC:
            hipLaunchKernelGGL((interp_kernel<PM_CIC, NCOMP>), dim3(blocks), dim3(BLK), 0, c->stream, grid, pos, np,
                               c->n, c->log2n, out);
            break;
        case PM_TSC:
            hipLaunchKernelGGL((interp_kernel<PM_TSC, NCOMP>), dim3(blocks), dim3(BLK), 0, c->stream, grid, pos, np,
                               c->n, c->log2n, out);
            break;
        default:
            pm_set_error("mass scheme %d, should be NGP(1), CIC(2) or TSC(3)", scheme);
            return PM_EINVAL;
    }
    PM_LAUNCH_CHECK();
    return PM_OK;
}

extern "C" int pm_interp(pm_ctx *c, int scheme, int ncomp, const float *d_grid, const float *d_pos, int64_t np,
                         float *d_out) {
    PM_REQUIRE(c && d_grid && np >= 0 && ((d_pos && d_out) || np == 0), "bad argument");
    if (np == 0) return PM_OK;
    if (ncomp == 3) return interp_launch<3>(c, scheme, d_grid, d_pos, np, d_out);
    if (ncomp == 1) return interp_launch<1>(c, scheme, d_grid, d_pos, np, d_out);
    pm_set_error("ncomp %d, should be 1 or 3", ncomp);
    return PM_EINVAL;
}

// ------------------------------------------------------------------------------------------------
// Morton keys (morton.py:42-137) and re-ordering (utils.py:1019-1116)

__device__ __forceinline__ int64_t interleave21(int64_t x) {
    x &= 0x1FFFFF;
    x = (x | x << 32) & 0x1F00000000FFFFLL;
    x = (x | x << 16) & 0x1F0000FF0000FFLL;
    x = (x | x << 8) & 0x100F00F00F00F00FLL;
    x = (x | x << 4) & 0x10C30C30C30C30C3LL;
    x = (x | x << 2) & 0x1249249249249249LL;
    return x;
}

__global__ __launch_bounds__(BLK) void morton_keys_kernel(const float *__restrict__ pos, int64_t np,
                                                          int64_t *__restrict__ keys, int64_t *__restrict__ iota) {
    for (int64_t p = (int64_t)blockIdx.x * BLK + threadIdx.x; p < np; p += (int64_t)gridDim.x * BLK) {
        const int64_t xx = interleave21((int64_t)floorf(pos[3 * p] * 2097152.0f));
        const int64_t yy = interleave21((int64_t)floorf(pos[3 * p + 1] * 2097152.0f));
        const int64_t zz = interleave21((int64_t)floorf(pos[3 * p + 2] * 2097152.0f));
        keys[p] = xx << 2 | yy << 1 | zz;
        if (iota) iota[p] = p;
    }
}

extern "C" int pm_morton_keys(pm_ctx *c, const float *d_pos, int64_t np, int64_t *d_keys) {
    PM_REQUIRE(c && np >= 0 && ((d_pos && d_keys) || np == 0), "bad argument");
    if (np == 0) return PM_OK;
    hipLaunchKernelGGL(morton_keys_kernel, dim3(pm_stream_blocks(np, BLK)), dim3(BLK), 0, c->stream, d_pos, np, d_keys,
                       (int64_t *)nullptr);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

__global__ __launch_bounds__(BLK) void gather3_kernel(const int64_t *__restrict__ perm, int64_t np,
                                                      const float *__restrict__ a, const float *__restrict__ b,
                                                      const float *__restrict__ cc, float *__restrict__ ao,
                                                      float *__restrict__ bo, float *__restrict__ co) {
    for (int64_t p = (int64_t)blockIdx.x * BLK + threadIdx.x; p < np; p += (int64_t)gridDim.x * BLK) {
        const int64_t s = perm[p];
        ao[3 * p] = a[3 * s], ao[3 * p + 1] = a[3 * s + 1], ao[3 * p + 2] = a[3 * s + 2];
        if (b) bo[3 * p] = b[3 * s], bo[3 * p + 1] = b[3 * s + 1], bo[3 * p + 2] = b[3 * s + 2];
        if (cc) co[3 * p] = cc[3 * s], co[3 * p + 1] = cc[3 * s + 1], co[3 * p + 2] = cc[3 * s + 2];
    }
}

extern "C" int pm_reorder_particles(pm_ctx *c, int64_t np, const float *d_pos, const float *d_vel, const float *d_acc,
                                    float *d_pos_out, float *d_vel_out, float *d_acc_out, int64_t *d_perm) {
    PM_REQUIRE(c && np >= 0, "bad argument");
    if (np == 0) return PM_OK;
    PM_REQUIRE(d_pos && d_pos_out && d_pos != d_pos_out, "pos / pos_out must be distinct non-NULL buffers");
    PM_REQUIRE((!d_vel) == (!d_vel_out) && (!d_acc) == (!d_acc_out), "in/out arrays must come in pairs");
    PM_REQUIRE(np < 2147483647LL, "rocPRIM radix sort is limited to 2^31-1 keys per call");
    // workspace: keys_in, keys_out, iota, perm (4 x int64[np]) + rocPRIM temp
    size_t tmp_bytes = 0;
    int64_t *nullk = nullptr;
    PM_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, nullk, nullk, nullk, nullk, (size_t)np, 0, 63, c->stream));
    const size_t arr = ((size_t)np * sizeof(int64_t) + 255) & ~(size_t)255;
    const size_t need = 4 * arr + tmp_bytes;
    if (c->sort_tmp_bytes < need) {
        if (c->sort_tmp) PM_HIP(hipFree(c->sort_tmp));
        c->sort_tmp = nullptr;
        c->sort_tmp_bytes = 0;
        PM_HIP(hipMalloc(&c->sort_tmp, need));
        c->sort_tmp_bytes = need;
    }
    char *w = (char *)c->sort_tmp;
    int64_t *keys_in = (int64_t *)w, *keys_out = (int64_t *)(w + arr), *iota = (int64_t *)(w + 2 * arr),
            *perm = (int64_t *)(w + 3 * arr);
    void *rp_tmp = w + 4 * arr;
    hipLaunchKernelGGL(morton_keys_kernel, dim3(pm_stream_blocks(np, BLK)), dim3(BLK), 0, c->stream, d_pos, np, keys_in,
                       iota);
    PM_LAUNCH_CHECK();
    // keys are non-negative 63-bit integers: bits [0, 63)
    PM_HIP(rocprim::radix_sort_pairs(rp_tmp, tmp_bytes, keys_in, keys_out, iota, perm, (size_t)np, 0, 63, c->stream));
    hipLaunchKernelGGL(gather3_kernel, dim3(pm_stream_blocks(np, BLK)), dim3(BLK), 0, c->stream, perm, np, d_pos, d_vel,
                       d_acc, d_pos_out, d_vel_out, d_acc_out);
    PM_LAUNCH_CHECK();
    if (d_perm) PM_HIP(hipMemcpyAsync(d_perm, perm, (size_t)np * sizeof(int64_t), hipMemcpyDeviceToDevice, c->stream));
    return PM_OK;
}
