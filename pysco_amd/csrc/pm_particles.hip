// Particle <-> mesh kernels: mass deposit (mesh.py:2240-2595), inverse interpolation
// (mesh.py:2600-3088), Morton keys (morton.py:42-137) and the particle re-ordering
// (utils.py:1019-1116).
//
// Deposit, MI355X design: global fp32 atomics run at ~1.3 TB/s chip-wide and only when a wave
// instruction touches contiguous bytes, so scattering 27 atomics per particle is ~20x off the HBM
// roofline.  Instead a workgroup takes a contiguous chunk of the (Morton-sorted) particle stream,
// accumulates it into an LDS tile of 16x16x16 cells + a one-cell halo, and flushes the tile with
// one global atomic per touched cell (rows of 18 contiguous floats).  The LDS tile is NOT fp32:
// measured on gfx950 (tools/micro/lds_atomics.hip) ds_add_f32 retires 0.33 lane-ops/clk/CU while
// ds_add_f64 retires 6.4 and ds_add_u64 7.8 -- a 20x difference -- so each fp32 weight (computed
// exactly as the reference does) is summed in 64-bit fixed point and rounded once at the flush.  A chunk that
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
__device__ __forceinline__ void deposit_global(float x, float y, float z, const pm_geom &geo, float *rho) {
    const int n = geo.n;
    const float nf = (float)n;
    const float xp = x * nf, yp = y * nf, zp = z * nf;
    const int i = cell_of(xp, n), j = cell_of(yp, n), k = cell_of(zp, n);
    auto at = [&](int ii, int jj, int kk) -> float * { return rho + pm_cell(geo, ii, jj, kk); };
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
__global__ __launch_bounds__(BLK) void deposit_atomic_kernel(const float *__restrict__ pos, int64_t np, pm_geom geo,
                                                             float *__restrict__ rho) {
    for (int64_t p = (int64_t)blockIdx.x * BLK + threadIdx.x; p < np; p += (int64_t)gridDim.x * BLK)
        deposit_global<SCHEME>(pos[3 * p], pos[3 * p + 1], pos[3 * p + 2], geo, rho);
}

// ------------------------------------------------------------------------------------------------
// LDS tiles shared by the tiled deposit and the tiled gather.
// A tile is 8 x 16 x 16 cells (x, y, z) -- a contiguous range of 2048 Morton codes, i.e. what one
// 2048-particle workgroup chunk of a sorted stream covers at one particle per cell -- plus a
// one-cell halo on every side (TSC/CIC reach i +- 1).

constexpr int PM_TI = 8, PM_TJ = 16, PM_TK = 16;      // tile extent in cells
constexpr int PM_LI = 3, PM_LJ = 4, PM_LK = 4;        // log2 of the above
constexpr int PM_HI = PM_TI + 2, PM_HJ = PM_TJ + 2, PM_HK = PM_TK + 2;
constexpr int PM_TILE = PM_HI * PM_HJ * PM_HK;        // 3240 cells
constexpr int PM_FSTRIDE = 3264;                      // cells per plane of the LDS force tile (51 * 64)
// LDS force tile: an (x, y) plane of float2 per cell, then a z plane.  The gather reads a cell's (x, y) with
// one ds_read_b64 straight into the register pair of a packed multiply/add (v_pk_mul_f32, v_pk_add_f32);
// with three separate component planes the compiler paid a v_mov per operand to build those pairs.
constexpr int PM_FZ = 2 * PM_FSTRIDE;                 // float offset of the z plane
typedef float pm_v2f __attribute__((ext_vector_type(2)));
typedef float pm_v4f __attribute__((ext_vector_type(4)));
typedef float pm_v3f __attribute__((ext_vector_type(3)));
typedef float pm_v3fu __attribute__((ext_vector_type(3), aligned(4)));  // a 12-byte particle record: one dwordx3 access
// byte offset of a __shared__ object inside the workgroup's LDS allocation (the address a ds_* instruction takes)
template <typename T>
__device__ __forceinline__ unsigned lds_offset(const T *p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
__device__ __forceinline__ int force_tile_index(int c, int m) { return m < 2 ? 2 * c + m : PM_FZ + c; }
__device__ __forceinline__ void force_tile_put(float *tile, int c, float gx, float gy, float gz) {
    *reinterpret_cast<pm_v2f *>(tile + 2 * c) = pm_v2f{gx, gy};
    tile[PM_FZ + c] = gz;
}
#ifndef PM_ROT_CLASSES
#define PM_ROT_CLASSES 27                             // orders in which the lanes of a dense range walk their 27 cells
#endif
constexpr int PM_PPT = 8;                             // particles per thread
constexpr int PM_CHUNK = BLK * PM_PPT;                // particles per workgroup
constexpr int PM_MAX_PASSES = 6;

struct TileId {
    int ti, tj, tk;
};

__device__ __forceinline__ bool in_tile(int i, int j, int k, const TileId &t) {
    return (i >> PM_LI) == t.ti && (j >> PM_LJ) == t.tj && (k >> PM_LK) == t.tk;
}

// Picks the tile of the first still-unprocessed particle of the workgroup's chunk (uniform result);
// returns false when nothing is left.  Two barriers.
__device__ __forceinline__ bool next_anchor(unsigned todo, const float *chunk_pos, float nf, int n, int *s_first,
                                            int *s_anchor, TileId &t) {
    const int tid = threadIdx.x;
    if (tid == 0) *s_first = 0x7fffffff;
    __syncthreads();
    if (todo) atomicMin(s_first, (__ffs((int)todo) - 1) * BLK + tid);
    __syncthreads();
    const int first = *s_first;
    if (first == 0x7fffffff) return false;
    if (tid == 0) {
        const float *q = chunk_pos + 3 * first;
        s_anchor[0] = cell_of(q[0] * nf, n) >> PM_LI;
        s_anchor[1] = cell_of(q[1] * nf, n) >> PM_LJ;
        s_anchor[2] = cell_of(q[2] * nf, n) >> PM_LK;
    }
    __syncthreads();
    t.ti = s_anchor[0], t.tj = s_anchor[1], t.tk = s_anchor[2];
    return true;
}

// global cell index of tile-local cell c (with halo, periodic)
__device__ __forceinline__ int64_t tile_cell_global(int c, const TileId &t, const pm_geom &geo) {
    const int lk = c % PM_HK, lij = c / PM_HK;
    const int lj = lij % PM_HJ, li = lij / PM_HJ;
    return pm_cell(geo, t.ti * PM_TI - 1 + li, t.tj * PM_TJ - 1 + lj, t.tk * PM_TK - 1 + lk);
}

// ------------------------------------------------------------------------------------------------
// deposit, LDS-tiled (see the header comment)

// LDS accumulators are 64-bit fixed point (weight * 2^30, truncated): measured on gfx950
// (tools/micro/lds_atomics.hip) ds_add_u64 retires 7.8 lane-ops/clk/CU against 6.4 for ds_add_f64 and
// 0.33 for ds_add_f32.  Weights are in [0, 1]; a contribution loses at most 2^-30 = 9e-10 (fp32 ulp of
// an O(1) density is 6e-8), a cell holds up to 2^34 particles, and -- integer adds being associative --
// a tile's sums no longer depend on the order in which its particles arrive.
typedef unsigned long long pm_acc_t;
__device__ __forceinline__ pm_acc_t acc_of_weight(float w) { return (pm_acc_t)(unsigned)(w * 1073741824.0f); }
__device__ __forceinline__ float weight_of_acc(pm_acc_t a) { return (float)((double)a * (1.0 / 1073741824.0)); }

template <int SCHEME, bool ROTATE = false>
__device__ __forceinline__ void deposit_lds(float xp, float yp, float zp, int i, int j, int k, const TileId &t,
                                            pm_acc_t *tile) {
    // local coordinates inside the haloed tile: cell (ti*TI-1) is local 0
    const int li = i - t.ti * PM_TI + 1, lj = j - t.tj * PM_TJ + 1, lk = k - t.tk * PM_TK + 1;
    auto at = [&](int a, int b, int c) -> pm_acc_t * { return tile + ((li + a) * PM_HJ + (lj + b)) * PM_HK + (lk + c); };
    if (ROTATE && SCHEME == PM_TSC) {
        // Lanes in the same cell (a clustered stream: hundreds of particles per cell) add to the same 27 addresses in
        // the same 27 instructions, and the LDS serialises them: measured at 1024^3 after 80 steps of the benchmark's
        // own evolution (61 % of the cells empty, 39 % of the particles in cells of 8 and more) the fused kick + deposit
        // takes 17.5 ms where conflict-free dummy addresses take 14.1.  Here every lane walks its (b, c) rows in an
        // order of its own (9 classes by lane), so that lanes of one cell meet 9 times less often: 15.5 ms.  64-bit
        // integer sums: any order gives the same bits.  (The rotated weights and offsets cost ~60 more vector
        // instructions per particle -- 0.4 ms on a uniform stream -- so only workgroups with a dense range take this
        // path: kdw_deposit_kernel.)
        float wx[3], wy[3], wz[3];
        tsc_weights(xp, i, wx);
        tsc_weights(yp, j, wy);
        tsc_weights(zp, k, wz);
        const int r = (int)(threadIdx.x & 63u) % PM_ROT_CLASSES, ra = r / 9, rb = (r / 3) % 3, rc = r % 3;
        float wxr[3], wyr[3], wzr[3];
        int oa[3], ob[3], oc[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            int ia = q + ra, ib = q + rb, ic = q + rc;
            ia = ia >= 3 ? ia - 3 : ia, ib = ib >= 3 ? ib - 3 : ib, ic = ic >= 3 ? ic - 3 : ic;
            wxr[q] = ia == 0 ? wx[0] : (ia == 1 ? wx[1] : wx[2]);
            wyr[q] = ib == 0 ? wy[0] : (ib == 1 ? wy[1] : wy[2]);
            wzr[q] = ic == 0 ? wz[0] : (ic == 1 ? wz[1] : wz[2]);
            oa[q] = (ia - 1) * PM_HJ * PM_HK, ob[q] = (ib - 1) * PM_HK, oc[q] = ic - 1;
        }
        pm_acc_t *const base = tile + (li * PM_HJ + lj) * PM_HK + lk;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float wxy = wxr[a] * wyr[b];
                pm_acc_t *const row = base + oa[a] + ob[b];
#pragma unroll
                for (int c = 0; c < 3; ++c) atomicAdd(row + oc[c], acc_of_weight(wxy * wzr[c]));
            }
        return;
    }
    if (SCHEME == PM_NGP) {
        atomicAdd(at(0, 0, 0), acc_of_weight(1.0f));
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
                    atomicAdd(at(a * sx, b * sy, c * sz), acc_of_weight(wx[a] * wy[b] * wz[c]));
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
                for (int c = 0; c < 3; ++c) atomicAdd(at(a - 1, b - 1, c - 1), acc_of_weight(wxy * wz[c]));
            }
    }
}

template <int SCHEME>
__global__ __launch_bounds__(BLK) void deposit_tiled_kernel(const float *__restrict__ pos, int64_t np, pm_geom geo,
                                                            float *__restrict__ rho) {
    __shared__ pm_acc_t tile[PM_TILE];  // 25.9 KB
    __shared__ int s_first;
    __shared__ int s_anchor[3];
    const int tid = threadIdx.x;
    const int n = geo.n;
    const float nf = (float)n;
    for (int c = tid; c < PM_TILE; c += BLK) tile[c] = 0;

    const int64_t base = (int64_t)blockIdx.x * PM_CHUNK;
    // Particles are re-read from L2 on every pass rather than held in registers or staged in LDS
    // (staging measured 25 % slower: it costs occupancy and a sorted chunk needs only 1-3 passes).
    unsigned todo = 0;
#pragma unroll
    for (int r = 0; r < PM_PPT; ++r)
        if (base + r * BLK + tid < np) todo |= 1u << r;

    TileId t;
    for (int pass = 0; pass < PM_MAX_PASSES; ++pass) {
        // (the first barrier inside also orders the tile zeroing / previous flush before the adds)
        if (!next_anchor(todo, pos + 3 * base, nf, n, &s_first, s_anchor, t)) break;
#pragma unroll 1
        for (int r = 0; r < PM_PPT; ++r) {
            if (todo >> r & 1u) {
                const float *q = pos + 3 * (base + r * BLK + tid);
                const float xp = q[0] * nf, yp = q[1] * nf, zp = q[2] * nf;
                const int i = cell_of(xp, n), j = cell_of(yp, n), k = cell_of(zp, n);
                if (in_tile(i, j, k, t)) {
                    deposit_lds<SCHEME>(xp, yp, zp, i, j, k, t, tile);
                    todo &= ~(1u << r);
                }
            }
        }
        __syncthreads();
        // flush: one global atomic per touched cell; rows of 18 contiguous floats along k
        for (int c = tid; c < PM_TILE; c += BLK) {
            const pm_acc_t v = tile[c];
            if (v != 0) {
                unsafeAtomicAdd(rho + tile_cell_global(c, t, geo), weight_of_acc(v));
                tile[c] = 0;
            }
        }
    }
    // stragglers (unsorted input): straight to global atomics
#pragma unroll 1
    for (int r = 0; r < PM_PPT; ++r)
        if (todo >> r & 1u) {
            const float *q = pos + 3 * (base + r * BLK + tid);
            deposit_global<SCHEME>(q[0], q[1], q[2], geo, rho);
        }
}

// ------------------------------------------------------------------------------------------------
// inverse interpolation (gather).  One particle per thread; the 27 (8, 1) cell values come through
// L1/L2, which the Morton-sorted stream keeps warm.  The sum keeps the reference's left-to-right
// order (mesh.py:3058-3086) so the result is bit-identical to the CPU oracle.

template <int SCHEME, int NCOMP>
__global__ __launch_bounds__(BLK) void interp_kernel(const float *__restrict__ grid, const float *__restrict__ pos,
                                                     int64_t np, pm_geom geo, float *__restrict__ out) {
    const int n = geo.n;
    const float nf = (float)n;
    for (int64_t p = (int64_t)blockIdx.x * BLK + threadIdx.x; p < np; p += (int64_t)gridDim.x * BLK) {
        const float xp = pos[3 * p] * nf, yp = pos[3 * p + 1] * nf, zp = pos[3 * p + 2] * nf;
        const int i = cell_of(xp, n), j = cell_of(yp, n), k = cell_of(zp, n);
        auto at = [&](int ii, int jj, int kk) -> const float * { return grid + NCOMP * pm_cell(geo, ii, jj, kk); };
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
                               pm_ctx_geom(c), out);
            break;
        case PM_CIC:
            hipLaunchKernelGGL((interp_kernel<PM_CIC, NCOMP>), dim3(blocks), dim3(BLK), 0, c->stream, grid, pos, np,
                               pm_ctx_geom(c), out);
            break;
        case PM_TSC:
            hipLaunchKernelGGL((interp_kernel<PM_TSC, NCOMP>), dim3(blocks), dim3(BLK), 0, c->stream, grid, pos, np,
                               pm_ctx_geom(c), out);
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
// gather, LDS-tiled and fused with the second leapfrog kick (integration.py:262) and the two CFL
// reductions of the next step (integration.py:294, 325).
//
// The direct kernel above is bound by the L1 tag rate (27 scattered 12-byte loads per particle).
// Here the workgroup stages the force tile of its chunk (8x16x16 cells + halo: an (x, y) float2 plane and
// a z plane, 39.2 KB) in LDS with coalesced loads and every particle does 27 x (ds_read_b64 + ds_read_b32).  Sum order is the
// reference's, so the accelerations are bit-identical to the direct kernel and to the oracle.

template <int SCHEME>
__device__ __forceinline__ void gather_lds(float xp, float yp, float zp, int i, int j, int k, const TileId &t,
                                           const float *tile, float acc[3]) {
    const int li = i - t.ti * PM_TI + 1, lj = j - t.tj * PM_TJ + 1, lk = k - t.tk * PM_TK + 1;
    // 24-bit multiplies: full rate, where v_mul_lo_u32 is quarter rate (li, lj, lk are small and >= 0)
    // 24-bit multiply-adds, as asm: from C (`* 18`, `__umul24`, shift-adds alike) the compiler canonicalises to
    // a multiply by 18 and selects the quarter-rate v_mul_lo_u32 (li, lj, lk are small and >= 0)
    int r0, c0;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r0) : "v"(li), "n"(PM_HJ), "v"(lj));
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(c0) : "v"(r0), "n"(PM_HK), "v"(lk));
    const pm_v2f *txy = reinterpret_cast<const pm_v2f *>(tile);
    const float *tz = tile + PM_FZ;
    // The running sums start from +0 (0 + w*g == w*g exactly) and add the terms in the reference's
    // left-to-right order (mesh.py:3058-3086); x and y ride in one packed register pair, which
    // rounds each lane exactly like the scalar multiply and add.
    pm_v2f axy = {0.f, 0.f};
    float az = 0.f;
    auto add = [&](int off, float w) {
        const pm_v2f g = txy[c0 + off];
        const float gz = tz[c0 + off];
        axy = axy + pm_v2f{w, w} * g;
        az = az + w * gz;
    };
    if (SCHEME == PM_NGP) {
        axy = txy[c0], az = tz[c0];
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
                for (int c = 0; c < 2; ++c) add((a * sx * PM_HJ + b * sy) * PM_HK + c * sz, wx[a] * wy[b] * wz[c]);
    } else {
        float wx[3], wy[3], wz[3];
        tsc_weights(xp, i, wx);
        tsc_weights(yp, j, wy);
        tsc_weights(zp, k, wz);
        // the x-plane loop stays rolled: nine cells in flight instead of 27 keeps the kernel under
        // 128 VGPRs; inside one plane every offset is an instruction immediate
        // The (x, y) reads are single ds_read_b64, written as asm: the compiler pairs adjacent 8-byte reads into
        // ds_read2_b64, which the LDS serves at half the rate (4 cycles per 8 bytes per lane against 2:
        // MI355X_MICROARCH.md, LDS table).  The compiler does not count asm reads in lgkmcnt: the wait below
        // is explicit and names every destination, so nothing is consumed before it.  (z stays on plain
        // 4-byte reads: fetched as aligned 8-byte pairs + a parity select it saved LDS cycles but cost 27
        // v_cndmask per particle, and with the LDS relieved the kernel is VALU-bound: 2 % slower.)
        const unsigned a_xy = lds_offset(txy + (c0 - (PM_HJ + 1) * PM_HK - 1));  // cell (-1, -1, -1)
#define PM_DS_READ_B64(dst, addr, off) asm("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define PM_DS_READ_B32(dst, addr, off) asm("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
        // One drain per x-plane: the nine z reads are issued behind the nine (x, y) reads and waited for together.  (As C
        // loads the compiler placed them behind the explicit wait of the (x, y) reads: two LDS round trips per plane.)
        const unsigned a_z = lds_offset(tz + (c0 - (PM_HJ + 1) * PM_HK - 1));
#pragma unroll 1
        for (int a = 0; a < 3; ++a) {
            const float wxa = a == 0 ? wx[0] : (a == 1 ? wx[1] : wx[2]);
            const unsigned pxy = a_xy + (unsigned)a * (PM_HJ * PM_HK * 8), pz = a_z + (unsigned)a * (PM_HJ * PM_HK * 4);
            pm_v2f g[3][3];
            float gz[3][3];
#define PM_ROW(b)                                     \
    PM_DS_READ_B64(g[b][0], pxy, (b * PM_HK + 0) * 8); \
    PM_DS_READ_B64(g[b][1], pxy, (b * PM_HK + 1) * 8); \
    PM_DS_READ_B64(g[b][2], pxy, (b * PM_HK + 2) * 8); \
    PM_DS_READ_B32(gz[b][0], pz, (b * PM_HK + 0) * 4); \
    PM_DS_READ_B32(gz[b][1], pz, (b * PM_HK + 1) * 4); \
    PM_DS_READ_B32(gz[b][2], pz, (b * PM_HK + 2) * 4)
            PM_ROW(0);
            PM_ROW(1);
            PM_ROW(2);
#undef PM_ROW
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(g[0][0]), "+v"(g[0][1]), "+v"(g[0][2]), "+v"(g[1][0]), "+v"(g[1][1]), "+v"(g[1][2]),
                           "+v"(g[2][0]), "+v"(g[2][1]), "+v"(g[2][2]), "+v"(gz[0][0]), "+v"(gz[0][1]), "+v"(gz[0][2]),
                           "+v"(gz[1][0]), "+v"(gz[1][1]), "+v"(gz[1][2]), "+v"(gz[2][0]), "+v"(gz[2][1]), "+v"(gz[2][2]));
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float wxy = wxa * wy[b];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float w = wxy * wz[c];
                    axy = axy + pm_v2f{w, w} * g[b][c];
                    az = az + w * gz[b][c];
                }
            }
        }
#undef PM_DS_READ_B64
#undef PM_DS_READ_B32
    }
    acc[0] = axy.x, acc[1] = axy.y, acc[2] = az;
}

// Same sum straight from global memory: the path of particles that are not inside the staged tile
// (unsorted or drifted streams).  Deliberately compact (rolled loops, not inlined): it is the rare
// path and must not set the register budget of the kernels that call it.
// The sums come back by value (three VGPRs): handed a pointer, the callers' accumulators lived in scratch,
// and the scratch reload's vmcnt(0) drained the particle prefetch queues once per particle.
struct Acc3 {
    float x, y, z;
};

template <int SCHEME>
__device__ __noinline__ Acc3 gather_global(const float *__restrict__ grid, float xp, float yp, float zp, int i, int j,
                                           int k, pm_geom geo) {
    auto at = [&](int ii, int jj, int kk) -> const float * { return grid + 3 * pm_cell(geo, ii, jj, kk); };
    float acc[3];
    acc[0] = acc[1] = acc[2] = 0.f;
    if (SCHEME == PM_NGP) {
        const float *g = at(i, j, k);
        acc[0] = g[0], acc[1] = g[1], acc[2] = g[2];
    } else if (SCHEME == PM_CIC) {
        float wx[2], wy[2], wz[2];
        int sx, sy, sz;
        cic_weights(xp, i, wx[0], wx[1], sx);
        cic_weights(yp, j, wy[0], wy[1], sy);
        cic_weights(zp, k, wz[0], wz[1], sz);
#pragma unroll 1
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float w = (a ? wx[1] : wx[0]) * wy[b] * wz[c];
                    const float *g = at(i + a * sx, j + b * sy, k + c * sz);
#pragma unroll
                    for (int m = 0; m < 3; ++m) acc[m] = acc[m] + w * g[m];
                }
    } else {
        float wx[3], wy[3], wz[3];
        tsc_weights(xp, i, wx);
        tsc_weights(yp, j, wy);
        tsc_weights(zp, k, wz);
#pragma unroll 1
        for (int a = 0; a < 3; ++a) {
            const float wxa = a == 0 ? wx[0] : (a == 1 ? wx[1] : wx[2]);
#pragma unroll 1
            for (int b = 0; b < 3; ++b) {
                const float wxy = wxa * (b == 0 ? wy[0] : (b == 1 ? wy[1] : wy[2]));
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float w = wxy * wz[c];
                    const float *g = at(i + a - 1, j + b - 1, k + c - 1);
#pragma unroll
                    for (int m = 0; m < 3; ++m) acc[m] = acc[m] + w * g[m];
                }
            }
        }
    }
    return Acc3{acc[0], acc[1], acc[2]};
}

template <int SCHEME, bool KICK>
__global__ __launch_bounds__(BLK) void interp_tiled_kernel(const float *__restrict__ grid,
                                                           const float *__restrict__ pos, int64_t np, pm_geom geo,
                                                           float *__restrict__ out, float *__restrict__ vel,
                                                           float mhalf_dt, unsigned *__restrict__ maxbits) {
    __shared__ __attribute__((aligned(16))) float tile[3 * PM_FSTRIDE];  // 39.2 KB
    __shared__ int s_first;
    __shared__ int s_anchor[3];
    __shared__ unsigned s_max[2][BLK / 64];
    const int tid = threadIdx.x;
    const int n = geo.n;
    const float nf = (float)n;
    const int64_t base = (int64_t)blockIdx.x * PM_CHUNK;
    unsigned todo = 0;
#pragma unroll
    for (int r = 0; r < PM_PPT; ++r)
        if (base + r * BLK + tid < np) todo |= 1u << r;
    unsigned ma = 0, mv = 0;  // bit patterns of max|a|, max|v| (integer order == float order, NaN on top)

    auto finish = [&](int64_t p, const float a[3]) {
        out[3 * p] = a[0], out[3 * p + 1] = a[1], out[3 * p + 2] = a[2];
        ma = max(ma, __float_as_uint(a[0]) & 0x7fffffffu);
        ma = max(ma, __float_as_uint(a[1]) & 0x7fffffffu);
        ma = max(ma, __float_as_uint(a[2]) & 0x7fffffffu);
        if (KICK) {
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const float v = vel[3 * p + m] + mhalf_dt * a[m];  // y += a*x (utils.py:297)
                vel[3 * p + m] = v;
                mv = max(mv, __float_as_uint(v) & 0x7fffffffu);
            }
        }
    };

    TileId t;
    for (int pass = 0; pass < PM_MAX_PASSES; ++pass) {
        // (the first barrier inside keeps this pass's tile fill behind the previous pass's reads)
        if (!next_anchor(todo, pos + 3 * base, nf, n, &s_first, s_anchor, t)) break;
        {   // tile fill: all loads in flight before the first LDS write
            constexpr int NQ = (PM_TILE + BLK - 1) / BLK;
            float gx[NQ], gy[NQ], gz[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int c = q * BLK + tid;
                if (c < PM_TILE) {
                    const float *g = grid + 3 * tile_cell_global(c, t, geo);
                    gx[q] = g[0], gy[q] = g[1], gz[q] = g[2];
                }
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int c = q * BLK + tid;
                if (c < PM_TILE) force_tile_put(tile, c, gx[q], gy[q], gz[q]);
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int r = 0; r < PM_PPT; ++r) {
            if (todo >> r & 1u) {
                const int64_t p = base + r * BLK + tid;
                const float xp = pos[3 * p] * nf, yp = pos[3 * p + 1] * nf, zp = pos[3 * p + 2] * nf;
                const int i = cell_of(xp, n), j = cell_of(yp, n), k = cell_of(zp, n);
                if (in_tile(i, j, k, t)) {
                    float a[3];
                    gather_lds<SCHEME>(xp, yp, zp, i, j, k, t, tile, a);
                    finish(p, a);
                    todo &= ~(1u << r);
                }
            }
        }
    }
#pragma unroll 1
    for (int r = 0; r < PM_PPT; ++r)
        if (todo >> r & 1u) {
            const int64_t p = base + r * BLK + tid;
            const float xp = pos[3 * p] * nf, yp = pos[3 * p + 1] * nf, zp = pos[3 * p + 2] * nf;
            const int i = cell_of(xp, n), j = cell_of(yp, n), k = cell_of(zp, n);
            const Acc3 r = gather_global<SCHEME>(grid, xp, yp, zp, i, j, k, geo);
            const float a[3] = {r.x, r.y, r.z};
            finish(p, a);
        }
    // workgroup max -> two global atomics
    for (int off = 32; off > 0; off >>= 1) {
        ma = max(ma, (unsigned)__shfl_down((int)ma, off, 64));
        mv = max(mv, (unsigned)__shfl_down((int)mv, off, 64));
    }
    if ((tid & 63) == 0) s_max[0][tid >> 6] = ma, s_max[1][tid >> 6] = mv;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < BLK / 64; ++w) ma = max(ma, s_max[0][w]), mv = max(mv, s_max[1][w]);
        // one same-address atomic costs ~12 ns at the L2 and half a million workgroups would queue up on
        // it; the maximum only grows, so a workgroup that cannot raise it (nearly all) just reads it
        if (ma > __atomic_load_n(maxbits, __ATOMIC_RELAXED)) atomicMax(maxbits, ma);
        if (KICK && mv > __atomic_load_n(maxbits + 1, __ATOMIC_RELAXED)) atomicMax(maxbits + 1, mv);
    }
}

// ------------------------------------------------------------------------------------------------
// Tile-range kernels: one workgroup per LDS tile, one pass, no anchor search.
//
// In a Morton-sorted stream (utils.reorder_particles: x is the most significant axis) the particles
// of one 8x16x16 tile form one contiguous range.  pm_tile_ranges finds, for every tile key t, the
// first particle whose key is >= t by bisection on the CURRENT positions and makes the table
// monotone with a running maximum, so [start[t], start[t+1]) is always a valid partition of
// [0, np) whatever the particle order is: when the stream is sorted the ranges are the tiles;
// particles that have drifted out of the tile they are filed under (or an unsorted stream) take
// the direct global-memory path inside the same kernel.  Correctness never depends on the order.
// (Tried and dropped: flushing each tile to a private slab with plain stores and summing the <= 8
// covering slabs per cell in a second kernel -- the assemble pass cost 0.98 ms at 512^3 against
// ~0.2 ms for the atomic flush it replaced.)

__device__ __forceinline__ unsigned spread3_10(unsigned v) {  // 10 bits -> every third bit
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__device__ __forceinline__ unsigned compact3_10(unsigned v) {
    v &= 0x09249249u;
    v = (v | (v >> 2)) & 0x030c30c3u;
    v = (v | (v >> 4)) & 0x0300f00fu;
    v = (v | (v >> 8)) & 0x030000ffu;
    v = (v | (v >> 16)) & 0x3ffu;
    return v;
}

// key of tile (ti, tj, tk) in stream order: cell-level Morton code (x most significant) >> 11
__device__ __forceinline__ unsigned tile_key(int ti, int tj, int tk) {
    return (unsigned)(ti & 1) |
           ((spread3_10((unsigned)tk) | (spread3_10((unsigned)tj) << 1) | (spread3_10((unsigned)(ti >> 1)) << 2)) << 1);
}

__device__ __forceinline__ TileId tile_of_key(unsigned key) {
    const unsigned rest = key >> 1;
    TileId t;
    t.tk = (int)compact3_10(rest);
    t.tj = (int)compact3_10(rest >> 1);
    t.ti = (int)((compact3_10(rest >> 2) << 1) | (key & 1u));
    return t;
}

__device__ __forceinline__ unsigned tile_key_of_pos(const float *__restrict__ q, float nf, int n) {
    const int i = cell_of(q[0] * nf, n), j = cell_of(q[1] * nf, n), k = cell_of(q[2] * nf, n);
    return tile_key(i >> PM_LI, j >> PM_LJ, k >> PM_LK);
}

__global__ __launch_bounds__(BLK) void tile_bounds_kernel(const float *__restrict__ pos, int64_t np, int n,
                                                          unsigned ntiles, int64_t *__restrict__ start) {
    const unsigned t = blockIdx.x * BLK + threadIdx.x;
    if (t > ntiles) return;
    const float nf = (float)n;
    int64_t lo = 0, hi = np;  // first p with key(p) >= t
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (tile_key_of_pos(pos + 3 * mid, nf, n) < t)
            lo = mid + 1;
        else
            hi = mid;
    }
    start[t] = (t == 0) ? 0 : (t == ntiles ? np : lo);
}

// A tile's range is processed PM_RANGE_CHUNK particles per workgroup: workgroup t < ntiles takes the
// first chunk of tile t, and the (rare) further chunks of over-full tiles -- clustered regions, or any
// range at all when the stream is not sorted -- are dealt to the surplus workgroups through a prefix
// table, so no workgroup ever loops over more than one chunk.
// (8192 since round 3: once the benchmark's own evolution has clustered the particles -- 1024^3, 80 steps: a tenth of
// the tiles hold more than 4096, a quarter of the particles live in them -- every surplus workgroup stages the
// tile again (the gather its force tile, the fused kick an accumulator tile it flushes with atomics): 10.5 %
// surplus workgroups with 4096, 1.5 % with 8192; gather 18.7 -> 18.2 ms, kick + deposit 16.6 -> 16.3, nothing
// either way on the uniform stream of the first steps)
constexpr int PM_RANGE_CHUNK = 8192;
constexpr int PM_SCAN_STAGE = 4096;  // strangers a workgroup of the straggler scan stages in LDS before appending
constexpr int PM_DENSE_RANGE = 2304;  // particles in a workgroup's range (of a tile of 2048 cells) from which it counts as dense

__global__ __launch_bounds__(BLK) void tile_extra_kernel(const int64_t *__restrict__ start, unsigned ntiles,
                                                         int64_t *__restrict__ extra) {
    const unsigned t = blockIdx.x * BLK + threadIdx.x;
    if (t > ntiles) return;
    int64_t x = 0;
    if (t < ntiles) {
        const int64_t len = start[t + 1] - start[t];
        x = len > PM_RANGE_CHUNK ? (len - 1) / PM_RANGE_CHUNK : 0;
    }
    extra[t] = x;
}

struct TileWork {
    const int64_t *start;   // [ntiles + 1] monotone particle offsets
    const int64_t *xstart;  // [ntiles + 1] exclusive prefix of the extra chunks per tile
    unsigned ntiles;        // tiles of the whole mesh (size of the tables)
    unsigned nlaunch;       // first-chunk workgroups = tiles inside the particle window (== ntiles on a single GPU)
    int ti0, lnti, lT;      // window: first tile in x, log2 of its tiles in x, log2 of the tiles per side in y / z
    unsigned boff;          // workgroup index = blockIdx.x + boff (a launch of the surplus workgroups only)
};

// Incoming stragglers per tile (pm_tile_lists): particles that the sorted order files under another
// tile but that are inside this one now.  With `use` set, a workgroup skips the strangers it meets in
// its own range (the tile they are in serves them from its list) and adds its own list to the range:
// every particle goes through LDS and the global straggler paths below are never taken.
struct TileLists {
    const unsigned *idx;     // particle indices grouped by current tile
    const unsigned *lstart;  // [ntiles + 1] offsets into idx
    int use;
};

// work item of this workgroup: tile key and particle range [s, e); false if it has none
// tile of first-chunk workgroup b < nlaunch (no memory access).  Workgroups go round-robin to the 8 XCDs
// (each with its own L2): deal every XCD runs of 64 consecutive tiles (a 32x64x64-cell Morton block) so the
// tile halos it re-reads are its own L2 hits.  On an x-slab (multi-GPU) only the tiles of the rank's planes
// are enumerated: x fastest inside the window, then a 2-D Morton walk over (y, z).
__device__ __forceinline__ unsigned compact2_16(unsigned v) {  // every second bit -> 16 bits
    v &= 0x55555555u;
    v = (v | (v >> 1)) & 0x33333333u;
    v = (v | (v >> 2)) & 0x0f0f0f0fu;
    v = (v | (v >> 4)) & 0x00ff00ffu;
    v = (v | (v >> 8)) & 0x0000ffffu;
    return v;
}

__device__ __forceinline__ unsigned tile_key_of_block(unsigned b, const TileWork &w) {
    constexpr unsigned G = 512;
    if (b < (w.nlaunch / G) * G) {
        const unsigned r = b % G;
        b = b - r + (r % 8) * (G / 8) + r / 8;
    }
    if (w.nlaunch == w.ntiles) return b;
    const unsigned til = b & ((1u << w.lnti) - 1u), yz = b >> w.lnti;
    return tile_key(w.ti0 + (int)til, (int)compact2_16(yz >> 1), (int)compact2_16(yz));
}

__device__ __forceinline__ bool tile_work_item(const TileWork &w, unsigned &key, int64_t &s, int64_t &e) {
    const unsigned b = blockIdx.x + w.boff;
    if (b < w.nlaunch) {
        key = tile_key_of_block(b, w);
        s = w.start[key];
        e = min(w.start[key + 1], s + PM_RANGE_CHUNK);
    } else {
        const int64_t x = (int64_t)(b - w.nlaunch);
        if (x >= w.xstart[w.ntiles]) return false;
        unsigned lo = 0, hi = w.ntiles;  // largest key with xstart[key] <= x
        while (hi - lo > 1) {
            const unsigned mid = (lo + hi) >> 1;
            if (w.xstart[mid] <= x)
                lo = mid;
            else
                hi = mid;
        }
        key = lo;
        s = w.start[lo] + (x - w.xstart[lo] + 1) * PM_RANGE_CHUNK;
        e = min(w.start[lo + 1], s + PM_RANGE_CHUNK);
    }
    return s < e;
}

// window of first-chunk workgroups: the tiles of the planes the particles live in (pm_set_tile_window)
static void tile_window(const pm_ctx *c, TileWork *w) {
    const int n = c->n;
    w->ntiles = (unsigned)(((int64_t)n * n * n) >> (PM_LI + PM_LJ + PM_LK));
    w->nlaunch = w->ntiles, w->ti0 = 0, w->lnti = 0, w->lT = 0, w->boff = 0;
    const int nti = c->tile_nx >> PM_LI, T = n >> PM_LJ;
    if (c->tile_nx < n && nti >= 1 && (nti & (nti - 1)) == 0 && (c->tile_x_lo % PM_TI) == 0 && (c->tile_nx % PM_TI) == 0) {
        w->ti0 = c->tile_x_lo >> PM_LI;
        w->lnti = __builtin_ctz((unsigned)nti);
        w->lT = __builtin_ctz((unsigned)T);
        w->nlaunch = (unsigned)nti * (unsigned)T * (unsigned)T;
    }
}

extern "C" int pm_set_tile_window(pm_ctx *c, int x_lo, int nx) {
    PM_REQUIRE(c && nx >= 1 && nx <= c->n && x_lo >= 0 && x_lo + nx <= c->n, "bad plane window");
    if (c->tile_x_lo != x_lo || c->tile_nx != nx) c->lists_pos = nullptr;  // prepared ranges belong to the old window
    c->tile_x_lo = x_lo, c->tile_nx = nx;
    return PM_OK;
}

// Buffers of the incremental re-sort (pm_gradient_interp_kick_rebucket): every tile's new population, the
// destination histogram of every workgroup, the fill counters, the widened counts and the two tile-offset tables.
struct RbBufs {
    unsigned *tile_count, *wg_hist, *fill;
    unsigned char *dcode;
    int64_t *cnt, *table[2];
    void *scan_tmp;
    size_t scan_tmp_bytes;
};

static int rebucket_bufs(pm_ctx *c, unsigned ntiles, int64_t np, RbBufs *b) {
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t maxblocks = (size_t)ntiles + (size_t)(np / PM_RANGE_CHUNK) + 2;  // >= nlaunch + np / PM_RANGE_CHUNK + 1
    const size_t t32 = up(((size_t)ntiles + 1) * sizeof(unsigned)), t64 = up(((size_t)ntiles + 1) * sizeof(int64_t));
    const size_t w32 = up(maxblocks * 27 * sizeof(unsigned));
    size_t tmp = 0;
    int64_t *nullp = nullptr;
    PM_HIP(rocprim::exclusive_scan(nullptr, tmp, nullp, nullp, (int64_t)0, (size_t)ntiles + 1, rocprim::plus<int64_t>(),
                                   c->stream));
    const size_t dbytes = up((size_t)np);
    const size_t need = 2 * t32 + w32 + 3 * t64 + up(tmp) + dbytes;
    if (c->rb_buf_bytes < need) {
        // Growing (a slab's particle count crept past the headroom): the tile-offset table of the stream the last
        // gather wrote lives in here and may already be some caller's `work.start` -- so callers size this buffer
        // BEFORE they take pointers into it (tile_lists_impl, pm_rebucket_prepare, the fused kick, the gather), and
        // the head of the buffer (everything whose place depends on the tiles only: counts, fill, both tables) moves
        // across with its contents, so a table on file stays on file.  The per-workgroup histograms and the codes
        // are recounted by whoever asked for the larger buffer.
        const size_t roomy = need + need / 16;  // (a slab's particle count creeps up and down with the migration)
        void *grown = nullptr;
        PM_HIP(hipMalloc(&grown, roomy));
        char *old = (char *)c->rb_buf;
        if (old) {
            const size_t head = 2 * t32 + 3 * t64;
            if (c->sorted_table && c->sorted_ntiles == ntiles && c->rb_buf_bytes >= head) {
                PM_HIP(hipMemcpyAsync(grown, old, head, hipMemcpyDeviceToDevice, c->stream));
                c->sorted_table = (int64_t *)((char *)grown + ((char *)c->sorted_table - old));
            } else {
                c->sorted_pos = nullptr, c->sorted_table = nullptr;
            }
            if ((const char *)c->lists_ptr[0] >= old && (const char *)c->lists_ptr[0] < old + c->rb_buf_bytes)
                c->lists_pos = nullptr;  // prepared ranges that read the old table: void (callers prepare afterwards)
            PM_HIP(hipStreamSynchronize(c->stream));
            PM_HIP(hipFree(old));
        }
        c->rb_buf = grown, c->rb_buf_bytes = roomy, c->stay_valid = 0;
    }
    // the parts whose size depends on the tiles only come first: the tile-offset tables stay where they are when the
    // particle count changes (a slab's migration), so the table of one step can be read while the next is written
    char *p = (char *)c->rb_buf;
    b->tile_count = (unsigned *)p, b->fill = (unsigned *)(p + t32);
    b->cnt = (int64_t *)(p + 2 * t32);
    b->table[0] = (int64_t *)(p + 2 * t32 + t64), b->table[1] = (int64_t *)(p + 2 * t32 + 2 * t64);
    b->scan_tmp = p + 2 * t32 + 3 * t64, b->scan_tmp_bytes = tmp;
    b->wg_hist = (unsigned *)(p + 2 * t32 + 3 * t64 + up(tmp));
    b->dcode = (unsigned char *)(p + 2 * t32 + 3 * t64 + up(tmp) + w32);
    return PM_OK;
}

static int tile_ranges(pm_ctx *c, const float *pos, int64_t np, TileWork *out, unsigned *out_blocks) {
    const int n = c->n;
    const unsigned ntiles = (unsigned)(((int64_t)n * n * n) >> (PM_LI + PM_LJ + PM_LK));
    const size_t bytes = ((size_t)(ntiles + 1) * sizeof(int64_t) + 255) & ~(size_t)255;
    size_t tmp_bytes = 0, tmp2 = 0;
    int64_t *nullp = nullptr;
    PM_HIP(rocprim::inclusive_scan(nullptr, tmp_bytes, nullp, nullp, (size_t)ntiles + 1, rocprim::maximum<int64_t>(),
                                   c->stream));
    PM_HIP(rocprim::exclusive_scan(nullptr, tmp2, nullp, nullp, (int64_t)0, (size_t)ntiles + 1,
                                   rocprim::plus<int64_t>(), c->stream));
    if (tmp2 > tmp_bytes) tmp_bytes = tmp2;
    const size_t need = 4 * bytes + tmp_bytes;
    if (c->tile_tmp_bytes < need) {
        if (c->tile_tmp) PM_HIP(hipFree(c->tile_tmp));
        c->tile_tmp = nullptr;
        c->tile_tmp_bytes = 0;
        PM_HIP(hipMalloc(&c->tile_tmp, need));
        c->tile_tmp_bytes = need;
    }
    char *base = (char *)c->tile_tmp;
    int64_t *raw = (int64_t *)base, *start = (int64_t *)(base + bytes), *extra = (int64_t *)(base + 2 * bytes),
            *xstart = (int64_t *)(base + 3 * bytes);
    void *scan_tmp = base + 4 * bytes;
    const unsigned tb = (ntiles + 1 + BLK - 1) / BLK;
    if (pos && c->sorted_pos == pos && c->sorted_np == np && c->sorted_ntiles == ntiles && c->rb_buf && c->sorted_table) {
        // this stream was written tile by tile (pm_gradient_interp_kick_rebucket): its offsets are on file
        start = c->sorted_table;
    } else {
        hipLaunchKernelGGL(tile_bounds_kernel, dim3(tb), dim3(BLK), 0, c->stream, pos, np, n, ntiles, raw);
        PM_LAUNCH_CHECK();
        PM_HIP(rocprim::inclusive_scan(scan_tmp, tmp_bytes, raw, start, (size_t)ntiles + 1, rocprim::maximum<int64_t>(),
                                       c->stream));
    }
    hipLaunchKernelGGL(tile_extra_kernel, dim3(tb), dim3(BLK), 0, c->stream, start, ntiles, extra);
    PM_LAUNCH_CHECK();
    PM_HIP(rocprim::exclusive_scan(scan_tmp, tmp_bytes, extra, xstart, (int64_t)0, (size_t)ntiles + 1,
                                   rocprim::plus<int64_t>(), c->stream));
    tile_window(c, out);
    out->start = start, out->xstart = xstart;
    // sum over tiles of ceil(len/CHUNK) <= nlaunch + np/CHUNK: surplus workgroups exit at once
    const int64_t blocks = (int64_t)out->nlaunch + np / PM_RANGE_CHUNK + 1;
    if (blocks > 2147483647LL) {
        pm_set_error("too many particles for one launch");
        return PM_EINVAL;
    }
    *out_blocks = (unsigned)blocks;
    return PM_OK;
}

// ---- pm_tile_lists: ranges + incoming-straggler lists, prepared once per set of positions ---------

// The list is appended to through PM_NSUB counters, each with its own segment of the buffers and its own
// 128-byte line: appends of different workgroups mostly hit different L2 channels instead of queueing
// on one address (~12 ns per same-address atomic: 6 ms for the 524288 workgroups of a 1024^3 step).
constexpr int PM_NSUB = 64, PM_SUB_STRIDE = 32;

__device__ __forceinline__ unsigned *sub_counter(unsigned *counts) {
    return counts + (blockIdx.x % PM_NSUB) * PM_SUB_STRIDE;
}

// (with `dc.dcode`: also the destination counts of pm_rebucket_prepare, in the same pass over the positions)
struct ScanCount {
    unsigned char *codes;  // NULL: off
    unsigned *tile_count, *wg_hist;
};

__global__ __launch_bounds__(BLK) void straggler_scan_kernel(const float *__restrict__ pos, TileWork work, int n,
                                                             unsigned *__restrict__ keys, unsigned *__restrict__ idx,
                                                             unsigned subcap, unsigned *__restrict__ counts, ScanCount sc) {
    unsigned key;
    int64_t s, e;
    if (!tile_work_item(work, key, s, e)) return;
    __shared__ unsigned s_k[PM_SCAN_STAGE], s_p[PM_SCAN_STAGE];  // 32 KB
    __shared__ unsigned s_cnt, s_base;
    __shared__ unsigned s_hist[27];
    const int tid = threadIdx.x;
    const float nf = (float)n;
    const unsigned seg0 = (blockIdx.x % PM_NSUB) * subcap;
    const int Ti = n >> PM_LI, Tj = n >> PM_LJ, Tk = n >> PM_LK;
    const bool windowed = work.nlaunch != work.ntiles;
    const int w_lo = work.ti0, w_hi = work.ti0 + (1 << work.lnti);
    if (tid == 0) s_cnt = 0;
    if (tid < 27) s_hist[tid] = 0;
    __syncthreads();
    for (int64_t p0 = s; p0 < e; p0 += BLK) {  // (whole waves in step: the ballot below)
        const int64_t p = p0 + tid;
        int d = -1;
        if (p < e) {
            const unsigned k = tile_key_of_pos(pos + 3 * p, nf, n);
            if (k != key) {
                const unsigned slot = atomicAdd(&s_cnt, 1u);
                if (slot < (unsigned)PM_SCAN_STAGE) {
                    s_k[slot] = k, s_p[slot] = (unsigned)p;
                } else {  // past the staging area (a range of strangers: an unsorted stream): straight to the global list
                    const unsigned g = atomicAdd(sub_counter(counts), 1u);
                    if (g < subcap) keys[seg0 + g] = k, idx[seg0 + g] = (unsigned)p;
                }
            }
            if (sc.codes) {
                // the particle's destination code, made safe (a byte the caller owns) and kept inside the window of
                // tiles this rank serves (a particle filed under an outside tile would never be served: it is about
                // to leave the rank anyway)
                const TileId ta = tile_of_key(k);
                d = min((int)sc.codes[p], 26);
                if (windowed) {
                    const int ti = (ta.ti + d / 9 - 1 + Ti) % Ti;
                    if (ti < w_lo || ti >= w_hi) d = 13;
                }
                if (d != (int)sc.codes[p]) sc.codes[p] = (unsigned char)d;
                if (k != key) {  // served by the tile it is in, from that tile's list: counted there one by one
                    atomicAdd(sc.tile_count + tile_key((ta.ti + d / 9 - 1 + Ti) % Ti, (ta.tj + (d / 3) % 3 - 1 + Tj) % Tj,
                                                       (ta.tk + d % 3 - 1 + Tk) % Tk), 1u);
                    d = -1;
                }
            }
        }
        if (sc.codes) {  // nearly everybody stays: one LDS atomic per wave for those (64 lanes on one address serialise)
            const unsigned long long m = __ballot(d == 13);
            if (m && (tid & 63) == __ffsll((long long)m) - 1) atomicAdd(&s_hist[13], (unsigned)__popcll(m));
            if (d >= 0 && d != 13) atomicAdd(&s_hist[d], 1u);
        }
    }
    __syncthreads();
    if (sc.codes && tid < 27) {
        const unsigned cnt_d = s_hist[tid];
        const TileId t = tile_of_key(key);
        sc.wg_hist[(size_t)blockIdx.x * 27 + tid] = cnt_d;
        if (cnt_d)
            atomicAdd(sc.tile_count + tile_key((t.ti + tid / 9 - 1 + Ti) % Ti, (t.tj + (tid / 3) % 3 - 1 + Tj) % Tj,
                                               (t.tk + tid % 3 - 1 + Tk) % Tk), cnt_d);
    }
    const unsigned cnt = min(s_cnt, (unsigned)PM_SCAN_STAGE);
    if (cnt == 0) return;
    if (tid == 0) s_base = atomicAdd(sub_counter(counts), cnt);
    __syncthreads();
    const unsigned base = s_base, seg = seg0;
    for (unsigned i = tid; i < cnt; i += BLK)
        if (base + i < subcap) keys[seg + base + i] = s_k[i], idx[seg + base + i] = s_p[i];
}

struct SubLists {
    unsigned count[PM_NSUB], offset[PM_NSUB];
};

// segments -> one contiguous list
__global__ __launch_bounds__(BLK) void list_compact_kernel(const unsigned *__restrict__ k_seg,
                                                           const unsigned *__restrict__ i_seg, unsigned subcap,
                                                           SubLists sub, unsigned *__restrict__ k_out,
                                                           unsigned *__restrict__ i_out) {
    const unsigned k = blockIdx.y, cnt = sub.count[k], off = sub.offset[k];
    for (unsigned i = blockIdx.x * BLK + threadIdx.x; i < cnt; i += gridDim.x * BLK)
        k_out[off + i] = k_seg[k * subcap + i], i_out[off + i] = i_seg[k * subcap + i];
}

__global__ __launch_bounds__(BLK) void list_bounds_kernel(const unsigned *__restrict__ keys, unsigned count,
                                                          unsigned ntiles, unsigned *__restrict__ lstart) {
    const unsigned t = blockIdx.x * BLK + threadIdx.x;
    if (t > ntiles) return;
    unsigned lo = 0, hi = count;  // first entry with key >= t
    while (lo < hi) {
        const unsigned mid = (lo + hi) >> 1;
        if (keys[mid] < t)
            lo = mid + 1;
        else
            hi = mid;
    }
    lstart[t] = lo;
}

static void tile_lists_drop(pm_ctx *c) {
    c->lists_pos = nullptr, c->lists_np = 0, c->lists_use = 0, c->lists_sticky = 0, c->stay_valid = 0;
}

// device buffers of the straggler lists: unsorted and sorted (key, index) pairs, per-tile offsets
struct ListBufs {
    unsigned *k_in, *i_in, *k_out, *i_out, *lstart, *counts;
    void *sort_tmp;
    size_t sort_tmp_bytes, cap, subcap;
    int bits;
};

static int list_bufs(pm_ctx *c, int64_t np, unsigned ntiles, ListBufs *b) {
    c->last_np = np;
    // room for every particle and some imbalance (20 bytes each: 21 GB at 1024^3 of 288): the sort and the list pass only cost what
    // the actual number of stragglers makes them cost, and there is no cliff when a sort has aged badly; a
    // sub-list can still overflow when the stragglers bunch up in few workgroups (then: order-independent kernels)
    b->subcap = (size_t)(np / PM_NSUB + np / (4 * PM_NSUB) + 2 * PM_RANGE_CHUNK);
    b->cap = b->subcap * PM_NSUB;
    const size_t off_l = ((b->cap * 4 * sizeof(unsigned)) + 255) & ~(size_t)255;
    const size_t off_c = off_l + ((((size_t)ntiles + 1) * sizeof(unsigned) + 255) & ~(size_t)255);
    const size_t off_t = off_c + PM_NSUB * PM_SUB_STRIDE * sizeof(unsigned);
    unsigned *nu = nullptr;
    b->bits = 32 - __builtin_clz(ntiles > 1 ? ntiles - 1 : 1);
    b->sort_tmp_bytes = 0;
    PM_HIP(rocprim::radix_sort_pairs(nullptr, b->sort_tmp_bytes, nu, nu, nu, nu, b->cap, 0, b->bits, c->stream));
    const size_t need = off_t + b->sort_tmp_bytes;
    if (c->list_buf_bytes < need) {
        if (c->list_buf) PM_HIP(hipFree(c->list_buf));
        c->list_buf = nullptr, c->list_buf_bytes = 0;
        PM_HIP(hipMalloc(&c->list_buf, need));
        c->list_buf_bytes = need;
    }
    b->k_in = (unsigned *)c->list_buf, b->i_in = b->k_in + b->cap, b->k_out = b->i_in + b->cap, b->i_out = b->k_out + b->cap;
    b->lstart = (unsigned *)((char *)c->list_buf + off_l);
    b->counts = (unsigned *)((char *)c->list_buf + off_c);
    b->sort_tmp = (char *)c->list_buf + off_t;
    return PM_OK;
}

// reads the straggler counts back (stream sync), compacts the segments and groups the entries by tile;
// *listed = entries grouped, *complete = every straggler is among them (no segment overflowed)
static int list_group(pm_ctx *c, const ListBufs &b, unsigned ntiles, const unsigned **idx, unsigned *listed,
                      int *complete) {
    unsigned h_counts[PM_NSUB * PM_SUB_STRIDE];
    if (int rb_rc = pm_read_back(c, h_counts, b.counts, sizeof(h_counts))) return rb_rc;
    SubLists sub;
    unsigned cnt = 0, biggest = 0;
    *complete = 1;
    c->last_stragglers = 0;
    for (int k = 0; k < PM_NSUB; ++k) {
        unsigned ck = h_counts[k * PM_SUB_STRIDE];
        c->last_stragglers += ck;
        if (ck > b.subcap) ck = (unsigned)b.subcap, *complete = 0;
        sub.count[k] = ck, sub.offset[k] = cnt;
        cnt += ck;
        if (ck > biggest) biggest = ck;
    }
    const unsigned *keys = b.k_out;
    *idx = b.i_out;
    if (cnt > 0) {
        const unsigned gx = (biggest + BLK * 8 - 1) / (BLK * 8);
        hipLaunchKernelGGL(list_compact_kernel, dim3(gx ? gx : 1, PM_NSUB), dim3(BLK), 0, c->stream, b.k_in, b.i_in,
                           (unsigned)b.subcap, sub, b.k_out, b.i_out);
        PM_LAUNCH_CHECK();
    }
    if (cnt > 1) {  // sort back into the (now free) segment buffers
        size_t tmp = b.sort_tmp_bytes;
        PM_HIP(rocprim::radix_sort_pairs(b.sort_tmp, tmp, b.k_out, b.k_in, b.i_out, b.i_in, (size_t)cnt, 0, b.bits,
                                         c->stream));
        keys = b.k_in, *idx = b.i_in;
    }
    hipLaunchKernelGGL(list_bounds_kernel, dim3((ntiles + 1 + BLK - 1) / BLK), dim3(BLK), 0, c->stream, keys, cnt, ntiles,
                       b.lstart);
    PM_LAUNCH_CHECK();
    *listed = cnt;
    c->last_complete = *complete;
    return PM_OK;
}

// Outcome of the most recent straggler scan on this context (pm_tile_lists, pm_kick_drift_wrap_deposit[_slab]):
// how many of the np particles were outside the tile they are filed under, and whether the lists hold them
// all (beyond one in four the kernels serve stragglers themselves, slowly).  *stragglers = -1 before any scan.
// A caller uses it to decide when to re-sort (utils.reorder_particles).
extern "C" int pm_tile_lists_info(pm_ctx *c, int64_t *stragglers, int64_t *np, int *complete) {
    PM_REQUIRE(c && stragglers && np && complete, "NULL argument");
    *stragglers = c->last_stragglers, *np = c->last_np, *complete = c->last_complete;
    return PM_OK;
}

static int tile_lists_impl(pm_ctx *c, const float *d_pos, int64_t np, unsigned char *d_codes) {
    PM_REQUIRE(c && np >= 0, "bad argument");
    if (!d_codes && c->lists_sticky && d_pos && c->lists_pos == d_pos && c->lists_np == np) {
        c->lists_sticky = 0;  // prepared by pm_kick_drift_wrap_deposit for exactly these positions
        return PM_OK;
    }
    tile_lists_drop(c);
    if (!d_pos || np == 0 || c->n < 16) return PM_OK;
    TileWork work;
    unsigned blocks = 0;
    int rc = 0;
    RbBufs rb;
    if (d_codes && np < 4294967296LL) {  // (before any pointer into the buffer is taken: rebucket_bufs may move it)
        rc = rebucket_bufs(c, (unsigned)(((int64_t)c->n * c->n * c->n) >> (PM_LI + PM_LJ + PM_LK)), np, &rb);
        if (rc) return rc;
    }
    rc = tile_ranges(c, d_pos, np, &work, &blocks);
    if (rc) return rc;
    const unsigned ntiles = work.ntiles;
    c->lists_ptr[0] = work.start, c->lists_ptr[1] = work.xstart, c->lists_ptr[2] = c->lists_ptr[3] = nullptr;
    c->lists_blocks = blocks;
    int use = 0;
    if (np < 4294967296LL) {
        ListBufs b;
        rc = list_bufs(c, np, ntiles, &b);
        if (rc) return rc;
        ScanCount sc{nullptr, nullptr, nullptr};
        if (d_codes) {  // the destination counts of the re-bucketing gather in the same pass (pm_tile_lists_codes)
            PM_HIP(hipMemsetAsync(rb.tile_count, 0, ((size_t)ntiles + 1) * sizeof(unsigned), c->stream));
            sc.codes = d_codes, sc.tile_count = rb.tile_count, sc.wg_hist = rb.wg_hist;
        }
        PM_HIP(hipMemsetAsync(b.counts, 0, PM_NSUB * PM_SUB_STRIDE * sizeof(unsigned), c->stream));
        hipLaunchKernelGGL(straggler_scan_kernel, dim3(blocks), dim3(BLK), 0, c->stream, d_pos, work, c->n, b.k_in, b.i_in,
                           (unsigned)b.subcap, b.counts, sc);
        PM_LAUNCH_CHECK();
        const unsigned *idx = nullptr;
        unsigned listed = 0;
        rc = list_group(c, b, ntiles, &idx, &listed, &use);
        if (rc) return rc;
        c->lists_ptr[2] = idx, c->lists_ptr[3] = b.lstart;
    }
    c->lists_pos = d_pos, c->lists_np = np, c->lists_use = use;
    if (d_codes && use) c->stay_valid = 1, c->stay_codes = d_codes;
    return PM_OK;
}

extern "C" int pm_tile_lists(pm_ctx *c, const float *d_pos, int64_t np) { return tile_lists_impl(c, d_pos, np, nullptr); }

// pm_tile_lists + pm_rebucket_prepare in one pass over the positions (the scan that finds the stragglers also counts
// every particle's destination): what a slab step calls after its migration.
extern "C" int pm_tile_lists_codes(pm_ctx *c, const float *d_pos, int64_t np, unsigned char *d_codes) {
    PM_REQUIRE(c && d_pos && d_codes && np > 0 && np < 4294967296LL && c->n >= 16, "bad argument");
    return tile_lists_impl(c, d_pos, np, d_codes);
}

// ranges (+ lists) for a tile-range launch: the prepared ones if they are for these positions
static int tile_work_for(pm_ctx *c, const float *pos, int64_t np, TileWork *work, TileLists *lists, unsigned *blocks) {
    if (c->lists_pos == pos && c->lists_np == np && pos) {
        work->start = (const int64_t *)c->lists_ptr[0], work->xstart = (const int64_t *)c->lists_ptr[1];
        tile_window(c, work);
        lists->idx = (const unsigned *)c->lists_ptr[2], lists->lstart = (const unsigned *)c->lists_ptr[3];
        lists->use = c->lists_use;
        *blocks = c->lists_blocks;
        return PM_OK;
    }
    tile_lists_drop(c);  // tile_ranges reuses the table the prepared ranges live in
    lists->idx = lists->lstart = nullptr, lists->use = 0;
    return tile_ranges(c, pos, np, work, blocks);
}

// Rolling depth-4 prefetch of the positions a thread will process (r, r+BLK, r+2BLK, ... of its range): the
// particle loop stays rolled (register budget) while four loads per lane are always in flight.
// Indices are 32-bit and relative to the range's first particle (a workgroup's range is at most
// PM_RANGE_CHUNK long): the range base stays in SGPRs and a lane's address is one 32-bit multiply, where
// absolute 64-bit indices cost a quarter-rate v_mad_u64_u32 chain per load.
struct PosQueue {
    float x[4], y[4], z[4];
    const float *base;
    unsigned last;
    __device__ __forceinline__ void load(int slot, unsigned r) {
        r = min(r, last);  // clamp: harmless reload of the range's last particle
        const float *pp = base + 3u * r;
        // read-once stream: nontemporal, so that it does not push the tile halos neighbours re-read out of the L2
        x[slot] = __builtin_nontemporal_load(pp), y[slot] = __builtin_nontemporal_load(pp + 1),
        z[slot] = __builtin_nontemporal_load(pp + 2);
    }
    __device__ __forceinline__ void prime(const float *range_base, unsigned r0, unsigned count) {
        base = range_base, last = count - 1u;
        load(0, r0), load(1, r0 + BLK), load(2, r0 + 2 * BLK), load(3, r0 + 3 * BLK);
    }
    template <int N>
    __device__ __forceinline__ void prime_slots(const float *range_base, unsigned r0, unsigned count) {  // the first N only
        base = range_base, last = count - 1u;
#pragma unroll
        for (int u = 0; u < N; ++u) load(u, r0 + u * BLK);
    }
    __device__ __forceinline__ void pop(unsigned r, float &px, float &py, float &pz) {  // r = particle being popped
        px = x[0], py = y[0], pz = z[0];
        x[0] = x[1], y[0] = y[1], z[0] = z[1];
        x[1] = x[2], y[1] = y[2], z[1] = z[2];
        x[2] = x[3], y[2] = y[3], z[2] = z[3];
        load(3, r + 4 * BLK);
    }
};

// the same for the velocities the fused kick updates (loaded at the top of an iteration they would be
// waited for at its end, one memory latency per particle; four iterations ahead they are simply there)
struct VelQueue {
    PosQueue q;
    __device__ __forceinline__ void prime(const float *range_base, unsigned r0, unsigned count) {
        q.prime(range_base, r0, count);
    }
    __device__ __forceinline__ void pop(unsigned r, float v[3]) { q.pop(r, v[0], v[1], v[2]); }
};

template <int SCHEME>
__global__ __launch_bounds__(BLK) void deposit_ranges_kernel(const float *__restrict__ pos, TileWork work,
                                                             TileLists lists, pm_geom geo, float *__restrict__ rho) {
    unsigned key;
    int64_t s, e;
    const bool has_range = tile_work_item(work, key, s, e);
    unsigned ls = 0, le = 0;  // incoming stragglers (first workgroup of the tile only)
    if (lists.use && blockIdx.x < work.nlaunch) ls = lists.lstart[key], le = lists.lstart[key + 1];
    if (!has_range && ls == le) return;
    __shared__ pm_acc_t tile[PM_TILE];  // 25.9 KB
    const int tid = threadIdx.x;
    const int n = geo.n;
    const float nf = (float)n;
    const TileId t = tile_of_key(key);
    PosQueue q;
    const unsigned cnt = has_range ? (unsigned)(e - s) : 0u;
    if (has_range) q.prime(pos + 3 * s, tid, cnt);
    for (int c = tid; c < PM_TILE; c += BLK) tile[c] = 0;
    __syncthreads();
    if (has_range) {
#pragma unroll 1
        for (unsigned r = tid; r < cnt; r += BLK) {
            float x, y, z;
            q.pop(r, x, y, z);
            const float xp = x * nf, yp = y * nf, zp = z * nf;
            const int i = cell_of(xp, n), j = cell_of(yp, n), k = cell_of(zp, n);
            if (in_tile(i, j, k, t)) {
                deposit_lds<SCHEME>(xp, yp, zp, i, j, k, t, tile);
            } else if (!lists.use) {  // filed under another tile (drifted / unsorted) and nobody else serves it
                deposit_global<SCHEME>(x, y, z, geo, rho);
            }
        }
    }
#pragma unroll 1
    for (unsigned l = ls + tid; l < le; l += BLK) {
        const int64_t p = lists.idx[l];
        const float x = pos[3 * p], y = pos[3 * p + 1], z = pos[3 * p + 2];
        const float xp = x * nf, yp = y * nf, zp = z * nf;
        const int i = cell_of(xp, n), j = cell_of(yp, n), k = cell_of(zp, n);
        if (in_tile(i, j, k, t))
            deposit_lds<SCHEME>(xp, yp, zp, i, j, k, t, tile);
        else
            deposit_global<SCHEME>(x, y, z, geo, rho);
    }
    __syncthreads();
    for (int c = tid; c < PM_TILE; c += BLK) {
        const pm_acc_t v = tile[c];
        if (v != 0) unsafeAtomicAdd(rho + tile_cell_global(c, t, geo), weight_of_acc(v));
    }
}

// ------------------------------------------------------------------------------------------------
// First leapfrog half + deposit in one kernel (integration.py:250-258 + mesh.py:2468): the streaming
// update is HBM-bound with the LDS idle, the deposit is LDS-atomic-bound with HBM idle; one workgroup
// per tile range does both -- v += -half_dt*a; x += dt*v; wrap; store; 27 LDS adds -- and the
// position is read once.  Ranges are those of the positions BEFORE the drift; a particle that ends up
// outside the tile it is filed under goes to the straggler list (LDS-aggregated append), which a
// second, small kernel deposits tile by tile after a sort by tile key.  The lists are left prepared
// (pm_tile_lists semantics) for the gather of the same step.

// stragglers staged in LDS per workgroup before the global append (beyond it: one global atomic per straggler).  256
// (was 1024): with the stream re-bucketed every step a workgroup has a handful, and the 6 KB let a fifth workgroup
// onto the CU (28 KB instead of 34: 15.15 -> 14.4 ms for the kernel at 1024^3)
constexpr int PM_SCAP = 256;
// Halo buffer of the fused kick + drift + deposit (pmstep.h: pm_kick_drift_wrap_deposit_khalo), three parts:
//   K  [tile column (ti, tj)][li 0..9][low | high][tk][lj 0..17]   the k-halo cells (lk = 0, 17) of every tile
//   Hi [tile column (ti, tj)][low | high][lj 0..17][k 0..n)         the i-halo planes (li = 0, 9), lk = 1..16
//   Hj [tile column (ti, tj)][low | high][li 1..8][k 0..n)          the j-halo rows (lj = 0, 17), li and lk inside
// Hi and Hj are whole lines along k, so that the FFT's z pass adds them to the density lines it loads.
// On a slab (multi-GPU) the buffer holds the rows of tiles of the rank's window only (ti0 .. ti0 + nti - 1, tile
// columns numbered from ti0), and the halo cells that leave the window through its two x faces are not kept in it:
// they belong to ghost planes of the density slab, which the neighbours receive (halo_dest below).
constexpr int PM_KHALO_TILE = 2 * PM_HI * PM_HJ;  // floats per tile in K
struct HaloBuf {
    float *k, *hi, *hj;
    int ti0, nti;   // rows of tiles the buffer covers
    int windowed;   // nti < n / 8: a slab
};
__host__ __device__ inline int64_t halo_k_floats(int n, int nti) { return (int64_t)nti * (n >> PM_LJ) * (n >> PM_LK) * PM_KHALO_TILE; }
__host__ __device__ inline int64_t halo_i_floats(int n, int nti) { return (int64_t)nti * (n >> PM_LJ) * 2 * PM_HJ * n; }
__host__ __device__ inline int64_t halo_j_floats(int n, int nti) { return (int64_t)nti * (n >> PM_LJ) * 2 * PM_TI * n; }
__host__ __device__ inline int64_t halo_floats(int n, int nti) { return halo_k_floats(n, nti) + halo_i_floats(n, nti) + halo_j_floats(n, nti); }
static HaloBuf halo_parts(float *base, int n, int ti0 = 0, int nti = -1) {
    HaloBuf h;
    if (nti < 0) nti = n >> PM_LI;
    h.k = base, h.hi = base ? base + halo_k_floats(n, nti) : nullptr, h.hj = base ? h.hi + halo_i_floats(n, nti) : nullptr;
    h.ti0 = ti0, h.nti = nti, h.windowed = nti < (n >> PM_LI);
    return h;
}

// where cell c of the haloed tile t lives: the tile's own cells in rho, its halo cells in the tile's entries of the
// halo buffer (every address has exactly one tile that maps to it)
__device__ __forceinline__ float *halo_cell(int c, const TileId &t, const HaloBuf &halo, const pm_geom &geo, float *rho) {
    const int n_ = geo.n, T_ = n_ >> PM_LJ;
    const int lij = c / PM_HK, lk = c - lij * PM_HK, li = lij / PM_HJ, lj = lij - li * PM_HJ;
    const int64_t col = (int64_t)(t.ti - halo.ti0) * T_ + t.tj;
    if (lk == 0 || lk == PM_HK - 1) return halo.k + ((((col * PM_HI + li) * 2 + (lk ? 1 : 0)) * T_ + t.tk) * PM_HJ + lj);
    if (li == 0 || li == PM_HI - 1) return halo.hi + (((col * 2 + (li ? 1 : 0)) * PM_HJ + lj) * n_ + t.tk * PM_TK + lk - 1);
    if (lj == 0 || lj == PM_HJ - 1)
        return halo.hj + (((col * 2 + (lj ? 1 : 0)) * PM_TI + (li - 1)) * n_ + t.tk * PM_TK + lk - 1);
    return rho + pm_cell(geo, t.ti * PM_TI - 1 + li, t.tj * PM_TJ - 1 + lj, t.tk * PM_TK - 1 + lk);
}

// On a slab: does cell c of the haloed tile t have an address the halo buffer scheme owns -- the tile is one of the
// window's, and the cell does not leave the window through an x face?  If not, the contribution goes to the cell
// itself with an atomic (ghost planes of the slab are cleared beforehand; own planes hold the owners' stores of an
// earlier launch, or -- the owners' own outward halo cells -- are ghost planes).
__device__ __forceinline__ bool halo_owns(int c, const TileId &t, const HaloBuf &halo) {
    if (!halo.windowed) return true;
    const int r = t.ti - halo.ti0;
    if (r < 0 || r >= halo.nti) return false;
    const int li = c / (PM_HJ * PM_HK);
    return !((li == 0 && r == 0) || (li == PM_HI - 1 && r == halo.nti - 1));
}
// where a contribution to cell c of tile t is ADDED (atomically) under the halo buffer scheme
__device__ __forceinline__ float *halo_dest(int c, const TileId &t, const HaloBuf &halo, const pm_geom &geo, float *rho) {
    return halo_owns(c, t, halo) ? halo_cell(c, t, halo, geo, rho) : rho + tile_cell_global(c, t, geo);
}

// Destination tile of the incremental re-sort.  A particle is FILED under the tile it is expected to be in after
// the next drift, x + dtp * v with v = v_half - (dt + dtp) / 2 * a_old: the velocity the next drift will use, with
// the acceleration of this step's first kick standing in for the one the gather is about to compute (dtp: the
// caller's guess of the next step, 0 = file under the current tile).  The fused kick + drift + deposit has all of
// that in registers; it records the decision as one byte per particle, which the gather obeys.  The displacement
// is clamped to one cell (the Courant criterion's bound, integration.py:324-326), so the destination is the tile
// the particle is in or one of its 26 neighbours: code d = 9 (dti + 1) + 3 (dtj + 1) + (dtk + 1), 13 = stays.
// Measured (512^3, tools/exp_predict_tiles.py): 0.05-0.3 % of the particles end up outside the tile they are
// filed under, where an exact sort sees every tile crosser of the step (4 % when the fastest particle moves one
// cell) and the velocity alone predicts 0.25-1.1 %.
struct DestCode {
    int d;
    unsigned key;  // tile key of the destination
};
__device__ __forceinline__ DestCode dest_of(const float x[3], const float v[3], float dtp, float nf, int n, const TileId &t) {
    const float h = 1.0f / nf;
    int q[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        float dx = dtp * v[m];
        dx = fminf(fmaxf(dx, -h), h);
        q[m] = (int)floorf((x[m] + dx) * nf);  // -1 .. n: not wrapped, so that the tile difference below is signed
    }
    const int di = (q[0] >> PM_LI) - t.ti, dj = (q[1] >> PM_LJ) - t.tj, dk = (q[2] >> PM_LK) - t.tk;
    DestCode r;
    // (a particle that is not in tile t at all -- only if positions were changed behind the lists -- stays put)
    const bool ok = di >= -1 && di <= 1 && dj >= -1 && dj <= 1 && dk >= -1 && dk <= 1;
    r.d = ok ? (di + 1) * 9 + (dj + 1) * 3 + (dk + 1) : 13;
    const int Ti = n >> PM_LI, Tj = n >> PM_LJ, Tk = n >> PM_LK;
    const int ti = ok ? (t.ti + di + Ti) % Ti : t.ti, tj = ok ? (t.tj + dj + Tj) % Tj : t.tj, tk = ok ? (t.tk + dk + Tk) % Tk : t.tk;
    r.key = tile_key(ti, tj, tk);
    return r;
}
__device__ __forceinline__ TileId tile_of_cell(int i, int j, int k) { return TileId{i >> PM_LI, j >> PM_LJ, k >> PM_LK}; }

// what the fused kick + drift + deposit records for the re-sort: every tile's new population, and per workgroup
// how many of the particles it serves go to each of the 27 destinations
struct DestCount {
    unsigned *tile_count;  // [ntiles] cleared (NULL: off)
    unsigned *wg_hist;     // [workgroups][27]
    unsigned char *dcode;  // [np] destination code of every particle, relative to the tile it is in
    float dtp;             // guess of the next dt
    float kick;            // -(dt + dtp) / 2: the velocity of the next drift is about v + kick * a (a: the
                           // acceleration this step's first kick used -- the new one is not known yet)
};

// slab decomposition: particles whose new x leaves [x_lo, x_hi) are listed for the migration (idx == NULL: off)
struct Leavers {
    float x_lo, x_hi;
    int64_t *idx;
    int cap;
    int *count;
};

// REB ("the kick re-buckets", struct Codes further down): what the last gather decided, and where the new stream goes
struct KickReb {
    const unsigned char *flag;  // [np] 13: the particle stays in the tile it is filed under
    const unsigned *wg_stay;    // [workgroups] stayers of the workgroup's range
    const unsigned *tile_stay;  // [ntiles] stayers of the tile
    const unsigned *inc_count;  // [ntiles] movers in the tile's incoming list
    const unsigned *segstart;   // [ntiles + 1] where the list lives in `entries`
    const unsigned *entries;    // old-stream indices of the movers, by destination tile
    const int64_t *newstart;    // [ntiles + 1] layout of the new stream: a tile's stayers, then its incoming list
    unsigned *fill;             // [ntiles] cleared: a tile's workgroups take their stayers' slices here
    float *pos_out, *vel_out;
    const unsigned *id;         // optional tags (NULL: none)
    unsigned *id_out;
};

template <int SCHEME, bool REB = false>
__global__ __launch_bounds__(BLK) void kdw_deposit_kernel(float *__restrict__ pos, float *__restrict__ vel,
                                                          const float *__restrict__ acc, float mhalf_dt, float dt,
                                                          TileWork work, pm_geom geo, float *__restrict__ rho,
                                                          unsigned *__restrict__ lkeys, unsigned *__restrict__ lidx,
                                                          unsigned subcap, unsigned *__restrict__ counts, Leavers lv,
                                                          HaloBuf halo, DestCount dc, KickReb kr = KickReb{}) {
    unsigned key;
    int64_t s, e;
    // Halo buffer: the first-chunk workgroup of a tile owns the tile's entries of K, Hi, Hj and the tile's own
    // cells of rho, and writes all of them with plain stores, empty tile or not (rho is not cleared first).
    const bool owner = halo.k && (blockIdx.x + work.boff) < work.nlaunch;
    TileId t0{0, 0, 0};
    if (owner) t0 = tile_of_key(tile_key_of_block(blockIdx.x + work.boff, work));
    // (a slab's outward halo cells: ghost planes, cleared by the launcher and added to)
    auto halo_store = [&](int c, float v) {
        if (halo_owns(c, t0, halo))
            *halo_cell(c, t0, halo, geo, rho) = v;
        else if (v != 0.f)
            unsafeAtomicAdd(rho + tile_cell_global(c, t0, geo), v);
    };
    // REB: the first workgroup of a tile also serves the tile's incoming list -- an empty range is no reason to leave
    const bool firstwg = (blockIdx.x + work.boff) < work.nlaunch;
    unsigned ninc = 0;
    if (!tile_work_item(work, key, s, e)) {
        if (REB && firstwg) {
            key = tile_key_of_block(blockIdx.x + work.boff, work);
            s = e = 0;
            ninc = min(kr.inc_count[key], kr.segstart[key + 1] - kr.segstart[key]);
        }
        if (ninc == 0) {
            if (owner)
                for (int c = threadIdx.x; c < PM_TILE; c += BLK) halo_store(c, 0.f);
            return;
        }
    } else if (REB && firstwg) {
        ninc = min(kr.inc_count[key], kr.segstart[key + 1] - kr.segstart[key]);
    }
    __shared__ pm_acc_t tile[PM_TILE];  // 25.9 KB
    __shared__ unsigned s_k[PM_SCAP], s_p[PM_SCAP];
    __shared__ unsigned s_cnt, s_base;
    __shared__ unsigned s_hist[27];  // served particles per destination (13: counted as the rest)
    __shared__ unsigned s_rbase, s_rslot;  // REB: the workgroup's slice of the tile's stayers, next free slot in it
    const int tid = threadIdx.x;
    const int n = geo.n;
    const float nf = (float)n;
    const TileId t = tile_of_key(key);
    for (int c = tid; c < PM_TILE; c += BLK) tile[c] = 0;
    if (tid == 0) s_cnt = 0;
    if (tid < 27) s_hist[tid] = 0;
    if (REB && tid == 0) {
        const unsigned mine = s < e ? kr.wg_stay[blockIdx.x + work.boff] : 0u;
        s_rbase = mine ? atomicAdd(kr.fill + key, mine) : 0u, s_rslot = 0;
    }
    // REB: the first BLK entries of the tile's incoming list are fetched NOW -- index, then (below, once it has arrived)
    // the particle's nine floats from wherever it sits in the old stream -- and ride in registers to the end of the
    // kernel: served there by loads of their own, the list cost the workgroup two exposed memory latencies with
    // nothing left to hide them behind
    int64_t pinc = -1;
    if (REB && (unsigned)tid < ninc) pinc = kr.entries[kr.segstart[key] + tid];
    __syncthreads();
    float xi[3] = {0.f, 0.f, 0.f}, vi[3] = {0.f, 0.f, 0.f}, ai[3] = {0.f, 0.f, 0.f};
    if (REB && pinc >= 0) {
#pragma unroll
        for (int m = 0; m < 3; ++m) xi[m] = pos[3 * pinc + m], vi[m] = vel[3 * pinc + m], ai[m] = acc[3 * pinc + m];
    }
    const unsigned seg = (blockIdx.x % PM_NSUB) * subcap;
    auto append = [&](unsigned k, unsigned p) {  // past the staging area: straight to the global list
        const unsigned g = atomicAdd(sub_counter(counts), 1u);
        if (g < subcap) lkeys[seg + g] = k, lidx[seg + g] = p;
    };
    // a range well above one particle per cell: lanes of a wave share cells (deposit_lds<..., ROTATE>)
    const bool dense = (e - s) + (int64_t)ninc >= PM_DENSE_RANGE;
    if constexpr (REB) {
        // kick, drift, wrap, deposit and the write into the NEW stream of one particle (old index p, new index d64)
        auto place = [&](float x[3], float v[3], const float a[3], int64_t d64, int64_t p) {
#pragma unroll
            for (int m = 0; m < 3; ++m) pm_kick_drift_wrap_one(x[m], v[m], a[m], mhalf_dt, dt);
            const pm_v3f vx = {x[0], x[1], x[2]}, vv = {v[0], v[1], v[2]};
            __builtin_nontemporal_store(vv, reinterpret_cast<pm_v3fu *>(kr.vel_out + 3 * d64));
            __builtin_nontemporal_store(vx, reinterpret_cast<pm_v3fu *>(kr.pos_out + 3 * d64));
            if (kr.id) kr.id_out[d64] = kr.id[p];
            const float xp = x[0] * nf, yp = x[1] * nf, zp = x[2] * nf;
            const int i = cell_of(xp, n), j = cell_of(yp, n), k = cell_of(zp, n);
            if (in_tile(i, j, k, t)) {
                if (dense)
                    deposit_lds<SCHEME, true>(xp, yp, zp, i, j, k, t, tile);
                else
                    deposit_lds<SCHEME>(xp, yp, zp, i, j, k, t, tile);
            } else {  // the guess of this step was off for it: served by the tile it is in, from that tile's list
                const unsigned k2 = tile_key(i >> PM_LI, j >> PM_LJ, k >> PM_LK);
                const unsigned slot = atomicAdd(&s_cnt, 1u);
                if (slot < PM_SCAP)
                    s_k[slot] = k2, s_p[slot] = (unsigned)d64;
                else
                    append(k2, (unsigned)d64);
            }
        };
        const unsigned lane = (unsigned)tid & 63u;
        const int64_t nbase = kr.newstart[key] + (int64_t)s_rbase;
        // the stayers of the range, compacted in order: one LDS atomic per wave and iteration hands out their slots
#pragma unroll 1
        for (int64_t p0 = s + tid; p0 - lane < e; p0 += 2 * BLK) {
            float x[2][3], v[2][3], a[2][3];
            unsigned fl[2] = {0u, 0u};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int64_t p = p0 + u * BLK;
                if (p < e) {
                    fl[u] = kr.flag[p];
#pragma unroll
                    for (int m = 0; m < 3; ++m)
                        x[u][m] = __builtin_nontemporal_load(pos + 3 * p + m), v[u][m] = __builtin_nontemporal_load(vel + 3 * p + m),
                        a[u][m] = __builtin_nontemporal_load(acc + 3 * p + m);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int64_t p = p0 + u * BLK;
                const bool act = p < e && fl[u] == 13u;
                const unsigned long long m = __ballot(act);
                if (m == 0ull) continue;
                const unsigned before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                unsigned wb = 0;
                if (act && before == 0) wb = atomicAdd(&s_rslot, (unsigned)__popcll(m));
                wb = __shfl((int)wb, __ffsll((long long)m) - 1, 64);
                if (act) place(x[u], v[u], a[u], nbase + (int64_t)(wb + before), p);
            }
        }
        // the tile's incoming list (first workgroup of the tile): behind all its stayers
        if (ninc) {
            const int64_t ibase = kr.newstart[key] + (int64_t)kr.tile_stay[key];
            const unsigned *const ent = kr.entries + kr.segstart[key];
            if (pinc >= 0) place(xi, vi, ai, ibase + tid, pinc);
#pragma unroll 1
            for (unsigned jn = tid + BLK; jn < ninc; jn += BLK) {  // (a long list: the rest where it is needed)
                const int64_t p = ent[jn];
                float x[3], v[3], a[3];
#pragma unroll
                for (int m = 0; m < 3; ++m) x[m] = pos[3 * p + m], v[m] = vel[3 * p + m], a[m] = acc[3 * p + m];
                place(x, v, a, ibase + jn, p);
            }
        }
    } else {
    // two particles per iteration: 18 dword loads in flight per lane
#pragma unroll 1
    for (int64_t p0 = s + tid; p0 < e; p0 += 2 * BLK) {
        float x[2][3], v[2][3], a[2][3];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t p = p0 + u * BLK;
            if (p < e) {
#pragma unroll
                for (int m = 0; m < 3; ++m)
                    x[u][m] = __builtin_nontemporal_load(pos + 3 * p + m), v[u][m] = __builtin_nontemporal_load(vel + 3 * p + m),
                    a[u][m] = __builtin_nontemporal_load(acc + 3 * p + m);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t p = p0 + u * BLK;
            if (p < e) {
#pragma unroll
                for (int m = 0; m < 3; ++m) pm_kick_drift_wrap_one(x[u][m], v[u][m], a[u][m], mhalf_dt, dt);
#pragma unroll
                for (int m = 0; m < 3; ++m) __builtin_nontemporal_store(v[u][m], vel + 3 * p + m);
#pragma unroll
                for (int m = 0; m < 3; ++m) __builtin_nontemporal_store(x[u][m], pos + 3 * p + m);
                if (lv.idx && (x[u][0] < lv.x_lo || x[u][0] >= lv.x_hi)) {  // slab: leaves this rank's planes
                    const int slot = atomicAdd(lv.count, 1);
                    if (slot < lv.cap) lv.idx[slot] = p;
                }
                const float xp = x[u][0] * nf, yp = x[u][1] * nf, zp = x[u][2] * nf;
                const int i = cell_of(xp, n), j = cell_of(yp, n), k = cell_of(zp, n);
                float veff[3];
#pragma unroll
                for (int m = 0; m < 3; ++m) veff[m] = v[u][m] + dc.kick * a[u][m];
                if (in_tile(i, j, k, t)) {
                    if (dense)
                        deposit_lds<SCHEME, true>(xp, yp, zp, i, j, k, t, tile);
                    else
                        deposit_lds<SCHEME>(xp, yp, zp, i, j, k, t, tile);
                    if (dc.dcode) {  // (codes without counts: a slab, whose migration comes before the counting)
                        const DestCode dd = dest_of(x[u], veff, dc.dtp, nf, n, t);
                        dc.dcode[p] = (unsigned char)dd.d;
                        if (dc.tile_count && dd.d != 13) atomicAdd(&s_hist[dd.d], 1u);
                    }
                } else {
                    const unsigned k2 = tile_key(i >> PM_LI, j >> PM_LJ, k >> PM_LK);
                    const unsigned slot = atomicAdd(&s_cnt, 1u);
                    if (slot < PM_SCAP)
                        s_k[slot] = k2, s_p[slot] = (unsigned)p;
                    else
                        append(k2, (unsigned)p);
                    // served by the tile it is in now, from that tile's list: counted at its destination one by one
                    if (dc.dcode) {
                        const DestCode dd = dest_of(x[u], veff, dc.dtp, nf, n, tile_of_cell(i, j, k));
                        dc.dcode[p] = (unsigned char)dd.d;
                        if (dc.tile_count) atomicAdd(dc.tile_count + dd.key, 1u);
                    }
                }
            }
        }
    }
    }  // (!REB)
    __syncthreads();
    // Flush.  The global float atomics execute at the memory side as 64-byte read-modify-writes, one per
    // touched 64-byte sector whatever it carries (at 1024^3 the flush was 5.0 of the kernel's 16.8 ms: 2.5 the
    // single k-halo cells at the ends of every row, 2.3 the rows of own cells, 0.2 the i/j-halo rows; without
    // it the kernel runs at the speed of the plain streaming update).  With a halo buffer the owner of a tile
    // uses no atomics at all: own cells are stored, halo cells go to the buffer, and the z pass of the FFT
    // adds the buffer to the density lines it loads (pm_fftpoisson.hip).
    for (int c = tid; c < PM_TILE; c += BLK) {
        const pm_acc_t w = tile[c];
        if (owner)
            halo_store(c, weight_of_acc(w));
        else if (w != 0)
            unsafeAtomicAdd(rho + tile_cell_global(c, t, geo), weight_of_acc(w));
    }
    // where the particles this workgroup serves (those of its range that are inside its tile) are filed next:
    // per workgroup for the slots the re-bucketing gather takes, per tile for the layout of the new stream
    if (dc.tile_count && tid < 27) {
        unsigned cnt_d = s_hist[tid];
        if (tid == 13) {
            unsigned moved = 0;
            for (int d = 0; d < 27; ++d) moved += s_hist[d];
            cnt_d = (unsigned)(e - s) - s_cnt - moved;
        }
        dc.wg_hist[(size_t)(blockIdx.x + work.boff) * 27 + tid] = cnt_d;
        if (cnt_d) {
            const int Ti = n >> PM_LI, Tj = n >> PM_LJ, Tk = n >> PM_LK;
            const int di = tid / 9 - 1, dj = (tid / 3) % 3 - 1, dk = tid % 3 - 1;
            atomicAdd(dc.tile_count + tile_key((t.ti + di + Ti) % Ti, (t.tj + dj + Tj) % Tj, (t.tk + dk + Tk) % Tk), cnt_d);
        }
    }
    const unsigned cnt = min(s_cnt, (unsigned)PM_SCAP);
    if (cnt == 0) return;
    if (tid == 0) s_base = atomicAdd(sub_counter(counts), cnt);
    __syncthreads();
    const unsigned base = s_base;
    for (unsigned i = tid; i < cnt; i += BLK)
        if (base + i < subcap) lkeys[seg + base + i] = s_k[i], lidx[seg + base + i] = s_p[i];
}

// deposit of the listed stragglers, one workgroup per tile with a non-empty incoming list.  With a halo buffer
// the workgroup is the only writer of its tile's own cells and of its tile's buffer entries while this kernel
// runs (the owners' stores and the surplus workgroups' atomics belong to earlier launches), so it adds with plain
// loads and stores of the touched cells, rows of 16 contiguous floats; without one, with global atomics --
// which the memory side executes as a 64-byte read-modify-write per touched sector (the 5 ms the halo buffer took
// out of the main kernel came back here once most tiles had a list: 5.2 ms at 15 % stragglers, 1024^3).
template <int SCHEME>
__global__ __launch_bounds__(BLK) void deposit_list_kernel(const float *__restrict__ pos, TileLists lists, pm_geom geo,
                                                           float *__restrict__ rho, HaloBuf halo) {
    const unsigned key = blockIdx.x;
    const unsigned ls = lists.lstart[key], le = lists.lstart[key + 1];
    if (ls == le) return;
    __shared__ pm_acc_t tile[PM_TILE];
    const int tid = threadIdx.x;
    const int n = geo.n;
    const float nf = (float)n;
    const TileId t = tile_of_key(key);
    for (int c = tid; c < PM_TILE; c += BLK) tile[c] = 0;
    __syncthreads();
    for (unsigned l = ls + tid; l < le; l += BLK) {
        const int64_t p = lists.idx[l];
        const float x = pos[3 * p], y = pos[3 * p + 1], z = pos[3 * p + 2];
        const float xp = x * nf, yp = y * nf, zp = z * nf;
        const int i = cell_of(xp, n), j = cell_of(yp, n), k = cell_of(zp, n);
        if (in_tile(i, j, k, t))
            deposit_lds<SCHEME>(xp, yp, zp, i, j, k, t, tile);
        else
            deposit_global<SCHEME>(x, y, z, geo, rho);
    }
    __syncthreads();
    if (halo.k && (halo.windowed || le - ls < 128)) {
        // a short list (the usual case once the stream is re-bucketed every step: ~10 particles per tile) touches a
        // few hundred cells: fire-and-forget atomics on the tile's own addresses, no load to wait for (the
        // read-modify-write form below cost 2.6 ms at 0.5 % stragglers, the atomics 0.4 ms)
        for (int c = tid; c < PM_TILE; c += BLK) {
            const pm_acc_t w = tile[c];
            if (w != 0) unsafeAtomicAdd(halo_dest(c, t, halo, geo, rho), weight_of_acc(w));
        }
    } else if (halo.k) {  // (single GPU only: on a slab a leaver's tile adds to own cells of the window's edge tiles)
        // read-modify-write of the touched cells, all the loads of a thread in flight before its first store (one
        // dependent load -> store per cell made the kernel latency-bound: 5.4 ms at 4 % stragglers, 1024^3)
        constexpr int NQ = (PM_TILE + BLK - 1) / BLK;
        float *dst[NQ];
        float add[NQ], old[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = q * BLK + tid;
            const pm_acc_t w = c < PM_TILE ? tile[c] : 0;
            add[q] = weight_of_acc(w);
            dst[q] = w != 0 ? halo_cell(c, t, halo, geo, rho) : nullptr;
        }
        // (unconditional loads, untouched cells from one cached dummy address: a guarded load is sunk to its use)
#pragma unroll
        for (int q = 0; q < NQ; ++q) old[q] = *(dst[q] ? dst[q] : rho);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (dst[q]) *dst[q] = old[q] + add[q];
    } else {
        for (int c = tid; c < PM_TILE; c += BLK) {
            const pm_acc_t w = tile[c];
            if (w != 0) unsafeAtomicAdd(rho + tile_cell_global(c, t, geo), weight_of_acc(w));
        }
    }
}

// f(di, dj, dk, w) for every cell of the particle's assignment footprint (offsets from its cell, fp32 weight in the
// reference's operation order) -- the weights of deposit_lds / deposit_global, for the flat list deposit below
template <int SCHEME, class F>
__device__ __forceinline__ void deposit_each(float xp, float yp, float zp, int i, int j, int k, F f) {
    if (SCHEME == PM_NGP) {
        f(0, 0, 0, 1.0f);
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
                for (int c = 0; c < 2; ++c) f(a * sx, b * sy, c * sz, wx[a] * wy[b] * wz[c]);
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
                for (int c = 0; c < 3; ++c) f(a - 1, b - 1, c - 1, wxy * wz[c]);
            }
    }
}

// The same deposit for SHORT lists (a stream that is re-bucketed every step leaves about one listed particle per
// tile): one WAVE per listed particle, a lane per cell of its footprint, added straight to the addresses the tile it
// is in owns (own cells in rho, halo cells in the tile's halo-buffer entries) with fire-and-forget atomics: 9 rows
// of 3 adjacent cells per wave instruction.  (One particle per lane -- 64 scattered sectors per instruction --
// was 4x slower than the per-tile kernel above, which spends 0.8 ms at 1024^3 on clearing and scanning half a
// million LDS tiles whatever the lists hold.)
template <int SCHEME>
__global__ __launch_bounds__(BLK) void deposit_list_flat_kernel(const float *__restrict__ pos,
                                                                const unsigned *__restrict__ idx, unsigned count,
                                                                pm_geom geo, float *__restrict__ rho, HaloBuf halo) {
    const int n = geo.n;
    const float nf = (float)n;
    const unsigned lane = threadIdx.x & 63u, nwaves = gridDim.x * (BLK / 64);
    for (unsigned l = blockIdx.x * (BLK / 64) + (threadIdx.x >> 6); l < count; l += nwaves) {
        const int64_t p = idx[l];
        const float x = pos[3 * p], y = pos[3 * p + 1], z = pos[3 * p + 2];
        const float xp = x * nf, yp = y * nf, zp = z * nf;
        const int i = cell_of(xp, n), j = cell_of(yp, n), k = cell_of(zp, n);
        const TileId t = tile_of_cell(i, j, k);
        const int c0 = ((i - t.ti * PM_TI + 1) * PM_HJ + (j - t.tj * PM_TJ + 1)) * PM_HK + (k - t.tk * PM_TK + 1);
        // the lane's cell of the footprint (offsets from the particle's cell) and its weight, products in the
        // reference's order; ONE atomic instruction per particle
        int di = 0, dj = 0, dk = 0;
        float w = 1.0f;
        bool on = lane == 0;
        auto pick = [](const float *v, int q) { return q == 0 ? v[0] : (q == 1 ? v[1] : v[2]); };
        if (SCHEME == PM_CIC) {
            float wx[3], wy[3], wz[3];
            int sx, sy, sz;
            cic_weights(xp, i, wx[0], wx[1], sx);
            cic_weights(yp, j, wy[0], wy[1], sy);
            cic_weights(zp, k, wz[0], wz[1], sz);
            wx[2] = wy[2] = wz[2] = 0.f;
            const int a = (lane >> 2) & 1, b = (lane >> 1) & 1, c = lane & 1;
            on = lane < 8, di = a * sx, dj = b * sy, dk = c * sz;
            w = pick(wx, a) * pick(wy, b) * pick(wz, c);
        } else if (SCHEME == PM_TSC) {
            float wx[3], wy[3], wz[3];
            tsc_weights(xp, i, wx);
            tsc_weights(yp, j, wy);
            tsc_weights(zp, k, wz);
            const int a = (int)lane / 9, b = ((int)lane / 3) % 3, c = (int)lane % 3;
            on = lane < 27, di = a - 1, dj = b - 1, dk = c - 1;
            w = (pick(wx, a) * pick(wy, b)) * pick(wz, c);
        }
        if (on) {
            if (halo.k)
                unsafeAtomicAdd(halo_dest(c0 + (di * PM_HJ + dj) * PM_HK + dk, t, halo, geo, rho), w);
            else
                unsafeAtomicAdd(rho + pm_cell(geo, i + di, j + dj, k + dk), w);
        }
    }
}

template <int SCHEME>
static int kdw_deposit_launch(pm_ctx *c, float *pos, float *vel, const float *acc, float half_dt, float dt, int64_t np,
                              float *rho, Leavers lv, float *khalo) {
    TileWork work;
    unsigned blocks = 0;
    tile_lists_drop(c);
    // destination counts for the incremental re-sort: single-GPU layout only (a slab's migration renumbers the
    // particles).  Their buffer is sized before the ranges are looked up: it holds the offsets table of a stream the
    // gather wrote, and growing moves it.
    RbBufs rb;
    tile_window(c, &work);
    const bool count_stay = !lv.idx && work.nlaunch == work.ntiles;
    int rc = count_stay ? rebucket_bufs(c, work.ntiles, np, &rb) : 0;
    if (rc) return rc;
    rc = tile_ranges(c, pos, np, &work, &blocks);  // ranges of the positions before the drift
    if (rc) return rc;
    const unsigned ntiles = work.ntiles;
    ListBufs b;
    rc = list_bufs(c, np, ntiles, &b);
    if (rc) return rc;
    DestCount dcount{nullptr, nullptr, nullptr, c->dt_pred, -0.5f * (dt + c->dt_pred) * c->acc_scale_pred};
    if (count_stay) {
        PM_HIP(hipMemsetAsync(rb.tile_count, 0, ((size_t)ntiles + 1) * sizeof(unsigned), c->stream));
        dcount.tile_count = rb.tile_count, dcount.wg_hist = rb.wg_hist, dcount.dcode = rb.dcode;
    } else if (c->codes_user) {
        dcount.dcode = c->codes_user;  // pm_set_destination_codes: the codes only, counted later (pm_rebucket_prepare)
    }
    const size_t cells = (size_t)c->slab_nx * c->n * c->n;
    const bool windowed = work.nlaunch != ntiles;
    const HaloBuf halo = windowed ? halo_parts(khalo, c->n, work.ti0, 1 << work.lnti) : halo_parts(khalo, c->n);
    const size_t halo_bytes = (size_t)halo_floats(c->n, halo.nti) * sizeof(float);
    if (!khalo) {
        PM_HIP(hipMemsetAsync(rho, 0, cells * sizeof(float), c->stream));  // owners store every cell otherwise
    } else if (windowed) {
        // a slab: the owners store every cell of the window's planes; the ghost planes either side only receive
        // additions (the edge tiles' outward halo cells, leavers' clouds)
        const int ghosts = (c->slab_nx - c->tile_nx) / 2;
        PM_REQUIRE(ghosts >= 1 && c->slab_nx == c->tile_nx + 2 * ghosts &&
                       c->slab_x0 == ((c->tile_x_lo - ghosts) % c->n + c->n) % c->n,
                   "halo buffer on a slab: the density slab must be the tile window + the same ghost planes either side");
        const size_t plane = (size_t)c->n * c->n * sizeof(float);
        PM_HIP(hipMemsetAsync(rho, 0, ghosts * plane, c->stream));
        PM_HIP(hipMemsetAsync(rho + (size_t)(ghosts + c->tile_nx) * c->n * c->n, 0, ghosts * plane, c->stream));
    }
    PM_HIP(hipMemsetAsync(b.counts, 0, PM_NSUB * PM_SUB_STRIDE * sizeof(unsigned), c->stream));
    if (!khalo) {
        hipLaunchKernelGGL((kdw_deposit_kernel<SCHEME>), dim3(blocks), dim3(BLK), 0, c->stream, pos, vel, acc, -half_dt, dt,
                           work, pm_ctx_geom(c), rho, b.k_in, b.i_in, (unsigned)b.subcap, b.counts, lv, halo, dcount);
    } else {
        // owners first (plain stores), then the surplus workgroups of over-full tiles, which add with atomics
        hipLaunchKernelGGL((kdw_deposit_kernel<SCHEME>), dim3(work.nlaunch), dim3(BLK), 0, c->stream, pos, vel, acc, -half_dt,
                           dt, work, pm_ctx_geom(c), rho, b.k_in, b.i_in, (unsigned)b.subcap, b.counts, lv, halo, dcount);
        if (blocks > work.nlaunch) {
            TileWork extra = work;
            extra.boff = work.nlaunch;
            hipLaunchKernelGGL((kdw_deposit_kernel<SCHEME>), dim3(blocks - work.nlaunch), dim3(BLK), 0, c->stream, pos, vel,
                               acc, -half_dt, dt, extra, pm_ctx_geom(c), rho, b.k_in, b.i_in, (unsigned)b.subcap, b.counts, lv,
                               halo, dcount);
        }
    }
    PM_LAUNCH_CHECK();
    const unsigned *idx = nullptr;
    unsigned listed = 0;
    int complete = 0;
    rc = list_group(c, b, ntiles, &idx, &listed, &complete);
    if (rc) return rc;
    c->lists_ptr[0] = work.start, c->lists_ptr[1] = work.xstart, c->lists_ptr[2] = idx, c->lists_ptr[3] = b.lstart;
    c->lists_blocks = blocks;
    if (!complete) {
        // a sub-list overflowed (stragglers bunched up in few workgroups): no complete lists.  The positions are
        // updated; deposit them again from scratch with the kernels that serve stragglers themselves.
        c->lists_pos = nullptr;
        PM_HIP(hipMemsetAsync(rho, 0, cells * sizeof(float), c->stream));
        if (khalo) PM_HIP(hipMemsetAsync(khalo, 0, halo_bytes, c->stream));
        hipLaunchKernelGGL((deposit_tiled_kernel<SCHEME>), dim3((unsigned)((np + PM_CHUNK - 1) / PM_CHUNK)), dim3(BLK), 0,
                           c->stream, pos, np, pm_ctx_geom(c), rho);
        PM_LAUNCH_CHECK();
        return PM_OK;
    }
    TileLists lists;
    lists.idx = idx, lists.lstart = b.lstart, lists.use = 1;
    if (listed > 0 && (int64_t)listed * 400 < np) {  // under 0.25 % (measured break-even at 1024^3: 0.3 %)
        hipLaunchKernelGGL((deposit_list_flat_kernel<SCHEME>), dim3(pm_stream_blocks((int64_t)listed * 64, BLK)), dim3(BLK), 0,
                           c->stream, pos, idx, listed, pm_ctx_geom(c), rho, halo);
        PM_LAUNCH_CHECK();
    } else if (listed > 0) {
        hipLaunchKernelGGL((deposit_list_kernel<SCHEME>), dim3(ntiles), dim3(BLK), 0, c->stream, pos, lists, pm_ctx_geom(c),
                           rho, halo);
        PM_LAUNCH_CHECK();
    }
    c->lists_pos = pos, c->lists_np = np, c->lists_use = 1, c->lists_sticky = 1, c->stay_valid = count_stay ? 1 : 0;
    c->stay_dt_pred = c->dt_pred, c->stay_codes = count_stay ? rb.dcode : nullptr;
    return PM_OK;
}

static int kdw_deposit_scheme(pm_ctx *c, int scheme, float *d_pos, float *d_vel, const float *d_acc, float half_dt,
                              float dt, int64_t np, float *d_rho, Leavers lv, float *khalo = nullptr) {
    switch (scheme) {
        case PM_NGP: return kdw_deposit_launch<PM_NGP>(c, d_pos, d_vel, d_acc, half_dt, dt, np, d_rho, lv, khalo);
        case PM_CIC: return kdw_deposit_launch<PM_CIC>(c, d_pos, d_vel, d_acc, half_dt, dt, np, d_rho, lv, khalo);
        case PM_TSC: return kdw_deposit_launch<PM_TSC>(c, d_pos, d_vel, d_acc, half_dt, dt, np, d_rho, lv, khalo);
    }
    pm_set_error("mass scheme %d, should be NGP(1), CIC(2) or TSC(3)", scheme);
    return PM_EINVAL;
}

extern "C" int pm_kick_drift_wrap_deposit(pm_ctx *c, int scheme, float *d_pos, float *d_vel, const float *d_acc,
                                          float half_dt, float dt, int64_t np, float *d_rho) {
    PM_REQUIRE(c && d_pos && d_vel && d_acc && d_rho && np > 0, "bad argument");
    PM_REQUIRE(c->n >= 16 && np < 4294967296LL, "pm_kick_drift_wrap_deposit needs n >= 16 and np < 2^32");
    PM_REQUIRE(c->slab_nx == c->n, "pm_kick_drift_wrap_deposit is the single-GPU form (no slab set)");
    return kdw_deposit_scheme(c, scheme, d_pos, d_vel, d_acc, half_dt, dt, np, d_rho, Leavers{0.f, 1.f, nullptr, 0, nullptr});
}

// The same with the k-halo cells of the tiles kept out of d_rho (pmstep.h): d_khalo[(n/8)(n/16)(n/16)][2][180].
extern "C" int pm_kick_drift_wrap_deposit_khalo(pm_ctx *c, int scheme, float *d_pos, float *d_vel, const float *d_acc,
                                                float half_dt, float dt, int64_t np, float *d_rho, float *d_khalo) {
    PM_REQUIRE(c && d_pos && d_vel && d_acc && d_rho && d_khalo && np > 0, "bad argument");
    PM_REQUIRE(c->n >= 16 && np < 4294967296LL, "pm_kick_drift_wrap_deposit_khalo needs n >= 16 and np < 2^32");
    PM_REQUIRE(c->slab_nx == c->n && c->tile_nx == c->n, "pm_kick_drift_wrap_deposit_khalo is the single-GPU form");
    return kdw_deposit_scheme(c, scheme, d_pos, d_vel, d_acc, half_dt, dt, np, d_rho, Leavers{0.f, 1.f, nullptr, 0, nullptr},
                              d_khalo);
}

// d_rho += the halo buffer: completes a density deposited by pm_kick_drift_wrap_deposit_khalo for consumers
// other than pm_poisson_fft_fused_khalo (one global atomic per non-zero entry, as the flush would have done)
__global__ __launch_bounds__(BLK) void khalo_apply_kernel(HaloBuf halo, pm_geom geo, float *__restrict__ rho) {
    const int n = geo.n, T = n >> PM_LJ;
    const int64_t nk = halo_k_floats(n, n >> PM_LI), ni = halo_i_floats(n, n >> PM_LI), nj = halo_j_floats(n, n >> PM_LI);
    for (int64_t e = (int64_t)blockIdx.x * BLK + threadIdx.x; e < nk + ni + nj; e += (int64_t)gridDim.x * BLK) {
        float v;
        int gi, gj, gk;
        if (e < nk) {
            v = halo.k[e];  // (((col * 10 + li) * 2 + side) * T + tk) * 18 + lj
            const int lj = (int)(e % PM_HJ);
            int64_t r = e / PM_HJ;
            const int tk = (int)(r % T);
            r /= T;
            const int side = (int)(r & 1), li = (int)((r >> 1) % PM_HI);
            const int64_t col = (r >> 1) / PM_HI;
            gi = (int)(col / T) * PM_TI - 1 + li, gj = (int)(col % T) * PM_TJ - 1 + lj, gk = tk * PM_TK - 1 + (side ? PM_HK - 1 : 0);
        } else if (e < nk + ni) {
            const int64_t f = e - nk;
            v = halo.hi[f];
            gk = (int)(f % n);
            const int64_t row = f / n;  // ((col * 2 + side) * 18 + lj)
            const int lj = (int)(row % PM_HJ), side = (int)((row / PM_HJ) & 1);
            const int64_t col = row / (2 * PM_HJ);
            gi = (int)(col / T) * PM_TI - 1 + (side ? PM_HI - 1 : 0), gj = (int)(col % T) * PM_TJ - 1 + lj;
        } else {
            const int64_t f = e - nk - ni;
            v = halo.hj[f];
            gk = (int)(f % n);
            const int64_t row = f / n;  // ((col * 2 + side) * 8 + li - 1)
            const int li = (int)(row % PM_TI) + 1, side = (int)((row / PM_TI) & 1);
            const int64_t col = row / (2 * PM_TI);
            gi = (int)(col / T) * PM_TI - 1 + li, gj = (int)(col % T) * PM_TJ - 1 + (side ? PM_HJ - 1 : 0);
        }
        if (v != 0.f) unsafeAtomicAdd(rho + pm_cell(geo, gi, gj, gk), v);
    }
}

extern "C" int pm_khalo_apply(pm_ctx *c, float *d_rho, const float *d_khalo) {
    PM_REQUIRE(c && d_rho && d_khalo, "NULL argument");
    PM_REQUIRE(c->n >= 16 && c->slab_nx == c->n, "pm_khalo_apply is the single-GPU form");
    const int64_t total = halo_floats(c->n, c->n >> PM_LI);
    hipLaunchKernelGGL(khalo_apply_kernel, dim3(pm_stream_blocks(total, BLK)), dim3(BLK), 0, c->stream,
                       halo_parts(const_cast<float *>(d_khalo), c->n), pm_ctx_geom(c), d_rho);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

extern "C" int64_t pm_khalo_floats(int n) {
    return n >= 16 ? halo_floats(n, n >> PM_LI) : 0;
}

// The same on a slab (pm_set_slab: d_rho is [nx_slab, n, n] starting at global plane x0, wide enough for
// the footprint of every particle AFTER the drift -- the caller bounds the drift and chooses the ghost
// width) plus the list of the particles whose new x left [x_lo, x_hi), as pm_kick_drift_wrap_slab returns it.
// The deposit of the leavers is done here, by their old owner; the prepared tile lists are dropped (the
// migration that follows renumbers the particles).
static int kdw_deposit_slab(pm_ctx *c, int scheme, float *d_pos, float *d_vel, const float *d_acc, float half_dt, float dt,
                            int64_t np, float *d_rho, float x_lo, float x_hi, int64_t *d_leave_idx, int cap, int *h_count,
                            float *d_khalo) {
    PM_REQUIRE(c && d_pos && d_vel && d_acc && d_rho && np > 0 && d_leave_idx && h_count && cap > 0, "bad argument");
    PM_REQUIRE(c->n >= 16 && np < 4294967296LL, "pm_kick_drift_wrap_deposit_slab needs n >= 16 and np < 2^32");
    int *d_count = reinterpret_cast<int *>(c->d_scratch + 16);
    PM_HIP(hipMemsetAsync(d_count, 0, sizeof(int), c->stream));
    int rc = kdw_deposit_scheme(c, scheme, d_pos, d_vel, d_acc, half_dt, dt, np, d_rho,
                                Leavers{x_lo, x_hi, d_leave_idx, cap, d_count}, d_khalo);
    tile_lists_drop(c);
    if (rc) return rc;
    if (int rb_rc = pm_read_back(c, h_count, d_count, sizeof(int))) return rb_rc;
    return PM_OK;
}

extern "C" int pm_kick_drift_wrap_deposit_slab(pm_ctx *c, int scheme, float *d_pos, float *d_vel, const float *d_acc,
                                               float half_dt, float dt, int64_t np, float *d_rho, float x_lo,
                                               float x_hi, int64_t *d_leave_idx, int cap, int *h_count) {
    return kdw_deposit_slab(c, scheme, d_pos, d_vel, d_acc, half_dt, dt, np, d_rho, x_lo, x_hi, d_leave_idx, cap, h_count,
                            nullptr);
}

// ... with the halo buffer (pm_kick_drift_wrap_deposit_khalo on a slab): needs a tile window (pm_set_tile_window) and
// the density slab = the window's planes + the same number (>= 1) of ghost planes either side.  The buffer holds the
// window's rows of tiles only (pm_khalo_floats_slab(n, nx_window) floats); the owners STORE the window's planes
// (d_rho is not cleared there), the ghost planes are cleared and receive the additions that leave the window through
// its x faces: the edge tiles' outward halo cells and the clouds of particles in tiles outside it (leavers).  The
// buffer is consumed by pm_fftp_z_forward_khalo once the caller has folded the ghost planes into the neighbours.
extern "C" int pm_kick_drift_wrap_deposit_slab_khalo(pm_ctx *c, int scheme, float *d_pos, float *d_vel, const float *d_acc,
                                                     float half_dt, float dt, int64_t np, float *d_rho, float x_lo,
                                                     float x_hi, int64_t *d_leave_idx, int cap, int *h_count,
                                                     float *d_khalo) {
    PM_REQUIRE(d_khalo, "NULL argument");
    PM_REQUIRE(c && c->tile_nx < c->n, "pm_kick_drift_wrap_deposit_slab_khalo needs a tile window (pm_set_tile_window)");
    TileWork w;
    tile_window(c, &w);
    PM_REQUIRE(w.nlaunch != w.ntiles, "the tile window must be a power-of-two number of tile rows, aligned to a row");
    return kdw_deposit_slab(c, scheme, d_pos, d_vel, d_acc, half_dt, dt, np, d_rho, x_lo, x_hi, d_leave_idx, cap, h_count,
                            d_khalo);
}

extern "C" int64_t pm_khalo_floats_slab(int n, int nx_window) {
    return (n >= 16 && nx_window >= PM_TI && nx_window % PM_TI == 0) ? halo_floats(n, nx_window >> PM_LI) : 0;
}

// ------------------------------------------------------------------------------------------------
// Gradient + gather + kick in one kernel (mesh.derivative + mesh.inv*_vec + the second kick): the
// workgroup stages the POTENTIAL of its tile (force tile + stencil reach R on every side) in LDS,
// differentiates it there with the reference's stencil (Stencil<ORDER,0>::eval, same bits as
// pm_gradient), parks the three components in registers, and rewrites the same LDS as the
// force tile the gather reads.  The [n,n,n,3] force grid (12.9 GB at 1024^3: one write, 1.6 reads
// with halos) and the gradient pass never exist.

template <int ORDER>
struct GradReach {
    static constexpr int R = (ORDER == 2 || ORDER == 3) ? 1 : (ORDER == 5 ? 2 : 3);
};

template <int ORDER>
__device__ __forceinline__ float grad_scale(int n) {
    return ORDER == 2 ? (float)n : (ORDER == 3 ? (float)(0.5 * n) : (ORDER == 5 ? (float)(n / 12.0) : (float)(n / 60.0)));
}

// force of one cell straight from the global potential (stragglers only)
template <int ORDER>
struct PhiFetch {
    const float *phi;
    pm_geom geo;
    float scale;
    __device__ __forceinline__ void operator()(int ii, int jj, int kk, float g[3]) const {
        constexpr int R = GradReach<ORDER>::R;
        float sa[7], sb[7];
#pragma unroll
        for (int o = 0; o < 7; ++o) sa[o] = sb[o] = 0.f;
#pragma unroll
        for (int o = -R; o <= R; ++o) sa[o + 3] = phi[pm_cell(geo, ii + o, jj, kk)];
        g[0] = Stencil<ORDER, 0>::eval(sa, sb, 0.f, scale);
#pragma unroll
        for (int o = -R; o <= R; ++o) sa[o + 3] = phi[pm_cell(geo, ii, jj + o, kk)];
        g[1] = Stencil<ORDER, 0>::eval(sa, sb, 0.f, scale);
#pragma unroll
        for (int o = -R; o <= R; ++o) sa[o + 3] = phi[pm_cell(geo, ii, jj, kk + o)];
        g[2] = Stencil<ORDER, 0>::eval(sa, sb, 0.f, scale);
    }
};

template <int SCHEME, int ORDER>
__device__ __noinline__ Acc3 gather_phi_global(const float *__restrict__ phi, float scale, float xp, float yp, float zp,
                                               int i, int j, int k, pm_geom geo) {
    PhiFetch<ORDER> fetch{phi, geo, scale};
    float acc[3];
    acc[0] = acc[1] = acc[2] = 0.f;
    float g[3];
    if (SCHEME == PM_NGP) {
        fetch(i, j, k, g);
        acc[0] = g[0], acc[1] = g[1], acc[2] = g[2];
    } else if (SCHEME == PM_CIC) {
        float wx[2], wy[2], wz[2];
        int sx, sy, sz;
        cic_weights(xp, i, wx[0], wx[1], sx);
        cic_weights(yp, j, wy[0], wy[1], sy);
        cic_weights(zp, k, wz[0], wz[1], sz);
#pragma unroll 1
        for (int q = 0; q < 8; ++q) {
            const int a = q >> 2, b = (q >> 1) & 1, c = q & 1;
            const float w = (a ? wx[1] : wx[0]) * (b ? wy[1] : wy[0]) * (c ? wz[1] : wz[0]);
            fetch(i + a * sx, j + b * sy, k + c * sz, g);
#pragma unroll
            for (int m = 0; m < 3; ++m) acc[m] = acc[m] + w * g[m];
        }
    } else {
        float wx[3], wy[3], wz[3];
        tsc_weights(xp, i, wx);
        tsc_weights(yp, j, wy);
        tsc_weights(zp, k, wz);
#pragma unroll 1
        for (int q = 0; q < 27; ++q) {
            const int a = q / 9, b = (q / 3) % 3, c = q % 3;
            const float wxa = a == 0 ? wx[0] : (a == 1 ? wx[1] : wx[2]);
            const float wyb = b == 0 ? wy[0] : (b == 1 ? wy[1] : wy[2]);
            const float wzc = c == 0 ? wz[0] : (c == 1 ? wz[1] : wz[2]);
            const float w = (wxa * wyb) * wzc;
            fetch(i + a - 1, j + b - 1, k + c - 1, g);
#pragma unroll
            for (int m = 0; m < 3; ++m) acc[m] = acc[m] + w * g[m];
        }
    }
    return Acc3{acc[0], acc[1], acc[2]};
}

// Incremental re-sort riding on the gather (pm_gradient_interp_kick_rebucket): the kernel reads every particle's
// position and velocity and writes its velocity and acceleration anyway; written OUT OF PLACE into the range of
// its destination tile (dest_of: the tile it is expected to be in after the next drift) the stream comes out
// sorted for 12 more bytes per particle (the position), where a sort by key costs a dozen passes.  The order
// inside a tile is the order of arrival (nothing depends on it: the tile sums of the deposit are integer sums).
struct __attribute__((packed, aligned(4))) Packed3 {
    float a, b, c;
};

constexpr int PM_SAMPLE_WORD = 32;  // offset (in 4-byte words) of the acceleration sample's sums from the maxima

struct Rebucket {
    const int64_t *newstart;  // [ntiles + 1] offsets of the re-bucketed stream
    unsigned *fill;           // [ntiles] cleared: whoever sends particles to a tile takes a slice of its range here
    const unsigned *wg_hist;  // [workgroups][27] particles the workgroup serves from its range, per destination
    const unsigned char *dcode;  // [np] destination code of every particle (dest_of, recorded by the fused kick)
    float *pos_out, *vel_out, *acc_out;
    const unsigned *id;       // optional particle tags carried along (NULL: none)
    unsigned *id_out;
    float *sink;              // 64 x 3 floats: where the lanes of the range loop that have no particle store
};
// p for the lanes set in `mask`, q for the others, without a branch the compiler could put back
typedef __attribute__((address_space(1))) float pm_gfloat;  // (global memory: rebuilt from integers a pointer is "flat")
__device__ __forceinline__ pm_gfloat *select_ptr(unsigned long long mask, float *p, float *q) {
    const unsigned plo = (unsigned)(uintptr_t)p, phi = (unsigned)((uintptr_t)p >> 32);
    const unsigned qlo = (unsigned)(uintptr_t)q, qhi = (unsigned)((uintptr_t)q >> 32);
    unsigned lo, hi;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"(qlo), "v"(plo), "s"(mask));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"(qhi), "v"(phi), "s"(mask));
    return (pm_gfloat *)(((uintptr_t)hi << 32) | lo);
}
typedef __attribute__((address_space(1))) pm_v3fu pm_gv3fu;

// "The kick re-buckets" (round 3): the gather stays IN PLACE -- it writes the kicked velocity and the acceleration
// where it read the particle, contiguous 12-byte records, and not the position at all -- and only DECIDES where every
// particle goes: the tile it is expected to be in after the next drift, x + dt' (v + kick - dt'/2 a) with the
// acceleration it has just computed (exactly what the next fused kick will do, but for the guess dt' of the step).
// A particle that stays in the tile it is filed under gets a flag byte; one that goes elsewhere has its index
// appended to the incoming list of its destination tile.  The NEXT fused kick + drift + deposit, which rewrites
// position and velocity anyway, writes them out of place: a tile's workgroup compacts its stayers to the front of
// the tile's new range and serves its incoming list behind them (kdw_deposit_kernel<SCHEME, true>).  Against the
// re-bucketing gather this drops the 12.9 GB of re-written positions and every scattered store of the step.
struct Codes {
    unsigned char *flag;       // [np] 13: stays in the tile it is filed under; 0: it is in its destination's incoming list
    unsigned *wg_stay;         // [workgroups] stayers served from the workgroup's range
    unsigned *tile_stay;       // [ntiles] cleared: stayers per tile
    unsigned *inc_count;       // [ntiles] cleared: movers appended to the tile's incoming list
    const unsigned *segstart;  // [ntiles + 1] where a tile's incoming list lives in `entries`
    unsigned *entries;         // stream indices of the movers, by destination tile
    unsigned *overflow;        // raised when a list was full (the caller then takes an in-place step)
    float dtp;                 // the caller's guess of the next dt (pm_set_predicted_dt)
};
constexpr int PM_MOV_STAGE = 288;  // movers a workgroup stages in LDS before it books their list slots (more: one by one)

// byte pointer p for the lanes set in `mask`, q for the others (see select_ptr)
__device__ __forceinline__ __attribute__((address_space(1))) unsigned char *select_byte_ptr(unsigned long long mask,
                                                                                           unsigned char *p, unsigned char *q) {
    const unsigned plo = (unsigned)(uintptr_t)p, phi = (unsigned)((uintptr_t)p >> 32);
    const unsigned qlo = (unsigned)(uintptr_t)q, qhi = (unsigned)((uintptr_t)q >> 32);
    unsigned lo, hi;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"(qlo), "v"(plo), "s"(mask));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"(qhi), "v"(phi), "s"(mask));
    return (__attribute__((address_space(1))) unsigned char *)(((uintptr_t)hi << 32) | lo);
}

// One workgroup per tile: gather + second kick + CFL maxima.  ORDER == 0: `src` is the [.,n,n,3] force
// grid; ORDER in {2,3,5,7}: `src` is the potential and the force tile is differentiated in LDS.
template <int SCHEME, int ORDER, bool KICK, bool REBUCKET = false, bool CODES = false>
__global__ __launch_bounds__(BLK) void interp_ranges_kernel(const float *__restrict__ src,
                                                            const float *__restrict__ pos, TileWork work,
                                                            TileLists lists, pm_geom geo, float *__restrict__ out,
                                                            float *__restrict__ vel, float mhalf_dt,
                                                            unsigned *__restrict__ maxbits, Rebucket rb, Codes cd = Codes{}) {
    static_assert(!(REBUCKET && CODES) && (!CODES || KICK), "one re-sort scheme at a time; the codes need the kicked velocity");
    // The tile of a first-chunk workgroup follows from blockIdx alone, so its fill loads are issued
    // together with the range lookup instead of one memory latency after it.
    unsigned key = 0;
    int64_t s = 0, e = 0;
    bool has_range = false;
    const bool first_chunk = blockIdx.x < work.nlaunch;
    if (first_chunk)
        key = tile_key_of_block(blockIdx.x, work);
    else if (!tile_work_item(work, key, s, e))
        return;
    else
        has_range = true;
    __shared__ __attribute__((aligned(16))) float tile[3 * PM_FSTRIDE];  // 39.2 KB: (x, y) plane + z plane (ORDER > 0: first the potential tile)
    __shared__ unsigned s_max[2][BLK / 64];
    // REBUCKET: per destination, where the workgroup's slice of that tile's new range starts, its size, and the
    // next free slot in it
    __shared__ int64_t s_dest[27];
    __shared__ unsigned s_mine[27], s_slot[27];
    // CODES: movers of the workgroup, staged as destination (5 bits) | rank among the workgroup's movers to that
    // destination (11) | index in the range (16), and the booking of their list slots
    __shared__ unsigned s_mov[CODES ? PM_MOV_STAGE : 1], s_nmov, s_dcnt[27], s_dbase[27], s_dkey[27], s_nstay;
    const int tid = threadIdx.x;
    const int n = geo.n;
    const float nf = (float)n;
    const float scale = grad_scale<ORDER>(n);
    if constexpr (CODES) {
        if (tid < 27) s_dcnt[tid] = 0;
        if (tid == 0) s_nmov = 0, s_nstay = 0;  // (ordered before their first use by the barriers of the tile fill)
    }
    const TileId t = tile_of_key(key);
    constexpr int R = GradReach<ORDER>::R;
    // Potential tile: DX x DY rows of DZ = 24 floats, row r starting at global k = 16 tk - 4.  A row is six
    // aligned float4 (n is a multiple of 16, so the periodic wrap never splits a quad): the fill is 6-9
    // 16-byte loads per thread whose LDS address is just 16 * q, where the 22-float rows of single dwords
    // took 27 loads per thread and a divide-by-22 decode each (the kernel is VALU-bound).
    constexpr int DX = PM_HI + 2 * R, DY = PM_HJ + 2 * R, DZ = 24, KOFF = 3;  // cell lk sits at row position lk + KOFF
    static_assert(PM_TK == 16 && PM_HK + KOFF + R <= DZ && KOFF >= R, "row window 16 tk - 4 .. 16 tk + 19");
    constexpr int NPHI = DX * DY * DZ;   // 7392 (R=2), 9216 (R=3)
    constexpr int NQUAD = NPHI / 4;
    constexpr int NQP = ORDER == 0 ? 1 : (NQUAD + BLK - 1) / BLK;
    pm_v4f gphi[NQP];
    if constexpr (ORDER != 0) {  // all loads in flight before anything waits
        const int gi0 = t.ti * PM_TI - 1 - R, gj0 = t.tj * PM_TJ - 1 - R, gk0 = t.tk * PM_TK - 4;
#pragma unroll
        for (int u = 0; u < NQP; ++u) {
            // clamped, not guarded: behind a guard the compiler sinks the load to its LDS write and waits
            // for each one in turn
            const int qd = min(u * BLK + tid, NQUAD - 1);
            const int c4 = qd % 6, ab = qd / 6;
            gphi[u] = *reinterpret_cast<const pm_v4f *>(src + pm_cell(geo, gi0 + ab / DY, gj0 + ab % DY, gk0 + 4 * c4));
        }
    }
    unsigned ls = 0, le = 0;  // incoming stragglers (first workgroup of the tile only)
    if (first_chunk) {
        s = work.start[key];
        e = min(work.start[key + 1], s + PM_RANGE_CHUNK);
        has_range = s < e;
        if (lists.use) ls = lists.lstart[key], le = lists.lstart[key + 1];
        if (!has_range && ls == le) return;
    }
    if constexpr (REBUCKET) {
        // slices of the destination tiles' new ranges for the particles this workgroup serves from its range, as
        // the fused kick counted them (made visible to all by the barriers of the tile fill); listed particles
        // take their slots one by one
        if (tid < 27) {
            const unsigned mine = has_range ? rb.wg_hist[(size_t)blockIdx.x * 27 + tid] : 0u;
            s_mine[tid] = mine, s_slot[tid] = 0;
            if (mine) {
                const int Ti = n >> PM_LI, Tj = n >> PM_LJ, Tk = n >> PM_LK;
                const int di = tid / 9 - 1, dj = (tid / 3) % 3 - 1, dk = tid % 3 - 1;
                const unsigned dkey = tile_key((t.ti + di + Ti) % Ti, (t.tj + dj + Tj) % Tj, (t.tk + dk + Tk) % Tk);
                s_dest[tid] = rb.newstart[dkey] + (int64_t)atomicAdd(rb.fill + dkey, mine);
            }
        }
    }
    PosQueue q;
    VelQueue qv;
    if (ORDER == 0 && has_range) {
        q.prime_slots<2>(pos + 3 * s, tid, (unsigned)(e - s));  // (two slots in use: the range loop below)
        if (KICK) qv.q.prime_slots<2>(vel + 3 * s, tid, (unsigned)(e - s));
    }
    if constexpr (ORDER == 0) {
        // tile fill, float-granular: consecutive lanes read consecutive floats of a tile row (18 cells x 3
        // components = 216 contiguous bytes), so a wave load is a few full lines instead of 64 scattered
        // 12-byte pieces (PMC: the per-cell dwordx3 fill fetched 2.4x the tile bytes).  All loads are in
        // flight before the first LDS write.
        constexpr int NF = PM_TILE * 3, NQ = (NF + BLK - 1) / BLK;  // 9720 floats, 38 per thread
        float g[NQ];
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int f = u * BLK + tid;
            if (f < NF) {
                const int c = f / 3, m = f - 3 * c;
                g[u] = src[3 * tile_cell_global(c, t, geo) + m];
            }
        }
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int f = u * BLK + tid;
            if (f < NF) {
                const int c = f / 3, m = f - 3 * c;
                tile[force_tile_index(c, m)] = g[u];
            }
        }
    } else {
        static_assert(NPHI <= 3 * PM_FSTRIDE, "potential tile must fit in the force tile");
        float *ptile = tile;
#pragma unroll
        for (int u = 0; u < NQP; ++u) {
            const int qd = u * BLK + tid;
            if (qd < NQUAD) reinterpret_cast<pm_v4f *>(ptile)[qd] = gphi[u];
        }
        __syncthreads();
        // differentiate.  Lane -> cell mapping chosen for the LDS banks (ds_read_b32: 32 lanes per group on 32
        // slots): a half-wave takes 4 consecutive (i, j) rows x 8 consecutive k; with the 24-float rows of the
        // potential tile its 32 reads fall on 32 different slots for every stencil offset, where consecutive
        // cells (18 + 14 of two rows) always collided two-fold.  A wave = 4 rows x k 0..15; the two last
        // k-cells of every row go in a tail pass.
        constexpr int NROWS = PM_HI * PM_HJ, NBLK = NROWS / 4;  // 180 rows, 45 blocks of 4 rows x 16 k
        static_assert(NROWS % 4 == 0 && PM_HK == 18 && BLK == 256, "differentiation blocks");
        constexpr int NCM = (NBLK + 3) / 4, NTAIL = (2 * NROWS + BLK - 1) / BLK, NC = NCM + NTAIL;  // 12 + 2
        float gx[NC], gy[NC], gz[NC];
        const int wv = tid >> 6, ln = tid & 63;
        const int kk = (ln & 7) + ((ln >> 5) << 3), jr = (ln >> 3) & 3;
        // cell (row r = li * 18 + lj, k) of pass slot u, or -1
        auto cell_row = [&](int u) { return u < NCM ? ((u * 4 + wv) < NBLK ? 4 * (u * 4 + wv) + jr : -1)
                                                    : ((tid + (u - NCM) * BLK) < 2 * NROWS ? (tid + (u - NCM) * BLK) >> 1 : -1); };
        auto cell_k = [&](int u) { return u < NCM ? kk : 16 + (tid & 1); };
#pragma unroll
        for (int u = 0; u < NC; ++u) {
            const int r = cell_row(u), k = cell_k(u);
            if (r >= 0) {
                const int li = r / PM_HJ, lj = r - li * PM_HJ;
                const float *ctr = ptile + ((li + R) * DY + (lj + R)) * DZ + (k + KOFF);
                float sa[7], sb[7];
#pragma unroll
                for (int o = 0; o < 7; ++o) sa[o] = sb[o] = 0.f;
#pragma unroll
                for (int o = -R; o <= R; ++o) sa[o + 3] = ctr[o * DY * DZ];
                gx[u] = Stencil<ORDER, 0>::eval(sa, sb, 0.f, scale);
#pragma unroll
                for (int o = -R; o <= R; ++o) sa[o + 3] = ctr[o * DZ];
                gy[u] = Stencil<ORDER, 0>::eval(sa, sb, 0.f, scale);
#pragma unroll
                for (int o = -R; o <= R; ++o) sa[o + 3] = ctr[o];
                gz[u] = Stencil<ORDER, 0>::eval(sa, sb, 0.f, scale);
            }
        }
        // the particle queues are primed only now: held across the fill and the differentiation their 24
        // registers pushed the kernel into scratch (20 bytes per lane written and read back by every thread)
        if (has_range) {
            q.prime_slots<2>(pos + 3 * s, tid, (unsigned)(e - s));
            if (KICK) qv.q.prime_slots<2>(vel + 3 * s, tid, (unsigned)(e - s));
        }
        __syncthreads();  // every stencil has read its potential: the buffer becomes the force tile
#pragma unroll
        for (int u = 0; u < NC; ++u) {
            const int r = cell_row(u);
            if (r >= 0) force_tile_put(tile, r * PM_HK + cell_k(u), gx[u], gy[u], gz[u]);
        }
    }
    __syncthreads();
    // gather, second kick (utils.py:297, y += a*x) and the CFL maxima in one pass: the acceleration never
    // leaves its registers (PMC: re-reading it in a separate coalesced pass cost a further 12.9 GB of fetch)
    unsigned ma = 0, mv = 0;  // bit patterns of max|a|, max|v| (integer order == float order, NaN on top)
    // every lane also sums |a_x| over the particles it serves and a subset of the workgroups (below) add their sums to a
    // global one: the mean |a_x| of a sample, whose step-to-step ratio the caller extrapolates for the re-sort's
    // prediction (pm_set_predicted_acc_scale).  Summed unconditionally -- two VALU operations per particle: a
    // branch on `sample` inside the particle loop splits its body and the compiler then drains the prefetch
    // queues every iteration (the kernel took 29 ms instead of 17).
    // (every 64th workgroup of a large launch -- the atomics of more would queue up --, every 8th of a medium one, all of a
    // small one: a 16-plane slab of a 64^3 mesh has 32 tiles, and one of them is no sample)
    const unsigned smask = gridDim.x >= 8192u ? 63u : (gridDim.x >= 512u ? 7u : 0u);
    const bool sample = (blockIdx.x & smask) == 0u;
    float sabs = 0.f, scnt = 0.f;
    unsigned lost = 0u;  // a re-bucketing pass found no slot for a particle (the caller hears of it: pm_last_maxabs)
    __shared__ float s_sum[2][BLK / 64];
    // (o, w, p): where the particle's acceleration and velocity go -- the range's own base with a 32-bit
    // relative index, or the arrays' base with the 64-bit index of a listed straggler
    // `src_index`: the particle's index in the input arrays (only read for the tags of a re-bucketing pass)
    auto serve = [&](float *__restrict__ o, float *__restrict__ w3, auto p, float x, float y, float z, const float v[3],
                     bool own_range, int64_t src_index, int dcode_known) {
        // (the particle's destination code: from the prefetch queue of the range loop, or -- listed particles --
        // loaded here so that its latency hides behind the gather)
        const int dcode_p = dcode_known;
        const float xp = x * nf, yp = y * nf, zp = z * nf;
        const int i = cell_of(xp, n), j = cell_of(yp, n), k = cell_of(zp, n);
        const bool inside = in_tile(i, j, k, t);
        // A stranger in the own range is served by the tile it is in now (from its list).  It is carried
        // through the LDS gather on a dummy cell and only its stores are suppressed: an early exit here
        // costs 10 % of the kernel (the compiler stops overlapping the loads with the LDS reads).
        const bool skip = !inside && own_range && lists.use;
        float a[3];
        if (inside || skip) {
            gather_lds<SCHEME>(xp, yp, zp, skip ? t.ti * PM_TI : i, skip ? t.tj * PM_TJ : j, skip ? t.tk * PM_TK : k, t,
                               tile, a);
        } else {  // no lists prepared (or positions changed behind them): the slow path, still correct
            Acc3 r;
            if constexpr (ORDER == 0)
                r = gather_global<SCHEME>(src, xp, yp, zp, i, j, k, geo);
            else
                r = gather_phi_global<SCHEME, ORDER>(src, scale, xp, yp, zp, i, j, k, geo);
            a[0] = r.x, a[1] = r.y, a[2] = r.z;
        }
        if constexpr (REBUCKET) {
            // (listed particles only: a re-bucketing pass serves its ranges in the loop of its own below)
            // A listed particle (it is inside this tile, or the lists are void) was counted one by one by the fused
            // kick, so it takes its slot from the destination's counter itself -- the lanes of a wave that stay in
            // this tile (nearly all) with one global atomic between them
            const int dd_d = min(dcode_p, 26);
            const TileId ta = tile_of_cell(i, j, k);
            const int Ti = n >> PM_LI, Tj = n >> PM_LJ, Tk = n >> PM_LK;
            const unsigned dkey = tile_key((ta.ti + dd_d / 9 - 1 + Ti) % Ti, (ta.tj + (dd_d / 3) % 3 - 1 + Tj) % Tj,
                                           (ta.tk + dd_d % 3 - 1 + Tk) % Tk);
            const bool here = dkey == key;
            const unsigned long long m = __ballot(here);
            const unsigned before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            unsigned wb = 0;
            if (here && before == 0) wb = atomicAdd(rb.fill + key, (unsigned)__popcll(m));
            wb = __shfl((int)wb, m ? __ffsll((long long)m) - 1 : (tid & 63), 64);
            const unsigned slot = here ? wb + before : atomicAdd(rb.fill + dkey, 1u);
            int64_t d64 = rb.newstart[dkey] + (int64_t)slot;
            if (d64 >= rb.newstart[dkey + 1]) d64 = -1, lost = 1u;  // (codes changed behind the counts: memory safety)
            if (d64 >= 0) {
                float w[3];
#pragma unroll
                for (int mm = 0; mm < 3; ++mm) {
                    ma = max(ma, __float_as_uint(a[mm]) & 0x7fffffffu);
                    w[mm] = v[mm] + mhalf_dt * a[mm];
                    mv = max(mv, __float_as_uint(w[mm]) & 0x7fffffffu);
                }
                sabs += fabsf(a[0]), scnt += 1.0f;
                const pm_v3f vp = {x, y, z}, va = {a[0], a[1], a[2]}, vw = {w[0], w[1], w[2]};
                __builtin_nontemporal_store(vp, reinterpret_cast<pm_v3fu *>(rb.pos_out + 3 * d64));
                __builtin_nontemporal_store(va, reinterpret_cast<pm_v3fu *>(rb.acc_out + 3 * d64));
                __builtin_nontemporal_store(vw, reinterpret_cast<pm_v3fu *>(rb.vel_out + 3 * d64));
                if (rb.id) rb.id_out[d64] = rb.id[src_index];
            }
        } else if (!skip) {
            float *op = o + 3 * p, *wp = w3 + 3 * p;
            sabs += fabsf(a[0]), scnt += 1.0f;
            float w[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                op[m] = a[m];
                ma = max(ma, __float_as_uint(a[m]) & 0x7fffffffu);
                if (KICK) {
                    w[m] = v[m] + mhalf_dt * a[m];
                    wp[m] = w[m];
                    mv = max(mv, __float_as_uint(w[m]) & 0x7fffffffu);
                }
            }
            if constexpr (CODES) {
                // a listed particle (filed under another tile, served here): always a mover -- into the incoming list
                // of the tile it is expected in after the next drift, one slot at a time (they are few)
                const float xs[3] = {x, y, z};
                float veff[3];
#pragma unroll
                for (int m = 0; m < 3; ++m) veff[m] = w[m] + (-0.5f * cd.dtp) * a[m];
                const DestCode dd = dest_of(xs, veff, cd.dtp, nf, n, tile_of_cell(i, j, k));
                const unsigned slot = atomicAdd(cd.inc_count + dd.key, 1u);
                if (slot < cd.segstart[dd.key + 1] - cd.segstart[dd.key])
                    cd.entries[cd.segstart[dd.key] + slot] = (unsigned)src_index;
                else
                    *cd.overflow = 1u;
            }
        }
    };
    constexpr int QD = 2;  // queue slots in use (measured at 1024^3: 4 slots 16.3 ms, 3: 16.3, 2: 15.95, 1: 16.4)
    if constexpr (CODES) {
        // The range of a pass that stays in place and decides (struct Codes): the memory pipeline of the re-bucketing
        // loop below -- four register slots two particles ahead, one basic block as far as vector memory goes -- with
        // stores that need no slot: velocity and acceleration go where the particle was read (a wave's 64 records are
        // 768 contiguous bytes), the flag byte beside them.
        if (has_range) {
            const unsigned cnt = (unsigned)(e - s), last = cnt - 1u;
            const unsigned lane = (unsigned)tid & 63u;
            float *const sink = rb.sink + 3u * lane;
            unsigned char *const bsink = reinterpret_cast<unsigned char *>(rb.sink + PM_SINK_BYTES_AT) + lane;
            float *const vout = vel + 3 * s, *const aout = out + 3 * s;
            unsigned char *const fout = cd.flag + s;
            const float kick_next = -0.5f * cd.dtp;
            constexpr int NS = 4;
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), once (see the re-bucketing loop)
            unsigned nstay = 0;
#pragma unroll 1
            for (unsigned r0 = tid; (unsigned)__builtin_amdgcn_readfirstlane((int)r0) < cnt; r0 += NS * BLK) {
#pragma unroll
                for (int u = 0; u < NS; ++u) {
                    const unsigned r = r0 + u * BLK;
                    const float x = q.x[u], y = q.y[u], z = q.z[u];
                    const float v[3] = {qv.q.x[u], qv.q.y[u], qv.q.z[u]};
                    q.load((u + 2) % NS, r + 2 * BLK);
                    qv.q.load((u + 2) % NS, r + 2 * BLK);
                    float a[3] = {0.f, 0.f, 0.f};
                    bool mine = false, stay = false;
                    int dcode = 13;
                    if ((unsigned)__builtin_amdgcn_readfirstlane((int)r) < cnt) {
                        const float xp = x * nf, yp = y * nf, zp = z * nf;
                        const int i = cell_of(xp, n), j = cell_of(yp, n), k = cell_of(zp, n);
                        // (a stranger in the own range is served by the tile it is in now, from its list)
                        mine = r < cnt && in_tile(i, j, k, t);
                        gather_lds<SCHEME>(xp, yp, zp, mine ? i : t.ti * PM_TI, mine ? j : t.tj * PM_TJ, mine ? k : t.tk * PM_TK,
                                           t, tile, a);
                    }
                    float w[3];
#pragma unroll
                    for (int mm = 0; mm < 3; ++mm) w[mm] = v[mm] + mhalf_dt * a[mm];
                    if (mine) {
#pragma unroll
                        for (int mm = 0; mm < 3; ++mm) {
                            ma = max(ma, __float_as_uint(a[mm]) & 0x7fffffffu);
                            mv = max(mv, __float_as_uint(w[mm]) & 0x7fffffffu);
                        }
                        sabs += fabsf(a[0]), scnt += 1.0f;
                        // where the next drift takes it: x + dt' (w - dt'/2 a), clamped to one cell (dest_of)
                        const float xs[3] = {x, y, z};
                        float veff[3];
#pragma unroll
                        for (int mm = 0; mm < 3; ++mm) veff[mm] = w[mm] + kick_next * a[mm];
#ifndef PM_EXP_NODEST
                        dcode = dest_of(xs, veff, cd.dtp, nf, n, t).d;
#endif
                        stay = dcode == 13;
                        nstay += stay ? 1u : 0u;
                        if (!stay) {  // (LDS only: a branch the counted waits need not know about)
                            const unsigned at = atomicAdd(&s_nmov, 1u);
                            if (at < (unsigned)PM_MOV_STAGE) {
                                const unsigned rank = atomicAdd(&s_dcnt[dcode], 1u);  // (< PM_MOV_STAGE: 11 bits)
                                s_mov[at] = ((unsigned)dcode << 27) | (rank << 16) | r;
                            } else {  // the staging area is full: book the slot now
                                const int Ti = n >> PM_LI, Tj = n >> PM_LJ, Tk = n >> PM_LK;
                                const unsigned dkey = tile_key((t.ti + dcode / 9 - 1 + Ti) % Ti, (t.tj + (dcode / 3) % 3 - 1 + Tj) % Tj,
                                                               (t.tk + dcode % 3 - 1 + Tk) % Tk);
                                const unsigned slot = atomicAdd(cd.inc_count + dkey, 1u);
                                if (slot < cd.segstart[dkey + 1] - cd.segstart[dkey])
                                    cd.entries[cd.segstart[dkey] + slot] = (unsigned)(s + r);
                                else
                                    *cd.overflow = 1u;
                            }
                        }
                    }
                    const pm_v3f va = {a[0], a[1], a[2]}, vw = {w[0], w[1], w[2]};
                    const unsigned long long mm_ = __ballot(mine);
                    __builtin_nontemporal_store(va, (pm_gv3fu *)select_ptr(mm_, aout + 3u * r, sink));
                    __builtin_nontemporal_store(vw, (pm_gv3fu *)select_ptr(mm_, vout + 3u * r, sink));
                    *select_byte_ptr(__ballot(r < cnt), fout + r, bsink) = stay ? (unsigned char)13 : (unsigned char)0;
                }
            }
            // stayers of the range: one count per workgroup, added to the tile's
            for (int off = 32; off > 0; off >>= 1) nstay += (unsigned)__shfl_down((int)nstay, off, 64);
            if (lane == 0) atomicAdd(&s_nstay, nstay);
        }
    } else if constexpr (REBUCKET) {
        // The range of a re-bucketing pass.  What this loop is made for is that a wave NEVER drains its memory queue:
        // vector-memory operations retire in order, so a wait for one load is a wait for every load and every STORE
        // issued before it.  (Until round 3 the destination byte was loaded inside the iteration that used it and the
        // stores were inline asm, which the compiler's wait counting does not see: every particle ended in
        // s_waitcnt vmcnt(0), i.e. behind the acknowledgement of its predecessor's three stores and the arrival of
        // the refills just issued -- two memory round trips per particle and wave, with four waves per SIMD to hide
        // them.)  Position, velocity and destination byte are now loaded two particles ahead, the stores are
        // ordinary (nontemporal) stores, and the body is ONE basic block as far as vector memory goes -- lanes
        // without a particle (the tail of the range, strangers the lists serve) run along on a dummy cell and store
        // to a 768-byte sink instead of branching around the stores -- so the counted waits the compiler derives
        // (vmcnt(N), N = everything issued since) leave the stores and the refills in flight.
        if (has_range) {
            const unsigned cnt = (unsigned)(e - s), last = cnt - 1u;
            const unsigned char *const dbase = rb.dcode + s;
            const unsigned lane = (unsigned)tid & 63u;
            float *const sink = rb.sink + 3u * lane;
            // Four register slots, two particles ahead: body u takes its particle from slot u and refills slot (u + 2) % 4,
            // which was emptied two bodies ago -- every load lands in the registers it is consumed from (no copies behind a
            // wait for a load just issued, which is what two slots refilled at the point of use came to) and has two
            // gathers' time to arrive.
            constexpr int NS = 4;
            unsigned dq[NS];  // (a dword each: as bytes the compiler packs the four into one register at the bottom of the
                              // loop, behind a wait for the loads)
#pragma unroll
            for (int u = 0; u < 2; ++u) dq[u] = dbase[min((unsigned)tid + u * BLK, last)];
            dq[2] = dq[3] = 13u;
            // (the primed loads are waited for here, once: the compiler's counted waits inside the loop are the minimum over
            // the ways into it, and coming from the priming there is next to nothing behind a slot's loads to count on)
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll 1
            for (unsigned r0 = tid; (unsigned)__builtin_amdgcn_readfirstlane((int)r0) < cnt; r0 += NS * BLK) {
#pragma unroll
                for (int u = 0; u < NS; ++u) {
                    const unsigned r = r0 + u * BLK;
                    const float x = q.x[u], y = q.y[u], z = q.z[u];
                    const float v[3] = {qv.q.x[u], qv.q.y[u], qv.q.z[u]};
                    const int dd_d = (int)min(dq[u], 26u);
                    q.load((u + 2) % NS, r + 2 * BLK);
                    qv.q.load((u + 2) % NS, r + 2 * BLK);
                    dq[(u + 2) % NS] = dbase[min(r + 2 * BLK, last)];
                    float a[3] = {0.f, 0.f, 0.f};
                    bool ok = false;
                    int64_t d64 = 0;
                    // (a wave past the end of the range -- the last trip of the loop -- skips the arithmetic, not the loads
                    // and stores: a branch around those would be a branch the counted waits have to allow for)
                    if ((unsigned)__builtin_amdgcn_readfirstlane((int)r) < cnt) {
                        const float xp = x * nf, yp = y * nf, zp = z * nf;
                        const int i = cell_of(xp, n), j = cell_of(yp, n), k = cell_of(zp, n);
                        // (a stranger in the own range is served by the tile it is in now, from its list)
                        const bool mine = r < cnt && in_tile(i, j, k, t);
                        gather_lds<SCHEME>(xp, yp, zp, mine ? i : t.ti * PM_TI, mine ? j : t.tj * PM_TJ, mine ? k : t.tk * PM_TK,
                                           t, tile, a);
                        // the destination the fused kick counted for it; stayers (nearly all) take consecutive slots of
                        // the own slice with one LDS atomic per wave and iteration, movers one each
                        const bool stay = mine && dd_d == 13;
                        const unsigned long long m = __ballot(stay);
                        const unsigned before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        unsigned wb = 0;
                        if (stay && before == 0) wb = atomicAdd(&s_slot[13], (unsigned)__popcll(m));  // the first staying lane
                        wb = __shfl((int)wb, m ? __ffsll((long long)m) - 1 : (int)lane, 64);
                        unsigned slot = wb + before;
                        if (mine && !stay) slot = atomicAdd(&s_slot[dd_d], 1u);
                        // (slot < s_mine always, unless the positions were changed behind the prepared lists: memory safety)
                        ok = mine && slot < s_mine[dd_d];
                        lost |= (unsigned)(mine && !ok);
                        d64 = s_dest[dd_d] + (int64_t)slot;
                    }
                    float w[3];
#pragma unroll
                    for (int mm = 0; mm < 3; ++mm) w[mm] = v[mm] + mhalf_dt * a[mm];
                    if (ok) {
#pragma unroll
                        for (int mm = 0; mm < 3; ++mm) {
                            ma = max(ma, __float_as_uint(a[mm]) & 0x7fffffffu);
                            mv = max(mv, __float_as_uint(w[mm]) & 0x7fffffffu);
                        }
                        sabs += fabsf(a[0]), scnt += 1.0f;
                    }
                    const pm_v3f vp = {x, y, z}, va = {a[0], a[1], a[2]}, vw = {w[0], w[1], w[2]};
                    // (the select is two v_cndmask written out: from `ok ? p : sink` the compiler makes a branch around
                    // each store, and a store that may not have been issued is a store its wait counting cannot rely on)
                    const unsigned long long okm = __ballot(ok);
                    __builtin_nontemporal_store(vp, (pm_gv3fu *)select_ptr(okm, rb.pos_out + 3 * d64, sink));
                    __builtin_nontemporal_store(va, (pm_gv3fu *)select_ptr(okm, rb.acc_out + 3 * d64, sink));
                    __builtin_nontemporal_store(vw, (pm_gv3fu *)select_ptr(okm, rb.vel_out + 3 * d64, sink));
                    if (rb.id) {
                        if (ok) rb.id_out[d64] = rb.id[s + r];
                    }
                }
            }
        }
    } else if (has_range) {
        float *const o = out + 3 * s, *const w3 = KICK ? vel + 3 * s : nullptr;
        const unsigned cnt = (unsigned)(e - s);
        // QD particles per trip, one per queue slot: slot u is consumed and refilled in place, where popping the head
        // and shifting the queue cost 18 register moves per particle (gather 17.2 -> 16.5 ms at 1024^3)
#pragma unroll 1
        for (unsigned r0 = tid; r0 < cnt; r0 += QD * BLK) {
#pragma unroll
            for (int u = 0; u < QD; ++u) {
                const unsigned r = r0 + u * BLK;
                if (r < cnt) {
                    const float x = q.x[u], y = q.y[u], z = q.z[u];
                    float v[3] = {0.f, 0.f, 0.f};
                    if (KICK) v[0] = qv.q.x[u], v[1] = qv.q.y[u], v[2] = qv.q.z[u];
                    q.load(u, r + QD * BLK);
                    if (KICK) qv.q.load(u, r + QD * BLK);
                    serve(o, w3, r, x, y, z, v, true, s + r, 13);
                }
            }
        }
    }
#pragma unroll 1
    for (unsigned l = ls + tid; l < le; l += BLK) {
        const int64_t p = lists.idx[l];
        float v[3] = {0.f, 0.f, 0.f};
        if (KICK) v[0] = vel[3 * p], v[1] = vel[3 * p + 1], v[2] = vel[3 * p + 2];
        serve(out, vel, p, pos[3 * p], pos[3 * p + 1], pos[3 * p + 2], v, false, p, REBUCKET ? (int)rb.dcode[p] : 13);
    }
    for (int off = 32; off > 0; off >>= 1) {
        ma = max(ma, (unsigned)__shfl_down((int)ma, off, 64));
        mv = max(mv, (unsigned)__shfl_down((int)mv, off, 64));
    }
    for (int off = 32; off > 0; off >>= 1) {
        sabs += __shfl_down(sabs, off, 64);
        scnt += __shfl_down(scnt, off, 64);
    }
    if ((tid & 63) == 0) s_max[0][tid >> 6] = ma, s_max[1][tid >> 6] = mv, s_sum[0][tid >> 6] = sabs, s_sum[1][tid >> 6] = scnt;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < BLK / 64; ++w) ma = max(ma, s_max[0][w]), mv = max(mv, s_max[1][w]);
        // one same-address atomic costs ~12 ns at the L2 and half a million workgroups would queue up on
        // it; the maximum only grows, so a workgroup that cannot raise it (nearly all) just reads it
        if (ma > __atomic_load_n(maxbits, __ATOMIC_RELAXED)) atomicMax(maxbits, ma);
        if (KICK && mv > __atomic_load_n(maxbits + 1, __ATOMIC_RELAXED)) atomicMax(maxbits + 1, mv);
        // the sample's sums live on a cache line of their own (PM_SAMPLE_WORD): on the line of the maxima, which
        // every workgroup reads, 260 000 atomics took the kernel from 17 to 30 ms
        if (sample) {
            for (int w = 1; w < BLK / 64; ++w) sabs += s_sum[0][w], scnt += s_sum[1][w];
            if (scnt > 0.f) {
                atomicAdd(reinterpret_cast<double *>(maxbits + PM_SAMPLE_WORD), (double)sabs);
                atomicAdd(maxbits + PM_SAMPLE_WORD + 2, (unsigned)scnt);
            }
        }
    }
    if constexpr (REBUCKET) {  // (never, unless positions or codes were changed behind the prepared counts)
        const unsigned long long lm = __ballot(lost != 0u);
        if (lm && (tid & 63) == 0) atomicAdd(maxbits + PM_SAMPLE_WORD + 3, (unsigned)__popcll(lm));
    }
    if constexpr (CODES) {
        // (the barrier above has every lane's staging and counting behind it)
        if (tid == 0) {
            cd.wg_stay[blockIdx.x + work.boff] = has_range ? s_nstay : 0u;
            if (has_range && s_nstay) atomicAdd(cd.tile_stay + key, s_nstay);
        }
        if (tid < 27 && s_dcnt[tid]) {  // one booking per destination: the workgroup's movers there sit side by side
            const int Ti = n >> PM_LI, Tj = n >> PM_LJ, Tk = n >> PM_LK;
            const unsigned dkey = tile_key((t.ti + tid / 9 - 1 + Ti) % Ti, (t.tj + (tid / 3) % 3 - 1 + Tj) % Tj,
                                           (t.tk + tid % 3 - 1 + Tk) % Tk);
            s_dkey[tid] = dkey, s_dbase[tid] = atomicAdd(cd.inc_count + dkey, s_dcnt[tid]);
        }
        __syncthreads();
        const unsigned nm = min(s_nmov, (unsigned)PM_MOV_STAGE);
        for (unsigned i = tid; i < nm; i += BLK) {
            const unsigned m = s_mov[i], d = m >> 27, rank = (m >> 16) & 2047u, r = m & 0xffffu;
            const unsigned dkey = s_dkey[d], slot = s_dbase[d] + rank;
            if (slot < cd.segstart[dkey + 1] - cd.segstart[dkey])
                cd.entries[cd.segstart[dkey] + slot] = (unsigned)(s + r);
            else
                *cd.overflow = 1u;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launchers

template <int SCHEME>
static int deposit_launch(pm_ctx *c, const float *pos, int64_t np, float *rho, int sorted_hint) {
    const int n = c->n;
    if (sorted_hint == 1 && n >= 16) {  // one workgroup per tile (Morton-sorted streams)
        TileWork work;
        TileLists lists;
        unsigned blocks = 0;
        int rc = tile_work_for(c, pos, np, &work, &lists, &blocks);
        if (rc) return rc;
        hipLaunchKernelGGL((deposit_ranges_kernel<SCHEME>), dim3(blocks), dim3(BLK), 0, c->stream, pos, work, lists,
                           pm_ctx_geom(c), rho);
    } else if (sorted_hint && n >= 16) {
        const int64_t blocks = (np + PM_CHUNK - 1) / PM_CHUNK;
        hipLaunchKernelGGL((deposit_tiled_kernel<SCHEME>), dim3((unsigned)blocks), dim3(BLK), 0, c->stream, pos, np,
                           pm_ctx_geom(c), rho);
    } else {
        const int blocks = pm_stream_blocks(np, BLK);
        hipLaunchKernelGGL((deposit_atomic_kernel<SCHEME>), dim3(blocks), dim3(BLK), 0, c->stream, pos, np,
                           pm_ctx_geom(c), rho);
    }
    PM_LAUNCH_CHECK();
    return PM_OK;
}

extern "C" int pm_deposit(pm_ctx *c, int scheme, const float *d_pos, int64_t np, float *d_rho, int sorted_hint) {
    PM_REQUIRE(c && d_rho && np >= 0 && (d_pos || np == 0), "bad argument");
    PM_REQUIRE((np + PM_CHUNK - 1) / PM_CHUNK < 2147483647LL, "too many particles for one launch");
    const size_t cells = (size_t)c->slab_nx * c->n * c->n;
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

template <int SCHEME>
static int interp_kick_launch(pm_ctx *c, const float *grid, const float *pos, int64_t np, float *out, float *vel,
                              float half_dt) {
    const int64_t blocks = (np + PM_CHUNK - 1) / PM_CHUNK;
    unsigned *mb = c->d_scratch + 8;
    if (c->gather_mode == 1) {  // one workgroup per tile
        TileWork work;
        TileLists lists;
        unsigned nblk = 0;
        int rc = tile_work_for(c, pos, np, &work, &lists, &nblk);
        if (rc) return rc;
        if (vel)
            hipLaunchKernelGGL((interp_ranges_kernel<SCHEME, 0, true>), dim3(nblk), dim3(BLK), 0, c->stream, grid, pos,
                               work, lists, pm_ctx_geom(c), out, vel, -half_dt, mb, Rebucket{});
        else
            hipLaunchKernelGGL((interp_ranges_kernel<SCHEME, 0, false>), dim3(nblk), dim3(BLK), 0, c->stream, grid, pos,
                               work, lists, pm_ctx_geom(c), out, vel, 0.f, mb, Rebucket{});
        PM_LAUNCH_CHECK();
        return PM_OK;
    }
    if (vel)
        hipLaunchKernelGGL((interp_tiled_kernel<SCHEME, true>), dim3((unsigned)blocks), dim3(BLK), 0, c->stream, grid,
                           pos, np, pm_ctx_geom(c), out, vel, -half_dt, mb);
    else
        hipLaunchKernelGGL((interp_tiled_kernel<SCHEME, false>), dim3((unsigned)blocks), dim3(BLK), 0, c->stream, grid,
                           pos, np, pm_ctx_geom(c), out, vel, 0.f, mb);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

extern "C" int pm_interp_kick(pm_ctx *c, int scheme, const float *d_force, const float *d_pos, int64_t np,
                              float *d_acc, float *d_vel, float half_dt) {
    PM_REQUIRE(c && d_force && np >= 0 && ((d_pos && d_acc) || np == 0), "bad argument");
    PM_REQUIRE(c->n >= 16, "pm_interp_kick needs n >= 16 (use pm_interp + pm_axpy below that)");
    PM_REQUIRE((np + PM_CHUNK - 1) / PM_CHUNK < 2147483647LL, "too many particles for one launch");
    PM_HIP(hipMemsetAsync(c->d_scratch + 8, 0, 2 * sizeof(unsigned), c->stream));
    PM_HIP(hipMemsetAsync(c->d_scratch + 8 + PM_SAMPLE_WORD, 0, 4 * sizeof(unsigned), c->stream));
    if (np == 0) return PM_OK;
    switch (scheme) {
        case PM_NGP: return interp_kick_launch<PM_NGP>(c, d_force, d_pos, np, d_acc, d_vel, half_dt);
        case PM_CIC: return interp_kick_launch<PM_CIC>(c, d_force, d_pos, np, d_acc, d_vel, half_dt);
        case PM_TSC: return interp_kick_launch<PM_TSC>(c, d_force, d_pos, np, d_acc, d_vel, half_dt);
    }
    pm_set_error("mass scheme %d, should be NGP(1), CIC(2) or TSC(3)", scheme);
    return PM_EINVAL;
}

template <int SCHEME, int ORDER>
static int grad_interp_kick_launch(pm_ctx *c, const float *phi, pm_geom geo, const float *pos, int64_t np, float *out,
                                   float *vel, float half_dt) {
    unsigned *mb = c->d_scratch + 8;
    TileWork work;
    TileLists lists;
    unsigned nblk = 0;
    int rc = tile_work_for(c, pos, np, &work, &lists, &nblk);
    if (rc) return rc;
    if (vel)
        hipLaunchKernelGGL((interp_ranges_kernel<SCHEME, ORDER, true>), dim3(nblk), dim3(BLK), 0, c->stream, phi, pos,
                           work, lists, geo, out, vel, -half_dt, mb, Rebucket{});
    else
        hipLaunchKernelGGL((interp_ranges_kernel<SCHEME, ORDER, false>), dim3(nblk), dim3(BLK), 0, c->stream, phi, pos,
                           work, lists, geo, out, vel, 0.f, mb, Rebucket{});
    PM_LAUNCH_CHECK();
    return PM_OK;
}

template <int SCHEME>
static int grad_interp_kick_order(pm_ctx *c, int order, const float *phi, pm_geom geo, const float *pos, int64_t np,
                                  float *out, float *vel, float half_dt) {
    switch (order) {
        case 2: return grad_interp_kick_launch<SCHEME, 2>(c, phi, geo, pos, np, out, vel, half_dt);
        case 3: return grad_interp_kick_launch<SCHEME, 3>(c, phi, geo, pos, np, out, vel, half_dt);
        case 5: return grad_interp_kick_launch<SCHEME, 5>(c, phi, geo, pos, np, out, vel, half_dt);
        case 7: return grad_interp_kick_launch<SCHEME, 7>(c, phi, geo, pos, np, out, vel, half_dt);
    }
    pm_set_error("gradient order %d, should be 2, 3, 5 or 7", order);
    return PM_EINVAL;
}

extern "C" int pm_gradient_interp_kick(pm_ctx *c, int order, int scheme, const float *d_phi, int nx_phi, int x0_phi,
                                       const float *d_pos, int64_t np, float *d_acc, float *d_vel, float half_dt) {
    PM_REQUIRE(c && d_phi && np >= 0 && ((d_pos && d_acc) || np == 0), "bad argument");
    PM_REQUIRE(c->n >= 16, "pm_gradient_interp_kick needs n >= 16 (use pm_gradient + pm_interp below that)");
    PM_REQUIRE(nx_phi >= 1 && nx_phi <= c->n, "bad potential slab");
    PM_REQUIRE(((uintptr_t)d_phi & 15u) == 0, "d_phi must be 16-byte aligned (the tile fill reads float4)");
    PM_REQUIRE((np + PM_CHUNK - 1) / PM_CHUNK < 2147483647LL, "too many particles for one launch");
    PM_HIP(hipMemsetAsync(c->d_scratch + 8, 0, 2 * sizeof(unsigned), c->stream));
    PM_HIP(hipMemsetAsync(c->d_scratch + 8 + PM_SAMPLE_WORD, 0, 4 * sizeof(unsigned), c->stream));
    if (np == 0) return PM_OK;
    pm_geom geo = pm_ctx_geom(c);  // tiles are filed on the full mesh; phi is the (slab + ghosts) array given
    geo.nx = nx_phi, geo.x0 = ((x0_phi % c->n) + c->n) % c->n;
    switch (scheme) {
        case PM_NGP: return grad_interp_kick_order<PM_NGP>(c, order, d_phi, geo, d_pos, np, d_acc, d_vel, half_dt);
        case PM_CIC: return grad_interp_kick_order<PM_CIC>(c, order, d_phi, geo, d_pos, np, d_acc, d_vel, half_dt);
        case PM_TSC: return grad_interp_kick_order<PM_TSC>(c, order, d_phi, geo, d_pos, np, d_acc, d_vel, half_dt);
    }
    pm_set_error("mass scheme %d, should be NGP(1), CIC(2) or TSC(3)", scheme);
    return PM_EINVAL;
}

// ---- incremental re-sort: the same kernel writing the particle arrays out of place, tile after tile ----------

__global__ __launch_bounds__(BLK) void rebucket_counts_kernel(const unsigned *__restrict__ tile_count, unsigned ntiles,
                                                              int64_t *__restrict__ cnt) {
    const unsigned t = blockIdx.x * BLK + threadIdx.x;
    if (t > ntiles) return;
    cnt[t] = t < ntiles ? (int64_t)tile_count[t] : 0;
}

template <int SCHEME, int ORDER>
static int rebucket_launch(pm_ctx *c, const float *phi, pm_geom geo, const float *pos, float *vel, int64_t np,
                           float half_dt, const TileWork &work, const TileLists &lists, unsigned nblk, const Rebucket &rb) {
    hipLaunchKernelGGL((interp_ranges_kernel<SCHEME, ORDER, true, true>), dim3(nblk), dim3(BLK), 0, c->stream, phi, pos, work,
                       lists, geo, (float *)nullptr, vel, -half_dt, c->d_scratch + 8, rb);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

template <int SCHEME>
static int rebucket_order(pm_ctx *c, int order, const float *phi, pm_geom geo, const float *pos, float *vel, int64_t np,
                          float half_dt, const TileWork &work, const TileLists &lists, unsigned nblk, const Rebucket &rb) {
    switch (order) {
        case 0: return rebucket_launch<SCHEME, 0>(c, phi, geo, pos, vel, np, half_dt, work, lists, nblk, rb);  // force grid
        case 2: return rebucket_launch<SCHEME, 2>(c, phi, geo, pos, vel, np, half_dt, work, lists, nblk, rb);
        case 3: return rebucket_launch<SCHEME, 3>(c, phi, geo, pos, vel, np, half_dt, work, lists, nblk, rb);
        case 5: return rebucket_launch<SCHEME, 5>(c, phi, geo, pos, vel, np, half_dt, work, lists, nblk, rb);
        case 7: return rebucket_launch<SCHEME, 7>(c, phi, geo, pos, vel, np, half_dt, work, lists, nblk, rb);
    }
    pm_set_error("gradient order %d, should be 0 (force grid), 2, 3, 5 or 7", order);
    return PM_EINVAL;
}

// Destination counts from the codes alone (pm_rebucket_prepare): what the fused kick counts on a single GPU, for a
// stream whose particles have moved between arrays since their codes were recorded (a slab's migration: the code,
// relative to the tile the particle is in, travels with it).  Same workgroup -> range mapping as the gather; a
// destination outside the window of tiles this rank serves is turned into "stays" (the particle is about to leave
// the rank; filed under an outside tile no workgroup would ever serve it).
__global__ __launch_bounds__(BLK) void rebucket_prepare_kernel(const float *__restrict__ pos, TileWork work, TileLists lists,
                                                               int n, unsigned char *__restrict__ codes,
                                                               unsigned *__restrict__ tile_count,
                                                               unsigned *__restrict__ wg_hist) {
    unsigned key = 0;
    int64_t s = 0, e = 0;
    const bool first = (blockIdx.x + work.boff) < work.nlaunch;
    const bool has_range = tile_work_item(work, key, s, e);
    if (!first && !has_range) return;
    __shared__ unsigned s_hist[27];
    const int tid = threadIdx.x;
    const float nf = (float)n;
    const TileId t = tile_of_key(key);
    const int Ti = n >> PM_LI, Tj = n >> PM_LJ, Tk = n >> PM_LK;
    const bool windowed = work.nlaunch != work.ntiles;
    const int w_lo = work.ti0, w_hi = work.ti0 + (1 << work.lnti);
    if (tid < 27) s_hist[tid] = 0;
    __syncthreads();
    // code of particle p, which is in tile ta, made safe (a byte the caller owns) and kept inside the window
    auto code_of = [&](int64_t p, const TileId &ta) {
        int d = min((int)codes[p], 26);
        if (windowed) {
            const int ti = (ta.ti + d / 9 - 1 + Ti) % Ti;
            if (ti < w_lo || ti >= w_hi) d = 13;
        }
        if (d != (int)codes[p]) codes[p] = (unsigned char)d;
        return d;
    };
    auto key_of = [&](const TileId &ta, int d) {
        return tile_key((ta.ti + d / 9 - 1 + Ti) % Ti, (ta.tj + (d / 3) % 3 - 1 + Tj) % Tj, (ta.tk + d % 3 - 1 + Tk) % Tk);
    };
    if (has_range)
        for (int64_t p0 = s; p0 < e; p0 += BLK) {  // (whole waves in step: the ballot below)
            const int64_t p = p0 + tid;
            int d = -1;  // strangers and the tail: counted by the tile they are in / not at all
            if (p < e) {
                const int i = cell_of(pos[3 * p] * nf, n), j = cell_of(pos[3 * p + 1] * nf, n), k = cell_of(pos[3 * p + 2] * nf, n);
                if (in_tile(i, j, k, t)) d = code_of(p, t);
            }
            // nearly everybody stays: one LDS atomic per wave for those (64 lanes on one address serialise)
            const unsigned long long m = __ballot(d == 13);
            if (m && (tid & 63) == __ffsll((long long)m) - 1) atomicAdd(&s_hist[13], (unsigned)__popcll(m));
            if (d >= 0 && d != 13) atomicAdd(&s_hist[d], 1u);
        }
    if (first && lists.use) {
        const unsigned ls = lists.lstart[key], le = lists.lstart[key + 1];
        for (unsigned l = ls + tid; l < le; l += BLK) {
            const int64_t p = lists.idx[l];
            const int i = cell_of(pos[3 * p] * nf, n), j = cell_of(pos[3 * p + 1] * nf, n), k = cell_of(pos[3 * p + 2] * nf, n);
            const TileId ta = tile_of_cell(i, j, k);
            atomicAdd(tile_count + key_of(ta, code_of(p, ta)), 1u);
        }
    }
    __syncthreads();
    if (tid < 27) {
        const unsigned cnt_d = s_hist[tid];
        wg_hist[(size_t)(blockIdx.x + work.boff) * 27 + tid] = cnt_d;
        if (cnt_d) atomicAdd(tile_count + key_of(t, tid), cnt_d);
    }
}

// Where the fused kick + drift + deposit of a SLAB records every particle's destination code (one byte per
// particle, NULL: nowhere); see pm_rebucket_prepare.
extern "C" int pm_set_destination_codes(pm_ctx *c, unsigned char *d_codes) {
    PM_REQUIRE(c, "NULL argument");
    c->codes_user = d_codes;
    return PM_OK;
}

// Counts the destinations recorded in d_codes for the particles d_pos / np as they are filed NOW, for the
// re-bucketing gather that follows.  Needs complete tile lists for exactly d_pos / np (pm_tile_lists).
extern "C" int pm_rebucket_prepare(pm_ctx *c, const float *d_pos, int64_t np, unsigned char *d_codes) {
    PM_REQUIRE(c && d_pos && d_codes && np > 0 && np < 4294967296LL, "bad argument");
    PM_REQUIRE(c->n >= 16, "pm_rebucket_prepare needs n >= 16");
    PM_REQUIRE(c->lists_pos == d_pos && c->lists_np == np && c->lists_use == 1,
               "no complete tile lists prepared for these positions (pm_tile_lists on the same d_pos first)");
    TileWork work;
    TileLists lists;
    unsigned nblk = 0;
    RbBufs b;
    int rc = rebucket_bufs(c, (unsigned)(((int64_t)c->n * c->n * c->n) >> (PM_LI + PM_LJ + PM_LK)), np, &b);  // (first: it may move)
    if (rc) return rc;
    PM_REQUIRE(c->lists_pos == d_pos, "the tile lists read a table that has just moved: call pm_tile_lists again");
    rc = tile_work_for(c, d_pos, np, &work, &lists, &nblk);
    if (rc) return rc;
    PM_HIP(hipMemsetAsync(b.tile_count, 0, ((size_t)work.ntiles + 1) * sizeof(unsigned), c->stream));
    hipLaunchKernelGGL(rebucket_prepare_kernel, dim3(nblk), dim3(BLK), 0, c->stream, d_pos, work, lists, c->n, d_codes,
                       b.tile_count, b.wg_hist);
    PM_LAUNCH_CHECK();
    c->stay_valid = 1, c->stay_codes = d_codes;
    return PM_OK;
}

// pm_gradient_interp_kick with the kick, writing position, kicked velocity and acceleration (and the optional
// tags) of every particle to the *_out arrays in tile order (pmstep.h).  Needs the ranges, lists and stay counts
// pm_kick_drift_wrap_deposit[_khalo] (or pm_tile_lists + pm_rebucket_prepare) left prepared for exactly d_pos / np.
static int gradient_interp_kick_rebucket(pm_ctx *c, int order, int scheme, const float *d_phi, pm_geom geo, const float *d_pos,
                                         const float *d_vel, int64_t np, float half_dt, float *d_pos_out, float *d_vel_out,
                                         float *d_acc_out, const unsigned *d_id, unsigned *d_id_out) {
    PM_REQUIRE(c && d_phi && d_pos && d_vel && d_pos_out && d_vel_out && d_acc_out && np > 0, "bad argument");
    PM_REQUIRE((!d_id) == (!d_id_out), "d_id / d_id_out must come as a pair");
    PM_REQUIRE(d_pos_out != d_pos && d_vel_out != d_vel, "the re-bucketed arrays are written out of place");
    PM_REQUIRE(c->n >= 16, "the re-bucketing gather needs n >= 16");
    PM_REQUIRE(((uintptr_t)d_phi & 15u) == 0, "d_phi must be 16-byte aligned (the tile fill reads float4)");
    PM_REQUIRE(c->lists_pos == d_pos && c->lists_np == np && c->lists_use == 1 && c->stay_valid && c->stay_codes,
               "no stay counts prepared for these positions (pm_kick_drift_wrap_deposit[_khalo] on the same d_pos, or "
               "pm_tile_lists + pm_rebucket_prepare, first)");
    TileWork work;
    TileLists lists;
    unsigned nblk = 0;
    RbBufs b;
    // (sized for this np by whoever prepared the counts -- the fused kick, pm_tile_lists_codes, pm_rebucket_prepare --
    // so this call only hands out the pointers; were it to grow the buffer, the counts would be gone: checked below)
    int rc = rebucket_bufs(c, (unsigned)(((int64_t)c->n * c->n * c->n) >> (PM_LI + PM_LJ + PM_LK)), np, &b);
    if (rc) return rc;
    PM_REQUIRE(c->stay_valid && c->lists_pos == d_pos, "the re-sort's buffer moved under the prepared counts");
    rc = tile_work_for(c, d_pos, np, &work, &lists, &nblk);
    if (rc) return rc;
    const unsigned ntiles = work.ntiles;
    int64_t *table = b.table[c->rb_cur ^ 1];  // (work.start may be the other table: this stream's own layout)
    const unsigned tb = (ntiles + 1 + BLK - 1) / BLK;
    hipLaunchKernelGGL(rebucket_counts_kernel, dim3(tb), dim3(BLK), 0, c->stream, b.tile_count, ntiles, b.cnt);
    PM_LAUNCH_CHECK();
    size_t tmp = b.scan_tmp_bytes;
    PM_HIP(rocprim::exclusive_scan(b.scan_tmp, tmp, b.cnt, table, (int64_t)0, (size_t)ntiles + 1, rocprim::plus<int64_t>(),
                                   c->stream));
    PM_HIP(hipMemsetAsync(b.fill, 0, ((size_t)ntiles + 1) * sizeof(unsigned), c->stream));
    PM_HIP(hipMemsetAsync(c->d_scratch + 8, 0, 2 * sizeof(unsigned), c->stream));
    PM_HIP(hipMemsetAsync(c->d_scratch + 8 + PM_SAMPLE_WORD, 0, 4 * sizeof(unsigned), c->stream));
    Rebucket rb;
    rb.newstart = table, rb.fill = b.fill, rb.wg_hist = b.wg_hist, rb.dcode = c->stay_codes;
    rb.pos_out = d_pos_out, rb.vel_out = d_vel_out, rb.acc_out = d_acc_out, rb.id = d_id, rb.id_out = d_id_out;
    rb.sink = reinterpret_cast<float *>(c->d_scratch + PM_SCRATCH_SINK);
    float *vel = const_cast<float *>(d_vel);  // read only by the re-bucketing instantiation
    switch (scheme) {
        case PM_NGP: rc = rebucket_order<PM_NGP>(c, order, d_phi, geo, d_pos, vel, np, half_dt, work, lists, nblk, rb); break;
        case PM_CIC: rc = rebucket_order<PM_CIC>(c, order, d_phi, geo, d_pos, vel, np, half_dt, work, lists, nblk, rb); break;
        case PM_TSC: rc = rebucket_order<PM_TSC>(c, order, d_phi, geo, d_pos, vel, np, half_dt, work, lists, nblk, rb); break;
        default: pm_set_error("mass scheme %d, should be NGP(1), CIC(2) or TSC(3)", scheme); rc = PM_EINVAL;
    }
    if (rc) return rc;
    tile_lists_drop(c);  // they describe the old layout
    c->rb_cur ^= 1;
    c->sorted_pos = d_pos_out, c->sorted_np = np, c->sorted_ntiles = ntiles, c->sorted_table = table;
    return PM_OK;
}

extern "C" int pm_gradient_interp_kick_rebucket(pm_ctx *c, int order, int scheme, const float *d_phi, const float *d_pos,
                                                const float *d_vel, int64_t np, float half_dt, float *d_pos_out,
                                                float *d_vel_out, float *d_acc_out, const unsigned *d_id,
                                                unsigned *d_id_out) {
    PM_REQUIRE(c, "NULL argument");
    PM_REQUIRE(c->slab_nx == c->n && c->tile_nx == c->n, "pm_gradient_interp_kick_rebucket is the single-GPU form");
    return gradient_interp_kick_rebucket(c, order, scheme, d_phi, pm_ctx_geom(c), d_pos, d_vel, np, half_dt, d_pos_out, d_vel_out,
                                         d_acc_out, d_id, d_id_out);
}

// The same from a potential slab with ghost planes (the geometry arguments of pm_gradient_interp_kick) and for the
// tiles of the window set by pm_set_tile_window: the form of a slab-decomposed step (order 2, 3, 5 or 7).
extern "C" int pm_gradient_interp_kick_rebucket_slab(pm_ctx *c, int order, int scheme, const float *d_phi, int nx_phi,
                                                     int x0_phi, const float *d_pos, const float *d_vel, int64_t np,
                                                     float half_dt, float *d_pos_out, float *d_vel_out, float *d_acc_out,
                                                     const unsigned *d_id, unsigned *d_id_out) {
    PM_REQUIRE(c && nx_phi >= 1 && nx_phi <= c->n, "bad potential slab");
    PM_REQUIRE(order != 0, "the slab form differentiates the potential (order 2, 3, 5 or 7)");
    pm_geom geo = pm_ctx_geom(c);  // tiles are filed on the full mesh; phi is the (slab + ghosts) array given
    geo.nx = nx_phi, geo.x0 = ((x0_phi % c->n) + c->n) % c->n;
    return gradient_interp_kick_rebucket(c, order, scheme, d_phi, geo, d_pos, d_vel, np, half_dt, d_pos_out, d_vel_out,
                                         d_acc_out, d_id, d_id_out);
}

// ---- "the kick re-buckets" (struct Codes / struct KickReb): host side -------------------------------------------

struct KrBufs {
    unsigned *tile_stay, *inc_count, *segcap, *segstart, *wg_stay, *overflow, *entries;
    int64_t *start, *xstart;  // the ranges the gather decided on: the very table the re-bucketing kick must walk
    unsigned char *flag;
    void *scan_tmp;
    size_t scan_tmp_bytes;
};

static int kr_bufs(pm_ctx *c, unsigned ntiles, int64_t np, KrBufs *b) {
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t maxblocks = (size_t)ntiles + (size_t)(np / PM_RANGE_CHUNK) + 2;
    const size_t t32 = up(((size_t)ntiles + 1) * sizeof(unsigned)), w32 = up(maxblocks * sizeof(unsigned));
    const size_t nent = (size_t)256 * ntiles + (size_t)(np / 4) + (size_t)(np / 2) + ntiles;  // sum of the list capacities (kr_segcap_kernel)
    size_t tmp = 0;
    unsigned *nullp = nullptr;
    PM_HIP(rocprim::exclusive_scan(nullptr, tmp, nullp, nullp, 0u, (size_t)ntiles + 1, rocprim::plus<unsigned>(), c->stream));
    const size_t t64 = up(((size_t)ntiles + 1) * sizeof(int64_t));
    const size_t need = 4 * t32 + w32 + 256 + up(tmp) + up((size_t)np) + up(nent * sizeof(unsigned)) + 2 * t64;
    if (c->kr_buf_bytes < need) {
        if (c->kr_buf) {
            PM_HIP(hipStreamSynchronize(c->stream));
            PM_HIP(hipFree(c->kr_buf));
        }
        c->kr_buf = nullptr, c->kr_buf_bytes = 0, c->kr_valid = 0;
        PM_HIP(hipMalloc(&c->kr_buf, need + need / 16));
        c->kr_buf_bytes = need + need / 16;
    }
    char *p = (char *)c->kr_buf;
    b->tile_stay = (unsigned *)p, b->inc_count = (unsigned *)(p + t32), b->segcap = (unsigned *)(p + 2 * t32);
    b->segstart = (unsigned *)(p + 3 * t32);
    b->wg_stay = (unsigned *)(p + 4 * t32), b->overflow = (unsigned *)(p + 4 * t32 + w32);
    b->scan_tmp = p + 4 * t32 + w32 + 256, b->scan_tmp_bytes = tmp;
    b->flag = (unsigned char *)(p + 4 * t32 + w32 + 256 + up(tmp));
    b->entries = (unsigned *)(p + 4 * t32 + w32 + 256 + up(tmp) + up((size_t)np));
    b->start = (int64_t *)(p + 4 * t32 + w32 + 256 + up(tmp) + up((size_t)np) + up(nent * sizeof(unsigned)));
    b->xstart = (int64_t *)((char *)b->start + t64);
    return PM_OK;
}

// capacity of a tile's incoming list: 256, a quarter of what the tile holds now, and what its six face neighbours can
// send it in one drift of at most a cell if they are evenly filled -- an eighth of a neighbour across an x face (the
// tile is 8 cells thick there), a sixteenth across a y or z face.  (Round 3's first version knew nothing of the
// neighbours: forty steps into the 1024^3 run a near-empty tile beside a cluster overflowed, every step from then on,
// and the stream aged: profiles/r03_long_run_s200w5.json, first version.)  A list that overflows all the same -- a
// cluster sitting on a face -- raises Codes::overflow, the next kick is taken in place and the caller re-buckets with
// its gather for a while (integration.leapfrog).
__global__ __launch_bounds__(BLK) void kr_segcap_kernel(const int64_t *__restrict__ start, unsigned ntiles, int n,
                                                        unsigned *__restrict__ cap) {
    const unsigned t = blockIdx.x * BLK + threadIdx.x;
    if (t > ntiles) return;
    if (t == ntiles) {
        cap[t] = 0u;
        return;
    }
    const TileId id = tile_of_key(t);
    const int Ti = n >> PM_LI, Tj = n >> PM_LJ, Tk = n >> PM_LK;
    auto len = [&](int ti, int tj, int tk) -> unsigned {
        const unsigned u = tile_key((ti + Ti) % Ti, (tj + Tj) % Tj, (tk + Tk) % Tk);
        return (unsigned)(start[u + 1] - start[u]);
    };
    const unsigned own = (unsigned)(start[t + 1] - start[t]);
    const unsigned fx = len(id.ti - 1, id.tj, id.tk) + len(id.ti + 1, id.tj, id.tk);
    const unsigned fyz = len(id.ti, id.tj - 1, id.tk) + len(id.ti, id.tj + 1, id.tk) + len(id.ti, id.tj, id.tk - 1) +
                         len(id.ti, id.tj, id.tk + 1);
    cap[t] = 256u + (own >> 2) + (fx >> 3) + (fyz >> 4);
}

// a tile's population in the new stream: its stayers + its incoming list
__global__ __launch_bounds__(BLK) void kr_counts_kernel(const unsigned *__restrict__ tile_stay, const unsigned *__restrict__ inc_count,
                                                        const unsigned *__restrict__ segstart, unsigned ntiles,
                                                        int64_t *__restrict__ cnt) {
    const unsigned t = blockIdx.x * BLK + threadIdx.x;
    if (t > ntiles) return;
    cnt[t] = t < ntiles ? (int64_t)tile_stay[t] + (int64_t)min(inc_count[t], segstart[t + 1] - segstart[t]) : 0;
}

template <int SCHEME, int ORDER>
static int codes_launch(pm_ctx *c, const float *phi, pm_geom geo, const float *pos, float *vel, int64_t np, float half_dt,
                        float *acc, const TileWork &work, const TileLists &lists, unsigned nblk, const Codes &cd) {
    Rebucket rb{};
    rb.sink = reinterpret_cast<float *>(c->d_scratch + PM_SCRATCH_SINK);
    hipLaunchKernelGGL((interp_ranges_kernel<SCHEME, ORDER, true, false, true>), dim3(nblk), dim3(BLK), 0, c->stream, phi, pos,
                       work, lists, geo, acc, vel, -half_dt, c->d_scratch + 8, rb, cd);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

template <int SCHEME>
static int codes_order(pm_ctx *c, int order, const float *phi, pm_geom geo, const float *pos, float *vel, int64_t np,
                       float half_dt, float *acc, const TileWork &work, const TileLists &lists, unsigned nblk, const Codes &cd) {
    switch (order) {
        case 0: return codes_launch<SCHEME, 0>(c, phi, geo, pos, vel, np, half_dt, acc, work, lists, nblk, cd);
        case 2: return codes_launch<SCHEME, 2>(c, phi, geo, pos, vel, np, half_dt, acc, work, lists, nblk, cd);
        case 3: return codes_launch<SCHEME, 3>(c, phi, geo, pos, vel, np, half_dt, acc, work, lists, nblk, cd);
        case 5: return codes_launch<SCHEME, 5>(c, phi, geo, pos, vel, np, half_dt, acc, work, lists, nblk, cd);
        case 7: return codes_launch<SCHEME, 7>(c, phi, geo, pos, vel, np, half_dt, acc, work, lists, nblk, cd);
    }
    pm_set_error("gradient order %d, should be 0 (force grid), 2, 3, 5 or 7", order);
    return PM_EINVAL;
}

// pm_gradient_interp_kick with the kick, IN PLACE (d_vel kicked, d_acc written, in the order of d_pos), which also decides
// where the next fused kick + drift + deposit files every particle (struct Codes; pm_set_predicted_dt's guess of the next
// dt): the first half of "the kick re-buckets".  Needs complete tile lists for exactly d_pos / np.
extern "C" int pm_gradient_interp_kick_codes(pm_ctx *c, int order, int scheme, const float *d_phi, const float *d_pos,
                                             float *d_vel, int64_t np, float half_dt, float *d_acc) {
    PM_REQUIRE(c && d_phi && d_pos && d_vel && d_acc && np > 0 && np < 4294967296LL, "bad argument");
    PM_REQUIRE(c->slab_nx == c->n && c->tile_nx == c->n, "pm_gradient_interp_kick_codes is the single-GPU form");
    PM_REQUIRE(c->n >= 16, "needs n >= 16");
    PM_REQUIRE(order == 0 || ((uintptr_t)d_phi & 15u) == 0, "d_phi must be 16-byte aligned (the tile fill reads float4)");
    PM_REQUIRE(c->lists_pos == d_pos && c->lists_np == np && c->lists_use == 1,
               "no complete tile lists prepared for these positions (the fused kick + drift + deposit, or pm_tile_lists, first)");
    const unsigned ntiles = (unsigned)(((int64_t)c->n * c->n * c->n) >> (PM_LI + PM_LJ + PM_LK));
    KrBufs b;
    int rc = kr_bufs(c, ntiles, np, &b);
    if (rc) return rc;
    TileWork work;
    TileLists lists;
    unsigned nblk = 0;
    rc = tile_work_for(c, d_pos, np, &work, &lists, &nblk);
    if (rc) return rc;
    const unsigned tb = (ntiles + 1 + BLK - 1) / BLK;
    hipLaunchKernelGGL(kr_segcap_kernel, dim3(tb), dim3(BLK), 0, c->stream, work.start, ntiles, c->n, b.segcap);
    PM_LAUNCH_CHECK();
    size_t tmp = b.scan_tmp_bytes;
    PM_HIP(rocprim::exclusive_scan(b.scan_tmp, tmp, b.segcap, b.segstart, 0u, (size_t)ntiles + 1, rocprim::plus<unsigned>(), c->stream));
    PM_HIP(hipMemsetAsync(b.tile_stay, 0, 2 * (((size_t)(ntiles + 1) * sizeof(unsigned) + 255) & ~(size_t)255), c->stream));  // + inc_count
    PM_HIP(hipMemsetAsync(c->d_scratch + 8, 0, 2 * sizeof(unsigned), c->stream));
    PM_HIP(hipMemsetAsync(c->d_scratch + 8 + PM_SAMPLE_WORD, 0, 5 * sizeof(unsigned), c->stream));
    Codes cd;
    cd.flag = b.flag, cd.wg_stay = b.wg_stay, cd.tile_stay = b.tile_stay, cd.inc_count = b.inc_count, cd.segstart = b.segstart;
    // (the overflow word sits with the CFL maxima: pm_last_maxabs, which every step calls anyway, reads it along)
    cd.entries = b.entries, cd.overflow = c->d_scratch + 8 + PM_SAMPLE_WORD + 4, cd.dtp = c->dt_pred;
    const pm_geom geo = pm_ctx_geom(c);
    switch (scheme) {
        case PM_NGP: rc = codes_order<PM_NGP>(c, order, d_phi, geo, d_pos, d_vel, np, half_dt, d_acc, work, lists, nblk, cd); break;
        case PM_CIC: rc = codes_order<PM_CIC>(c, order, d_phi, geo, d_pos, d_vel, np, half_dt, d_acc, work, lists, nblk, cd); break;
        case PM_TSC: rc = codes_order<PM_TSC>(c, order, d_phi, geo, d_pos, d_vel, np, half_dt, d_acc, work, lists, nblk, cd); break;
        default: pm_set_error("mass scheme %d, should be NGP(1), CIC(2) or TSC(3)", scheme); rc = PM_EINVAL;
    }
    if (rc) return rc;
    // (an in-place stream is only roughly tile-sorted: ranges looked up again after the next drift would come out
    // different, and the per-workgroup counts belong to THESE)
    PM_HIP(hipMemcpyAsync(b.start, work.start, ((size_t)ntiles + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, c->stream));
    PM_HIP(hipMemcpyAsync(b.xstart, work.xstart, ((size_t)ntiles + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, c->stream));
    c->kr_valid = 1, c->kr_checked = 0, c->kr_pos = d_pos, c->kr_np = np, c->kr_blocks = nblk;
    return PM_OK;
}

// *h_ready = 1 if pm_kick_drift_wrap_deposit_rebucket can take the next step of d_pos / np: the last gather left its
// decisions for exactly these arrays and no incoming list overflowed (one small read-back, once).
extern "C" int pm_rebucket_kick_ready(pm_ctx *c, const float *d_pos, int64_t np, int *h_ready) {
    PM_REQUIRE(c && h_ready, "NULL argument");
    *h_ready = 0;
    if (!c->kr_valid || c->kr_pos != d_pos || c->kr_np != np || !c->kr_buf) return PM_OK;
    if (!c->kr_checked) {
        const unsigned ntiles = (unsigned)(((int64_t)c->n * c->n * c->n) >> (PM_LI + PM_LJ + PM_LK));
        KrBufs b;
        int rc = kr_bufs(c, ntiles, np, &b);
        if (rc) return rc;
        unsigned over = 0;
        rc = pm_read_back(c, &over, c->d_scratch + 8 + PM_SAMPLE_WORD + 4, sizeof(over));
        if (rc) return rc;
        c->kr_checked = 1;
        if (over) c->kr_valid = 0;
    }
    *h_ready = c->kr_valid;
    return PM_OK;
}

template <int SCHEME>
static int kdw_rebucket_launch(pm_ctx *c, const float *pos, const float *vel, const float *acc, float half_dt, float dt,
                               int64_t np, float *rho, float *khalo, float *pos_out, float *vel_out, const unsigned *id,
                               unsigned *id_out) {
    const unsigned ntiles = (unsigned)(((int64_t)c->n * c->n * c->n) >> (PM_LI + PM_LJ + PM_LK));
    KrBufs kb;
    int rc = kr_bufs(c, ntiles, np, &kb);
    if (rc) return rc;
    RbBufs rb;
    rc = rebucket_bufs(c, ntiles, np, &rb);  // (before the ranges are looked up: it may move the table on file)
    if (rc) return rc;
    PM_REQUIRE(c->kr_valid, "the re-sort's buffers moved under the gather's decisions");
    TileWork work;
    tile_lists_drop(c);
    tile_window(c, &work);
    work.start = kb.start, work.xstart = kb.xstart;  // the old stream's ranges: the very table the gather decided on
    const unsigned blocks = c->kr_blocks;
    ListBufs lb;
    rc = list_bufs(c, np, ntiles, &lb);
    if (rc) return rc;
    // layout of the new stream
    int64_t *table = rb.table[c->rb_cur ^ 1];
    const unsigned tb = (ntiles + 1 + BLK - 1) / BLK;
    hipLaunchKernelGGL(kr_counts_kernel, dim3(tb), dim3(BLK), 0, c->stream, kb.tile_stay, kb.inc_count, kb.segstart, ntiles, rb.cnt);
    PM_LAUNCH_CHECK();
    size_t tmp = rb.scan_tmp_bytes;
    PM_HIP(rocprim::exclusive_scan(rb.scan_tmp, tmp, rb.cnt, table, (int64_t)0, (size_t)ntiles + 1, rocprim::plus<int64_t>(), c->stream));
    PM_HIP(hipMemsetAsync(rb.fill, 0, ((size_t)ntiles + 1) * sizeof(unsigned), c->stream));
    KickReb kr;
    kr.flag = kb.flag, kr.wg_stay = kb.wg_stay, kr.tile_stay = kb.tile_stay, kr.inc_count = kb.inc_count, kr.segstart = kb.segstart;
    kr.entries = kb.entries, kr.newstart = table, kr.fill = rb.fill, kr.pos_out = pos_out, kr.vel_out = vel_out, kr.id = id, kr.id_out = id_out;
    const size_t cells = (size_t)c->n * c->n * c->n;
    const HaloBuf halo = halo_parts(khalo, c->n);
    const size_t halo_bytes = (size_t)halo_floats(c->n, halo.nti) * sizeof(float);
    if (!khalo) PM_HIP(hipMemsetAsync(rho, 0, cells * sizeof(float), c->stream));
    PM_HIP(hipMemsetAsync(lb.counts, 0, PM_NSUB * PM_SUB_STRIDE * sizeof(unsigned), c->stream));
    const Leavers lv{0.f, 0.f, nullptr, 0, nullptr};
    const DestCount dcount{nullptr, nullptr, nullptr, 0.f, 0.f};
    float *posw = const_cast<float *>(pos), *velw = const_cast<float *>(vel);  // (read only by this instantiation)
    hipLaunchKernelGGL((kdw_deposit_kernel<SCHEME, true>), dim3(khalo ? work.nlaunch : blocks), dim3(BLK), 0, c->stream, posw, velw, acc,
                       -half_dt, dt, work, pm_ctx_geom(c), rho, lb.k_in, lb.i_in, (unsigned)lb.subcap, lb.counts, lv, halo, dcount, kr);
    if (khalo && blocks > work.nlaunch) {  // owners first (plain stores), then the surplus workgroups (atomics)
        TileWork extra = work;
        extra.boff = work.nlaunch;
        hipLaunchKernelGGL((kdw_deposit_kernel<SCHEME, true>), dim3(blocks - work.nlaunch), dim3(BLK), 0, c->stream, posw, velw,
                           acc, -half_dt, dt, extra, pm_ctx_geom(c), rho, lb.k_in, lb.i_in, (unsigned)lb.subcap, lb.counts, lv, halo,
                           dcount, kr);
    }
    PM_LAUNCH_CHECK();
    // the new stream: its offsets are on file, its ranges and straggler lists are what the gather of this step needs
    c->rb_cur ^= 1;
    c->sorted_pos = pos_out, c->sorted_np = np, c->sorted_ntiles = ntiles, c->sorted_table = table;
    c->kr_valid = 0;
    TileWork nwork;
    unsigned nblocks = 0;
    rc = tile_ranges(c, pos_out, np, &nwork, &nblocks);
    if (rc) return rc;
    const unsigned *idx = nullptr;
    unsigned listed = 0;
    int complete = 0;
    rc = list_group(c, lb, ntiles, &idx, &listed, &complete);
    if (rc) return rc;
    c->lists_ptr[0] = nwork.start, c->lists_ptr[1] = nwork.xstart, c->lists_ptr[2] = idx, c->lists_ptr[3] = lb.lstart;
    c->lists_blocks = nblocks;
    if (!complete) {  // (stragglers bunched up beyond the lists: deposit the new stream again from scratch)
        c->lists_pos = nullptr;
        PM_HIP(hipMemsetAsync(rho, 0, cells * sizeof(float), c->stream));
        if (khalo) PM_HIP(hipMemsetAsync(khalo, 0, halo_bytes, c->stream));
        hipLaunchKernelGGL((deposit_tiled_kernel<SCHEME>), dim3((unsigned)((np + PM_CHUNK - 1) / PM_CHUNK)), dim3(BLK), 0, c->stream,
                           pos_out, np, pm_ctx_geom(c), rho);
        PM_LAUNCH_CHECK();
        return PM_OK;
    }
    TileLists lists;
    lists.idx = idx, lists.lstart = lb.lstart, lists.use = 1;
    if (listed > 0 && (int64_t)listed * 400 < np) {
        hipLaunchKernelGGL((deposit_list_flat_kernel<SCHEME>), dim3(pm_stream_blocks((int64_t)listed * 64, BLK)), dim3(BLK), 0, c->stream,
                           pos_out, idx, listed, pm_ctx_geom(c), rho, halo);
        PM_LAUNCH_CHECK();
    } else if (listed > 0) {
        hipLaunchKernelGGL((deposit_list_kernel<SCHEME>), dim3(ntiles), dim3(BLK), 0, c->stream, pos_out, lists, pm_ctx_geom(c), rho, halo);
        PM_LAUNCH_CHECK();
    }
    c->lists_pos = pos_out, c->lists_np = np, c->lists_use = 1, c->lists_sticky = 1, c->stay_valid = 0, c->stay_codes = nullptr;
    return PM_OK;
}

// The second half of "the kick re-buckets": pm_kick_drift_wrap_deposit[_khalo] (d_khalo NULL: plain flush) writing position
// and velocity OUT OF PLACE into d_pos_out / d_vel_out, tile after tile, as the last pm_gradient_interp_kick_codes on
// exactly d_pos / np decided (pm_rebucket_kick_ready).  d_pos / d_vel / d_acc are only read.  Leaves the new stream's
// tile offsets on file and its ranges + straggler lists prepared, like the in-place form.
extern "C" int pm_kick_drift_wrap_deposit_rebucket(pm_ctx *c, int scheme, const float *d_pos, const float *d_vel,
                                                   const float *d_acc, float half_dt, float dt, int64_t np, float *d_rho,
                                                   float *d_khalo, float *d_pos_out, float *d_vel_out, const unsigned *d_id,
                                                   unsigned *d_id_out) {
    PM_REQUIRE(c && d_pos && d_vel && d_acc && d_rho && d_pos_out && d_vel_out && np > 0 && np < 4294967296LL, "bad argument");
    PM_REQUIRE(c->slab_nx == c->n && c->tile_nx == c->n && c->n >= 16, "single-GPU form, n >= 16");
    PM_REQUIRE(d_pos_out != d_pos && d_vel_out != d_vel, "the new stream is written out of place");
    PM_REQUIRE((!d_id) == (!d_id_out), "d_id / d_id_out must come as a pair");
    PM_REQUIRE(c->kr_valid && c->kr_pos == d_pos && c->kr_np == np,
               "no decisions on file for these arrays (pm_gradient_interp_kick_codes on the same d_pos first; pm_rebucket_kick_ready)");
    switch (scheme) {
        case PM_NGP: return kdw_rebucket_launch<PM_NGP>(c, d_pos, d_vel, d_acc, half_dt, dt, np, d_rho, d_khalo, d_pos_out, d_vel_out, d_id, d_id_out);
        case PM_CIC: return kdw_rebucket_launch<PM_CIC>(c, d_pos, d_vel, d_acc, half_dt, dt, np, d_rho, d_khalo, d_pos_out, d_vel_out, d_id, d_id_out);
        case PM_TSC: return kdw_rebucket_launch<PM_TSC>(c, d_pos, d_vel, d_acc, half_dt, dt, np, d_rho, d_khalo, d_pos_out, d_vel_out, d_id, d_id_out);
    }
    pm_set_error("mass scheme %d, should be NGP(1), CIC(2) or TSC(3)", scheme);
    return PM_EINVAL;
}

// A slab's migration changes the particle count of a stream that was written tile by tile: leavers' slots are refilled
// (by arrivals or from the tail) and the surplus arrivals appended.  The tile offsets on file stay a partition of the
// stream if they are clamped to the new count and the last one set to it -- the few particles that changed slots are
// strangers in their ranges, which the lists take care of -- where finding ranges again from the positions (a running
// maximum of the tile keys along the stream) lets ONE arrival with a high key in an early slot file a tenth of the
// rank's particles under its tile.
// (the surplus at the end must belong to a tile some workgroup serves: every offset past the last tile of the rank's
// window -- ti_max < 0: no window, the last tile of the mesh -- becomes the new count)
__global__ __launch_bounds__(BLK) void table_resize_kernel(int64_t *__restrict__ table, unsigned ntiles, int64_t np_new,
                                                           int ti_max, int tj_max, int tk_max) {
    const unsigned t = blockIdx.x * BLK + threadIdx.x;
    if (t > ntiles) return;
    const unsigned kmax = ti_max < 0 ? ntiles - 1u : tile_key(ti_max, tj_max, tk_max);
    table[t] = t > kmax ? np_new : min(table[t], np_new);
}

extern "C" int pm_sorted_stream_resized(pm_ctx *c, const float *d_pos, int64_t np_new) {
    PM_REQUIRE(c && d_pos && np_new > 0, "bad argument");
    if (c->sorted_pos != d_pos || !c->sorted_table || !c->rb_buf) return PM_OK;  // nothing on file for this stream
    if (c->sorted_np == np_new) return PM_OK;
    TileWork w;
    tile_window(c, &w);  // (the window in force now: the one the caller's tile-range calls will run with)
    const bool windowed = w.nlaunch != w.ntiles;
    hipLaunchKernelGGL(table_resize_kernel, dim3((c->sorted_ntiles + 1 + BLK - 1) / BLK), dim3(BLK), 0, c->stream, c->sorted_table,
                       c->sorted_ntiles, np_new, windowed ? w.ti0 + (1 << w.lnti) - 1 : -1, (c->n >> PM_LJ) - 1,
                       (c->n >> PM_LK) - 1);
    PM_LAUNCH_CHECK();
    c->sorted_np = np_new;
    tile_lists_drop(c);
    return PM_OK;
}

// The caller's guess of the NEXT step's dt, used by pm_kick_drift_wrap_deposit[_khalo] to count and by
// pm_gradient_interp_kick_rebucket to choose the tile every particle is filed under (dest_of).  0 (default): the
// tile it is in.  Any value gives correct results; a good guess keeps the straggler lists short.
extern "C" int pm_set_predicted_dt(pm_ctx *c, float dt_next) {
    PM_REQUIRE(c && dt_next >= 0.f && dt_next == dt_next, "bad argument");
    c->dt_pred = dt_next;
    return PM_OK;
}

// How much the accelerations are expected to grow over the next step (|a_new| ~ scale |a_old|, 1 = no guess): the
// re-sort's prediction of the next drift uses scale * a_old where the new acceleration is not known yet (dest_of).
extern "C" int pm_set_predicted_acc_scale(pm_ctx *c, float scale) {
    PM_REQUIRE(c && scale > 0.f && scale < 1.0e6f, "bad argument");
    c->acc_scale_pred = scale;
    return PM_OK;
}

extern "C" int pm_last_maxabs(pm_ctx *c, float *h_max_acc, float *h_max_vel) {
    PM_REQUIRE(c && h_max_acc && h_max_vel, "NULL argument");
    unsigned bits[PM_SAMPLE_WORD + 5];
    memset(bits, 0, sizeof(bits));
    if (int rb_rc = pm_read_back(c, bits, c->d_scratch + 8, sizeof(bits))) return rb_rc;
    memcpy(h_max_acc, &bits[0], sizeof(float));
    memcpy(h_max_vel, &bits[1], sizeof(float));
    double sum;
    memcpy(&sum, &bits[PM_SAMPLE_WORD], sizeof(double));
    const unsigned cnt = bits[PM_SAMPLE_WORD + 2];
    c->last_acc_meanabs = cnt >= 1024u ? sum / (double)cnt : 0.0;  // (came with the same copy; a handful of particles is no sample)
    if (c->kr_valid && !c->kr_checked) {  // an incoming list of pm_gradient_interp_kick_codes overflowed: its decisions are void
        c->kr_checked = 1;
        if (bits[PM_SAMPLE_WORD + 4] != 0u) c->kr_valid = 0;
    }
    if (bits[PM_SAMPLE_WORD + 3] != 0u) {
        // the re-bucketing gather found no slot for some particles: their records were NOT written (positions or
        // destination codes changed between the count and the gather).  The output arrays are incomplete.
        pm_set_error("the re-bucketing gather dropped particles in %u waves: positions or destination codes changed behind "
                     "the prepared counts (call pm_tile_lists + pm_rebucket_prepare again, or reorder the particles)",
                     bits[PM_SAMPLE_WORD + 3]);
        return PM_EINVAL;
    }
    return PM_OK;
}

// Mean |a_x| over the sample of particles the last fused gather summed (every 64th tile of a large launch, every tile of a small one; 0 below 1024 particles), as read back by the
// last pm_last_maxabs; 0 if that gather took no sample.  Host value, no synchronisation.
extern "C" int pm_last_acc_meanabs(pm_ctx *c, double *h_meanabs) {
    PM_REQUIRE(c && h_meanabs, "NULL argument");
    *h_meanabs = c->last_acc_meanabs;
    return PM_OK;
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
