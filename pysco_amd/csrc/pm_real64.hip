// The fp64 device mode (pm_ctx_create_real(..., real_bytes = 8)): the Newtonian FFT step of PySCo's hot path with
// every array and every scalar in double -- kick + drift + wrap (integration.py:192-264, utils.py:220-298, 1120-1149),
// NGP / CIC / TSC deposit (mesh.py:2240-2595), r2c -> Green's function -> c2r (solver.py:453-522, fourier.py:460-595),
// gradient of order 2 / 3 / 5 / 7 (mesh.py:639-850), inverse interpolation (mesh.py:2600-3088) + second kick + CFL
// maxima, max_abs (utils.py:410-430) and the binned P(k) (fourier.py:22-100).
//
// The reference has no fp64 mode (it is float32 end to end: mesh.py:2467, fourier.py:21); this one exists to state a
// tolerance between the build's own fp64 CPU and fp64 device paths (BASELINE.md section 6:
// <= 1e-10), i.e. to separate rounding from algorithmic differences.  Same formulas and operation order as the fp32
// kernels; the data layout is the fp32 mode's with 8-byte elements ([Np,3] particles, [n,n,n] grids, [n,n,n,3] force).
//
// Design for MI355X: the deposit accumulates a workgroup's chunk of the (Morton-sorted) stream in an LDS tile of
// doubles with ds_add_f64 -- measured at 6.4 lane-operations per clock and CU (tools/micro/lds_atomics.hip), against
// ~1.3 TB/s of added bytes for memory-side atomics -- and flushes the tile with one global_atomic_add_f64 per touched
// cell; a chunk that spans several tiles re-anchors the tile, particles outside the first few tiles go straight to
// global atomics: the result never depends on the order of the stream, only the speed does.  The gather differentiates
// nothing on the fly: the force grid is written once ([n,n,n,3] doubles: 3.2 GB at 512^3) and a workgroup stages the
// 10 x 18 x 18 force tile of its chunk in LDS (3 x 25.9 KB).  rocFFT double-precision plans do the transforms.
#include "pm_common.h"

#include <rocprim/rocprim.hpp>

#define BLK 256

namespace {

constexpr int TI = 8, TJ = 16, TK = 16, LI = 3, LJ = 4, LK = 4;  // tile of 8 x 16 x 16 cells: 2048 Morton codes
constexpr int HI = TI + 2, HJ = TJ + 2, HK = TK + 2;
constexpr int TILE = HI * HJ * HK;   // 3240 cells with the one-cell halo TSC / CIC reach
constexpr int PPT = 8, CHUNK = BLK * PPT;

__device__ __forceinline__ int cell_of(double xp, int n) {
    const int i = (int)xp;  // np.int16(x_part): truncation (mesh.py:2506)
    return min(max(i, 0), n - 1);
}

__device__ __forceinline__ void tsc_weights(double xp, int i, double w[3]) {  // mesh.py:2510-2520
    const double d = xp - 0.5 - (double)i;
    w[1] = 0.75 - d * d;
    const double a = 0.5 - d, b = 0.5 + d;
    w[0] = 0.5 * (a * a);
    w[2] = 0.5 * (b * b);
}

__device__ __forceinline__ void cic_weights(double xp, int i, double &w0, double &w1, int &sgn) {  // mesh.py:2312-2330
    const double d = xp - 0.5 - (double)i;
    sgn = (d > 0.0) - (d < 0.0);
    w1 = fabs(d);
    w0 = 1.0 - w1;
}

__device__ __forceinline__ int64_t cell_index(int n, int log2n, int i, int j, int k) {
    const int m = n - 1;
    return ((((int64_t)(i & m) << log2n) + (j & m)) << log2n) + (k & m);
}

__device__ __forceinline__ double wrap01(double y) {  // utils.py:1120-1149 (eps is a Python float there too)
    const double eps = -(1.0 / 33554432.0) * (1.0 + 1e-6);
    if (y < 0.0) return y > eps ? 0.0 : y + 1.0;
    return y >= 1.0 ? y - 1.0 : y;
}

// ---- kick + drift + wrap ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLK) void kdw64_kernel(double *__restrict__ pos, double *__restrict__ vel,
                                                    const double *__restrict__ acc, double mhalf_dt, double dt, int64_t n3) {
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < n3; t += (int64_t)gridDim.x * BLK) {
        const double v = vel[t] + mhalf_dt * acc[t];  // y += a * x (utils.py:297): product and sum rounded separately
        vel[t] = v;
        pos[t] = wrap01(pos[t] + dt * v);
    }
}

__global__ __launch_bounds__(BLK) void axpb64_kernel(double *__restrict__ x, double a, double b, int64_t n) {
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < n; t += (int64_t)gridDim.x * BLK) x[t] = a * x[t] + b;
}

// integer order of the bit patterns == order of the magnitudes, NaN on top (np.max(np.abs(x)) propagates NaN)
__global__ __launch_bounds__(BLK) void maxabs64_kernel(const double *__restrict__ x, int64_t n, unsigned long long *out) {
    unsigned long long m = 0;
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < n; t += (int64_t)gridDim.x * BLK)
        m = max(m, (unsigned long long)__double_as_longlong(x[t]) & 0x7fffffffffffffffull);
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned long long)__shfl_down((long long)m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, m);
}

// ---- deposit ------------------------------------------------------------------------------------------------------
struct Tile {
    int ti, tj, tk;
};

template <int SCHEME>
__device__ __forceinline__ void deposit_global(double xp, double yp, double zp, int i, int j, int k, int n, int log2n,
                                               double *rho) {
    auto at = [&](int a, int b, int c) { return rho + cell_index(n, log2n, a, b, c); };
    if (SCHEME == PM_NGP) {
        atomicAdd(at(i, j, k), 1.0);
    } else if (SCHEME == PM_CIC) {
        double wx[2], wy[2], wz[2];
        int sx, sy, sz;
        cic_weights(xp, i, wx[0], wx[1], sx), cic_weights(yp, j, wy[0], wy[1], sy), cic_weights(zp, k, wz[0], wz[1], sz);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) atomicAdd(at(i + a * sx, j + b * sy, k + c * sz), wx[a] * wy[b] * wz[c]);
    } else {
        double wx[3], wy[3], wz[3];
        tsc_weights(xp, i, wx), tsc_weights(yp, j, wy), tsc_weights(zp, k, wz);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const double wxy = wx[a] * wy[b];
#pragma unroll
                for (int c = 0; c < 3; ++c) atomicAdd(at(i + a - 1, j + b - 1, k + c - 1), wxy * wz[c]);
            }
    }
}

template <int SCHEME>
__device__ __forceinline__ void deposit_lds(double xp, double yp, double zp, int i, int j, int k, const Tile &t, double *tile) {
    const int li = i - t.ti * TI + 1, lj = j - t.tj * TJ + 1, lk = k - t.tk * TK + 1;
    auto at = [&](int a, int b, int c) { return tile + ((li + a) * HJ + (lj + b)) * HK + (lk + c); };
    if (SCHEME == PM_NGP) {
        atomicAdd(at(0, 0, 0), 1.0);
    } else if (SCHEME == PM_CIC) {
        double wx[2], wy[2], wz[2];
        int sx, sy, sz;
        cic_weights(xp, i, wx[0], wx[1], sx), cic_weights(yp, j, wy[0], wy[1], sy), cic_weights(zp, k, wz[0], wz[1], sz);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) atomicAdd(at(a * sx, b * sy, c * sz), wx[a] * wy[b] * wz[c]);
    } else {
        double wx[3], wy[3], wz[3];
        tsc_weights(xp, i, wx), tsc_weights(yp, j, wy), tsc_weights(zp, k, wz);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const double wxy = wx[a] * wy[b];
#pragma unroll
                for (int c = 0; c < 3; ++c) atomicAdd(at(a - 1, b - 1, c - 1), wxy * wz[c]);
            }
    }
}

// A workgroup serves its chunk of the stream tile by tile.  Every thread keeps its particles' tile keys in registers.
// First the chunk is classified: a small hash table in LDS (in the tile's own storage, before it is used) counts the
// chunk's particles per tile, and a particle whose tile holds fewer than DENSE_MIN of them is SPARSE: it is served one
// by one through L2 (global atomics / global loads) in one sweep at the end, every lane busy.  The dense tiles -- at
// most CHUNK / DENSE_MIN of them -- get a pass each through the LDS tile.  A tile pass moves a tile's worth of data
// whatever it serves (25.9 KB zeroed and flushed, or 77.8 KB of force staged) and runs the service code for every wave
// with a member: a chunk in a void (forty tiles with fifty particles each) or of an aged stream must not pay that
// forty times for one or two busy lanes per wave.  Measured at 512^3, twenty steps into the run (clustered): gather
// 19.0 -> ms, deposit 8.2 -> ms with a tile pass for every tile of the chunk.
constexpr int DENSE_MIN = 32, PASS_MAX = 64, HASH_SLOTS = 1024, HASH_PROBES = 8;
static_assert(PASS_MAX * DENSE_MIN >= 2048 && HASH_SLOTS * 2 * sizeof(int) <= TILE * sizeof(double), "chunk classification");

__device__ __forceinline__ int tile_key_of(int i, int j, int k) { return ((i >> LI) << 20) | ((j >> LJ) << 10) | (k >> LK); }
__device__ __forceinline__ Tile tile_of_key(int key) { return Tile{key >> 20, (key >> 10) & 1023, key & 1023}; }
__device__ __forceinline__ unsigned hash_of_key(int key) { return ((unsigned)key * 2654435761u) >> 22; }  // 10 bits

// Classifies the thread's NP particles (key[r] >= 0 where todo has bit r): returns the mask of the SPARSE ones.  `table`
// is HASH_SLOTS x {key, count} ints of LDS scratch; two barriers inside, one more needed before the storage is reused.
template <int NT, int NP>
__device__ __forceinline__ unsigned classify_sparse(unsigned todo, const int (&key)[NP], int *table) {
    const int tid = threadIdx.x;
    for (int c = tid; c < HASH_SLOTS; c += NT) table[2 * c] = -1, table[2 * c + 1] = 0;
    __syncthreads();
    // (a wave of the sorted stream holds one or two different keys: the lanes that share the first lane's key count
    // as one insertion -- 2048 single insertions of the same key were 2048 same-address LDS atomics in a row)
    int slot[NP];
#pragma unroll
    for (int r = 0; r < NP; ++r) {
        slot[r] = -1;
        bool pending = todo >> r & 1u;
        for (int round = 0; round < 3 && __ballot(pending); ++round) {
            const unsigned long long act = __ballot(pending);
            const int lead = __ffsll((long long)act) - 1;
            const int k0 = __shfl(key[r], lead, 64);
            const unsigned long long same = __ballot(pending && key[r] == k0);
            int found = -1;
            if ((int)(tid & 63) == lead) {
                unsigned h = hash_of_key(k0);
                for (int probe = 0; probe < HASH_PROBES; ++probe, h = (h + 1) & (HASH_SLOTS - 1)) {
                    const int old = atomicCAS(&table[2 * h], -1, k0);
                    if (old == -1 || old == k0) {
                        atomicAdd(&table[2 * h + 1], __popcll(same));
                        found = (int)h;
                        break;
                    }
                }
            }
            found = __shfl(found, lead, 64);
            if (pending && key[r] == k0) slot[r] = found, pending = false;
        }
        if (pending) {  // a wave with many different keys (an unsorted stream): one by one
            unsigned h = hash_of_key(key[r]);
            for (int probe = 0; probe < HASH_PROBES; ++probe, h = (h + 1) & (HASH_SLOTS - 1)) {
                const int old = atomicCAS(&table[2 * h], -1, key[r]);
                if (old == -1 || old == key[r]) {
                    atomicAdd(&table[2 * h + 1], 1);
                    slot[r] = (int)h;
                    break;
                }
            }
        }
    }
    __syncthreads();
    unsigned sparse = 0;
#pragma unroll
    for (int r = 0; r < NP; ++r)
        if ((todo >> r & 1u) && (slot[r] < 0 || table[2 * slot[r] + 1] < DENSE_MIN)) sparse |= 1u << r;  // (no slot: a crowded table)
    return sparse;
}

// One pass's choice among the particles of `todo` (the dense ones): the tile key of the first of the chunk (uniform) and
// this thread's slots in that tile.  Returns false when nothing is left.  NT threads, NP slots each; three barriers.
template <int NT, int NP>
__device__ __forceinline__ bool next_tile(unsigned todo, const int (&key)[NP], int *s_pick, int &akey, unsigned &mine) {
    const int tid = threadIdx.x;
    if (tid == 0) s_pick[0] = 0x7fffffff;
    __syncthreads();
    if (todo) atomicMin(&s_pick[0], (__ffs((int)todo) - 1) * NT + tid);
    __syncthreads();
    const int first = s_pick[0];
    if (first == 0x7fffffff) return false;
    if (first % NT == tid) {
        int kk = 0;
#pragma unroll
        for (int r = 0; r < NP; ++r) kk = (first / NT == r) ? key[r] : kk;
        s_pick[2] = kk;
    }
    __syncthreads();
    akey = s_pick[2];
    mine = 0;
#pragma unroll
    for (int r = 0; r < NP; ++r)
        if ((todo >> r & 1u) && key[r] == akey) mine |= 1u << r;
    return true;
}

// KDW: the first kick, the drift and the wrap of the leapfrog (kdw64_kernel's expressions) applied where the particle is
// first read -- position and velocity updated in place -- so that the deposit does not read the positions a pass of
// their own has just written (pm_real64_kick_drift_wrap_deposit)
template <int SCHEME, bool KDW = false>
__global__ __launch_bounds__(BLK) void deposit64_kernel(double *pos, int64_t np, int n, int log2n, double *__restrict__ rho,
                                                        double *vel = nullptr, const double *acc = nullptr, double mhalf_dt = 0.0,
                                                        double dt = 0.0) {
    __shared__ double tile[TILE];  // 25.9 KB
    __shared__ int s_pick[3];
    const int tid = threadIdx.x;
    const double nf = (double)n;
    const int64_t base = (int64_t)blockIdx.x * CHUNK;
    // (the tile keys stay in registers over the passes; the positions are read again where a particle is served: the
    // service loops stay rolled -- eight unrolled copies of 27 atomics were 512 registers and a 35 KB kernel)
    int key[PPT];
    unsigned todo = 0;
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
        const int64_t p = base + r * BLK + tid;
        key[r] = -1;
        if (p < np) {
            double x[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                x[m] = pos[3 * p + m];
                if (KDW) {
                    const double v = vel[3 * p + m] + mhalf_dt * acc[3 * p + m];  // (utils.py:297: product and sum rounded separately)
                    vel[3 * p + m] = v;
                    x[m] = wrap01(x[m] + dt * v);
                    pos[3 * p + m] = x[m];  // (read again below by this very thread)
                }
            }
            const double xp = x[0] * nf, yp = x[1] * nf, zp = x[2] * nf;
            key[r] = tile_key_of(cell_of(xp, n), cell_of(yp, n), cell_of(zp, n));
            todo |= 1u << r;
        }
    }
    const bool tiled = n >= 16;
    unsigned left = todo;  // served through L2 at the end: the sparse ones (everything when n < 16)
    if (tiled) {
        left = classify_sparse<BLK, PPT>(todo, key, reinterpret_cast<int *>(tile));  // (the table's storage is the tile's)
        todo &= ~left;
        __syncthreads();
        for (int c = tid; c < TILE; c += BLK) tile[c] = 0.0;
    }
    for (int pass = 0; tiled && pass < PASS_MAX; ++pass) {
        int akey;
        unsigned mine;
        if (!next_tile<BLK, PPT>(todo, key, s_pick, akey, mine)) break;  // (its barriers close the zeroing / the last flush)
        const Tile t = tile_of_key(akey);
#pragma unroll 1
        for (int r = 0; r < PPT; ++r)
            if (mine >> r & 1u) {
                const double *q = pos + 3 * (base + r * BLK + tid);
                const double xp = q[0] * nf, yp = q[1] * nf, zp = q[2] * nf;
                deposit_lds<SCHEME>(xp, yp, zp, cell_of(xp, n), cell_of(yp, n), cell_of(zp, n), t, tile);
            }
        todo &= ~mine;
        {
            __syncthreads();
            for (int c = tid; c < TILE; c += BLK) {  // flush: one global atomic per touched cell, rows of 18 along k
                const double v = tile[c];
                if (v != 0.0) {
                    const int lk = c % HK, lij = c / HK, lj = lij % HJ, li = lij / HJ;
                    atomicAdd(rho + cell_index(n, log2n, t.ti * TI - 1 + li, t.tj * TJ - 1 + lj, t.tk * TK - 1 + lk), v);
                    tile[c] = 0.0;
                }
            }
        }
    }
    left |= todo;
#pragma unroll 1
    for (int r = 0; r < PPT; ++r)  // the sparse ones
        if (left >> r & 1u) {
            const double *q = pos + 3 * (base + r * BLK + tid);
            const double xp = q[0] * nf, yp = q[1] * nf, zp = q[2] * nf;
            deposit_global<SCHEME>(xp, yp, zp, cell_of(xp, n), cell_of(yp, n), cell_of(zp, n), n, log2n, rho);
        }
}

// ---- Green's function (fourier.py:460-595) ------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(BLK) void greens64_kernel(double2 *__restrict__ x, int n, int p, const double *__restrict__ k2,
                                                       const double *__restrict__ s, double c0) {
    const int nh = (n >> 1) + 1;
    const int64_t total = (int64_t)n * n * nh;
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLK) {
        const int k = (int)(t % nh);
        const int64_t ij = t / nh;
        const int j = (int)(ij % n), i = (int)(ij / n);
        double g;
        if (KIND == PM_GREEN_7PT) {
            g = c0 / ((s[i] + s[j]) + s[k]);
        } else {
            const double invk2 = c0 / ((k2[i] + k2[j]) + k2[k]);
            if (KIND == PM_GREEN_COMPENSATED) {  // w^(-2p) for the small integer p of the mass scheme: products, one division
                const double w = (s[i] * s[j]) * s[k], w2 = w * w;
                double wp = w2;
                for (int e = 1; e < p; ++e) wp *= w2;
                g = invk2 / wp;  // (pow(w, -2p) to an ulp or two: 1e-16, against the 1e-10 this mode is held to)
            } else {
                g = invk2;
            }
        }
        double2 v = x[t];
        if (t == 0) v = make_double2(0.0, 0.0);  // x[0,0,0] = 0 (fourier.py:491, 544, 595)
        else v.x *= g, v.y *= g;
        x[t] = v;
    }
}

// ---- gradient (mesh.py:639-850): [n,n,n] -> [n,n,n,3] -------------------------------------------------------------
template <int ORDER>
__device__ __forceinline__ double stencil64(const double *a, double scale) {  // a[3] is the centre
    if (ORDER == 2) return scale * (-a[3] + a[4]);
    if (ORDER == 3) return scale * (-a[2] + a[4]);
    if (ORDER == 5) return scale * (8.0 * (-a[2] + a[4]) + a[1] - a[5]);
    return scale * (45.0 * (-a[2] + a[4]) + 9.0 * (a[1] - a[5]) - a[0] + a[6]);
}

template <int ORDER>
__global__ __launch_bounds__(BLK) void gradient64_kernel(const double *__restrict__ phi, int n, int log2n,
                                                         double *__restrict__ force) {
    constexpr int R = (ORDER == 2 || ORDER == 3) ? 1 : (ORDER == 5 ? 2 : 3);
    const double scale = ORDER == 2 ? (double)n : (ORDER == 3 ? 0.5 * n : (ORDER == 5 ? n / 12.0 : n / 60.0));
    const int64_t total = (int64_t)n * n * n;
    const int m = n - 1;
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLK) {
        const int k = (int)(t & m), j = (int)((t >> log2n) & m), i = (int)(t >> (2 * log2n));
        double a[7];
#pragma unroll
        for (int o = 0; o < 7; ++o) a[o] = 0.0;
#pragma unroll
        for (int o = -R; o <= R; ++o) a[o + 3] = phi[cell_index(n, log2n, i + o, j, k)];
        const double gx = stencil64<ORDER>(a, scale);
#pragma unroll
        for (int o = -R; o <= R; ++o) a[o + 3] = phi[cell_index(n, log2n, i, j + o, k)];
        const double gy = stencil64<ORDER>(a, scale);
#pragma unroll
        for (int o = -R; o <= R; ++o) a[o + 3] = phi[cell_index(n, log2n, i, j, k + o)];
        const double gz = stencil64<ORDER>(a, scale);
        force[3 * t] = gx, force[3 * t + 1] = gy, force[3 * t + 2] = gz;
    }
}

// Two consecutive k-cells per thread (n >= 8, 16-byte aligned arrays): the x and y neighbours arrive as double2, the z
// window as W + 1 aligned double2 (W = the stencil reach rounded up to even), the six components leave as three
// double2 -- 11 loads and 3 stores of 16 bytes for two cells at order 5, where the kernel above issues 24 and 6 of 8
// (1.9 -> ms at 512^3).  Same expression per cell (stencil64).
template <int ORDER>
__global__ __launch_bounds__(BLK) void gradient64_rows_kernel(const double *__restrict__ phi, int n, int log2n,
                                                              double *__restrict__ force) {
    constexpr int R = (ORDER == 2 || ORDER == 3) ? 1 : (ORDER == 5 ? 2 : 3), W = (R + 1) & ~1;
    const double scale = ORDER == 2 ? (double)n : (ORDER == 3 ? 0.5 * n : (ORDER == 5 ? n / 12.0 : n / 60.0));
    const int64_t total = ((int64_t)n * n * n) >> 1;
    const int m = n - 1;
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLK) {
        const int k = (int)((t << 1) & m), j = (int)((t >> (log2n - 1)) & m), i = (int)(t >> (2 * log2n - 1));
        double ax[2][7], ay[2][7], az[2][7], w[2 * W + 2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int o = 0; o < 7; ++o) ax[c][o] = ay[c][o] = az[c][o] = 0.0;
#pragma unroll
        for (int o = -R; o <= R; ++o) {
            if (o == 0 && ORDER != 2) continue;  // (only the order-2 stencil reads the centre)
            const double2 vx = *reinterpret_cast<const double2 *>(phi + cell_index(n, log2n, i + o, j, k));
            const double2 vy = *reinterpret_cast<const double2 *>(phi + cell_index(n, log2n, i, j + o, k));
            ax[0][o + 3] = vx.x, ax[1][o + 3] = vx.y, ay[0][o + 3] = vy.x, ay[1][o + 3] = vy.y;
        }
#pragma unroll
        for (int q = 0; q <= W; ++q) {
            const double2 v = *reinterpret_cast<const double2 *>(phi + cell_index(n, log2n, i, j, k - W + 2 * q));
            w[2 * q] = v.x, w[2 * q + 1] = v.y;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int o = -R; o <= R; ++o) az[c][o + 3] = w[W + c + o];
        double g[6];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            g[3 * c] = stencil64<ORDER>(ax[c], scale);
            g[3 * c + 1] = stencil64<ORDER>(ay[c], scale);
            g[3 * c + 2] = stencil64<ORDER>(az[c], scale);
        }
        double2 *o2 = reinterpret_cast<double2 *>(force + 6 * t);
        o2[0] = make_double2(g[0], g[1]), o2[1] = make_double2(g[2], g[3]), o2[2] = make_double2(g[4], g[5]);
    }
}

// The same marching along x (n >= 128; the fp32 mode's gradient_march_kernel in double): a workgroup owns a patch of 8
// rows x 32 double2 (64 k-cells) and walks 32 planes; a thread keeps its two cells of planes i - R .. i + R in registers
// (every plane of phi is loaded once along x), its y neighbours are the loads of the threads beside it (L1), its z
// window two more double2.  Same expression per cell (stencil64).
template <int ORDER>
__global__ __launch_bounds__(BLK) void gradient64_march_kernel(const double *__restrict__ phi, int n, int log2n, int seg,
                                                               double *__restrict__ force) {
    constexpr int R = (ORDER == 2 || ORDER == 3) ? 1 : (ORDER == 5 ? 2 : 3), W = (R + 1) & ~1;
    const double scale = ORDER == 2 ? (double)n : (ORDER == 3 ? 0.5 * n : (ORDER == 5 ? n / 12.0 : n / 60.0));
    const int nq = n >> 1, pk = nq >> 5, pj = n >> 3;
    const int patch = (int)(blockIdx.x % (pk * pj));
    const int i0 = (int)(blockIdx.x / (pk * pj)) * seg;
    const int k = ((patch % pk) * 32 + (threadIdx.x & 31)) << 1, j = (patch / pk) * 8 + (threadIdx.x >> 5);
    auto ld = [&](int ii, int jj, int kk) { return *reinterpret_cast<const double2 *>(phi + cell_index(n, log2n, ii, jj, kk)); };
    double2 w[2 * R + 1];
#pragma unroll
    for (int o = 0; o < 2 * R; ++o) w[o + 1] = ld(i0 - R + o, j, k);
#pragma unroll 1
    for (int i = i0; i < i0 + seg; ++i) {
#pragma unroll
        for (int o = 0; o < 2 * R; ++o) w[o] = w[o + 1];
        w[2 * R] = ld(i + R, j, k);
        double ax[2][7], ay[2][7], az[2][7], zw[2 * W + 2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int o = 0; o < 7; ++o) ax[c][o] = ay[c][o] = az[c][o] = 0.0;
#pragma unroll
        for (int o = -R; o <= R; ++o) {
            ax[0][o + 3] = w[o + R].x, ax[1][o + 3] = w[o + R].y;
            if (o == 0) continue;
            const double2 vy = ld(i, j + o, k);
            ay[0][o + 3] = vy.x, ay[1][o + 3] = vy.y;
        }
        ay[0][3] = w[R].x, ay[1][3] = w[R].y;
#pragma unroll
        for (int q = 0; q <= W; ++q) {
            const double2 v = (2 * q == W) ? w[R] : ld(i, j, k - W + 2 * q);
            zw[2 * q] = v.x, zw[2 * q + 1] = v.y;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int o = -R; o <= R; ++o) az[c][o + 3] = zw[W + c + o];
        double g[6];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            g[3 * c] = stencil64<ORDER>(ax[c], scale);
            g[3 * c + 1] = stencil64<ORDER>(ay[c], scale);
            g[3 * c + 2] = stencil64<ORDER>(az[c], scale);
        }
        double *o = force + 3 * cell_index(n, log2n, i, j, k);  // (read once, by the gather's tile fills: nontemporal)
#pragma unroll
        for (int m = 0; m < 6; ++m) __builtin_nontemporal_store(g[m], o + m);
    }
}

// ---- gather + second kick + CFL maxima ----------------------------------------------------------------------------
// 27 (8, 1) cells read through `at(a, b, c, m)`, summed in the reference's left-to-right order (mesh.py:3058-3086)
template <int SCHEME, class F>
__device__ __forceinline__ void gather27(double xp, double yp, double zp, int i, int j, int k, F at, double acc[3]) {
    if (SCHEME == PM_NGP) {
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[m] = at(0, 0, 0, m);
    } else if (SCHEME == PM_CIC) {
        double wx[2], wy[2], wz[2];
        int sx, sy, sz;
        cic_weights(xp, i, wx[0], wx[1], sx), cic_weights(yp, j, wy[0], wy[1], sy), cic_weights(zp, k, wz[0], wz[1], sz);
        bool first = true;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const double w = wx[a] * wy[b] * wz[c];
#pragma unroll
                    for (int m = 0; m < 3; ++m) acc[m] = first ? w * at(a * sx, b * sy, c * sz, m) : acc[m] + w * at(a * sx, b * sy, c * sz, m);
                    first = false;
                }
    } else {
        double wx[3], wy[3], wz[3];
        tsc_weights(xp, i, wx), tsc_weights(yp, j, wy), tsc_weights(zp, k, wz);
        bool first = true;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const double wxy = wx[a] * wy[b];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double w = wxy * wz[c];
#pragma unroll
                    for (int m = 0; m < 3; ++m) acc[m] = first ? w * at(a - 1, b - 1, c - 1, m) : acc[m] + w * at(a - 1, b - 1, c - 1, m);
                    first = false;
                }
            }
    }
}

// The workgroup is 512 threads with four particles each (the 2048-particle chunk of the deposit): the 78 KB tile allows
// two workgroups per CU either way, and with 256 threads that was two waves per SIMD to hide every latency of the
// kernel behind.  Positions (and the velocities the kick updates) are loaded once, up front, and stay in registers
// over the passes (deposit64_kernel's scheme: dense tiles through LDS, sparse ones through L2).
constexpr int IB = 512, IPT = CHUNK / IB, FILL_U = 10, AGE_SAMPLE = 16;

template <int SCHEME, bool KICK>
__global__ __launch_bounds__(IB) void interp64_kernel(const double *__restrict__ force, const double *__restrict__ pos,
                                                      int64_t np, int n, int log2n, double *__restrict__ out,
                                                      double *__restrict__ vel, double mhalf_dt,
                                                      unsigned long long *__restrict__ maxbits) {
    extern __shared__ double tile[];  // [3][TILE]: 77.8 KB (two workgroups per CU)
    __shared__ int s_pick[3];
    const int tid = threadIdx.x;
    const double nf = (double)n;
    const int64_t base = (int64_t)blockIdx.x * CHUNK;
    // (the tile keys stay in registers over the passes; position and velocity are read where a particle is served, in
    // rolled loops: one copy each of the LDS gather and of the L2 gather)
    int key[IPT];
    unsigned todo = 0;
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const int64_t p = base + r * IB + tid;
        key[r] = -1;
        if (p < np) {
            const double xp = pos[3 * p] * nf, yp = pos[3 * p + 1] * nf, zp = pos[3 * p + 2] * nf;
            key[r] = tile_key_of(cell_of(xp, n), cell_of(yp, n), cell_of(zp, n));
            todo |= 1u << r;
        }
    }
    unsigned long long ma = 0, mv = 0;
    unsigned direct = 0, npass = 0;  // particles served through L2, tile passes: what pm_real64_stream_age reports
    // serves particle slot r: dense -> from the LDS tile of t, else through L2
    // a particle's record on its way to the service loop: loaded one particle ahead (the first of a pass before the
    // tile fill), so that its two memory latencies lie behind the fill / the previous particle's 27 cells
    struct Slot {
        double x[3], v[3];
    };
    auto load = [&](int r) {
        Slot q;
        const int64_t p = min(base + (int64_t)max(r, 0) * IB + tid, np - 1);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            q.x[m] = pos[3 * p + m];
            q.v[m] = KICK ? __builtin_nontemporal_load(vel + 3 * p + m) : 0.0;
        }
        return q;
    };
    auto serve = [&](int r, const Slot &q, bool dense, const Tile &t) {
        const int64_t p = base + r * IB + tid;
        const double xp = q.x[0] * nf, yp = q.x[1] * nf, zp = q.x[2] * nf;
        const double *v = q.v;
        const int i = cell_of(xp, n), j = cell_of(yp, n), k = cell_of(zp, n);
        double a[3];
        if (dense) {
            const int c0 = ((i - t.ti * TI + 1) * HJ + (j - t.tj * TJ + 1)) * HK + (k - t.tk * TK + 1);
            gather27<SCHEME>(xp, yp, zp, i, j, k, [&](int da, int db, int dc, int m) { return tile[m * TILE + c0 + (da * HJ + db) * HK + dc]; }, a);
        } else {
            gather27<SCHEME>(xp, yp, zp, i, j, k,
                             [&](int da, int db, int dc, int m) { return force[3 * cell_index(n, log2n, i + da, j + db, k + dc) + m]; }, a);
            ++direct;
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            __builtin_nontemporal_store(a[m], out + 3 * p + m);
            ma = max(ma, (unsigned long long)__double_as_longlong(a[m]) & 0x7fffffffffffffffull);
            if (KICK) {
                const double w = v[m] + mhalf_dt * a[m];
                __builtin_nontemporal_store(w, vel + 3 * p + m);
                mv = max(mv, (unsigned long long)__double_as_longlong(w) & 0x7fffffffffffffffull);
            }
        }
    };
    // the particles of `mask`, lowest slot first, `first` being the record of the first one
    auto sweep = [&](unsigned mask, const Slot &first, bool dense, const Tile &t) {
        Slot nxt = first;
        int r = mask ? __ffs((int)mask) - 1 : -1;
        while (mask) {
            mask &= mask - 1u;
            const int rn = mask ? __ffs((int)mask) - 1 : -1;
            const Slot cur = nxt;
            if (rn >= 0) nxt = load(rn);
            serve(r, cur, dense, t);
            r = rn;
        }
    };
    const bool tiled = n >= 16;
    unsigned left = todo;  // served through L2 at the end: the sparse ones (everything when n < 16)
    if (tiled) {
        left = classify_sparse<IB, IPT>(todo, key, reinterpret_cast<int *>(tile));
        todo &= ~left;
    }
    for (int pass = 0; tiled && pass < PASS_MAX; ++pass) {
        int akey;
        unsigned mine;
        if (!next_tile<IB, IPT>(todo, key, s_pick, akey, mine)) break;  // (its barriers close the table's / the last pass's reads)
        const Tile t = tile_of_key(akey);
        ++npass;
        const bool dense = true;
        const Slot first = load(mine ? __ffs((int)mine) - 1 : 0);  // (in flight during the fill)
        {
            // the force tile: consecutive lanes, consecutive doubles of a row of 18 x 3; FILL_U loads in flight per thread
            // before the first is stored (one load per trip was 38 memory latencies in a row per pass: most of the kernel)
#pragma unroll 1
            for (int f0 = tid; f0 < 3 * TILE; f0 += IB * FILL_U) {
                double fv[FILL_U];
#pragma unroll
                for (int u = 0; u < FILL_U; ++u) {
                    const int f = min(f0 + u * IB, 3 * TILE - 1);
                    const int c = f / 3, m = f - 3 * c;
                    const int lk = c % HK, lij = c / HK, lj = lij % HJ, li = lij / HJ;
                    fv[u] = force[3 * cell_index(n, log2n, t.ti * TI - 1 + li, t.tj * TJ - 1 + lj, t.tk * TK - 1 + lk) + m];
                }
#pragma unroll
                for (int u = 0; u < FILL_U; ++u) {
                    const int f = f0 + u * IB;
                    if (f < 3 * TILE) tile[(f % 3) * TILE + f / 3] = fv[u];
                }
            }
            __syncthreads();
        }
        sweep(mine, first, dense, t);
        todo &= ~mine;
    }
    left |= todo;
    if (__ballot(left != 0u)) sweep(left, load(left ? __ffs((int)left) - 1 : 0), false, Tile{0, 0, 0});  // the sparse ones
    for (int off = 32; off > 0; off >>= 1) {
        ma = max(ma, (unsigned long long)__shfl_down((long long)ma, off, 64));
        mv = max(mv, (unsigned long long)__shfl_down((long long)mv, off, 64));
        direct += (unsigned)__shfl_down((int)direct, off, 64);
    }
    if ((tid & 63) == 0) {
        if (ma > __atomic_load_n(maxbits, __ATOMIC_RELAXED)) atomicMax(maxbits, ma);
        if (KICK && mv > __atomic_load_n(maxbits + 1, __ATOMIC_RELAXED)) atomicMax(maxbits + 1, mv);
        // (one workgroup in AGE_SAMPLE reports: every one of them on one address was 5 ms of the kernel at 512^3)
        if (direct && tiled && blockIdx.x % AGE_SAMPLE == 0) atomicAdd(maxbits + 2, (unsigned long long)direct * AGE_SAMPLE);
        if (tid == 0 && npass && blockIdx.x % AGE_SAMPLE == 0) atomicAdd(maxbits + 3, (unsigned long long)npass * AGE_SAMPLE);
    }
}

// ---- P(k) (fourier.py:22-100), every product and sum in double -----------------------------------------------------
__global__ __launch_bounds__(BLK) void pk64_kernel(const double2 *__restrict__ x, int n, int log2n, int p,
                                                   const double *__restrict__ sincs, double *__restrict__ hist) {
    extern __shared__ double lds[];  // 3 n: ksum, pksum, nmodes
    for (int b = threadIdx.x; b < 3 * n; b += BLK) lds[b] = 0.0;
    __syncthreads();
    const int nh = (n >> 1) + 1, middle = n >> 1, mask = n - 1;
    const int64_t total = (int64_t)n * n * nh;
    for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLK) {
        if (t == 0) continue;  // the k = 0 mode is skipped (fourier.py:80-82)
        const int k = (int)(t % nh);
        const int64_t ij = t / nh;
        const int j = (int)(ij & mask), i = (int)(ij >> log2n);
        const double kx = (i >= middle) ? (double)(i - n) : (double)i, ky = (j >= middle) ? (double)(j - n) : (double)j;
        const double kz = (double)k;
        const double comp = pow((sincs[i] * sincs[j]) * sincs[k], (double)(-2 * p));
        const double2 v = x[t];
        const double knorm = sqrt((kx * kx + ky * ky) + kz * kz);
        const int kidx = (int)(knorm + 0.5);
        atomicAdd(&lds[kidx], knorm);
        atomicAdd(&lds[n + kidx], (v.x * v.x + v.y * v.y) * comp);
        atomicAdd(&lds[2 * n + kidx], 1.0);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < 3 * n; b += BLK)
        if (lds[b] != 0.0) atomicAdd(&hist[b], lds[b]);
}

int blocks_for(int64_t items, int per_block) {
    int64_t b = (items + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > 262144 ? 262144 : b));
}

int build_plans64(pm_ctx *c) {
    if (c->plan64_r2c) return PM_OK;
    pm_fft_setup_once();
    const size_t len[3] = {(size_t)c->n, (size_t)c->n, (size_t)c->n};
    PM_FFT(rocfft_plan_create(&c->plan64_r2c, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                              rocfft_precision_double, 3, len, 1, nullptr));
    rocfft_plan_description d = nullptr;
    PM_FFT(rocfft_plan_description_create(&d));
    // ifft_3D_real carries 1/n^3 (np.fft.irfftn, fourier.py:277)
    PM_FFT(rocfft_plan_description_set_scale_factor(d, 1.0 / ((double)c->n * c->n * c->n)));
    PM_FFT(rocfft_plan_create(&c->plan64_c2r, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                              rocfft_precision_double, 3, len, 1, d));
    PM_FFT(rocfft_plan_description_destroy(d));
    size_t w1 = 0, w2 = 0;
    PM_FFT(rocfft_plan_get_work_buffer_size(c->plan64_r2c, &w1));
    PM_FFT(rocfft_plan_get_work_buffer_size(c->plan64_c2r, &w2));
    c->fft64_work_bytes = w1 > w2 ? w1 : w2;
    if (c->fft64_work_bytes) PM_HIP(hipMalloc(&c->fft64_work, c->fft64_work_bytes));
    PM_FFT(rocfft_execution_info_create(&c->info64_r2c));
    PM_FFT(rocfft_execution_info_create(&c->info64_c2r));
    if (c->fft64_work_bytes) {
        PM_FFT(rocfft_execution_info_set_work_buffer(c->info64_r2c, c->fft64_work, c->fft64_work_bytes));
        PM_FFT(rocfft_execution_info_set_work_buffer(c->info64_c2r, c->fft64_work, c->fft64_work_bytes));
    }
    const size_t nspec = (size_t)c->n * c->n * (c->n / 2 + 1);
    PM_HIP(hipMalloc(&c->d_spec64, nspec * 2 * sizeof(double)));
    PM_HIP(hipMalloc((void **)&c->d_tab64, 3 * (size_t)c->n * sizeof(double)));
    return PM_OK;
}

int fft64(pm_ctx *c, bool forward, void *in, void *out) {
    // (rocFFT reads the stream from the execution info: set on every call, the caller may have changed it)
    rocfft_execution_info info = forward ? c->info64_r2c : c->info64_c2r;
    PM_FFT(rocfft_execution_info_set_stream(info, c->stream));
    void *i[1] = {in}, *o[1] = {out};
    PM_FFT(rocfft_execute(forward ? c->plan64_r2c : c->plan64_c2r, i, o, info));
    return PM_OK;
}

}  // namespace

#define PM_REQUIRE_F64(c) PM_REQUIRE((c) && (c)->real_bytes == 8, "this context was not created with real_bytes = 8 (pm_ctx_create_real)")

extern "C" int pm_real64_kick_drift_wrap(pm_ctx *c, double *d_pos, double *d_vel, const double *d_acc, double half_dt,
                                         double dt, int64_t np) {
    PM_REQUIRE_F64(c);
    PM_REQUIRE(d_pos && d_vel && d_acc && np >= 0, "bad argument");
    if (np == 0) return PM_OK;
    hipLaunchKernelGGL(kdw64_kernel, dim3(blocks_for(3 * np, BLK * 4)), dim3(BLK), 0, c->stream, d_pos, d_vel, d_acc, -half_dt,
                       dt, 3 * np);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

extern "C" int pm_real64_scale_add(pm_ctx *c, double *d_x, double a, double b, int64_t n) {
    PM_REQUIRE_F64(c);
    PM_REQUIRE(d_x && n >= 0, "bad argument");
    if (n == 0) return PM_OK;
    hipLaunchKernelGGL(axpb64_kernel, dim3(blocks_for(n, BLK * 4)), dim3(BLK), 0, c->stream, d_x, a, b, n);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

extern "C" int pm_real64_max_abs(pm_ctx *c, const double *d_x, int64_t n, double *h_out) {
    PM_REQUIRE_F64(c);
    PM_REQUIRE(d_x && h_out && n > 0, "bad argument");
    unsigned long long *d_bits = reinterpret_cast<unsigned long long *>(c->d_scratch + 48);
    PM_HIP(hipMemsetAsync(d_bits, 0, sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(maxabs64_kernel, dim3(blocks_for(n, BLK * 8)), dim3(BLK), 0, c->stream, d_x, n, d_bits);
    PM_LAUNCH_CHECK();
    return pm_read_back(c, h_out, d_bits, sizeof(double));
}

template <bool KDW>
static int deposit64_launch(pm_ctx *c, int scheme, double *d_pos, int64_t np, double *d_rho, double *d_vel, const double *d_acc,
                            double mhalf_dt, double dt) {
    const int n = c->n;
    PM_HIP(hipMemsetAsync(d_rho, 0, (size_t)n * n * n * sizeof(double), c->stream));
    if (np == 0) return PM_OK;
    const dim3 grid((unsigned)((np + CHUNK - 1) / CHUNK));
    switch (scheme) {
        case PM_NGP: hipLaunchKernelGGL((deposit64_kernel<PM_NGP, KDW>), grid, dim3(BLK), 0, c->stream, d_pos, np, n, c->log2n, d_rho, d_vel, d_acc, mhalf_dt, dt); break;
        case PM_CIC: hipLaunchKernelGGL((deposit64_kernel<PM_CIC, KDW>), grid, dim3(BLK), 0, c->stream, d_pos, np, n, c->log2n, d_rho, d_vel, d_acc, mhalf_dt, dt); break;
        case PM_TSC: hipLaunchKernelGGL((deposit64_kernel<PM_TSC, KDW>), grid, dim3(BLK), 0, c->stream, d_pos, np, n, c->log2n, d_rho, d_vel, d_acc, mhalf_dt, dt); break;
        default: pm_set_error("mass scheme %d, should be NGP(1), CIC(2) or TSC(3)", scheme); return PM_EINVAL;
    }
    PM_LAUNCH_CHECK();
    return PM_OK;
}

extern "C" int pm_real64_deposit(pm_ctx *c, int scheme, const double *d_pos, int64_t np, double *d_rho) {
    PM_REQUIRE_F64(c);
    PM_REQUIRE(d_pos && d_rho && np >= 0, "bad argument");
    return deposit64_launch<false>(c, scheme, const_cast<double *>(d_pos), np, d_rho, nullptr, nullptr, 0.0, 0.0);
}

// pm_real64_kick_drift_wrap followed by pm_real64_deposit of the new positions, in one pass over the particles (the
// same expressions, particle by particle: the same position, velocity and -- up to the order of the double sums -- density)
extern "C" int pm_real64_kick_drift_wrap_deposit(pm_ctx *c, int scheme, double *d_pos, double *d_vel, const double *d_acc,
                                                 double half_dt, double dt, int64_t np, double *d_rho) {
    PM_REQUIRE_F64(c);
    PM_REQUIRE(d_pos && d_vel && d_acc && d_rho && np >= 0, "bad argument");
    return deposit64_launch<true>(c, scheme, d_pos, np, d_rho, d_vel, d_acc, -half_dt, dt);
}

extern "C" int pm_real64_poisson_fft(pm_ctx *c, int kind, int p, const double *d_rhs, double *d_phi, const double *h_k2,
                                     const double *h_s, double c0) {
    PM_REQUIRE_F64(c);
    PM_REQUIRE(d_rhs && d_phi && h_k2 && h_s, "NULL argument");
    PM_REQUIRE(kind != PM_GREEN_COMPENSATED || (p >= 1 && p <= 16), "the compensated Green's function needs the mass-scheme index p >= 1");
    int rc = build_plans64(c);
    if (rc) return rc;
    const int n = c->n;
    rc = fft64(c, true, const_cast<double *>(d_rhs), c->d_spec64);  // (rocFFT's not-in-place real transforms may overwrite their input: callers pass a density they no longer need)
    if (rc) return rc;
    PM_HIP(hipMemcpyAsync(c->d_tab64, h_k2, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    PM_HIP(hipMemcpyAsync(c->d_tab64 + n, h_s, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const int64_t total = (int64_t)n * n * (n / 2 + 1);
    const dim3 grid(blocks_for(total, BLK * 2));
    double2 *x = reinterpret_cast<double2 *>(c->d_spec64);
    switch (kind) {
        case PM_GREEN_PLAIN: hipLaunchKernelGGL((greens64_kernel<PM_GREEN_PLAIN>), grid, dim3(BLK), 0, c->stream, x, n, p, c->d_tab64, c->d_tab64 + n, c0); break;
        case PM_GREEN_COMPENSATED: hipLaunchKernelGGL((greens64_kernel<PM_GREEN_COMPENSATED>), grid, dim3(BLK), 0, c->stream, x, n, p, c->d_tab64, c->d_tab64 + n, c0); break;
        case PM_GREEN_7PT: hipLaunchKernelGGL((greens64_kernel<PM_GREEN_7PT>), grid, dim3(BLK), 0, c->stream, x, n, p, c->d_tab64, c->d_tab64 + n, c0); break;
        default: pm_set_error("unknown Green's function kind %d", kind); return PM_EINVAL;
    }
    PM_LAUNCH_CHECK();
    return fft64(c, false, c->d_spec64, d_phi);
}

extern "C" int pm_real64_gradient(pm_ctx *c, int order, const double *d_phi, double *d_force) {
    PM_REQUIRE_F64(c);
    PM_REQUIRE(d_phi && d_force, "NULL argument");
    const int n = c->n;
    if (n >= 128 && pm_aligned16(d_phi) && pm_aligned16(d_force)) {
        const int seg = 32;
        const dim3 gm((unsigned)(((n >> 1) >> 5) * (n >> 3) * (n / seg)));
        switch (order) {
            case 2: hipLaunchKernelGGL((gradient64_march_kernel<2>), gm, dim3(BLK), 0, c->stream, d_phi, n, c->log2n, seg, d_force); break;
            case 3: hipLaunchKernelGGL((gradient64_march_kernel<3>), gm, dim3(BLK), 0, c->stream, d_phi, n, c->log2n, seg, d_force); break;
            case 5: hipLaunchKernelGGL((gradient64_march_kernel<5>), gm, dim3(BLK), 0, c->stream, d_phi, n, c->log2n, seg, d_force); break;
            case 7: hipLaunchKernelGGL((gradient64_march_kernel<7>), gm, dim3(BLK), 0, c->stream, d_phi, n, c->log2n, seg, d_force); break;
            default: pm_set_error("gradient_order=%d, should be 2, 3, 5 or 7", order); return PM_EINVAL;
        }
        PM_LAUNCH_CHECK();
        return PM_OK;
    }
    if (n >= 8 && pm_aligned16(d_phi) && pm_aligned16(d_force)) {
        const dim3 grid2(blocks_for((int64_t)n * n * n / 2, BLK));
        switch (order) {
            case 2: hipLaunchKernelGGL((gradient64_rows_kernel<2>), grid2, dim3(BLK), 0, c->stream, d_phi, n, c->log2n, d_force); break;
            case 3: hipLaunchKernelGGL((gradient64_rows_kernel<3>), grid2, dim3(BLK), 0, c->stream, d_phi, n, c->log2n, d_force); break;
            case 5: hipLaunchKernelGGL((gradient64_rows_kernel<5>), grid2, dim3(BLK), 0, c->stream, d_phi, n, c->log2n, d_force); break;
            case 7: hipLaunchKernelGGL((gradient64_rows_kernel<7>), grid2, dim3(BLK), 0, c->stream, d_phi, n, c->log2n, d_force); break;
            default: pm_set_error("gradient_order=%d, should be 2, 3, 5 or 7", order); return PM_EINVAL;
        }
        PM_LAUNCH_CHECK();
        return PM_OK;
    }
    const dim3 grid(blocks_for((int64_t)n * n * n, BLK));
    switch (order) {
        case 2: hipLaunchKernelGGL((gradient64_kernel<2>), grid, dim3(BLK), 0, c->stream, d_phi, n, c->log2n, d_force); break;
        case 3: hipLaunchKernelGGL((gradient64_kernel<3>), grid, dim3(BLK), 0, c->stream, d_phi, n, c->log2n, d_force); break;
        case 5: hipLaunchKernelGGL((gradient64_kernel<5>), grid, dim3(BLK), 0, c->stream, d_phi, n, c->log2n, d_force); break;
        case 7: hipLaunchKernelGGL((gradient64_kernel<7>), grid, dim3(BLK), 0, c->stream, d_phi, n, c->log2n, d_force); break;
        default: pm_set_error("gradient_order=%d, should be 2, 3, 5 or 7", order); return PM_EINVAL;
    }
    PM_LAUNCH_CHECK();
    return PM_OK;
}

template <int SCHEME>
static int interp64_launch(pm_ctx *c, const double *force, const double *pos, int64_t np, double *out, double *vel,
                           double half_dt, unsigned long long *bits) {
    const dim3 grid((unsigned)((np + CHUNK - 1) / CHUNK));
    const size_t lds = 3 * TILE * sizeof(double);
    if (vel) {
        PM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&interp64_kernel<SCHEME, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((interp64_kernel<SCHEME, true>), grid, dim3(IB), lds, c->stream, force, pos, np, c->n, c->log2n, out,
                           vel, -half_dt, bits);
    } else {
        PM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&interp64_kernel<SCHEME, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((interp64_kernel<SCHEME, false>), grid, dim3(IB), lds, c->stream, force, pos, np, c->n, c->log2n, out,
                           vel, 0.0, bits);
    }
    PM_LAUNCH_CHECK();
    return PM_OK;
}

// inv<scheme>_vec of the [n,n,n,3] force at the particles -> d_acc; with d_vel: v += -half_dt * a in the same pass
// (the second kick of the leapfrog); max|a| and (with d_vel) max|v| come back for the next step's dt criteria.
extern "C" int pm_real64_interp_kick(pm_ctx *c, int scheme, const double *d_force, const double *d_pos, int64_t np,
                                     double *d_acc, double *d_vel, double half_dt, double *h_max_acc, double *h_max_vel) {
    PM_REQUIRE_F64(c);
    PM_REQUIRE(d_force && d_pos && d_acc && np > 0, "bad argument");
    unsigned long long *d_bits = reinterpret_cast<unsigned long long *>(c->d_scratch + 48);
    PM_HIP(hipMemsetAsync(d_bits, 0, 4 * sizeof(unsigned long long), c->stream));
    int rc;
    switch (scheme) {
        case PM_NGP: rc = interp64_launch<PM_NGP>(c, d_force, d_pos, np, d_acc, d_vel, half_dt, d_bits); break;
        case PM_CIC: rc = interp64_launch<PM_CIC>(c, d_force, d_pos, np, d_acc, d_vel, half_dt, d_bits); break;
        case PM_TSC: rc = interp64_launch<PM_TSC>(c, d_force, d_pos, np, d_acc, d_vel, half_dt, d_bits); break;
        default: pm_set_error("mass scheme %d, should be NGP(1), CIC(2) or TSC(3)", scheme); return PM_EINVAL;
    }
    if (rc) return rc;
    double m[4] = {0.0, 0.0, 0.0, 0.0};
    c->age64_np = -1;
    if (h_max_acc || h_max_vel) {
        rc = pm_read_back(c, m, d_bits, sizeof(m));
        if (rc) return rc;
        unsigned long long aged, passes;
        memcpy(&aged, &m[2], sizeof(aged));
        memcpy(&passes, &m[3], sizeof(passes));
        c->age64_count = (int64_t)aged, c->age64_passes = (int64_t)passes, c->age64_np = np;
    }
    if (h_max_acc) *h_max_acc = m[0];
    if (h_max_vel) *h_max_vel = m[1];
    return PM_OK;
}

// The age of the particle stream as the last pm_real64_interp_kick (with read-back) found it: how many particles it
// served one by one through L2 -- members of a tile with fewer than DENSE_MIN of the chunk's particles -- of how many,
// and how many tile passes its *h_chunks workgroups made.  A Morton-sorted stream gives the few particles of near-empty
// tiles and at most (tiles + chunks) passes (two per chunk at one particle per cell); a shuffled one nearly every
// particle; the in-place steps let both grow (the kernels serve every particle whatever the order: this is a performance matter only), and the host
// re-sorts past a threshold (pysco_amd.main.reorder_policy), as it does in the fp32 mode on the straggler fraction.
// *h_np = -1 before any gather.
extern "C" int pm_real64_stream_age(pm_ctx *c, int64_t *h_aged, int64_t *h_np, int64_t *h_passes, int64_t *h_chunks) {
    PM_REQUIRE_F64(c);
    PM_REQUIRE(h_aged && h_np && h_passes && h_chunks, "NULL argument");
    *h_aged = c->age64_count, *h_np = c->age64_np, *h_passes = c->age64_passes;
    *h_chunks = c->age64_np > 0 ? (c->age64_np + CHUNK - 1) / CHUNK : 0;
    return PM_OK;
}

// utils.reorder_particles (utils.py:1019-1116) for float64 arrays: stable sort by Morton key (morton.py:42-137, 21 bits
// per axis), velocity / acceleration (optional, in pairs) carried along, d_perm (optional): new[i] = old[perm[i]].
// key_bits (1..21): only the top key_bits bits per axis decide the order; -1: only the tile of the context's mesh does
// (all the kernels care about: 16 bits at 512^3, two radix passes); 0 = all 21 bits per axis, the reference's key.
namespace {
__device__ __forceinline__ unsigned long long interleave21(unsigned long long x) {
    x &= 0x1FFFFFull;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
__global__ __launch_bounds__(BLK) void morton_keys64_kernel(const double *__restrict__ pos, int64_t np,
                                                            unsigned long long *__restrict__ keys, unsigned *__restrict__ iota) {
    for (int64_t p = (int64_t)blockIdx.x * BLK + threadIdx.x; p < np; p += (int64_t)gridDim.x * BLK) {
        const unsigned long long xx = interleave21((unsigned long long)(long long)floor(pos[3 * p] * 2097152.0));
        const unsigned long long yy = interleave21((unsigned long long)(long long)floor(pos[3 * p + 1] * 2097152.0));
        const unsigned long long zz = interleave21((unsigned long long)(long long)floor(pos[3 * p + 2] * 2097152.0));
        keys[p] = xx << 2 | yy << 1 | zz;
        iota[p] = (unsigned)p;
    }
}
__global__ __launch_bounds__(BLK) void gather3_64_kernel(const unsigned *__restrict__ perm, int64_t np,
                                                         const double *__restrict__ a, const double *__restrict__ b,
                                                         const double *__restrict__ cc, double *__restrict__ ao,
                                                         double *__restrict__ bo, double *__restrict__ co,
                                                         int64_t *__restrict__ perm_out) {
    for (int64_t p = (int64_t)blockIdx.x * BLK + threadIdx.x; p < np; p += (int64_t)gridDim.x * BLK) {
        const int64_t s = perm[p];
        ao[3 * p] = a[3 * s], ao[3 * p + 1] = a[3 * s + 1], ao[3 * p + 2] = a[3 * s + 2];
        if (b) bo[3 * p] = b[3 * s], bo[3 * p + 1] = b[3 * s + 1], bo[3 * p + 2] = b[3 * s + 2];
        if (cc) co[3 * p] = cc[3 * s], co[3 * p + 1] = cc[3 * s + 1], co[3 * p + 2] = cc[3 * s + 2];
        if (perm_out) perm_out[p] = s;
    }
}
}  // namespace

// keys in / out, values in / out and rocPRIM's scratch of the sort, in the context's sort workspace (grown on demand: a
// device allocation of a few GB, ~0.15 s -- pm_real64_reserve_sort takes it out of the step loop)
static int sort64_workspace(pm_ctx *c, int64_t np, size_t *tmp_bytes, size_t *karr, size_t *varr) {
    unsigned long long *nullk = nullptr;
    unsigned *nullv = nullptr;
    *tmp_bytes = 0;
    PM_HIP(rocprim::radix_sort_pairs(nullptr, *tmp_bytes, nullk, nullk, nullv, nullv, (size_t)np, 0, 63, c->stream));
    *karr = ((size_t)np * sizeof(unsigned long long) + 255) & ~(size_t)255;
    *varr = ((size_t)np * sizeof(unsigned) + 255) & ~(size_t)255;
    const size_t need = 2 * *karr + 2 * *varr + *tmp_bytes;
    if (c->sort_tmp_bytes < need) {
        if (c->sort_tmp) PM_HIP(hipFree(c->sort_tmp));
        c->sort_tmp = nullptr, c->sort_tmp_bytes = 0;
        PM_HIP(hipMalloc(&c->sort_tmp, need));
        c->sort_tmp_bytes = need;
    }
    return PM_OK;
}

extern "C" int pm_real64_reserve_sort(pm_ctx *c, int64_t np) {
    PM_REQUIRE_F64(c);
    PM_REQUIRE(np > 0 && np < 2147483647LL, "bad argument");
    size_t a, b, d;
    return sort64_workspace(c, np, &a, &b, &d);
}

extern "C" int pm_real64_reorder_particles(pm_ctx *c, int64_t np, const double *d_pos, const double *d_vel,
                                           const double *d_acc, double *d_pos_out, double *d_vel_out, double *d_acc_out,
                                           int64_t *d_perm, int key_bits) {
    PM_REQUIRE_F64(c);
    PM_REQUIRE(np >= 0 && key_bits >= -1 && key_bits <= 21, "bad argument");
    if (np == 0) return PM_OK;
    PM_REQUIRE(d_pos && d_pos_out && d_pos != d_pos_out, "pos / pos_out must be distinct non-NULL buffers");
    PM_REQUIRE((!d_vel) == (!d_vel_out) && (!d_acc) == (!d_acc_out), "in/out arrays must come in pairs");
    PM_REQUIRE(np < 2147483647LL, "rocPRIM radix sort is limited to 2^31-1 keys per call");
    // (-1: to the tiles of this context's mesh -- a tile is 2^11 consecutive cell codes, LI + LJ + LK bits)
    const int tile_bits = 3 * c->log2n - (LI + LJ + LK);
    const unsigned begin_bit = key_bits > 0 ? 3u * (unsigned)(21 - key_bits) : key_bits < 0 ? (unsigned)(63 - (tile_bits > 1 ? tile_bits : 1)) : 0u;
    size_t tmp_bytes = 0, karr = 0, varr = 0;
    int rc = sort64_workspace(c, np, &tmp_bytes, &karr, &varr);
    if (rc) return rc;
    char *w = (char *)c->sort_tmp;
    unsigned long long *keys_in = (unsigned long long *)w, *keys_out = (unsigned long long *)(w + karr);
    unsigned *iota = (unsigned *)(w + 2 * karr), *perm = (unsigned *)(w + 2 * karr + varr);
    void *rp_tmp = w + 2 * karr + 2 * varr;
    hipLaunchKernelGGL(morton_keys64_kernel, dim3(blocks_for(np, BLK)), dim3(BLK), 0, c->stream, d_pos, np, keys_in, iota);
    PM_LAUNCH_CHECK();
    PM_HIP(rocprim::radix_sort_pairs(rp_tmp, tmp_bytes, keys_in, keys_out, iota, perm, (size_t)np, begin_bit, 63, c->stream));
    hipLaunchKernelGGL(gather3_64_kernel, dim3(blocks_for(np, BLK)), dim3(BLK), 0, c->stream, perm, np, d_pos, d_vel, d_acc,
                       d_pos_out, d_vel_out, d_acc_out, d_perm);
    PM_LAUNCH_CHECK();
    c->age64_np = -1;  // (what the last gather found no longer describes this stream)
    return PM_OK;
}

// Binned, window-deconvolved power spectrum of a real grid (fourier.fourier_grid_to_Pk o fourier.fft_3D_real): sums per
// bin of |k|, of |delta_k|^2 W^(-2p) and of the mode count, n entries each (the caller divides, fourier.py:96-100).
extern "C" int pm_real64_pk(pm_ctx *c, int p, double *d_rho, const double *h_sinc, double *h_ksum, double *h_pksum,
                            double *h_nmodes) {
    PM_REQUIRE_F64(c);
    PM_REQUIRE(d_rho && h_sinc && h_ksum && h_pksum && h_nmodes, "NULL argument");
    const int n = c->n;
    PM_REQUIRE((size_t)3 * n * sizeof(double) <= 64 * 1024, "n too large for the LDS histogram");
    int rc = build_plans64(c);
    if (rc) return rc;
    rc = fft64(c, true, d_rho, c->d_spec64);
    if (rc) return rc;
    PM_HIP(hipMemcpyAsync(c->d_tab64 + 2 * n, h_sinc, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    PM_HIP(hipMemsetAsync(c->d_hist, 0, 3 * (size_t)n * sizeof(double), c->stream));
    int blocks = blocks_for((int64_t)n * n * (n / 2 + 1), BLK * 16);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(pk64_kernel, dim3(blocks), dim3(BLK), 3 * n * sizeof(double), c->stream,
                       reinterpret_cast<const double2 *>(c->d_spec64), n, c->log2n, p, c->d_tab64 + 2 * n, c->d_hist);
    PM_LAUNCH_CHECK();
    PM_HIP(hipMemcpyAsync(h_ksum, c->d_hist, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PM_HIP(hipMemcpyAsync(h_pksum, c->d_hist + n, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    return pm_read_back(c, h_nmodes, c->d_hist + 2 * n, n * sizeof(double));
}
