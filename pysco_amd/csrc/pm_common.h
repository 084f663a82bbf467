// Internal header shared by the libpmstep translation units (gfx950 only).
#pragma once
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/pmstep.h"

#define PM_OK 0
#define PM_EINVAL (-1)
#define PM_EHIP (-2)
#define PM_EFFT (-3)
#define PM_ENOSUP (-4)

void pm_set_error(const char *fmt, ...);

#define PM_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            pm_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_));     \
            return PM_EHIP;                                                                       \
        }                                                                                         \
    } while (0)

#define PM_FFT(call)                                                                              \
    do {                                                                                          \
        rocfft_status s_ = (call);                                                                \
        if (s_ != rocfft_status_success) {                                                        \
            pm_set_error("%s:%d %s -> rocfft_status %d", __FILE__, __LINE__, #call, (int)s_);     \
            return PM_EFFT;                                                                       \
        }                                                                                         \
    } while (0)

#define PM_REQUIRE(cond, msg)                                                                     \
    do {                                                                                          \
        if (!(cond)) {                                                                            \
            pm_set_error("%s: %s", __func__, msg);                                                \
            return PM_EINVAL;                                                                     \
        }                                                                                         \
    } while (0)

#define PM_LAUNCH_CHECK() PM_HIP(hipGetLastError())

// Grid geometry seen by the particle/stencil kernels.  y and z are always the full periodic
// n = 2^log2n cells; x is either the full periodic axis (nx == n, x0 == 0) or, in slab mode, the
// nx local planes (ghosts included) that start at global plane x0 (periodic origin).
struct pm_geom {
    int n, log2n, mask;
    int nx, x0;
};

#ifdef __HIPCC__
// linear index of global cell (gi, gj, gk) in the local array [nx, n, n]; out-of-slab planes are
// clamped (memory safety only: callers keep particles inside their slab + ghosts)
__device__ __forceinline__ int64_t pm_cell(const pm_geom &g, int gi, int gj, int gk) {
    int li = (gi - g.x0) & g.mask;
    li = li < g.nx ? li : g.nx - 1;
    return ((((int64_t)li << g.log2n) + (gj & g.mask)) << g.log2n) + (gk & g.mask);
}

// utils.periodic_wrap (utils.py:1120-1149); eps = -(2^-25)*(1+1e-6) as a double
// Written as selects: the early-return form compiled to three branches per coordinate inside the
// particle loops.  For a float y, (double)y > eps is y >= -(2^-25)(1 + 8*2^-23): eight ulps below
// -2^-25 is the last float above eps, so the test needs no fp64.
__device__ __forceinline__ float pm_wrap01(float y) {
    const float eps_f = __uint_as_float(0xb3000008u);  // -(2^-25)(1 + 8*2^-23)
    const float neg = (y >= eps_f) ? 0.0f : y + 1.0f;
    float r = (y >= 1.0f) ? y - 1.0f : y;
    r = (y < 0.0f) ? neg : r;
    return r;
}

// leapfrog first half for one coordinate (integration.py:250-258): v += -half_dt*a; x += dt*v; wrap(x)
// (the reference's a == +-1 special cases of y += a*x, utils.py:289-295, need no branch: 1*v and -1*v are
// exact, so x + dt*v rounds once exactly like x + v and x - v; the build never contracts to an fma)
__device__ __forceinline__ void pm_kick_drift_wrap_one(float &x, float &v, float a, float mhalf_dt, float dt) {
    v = v + mhalf_dt * a;
    x = x + dt * v;
    x = pm_wrap01(x);
}

// 1-D gradient stencils of mesh.py:639-850 (and the f(R) fused variants :860-2069), shared by the grid
// gradient kernels (pm_grid.hip) and the gather that differentiates its own phi tile (pm_particles.hip)
template <int FRN>
__device__ __forceinline__ float fr_pow(float b) {
    return FRN == 1 ? b * b : b * b * b;
}

template <int ORDER, int FRN>
struct Stencil {
    // value of the 1-D stencil given a[-3..3] (and b when FRN>0); operation order as the reference
    static __device__ __forceinline__ float eval(const float *a, const float *b, float f, float scale) {
        // a[3] is the centre
        if (ORDER == 2) {
            if (FRN == 0) return scale * (-a[3] + a[4]);
            return scale * (-a[3] + a[4] + f * (-fr_pow<FRN>(b[3]) + fr_pow<FRN>(b[4])));
        } else if (ORDER == 3) {
            if (FRN == 0) return scale * (-a[2] + a[4]);
            return scale * (-a[2] + a[4] + f * (-fr_pow<FRN>(b[2]) + fr_pow<FRN>(b[4])));
        } else if (ORDER == 5) {
            if (FRN == 0) return scale * (8.0f * (-a[2] + a[4]) + a[1] - a[5]);
            return scale * (8.0f * (-a[2] + a[4] + f * (-fr_pow<FRN>(b[2]) + fr_pow<FRN>(b[4]))) + a[1] - a[5] +
                            f * (fr_pow<FRN>(b[1]) - fr_pow<FRN>(b[5])));
        } else {
            if (FRN == 0) return scale * (45.0f * (-a[2] + a[4]) + 9.0f * (a[1] - a[5]) - a[0] + a[6]);
            return scale * (45.0f * (-a[2] + a[4] + f * (-fr_pow<FRN>(b[2]) + fr_pow<FRN>(b[4]))) +
                            9.0f * (+a[1] - a[5] + f * (fr_pow<FRN>(b[1]) - fr_pow<FRN>(b[5]))) - a[0] + a[6] +
                            f * (-fr_pow<FRN>(b[0]) + fr_pow<FRN>(b[6])));
        }
    }
};

// Small device -> host read-back on the context's stream, synchronised: through pinned memory (a copy into pageable
// memory is staged by the runtime and costs 10-20 us more; the step does five of these at 128^3).
int pm_read_back(pm_ctx *c, void *h_dst, const void *d_src, size_t bytes);
void pm_fft_setup_once();  // rocfft_setup(), once per process
constexpr size_t PM_PINNED_BYTES = 16384;

#endif

constexpr int PM_SCRATCH_SINK = 64, PM_SINK_BYTES_AT = 192, PM_SCRATCH_WORDS = 288;  // sink: 192 words of records + 64 flag bytes

struct pm_ctx {
    int n;          // cells per side
    int log2n;
    int slab_nx, slab_x0;  // geometry of the particle <-> mesh kernels (pm_set_slab); default n, 0
    hipStream_t stream;
    // rocFFT
    rocfft_plan plan_r2c, plan_c2r;
    rocfft_execution_info info_r2c, info_c2r;
    void *fft_work;
    size_t fft_work_bytes;
    // the same pair on the 3n/2 grid of the de-aliased initial conditions (pm_fft_r2c_padded / pm_fft_c2r_padded)
    rocfft_plan pad_r2c, pad_c2r;
    rocfft_execution_info pad_info_r2c, pad_info_c2r;
    void *pad_work;
    void *d_spec;  // lazily allocated half-spectrum scratch for pm_poisson_fft
    // the fp64 device mode (pm_ctx_create_real(..., 8); pm_real64.hip): double-precision plans, spectrum, tables
    int real_bytes;  // 4 (default) or 8
    rocfft_plan plan64_r2c, plan64_c2r;
    rocfft_execution_info info64_r2c, info64_c2r;
    void *fft64_work, *d_spec64;
    size_t fft64_work_bytes;
    double *d_tab64;  // 3 n doubles
    void *fp_B, *fp_C, *fp_tw;  // fused FFT Poisson solve: two transposed half-spectra + twiddle table
    size_t fp_B_bytes, fp_C_bytes;
    float *h_pow;               // host staging of the per-axis sinc^(-2p) table
    // slab-decomposed FFT pieces (multi-GPU): batched 2-D r2c/c2r over (y,z), strided 1-D c2c over x
    rocfft_plan plan_s[4];            // 0: 2-D r2c, 1: 2-D c2r, 2: 1-D forward, 3: 1-D backward
    rocfft_execution_info info_s[4];
    void *work_s[4];
    long long key_s[4];               // batch (2-D) or stride (1-D) the cached plan was built for
    // small device scratch: [0..63] words for reductions / flags, [PM_SCRATCH_SINK, +192) the store sink of the
    // re-bucketing gather (lanes without a particle); then 3 tables of n floats
    unsigned *d_scratch;
    float *d_tab;      // 3*n floats (k2, s, spare)
    double *d_hist;    // 3*n doubles (P(k) bins)
    hipEvent_t ev0, ev1;
    // tile-range table of the one-workgroup-per-tile kernels (grown on demand)
    void *tile_tmp;
    size_t tile_tmp_bytes;
    int gather_mode;  // 1: tile-range gather (default), 2: chunk/anchor gather
    int tile_x_lo, tile_nx;  // planes the particles live in (pm_set_tile_window); 0, n on a single GPU
    // prepared tile ranges + straggler lists (pm_tile_lists): valid for exactly these positions
    const float *lists_pos;
    int64_t lists_np;
    unsigned lists_blocks;
    int lists_use;       // 1: every straggler is in the per-tile incoming lists
    int64_t last_stragglers, last_np;  // of the most recent straggler scan (pm_tile_lists_info); -1 before any
    int last_complete;
    int lists_sticky;    // prepared by pm_kick_drift_wrap_deposit: the next pm_tile_lists on them keeps them
    const void *lists_ptr[4];  // start, xstart (tile_tmp); idx, lstart (list_buf)
    void *list_buf;      // keys/indices (in, out), per-tile list offsets, sort scratch
    size_t list_buf_bytes;
    // incremental re-sort (pm_gradient_interp_kick_rebucket): per-tile / per-workgroup counts of the particles that
    // stayed in the tile they are filed under (written by the fused kick + drift + deposit), the tile offsets of
    // the re-bucketed stream (two tables: the one the current layout was written with, the one being built)
    void *rb_buf;
    size_t rb_buf_bytes;
    // "the kick re-buckets" (pm_gradient_interp_kick_codes -> pm_kick_drift_wrap_deposit_rebucket): flags, per-workgroup and
    // per-tile stay counts, incoming lists; valid for exactly (kr_pos, kr_np) until the next fused kick consumes them
    void *kr_buf;
    size_t kr_buf_bytes;
    int kr_valid, kr_checked;
    unsigned kr_blocks;  // workgroups of the gather's launch (the ranges it walked: KrBufs::start / xstart)
    const float *kr_pos;
    int64_t kr_np;
    int stay_valid;            // the destination counts belong to the prepared lists (lists_pos)
    unsigned char *codes_user;        // where the fused kick records the destination codes on a slab (pm_set_destination_codes)
    const unsigned char *stay_codes;  // the codes the destination counts were made from (the gather's Rebucket::dcode)
    float dt_pred, stay_dt_pred;  // guess of the next step's dt (pm_set_predicted_dt); the one the counts were made with
    float acc_scale_pred;      // expected |a_new| / |a_old| over the next step (pm_set_predicted_acc_scale), 1 = none
    double last_acc_meanabs;   // mean |a_x| of the last gather's sample (pm_last_acc_meanabs)
    const float *sorted_pos;   // positions that were written tile by tile with table rb_cur (a hint: any table that
    int64_t sorted_np;         // partitions [0, np) gives correct results, an exact one saves the bisection)
    unsigned sorted_ntiles;
    int fr_march_gradient;     // pm_gradient_fr: marching kernel from 128^3 (default on)
    int64_t *sorted_table;     // ... and where those offsets are (inside rb_buf)
    int rb_cur;
    int mg_fused_sweep;  // 1 (default): single-pass red + black sweep for n >= 256 (pm_set_fused_sweep; tests compare)
    int fr_rows_sweep;    // the f(R) smoothers' row kernels for n >= 16 (pm_set_fr_rows_sweep; tests compare)
    int fr_fused_restrict;  // pm_fr_vcycle: operator + the three restrictions in one marching pass (pm_set_fr_fused_restrict)
    int mond_fast_nu;     // pm_mond_rhs, simple function: hardware rsq / sqrt (pm_set_mond_fast_nu; tests compare)
    int mg_fused_coarse;  // pm_lap_vcycle runs the levels from 16^3 down in one workgroup (pm_set_fused_coarse)
    void *mg_buf;        // scratch grids of pm_lap_vcycle (grown on demand)
    size_t mg_buf_bytes;
    // captured V-cycles (pm_lap_vcycle, pm_fr_vcycle): a cycle is 40-90 small launches, replayed as one hipGraph when the
    // same arrays come back (keyed by every argument and the arena)
    struct VcGraph {
        const void *x, *b, *arena;
        int kind, n, npre, npost;
        float q;
        hipGraphExec_t exec;
        unsigned long long last_use;
    } vc_graphs[8];
    int vc_graph_mode;           // 1: capture and replay, 0: plain launches (also after any capture failure)
    unsigned vc_hits, vc_misses;
    unsigned long long vc_clock;
    hipStream_t capture_stream;  // captures happen here (the caller's stream may be the legacy default stream)
    void *h_pinned;              // 16 KB of pinned host memory: landing zone of the small read-backs (pm_read_back)
    // sort workspace (grown on demand)
    void *sort_tmp;
    size_t sort_tmp_bytes;
    int64_t age64_count, age64_np, age64_passes;  // fp64 mode: what the last gather saw of the stream's age (pm_real64_stream_age)
};

static inline pm_geom pm_ctx_geom(const pm_ctx *c) {
    pm_geom g;
    g.n = c->n, g.log2n = c->log2n, g.mask = c->n - 1, g.nx = c->slab_nx, g.x0 = c->slab_x0;
    return g;
}

static inline bool pm_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// grid size for streaming kernels: enough blocks to fill 256 CUs x 8, grid-stride the rest
static inline int pm_stream_blocks(int64_t items, int per_block) {
    int64_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > 256 * 16) b = 256 * 16;
    return (int)b;
}
