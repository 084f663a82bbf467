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

struct pm_ctx {
    int n;          // cells per side
    int log2n;
    hipStream_t stream;
    // rocFFT
    rocfft_plan plan_r2c, plan_c2r;
    rocfft_execution_info info_r2c, info_c2r;
    void *fft_work;
    size_t fft_work_bytes;
    void *d_spec;  // lazily allocated half-spectrum scratch for pm_poisson_fft
    // small device scratch: [0..63] words for reductions / flags, then 3 tables of n floats
    unsigned *d_scratch;
    float *d_tab;      // 3*n floats (k2, s, spare)
    double *d_hist;    // 3*n doubles (P(k) bins)
    hipEvent_t ev0, ev1;
    // sort workspace (grown on demand)
    void *sort_tmp;
    size_t sort_tmp_bytes;
};

static inline bool pm_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// grid size for streaming kernels: enough blocks to fill 256 CUs x 8, grid-stride the rest
static inline int pm_stream_blocks(int64_t items, int per_block) {
    int64_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > 256 * 16) b = 256 * 16;
    return (int)b;
}
