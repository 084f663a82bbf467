// Context, error reporting, rocFFT plans (fourier.py:104-147, 251-294 replaced by persistent plans).
#include <cstring>
#include <mutex>

#include "pm_common.h"

static thread_local char g_err[512] = "";

void pm_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *pm_last_error(void) { return g_err; }
extern "C" int pm_version(void) { return 100; }

static std::once_flag g_fft_once;

void pm_fft_setup_once() {
    std::call_once(g_fft_once, [] { rocfft_setup(); });
}

static int build_plans(pm_ctx *c) {
    if (c->plan_r2c) return PM_OK;
    pm_fft_setup_once();
    const size_t len[3] = {(size_t)c->n, (size_t)c->n, (size_t)c->n};  // fastest first
    PM_FFT(rocfft_plan_create(&c->plan_r2c, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                              rocfft_precision_single, 3, len, 1, nullptr));
    rocfft_plan_description d = nullptr;
    PM_FFT(rocfft_plan_description_create(&d));
    // ifft_3D_real carries 1/n^3 (PyFFTW normalise_idft / np.fft.irfftn, fourier.py:277)
    PM_FFT(rocfft_plan_description_set_scale_factor(d, 1.0 / ((double)c->n * c->n * c->n)));
    PM_FFT(rocfft_plan_create(&c->plan_c2r, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                              rocfft_precision_single, 3, len, 1, d));
    PM_FFT(rocfft_plan_description_destroy(d));
    size_t w1 = 0, w2 = 0;
    PM_FFT(rocfft_plan_get_work_buffer_size(c->plan_r2c, &w1));
    PM_FFT(rocfft_plan_get_work_buffer_size(c->plan_c2r, &w2));
    c->fft_work_bytes = w1 > w2 ? w1 : w2;
    if (c->fft_work_bytes) PM_HIP(hipMalloc(&c->fft_work, c->fft_work_bytes));
    PM_FFT(rocfft_execution_info_create(&c->info_r2c));
    PM_FFT(rocfft_execution_info_create(&c->info_c2r));
    if (c->fft_work_bytes) {
        PM_FFT(rocfft_execution_info_set_work_buffer(c->info_r2c, c->fft_work, c->fft_work_bytes));
        PM_FFT(rocfft_execution_info_set_work_buffer(c->info_c2r, c->fft_work, c->fft_work_bytes));
    }
    PM_FFT(rocfft_execution_info_set_stream(c->info_r2c, c->stream));
    PM_FFT(rocfft_execution_info_set_stream(c->info_c2r, c->stream));
    return PM_OK;
}

// SURVEY.md section 8(b)'s pm_create(N, Np, ngpu, real_bytes): the arithmetic type of a context.  4: the fp32 mode (the
// reference's precision, every entry point of pmstep.h but the pm_real64_* ones); 8: the fp64 device mode of the
// Newtonian FFT step (pm_real64_*: arrays of double).
extern "C" int pm_ctx_create_real(pm_ctx **out, int n, void *stream, int real_bytes) {
    PM_REQUIRE(real_bytes == 4 || real_bytes == 8, "real_bytes must be 4 (float) or 8 (double)");
    int rc = pm_ctx_create(out, n, stream);
    if (rc == PM_OK) (*out)->real_bytes = real_bytes;
    return rc;
}

extern "C" int pm_ctx_real_bytes(pm_ctx *c) { return c ? c->real_bytes : 0; }

extern "C" int pm_ctx_create(pm_ctx **out, int n, void *stream) {
    PM_REQUIRE(out != nullptr, "out is NULL");
    PM_REQUIRE(n >= 4 && n <= 16384 && (n & (n - 1)) == 0, "n must be a power of two in [4, 16384]");
    pm_ctx *c = new pm_ctx();
    memset(c, 0, sizeof(*c));
    c->acc_scale_pred = 1.0f;
    c->real_bytes = 4;
    c->age64_np = -1;
    c->vc_graph_mode = 1, c->mg_fused_coarse = 1, c->fr_rows_sweep = 1, c->fr_march_gradient = 1, c->mond_fast_nu = 1, c->fr_fused_restrict = 1;
    if (hipHostMalloc(&c->h_pinned, PM_PINNED_BYTES, hipHostMallocDefault) != hipSuccess) c->h_pinned = nullptr, (void)hipGetLastError();
    c->n = n;
    c->log2n = __builtin_ctz((unsigned)n);
    c->slab_nx = n;
    c->slab_x0 = 0;
    c->gather_mode = 1;
    c->mg_fused_sweep = 1;
    c->tile_x_lo = 0, c->tile_nx = n;
    c->last_stragglers = -1, c->last_np = 0, c->last_complete = 1;
    c->stream = (hipStream_t)stream;
    PM_HIP(hipMalloc((void **)&c->d_scratch, PM_SCRATCH_WORDS * sizeof(unsigned)));
    PM_HIP(hipMemset(c->d_scratch, 0, PM_SCRATCH_WORDS * sizeof(unsigned)));
    PM_HIP(hipMalloc((void **)&c->d_tab, 3 * (size_t)n * sizeof(float)));
    PM_HIP(hipMalloc((void **)&c->d_hist, 3 * (size_t)n * sizeof(double)));
    PM_HIP(hipEventCreate(&c->ev0));
    PM_HIP(hipEventCreate(&c->ev1));
    *out = c;
    return PM_OK;
}

extern "C" int pm_ctx_destroy(pm_ctx *c) {
    if (!c) return PM_OK;
    hipStreamSynchronize(c->stream);
    if (c->plan_r2c) rocfft_plan_destroy(c->plan_r2c);
    if (c->plan_c2r) rocfft_plan_destroy(c->plan_c2r);
    if (c->info_r2c) rocfft_execution_info_destroy(c->info_r2c);
    if (c->info_c2r) rocfft_execution_info_destroy(c->info_c2r);
    if (c->fft_work) hipFree(c->fft_work);
    if (c->pad_r2c) rocfft_plan_destroy(c->pad_r2c);
    if (c->pad_c2r) rocfft_plan_destroy(c->pad_c2r);
    if (c->pad_info_r2c) rocfft_execution_info_destroy(c->pad_info_r2c);
    if (c->pad_info_c2r) rocfft_execution_info_destroy(c->pad_info_c2r);
    if (c->pad_work) hipFree(c->pad_work);
    if (c->d_spec) hipFree(c->d_spec);
    if (c->plan64_r2c) rocfft_plan_destroy(c->plan64_r2c);
    if (c->plan64_c2r) rocfft_plan_destroy(c->plan64_c2r);
    if (c->info64_r2c) rocfft_execution_info_destroy(c->info64_r2c);
    if (c->info64_c2r) rocfft_execution_info_destroy(c->info64_c2r);
    if (c->fft64_work) hipFree(c->fft64_work);
    if (c->d_spec64) hipFree(c->d_spec64);
    if (c->d_tab64) hipFree(c->d_tab64);
    if (c->fp_B) hipFree(c->fp_B);
    if (c->fp_C) hipFree(c->fp_C);
    if (c->fp_tw) hipFree(c->fp_tw);
    delete[] c->h_pow;
    for (int i = 0; i < 4; ++i) {
        if (c->plan_s[i]) rocfft_plan_destroy(c->plan_s[i]);
        if (c->info_s[i]) rocfft_execution_info_destroy(c->info_s[i]);
        if (c->work_s[i]) hipFree(c->work_s[i]);
    }
    if (c->sort_tmp) hipFree(c->sort_tmp);
    if (c->tile_tmp) hipFree(c->tile_tmp);
    if (c->list_buf) hipFree(c->list_buf);
    if (c->rb_buf) hipFree(c->rb_buf);
    if (c->kr_buf) hipFree(c->kr_buf);
    if (c->mg_buf) hipFree(c->mg_buf);
    for (auto &g : c->vc_graphs)
        if (g.exec) hipGraphExecDestroy(g.exec);
    if (c->capture_stream) hipStreamDestroy(c->capture_stream);
    if (c->h_pinned) hipHostFree(c->h_pinned);
    hipFree(c->d_scratch);
    hipFree(c->d_tab);
    hipFree(c->d_hist);
    hipEventDestroy(c->ev0);
    hipEventDestroy(c->ev1);
    delete c;
    return PM_OK;
}

extern "C" int pm_ctx_set_stream(pm_ctx *c, void *stream) {
    PM_REQUIRE(c != nullptr, "ctx is NULL");
    c->stream = (hipStream_t)stream;
    if (c->info_r2c) {
        PM_FFT(rocfft_execution_info_set_stream(c->info_r2c, c->stream));
        PM_FFT(rocfft_execution_info_set_stream(c->info_c2r, c->stream));
    }
    if (c->pad_info_r2c) {
        PM_FFT(rocfft_execution_info_set_stream(c->pad_info_r2c, c->stream));
        PM_FFT(rocfft_execution_info_set_stream(c->pad_info_c2r, c->stream));
    }
    for (int i = 0; i < 4; ++i)
        if (c->info_s[i]) PM_FFT(rocfft_execution_info_set_stream(c->info_s[i], c->stream));
    return PM_OK;
}

int pm_read_back(pm_ctx *c, void *h_dst, const void *d_src, size_t bytes) {
    void *land = (c->h_pinned && bytes <= PM_PINNED_BYTES) ? c->h_pinned : h_dst;
    PM_HIP(hipMemcpyAsync(land, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    PM_HIP(hipStreamSynchronize(c->stream));
    if (land != h_dst) memcpy(h_dst, land, bytes);
    return PM_OK;
}

extern "C" int pm_set_slab(pm_ctx *c, int nx, int x0) {
    PM_REQUIRE(c != nullptr, "ctx is NULL");
    PM_REQUIRE(nx >= 1 && nx <= c->n + 16 && x0 >= -c->n && x0 < c->n, "bad slab geometry");
    c->slab_nx = nx;
    c->slab_x0 = ((x0 % c->n) + c->n) % c->n;
    return PM_OK;
}

extern "C" int pm_set_gather_mode(pm_ctx *c, int mode) {
    PM_REQUIRE(c != nullptr && (mode == 1 || mode == 2), "mode must be 1 (tile ranges) or 2 (chunk/anchor)");
    c->gather_mode = mode;
    return PM_OK;
}

extern "C" int pm_ctx_sync(pm_ctx *c) {
    PM_REQUIRE(c != nullptr, "ctx is NULL");
    PM_HIP(hipStreamSynchronize(c->stream));
    return PM_OK;
}

extern "C" int pm_timer_start(pm_ctx *c) {
    PM_REQUIRE(c != nullptr, "ctx is NULL");
    PM_HIP(hipEventRecord(c->ev0, c->stream));
    return PM_OK;
}

extern "C" int pm_timer_stop_ms(pm_ctx *c, float *h_ms) {
    PM_REQUIRE(c != nullptr && h_ms != nullptr, "NULL argument");
    PM_HIP(hipEventRecord(c->ev1, c->stream));
    PM_HIP(hipEventSynchronize(c->ev1));
    PM_HIP(hipEventElapsedTime(h_ms, c->ev0, c->ev1));
    return PM_OK;
}

extern "C" int pm_fft_r2c(pm_ctx *c, const float *d_x, void *d_xk) {
    PM_REQUIRE(c && d_x && d_xk, "NULL argument");
    int rc = build_plans(c);
    if (rc) return rc;
    void *in[1] = {const_cast<float *>(d_x)};
    void *out[1] = {d_xk};
    PM_FFT(rocfft_execute(c->plan_r2c, in, out, c->info_r2c));
    return PM_OK;
}

extern "C" int pm_fft_c2r(pm_ctx *c, void *d_xk, float *d_x) {
    PM_REQUIRE(c && d_x && d_xk, "NULL argument");
    int rc = build_plans(c);
    if (rc) return rc;
    void *in[1] = {d_xk};
    void *out[1] = {d_x};
    PM_FFT(rocfft_execute(c->plan_c2r, in, out, c->info_c2r));
    return PM_OK;
}

// The pair of plans on the padded grid of Orszag's 3/2 rule (initial_conditions.py:1859-1928: m = 3 n / 2, not a power
// of two -- rocFFT takes any length; the hand-written LDS FFT passes are radix-2 and stay on the mesh sizes).
static int build_padded_plans(pm_ctx *c) {
    if (c->pad_r2c) return PM_OK;
    pm_fft_setup_once();
    const size_t m = (size_t)c->n * 3 / 2;
    const size_t len[3] = {m, m, m};
    PM_FFT(rocfft_plan_create(&c->pad_r2c, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                              rocfft_precision_single, 3, len, 1, nullptr));
    rocfft_plan_description d = nullptr;
    PM_FFT(rocfft_plan_description_create(&d));
    PM_FFT(rocfft_plan_description_set_scale_factor(d, 1.0 / ((double)m * m * m)));  // ifft_3D_real's 1/m^3
    PM_FFT(rocfft_plan_create(&c->pad_c2r, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                              rocfft_precision_single, 3, len, 1, d));
    PM_FFT(rocfft_plan_description_destroy(d));
    size_t w1 = 0, w2 = 0;
    PM_FFT(rocfft_plan_get_work_buffer_size(c->pad_r2c, &w1));
    PM_FFT(rocfft_plan_get_work_buffer_size(c->pad_c2r, &w2));
    const size_t w = w1 > w2 ? w1 : w2;
    if (w) PM_HIP(hipMalloc(&c->pad_work, w));
    PM_FFT(rocfft_execution_info_create(&c->pad_info_r2c));
    PM_FFT(rocfft_execution_info_create(&c->pad_info_c2r));
    if (w) {
        PM_FFT(rocfft_execution_info_set_work_buffer(c->pad_info_r2c, c->pad_work, w));
        PM_FFT(rocfft_execution_info_set_work_buffer(c->pad_info_c2r, c->pad_work, w));
    }
    PM_FFT(rocfft_execution_info_set_stream(c->pad_info_r2c, c->stream));
    PM_FFT(rocfft_execution_info_set_stream(c->pad_info_c2r, c->stream));
    return PM_OK;
}

extern "C" int pm_fft_r2c_padded(pm_ctx *c, const float *d_x, void *d_xk) {
    PM_REQUIRE(c && d_x && d_xk, "NULL argument");
    int rc = build_padded_plans(c);
    if (rc) return rc;
    void *in[1] = {const_cast<float *>(d_x)};
    void *out[1] = {d_xk};
    PM_FFT(rocfft_execute(c->pad_r2c, in, out, c->pad_info_r2c));
    return PM_OK;
}

extern "C" int pm_fft_c2r_padded(pm_ctx *c, void *d_xk, float *d_x) {
    PM_REQUIRE(c && d_x && d_xk, "NULL argument");
    int rc = build_padded_plans(c);
    if (rc) return rc;
    void *in[1] = {d_xk};
    void *out[1] = {d_x};
    PM_FFT(rocfft_execute(c->pad_c2r, in, out, c->pad_info_c2r));
    return PM_OK;
}

extern "C" int pm_poisson_fft(pm_ctx *c, int kind, int p, const float *d_rhs, float *d_phi, const float *h_k2,
                              const float *h_s, float c0) {
    PM_REQUIRE(c && d_rhs && d_phi, "NULL argument");
    const size_t nspec = (size_t)c->n * c->n * (c->n / 2 + 1);
    if (!c->d_spec) PM_HIP(hipMalloc(&c->d_spec, nspec * 2 * sizeof(float)));
    int rc = pm_fft_r2c(c, d_rhs, c->d_spec);
    if (rc) return rc;
    rc = pm_greens(c, kind, p, c->d_spec, h_k2, h_s, c0);
    if (rc) return rc;
    return pm_fft_c2r(c, c->d_spec, d_phi);
}

// ------------------------------------------------------------------------------------------------
// Slab-decomposed FFT pieces.  A rank owns nb x-planes: (1) batched 2-D r2c over (y,z) on its planes,
// (2) all-to-all transpose done by the caller (RCCL), after which the rank holds all n x-planes of
// m = (n/G)*(n/2+1) (y,kz) columns, (3) strided 1-D c2c along x.  The inverse mirrors it; the 1/n^3
// of ifft_3D_real is folded into the 2-D c2r plan.

static int slab_plan(pm_ctx *c, int which, long long key) {
    if (c->plan_s[which] && c->key_s[which] == key) return PM_OK;
    pm_fft_setup_once();
    if (c->plan_s[which]) {
        PM_HIP(hipStreamSynchronize(c->stream));
        rocfft_plan_destroy(c->plan_s[which]);
        c->plan_s[which] = nullptr;
    }
    if (c->info_s[which]) {
        rocfft_execution_info_destroy(c->info_s[which]);
        c->info_s[which] = nullptr;
    }
    if (c->work_s[which]) {
        PM_HIP(hipFree(c->work_s[which]));
        c->work_s[which] = nullptr;
    }
    const size_t n = (size_t)c->n;
    rocfft_plan_description d = nullptr;
    PM_FFT(rocfft_plan_description_create(&d));
    if (which == 0 || which == 1) {
        const size_t len[2] = {n, n};  // z fastest, then y
        if (which == 1) PM_FFT(rocfft_plan_description_set_scale_factor(d, 1.0 / ((double)n * n * n)));
        PM_FFT(rocfft_plan_create(&c->plan_s[which], rocfft_placement_notinplace,
                                  which == 0 ? rocfft_transform_type_real_forward : rocfft_transform_type_real_inverse,
                                  rocfft_precision_single, 2, len, (size_t)key, d));
    } else {
        const size_t len[1] = {n};
        const size_t stride[1] = {(size_t)key};
        PM_FFT(rocfft_plan_description_set_data_layout(d, rocfft_array_type_complex_interleaved,
                                                       rocfft_array_type_complex_interleaved, nullptr, nullptr, 1,
                                                       stride, 1, 1, stride, 1));
        PM_FFT(rocfft_plan_create(&c->plan_s[which], rocfft_placement_inplace,
                                  which == 2 ? rocfft_transform_type_complex_forward
                                             : rocfft_transform_type_complex_inverse,
                                  rocfft_precision_single, 1, len, (size_t)key, d));
    }
    PM_FFT(rocfft_plan_description_destroy(d));
    size_t w = 0;
    PM_FFT(rocfft_plan_get_work_buffer_size(c->plan_s[which], &w));
    PM_FFT(rocfft_execution_info_create(&c->info_s[which]));
    if (w) {
        PM_HIP(hipMalloc(&c->work_s[which], w));
        PM_FFT(rocfft_execution_info_set_work_buffer(c->info_s[which], c->work_s[which], w));
    }
    PM_FFT(rocfft_execution_info_set_stream(c->info_s[which], c->stream));
    c->key_s[which] = key;
    return PM_OK;
}

extern "C" int pm_fft2d_r2c(pm_ctx *c, const float *d_x, void *d_xk, int nplanes) {
    PM_REQUIRE(c && d_x && d_xk && nplanes >= 1, "bad argument");
    int rc = slab_plan(c, 0, nplanes);
    if (rc) return rc;
    void *in[1] = {const_cast<float *>(d_x)};
    void *out[1] = {d_xk};
    PM_FFT(rocfft_execute(c->plan_s[0], in, out, c->info_s[0]));
    return PM_OK;
}

extern "C" int pm_fft2d_c2r(pm_ctx *c, void *d_xk, float *d_x, int nplanes) {
    PM_REQUIRE(c && d_x && d_xk && nplanes >= 1, "bad argument");
    int rc = slab_plan(c, 1, nplanes);
    if (rc) return rc;
    void *in[1] = {d_xk};
    void *out[1] = {d_x};
    PM_FFT(rocfft_execute(c->plan_s[1], in, out, c->info_s[1]));
    return PM_OK;
}

extern "C" int pm_fft1d_x(pm_ctx *c, void *d_data, int64_t ncolumns, int inverse) {
    PM_REQUIRE(c && d_data && ncolumns >= 1, "bad argument");
    const int which = inverse ? 3 : 2;
    int rc = slab_plan(c, which, ncolumns);
    if (rc) return rc;
    void *in[1] = {d_data};
    PM_FFT(rocfft_execute(c->plan_s[which], in, nullptr, c->info_s[which]));
    return PM_OK;
}
