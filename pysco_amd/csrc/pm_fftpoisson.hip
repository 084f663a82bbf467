// Fused FFT Poisson solve: phi = IFFT( G(k) * FFT(rho) ) in five passes over HBM, no rocFFT.
//
// rocFFT's 3-D r2c + Green's function + c2r costs 7 + 1 + 7 kernels at 1024^3 (25 ms), more than half
// of it in stand-alone transposes.  Here every pass is a batched 1-D FFT done entirely in LDS whose
// global side is contiguous on one end and transposed (64/128-byte segments) on the other, so the
// array is never transposed on its own, and the x pass does forward FFT -> Green's function ->
// inverse FFT on each line while it sits in LDS:
//
//   P1  rho[x][y][z] (real)   --z r2c-->            B[x/8][kz][x%8][y] two real lines per complex FFT
//   P2  B                     --y forward-->        C[ky][kz][x]  (single GPU: C[kz][ky][x])
//   P3  C[ky][kz][x]          --x forward, *G(k)/n^3, x inverse-->  C   (in place, contiguous lines)
//   P4  C                     --y inverse-->        B
//   P5  B                     --z c2r-->            phi[x][y][z] (real)
// Layouts are chosen for the locality of the transposed (scattered) side of each pass -- measured: the
// 64/128-byte pieces of one workgroup a whole plane apart cost 15-20 % of the pass against a few KB apart.
// B is blocked by 8 x-planes, so the 8 lines of a y-pass workgroup are one contiguous 64 KB and the pieces a
// z-pass workgroup scatters over kz are 64 KB apart; on a single GPU C is kz-major, so the pieces a y-pass
// workgroup scatters over ky are one x-line apart (the slab entry points keep C[ky][kz][x_local], whose
// ky blocks are what the all-to-all ships).
//
// 1-D FFTs: one wavefront per line, Stockham autosort stages of radix 16 / 8 / 4 with the butterfly in
// registers (1024 = 16 * 16 * 4: a lane holds 16 points) and the line exchanged through LDS between
// stages.  Natural order in and out (no digit reversal, no permutation tables), no barrier inside a
// stage: the lanes of a wave read all their inputs before the first output is written.
// Replaces solver.fft (solver.py:453-522 without the P(k) branch) + fourier.py:104-147, 251-294, 460-595.
#include "pm_common.h"
#include <type_traits>

#define FBLK 512  // 8 wavefronts = 8 lines

// Complex arithmetic on 2-vectors: one packed instruction per complex add, three per multiply, one per
// multiplication by +-i.  The half swaps and single-half sign flips ride on the VOP3P op_sel / neg
// modifiers; the compiler folds a swap or a negation of BOTH halves by itself but builds (-x, y) with a
// v_xor and a v_mov, and left to its SLP vectoriser it pairs halves of different values and shuffles
// them back (a third of a 1024-point transform's VALU instructions) -- hence four one-line asm helpers.
typedef float cv2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cv2 cv(float2 a) { return cv2{a.x, a.y}; }
__device__ __forceinline__ float2 cf(cv2 a) { return make_float2(a.x, a.y); }
#define FFTP_PK_ADD(name, mods)                                                          \
    __device__ __forceinline__ cv2 name(cv2 a, cv2 b) {                                   \
        cv2 r;                                                                            \
        asm("v_pk_add_f32 %0, %1, %2 " mods : "=v"(r) : "v"(a), "v"(b));                   \
        return r;                                                                         \
    }
FFTP_PK_ADD(add_neglo, "neg_lo:[0,1]")                                  // a + (-b.x,  b.y)
FFTP_PK_ADD(add_neghi, "neg_hi:[0,1]")                                  // a + ( b.x, -b.y)
FFTP_PK_ADD(add_swap_neglo, "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")  // a + (-b.y,  b.x) = a + i b
FFTP_PK_ADD(add_swap_neghi, "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")  // a + ( b.y, -b.x) = a - i b
#undef FFTP_PK_ADD
// a b = (a.x b.x - a.y b.y, a.x b.y + a.y b.x): a packed multiply and a packed fma (the transform is held to a
// tolerance, not to the bits of a CPU FFT, so the fused rounding is welcome here; the build's -ffp-contract=off
// protects the kernels that reproduce the reference's fp32 arithmetic)
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    const cv2 t = cv2{a.x, a.x} * cv(b);
    cv2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(cv(a)), "v"(cv(b)), "v"(t));
    return cf(r);  // (a.y * -b.y + t.x, a.y * b.x + t.y)
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return cf(cv(a) + cv(b)); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return cf(cv(a) - cv(b)); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return cf(add_swap_neghi(cv2{0.f, 0.f}, cv(a))); }  // a * (-i)
__device__ __forceinline__ float2 mul_pi(float2 a) { return cf(add_swap_neglo(cv2{0.f, 0.f}, cv(a))); }  // a * (+i)

// LDS addressing: element e of a line lives at PADI(e) = e + e/16 and lines are NP(n) = n + n/16 + 4 apart.
// The extra slot per 16 elements spreads the stride-16 (and stride-8) scatter of a stage's outputs over
// the banks (16 lanes x 128 B apart would be a 16-way conflict); the 4 extra elements per line put the
// 8 lines 8 banks apart, so the transposed passes' (line, frequency) accesses -- 8 lines x 4 consecutive
// frequencies per 32-lane group -- hit 32 different bank pairs instead of 4.
#define PADI(e) ((e) + ((e) >> 4))
#define NP(n) ((n) + ((n) >> 4) + 4)

// R-point DFTs in registers, natural order in and out.  INV: conjugate transform (unnormalised).
template <bool INV>
__device__ __forceinline__ float2 mul_j(float2 a) {  // a * (-i) forward, a * (+i) inverse
    return INV ? mul_pi(a) : mul_mi(a);
}
template <bool INV>
__device__ __forceinline__ void dft4(float2 &a0, float2 &a1, float2 &a2, float2 &a3) {
    const cv2 s02 = cv(a0) + cv(a2), d02 = cv(a0) - cv(a2), s13 = cv(a1) + cv(a3), d = cv(a1) - cv(a3);
    // a1 = d02 + j d, a3 = d02 - j d with j = -i (forward) or +i (inverse)
    a0 = cf(s02 + s13), a2 = cf(s02 - s13);
    a1 = cf(INV ? add_swap_neglo(d02, d) : add_swap_neghi(d02, d));
    a3 = cf(INV ? add_swap_neghi(d02, d) : add_swap_neglo(d02, d));
}
template <bool INV>
__device__ __forceinline__ float2 cmul_w(float2 a, float c, float sn) {  // a * exp(-/+ i theta), (c, sn) = (cos, sin) theta
    // forward: (a.x c + a.y sn, a.y c - a.x sn); inverse: (a.x c - a.y sn, a.y c + a.x sn)
    const cv2 t = cv(a) * cv2{c, c}, u = cv2{a.y, a.x} * cv2{sn, sn};
    return cf(INV ? add_neglo(t, u) : add_neghi(t, u));
}
template <int R, bool INV>
struct Dft;
template <bool INV>
struct Dft<4, INV> {
    static __device__ __forceinline__ void run(float2 *a) { dft4<INV>(a[0], a[1], a[2], a[3]); }
};
template <bool INV>
struct Dft<8, INV> {
    static __device__ __forceinline__ void run(float2 *a) {
        // even / odd subsequences, then X[k] = E[k] + w8^k O[k], X[k+4] = E[k] - w8^k O[k]
        dft4<INV>(a[0], a[2], a[4], a[6]);
        dft4<INV>(a[1], a[3], a[5], a[7]);
        const float h = 0.70710678118654752f;
        const float2 o0 = a[1], o1 = cmul_w<INV>(a[3], h, h), o2 = mul_j<INV>(a[5]), o3 = cmul_w<INV>(a[7], -h, h);
        const float2 e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6];
        a[0] = cadd(e0, o0), a[1] = cadd(e1, o1), a[2] = cadd(e2, o2), a[3] = cadd(e3, o3);
        a[4] = csub(e0, o0), a[5] = csub(e1, o1), a[6] = csub(e2, o2), a[7] = csub(e3, o3);
    }
};
template <bool INV>
struct Dft<16, INV> {
    static __device__ __forceinline__ void run(float2 *a) {
        // m = m0 + 4 m1, k = k1 + 4 k2:  Y[m0][k1] = DFT4 over m1;  X[k1 + 4 k2] = DFT4 over m0 of w16^(m0 k1) Y
#pragma unroll
        for (int m0 = 0; m0 < 4; ++m0) dft4<INV>(a[m0], a[m0 + 4], a[m0 + 8], a[m0 + 12]);  // a[m0 + 4 k1] = Y[m0][k1]
        const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
        a[5] = cmul_w<INV>(a[5], c1, s1);    // w16^1
        a[6] = cmul_w<INV>(a[6], h, h);      // w16^2
        a[7] = cmul_w<INV>(a[7], s1, c1);    // w16^3
        a[9] = cmul_w<INV>(a[9], h, h);      // w16^2
        a[10] = mul_j<INV>(a[10]);           // w16^4
        a[11] = cmul_w<INV>(a[11], -h, h);   // w16^6
        a[13] = cmul_w<INV>(a[13], s1, c1);  // w16^3
        a[14] = cmul_w<INV>(a[14], -h, h);   // w16^6
        a[15] = cmul_w<INV>(a[15], -c1, -s1);  // w16^9
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) dft4<INV>(a[4 * k1], a[4 * k1 + 1], a[4 * k1 + 2], a[4 * k1 + 3]);  // a[k2 + 4 k1]
        // natural order: X[k1 + 4 k2] sits in a[k2 + 4 k1] (a 4 x 4 transpose of register names)
        float2 t;
#define FFTP_SWAP(i, j) t = a[i], a[i] = a[j], a[j] = t
        FFTP_SWAP(1, 4), FFTP_SWAP(2, 8), FFTP_SWAP(3, 12), FFTP_SWAP(6, 9), FFTP_SWAP(7, 13), FFTP_SWAP(11, 14);
#undef FFTP_SWAP
    }
};

// One Stockham stage of radix R on the line `ln` (padded, LDS), executed by one wavefront: butterfly j
// reads in[j + t n/R], multiplies by w^(t (j mod Ns)) (w = exp(-+2 pi i / (Ns R))), does the R-point DFT and
// writes out[(j / Ns) Ns R + (j mod Ns) + t Ns].  BPL = butterflies per lane (n / R / 64, at least 1).
// tw[k] = exp(-2 pi i k / n).
// (in three pieces, so that a pass can feed a stage from global memory or registers and take its outputs there:
// fftp_x_green's fused form)
template <int R, int BPL>
__device__ __forceinline__ void stage_load(float2 (&a)[BPL][R], const float2 *ln, int n) {
    const int lane = threadIdx.x & 63;
    const int nb = n / R;              // butterflies in the line
#pragma unroll
    for (int i = 0; i < BPL; ++i) {
        const int j = lane + 64 * i;
        if (j < nb) {
            // PADI(j + t nb) = PADI(j) + t (nb + nb/16) when nb is a multiple of 16: one add per element
            // instead of the shift-add-add of PADI (a third of the stage's VALU work was address arithmetic)
            if (nb >= 16) {
                const float2 *src = ln + PADI(j);
                const int rs = nb + (nb >> 4);
#pragma unroll
                for (int t = 0; t < R; ++t) a[i][t] = src[t * rs];
            } else {
#pragma unroll
                for (int t = 0; t < R; ++t) a[i][t] = ln[PADI(j + t * nb)];
            }
        }
    }
}

template <int R, int BPL, bool INV>
__device__ __forceinline__ void stage_compute(float2 (&a)[BPL][R], const float2 *tw, int n, int lNs) {
    const int lane = threadIdx.x & 63;
    const int nb = n / R;
    const int Ns = 1 << lNs;
    const int tstep = n >> (lNs + (R == 16 ? 4 : (R == 8 ? 3 : 2)));  // n / (Ns R)
#pragma unroll
    for (int i = 0; i < BPL; ++i) {
        const int j = lane + 64 * i;
        if (j < nb) {
            if (lNs > 0) {  // twiddles w^t from one table entry: powers by repeated squaring / products
                const int r = j & (Ns - 1);
                float2 w1 = tw[r * tstep];
                if (INV) w1 = cconj(w1);
                float2 w[R];
                w[1] = w1;
#pragma unroll
                for (int t = 2; t < R; ++t) w[t] = (t & 1) ? cmul(w[t - 1], w1) : cmul(w[t / 2], w[t / 2]);
#pragma unroll
                for (int t = 1; t < R; ++t) a[i][t] = cmul(a[i][t], w[t]);
            }
            Dft<R, INV>::run(a[i]);
        }
    }
}

template <int R, int BPL>
__device__ __forceinline__ void stage_store(const float2 (&a)[BPL][R], float2 *ln, int n, int lNs) {
    const int lane = threadIdx.x & 63;
    const int nb = n / R;
    const int Ns = 1 << lNs;
#pragma unroll
    for (int i = 0; i < BPL; ++i) {
        const int j = lane + 64 * i;
        if (j < nb) {
            const int base = ((j >> lNs) << (lNs + (R == 16 ? 4 : (R == 8 ? 3 : 2)))) + (j & (Ns - 1));
            if (lNs == 0 || lNs >= 4) {
                // Ns = 1: base is a multiple of R <= 16, so base + t stays in base's 16-group; Ns a multiple
                // of 16: PADI(base + t Ns) = PADI(base) + t (Ns + Ns/16).  Either way immediate offsets.
                float2 *dst = ln + PADI(base);
                const int ws = lNs == 0 ? 1 : Ns + (Ns >> 4);
#pragma unroll
                for (int t = 0; t < R; ++t) dst[t * ws] = a[i][t];
            } else {
#pragma unroll
                for (int t = 0; t < R; ++t) ln[PADI(base + (t << lNs))] = a[i][t];
            }
        }
    }
}

template <int R, int BPL, bool INV>
__device__ __forceinline__ void stockham_stage(float2 *ln, const float2 *tw, int n, int log2n, int lNs) {
    float2 a[BPL][R];
    stage_load<R, BPL>(a, ln, n);
    stage_compute<R, BPL, INV>(a, tw, n, lNs);
    stage_store<R, BPL>(a, ln, n, lNs);
}

// forward / inverse FFT of the 8 lines of the workgroup (wave w owns line w); barriers only publish a
// stage's outputs (the next stage of the same wave reads what other lanes of that wave wrote)
// Orders a wavefront's LDS writes before its later LDS reads (the hardware executes one wave's LDS
// operations in order; this only stops the compiler from moving them) without stopping the other waves.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// WAVE_ONLY: the line was loaded by, and will be stored by, its own wave (contiguous lines): no workgroup
// barrier at all, so the 8 waves drift apart and one wave's loads overlap another's butterflies.
template <bool INV, bool WAVE_ONLY = false>
__device__ __forceinline__ void fft_lines_lds(float2 *buf, const float2 *tw, int n, int log2n) {
#define FFTP_SYNC() (WAVE_ONLY ? wave_lds_fence() : __syncthreads())
    float2 *ln = buf + (threadIdx.x >> 6) * NP(n);
    switch (log2n) {
        case 10:  // 16 * 16 * 4
            stockham_stage<16, 1, INV>(ln, tw, n, log2n, 0);
            FFTP_SYNC();
            stockham_stage<16, 1, INV>(ln, tw, n, log2n, 4);
            FFTP_SYNC();
            stockham_stage<4, 4, INV>(ln, tw, n, log2n, 8);
            break;
        case 9:  // 8 * 8 * 8
            stockham_stage<8, 1, INV>(ln, tw, n, log2n, 0);
            FFTP_SYNC();
            stockham_stage<8, 1, INV>(ln, tw, n, log2n, 3);
            FFTP_SYNC();
            stockham_stage<8, 1, INV>(ln, tw, n, log2n, 6);
            break;
        case 8:  // 16 * 16
            stockham_stage<16, 1, INV>(ln, tw, n, log2n, 0);
            FFTP_SYNC();
            stockham_stage<16, 1, INV>(ln, tw, n, log2n, 4);
            break;
        case 7:  // 8 * 16
            stockham_stage<8, 1, INV>(ln, tw, n, log2n, 0);
            FFTP_SYNC();
            stockham_stage<16, 1, INV>(ln, tw, n, log2n, 3);
            break;
        case 6:  // 8 * 8
            stockham_stage<8, 1, INV>(ln, tw, n, log2n, 0);
            FFTP_SYNC();
            stockham_stage<8, 1, INV>(ln, tw, n, log2n, 3);
            break;
        default:  // 32 = 8 * 4
            stockham_stage<8, 1, INV>(ln, tw, n, log2n, 0);
            FFTP_SYNC();
            stockham_stage<4, 1, INV>(ln, tw, n, log2n, 3);
            break;
    }
    FFTP_SYNC();
#undef FFTP_SYNC
}
__device__ __forceinline__ void fft_forward_lds(float2 *buf, const float2 *tw, int n, int log2n, int) {
    fft_lines_lds<false>(buf, tw, n, log2n);
}
__device__ __forceinline__ void fft_inverse_lds(float2 *buf, const float2 *tw, int n, int log2n, int) {
    fft_lines_lds<true>(buf, tw, n, log2n);
}

// Global -> LDS loops.  Written as `for (...) lds[..] = global[..]` the compiler keeps the loop rolled and
// waits for every load before the next is issued (vmcnt(0) per iteration): a thread of an n = 1024 pass
// then pays eight memory latencies in a row and the passes ran at 2.5-4.3 TB/s.  `batched` issues up to U
// loads first and only then consumes them.  Loads past the end are clamped to the last item (and their
// result dropped) rather than guarded: behind a guard the compiler sinks a load down to its use again.
typedef float fft_v4f __attribute__((ext_vector_type(4)));  // (an array of HIP float4 is not promoted to registers)
typedef float fft_v2f __attribute__((ext_vector_type(2)));

template <int U, typename Load, typename Use>
__device__ __forceinline__ void batched(int first, int total, int stride, Load load, Use use) {
    for (int t0 = first; t0 < total; t0 += U * stride) {
        decltype(load(0)) v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = load(min(t0 + u * stride, total - 1));
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (t0 + u * stride < total) use(t0 + u * stride, v[u]);
    }
}

// One batch, no loop: for trip counts known to fit (total <= U * stride).  Before a LOOP the compiler drains
// vmcnt whenever registers loaded earlier stay live across it, which would serialise those loads with this batch.
template <int U, typename Load, typename Use>
__device__ __forceinline__ void batched_once(int first, int total, int stride, Load load, Use use) {
    if (first < total) {
        decltype(load(0)) v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = load(min(first + u * stride, total - 1));
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (first + u * stride < total) use(first + u * stride, v[u]);
    }
}

__device__ __forceinline__ void load_twiddles(float2 *tw, const float2 *__restrict__ g_tw, int n, int) {
    batched<2>(threadIdx.x, n, FBLK, [&](int t) { return *reinterpret_cast<const fft_v2f *>(g_tw + t); },
               [&](int t, fft_v2f v) { tw[t] = make_float2(v.x, v.y); });
}

// 8 contiguous complex lines <-> LDS: 16-byte global accesses, two 8-byte LDS accesses (a pair
// (e, e+1) with e even never straddles a 16-group, but its padded address may be odd)
__device__ __forceinline__ void lds_load_lines(float2 *buf, const float2 *__restrict__ src, int64_t line_stride,
                                               int n, int log2n) {
    batched<8>(
        threadIdx.x, 8 * n / 2, FBLK,
        [&](int t) {
            const int line = (2 * t) >> log2n, e = (2 * t) & (n - 1);
            return *reinterpret_cast<const fft_v4f *>(src + line * line_stride + e);
        },
        [&](int t, fft_v4f v) {
            const int line = (2 * t) >> log2n, e = (2 * t) & (n - 1);
            float2 *d = buf + line * NP(n) + PADI(e);
            d[0] = make_float2(v.x, v.y);
            d[1] = make_float2(v.z, v.w);
        });
}
__device__ __forceinline__ void lds_store_lines(const float2 *buf, float2 *__restrict__ dst, int64_t line_stride, int n,
                                                int log2n) {
    for (int t = threadIdx.x; t < 8 * n / 2; t += FBLK) {
        const int line = (2 * t) >> log2n, e = (2 * t) & (n - 1);
        const float2 *s2 = buf + line * NP(n) + PADI(e);
        *reinterpret_cast<float4 *>(dst + line * line_stride + e) = make_float4(s2[0].x, s2[0].y, s2[1].x, s2[1].y);
    }
}

// ---- P1: z r2c, two real lines per complex FFT, transposed write ---------------------------------
// workgroup = 16 real lines (x fixed, y0..y0+15) = 8 complex lines
// (khalo on a slab: `nti` rows of tiles -- the planes of this launch -- and `slab_edges` set: the first and the last
// plane take nothing from a neighbouring column in x; what the neighbouring RANK's tiles add there came in through
// the ghost planes of the density)
__global__ __launch_bounds__(FBLK) void fftp_z_r2c(const float *__restrict__ rho, float2 *__restrict__ B,
                                                   const float2 *__restrict__ g_tw, int n, int log2n, int nx,
                                                   const float *__restrict__ khalo, int nti, int slab_edges) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    float2 *buf = lds, *tw = lds + 8 * NP(n);
    const int nh = (n >> 1) + 1;
    const int64_t line0 = (int64_t)blockIdx.x * 16;  // flat (x*n + y) index of the first real line
    load_twiddles(tw, g_tw, n, log2n);
    // k-halo buffer (pm_kick_drift_wrap_deposit_khalo): the contributions of every tile to the two k-cells just
    // outside it are not in rho but in khalo[((ti Tj + tj) Tk + tk)][low | high][li][lj] (tiles of 8 x 16 x 16
    // cells, li in 0..9 and lj in 0..17 with the one-cell halo).  This workgroup holds the 16 lines
    // (x, y0 .. y0 + 15), i.e. row li = x % 8 + 1 of the tile column (x / 8, y0 / 16): cell k = 16 m + 15
    // receives the low halo of the tiles at tk = m + 1, cell k = 16 m the high halo of those at tk = m - 1, from
    // the own tile column and -- on its faces -- from the neighbouring ones.  16 lanes read 16 consecutive lj
    // (64-byte pieces); the loads are issued here, ahead of the density's, unguarded (a lane off the j faces
    // re-reads its own entry and drops it), and are consumed after the line buffer is filled.
    constexpr int KH_PER_THREAD = 4;  // 2 x (n / 16) x 16 entries over FBLK threads, n <= 1024
    const int kT = n >> 4;
    auto fill = [&]() {  // contiguous load of 16*n floats; real line 2m -> re, 2m+1 -> im of complex line m
        const float4 *src = reinterpret_cast<const float4 *>(rho + line0 * n);
        const int nq = n >> 2;
        batched_once<8>(  // 16 n / 4 quads <= 8 FBLK for n <= 1024
            threadIdx.x, 16 * nq, FBLK, [&](int t) { return *reinterpret_cast<const fft_v4f *>(src + t); },
            [&](int t, fft_v4f v) {
                const int rl = t / nq, z0 = (t - rl * nq) << 2;
                // z0 is a multiple of 4: z0..z0+3 share a 16-group, so the four slots stay contiguous
                float *dst = reinterpret_cast<float *>(buf + (rl >> 1) * NP(n) + PADI(z0)) + (rl & 1);
                dst[0] = v.x, dst[2] = v.y, dst[4] = v.z, dst[6] = v.w;
            });
        __syncthreads();
    };
    // fill + halo buffer: two straight-line variants (interior row of the tile column / row on an i face) of the
    // whole front of the kernel, so that the halo loads stay in flight across the density loads (with the face
    // test inside, every pair of loads got its own basic block and vmcnt(0); joined after a branch, the loaded
    // values were copied -- and waited for -- at the join).
    // The buffer (pm_particles.hip, HaloBuf) has three parts: K [tile column][li][low|high][tk][lj] (the k-halo cells),
    // Hi [tile column][low|high][lj 0..17][k] and Hj [tile column][low|high][li 1..8][k] (whole lines along k).
    // The 16 lines (x, y0 .. y0 + 15) of this workgroup are row li = x % 8 + 1 of tile column (x / 8, y0 / 16):
    //  * line y0 also receives the high j-halo row of column tj - 1, line y0 + 15 the low one of tj + 1 (Hj);
    //  * on an i face (x % 8 == 0 or 7) every line receives a row of the neighbouring column's i-halo plane
    //    (Hi, lj = 1..16), and lines y0 / y0 + 15 the edge rows lj = 17 / 0 of the diagonal columns;
    //  * K as before.
    auto front = [&](auto with_iface) {
        constexpr bool IFACE = decltype(with_iface)::value;
        const int Ti = nti, nq = n >> 2;
        const int x = (int)(line0 >> log2n), y0 = (int)(line0 & (n - 1));
        const int ti = x >> 3, tj = y0 >> 4, xi = x & 7;
        const int di = xi == 0 ? -1 : 1, iside = xi == 0 ? 1 : 0;  // the neighbour's high plane (li = 9) or low one (li = 0)
        const int tin = (ti + di + Ti) % Ti, lin = xi + 1 - 8 * di;
        const int64_t nk = (int64_t)Ti * kT * kT * 360;
        const float *hi = khalo + nk, *hj = hi + (int64_t)Ti * kT * 36 * n;
        float kraw[KH_PER_THREAD][IFACE ? 4 : 2];  // own tile column, its j-face neighbour[, the i-face one, the edge one]
#pragma unroll
        for (int u = 0; u < KH_PER_THREAD; ++u) {
            const int idx = min((int)threadIdx.x + u * FBLK, 32 * kT - 1);
            const int rl = idx & 15, sm = idx >> 4, side = sm >= kT, m = sm - (side ? kT : 0);
            const int tk = side ? (m + kT - 1) % kT : (m + 1) % kT;
            const bool jf = rl == 0 || rl == 15;
            const int tjn = rl == 0 ? (tj + kT - 1) % kT : (tj + 1) % kT, ljn = rl == 0 ? 17 : 0;
            auto at = [&](int tii, int tjj, int li, int lj) {
                return khalo[(((((int64_t)tii * kT + tjj) * 10 + li) * 2 + side) * kT + tk) * 18 + lj];
            };
            // unconditional loads: off the j faces the second one re-reads the first entry and is dropped
            const int tj2 = jf ? tjn : tj, lj2 = jf ? ljn : rl + 1;
            kraw[u][0] = at(ti, tj, xi + 1, rl + 1);
            kraw[u][1] = at(ti, tj2, xi + 1, lj2);
            if constexpr (IFACE) kraw[u][2] = at(tin, tj, lin, rl + 1), kraw[u][3] = at(tin, tj2, lin, lj2);
        }
        // j-halo rows for lines 0 and 15 (+ the edge rows on an i face): one quad per thread (2 n / 4 <= FBLK)
        const int jt = min((int)threadIdx.x, 2 * nq - 1), jw = jt / nq, jq = jt - jw * nq;
        const int tjx = jw ? (tj + 1) % kT : (tj + kT - 1) % kT;
        fft_v4f jv = *reinterpret_cast<const fft_v4f *>(hj + ((((int64_t)ti * kT + tjx) * 2 + (jw ? 0 : 1)) * 8 + xi) * n + 4 * jq);
        fft_v4f ev = {0.f, 0.f, 0.f, 0.f};
        if constexpr (IFACE)
            ev = *reinterpret_cast<const fft_v4f *>(hi + ((((int64_t)tin * kT + tjx) * 2 + iside) * 18 + (jw ? 0 : 17)) * n + 4 * jq);
        // the neighbouring column's i-halo plane, one row per line: fetched here too (8 quads per thread)
        fft_v4f iv[IFACE ? 8 : 1];
        if constexpr (IFACE) {
            const float *src = hi + (((int64_t)tin * kT + tj) * 2 + iside) * 18 * n + n;  // row lj = 1
#pragma unroll
            for (int u = 0; u < 8; ++u)
                iv[u] = *reinterpret_cast<const fft_v4f *>(src + 4 * (int64_t)min((int)threadIdx.x + u * FBLK, 16 * nq - 1));
        }
        fill();
#pragma unroll
        for (int u = 0; u < KH_PER_THREAD; ++u) {  // K: the 2 x T boundary cells of each of the 16 lines
            const int idx = threadIdx.x + u * FBLK;
            if (idx < 32 * kT) {
                const int rl = idx & 15, sm = idx >> 4, side = sm >= kT, m = sm - (side ? kT : 0);
                const bool jf = rl == 0 || rl == 15;
                float sum = kraw[u][0] + (jf ? kraw[u][1] : 0.f);
                if constexpr (IFACE) sum = (sum + kraw[u][2]) + (jf ? kraw[u][3] : 0.f);
                float *dst = reinterpret_cast<float *>(buf + (rl >> 1) * NP(n) + PADI(side ? 16 * m : 16 * m + 15)) + (rl & 1);
                *dst += sum;
            }
        }
        __syncthreads();
        auto add4 = [&](int rl, int z0, fft_v4f v) {
            float *dst = reinterpret_cast<float *>(buf + (rl >> 1) * NP(n) + PADI(z0)) + (rl & 1);
            dst[0] += v.x, dst[2] += v.y, dst[4] += v.z, dst[6] += v.w;
        };
        if ((int)threadIdx.x < 2 * nq) add4(jw ? 15 : 0, 4 * jq, IFACE ? jv + ev : jv);
        __syncthreads();
        if constexpr (IFACE) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = threadIdx.x + u * FBLK;
                if (t < 16 * nq) {
                    const int rl = t / nq;
                    add4(rl, (t - rl * nq) << 2, iv[u]);
                }
            }
            __syncthreads();
        }
    };
    const int xplane = (int)(line0 >> log2n);
    if (!khalo)
        fill();
    else if (((xplane & 7) == 0 || (xplane & 7) == 7) && !(slab_edges && (xplane == 0 || xplane == nx - 1)))
        front(std::true_type{});
    else
        front(std::false_type{});
    fft_forward_lds(buf, tw, n, log2n, 8);
    // unpack and write B[x][kz][y]: thread = (complex line m, kz) -> the two real lines' spectra A, B are
    // adjacent in y: one 16-byte store; 8 threads cover the 128 B of one kz
    const int x = (int)(line0 >> log2n), y0 = (int)(line0 & (n - 1));
    const int m = threadIdx.x & 7;
    const float2 *ln = buf + m * NP(n);
    for (int kz = threadIdx.x >> 3; kz < nh; kz += FBLK / 8) {
        const float2 zk = ln[PADI(kz)], zn = cconj(ln[PADI((n - kz) & (n - 1))]);
        const float2 d = csub(zk, zn);
        // A[k] = (Z[k] + conj Z[n-k]) / 2 ;  B[k] = (Z[k] - conj Z[n-k]) / (2i)
        const float4 v = make_float4(0.5f * (zk.x + zn.x), 0.5f * (zk.y + zn.y), 0.5f * d.y, -0.5f * d.x);
        *reinterpret_cast<float4 *>(B + ((((int64_t)(x >> 3) * nh + kz) << 3) + (x & 7)) * n + y0 + 2 * m) = v;
    }
}

// element (ky, kz, x) of C: C[ky][kz][x] with nx local x values (slab layout), C[kz][ky][x] on a single GPU.
// (Tried: (ky pair, 8 x) tiles so that a y-pass workgroup moves whole 128-byte lines -- the y passes gained
// what the x pass, then reading 64-byte pieces, lost.)
__device__ __forceinline__ int64_t c_index(int ky, int kz, int x, int n, int nh, int nx, int kzmajor) {
    return (kzmajor ? (int64_t)kz * n + ky : (int64_t)ky * nh + kz) * nx + x;
}

// ---- P2 / P4: y transform with transposed global side ------------------------------------------------
// workgroup = 8 lines (kz fixed, x0..x0+7).  FORWARD: B -> C, else C -> B.
template <bool FORWARD>
__global__ __launch_bounds__(FBLK) void fftp_y(float2 *__restrict__ B, float2 *__restrict__ C,
                                               const float2 *__restrict__ g_tw, int n, int log2n, int nx,
                                               int kzmajor, int kz0, int nkz) {
    // C holds the kz range [kz0, kz0 + nkz) only (slab layout: [ky][nkz][nx]; the whole spectrum is kz0 = 0,
    // nkz = n/2 + 1); B always holds all of it
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    float2 *buf = lds, *tw = lds + 8 * NP(n);
    const int nh = (n >> 1) + 1;
    // A workgroup touches C in 64-byte pieces (8 x-values per (ky, kz)); the other half of each 128-byte
    // line belongs to the next group of 8 lines.  Workgroups b and b + 8 land on the same XCD back to back
    // (round-robin dispatch), so they are given such neighbouring groups: the halves meet in that XCD's L2
    // and HBM sees whole lines (PMC before: the inverse pass fetched 2x its bytes).
    unsigned grp = blockIdx.x;
    if (grp < (gridDim.x & ~15u)) grp = (grp & ~15u) + ((grp & 7u) << 1) + ((grp >> 3) & 1u);
    const int64_t line0 = (int64_t)grp * 8;  // flat (kz_local*nx + x), nx = local x planes (multiple of 8)
    const int kzl = (int)(line0 / nx), x0 = (int)(line0 - (int64_t)kzl * nx), kz = kz0 + kzl;
    load_twiddles(tw, g_tw, n, log2n);
    // transposed side: thread = (ky, pair of x) -> one 16-byte access; 4 lanes cover the 64-byte piece
    const int lx = (threadIdx.x & 3) << 1;
    if (FORWARD) {
        lds_load_lines(buf, B + (((int64_t)(x0 >> 3) * nh + kz) << 3) * n, n, n, log2n);
        __syncthreads();
        fft_forward_lds(buf, tw, n, log2n, 8);
        for (int ky = threadIdx.x >> 2; ky < n; ky += FBLK / 4) {
            const float2 a = buf[lx * NP(n) + PADI(ky)], b = buf[(lx + 1) * NP(n) + PADI(ky)];
            *reinterpret_cast<float4 *>(C + c_index(ky, kzl, x0 + lx, n, nkz, nx, kzmajor)) = make_float4(a.x, a.y, b.x, b.y);
        }
    } else {
        batched<8>(
            threadIdx.x >> 2, n, FBLK / 4,
            [&](int ky) {
                return *reinterpret_cast<const fft_v4f *>(C + c_index(ky, kzl, x0 + lx, n, nkz, nx, kzmajor));
            },
            [&](int ky, fft_v4f v) {
                buf[lx * NP(n) + PADI(ky)] = make_float2(v.x, v.y);
                buf[(lx + 1) * NP(n) + PADI(ky)] = make_float2(v.z, v.w);
            });
        __syncthreads();
        fft_inverse_lds(buf, tw, n, log2n, 8);
        lds_store_lines(buf, B + (((int64_t)(x0 >> 3) * nh + kz) << 3) * n, n, n, log2n);
    }
}

// A line of the x pass is stored as nseg segments of nx elements: element x of line l lives at
// ((x / nx) * nlines_total + l) * nx + x % nx (nseg = 1: plain contiguous lines).  This is the layout an
// all-to-all over x-slabs delivers: [source rank][ky_local][kz][x_local].

// ---- P3: x forward, Green's function, x inverse, in place ----------------------------------------------
// workgroup = 8 consecutive lines (ky_local, kz) of C viewed as [nseg][nky*nh lines][nx]; ky0 = global
// index of the first local ky (0 on a single GPU)
// The x pass of one line for n = 64 R1 (1024 = 16 * 16 * 4, 512 = 8 * 8 * 8), the line passing through LDS four times
// instead of nine: the first forward stage takes its inputs straight from global memory (butterfly `lane` reads
// elements lane + 64 t: consecutive lanes, consecutive addresses), the last forward stage leaves element lane + 64 m
// in the lane that the first INVERSE stage wants it in (both cover exactly the elements = lane mod 64), so the
// Green's function is applied and the hand-over made in registers, and the last inverse stage stores to global
// memory.  Same stages, same twiddles, same Green's expression as the generic form below.
template <int KIND, int R1, int R2, int R3, class XAddr>
__device__ __forceinline__ void xgreen_line_fused(float2 *__restrict__ C, XAddr xaddr, float2 *ln, const float2 *tw, int n,
                                                  int log2n, const float *__restrict__ k2, const float *__restrict__ s,
                                                  float c0, float norm, int ky, int kz) {
    constexpr int BPL3 = R1 / R3;  // butterflies per lane of the last stage (n / R3 / 64)
    constexpr int L1 = R1 == 16 ? 4 : 3, L2 = R2 == 16 ? 4 : 3;
    const int lane = threadIdx.x & 63;
    float2 a1[1][R1];
#pragma unroll
    for (int t = 0; t < R1; ++t) {
        const fft_v2f v = *reinterpret_cast<const fft_v2f *>(C + xaddr(lane + 64 * t));
        a1[0][t] = make_float2(v.x, v.y);
    }
    stage_compute<R1, 1, false>(a1, tw, n, 0);
    stage_store<R1, 1>(a1, ln, n, 0);
    wave_lds_fence();
    stockham_stage<R2, 1, false>(ln, tw, n, log2n, L1);
    wave_lds_fence();
    float2 a3[BPL3][R3];
    stage_load<R3, BPL3>(a3, ln, n);
    stage_compute<R3, BPL3, false>(a3, tw, n, L1 + L2);
    // output t of butterfly lane + 64 i is element lane + 64 i + t (n / R3) = lane + 64 (i + t BPL3)
    float2 b1[1][R1];
#pragma unroll
    for (int i = 0; i < BPL3; ++i)
#pragma unroll
        for (int t = 0; t < R3; ++t) {
            const int m = i + t * BPL3, kx = lane + 64 * m;
            float g;  // fourier.py:460-595; for the compensated kernel `s` already holds sinc^(-2p) per axis
            if (KIND == PM_GREEN_7PT) {
                g = c0 / ((s[kx] + s[ky]) + s[kz]);
            } else {
                const float invk2 = c0 / ((k2[kx] + k2[ky]) + k2[kz]);
                g = (KIND == PM_GREEN_COMPENSATED) ? ((s[kx] * s[ky]) * s[kz]) * invk2 : invk2;
            }
            g *= norm;  // 1/n^3 of the inverse transform
            float2 v = a3[i][t];
            if ((kx | ky | kz) == 0)
                v = make_float2(0.f, 0.f);
            else
                v.x *= g, v.y *= g;
            b1[0][m] = v;
        }
    stage_compute<R1, 1, true>(b1, tw, n, 0);
    wave_lds_fence();  // (every lane has read its stage-3 inputs before the line is overwritten)
    stage_store<R1, 1>(b1, ln, n, 0);
    wave_lds_fence();
    stockham_stage<R2, 1, true>(ln, tw, n, log2n, L1);
    wave_lds_fence();
    float2 c3[BPL3][R3];
    stage_load<R3, BPL3>(c3, ln, n);
    stage_compute<R3, BPL3, true>(c3, tw, n, L1 + L2);
#pragma unroll
    for (int i = 0; i < BPL3; ++i)
#pragma unroll
        for (int t = 0; t < R3; ++t)
            *reinterpret_cast<fft_v2f *>(C + xaddr(lane + 64 * (i + t * BPL3))) = fft_v2f{c3[i][t].x, c3[i][t].y};
}

template <int KIND>
__global__ __launch_bounds__(FBLK) void fftp_x_green(float2 *__restrict__ C, const float2 *__restrict__ g_tw, int n,
                                                     int log2n, int p, const float *__restrict__ k2,
                                                     const float *__restrict__ s, float c0, float norm, int nx,
                                                     int nky, int ky0, int kzmajor, int kz0, int nkz) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    float2 *buf = lds, *tw = lds + 8 * NP(n);
    const int nh = (n >> 1) + 1;
    const int64_t line0 = (int64_t)blockIdx.x * 8;
    load_twiddles(tw, g_tw, n, log2n);
    const int64_t nlines_total = (int64_t)nky * nkz;
    __syncthreads();  // twiddle table; from here on every wave works on its own line, no workgroup barrier
    const int l = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t line = line0 + l;
    float2 *ln = buf + l * NP(n);
    // line index: ky_local * nkz + kz_local, or kz_local * n + ky on a single GPU (kzmajor, where nseg = 1)
    const int kyl = kzmajor ? (int)(line & (n - 1)) : (int)(line / nkz);
    const int kzl = kzmajor ? (int)(line >> log2n) : (int)(line - (int64_t)kyl * nkz);
    const int kz = kz0 + kzl, ky = ky0 + kyl;
    auto xaddr = [&](int e) -> int64_t {
        if (kzmajor) return c_index(kyl, kzl, e, n, nkz, nx, 1);
        const int seg = e / nx, xl = e - seg * nx;
        return ((int64_t)seg * nlines_total + line) * nx + xl;
    };
    if (log2n == 10) {
        xgreen_line_fused<KIND, 16, 16, 4>(C, xaddr, ln, tw, n, log2n, k2, s, c0, norm, ky, kz);
        return;
    }
    if (log2n == 9) {
        xgreen_line_fused<KIND, 8, 8, 8>(C, xaddr, ln, tw, n, log2n, k2, s, c0, norm, ky, kz);
        return;
    }
    batched<8>(  // 16-byte accesses: elements (e, e+1), e even; the clamp keeps e even (n - 2)
        lane, n >> 1, 64, [&](int h) { return *reinterpret_cast<const fft_v4f *>(C + xaddr(2 * h)); },
        [&](int h, fft_v4f v) {
            float2 *d = ln + PADI(2 * h);
            d[0] = make_float2(v.x, v.y);
            d[1] = make_float2(v.z, v.w);
        });
    wave_lds_fence();
    fft_lines_lds<false, true>(buf, tw, n, log2n);
    {
        for (int kx = lane; kx < n; kx += 64) {
            float g;  // fourier.py:460-595; for the compensated kernel `s` already holds sinc^(-2p) per axis
            if (KIND == PM_GREEN_7PT) {
                g = c0 / ((s[kx] + s[ky]) + s[kz]);
            } else {
                const float invk2 = c0 / ((k2[kx] + k2[ky]) + k2[kz]);
                g = (KIND == PM_GREEN_COMPENSATED) ? ((s[kx] * s[ky]) * s[kz]) * invk2 : invk2;
            }
            g *= norm;  // 1/n^3 of the inverse transform
            float2 *slot = ln + PADI(kx);
            float2 v = *slot;
            if ((kx | ky | kz) == 0)
                v = make_float2(0.f, 0.f);
            else
                v.x *= g, v.y *= g;
            *slot = v;
        }
    }
    wave_lds_fence();
    fft_lines_lds<true, true>(buf, tw, n, log2n);
    for (int e = 2 * lane; e < n; e += 128) {
        const float2 *s2 = ln + PADI(e);
        *reinterpret_cast<float4 *>(C + xaddr(e)) = make_float4(s2[0].x, s2[0].y, s2[1].x, s2[1].y);
    }
}

// ---- P5: z c2r, transposed read ---------------------------------------------------------------------------
__global__ __launch_bounds__(FBLK) void fftp_z_c2r(const float2 *__restrict__ B, float *__restrict__ phi,
                                                   const float2 *__restrict__ g_tw, int n, int log2n, int nx) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    float2 *buf = lds, *tw = lds + 8 * NP(n);
    const int nh = (n >> 1) + 1;
    const int64_t line0 = (int64_t)blockIdx.x * 16;
    const int x = (int)(line0 >> log2n), y0 = (int)(line0 & (n - 1));
    load_twiddles(tw, g_tw, n, log2n);
    // Z = A + iB: thread = (complex line m, kz) loads the adjacent pair (A[kz], B[kz]) with one 16-byte
    // load and writes Z[k] = A + iB and Z[n-k] = conj(A) + i conj(B)
    {
        const int m = threadIdx.x & 7;
        float2 *ln = buf + m * NP(n);
        batched<8>(threadIdx.x >> 3, nh, FBLK / 8, [&](int kz) {
            return *reinterpret_cast<const fft_v4f *>(B + ((((int64_t)(x >> 3) * nh + kz) << 3) + (x & 7)) * n + y0 + 2 * m);
        }, [&](int kz, fft_v4f v) {
            const bool edge = (kz == 0) || (kz == (n >> 1));
            if (edge) v.y = 0.f, v.w = 0.f;  // c2r ignores the imaginary part of DC / Nyquist
            // A = (v.x, v.y), Bh = (v.z, v.w):  A + i Bh = (v.x - v.w, v.y + v.z)
            ln[PADI(kz)] = make_float2(v.x - v.w, v.y + v.z);
            // conj(A) + i conj(Bh) = (v.x + v.w, -v.y + v.z)
            if (!edge) ln[PADI(n - kz)] = make_float2(v.x + v.w, v.z - v.y);
        });
    }
    __syncthreads();
    fft_inverse_lds(buf, tw, n, log2n, 8);
    {
        float4 *dst = reinterpret_cast<float4 *>(phi + line0 * n);
        const int nq = n >> 2;
        for (int t = threadIdx.x; t < 16 * nq; t += FBLK) {
            const int rl = t / nq, z0 = (t - rl * nq) << 2;
            const float *src = reinterpret_cast<const float *>(buf + (rl >> 1) * NP(n) + PADI(z0)) + (rl & 1);
            dst[t] = make_float4(src[0], src[2], src[4], src[6]);
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------------------

// The work arrays of the five passes are rewritten in place (C by the x pass) or read and written whole by every
// pass; how fast an in-place pass over a multi-GB allocation runs depends on the physical memory it landed in (5.08
// or 4.5 TB/s, DESIGN.md section 5 "placement").  Large buffers are therefore chosen: a few candidate allocations, an
// in-place pass over each (x * 1, the multiplier known at run time only), the fastest kept, the others freed.
typedef float fftp_pv4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void fftp_probe_kernel(fftp_pv4 *__restrict__ a, int64_t quads, float one) {
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (int64_t)gridDim.x * 256)
        __builtin_nontemporal_store(__builtin_nontemporal_load(a + q) * one, a + q);
}

static int fftp_alloc_chosen(pm_ctx *c, size_t bytes, void **out) {
    *out = nullptr;
    constexpr int K = 5;
    size_t free_b = 0, total_b = 0;
    if (bytes < ((size_t)1 << 30) || hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < (K + 1) * bytes) {
        (void)hipGetLastError();
        PM_HIP(hipMalloc(out, bytes));
        return PM_OK;
    }
    void *cand[K] = {nullptr};
    float ms[K];
    int got = 0;
    for (; got < K; ++got)
        if (hipMalloc(&cand[got], bytes) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
    if (got == 0) {
        pm_set_error("out of device memory for the FFT work arrays");
        return PM_EHIP;
    }
    const int64_t quads = (int64_t)(bytes / 16);
    int best = 0;
    for (int k = 0; k < got; ++k) {
        ms[k] = 1e30f;
        (void)hipMemsetAsync(cand[k], 0, bytes, c->stream);
        for (int rep = 0; rep < 3; ++rep) {  // (the first pass also faults the pages in)
            (void)hipEventRecord(c->ev0, c->stream);
            hipLaunchKernelGGL(fftp_probe_kernel, dim3(256 * 16), dim3(256), 0, c->stream, (fftp_pv4 *)cand[k], quads, 1.0f);
            (void)hipEventRecord(c->ev1, c->stream);
            (void)hipEventSynchronize(c->ev1);
            float t = 1e30f;
            if (hipEventElapsedTime(&t, c->ev0, c->ev1) == hipSuccess && rep > 0 && t < ms[k]) ms[k] = t;
        }
        if (ms[k] < ms[best]) best = k;
    }
    (void)hipGetLastError();
    for (int k = 0; k < got; ++k)
        if (k != best) (void)hipFree(cand[k]);
    *out = cand[best];
    return PM_OK;
}

static int ensure_fused_buffers(pm_ctx *c, size_t b_bytes, size_t c_bytes) {
    const int n = c->n;
    if (!c->fp_tw) {
        PM_HIP(hipMalloc(&c->fp_tw, (size_t)n * sizeof(float2)));
        c->h_pow = new float[n];
        float2 *h = new float2[n];
        for (int t = 0; t < n; ++t) {
            const double a = -2.0 * 3.14159265358979323846 * t / n;
            h[t] = make_float2((float)cos(a), (float)sin(a));
        }
        hipError_t e = hipMemcpy(c->fp_tw, h, (size_t)n * sizeof(float2), hipMemcpyHostToDevice);
        delete[] h;
        PM_HIP(e);
        // up to 80 KB of dynamic LDS (78 KB at n = 1024)
        PM_HIP(hipFuncSetAttribute((const void *)fftp_z_r2c, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        PM_HIP(hipFuncSetAttribute((const void *)fftp_y<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        PM_HIP(hipFuncSetAttribute((const void *)fftp_y<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        PM_HIP(hipFuncSetAttribute((const void *)fftp_x_green<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        PM_HIP(hipFuncSetAttribute((const void *)fftp_x_green<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        PM_HIP(hipFuncSetAttribute((const void *)fftp_x_green<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        PM_HIP(hipFuncSetAttribute((const void *)fftp_z_c2r, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    }
    if (c->fp_B_bytes < b_bytes) {
        if (c->fp_B) PM_HIP(hipFree(c->fp_B));
        c->fp_B = nullptr, c->fp_B_bytes = 0;
        if (int rc = fftp_alloc_chosen(c, b_bytes, &c->fp_B)) return rc;
        c->fp_B_bytes = b_bytes;
    }
    if (c->fp_C_bytes < c_bytes) {
        if (c->fp_C) PM_HIP(hipFree(c->fp_C));
        c->fp_C = nullptr, c->fp_C_bytes = 0;
        if (int rc = fftp_alloc_chosen(c, c_bytes, &c->fp_C)) return rc;
        c->fp_C_bytes = c_bytes;
    }
    return PM_OK;
}

static inline size_t fftp_lds_bytes(int n) {
    return (size_t)(8 * NP(n) + n) * sizeof(float2);
}

#define FFTP_CHECK_N(c)                                                                                        \
    PM_REQUIRE((c)->n >= 32 && (c)->n <= 1024, "LDS-FFT passes support 32 <= n <= 1024 (use the rocFFT entry points otherwise)")

// ---- stages (each a C-ABI entry point so that a slab driver can pipeline them against its all-to-alls) ----
// P1: d_rhs [nx, n, n] real -> the context's B (all kz)
static int fftp_z_forward(pm_ctx *c, const float *d_rhs, int nx, const float *d_khalo = nullptr) {
    PM_REQUIRE(c && d_rhs, "NULL argument");
    FFTP_CHECK_N(c);
    const int n = c->n, log2n = c->log2n, nh = n / 2 + 1;
    PM_REQUIRE(nx >= 8 && nx <= n && (nx % 8) == 0, "nx must be a multiple of 8 in [8, n]");
    PM_REQUIRE(pm_aligned16(d_rhs), "buffers must be 16-byte aligned");
    int rc = ensure_fused_buffers(c, (size_t)nh * nx * n * sizeof(float2), 0);
    if (rc) return rc;
    hipLaunchKernelGGL(fftp_z_r2c, dim3((unsigned)((int64_t)nx * n / 16)), dim3(FBLK), fftp_lds_bytes(n), c->stream,
                       d_rhs, (float2 *)c->fp_B, (const float2 *)c->fp_tw, n, log2n, nx, d_khalo, nx >> 3, nx < n ? 1 : 0);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

// P2 / P4 for the kz range [kz0, kz0 + nkz): B <-> d_C [n (ky), nkz, nx] (slab layout) or kz-major
static int fftp_y_pass(pm_ctx *c, bool forward, void *d_C, int nx, int kz0, int nkz, int kzmajor) {
    PM_REQUIRE(c && d_C, "NULL argument");
    FFTP_CHECK_N(c);
    const int n = c->n, log2n = c->log2n, nh = n / 2 + 1;
    PM_REQUIRE(nx >= 8 && nx <= n && (nx % 8) == 0, "nx must be a multiple of 8 in [8, n]");
    PM_REQUIRE(kz0 >= 0 && nkz >= 1 && kz0 + nkz <= nh, "bad kz range");
    PM_REQUIRE(pm_aligned16(d_C), "buffers must be 16-byte aligned");
    int rc = ensure_fused_buffers(c, (size_t)nh * nx * n * sizeof(float2), 0);
    if (rc) return rc;
    const float2 *tw = (const float2 *)c->fp_tw;
    const unsigned grid = (unsigned)((int64_t)nkz * nx / 8);
    if (forward)
        hipLaunchKernelGGL((fftp_y<true>), dim3(grid), dim3(FBLK), fftp_lds_bytes(n), c->stream, (float2 *)c->fp_B,
                           (float2 *)d_C, tw, n, log2n, nx, kzmajor, kz0, nkz);
    else
        hipLaunchKernelGGL((fftp_y<false>), dim3(grid), dim3(FBLK), fftp_lds_bytes(n), c->stream, (float2 *)c->fp_B,
                           (float2 *)d_C, tw, n, log2n, nx, kzmajor, kz0, nkz);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

// P5: the context's B -> d_phi [nx, n, n] real
static int fftp_z_inverse(pm_ctx *c, int nx, float *d_phi) {
    PM_REQUIRE(c && d_phi, "NULL argument");
    FFTP_CHECK_N(c);
    const int n = c->n, log2n = c->log2n, nh = n / 2 + 1;
    PM_REQUIRE(nx >= 8 && nx <= n && (nx % 8) == 0, "nx must be a multiple of 8 in [8, n]");
    PM_REQUIRE(pm_aligned16(d_phi), "buffers must be 16-byte aligned");
    int rc = ensure_fused_buffers(c, (size_t)nh * nx * n * sizeof(float2), 0);
    if (rc) return rc;
    hipLaunchKernelGGL(fftp_z_c2r, dim3((unsigned)((int64_t)nx * n / 16)), dim3(FBLK), fftp_lds_bytes(n), c->stream,
                       (const float2 *)c->fp_B, d_phi, (const float2 *)c->fp_tw, n, log2n, nx);
    PM_LAUNCH_CHECK();
    return PM_OK;
}

extern "C" int pm_fftp_z_forward(pm_ctx *c, const float *d_rhs, int nx) { return fftp_z_forward(c, d_rhs, nx); }
// ... of a density whose halo contributions are still in the buffer pm_kick_drift_wrap_deposit_slab_khalo filled
// (pm_khalo_floats_slab(n, nx) floats: the nx / 8 rows of tiles of these planes)
extern "C" int pm_fftp_z_forward_khalo(pm_ctx *c, const float *d_rhs, int nx, const float *d_khalo) {
    PM_REQUIRE(d_khalo, "NULL argument");
    PM_REQUIRE(c && c->n >= 16, "the halo buffer needs n >= 16");
    return fftp_z_forward(c, d_rhs, nx, d_khalo);
}
extern "C" int pm_fftp_y_forward(pm_ctx *c, int nx, int kz0, int nkz, void *d_C) {
    return fftp_y_pass(c, true, d_C, nx, kz0, nkz, 0);
}
extern "C" int pm_fftp_y_inverse(pm_ctx *c, const void *d_C, int nx, int kz0, int nkz) {
    return fftp_y_pass(c, false, const_cast<void *>(d_C), nx, kz0, nkz, 0);
}
extern "C" int pm_fftp_z_inverse(pm_ctx *c, int nx, float *d_phi) { return fftp_z_inverse(c, nx, d_phi); }

// P1 + P2 on nx local x-planes: d_rhs [nx, n, n] real -> d_C [n (ky), n/2+1 (kz), nx] complex
static int fftp_forward_impl(pm_ctx *c, const float *d_rhs, int nx, void *d_C, int kzmajor,
                             const float *d_khalo = nullptr) {
    int rc = fftp_z_forward(c, d_rhs, nx, d_khalo);
    if (rc) return rc;
    return fftp_y_pass(c, true, d_C, nx, 0, c->n / 2 + 1, kzmajor);
}

extern "C" int pm_fftp_forward(pm_ctx *c, const float *d_rhs, int nx, void *d_C) {
    PM_REQUIRE(d_C, "NULL argument");
    return fftp_forward_impl(c, d_rhs, nx, d_C, 0);
}

// P4 + P5: d_C [n, n/2+1, nx] -> d_phi [nx, n, n] real (d_C is read only)
static int fftp_inverse_impl(pm_ctx *c, const void *d_C, int nx, float *d_phi, int kzmajor) {
    PM_REQUIRE(c, "NULL argument");
    int rc = fftp_y_pass(c, false, const_cast<void *>(d_C), nx, 0, c->n / 2 + 1, kzmajor);
    if (rc) return rc;
    return fftp_z_inverse(c, nx, d_phi);
}

extern "C" int pm_fftp_inverse(pm_ctx *c, const void *d_C, int nx, float *d_phi) {
    return fftp_inverse_impl(c, d_C, nx, d_phi, 0);
}

// P3 in place on d_C laid out [nseg = n/nx][nky][nkz][nx]: the lines (ky0 + ky_local, kz0 + kz_local) along x
static int fftp_xgreen_impl(pm_ctx *c, int kind, int p, void *d_C, int nx, int nky, int ky0, const float *h_k2,
                            const float *h_s, float c0, int kzmajor, int kz0, int nkz) {
    PM_REQUIRE(c && d_C && h_k2 && h_s, "NULL argument");
    FFTP_CHECK_N(c);
    const int n = c->n, log2n = c->log2n, nh = n / 2 + 1;
    PM_REQUIRE(nx >= 8 && nx <= n && (n % nx) == 0 && (nx % 2) == 0, "nx must divide n");
    PM_REQUIRE(nky >= 8 && (nky % 8) == 0 && ky0 >= 0 && ky0 + nky <= n, "bad ky range (multiple of 8 lines)");
    PM_REQUIRE(kz0 >= 0 && nkz >= 1 && kz0 + nkz <= nh, "bad kz range");
    PM_REQUIRE(kind == PM_GREEN_PLAIN || kind == PM_GREEN_COMPENSATED || kind == PM_GREEN_7PT,
               "unknown Green's function kind");
    PM_REQUIRE(pm_aligned16(d_C), "buffers must be 16-byte aligned");
    int rc = ensure_fused_buffers(c, 0, 0);
    if (rc) return rc;
    PM_HIP(hipMemcpyAsync(c->d_tab, h_k2, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (kind == PM_GREEN_COMPENSATED) {
        // (sx*sy*sz)^(-2p) = sx^(-2p) * sy^(-2p) * sz^(-2p): one pow per axis entry on the host instead
        // of one per mode on the device (a few ulp from the reference's single pow, fourier.py:543)
        float *hp = c->h_pow;
        for (int i = 0; i < n; ++i) hp[i] = (float)pow((double)h_s[i], (double)(-2 * p));
        PM_HIP(hipMemcpyAsync(c->d_tab + n, hp, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    } else {
        PM_HIP(hipMemcpyAsync(c->d_tab + n, h_s, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    }
    const float2 *tw = (const float2 *)c->fp_tw;
    const size_t lds = fftp_lds_bytes(n);
    const float norm = (float)(1.0 / ((double)n * n * n));
    const unsigned g3 = (unsigned)((int64_t)nky * nkz / 8);
    float2 *C = (float2 *)d_C;
    switch (kind) {
        case PM_GREEN_PLAIN:
            hipLaunchKernelGGL((fftp_x_green<PM_GREEN_PLAIN>), dim3(g3), dim3(FBLK), lds, c->stream, C, tw, n, log2n, p,
                               c->d_tab, c->d_tab + n, c0, norm, nx, nky, ky0, kzmajor, kz0, nkz);
            break;
        case PM_GREEN_COMPENSATED:
            hipLaunchKernelGGL((fftp_x_green<PM_GREEN_COMPENSATED>), dim3(g3), dim3(FBLK), lds, c->stream, C, tw, n,
                               log2n, p, c->d_tab, c->d_tab + n, c0, norm, nx, nky, ky0, kzmajor, kz0, nkz);
            break;
        default:
            hipLaunchKernelGGL((fftp_x_green<PM_GREEN_7PT>), dim3(g3), dim3(FBLK), lds, c->stream, C, tw, n, log2n, p,
                               c->d_tab, c->d_tab + n, c0, norm, nx, nky, ky0, kzmajor, kz0, nkz);
            break;
    }
    PM_LAUNCH_CHECK();
    return PM_OK;
}

extern "C" int pm_fftp_xgreen(pm_ctx *c, int kind, int p, void *d_C, int nx, int nky, int ky0, const float *h_k2,
                              const float *h_s, float c0) {
    PM_REQUIRE(c, "NULL argument");
    return fftp_xgreen_impl(c, kind, p, d_C, nx, nky, ky0, h_k2, h_s, c0, 0, 0, c->n / 2 + 1);
}

// the same for a spectrum holding the kz range [kz0, kz0 + nkz) only: [nseg][nky][nkz][nx]
extern "C" int pm_fftp_xgreen_kz(pm_ctx *c, int kind, int p, void *d_C, int nx, int nky, int ky0, int kz0, int nkz,
                                 const float *h_k2, const float *h_s, float c0) {
    return fftp_xgreen_impl(c, kind, p, d_C, nx, nky, ky0, h_k2, h_s, c0, 0, kz0, nkz);
}

static int poisson_fft_fused(pm_ctx *c, int kind, int p, const float *d_rhs, const float *d_khalo, float *d_phi,
                             const float *h_k2, const float *h_s, float c0) {
    PM_REQUIRE(c && d_rhs && d_phi && h_k2 && h_s, "NULL argument");
    FFTP_CHECK_N(c);
    const int n = c->n;
    int rc = ensure_fused_buffers(c, 0, (size_t)n * n * (n / 2 + 1) * sizeof(float2));
    if (rc) return rc;
    rc = fftp_forward_impl(c, d_rhs, n, c->fp_C, 1, d_khalo);  // single GPU: C is kz-major (see the header)
    if (rc) return rc;
    rc = fftp_xgreen_impl(c, kind, p, c->fp_C, n, n, 0, h_k2, h_s, c0, 1, 0, n / 2 + 1);
    if (rc) return rc;
    return fftp_inverse_impl(c, c->fp_C, n, d_phi, 1);
}

extern "C" int pm_poisson_fft_fused(pm_ctx *c, int kind, int p, const float *d_rhs, float *d_phi, const float *h_k2,
                                    const float *h_s, float c0) {
    return poisson_fft_fused(c, kind, p, d_rhs, nullptr, d_phi, h_k2, h_s, c0);
}

// The same for a density whose k-halo contributions are still in d_khalo (pm_kick_drift_wrap_deposit_khalo)
extern "C" int pm_poisson_fft_fused_khalo(pm_ctx *c, int kind, int p, const float *d_rhs, const float *d_khalo,
                                          float *d_phi, const float *h_k2, const float *h_s, float c0) {
    PM_REQUIRE(d_khalo, "NULL argument");
    return poisson_fft_fused(c, kind, p, d_rhs, d_khalo, d_phi, h_k2, h_s, c0);
}
