// Microbenchmark (development tool): issue cost of the vector instructions the gather kernel is made of, on gfx950,
// at 1, 2 and 4 waves per SIMD.  Cycles come from s_memtime inside the kernel (shader clock), so the chip's clock
// under load does not enter.  Prints cycles per wave-instruction per SIMD (= what a SIMD needs for one such
// instruction when its waves keep it busy).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
constexpr int LOOPS = 256;  // x 64 instructions

typedef float v2f __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void k(uint64_t *out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 * 0.5f, a2 = a0 + 3.f, a3 = a0 - 7.f;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a2}, p3 = {a3, a0};
    unsigned u0 = threadIdx.x * 7u + 1u, u1 = u0 ^ 0x55u, u2 = u0 + 99u, u3 = u0 * 3u;
    uint64_t w0 = u0, w1 = u1;
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u;
    // LDS addresses: conflict-free (consecutive 8-byte slots) and a 4 x 4 x 4 block of the 10 x 18 x 18 tile
    const unsigned a_lin8 = lane * 8u, a_lin4 = lane * 4u;
    const unsigned cell = (lane & 3u) + ((lane >> 2) & 3u) * 18u + (lane >> 4) * 324u;
    const unsigned a_blk8 = cell * 8u, a_blk4 = cell * 4u;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < LOOPS; ++it) {
        if (OP == 0) asm volatile(REP64("v_mul_f32 %0, %1, %2\n v_mul_f32 %1, %2, %3\n") : "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3));
        if (OP == 1) asm volatile(REP64("v_pk_mul_f32 %0, %1, %2\n v_pk_mul_f32 %1, %2, %3\n") : "+v"(p0), "+v"(p1) : "v"(p2), "v"(p3));
        if (OP == 2) asm volatile(REP64("v_pk_add_f32 %0, %1, %2\n v_pk_add_f32 %1, %2, %3\n") : "+v"(p0), "+v"(p1) : "v"(p2), "v"(p3));
        if (OP == 3) asm volatile(REP64("v_fma_f32 %0, %1, %2, %3\n v_fma_f32 %1, %2, %3, %0\n") : "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3));
        if (OP == 4) asm volatile(REP64("v_mad_u32_u24 %0, %1, %2, %3\n v_mad_u32_u24 %1, %2, %3, %0\n") : "+v"(u0), "+v"(u1) : "v"(u2), "v"(u3));
        if (OP == 5) asm volatile(REP64("v_mul_lo_u32 %0, %1, %2\n v_mul_lo_u32 %1, %2, %3\n") : "+v"(u0), "+v"(u1) : "v"(u2), "v"(u3));
        if (OP == 6) asm volatile(REP64("v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n") : "+v"(u0), "+v"(u1) : "v"(u2), "v"(u3) : "vcc");
        if (OP == 7) asm volatile(REP64("v_mad_u64_u32 %0, vcc, %2, %3, %1\n v_mad_u64_u32 %1, vcc, %3, %2, %0\n") : "+v"(w0), "+v"(w1) : "v"(u2), "v"(u3) : "vcc");
        if (OP == 8) asm volatile(REP64("v_lshl_add_u64 %0, %1, 2, %0\n v_lshl_add_u64 %1, %0, 2, %1\n") : "+v"(w0), "+v"(w1));
        if (OP == 9) asm volatile(REP64("v_cvt_i32_f32 %0, %2\n v_cvt_f32_i32 %1, %3\n") : "+v"(u0), "+v"(a1) : "v"(a2), "v"(u3));
        if (OP == 10) asm volatile(REP64("v_max_u32 %0, %1, %2\n v_and_b32 %1, %2, %3\n") : "+v"(u0), "+v"(u1) : "v"(u2), "v"(u3));
        if (OP == 11) asm volatile(REP64("v_pk_fma_f32 %0, %1, %2, %3\n v_pk_fma_f32 %1, %2, %3, %0\n") : "+v"(p0), "+v"(p1) : "v"(p2), "v"(p3));
        if (OP == 12) asm volatile(REP64("v_mov_b32 %0, %2\n v_mov_b32 %1, %3\n") : "+v"(u0), "+v"(u1) : "v"(u2), "v"(u3));
        // LDS reads: 16 per wait
        if (OP == 20) asm volatile(REP8(REP8("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:512\n") "s_waitcnt lgkmcnt(0)\n") : "+v"(p0), "+v"(p1) : "v"(a_lin8));
        if (OP == 21) asm volatile(REP8(REP8("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:8\n") "s_waitcnt lgkmcnt(0)\n") : "+v"(p0), "+v"(p1) : "v"(a_blk8));
        if (OP == 22) asm volatile(REP8(REP8("ds_read_b32 %0, %2\n ds_read_b32 %1, %2 offset:256\n") "s_waitcnt lgkmcnt(0)\n") : "+v"(a0), "+v"(a1) : "v"(a_lin4));
        if (OP == 23) asm volatile(REP8(REP8("ds_read_b32 %0, %2\n ds_read_b32 %1, %2 offset:4\n") "s_waitcnt lgkmcnt(0)\n") : "+v"(a0), "+v"(a1) : "v"(a_blk4));
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float r = a0 + a1 + p0.x + p0.y + p1.x + p1.y + (float)(u0 + u1) + (float)(w0 + w1);
    if (lane == 0) out[(size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (r == 1.2345f) out[0] = 0;
}

template <int OP>
void run(const char *name, int instr_per_rep = 128) {
    uint64_t *d;
    const int blocks = 256;
    hipMalloc(&d, blocks * 16 * sizeof(uint64_t));
    printf("%-34s", name);
    for (int threads : {256, 512, 1024}) {  // 1, 2, 4 waves per SIMD (one workgroup per CU)
        hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(threads), 0, 0, d, 1.0f);
        hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(threads), 0, 0, d, 2.0f);
        hipDeviceSynchronize();
        std::vector<uint64_t> h(blocks * threads / 64);
        hipMemcpy(h.data(), d, h.size() * sizeof(uint64_t), hipMemcpyDeviceToHost);
        double sum = 0;
        for (auto v : h) sum += (double)v;
        const double cyc_per_wave = sum / h.size();
        const double per_instr_wave = cyc_per_wave / ((double)LOOPS * instr_per_rep);
        const int waves_per_simd = threads / 256;
        printf("  %dw/SIMD: %6.2f cyc/instr/wave = %5.2f cyc/instr/SIMD", waves_per_simd, per_instr_wave,
               per_instr_wave / waves_per_simd);
    }
    printf("\n");
    hipFree(d);
}

int main() {
    run<0>("v_mul_f32");
    run<1>("v_pk_mul_f32");
    run<2>("v_pk_add_f32");
    run<3>("v_fma_f32");
    run<11>("v_pk_fma_f32");
    run<4>("v_mad_u32_u24");
    run<5>("v_mul_lo_u32");
    run<6>("v_cndmask_b32");
    run<7>("v_mad_u64_u32");
    run<8>("v_lshl_add_u64");
    run<9>("v_cvt_i32_f32 / v_cvt_f32_i32");
    run<10>("v_max_u32 / v_and_b32");
    run<12>("v_mov_b32");
    run<20>("ds_read_b64 conflict-free");
    run<21>("ds_read_b64 4x4x4 cell block");
    run<22>("ds_read_b32 conflict-free");
    run<23>("ds_read_b32 4x4x4 cell block");
    return 0;
}
