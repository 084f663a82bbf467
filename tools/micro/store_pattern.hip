// Micro-benchmark: HBM write traffic (rocprofv3 --pmc WRITE_SIZE) and time of three ways to write a fresh
// [np,3] float array: 12 bytes per lane (global_store_dwordx3), three dword stores per lane, and 16-byte
// aligned float4 stores.   hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void store_x3(float *__restrict__ out, int64_t np) {
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < np; p += (int64_t)gridDim.x * 256) {
        const float v = (float)p;
        out[3 * p] = v, out[3 * p + 1] = v + 1.f, out[3 * p + 2] = v + 2.f;
    }
}
__global__ __launch_bounds__(256) void store_3x1(float *__restrict__ out, int64_t np) {
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < np; p += (int64_t)gridDim.x * 256) {
        const float v = (float)p;
        __builtin_nontemporal_store(v, out + 3 * p);
        __builtin_nontemporal_store(v + 1.f, out + 3 * p + 1);
        __builtin_nontemporal_store(v + 2.f, out + 3 * p + 2);
    }
}
__global__ __launch_bounds__(256) void store_x4(float4 *__restrict__ out, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float v = (float)i;
        out[i] = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
    }
}
// tile-range-like: each workgroup writes a contiguous range of 2048 particles (as the gather does)
__global__ __launch_bounds__(256) void store_x3_ranges(float *__restrict__ out, int64_t np) {
    const int64_t s = (int64_t)blockIdx.x * 2048, e = s + 2048 < np ? s + 2048 : np;
    for (int64_t p = s + threadIdx.x; p < e; p += 256) {
        const float v = (float)p;
        out[3 * p] = v, out[3 * p + 1] = v + 1.f, out[3 * p + 2] = v + 2.f;
    }
}

int main() {
    const int64_t np = 1LL << 28;  // 3.2 GB
    float *out;
    hipMalloc(&out, np * 12);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        store_x3<<<8192, 256>>>(out, np);
        hipEventRecord(e1), hipEventSynchronize(e1), hipEventElapsedTime(&ms, e0, e1);
        printf("dwordx3 grid-stride   %7.3f ms  %7.1f GB/s\n", ms, np * 12 / ms / 1e6);
        hipEventRecord(e0);
        store_3x1<<<8192, 256>>>(out, np);
        hipEventRecord(e1), hipEventSynchronize(e1), hipEventElapsedTime(&ms, e0, e1);
        printf("3 x dword nontemporal %7.3f ms  %7.1f GB/s\n", ms, np * 12 / ms / 1e6);
        hipEventRecord(e0);
        store_x4<<<8192, 256>>>((float4 *)out, np * 3 / 4);
        hipEventRecord(e1), hipEventSynchronize(e1), hipEventElapsedTime(&ms, e0, e1);
        printf("float4                %7.3f ms  %7.1f GB/s\n", ms, np * 12 / ms / 1e6);
        hipEventRecord(e0);
        store_x3_ranges<<<(unsigned)(np / 2048), 256>>>(out, np);
        hipEventRecord(e1), hipEventSynchronize(e1), hipEventElapsedTime(&ms, e0, e1);
        printf("dwordx3 2048-ranges   %7.3f ms  %7.1f GB/s\n", ms, np * 12 / ms / 1e6);
    }
    hipFree(out);
    return 0;
}
