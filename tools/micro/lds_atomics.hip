// Microbenchmark: LDS atomic throughput on gfx950 by type and address pattern (development tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 2048
#define TILE 5832

template <int MODE, int PATTERN>
__global__ __launch_bounds__(256) void k(float *out, int seed) {
    __shared__ float tile[TILE * 2];
    for (int c = threadIdx.x; c < TILE * 2; c += 256) tile[c] = 0.f;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + seed;
    int lane = threadIdx.x & 63;
    for (int it = 0; it < ITERS; ++it) {
        int idx;
        if (PATTERN == 0) idx = (threadIdx.x + it * 67) % TILE;                 // conflict-free, consecutive lanes
        else if (PATTERN == 1) { s = s * 1664525u + 1013904223u; idx = (s >> 8) % TILE; }  // random
        else if (PATTERN == 2) idx = ((lane >> 2) + it * 131) % TILE;             // 4 lanes per address
        else idx = ((lane & 3) + ((lane >> 2) & 3) * 18 + (lane >> 4) * 324 + (it * 7) % 4000) % TILE;  // 4x4x4 block of the 18^3 tile
        if (MODE == 0) atomicAdd(&tile[idx], 1.0f);
        else if (MODE == 1) atomicAdd((int *)&tile[idx], 1);
        else if (MODE == 2) atomicAdd((unsigned long long *)&tile[2 * idx], 1ull);
        else if (MODE == 3) tile[idx] += 1.0f;  // non-atomic RMW (racy, reference rate)
        else if (MODE == 4) atomicAdd((double *)&tile[2 * idx], 1.0);
    }
    __syncthreads();
    float acc = 0;
    for (int c = threadIdx.x; c < TILE * 2; c += 256) acc += tile[c];
    if (acc == 12345.678f) out[0] = acc;
}

template <int MODE, int PATTERN>
void run(const char *name) {
    float *out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int blocks = 256 * 4;
    hipLaunchKernelGGL((k<MODE, PATTERN>), dim3(blocks), dim3(256), 0, 0, out, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, PATTERN>), dim3(blocks), dim3(256), 0, 0, out, 2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * ITERS;
    printf("%-34s %8.3f ms  %8.2f Gops/s  %6.2f lane-ops/clk/CU @2.4GHz\n", name, ms, ops / ms / 1e6,
           ops / (ms * 1e-3) / 256 / 2.4e9);
    hipFree(out);
}

int main() {
    run<0, 0>("f32 add, consecutive");
    run<0, 1>("f32 add, random");
    run<0, 2>("f32 add, 4 lanes/address");
    run<0, 3>("f32 add, 4x4x4 cell block");
    run<1, 0>("i32 add, consecutive");
    run<1, 1>("i32 add, random");
    run<1, 2>("i32 add, 4 lanes/address");
    run<1, 3>("i32 add, 4x4x4 cell block");
    run<2, 0>("u64 add, consecutive");
    run<2, 1>("u64 add, random");
    run<2, 3>("u64 add, 4x4x4 cell block");
    run<4, 0>("f64 add, consecutive");
    run<4, 1>("f64 add, random");
    run<3, 0>("plain RMW, consecutive");
    run<3, 1>("plain RMW, random");
    return 0;
}
