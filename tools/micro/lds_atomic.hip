// LDS atomic throughput on gfx950: ds_add_f32 / ds_add_u32 / ds_add_u64 / ds_add_f64(?) with conflict-free and random addresses.
// hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/micro/lds_atomic.hip -o tools/micro/lds_atomic && tools/micro/lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int KIND, bool RANDOM>
__global__ __launch_bounds__(512) void k(unsigned* out, int iters) {
    __shared__ unsigned long long cells[9216];                 // 72 KB
    for (int i = threadIdx.x; i < 9216; i += 512) cells[i] = 0;
    __syncthreads();
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            unsigned idx;
            if (RANDOM) { x = x * 1664525u + 1013904223u; idx = (x >> 8) % 9000u; }
            else idx = (threadIdx.x + u * 512 + it * 64) % 9000u;
            if (KIND == 0) atomicAdd(reinterpret_cast<float*>(cells) + idx, 1.0f);
            if (KIND == 1) atomicAdd(reinterpret_cast<unsigned*>(cells) + idx, 1u);
            if (KIND == 2) atomicAdd(cells + idx, 1ull);
            if (KIND == 3) atomicAdd(reinterpret_cast<double*>(cells) + idx, 1.0);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (unsigned)cells[5];
}

template <int KIND, bool RANDOM>
void run(const char* name, unsigned* out) {
    const int iters = 256, blocks = 512;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, RANDOM>), dim3(blocks), dim3(512), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, RANDOM>), dim3(blocks), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 512 * iters * 8;
    printf("%-28s %8.1f us  %7.1f G lane-atomics/s  (%.2f lanes/clk/CU at 2.4 GHz, 256 CUs)\n", name, ms * 1e3, n / ms / 1e6, n / (ms * 1e-3) / 2.4e9 / 256);
}

int main() {
    unsigned* out;
    hipMalloc(&out, 4096);
    run<0, false>("ds_add_f32 conflict-free", out);
    run<0, true>("ds_add_f32 random", out);
    run<1, false>("ds_add_u32 conflict-free", out);
    run<1, true>("ds_add_u32 random", out);
    run<2, false>("ds_add_u64 conflict-free", out);
    run<2, true>("ds_add_u64 random", out);
    run<3, false>("ds_add_f64 conflict-free", out);
    run<3, true>("ds_add_f64 random", out);
    return 0;
}
