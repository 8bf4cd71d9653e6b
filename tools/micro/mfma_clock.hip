// Microbenchmark (gfx950): what a dense MFMA loop sustains with EVERY CU issuing - cycles per MFMA (s_memtime), the clock the chip
// holds meanwhile (s_memtime ticks per s_memrealtime tick x 100 MHz) and the resulting TFLOP/s - for the fp32 MFMA, the bf16 MFMAs of
// both shapes, and the bf16 MFMA with LDS operand reads.  Random operands (a clock measured on zeros reads high), one or two waves per SIMD.
// hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_clock.hip -o tools/micro/mfma_clock && tools/micro/mfma_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: v_mfma_f32_16x16x4_f32   1: v_mfma_f32_32x32x2_f32   2: v_mfma_f32_16x16x32_bf16   3: v_mfma_f32_32x32x16_bf16
// 4: 16x16x32 bf16 with one ds_read_b128 per two MFMAs
template <int MODE>
__global__ void loop(const unsigned* seed, float* out, int iters, unsigned long long* stamps) {
    __shared__ u32x4 lds[256];
    const int tid = threadIdx.x;
    // random bf16 pairs in [1, 2) x sign: finite, non-trivial bit patterns
    unsigned s = seed[(blockIdx.x * blockDim.x + tid) & 65535];
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s & 0x807f807fu) | 0x3f803f80u; };
    u32x4 a = {rnd(), rnd(), rnd(), rnd()}, b[4];
    for (int q = 0; q < 4; ++q) b[q] = u32x4{rnd(), rnd(), rnd(), rnd()};
    if (tid < 256) lds[tid] = u32x4{rnd(), rnd(), rnd(), rnd()};
    float fa = __uint_as_float((rnd() & 0x807fffffu) | 0x3f800000u), fb = __uint_as_float((rnd() & 0x807fffffu) | 0x3f800000u);
    f32x4 acc4[4];
    f32x16 acc16[2];
    for (int q = 0; q < 4; ++q) acc4[q] = f32x4{0, 0, 0, 0};
    for (int r = 0; r < 16; ++r) { acc16[0][r] = 0.f; acc16[1][r] = 0.f; }
    const unsigned ladr = (tid & 63) * 16;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (MODE == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc4[m & 3]) : "v"(fa), "v"(fb));
            if (MODE == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc16[m & 1]) : "v"(fa), "v"(fb));
            if (MODE == 2 || MODE == 4) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc4[m & 3]) : "v"(a), "v"(b[m & 3]));
            if (MODE == 3) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc16[m & 1]) : "v"(a), "v"(b[m & 3]));
            if (MODE == 4 && (m & 1)) asm volatile("ds_read_b128 %0, %1" : "=v"(b[(m + 2) & 3]) : "v"(ladr));
        }
        if (MODE == 4) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    asm volatile("s_nop 15\n s_nop 15");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    for (int q = 0; q < 4; ++q) sum += acc4[q][0] + acc4[q][3];
    sum += acc16[0][0] + acc16[1][5];
    out[blockIdx.x * blockDim.x + tid] = sum;
    if ((tid & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (tid >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

template <int MODE>
void run(const char* name, double flop_per_mfma, int threads, int blocks, const unsigned* seed, float* out, unsigned long long* st) {
    const int iters = 20000;
    const int waves = blocks * threads / 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {                        // (the third run's numbers: clocks settle over the first ones)
        hipEventRecord(e0);
        hipLaunchKernelGGL(loop<MODE>, dim3(blocks), dim3(threads), 0, 0, seed, out, iters, st);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * waves);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc(waves), ghz(waves);
    for (int w = 0; w < waves; ++w) { cyc[w] = (double)h[2 * w] / (iters * 16.0); ghz[w] = (double)h[2 * w] / ((double)h[2 * w + 1] * 10.0) ; }   // ticks per 10 ns
    std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
    const double tf = flop_per_mfma * 16.0 * iters * waves / (ms * 1e-3) / 1e12;
    printf("%-44s %d waves/SIMD: %6.2f cycles per MFMA per wave (median), clock %.2f GHz (median; min %.2f max %.2f), %7.1f TFLOP/s, %.2f ms\n", name,
           threads / 256, cyc[waves / 2], ghz[waves / 2], ghz[0], ghz[waves - 1], tf, ms);
}

int main() {
    unsigned* seed; float* out; unsigned long long* st;
    std::vector<unsigned> hs(65536);
    srand(7);
    for (auto& v : hs) v = (unsigned)rand() * 2654435761u + (unsigned)rand();
    hipMalloc(&seed, 65536 * 4); hipMemcpy(seed, hs.data(), 65536 * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 512 * 4 * 2); hipMalloc(&st, 256 * 8 * 2 * 8 * 2);
    for (int threads : {256, 512}) {
        run<0>("v_mfma_f32_16x16x4_f32", 2.0 * 16 * 16 * 4, threads, 256, seed, out, st);
        run<1>("v_mfma_f32_32x32x2_f32", 2.0 * 32 * 32 * 2, threads, 256, seed, out, st);
        run<2>("v_mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32, threads, 256, seed, out, st);
        run<3>("v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16, threads, 256, seed, out, st);
        run<4>("v_mfma_f32_16x16x32_bf16 + ds_read_b128 / 2", 2.0 * 16 * 16 * 32, threads, 256, seed, out, st);
    }
    return 0;
}
