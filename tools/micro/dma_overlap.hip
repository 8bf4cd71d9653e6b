// Does LDS-DMA (global_load_lds_dwordx4) make progress underneath a CU's MFMA work?  Every kernel of this library that is fed by LDS-DMA
// double-buffers "request tile i+1, multiply tile i" - and two of them measure as the SUM of their operand delivery and their multiplies
// (wgrad_enc_kernel: 90 us without MFMAs + 58 us of MFMAs = 159 us; conv_enc12.hip's phase A).  Per iteration a 4-wave block requests
// BYTES of fresh HBM data into the LDS stage it is not using and issues NM MFMAs per wave whose operands come (a) from registers,
// (b) from the other LDS stage (ds_read_b32 each, as the kernels do), then waits and meets at a barrier.  Two blocks per CU.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/dma_overlap.hip -o tools/micro/dma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int NDMA, int NM, int MODE>      // NDMA: 1 KB wave-instructions per wave and iteration; MODE 0: no MFMA, 1: register operands, 2: LDS operands
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* __restrict__ out, int iters, size_t stride_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int STAGE = 4 * (NDMA > 0 ? NDMA : 1) * 256;           // floats
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* base = src + (size_t)blockIdx.x * stride_floats;
    f32x4 acc[4] = {};
    float a = (float)lane * 1e-3f, b = 1.f;
    auto issue = [&](int it) {
        float* st = lds + (it & 1) * STAGE;
#pragma unroll
        for (int q = 0; q < NDMA; ++q) {
            const float* gp = base + ((size_t)it * NDMA * 4 + q * 4 + wave) * 256 + lane * 4;
            __builtin_amdgcn_global_load_lds(GLB_PTR(gp), LDS_PTR(st + (q * 4 + wave) * 256), 16, 0, 0);
        }
    };
    if (NDMA > 0) issue(0);
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (NDMA > 0 && it + 1 < iters) issue(it + 1);
        const float* cur = lds + (it & 1) * STAGE;
        if (MODE == 1) {
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
        } else if (MODE == 2) {
#pragma unroll
            for (int m0 = 0; m0 < NM; m0 += 16) {
                float v[16];
#pragma unroll
                for (int m = 0; m < 16; ++m) v[m] = cur[(m0 + m) * 64 % (STAGE - 64) + lane];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 16; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[m], b, acc[m & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s[0] == 12345.f) out[threadIdx.x] = s[1];
}

// the same traffic through REGISTERS: global_load_dwordx4 at the top of the iteration, the MFMAs, then ds_write_b128 of what arrived
template <int NDMA, int NM>
__global__ __launch_bounds__(256) void kreg(const float* __restrict__ src, float* __restrict__ out, int iters, size_t stride_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int STAGE = 4 * NDMA * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* base = src + (size_t)blockIdx.x * stride_floats;
    f32x4 acc[4] = {};
    float a = (float)lane * 1e-3f, b = 1.f;
    f32x4 r[NDMA];
    auto load = [&](int it) {
#pragma unroll
        for (int q = 0; q < NDMA; ++q) r[q] = *reinterpret_cast<const f32x4*>(base + ((size_t)it * NDMA * 4 + q * 4 + wave) * 256 + lane * 4);
    };
    load(0);
    for (int it = 0; it < iters; ++it) {
        float* st = lds + (it & 1) * STAGE;
#pragma unroll
        for (int q = 0; q < NDMA; ++q) *reinterpret_cast<f32x4*>(st + (q * 4 + wave) * 256 + lane * 4) = r[q];   // (waits for the loads)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (it + 1 < iters) load(it + 1);
        __builtin_amdgcn_sched_barrier(0);
        const float* cur = st;
#pragma unroll
        for (int m0 = 0; m0 < NM; m0 += 16) {
            float v[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) v[m] = cur[(m0 + m) * 64 % (STAGE - 64) + lane];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 16; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[m], b, acc[m & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s[0] == 12345.f) out[threadIdx.x] = s[1];
}

template <int NDMA, int NM>
float runreg(const float* src, float* out, int blocks, int iters, size_t stride) {
    const int lds = 2 * 4 * NDMA * 1024;
    hipFuncSetAttribute((const void*)kreg<NDMA, NM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kreg<NDMA, NM>), dim3(blocks), dim3(256), lds, 0, src, out, iters, stride);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kreg<NDMA, NM>), dim3(blocks), dim3(256), lds, 0, src, out, iters, stride);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

template <int NDMA, int NM, int MODE>
float run(const float* src, float* out, int blocks, int iters, size_t stride) {
    const int lds = 2 * 4 * (NDMA > 0 ? NDMA : 1) * 1024;
    hipFuncSetAttribute((const void*)k<NDMA, NM, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NDMA, NM, MODE>), dim3(blocks), dim3(256), lds, 0, src, out, iters, stride);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NDMA, NM, MODE>), dim3(blocks), dim3(256), lds, 0, src, out, iters, stride);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

int main() {
    const int blocks = 512, iters = 16;
    constexpr int NDMA = 8;                                           // 32 KB per block and iteration (wgrad_enc_kernel's stage)
    const size_t stride = (size_t)iters * NDMA * 4 * 256;             // floats per block: fresh data every iteration
    float *src, *out;
    hipMalloc(&src, (size_t)blocks * stride * 4);                     // 512 x 16 x 32 KB = 268 MB: more than the Infinity Cache
    hipMalloc(&out, 4096);
    hipMemset(src, 0, (size_t)blocks * stride * 4);
    const double mb = (double)blocks * iters * NDMA * 4 * 1024 / 1e6;
    printf("%d blocks x %d iterations x %d KB = %.0f MB per launch; MFMAs per wave and iteration: 144 (4.6k cycles) / 288\n", blocks, iters, NDMA * 4, mb);
    const float d0 = run<NDMA, 0, 0>(src, out, blocks, iters, stride);
    const float m1 = run<0, 144, 1>(src, out, blocks, iters, stride);
    const float m2 = run<0, 144, 2>(src, out, blocks, iters, stride);
    const float b1 = run<NDMA, 144, 1>(src, out, blocks, iters, stride);
    const float b2 = run<NDMA, 144, 2>(src, out, blocks, iters, stride);
    const float m1x = run<0, 288, 1>(src, out, blocks, iters, stride);
    const float b1x = run<NDMA, 288, 1>(src, out, blocks, iters, stride);
    const float b2x = run<NDMA, 288, 2>(src, out, blocks, iters, stride);
    printf("DMA only                      %7.1f us  (%.2f TB/s)\n", d0, mb / d0);
    printf("144 MFMAs, register operands  %7.1f us\n", m1);
    printf("144 MFMAs, LDS operands       %7.1f us\n", m2);
    printf("DMA + 144 MFMAs (registers)   %7.1f us   (sum %.1f, max %.1f)\n", b1, d0 + m1, d0 > m1 ? d0 : m1);
    printf("DMA + 144 MFMAs (LDS)         %7.1f us   (sum %.1f, max %.1f)\n", b2, d0 + m2, d0 > m2 ? d0 : m2);
    printf("288 MFMAs, register operands  %7.1f us\n", m1x);
    printf("DMA + 288 MFMAs (registers)   %7.1f us   (sum %.1f, max %.1f)\n", b1x, d0 + m1x, d0 > m1x ? d0 : m1x);
    printf("DMA + 288 MFMAs (LDS)         %7.1f us\n", b2x);
    const float r0 = runreg<NDMA, 0>(src, out, blocks, iters, stride);
    const float r1 = runreg<NDMA, 144>(src, out, blocks, iters, stride);
    const float r2 = runreg<NDMA, 288>(src, out, blocks, iters, stride);
    printf("through registers, no MFMAs   %7.1f us\n", r0);
    printf("through registers + 144 (LDS) %7.1f us\n", r1);
    printf("through registers + 288 (LDS) %7.1f us\n", r2);
    return 0;
}
