// Microbenchmark (gfx950): does a wave's f32 VALU work overlap its own / its SIMD partner's f32-input MFMAs?
// One block of 256 (1 wave per SIMD) or 512 threads (2 per SIMD) per CU; each wave runs LOOPS x {NM mfma, NV v_fma}.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NM, int NV, int ROLE>   // ROLE 0: every wave runs both; 1: waves 0-3 mfma only, waves 4-7 valu only
__global__ void k(float* out, int loops, unsigned long long* cyc) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    const float a = threadIdx.x * 0.5f, b = 1.0001f;
    const int wave = threadIdx.x >> 6;
    const bool do_m = ROLE == 0 || wave < 4, do_v = ROLE == 0 || wave >= 4;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < loops; ++l) {
        if (do_m) {
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 7], 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], b, a);
        }
        if (ROLE == 0) {   // interleave hint: let the scheduler mix them
#pragma unroll
            for (int m = 0; m < NM; ++m) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, (NV + NM - 1) / (NM > 0 ? NM : 1), 0); }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int NM, int NV, int ROLE>
void run(const char* name, int threads) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    hipMemset(cyc, 0, 256 * 8 * 8);
    const int loops = 2000;
    hipLaunchKernelGGL((k<NM, NV, ROLE>), dim3(256), dim3(threads), 0, 0, out, loops, cyc);
    hipLaunchKernelGGL((k<NM, NV, ROLE>), dim3(256), dim3(threads), 0, 0, out, loops, cyc);
    hipDeviceSynchronize();
    unsigned long long h[256 * 8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s0 = 0, s4 = 0; int n = 0;
    for (int b = 0; b < 256; ++b) { s0 += h[b * 8]; s4 += h[b * 8 + (threads > 256 ? 4 : 0)]; ++n; }
    printf("%-44s threads %3d: cycles/iter wave0 %.1f  wave4 %.1f   (NM=%d x32 = %d, NV=%d x4 = %d)\n", name, threads, s0 / n / loops, s4 / n / loops, NM, NM * 32, NV, NV * 4);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<16, 0, 0>("mfma only", 256);
    run<0, 48, 0>("valu only", 256);
    run<16, 48, 0>("same wave: 16 mfma + 48 fma", 256);
    run<16, 96, 0>("same wave: 16 mfma + 96 fma", 256);
    run<16, 0, 0>("2 waves/SIMD mfma only", 512);
    run<0, 48, 0>("2 waves/SIMD valu only", 512);
    run<16, 48, 0>("2 waves/SIMD both do 16 mfma + 48 fma", 512);
    run<16, 128, 1>("partner split: w0-3 16 mfma | w4-7 128 fma", 512);
    run<16, 64, 1>("partner split: w0-3 16 mfma | w4-7 64 fma", 512);
    return 0;
}
