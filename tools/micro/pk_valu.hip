// Microbenchmark (gfx950): issue cost of packed f32 VALU (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mov_b32) against scalar v_fma_f32,
// one wave per SIMD (256 threads) or two (512), alone and beside f32-input MFMAs (v_mfma_f32_16x16x4_f32).
// hipcc -O3 --offload-arch=gfx950 tools/micro/pk_valu.hip -o tools/micro/pk_valu && tools/micro/pk_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// MODE 0: NV scalar v_fma, CH independent chains; 1: NV v_pk_fma_f32; 2: NV v_pk_add_f32; 3: NV v_pk_mov_b32 (op_sel transposes)
template <int NM, int NV, int CH, int MODE>
__global__ void k(float* out, int loops, unsigned long long* cyc) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f - i};
    const float a = threadIdx.x * 0.5f, b = 1.0001f;
    const f32x2 a2 = {a, a + 1.f}, b2 = {b, b};
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < loops; ++l) {
        constexpr int PER = NM > 0 ? (NV + NM - 1) / NM : NV;          // VALU between two MFMAs (volatile asm keeps the order)
        constexpr int GROUPS = NM > 0 ? NM : 1;
#pragma unroll
        for (int m = 0; m < GROUPS; ++m) {
            if constexpr (NM > 0) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "v"(b));
            }
#pragma unroll
            for (int qq = 0; qq < PER; ++qq) {
                const int q = m * PER + qq;
                if (q >= NV) break;
                f32x2& x = v[q % CH];
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(b), "v"(a));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b2), "v"(a2));
                if (MODE == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(a2));
                if (MODE == 3) asm volatile("v_pk_mov_b32 %0, %0, %1 op_sel:[1,0]" : "+v"(x) : "v"(a2));
                if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(b2));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i][0] + v[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int NM, int NV, int CH, int MODE>
void run(const char* name, int threads) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    hipMemset(cyc, 0, 256 * 8 * 8);
    const int loops = 2000;
    hipLaunchKernelGGL((k<NM, NV, CH, MODE>), dim3(256), dim3(threads), 0, 0, out, loops, cyc);
    hipLaunchKernelGGL((k<NM, NV, CH, MODE>), dim3(256), dim3(threads), 0, 0, out, loops, cyc);
    hipDeviceSynchronize();
    unsigned long long h[256 * 8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s0 = 0, s4 = 0;
    for (int b = 0; b < 256; ++b) { s0 += h[b * 8]; s4 += h[b * 8 + (threads > 256 ? 4 : 0)]; }
    printf("%-40s thr %3d NM %2d NV %2d chains %d: cycles/iter wave0 %7.1f wave4 %7.1f  -> %.2f cyc per VALU beyond 32/MFMA\n", name, threads, NM, NV, CH,
           s0 / 256 / loops, s4 / 256 / loops, NV ? (s0 / 256 / loops - 32.0 * NM) / NV : 0.0);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0, 48, 8, 0>("v_fma_f32 x48, 8 chains", 256);
    run<0, 48, 1, 0>("v_fma_f32 x48, 1 chain (dependent)", 256);
    run<0, 48, 2, 0>("v_fma_f32 x48, 2 chains", 256);
    run<0, 48, 4, 0>("v_fma_f32 x48, 4 chains", 256);
    run<0, 48, 8, 1>("v_pk_fma_f32 x48, 8 chains", 256);
    run<0, 48, 1, 1>("v_pk_fma_f32 x48, 1 chain", 256);
    run<0, 48, 8, 2>("v_pk_add_f32 x48, 8 chains", 256);
    run<0, 48, 8, 4>("v_pk_mul_f32 x48, 8 chains", 256);
    run<0, 48, 8, 3>("v_pk_mov_b32 x48, 8 chains", 256);
    run<0, 48, 8, 0>("2/SIMD v_fma_f32 x48, 8 chains", 512);
    run<0, 48, 1, 0>("2/SIMD v_fma_f32 x48, 1 chain", 512);
    run<0, 48, 8, 1>("2/SIMD v_pk_fma_f32 x48, 8 chains", 512);
    run<16, 0, 8, 0>("mfma only", 256);
    run<16, 48, 8, 0>("16 mfma + 48 v_fma (8 chains)", 256);
    run<16, 48, 2, 0>("16 mfma + 48 v_fma (2 chains)", 256);
    run<16, 24, 8, 1>("16 mfma + 24 v_pk_fma (8 chains)", 256);
    run<16, 48, 8, 1>("16 mfma + 48 v_pk_fma (8 chains)", 256);
    run<16, 24, 8, 2>("16 mfma + 24 v_pk_add (8 chains)", 256);
    run<16, 24, 8, 3>("16 mfma + 24 v_pk_mov (8 chains)", 256);
    run<16, 48, 8, 0>("2/SIMD 16 mfma + 48 v_fma", 512);
    run<16, 24, 8, 1>("2/SIMD 16 mfma + 24 v_pk_fma", 512);
    return 0;
}
