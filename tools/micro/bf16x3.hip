// Microbenchmark (gfx950): fp32 products as three bf16 pieces per operand (a = a0 + a1 + a2 exactly, 8 significand bits each, by
// truncation) and six bf16 MFMAs (a0b0, a0b1, a1b0, a0b2, a1b1, a2b0; the dropped terms are below 2^-24 of |a||b|).
//   (1) accuracy: a 32 x 32 x 144 product by v_mfma_f32_32x32x16_bf16 in that form against float64, beside the fp32 MFMA's error;
//   (2) issue rate of the bf16 MFMAs alone, with the ds_read_b128 operand reads, and with the VALU work of the split beside them.
// hipcc -O3 --offload-arch=gfx950 tools/micro/bf16x3.hip -o tools/micro/bf16x3 && tools/micro/bf16x3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float a, unsigned& p0, unsigned& p1, unsigned& p2) {
    const float a0 = __uint_as_float(__float_as_uint(a) & 0xffff0000u);
    const float r1 = a - a0;
    const float a1 = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
    const float r2 = r1 - a1;                                   // 8 bits left: a bf16 as it stands
    p0 = __float_as_uint(a0) >> 16; p1 = __float_as_uint(a1) >> 16; p2 = __float_as_uint(r2) >> 16;
}

// A [32][K] row-major, B [K][32] row-major, D [32][32]; one wave
template <int MODE>
__global__ void gemm(const float* A, const float* B, float* D, int K) {
    const int lane = threadIdx.x, i = lane & 31, g = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (MODE == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k + g], B[(k + g) * 32 + i], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            union { bf16x8 v; unsigned short s[8]; } a[3], b[3];
            for (int e = 0; e < 8; ++e) {
                unsigned p0, p1, p2;
                split3(A[i * K + k + 8 * g + e], p0, p1, p2); a[0].s[e] = p0; a[1].s[e] = p1; a[2].s[e] = p2;
                split3(B[(k + 8 * g + e) * 32 + i], p0, p1, p2); b[0].s[e] = p0; b[1].s[e] = p1; b[2].s[e] = p2;
            }
            // small terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2].v, b[0].v, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[1].v, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[2].v, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[0].v, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[1].v, acc, 0, 0, 0);
            if (MODE == 2) {                                    // all nine
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2].v, b[2].v, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2].v, b[1].v, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[2].v, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[0].v, acc, 0, 0, 0);
        }
    }
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * g) * 32 + i] = acc[r];
}

// NM MFMAs per iteration; NL ds_read_b128 and NV VALU ops spread between them
template <int SHAPE, int NM, int NL, int NV>
__global__ void rate(float* out, int loops, unsigned long long* cyc) {
    __shared__ u32x4 lds[1024];
    for (int t = threadIdx.x; t < 1024; t += blockDim.x) lds[t] = u32x4{(unsigned)t, 1u, 2u, 3u};
    f32x16 acc[2]; f32x4 acc4[4];
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    for (int r = 0; r < 4; ++r) acc4[r] = f32x4{0, 0, 0, 0};
    u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b[4] = {a, a, a, a};
    float v[8];
    for (int q = 0; q < 8; ++q) v[q] = threadIdx.x * 0.001f + q;
    const unsigned ladr = (threadIdx.x & 63) * 16;
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < loops; ++l) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (SHAPE == 32) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 1]) : "v"(a), "v"(b[m & 3]));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc4[m & 3]) : "v"(a), "v"(b[m & 3]));
            if ((m * NL) / NM != ((m + 1) * NL) / NM) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[m & 3]) : "v"(ladr), "n"(0));
#pragma unroll
            for (int q = (m * NV) / NM; q < ((m + 1) * NV) / NM; ++q) {
                if (q & 1) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(v[q & 7]));
                else asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[q & 7]) : "v"(v[(q + 1) & 7]));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
    for (int r = 0; r < 4; ++r) s += acc4[r][0] + acc4[r][3];
    for (int q = 0; q < 8; ++q) s += v[q];
    s += __uint_as_float(b[0][0]) + __uint_as_float(b[1][1]) + __uint_as_float(b[2][2]) + __uint_as_float(b[3][3]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int SHAPE, int NM, int NL, int NV>
void run(const char* name, int threads) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    hipMemset(cyc, 0, 256 * 8 * 8);
    const int loops = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((rate<SHAPE, NM, NL, NV>), dim3(256), dim3(threads), 0, 0, out, loops, cyc);
    hipDeviceSynchronize();
    unsigned long long h[256 * 8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s0 = 0;
    for (int b = 0; b < 256; ++b) s0 += h[b * 8];
    // s_memtime counts at 100 MHz on gfx950; the ratio to a known rate is what matters, so also print per-MFMA
    printf("%-52s waves/SIMD %d: %8.1f ticks/iter, %6.2f per MFMA\n", name, threads / 256, s0 / 256 / loops, s0 / 256 / loops / NM);
    hipFree(out); hipFree(cyc);
}

int main() {
    const int K = 144;
    float *hA = (float*)malloc(32 * K * 4), *hB = (float*)malloc(K * 32 * 4), hD[1024];
    srand(1);
    for (int t = 0; t < 32 * K; ++t) { hA[t] = (float)rand() / RAND_MAX * 2 - 1; hB[t] = ((float)rand() / RAND_MAX * 2 - 1) * (1 + (t % 7)); }
    float *A, *B, *D;
    hipMalloc(&A, 32 * K * 4); hipMalloc(&B, 32 * K * 4); hipMalloc(&D, 4096);
    hipMemcpy(A, hA, 32 * K * 4, hipMemcpyHostToDevice); hipMemcpy(B, hB, 32 * K * 4, hipMemcpyHostToDevice);
    const char* names[3] = {"fp32 MFMA 32x32x2", "bf16 x6 (32x32x16)", "bf16 x9 (32x32x16)"};
    for (int mode = 0; mode < 3; ++mode) {
        if (mode == 0) hipLaunchKernelGGL(gemm<0>, dim3(1), dim3(64), 0, 0, A, B, D, K);
        if (mode == 1) hipLaunchKernelGGL(gemm<1>, dim3(1), dim3(64), 0, 0, A, B, D, K);
        if (mode == 2) hipLaunchKernelGGL(gemm<2>, dim3(1), dim3(64), 0, 0, A, B, D, K);
        hipMemcpy(hD, D, 4096, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double s = 0, m = 0;
                for (int k = 0; k < K; ++k) { s += (double)hA[i * K + k] * hB[k * 32 + j]; m += fabs((double)hA[i * K + k] * hB[k * 32 + j]); }
                worst = fmax(worst, fabs(hD[i * 32 + j] - s) / m); scale = fmax(scale, m);
            }
        printf("%-22s K=%d: worst |error| / sum|a b| = %.3e   (2^-24 = 5.96e-08)\n", names[mode], K, worst);
    }
    run<32, 12, 0, 0>("32x32x16 bf16 x12", 256);
    run<32, 12, 0, 0>("32x32x16 bf16 x12", 512);
    run<16, 12, 0, 0>("16x16x32 bf16 x12", 256);
    run<16, 12, 0, 0>("16x16x32 bf16 x12", 512);
    run<32, 12, 6, 0>("32x32x16 x12 + 6 ds_read_b128", 256);
    run<32, 12, 6, 0>("32x32x16 x12 + 6 ds_read_b128", 512);
    run<32, 12, 12, 0>("32x32x16 x12 + 12 ds_read_b128", 512);
    run<32, 12, 0, 24>("32x32x16 x12 + 24 VALU", 256);
    run<32, 12, 0, 24>("32x32x16 x12 + 24 VALU", 512);
    run<32, 12, 0, 48>("32x32x16 x12 + 48 VALU", 512);
    run<32, 12, 6, 24>("32x32x16 x12 + 6 ds_read + 24 VALU", 512);
    run<16, 12, 6, 24>("16x16x32 x12 + 6 ds_read + 24 VALU", 512);
    run<16, 12, 12, 0>("16x16x32 x12 + 12 ds_read_b128", 512);
    return 0;
}
