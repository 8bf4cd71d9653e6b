// HBM read rate of LDS-DMA against the CONTIGUOUS RUN a block reads: every block streams its own region, but as runs of P bytes at a
// pitch of 5120 bytes (a 1280-float image row) - the pattern of a tile's rows - instead of one contiguous range.  What do the encoder's
// 528-byte (conv_enc1.hip), 544-byte (conv_wino4.hip at C = 16) and 128 / 160-byte (wgrad_enc.hip) row pieces cost?
//   hipcc -O3 --offload-arch=gfx950 tools/micro/dma_pieces.hip -o tools/micro/dma_pieces
#include <hip/hip_runtime.h>
#include <cstdio>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// a block reads `rows` runs of P bytes per iteration (NDMA x 4 KB in all), run r of iteration it at base + (it * rows + r) * pitch
template <int NDMA>
__global__ __launch_bounds__(256) void k(const char* __restrict__ src, float* __restrict__ out, int iters, int P, size_t pitch, size_t region) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int STAGE = 4 * NDMA * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = src + (size_t)blockIdx.x * region;
    const int ppr = P / 16;                                           // 16-byte pieces per run
    for (int it = 0; it < iters; ++it) {
        float* st = lds + (it & 1) * STAGE;
#pragma unroll
        for (int q = 0; q < NDMA; ++q) {
            const int piece = (q * 4 + wave) * 64 + lane;             // 0 .. NDMA * 256 - 1 of this iteration
            const int r = piece / ppr, c = piece - r * ppr;
            const char* gp = base + ((size_t)it * (NDMA * 256 / ppr) + r) * pitch + (size_t)c * 16;
            __builtin_amdgcn_global_load_lds(GLB_PTR(gp), LDS_PTR(st + (q * 4 + wave) * 256), 16, 0, 0);
        }
        if (it > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");       // the iteration before has landed
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lds[threadIdx.x] == 12345.f) out[threadIdx.x] = 1.f;
}

int main() {
    constexpr int NDMA = 8;
    const int blocks = 512, iters = 16;
    const size_t bytes_per_block = (size_t)iters * NDMA * 4096;
    float* out;
    hipMalloc(&out, 4096);
    hipFuncSetAttribute((const void*)k<NDMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("512 blocks x 16 iterations x 32 KB = %.0f MB read per launch, two blocks per CU, one iteration ahead\n", blocks * bytes_per_block / 1e6);
    for (int P : {64, 128, 256, 512, 1024, 2048, 4096}) {
        for (int mode = 0; mode < 2; ++mode) {
            // mode 0: pitch = P (contiguous region per block); mode 1: runs of P bytes at a 5120-byte pitch (tile rows of a 1280-wide fp32 image)
            const size_t pitch = mode == 0 ? (size_t)P : 5120;
            if (mode == 1 && P > 5120) continue;
            const size_t region = (bytes_per_block / P) * pitch;
            char* src;
            const size_t total = (size_t)blocks * region;
            if (hipMalloc(&src, total) != hipSuccess) { printf("P %d: allocation of %.1f GB failed\n", P, total / 1e9); continue; }
            hipMemset(src, 0, total);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k<NDMA>, dim3(blocks), dim3(256), 2 * NDMA * 4096, 0, src, out, iters, P, pitch, region);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<NDMA>, dim3(blocks), dim3(256), 2 * NDMA * 4096, 0, src, out, iters, P, pitch, region);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("runs of %4d B %s: %7.1f us  %5.2f TB/s\n", P, mode == 0 ? "back to back      " : "at a 5120 B pitch ", ms * 1e3, blocks * bytes_per_block / (ms * 1e-3) / 1e12);
            hipFree(src);
        }
    }
    return 0;
}
