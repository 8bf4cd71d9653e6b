// Where do the blocks of a small launch land?  dispatch_map <blocks> <threads> <lds bytes>: histogram of blocks per CU.
// (HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID register: xcc_id [3:0])
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void where(unsigned* out, int spin) {
    extern __shared__ float pad[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float v = threadIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;     // keep the block alive so that later blocks see occupied CUs
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
    if (v == 12345.f) pad[0] = v;
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 200, threads = argc > 2 ? atoi(argv[2]) : 256, lds = argc > 3 ? atoi(argv[3]) : 0;
    unsigned* d;
    hipMalloc(&d, blocks * 8);
    if (lds > 65536) hipFuncSetAttribute(reinterpret_cast<const void*>(&where), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(where, dim3(blocks), dim3(threads), lds, 0, d, 200000);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
    std::vector<unsigned> h(2 * blocks);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_cu;
    for (int b = 0; b < blocks; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15;
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
    }
    std::map<int, int> hist;
    for (auto& kv : per_cu) hist[kv.second]++;
    printf("%d blocks x %d threads, %d B dynamic LDS: %zu distinct CUs;", blocks, threads, lds, per_cu.size());
    for (auto& kv : hist) printf(" %d CUs with %d block(s);", kv.second, kv.first);
    printf("\n");
    return 0;
}
