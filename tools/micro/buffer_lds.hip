// buffer_load_dwordx4 ... lds on gfx950: out-of-range lanes (voffset >= num_records) must land as zeros in LDS, soffset moves the window.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(const float* src, float* out, int bytes, int so) {
    __shared__ __attribute__((aligned(16))) float lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = -1.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, bytes, 0x00020000);
    const unsigned voff = (threadIdx.x % 5 == 3) ? 0x7ffffff0u : threadIdx.x * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, so, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}

int main() {
    const int n = 1024;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)(i + 1);
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 256 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int so : {0, 1024}) {
        // window of 300 floats: lanes whose 16 bytes start at or beyond 1200 bytes are out of range as well
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, so == 0 ? 1200 : 4096, so);
        std::vector<float> r(256);
        hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < 64; ++t)
            for (int e = 0; e < 4; ++e) {
                const bool oob = (t % 5 == 3) || (so == 0 && t * 16 + e * 4 >= 1200);
                const float want = oob ? 0.f : (float)(so / 4 + t * 4 + e + 1);
                if (r[t * 4 + e] != want) { if (bad < 6) printf("  lane %d elem %d: got %g want %g\n", t, e, r[t * 4 + e], want); ++bad; }
            }
        printf("soffset %d: %d mismatches\n", so, bad);
    }
    return 0;
}
