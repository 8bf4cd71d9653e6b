#include <hip/hip_ext.h>

#include "eraft_kernels.h"

namespace {

__global__ __launch_bounds__(256) void pad_kernel(const float* __restrict__ in, float* __restrict__ out, int nc, int h, int w,
                                                  int left, int top, int oh, int ow) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)nc * oh * ow) return;
    const int x = idx % ow, y = (idx / ow) % oh;
    const long c = idx / ((long)ow * oh);
    const int sy = min(max(y - top, 0), h - 1), sx = min(max(x - left, 0), w - 1);
    out[idx] = in[(c * h + sy) * w + sx];
}

// four output pixels per thread (16-byte stores); the replicate clamp is per pixel (in1 != NULL: planes [nc, 2 nc) of the output come
// from in1 - both event volumes of a sample as one launch).  Block = 32 output rows x 32 threads along a row, rows on grid.y / grid.z: the
// first form took a flat 64-bit index apart with three 64-bit divisions per thread and ran at 3.9 TB/s of traffic on the training step's
// 234 MB (60 us of a 2.1 ms step) and 2.9 TB/s on EEMFlow+'s 1280x720 pair.
__global__ __launch_bounds__(256) void pad4_kernel(const float* __restrict__ in, float* __restrict__ out, int nc, int h, int w,
                                                   int left, int top, int oh, int ow, const float* __restrict__ in1) {
    const int ow4 = ow >> 2;
    const int x4 = blockIdx.x * 32 + (threadIdx.x & 31);
    const unsigned row0 = (blockIdx.y + blockIdx.z * gridDim.y) * 32u + (threadIdx.x >> 5);     // c * oh + y; the thread's rows: row0 + 8 k
    const unsigned rows = (unsigned)(in1 ? 2 * nc : nc) * (unsigned)oh;
    if (x4 >= ow4) return;
    int sx[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) sx[e] = min(max(x4 * 4 + e - left, 0), w - 1);
    // four rows per thread, all sixteen loads requested before the first store: a wave keeps 4 KB in flight instead of 1 (one row per
    // thread moved the training step's 234 MB at 4.0 TB/s: 32 waves x 1 KB per CU over a ~2 us round trip)
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned row = min(row0 + 8u * k, rows - 1);
        unsigned c = row / (unsigned)oh;
        const int y = (int)(row - c * (unsigned)oh);
        const float* base = in;
        if (in1 && c >= (unsigned)nc) { c -= nc; base = in1; }
        const int sy = min(max(y - top, 0), h - 1);
        const float* src = base + ((size_t)c * h + sy) * w;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[k][e] = src[sx[e]];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned row = row0 + 8u * k;
        if (row < rows) reinterpret_cast<f32x4*>(out)[(size_t)row * ow4 + x4] = v[k];
    }
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// one block per (n, c) plane; mean, then biased variance around the mean, then apply
__global__ __launch_bounds__(256) void instnorm_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                       const float* __restrict__ res, int hw, int relu_inner) {
    __shared__ float sh[4];
    const float* p = x + (size_t)blockIdx.x * hw;
    float s = 0.f;
    for (int i = threadIdx.x; i < hw; i += 256) s += p[i];
    const float mean = block_sum(s, sh) / (float)hw;
    float q = 0.f;
    for (int i = threadIdx.x; i < hw; i += 256) { const float d = p[i] - mean; q += d * d; }
    const float var = block_sum(q, sh) / (float)hw;
    const float rstd = 1.f / sqrtf(var + 1e-5f);
    float* o = out + (size_t)blockIdx.x * hw;
    const float* r = res ? res + (size_t)blockIdx.x * hw : nullptr;
    for (int i = threadIdx.x; i < hw; i += 256) {
        float v = (p[i] - mean) * rstd;
        if (relu_inner) v = v > 0.f ? v : 0.f;
        if (r) { v += r[i]; v = v > 0.f ? v : 0.f; }
        o[i] = v;
    }
}

// the same with the plane held in registers (1024 threads x up to NV float4): one read and one write of the plane
// instead of three reads; mean first, then the variance around it, as above
template <int NV>
__global__ __launch_bounds__(1024) void instnorm_reg_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                            const float* __restrict__ res, int hw, int relu_inner) {
    __shared__ float sh[16];
    const f32x4* p = reinterpret_cast<const f32x4*>(x + (size_t)blockIdx.x * hw);
    const int n4 = hw >> 2;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = k * 1024 + threadIdx.x;
        v[k] = i < n4 ? p[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
    }
    auto bsum = [&](float t) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) t += __shfl_xor(t, d);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = t;
        __syncthreads();
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) r += sh[k];
        return r;
    };
    const float mean = bsum(s) / (float)hw;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (k * 1024 + (int)threadIdx.x < n4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[k][e] - mean; q += d * d; }
        }
    }
    const float var = bsum(q) / (float)hw;
    const float rstd = 1.f / sqrtf(var + 1e-5f);
    f32x4* o = reinterpret_cast<f32x4*>(out + (size_t)blockIdx.x * hw);
    const f32x4* r = res ? reinterpret_cast<const f32x4*>(res + (size_t)blockIdx.x * hw) : nullptr;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = k * 1024 + threadIdx.x;
        if (i >= n4) continue;
        f32x4 rv = r ? r[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = (v[k][e] - mean) * rstd;
            if (relu_inner) t = t > 0.f ? t : 0.f;
            if (r) { t += rv[e]; t = t > 0.f ? t : 0.f; }
            w[e] = t;
        }
        o[i] = w;
    }
}

// Planes too large for registers (HREM's 1280x720 through the feature network: 360x640 = 230 400 values per plane, 128 planes): several
// blocks per plane.  Pass 1 leaves each chunk's sum and sum of squares (accumulated in double) in a scratch of the caller, pass 2 adds a
// plane's chunks, derives mean and variance in double and applies them.  One block per plane looping three times over 900 KB was 922 us
// per call - a fifth of the 1280x720 forward - on half of the CUs; these two passes run at the memory system's pace.
constexpr int kNormChunk4 = 8192;                                 // float4 per block: 8 per thread
__global__ __launch_bounds__(1024) void instnorm_stats_kernel(const float* __restrict__ x, int hw, int chunks, double* __restrict__ stats) {
    __shared__ double sh[32];
    const f32x4* p = reinterpret_cast<const f32x4*>(x + (size_t)blockIdx.y * hw);
    const int n4 = hw >> 2, i0 = blockIdx.x * kNormChunk4;
    float sf = 0.f, qf = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = i0 + k * 1024 + threadIdx.x;
        if (i < n4) {
            const f32x4 v = p[i];
            sf += (v[0] + v[1]) + (v[2] + v[3]);
            qf += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
    }
    double s = sf, q = qf;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { s += __shfl_xor(s, d); q += __shfl_xor(q, d); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sh[wave * 2] = s; sh[wave * 2 + 1] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double S = 0.0, Q = 0.0;
        for (int k = 0; k < 16; ++k) { S += sh[k * 2]; Q += sh[k * 2 + 1]; }
        stats[((size_t)blockIdx.y * chunks + blockIdx.x) * 2] = S;
        stats[((size_t)blockIdx.y * chunks + blockIdx.x) * 2 + 1] = Q;
    }
}

__global__ __launch_bounds__(1024) void instnorm_apply_kernel(const float* __restrict__ x, float* __restrict__ out, const float* __restrict__ res,
                                                              int hw, int chunks, const double* __restrict__ stats, int relu_inner) {
    double S = 0.0, Q = 0.0;
    for (int k = 0; k < chunks; ++k) { S += stats[((size_t)blockIdx.y * chunks + k) * 2]; Q += stats[((size_t)blockIdx.y * chunks + k) * 2 + 1]; }
    const double m = S / (double)hw;
    double var = Q / (double)hw - m * m;
    var = var < 0.0 ? 0.0 : var;
    const float mean = (float)m, rstd = 1.f / sqrtf((float)var + 1e-5f);
    const size_t base = (size_t)blockIdx.y * hw;
    const f32x4* p = reinterpret_cast<const f32x4*>(x + base);
    f32x4* o = reinterpret_cast<f32x4*>(out + base);
    const f32x4* r = res ? reinterpret_cast<const f32x4*>(res + base) : nullptr;
    const int n4 = hw >> 2, i0 = blockIdx.x * kNormChunk4;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = i0 + k * 1024 + threadIdx.x;
        if (i >= n4) continue;
        const f32x4 v = p[i];
        const f32x4 rv = r ? r[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = (v[e] - mean) * rstd;
            if (relu_inner) t = t > 0.f ? t : 0.f;
            if (r) { t += rv[e]; t = t > 0.f ? t : 0.f; }
            w[e] = t;
        }
        o[i] = w;
    }
}

// D[p1][p2] tiles of 64x64 per wave on v_mfma_f32_32x32x2_f32; both operands are read with unit stride along
// the pixel index (A[i=p1][k=c] = f1[c][p1], B[k=c][j=p2] = f2[c][p2]) and each accumulator register is a
// 128-B run of p2 for one p1 row.
// any hw (4-byte operand loads); the 8-byte form below needs an even pixel count
__global__ __launch_bounds__(256) void allpairs_odd_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                       float* __restrict__ out, int c, int hw, float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int b = blockIdx.z;
    const int m0 = blockIdx.y * 128 + (wave >> 1) * 64;      // p1 tile origin
    const int n0 = blockIdx.x * 128 + (wave & 1) * 64;       // p2 tile origin
    if (m0 >= hw || n0 >= hw) return;
    const float* a = f1 + (size_t)b * c * hw;
    const float* bb = f2 + (size_t)b * c * hw;
    int ma[2], nb[2];
    bool mv[2], nv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        ma[t] = m0 + t * 32 + j; mv[t] = ma[t] < hw; ma[t] = mv[t] ? ma[t] : 0;
        nb[t] = n0 + t * 32 + j; nv[t] = nb[t] < hw; nb[t] = nv[t] ? nb[t] : 0;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][t][r] = 0.f;
#pragma unroll 4
    for (int k = 0; k < c; k += 2) {
        const int kk = k + h;
        const bool kv = kk < c;
        const size_t ko = (size_t)(kv ? kk : 0) * hw;
        float av[2], bv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float x = a[ko + ma[t]], y = bb[ko + nb[t]];
            av[t] = (kv && mv[t]) ? x : 0.f;
            bv[t] = (kv && nv[t]) ? y : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[t], acc[s][t], 0, 0, 0);
    }
    float* o = out + (size_t)b * hw * hw;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int p2 = n0 + t * 32 + j;
            if (p2 >= hw) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p1 = m0 + s * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (p1 < hw) o[(size_t)p1 * hw + p2] = acc[s][t][r] * scale;
            }
        }
}

__global__ __launch_bounds__(256) void allpairs_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                       float* __restrict__ out, int c, int hw, float scale) {
    // MFMA row / column j of tile t stands for pixel 2j + t of the wave's 64-pixel span: one 8-byte load per operand and k-step
    // feeds both tiles, and the two column tiles of a lane are neighbouring p2 -> 8-byte stores in 256-byte runs.  (hw is even.)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int b = blockIdx.z;
    const int m0 = blockIdx.y * 128 + (wave >> 1) * 64;      // p1 span origin
    const int n0 = blockIdx.x * 128 + (wave & 1) * 64;       // p2 span origin
    if (m0 >= hw || n0 >= hw) return;
    const float* a = f1 + (size_t)b * c * hw;
    const float* bb = f2 + (size_t)b * c * hw;
    const int ma = m0 + 2 * j, nb = n0 + 2 * j;
    const bool mv = ma < hw, nv = nb < hw;
    const int mac = mv ? ma : 0, nbc = nv ? nb : 0;
    f32x16 acc[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][t][r] = 0.f;
#pragma unroll 4
    for (int k = 0; k < c; k += 2) {
        const int kk = k + h;
        const bool kv = kk < c;
        const size_t ko = (size_t)(kv ? kk : 0) * hw;
        f32x2 av = *reinterpret_cast<const f32x2*>(a + ko + mac);
        f32x2 bv = *reinterpret_cast<const f32x2*>(bb + ko + nbc);
        if (!(kv && mv)) av = f32x2{0.f, 0.f};
        if (!(kv && nv)) bv = f32x2{0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[t], acc[s][t], 0, 0, 0);
    }
    float* o = out + (size_t)b * hw * hw;
    if (!nv) return;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p1 = m0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * h) + s;
            if (p1 < hw) *reinterpret_cast<f32x2*>(o + (size_t)p1 * hw + nb) = f32x2{acc[s][0][r] * scale, acc[s][1][r] * scale};
        }
}

// The same product with both operands staged through LDS (round 6).  allpairs_kernel above feeds every MFMA from L2: a wave loads 1 KB
// per k-step for four MFMAs - 16 FLOP per byte, 4.6 TB/s of L2 reads at the 73 TFLOP/s it reaches (161 us at 60x80, 645 at batch 4).
// Here a block's 128 x 128 tile walks the channels in chunks of 16: the chunk's [16][128] slabs of both feature maps (8 KB each, rows of
// 512 contiguous bytes) arrive by 16-byte LDS-DMA, double-buffered, one barrier per chunk; a wave's two operands of a k-step are one
// 8-byte LDS read each (same pixel mapping: row j of tile t = pixel 2 j + t), so each staged byte feeds both waves that need it.
// 32 KB of LDS and 64 accumulator registers per block: four blocks per CU cover each other's waits.  Needs c % 16 == 0, hw % 4 == 0 and
// 16-byte aligned maps (the launch checks); pixels past hw read a zero page.
#define AP_LDS(p) ((__attribute__((address_space(3))) void*)(p))
#define AP_GLB(p) ((const __attribute__((address_space(1))) void*)(p))
__global__ __launch_bounds__(256, 4) void allpairs_lds_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                              float* __restrict__ out, int c, int hw, float scale,
                                                              const float* __restrict__ zero_page) {
    constexpr int KC = 16, TP = 128;                          // channels per chunk, pixels per block side
    constexpr int SLAB = KC * TP;                             // floats of one operand's chunk
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * SLAB];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int b = blockIdx.z;
    const int mb = blockIdx.y * TP, nb0 = blockIdx.x * TP;
    const float* a = f1 + (size_t)b * c * hw;
    const float* bb = f2 + (size_t)b * c * hw;
    // DMA plan: a slab is 16 rows x 32 pieces of 16 bytes = 512 slots = 2 instructions of the block per operand; slot = (row, piece)
    const char* ga[2];
    const char* gb[2];
    unsigned step_a[2], step_b[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int slot = (wave + 4 * k) * 64 + lane;          // 0 .. 511
        const int row = slot >> 5, pc = slot & 31;
        const int pa = mb + 4 * pc, pb = nb0 + 4 * pc;
        const bool oka = pa < hw, okb = pb < hw;              // (hw % 4 == 0: a piece is inside or outside as a whole)
        ga[k] = oka ? reinterpret_cast<const char*>(a + (size_t)row * hw + pa) : reinterpret_cast<const char*>(zero_page);
        gb[k] = okb ? reinterpret_cast<const char*>(bb + (size_t)row * hw + pb) : reinterpret_cast<const char*>(zero_page);
        step_a[k] = oka ? (unsigned)(KC * hw) * 4u : 0u;      // bytes to the same slot of the next chunk
        step_b[k] = okb ? (unsigned)(KC * hw) * 4u : 0u;
    }
    auto issue = [&](int stage) __attribute__((always_inline)) {
        float* sa = lds + stage * 2 * SLAB;
        float* sb = sa + SLAB;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            __builtin_amdgcn_global_load_lds(AP_GLB(ga[k]), AP_LDS(sa + (wave + 4 * k) * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(AP_GLB(gb[k]), AP_LDS(sb + (wave + 4 * k) * 256), 16, 0, 0);
            ga[k] += step_a[k];
            gb[k] += step_b[k];
        }
    };
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;    // this wave's 64 x 64 corner of the block tile
    f32x16 acc[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][t][r] = 0.f;
    const int nchunks = c / KC;
    issue(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of chunk ch have landed
        __builtin_amdgcn_s_barrier();                        // everyone's have; everyone is through with the other stage
        asm volatile("" ::: "memory");
        if (ch + 1 < nchunks) issue((ch + 1) & 1);
        const float* sa = lds + (ch & 1) * 2 * SLAB + wm + 2 * j;
        const float* sb = lds + (ch & 1) * 2 * SLAB + SLAB + wn + 2 * j;
#pragma unroll
        for (int k = 0; k < KC; k += 2) {
            const f32x2 av = *reinterpret_cast<const f32x2*>(sa + (k + h) * TP);
            const f32x2 bv = *reinterpret_cast<const f32x2*>(sb + (k + h) * TP);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[t], acc[s][t], 0, 0, 0);
        }
    }
    float* o = out + (size_t)b * hw * hw;
    const int m0 = mb + wm, nb = nb0 + wn + 2 * j;
    if (nb >= hw) return;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p1 = m0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * h) + s;
            if (p1 < hw) *reinterpret_cast<f32x2*>(o + (size_t)p1 * hw + nb) = f32x2{acc[s][0][r] * scale, acc[s][1][r] * scale};
        }
}

__global__ __launch_bounds__(256) void pool2_kernel(const float* __restrict__ in, float* __restrict__ out, long planes, int h,
                                                    int w) {
    const int oh = h / 2, ow = w / 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * oh * ow) return;
    const int x = idx % ow, y = (idx / ow) % oh;
    const long p = idx / ((long)ow * oh);
    const float* s = in + (p * h + 2 * y) * w + 2 * x;
    out[idx] = (s[0] + s[1] + s[w] + s[w + 1]) * 0.25f;
}

// Three avg_pool2d(2, 2) levels in one launch (the correlation pyramid, model/corr.py:24-27; EEMFlow+'s feature levels 4 .. 6,
// EEMFlow+.py:170-175).  Block = one plane's band of 8 input rows: 4 rows of level 1 (kept in LDS), 2 of level 2, 1 of level 3 - bands are
// independent because every level halves with floor.  pool2_kernel's own expression per output, so the three levels are bit for bit
// those of three launches; the input is read once and the two smaller levels never come back from memory.
__global__ __launch_bounds__(128) void pool2x3_kernel(const float* __restrict__ in, float* __restrict__ o1, float* __restrict__ o2,
                                                      float* __restrict__ o3, int h, int w, int nb) {
    extern __shared__ float pl[];                                // level 1 [4][w1], level 2 [2][w2]
    const int h1 = h / 2, w1 = w / 2, h2 = h1 / 2, w2 = w1 / 2, h3 = h2 / 2, w3 = w2 / 2;
    const long p = blockIdx.x / nb;
    const int b = blockIdx.x - p * nb, tid = threadIdx.x;
    float* l1 = pl;
    float* l2 = pl + 4 * w1;
    const int r1 = min(4, h1 - 4 * b), r2 = min(2, h2 - 2 * b);
    const float* s = in + (p * h + 8 * b) * w;
    for (int i = tid; i < r1 * w1; i += 128) {
        const int y = i / w1, x = i - y * w1;
        const float* q = s + 2 * y * w + 2 * x;
        const float v = (q[0] + q[1] + q[w] + q[w + 1]) * 0.25f;
        l1[y * w1 + x] = v;
        o1[(p * h1 + 4 * b + y) * w1 + x] = v;
    }
    __syncthreads();
    for (int i = tid; i < r2 * w2; i += 128) {
        const int y = i / w2, x = i - y * w2;
        const float* q = l1 + 2 * y * w1 + 2 * x;
        const float v = (q[0] + q[1] + q[w1] + q[w1 + 1]) * 0.25f;
        l2[y * w2 + x] = v;
        o2[(p * h2 + 2 * b + y) * w2 + x] = v;
    }
    if (b >= h3) return;
    __syncthreads();
    for (int x = tid; x < w3; x += 128) {
        const float* q = l2 + 2 * x;
        o3[(p * h3 + b) * w3 + x] = (q[0] + q[1] + q[w2] + q[w2 + 1]) * 0.25f;
    }
}

// grid_sample(align_corners=True, zeros) at pixel coordinates, computed like the reference: normalise with
// (size-1), un-normalise again, floor, 4 taps (model/model_utils.py:7-21)
// FLAT: the four corners as unconditional loads from clamped cells, the bounds applied to the values - one round trip per sample
// instead of four dependent ones (a load in one arm of a lane-dependent conditional is a branch + s_waitcnt vmcnt(0)); it also loads
// the corners that fall outside the map, which the coarse levels have many of, so launches that fill the chip (batch 4) keep the
// conditional form (E-RAFT 640x480: batch 1 149.0 -> 150.8 frames/s with FLAT, batch 4 198 -> 194)
template <bool FLAT>
__device__ __forceinline__ float sample_bilinear(const float* __restrict__ img, int h, int w, float x, float y) {
    const float xn = 2.f * x / (float)(w - 1) - 1.f, yn = 2.f * y / (float)(h - 1) - 1.f;
    const float ix = ((xn + 1.f) * 0.5f) * (float)(w - 1), iy = ((yn + 1.f) * 0.5f) * (float)(h - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float tx = ix - fx, ty = iy - fy;
    float v00, v01, v10, v11;
    if (FLAT) {
        const int ya = min(max(y0, 0), h - 1), yb = min(max(y0 + 1, 0), h - 1), xa = min(max(x0, 0), w - 1), xb = min(max(x0 + 1, 0), w - 1);
        const float r00 = img[(size_t)ya * w + xa], r01 = img[(size_t)ya * w + xb], r10 = img[(size_t)yb * w + xa], r11 = img[(size_t)yb * w + xb];
        const bool iy0 = y0 >= 0 && y0 < h, iy1 = y0 + 1 >= 0 && y0 + 1 < h, ix0 = x0 >= 0 && x0 < w, ix1 = x0 + 1 >= 0 && x0 + 1 < w;
        v00 = (iy0 && ix0) ? r00 : 0.f; v01 = (iy0 && ix1) ? r01 : 0.f; v10 = (iy1 && ix0) ? r10 : 0.f; v11 = (iy1 && ix1) ? r11 : 0.f;
    } else {
        auto at = [&](int yy, int xx) -> float {
            return (yy >= 0 && yy < h && xx >= 0 && xx < w) ? img[(size_t)yy * w + xx] : 0.f;
        };
        v00 = at(y0, x0); v01 = at(y0, x0 + 1); v10 = at(y0 + 1, x0); v11 = at(y0 + 1, x0 + 1);
    }
    return v00 * (1.f - tx) * (1.f - ty) + v01 * tx * (1.f - ty) + v10 * (1.f - tx) * ty + v11 * tx * ty;
}

__global__ __launch_bounds__(256) void lookup_kernel(LookupArgs a) {
    const int hw = a.h * a.w;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)a.batch * 324 * hw) return;
    const int p = idx % hw;
    const int ch = (idx / hw) % 324;
    const int b = idx / ((long)hw * 324);
    const int lvl = ch / 81, k = ch - lvl * 81;
    const int i = k / 9, jj = k - i * 9;
    const float cx = a.coords[((size_t)b * 2 + 0) * hw + p], cy = a.coords[((size_t)b * 2 + 1) * hw + p];
    const float sc = (float)(1 << lvl);
    const float x = cx / sc + (float)(i - 4);          // the reference adds (dy[i], dx[j]) to (x, y)
    const float y = cy / sc + (float)(jj - 4);
    const float* img = a.pyr[lvl] + ((size_t)b * hw + p) * a.ph[lvl] * a.pw[lvl];
    a.out[((size_t)b * a.out_ctotal + ch) * hw + p] = sample_bilinear<false>(img, a.ph[lvl], a.pw[lvl], x, y);
    if (a.flow_dst != nullptr && ch < 2)
        a.flow_dst[((size_t)b * a.flow_ctotal + a.flow_coff + ch) * hw + p] = (ch ? cy : cx) - a.coords0[((size_t)b * 2 + ch) * hw + p];
}

// The same lookup with both sides coalesced.  lookup_kernel above gives consecutive lanes consecutive PIXELS: its stores are
// 256-byte runs but every lane samples a different pixel's correlation map (19 KB apart at level 0).  Here a block owns 64
// consecutive pixels of one pyramid level; lanes walk the flattened (pixel, tap) pairs, so the 64 samples of an instruction
// come from the 10x10 windows of one or two maps; the values pass through an LDS tile [tap][pixel] and leave as 256-byte runs.
// PX pixels per block: 64, or 16 for the launches that leave the chip mostly empty (60x80 at batch 1: 300 blocks of 64 pixels = 4 waves on
// a CU, each a chain of ~20 dependent sample round trips to a volume that lives in the Infinity Cache; 1 200 blocks of 16 pixels = 5 round
// trips each with ~19 waves per CU to hide them; the stores become 64-byte runs of a 6 MB output: 14.7 -> 11.1 us; 8 pixels: 11.5 - what
// is left is the volume's ~12 MB of scattered 128-byte lines).  Same samples, same values.
template <bool FLAT, int PX>
__global__ __launch_bounds__(256) void lookup_tiled_kernel(LookupArgs a) {
    __shared__ float tile[81][PX + 1];
    __shared__ float cxs[PX], cys[PX];
    const int hw = a.h * a.w;
    const int lvl = blockIdx.y, b = blockIdx.z;
    const int p0 = blockIdx.x * PX;
    const int npx = min(PX, hw - p0);
    const int tid = threadIdx.x;
    if (tid < PX) {
        const int p = p0 + min(tid, npx - 1);
        cxs[tid] = a.coords[((size_t)b * 2 + 0) * hw + p];
        cys[tid] = a.coords[((size_t)b * 2 + 1) * hw + p];
    }
    __syncthreads();
    if (a.flow_dst != nullptr && lvl == 0 && tid < 2 * npx) {           // flow = coords1 - coords0 of this block's pixels
        const int ch = tid >= npx ? 1 : 0, px = tid - ch * npx;
        const float c1v = ch ? cys[px] : cxs[px];
        a.flow_dst[((size_t)b * a.flow_ctotal + a.flow_coff + ch) * hw + p0 + px] = c1v - a.coords0[((size_t)b * 2 + ch) * hw + p0 + px];
    }
    const float inv = 1.f / (float)(1 << lvl);
    const int ph = a.ph[lvl], pw = a.pw[lvl];
    const float* maps = a.pyr[lvl] + ((size_t)b * hw + p0) * ph * pw;
    for (int item = tid; item < npx * 81; item += 256) {
        const int px = item / 81, k = item - px * 81;
        const int i = k / 9, jj = k - i * 9;
        const float x = cxs[px] * inv + (float)(i - 4);          // as lookup_kernel: the reference adds (dy[i], dx[j]) to (x, y)
        const float y = cys[px] * inv + (float)(jj - 4);
        tile[k][px] = sample_bilinear<FLAT>(maps + (size_t)px * ph * pw, ph, pw, x, y);
    }
    __syncthreads();
    float* out = a.out + ((size_t)b * a.out_ctotal + lvl * 81) * hw + p0;
    for (int e = tid; e < 81 * PX; e += 256) {
        const int k = e / PX, px = e % PX;
        if (px < npx) out[(size_t)k * hw + px] = tile[k][px];
    }
}

// On-the-fly form of the same lookup (the `alt_cuda_corr` pattern, model/flowformer/corr.py:60-91): no all-pairs
// volume.  avg_pool2d of the volume over its last two dimensions is the correlation with avg_pool2d of fmap2 (linear), so a pixel's
// 81 taps at level l are bilinear samples of  corr_l(p, q) = <fmap1[:, p], pool^l(fmap2)[:, q]> / sqrt(C)  over the 10 x 10 integer
// cells its window touches.  One block per (pixel, level): the fmap1 column in LDS, a thread per window cell (a C-long dot product,
// cells of a window row are consecutive floats of a feature plane), then a thread per tap with lookup_kernel's own corner / weight
// arithmetic on the 10 x 10 cells (cells outside the map hold 0; a corner that rounding moves out of the window beside an integer
// coordinate carries a weight ~1e-7 and is dropped).  C * 100 reads per pixel, level and iteration instead of 81 x 4: this path trades
// ~250x the lookup's memory traffic for not holding B * (HW)^2 * 4/3 floats - 829 MB per sample at 1280x720 - and is off by default.
__global__ __launch_bounds__(128) void altcorr_kernel(AltCorrArgs a) {
    extern __shared__ float f1s[];                               // [C] then win[100]
    float* win = f1s + a.c;
    const int hw = a.h * a.w;
    const int p = blockIdx.x, lvl = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x;
    for (int ch = tid; ch < a.c; ch += 128) f1s[ch] = a.f1[((size_t)b * a.c + ch) * hw + p];
    const float cx = a.coords[((size_t)b * 2 + 0) * hw + p], cy = a.coords[((size_t)b * 2 + 1) * hw + p];
    const float sc = (float)(1 << lvl);
    const int h = a.ph[lvl], w = a.pw[lvl];
    auto tap = [](float c, float scv, int off, int size, int& i0, float& t) {
        const float v = c / scv + (float)(off - 4);
        const float vn = 2.f * v / (float)(size - 1) - 1.f;
        const float iv = ((vn + 1.f) * 0.5f) * (float)(size - 1);
        const float f = floorf(iv);
        i0 = (int)fminf(fmaxf(f, -16.f), (float)size + 16.f);    // for the conversion only: a clamped window has no cell in the map
        t = iv - f;
    };
    int xb, yb; float t0;
    tap(cx, sc, 0, w, xb, t0);
    tap(cy, sc, 0, h, yb, t0);
    __syncthreads();
    if (tid < 100) {
        const int Y = yb + tid / 10, X = xb + tid % 10;
        float acc = 0.f;
        if (Y >= 0 && Y < h && X >= 0 && X < w) {
            const float* q = a.f2[lvl] + (size_t)b * a.c * h * w + (size_t)Y * w + X;
            const size_t plane = (size_t)h * w;
#pragma unroll 8
            for (int ch = 0; ch < a.c; ++ch) acc += f1s[ch] * q[ch * plane];
        }
        win[tid] = acc * a.scale;
    }
    __syncthreads();
    if (tid < 81) {
        const int i = tid / 9, jj = tid - i * 9;                 // channel i * 9 + jj samples x + (i - 4), y + (jj - 4)
        int x0, y0; float tx, ty;
        tap(cx, sc, i, w, x0, tx);
        tap(cy, sc, jj, h, y0, ty);
        auto at = [&](int yy, int xx) -> float {
            const int r = yy - yb, cix = xx - xb;
            return (r >= 0 && r < 10 && cix >= 0 && cix < 10) ? win[r * 10 + cix] : 0.f;
        };
        const float v = at(y0, x0) * (1.f - tx) * (1.f - ty) + at(y0, x0 + 1) * tx * (1.f - ty) + at(y0 + 1, x0) * (1.f - tx) * ty +
                        at(y0 + 1, x0 + 1) * tx * ty;
        a.out[((size_t)b * a.out_ctotal + lvl * 81 + tid) * hw + p] = v;
    }
}

__global__ __launch_bounds__(256) void coords_init_kernel(float* c0, float* c1, const float* init, int batch, int h, int w) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int hw = h * w;
    if (idx >= (long)batch * 2 * hw) return;
    const int p = idx % hw, ch = (idx / hw) % 2;
    const float v = ch == 0 ? (float)(p % w) : (float)(p / w);
    c0[idx] = v;
    c1[idx] = init ? v + init[idx] : v;
}

__global__ __launch_bounds__(256) void flow_kernel(const float* c0, const float* c1, float* dst, int dst_ctotal, int dst_coff,
                                                   int batch, int hw) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)batch * 2 * hw) return;
    const int p = idx % hw, ch = (idx / hw) % 2, b = idx / (2L * hw);
    dst[((size_t)b * dst_ctotal + dst_coff + ch) * hw + p] = c1[idx] - c0[idx];
}

__global__ __launch_bounds__(256) void axpy_kernel(float* y, const float* x, long n) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < n) y[idx] += x[idx];
}

__global__ __launch_bounds__(256) void sum_kernel(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ b, long n) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < n) out[idx] = a[idx] + b[idx];
}

// out[n][c][p] = a[n][a_coff + c][p] * b[n][c][p]: r * h of the GRU when z and r come out of one 256-cout launch
__global__ __launch_bounds__(256) void mul_channels_kernel(float* __restrict__ out, const float* __restrict__ a, int a_ctotal, int a_coff,
                                                            const float* __restrict__ b, int c, long hw, long n) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const long chw = (long)c * hw;
    const long img = idx / chw, rem = idx - img * chw;
    out[idx] = a[(img * a_ctotal + a_coff) * hw + rem] * b[idx];
}

// one thread per output pixel (both flow channels): softmax over the 9 mask logits of its (sub-pixel, cell),
// convex combination of the 3x3 neighbourhood of 8*flow (zero outside, F.unfold padding=1)
__global__ __launch_bounds__(256) void convex_up_kernel(const float* __restrict__ c0, const float* __restrict__ c1,
                                                        const float* __restrict__ mask, float* __restrict__ out, int batch,
                                                        int h, int w, int top, int left, int oh, int ow,
                                                        const float* __restrict__ delta, float* __restrict__ c1n) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)batch * oh * ow) {                                  // the launch's tail: coords1_next = coords1 + delta, every cell
        const long e = idx - (long)batch * oh * ow;                       // (cells of the padding have no output pixel of their own)
        if (c1n != nullptr && e < (long)batch * 2 * h * w) c1n[e] = c1[e] + delta[e];
        return;
    }
    const int X = idx % ow + left, Y = (idx / ow) % oh + top;
    const int b = idx / ((long)ow * oh);
    const int x = X >> 3, y = Y >> 3, sx = X & 7, sy = Y & 7;
    const int hw = h * w;
    const float* m = mask + ((size_t)b * 576 + sy * 8 + sx) * hw + y * w + x;
    float lg[9], mx = -3.4e38f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { lg[k] = m[(size_t)k * 64 * hw]; mx = fmaxf(mx, lg[k]); }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { lg[k] = expf(lg[k] - mx); den += lg[k]; }
    // the nine neighbours' coordinates: loads unconditional from clamped cells, the bounds test applied to the weight (a load inside
    // a lane-dependent branch is followed by the compiler's s_waitcnt vmcnt(0): up to 54 dependent round trips per pixel before)
    float fx[9], fy[9];
    bool in[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        in[k] = yy >= 0 && yy < h && xx >= 0 && xx < w;
        const size_t q = (size_t)b * 2 * hw + min(max(yy, 0), h - 1) * w + min(max(xx, 0), w - 1);
        // (c1 + delta) - c0: the sum first, as coords1 = coords1 + delta_flow then coords1 - coords0 (model/eraft.py:149,152)
        if (delta) {                                                     // (uniform)
            fx[k] = (c1[q] + delta[q]) - c0[q];
            fy[k] = (c1[q + hw] + delta[q + hw]) - c0[q + hw];
        } else {
            fx[k] = c1[q] - c0[q];
            fy[k] = c1[q + hw] - c0[q + hw];
        }
    }
    float u = 0.f, v = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        if (in[k]) {
            const float wgt = lg[k] / den;
            u += wgt * (8.f * fx[k]);
            v += wgt * (8.f * fy[k]);
        }
    }
    const size_t o = (size_t)b * 2 * oh * ow + (size_t)(Y - top) * ow + (X - left);
    out[o] = u;
    out[o + (size_t)oh * ow] = v;
}

inline unsigned blocks(long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

// pad4_kernel's grid: 32 threads along a row of ow4 16-byte pieces, 32 rows per block, the row blocks folded over grid.y and grid.z
static dim3 pad4_grid(long rows, int ow4) {
    const long rb = (rows + 31) / 32;
    const unsigned gy = (unsigned)(rb < 32768 ? rb : 32768), gz = (unsigned)((rb + gy - 1) / gy);
    return dim3((unsigned)((ow4 + 31) / 32), gy ? gy : 1, gz ? gz : 1);
}

int er_pad2_launch(const float* in0, const float* in1, float* out, int nc, int h, int w, int left, int right, int top, int bottom,
                   hipStream_t st) {
    const int oh = h + top + bottom, ow = w + left + right;
    if ((ow & 3) == 0 && ((uintptr_t)out & 15) == 0) {
        hipLaunchKernelGGL(pad4_kernel, pad4_grid(2L * nc * oh, ow / 4), dim3(256), 0, st, in0, out, nc, h, w, left, top, oh, ow, in1);
        EEM_HIP_CHECK(hipGetLastError());
        return EEM_OK;
    }
    const int rc = er_pad_launch(in0, out, nc, h, w, left, right, top, bottom, st);
    return rc != EEM_OK ? rc : er_pad_launch(in1, out + (size_t)nc * oh * ow, nc, h, w, left, right, top, bottom, st);
}

int er_pad_launch(const float* in, float* out, int nc, int h, int w, int left, int right, int top, int bottom, hipStream_t st) {
    const int oh = h + top + bottom, ow = w + left + right;
    if ((ow & 3) == 0 && ((uintptr_t)out & 15) == 0)
        hipLaunchKernelGGL(pad4_kernel, pad4_grid((long)nc * oh, ow / 4), dim3(256), 0, st, in, out, nc, h, w, left, top, oh, ow, (const float*)nullptr);
    else
        hipLaunchKernelGGL(pad_kernel, dim3(blocks((long)nc * oh * ow)), dim3(256), 0, st, in, out, nc, h, w, left, top, oh, ow);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

size_t er_instnorm_scratch_doubles(int planes, int hw) {
    return hw > 4 * 1024 * 20 ? (size_t)planes * ceil_div(hw >> 2, kNormChunk4) * 2 : 0;
}

int er_instnorm_launch(const float* x, float* out, const float* res, int planes, int hw, int relu_inner, hipStream_t st, double* stats,
                       size_t stats_cap) {
    const bool al = (hw & 3) == 0 && (((uintptr_t)x | (uintptr_t)out | (uintptr_t)res) & 15) == 0;
    if (al && hw > 4 * 1024 * 20 && stats && stats_cap >= er_instnorm_scratch_doubles(planes, hw)) {
        const int chunks = ceil_div(hw >> 2, kNormChunk4);
        hipLaunchKernelGGL(instnorm_stats_kernel, dim3(chunks, planes), dim3(1024), 0, st, x, hw, chunks, stats);
        hipLaunchKernelGGL(instnorm_apply_kernel, dim3(chunks, planes), dim3(1024), 0, st, x, out, res, hw, chunks, stats, relu_inner);
        EEM_HIP_CHECK(hipGetLastError());
        return EEM_OK;
    }
    if (al && hw <= 4 * 1024 * 5) hipLaunchKernelGGL(instnorm_reg_kernel<5>, dim3(planes), dim3(1024), 0, st, x, out, res, hw, relu_inner);
    else if (al && hw <= 4 * 1024 * 20) hipLaunchKernelGGL(instnorm_reg_kernel<20>, dim3(planes), dim3(1024), 0, st, x, out, res, hw, relu_inner);
    else hipLaunchKernelGGL(instnorm_kernel, dim3(planes), dim3(256), 0, st, x, out, res, hw, relu_inner);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int er_allpairs_launch(const float* f1, const float* f2, float* out, int batch, int c, int hw, hipStream_t st) {
    const bool even = hw % 2 == 0 && ((uintptr_t)f1 & 7) == 0 && ((uintptr_t)f2 & 7) == 0 && ((uintptr_t)out & 7) == 0;
    dim3 grid(ceil_div(hw, 128), ceil_div(hw, 128), batch);
    // the LDS-staged form where it applies (EEM_ALLPAIRS_L2=1, read per call: the form that feeds the MFMAs from L2)
    const char* el2 = getenv("EEM_ALLPAIRS_L2");
    static float* zero_pages[16] = {};                       // 256 bytes of zeros per device for the pieces past the last pixel (never freed)
    int dev = 0;
    if (even && c % 16 == 0 && hw % 4 == 0 && (((uintptr_t)f1 | (uintptr_t)f2) & 15) == 0 && !(el2 && el2[0] == '1') &&
        hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16) {
        if (!zero_pages[dev]) {
            EEM_HIP_CHECK(hipMalloc(&zero_pages[dev], 256));
            EEM_HIP_CHECK(hipMemset(zero_pages[dev], 0, 256));
        }
        hipLaunchKernelGGL(allpairs_lds_kernel, grid, dim3(256), 0, st, f1, f2, out, c, hw, 1.0f / sqrtf((float)c), zero_pages[dev]);
        EEM_HIP_CHECK(hipGetLastError());
        return EEM_OK;
    }
    if (even) hipLaunchKernelGGL(allpairs_kernel, grid, dim3(256), 0, st, f1, f2, out, c, hw, 1.0f / sqrtf((float)c));
    else hipLaunchKernelGGL(allpairs_odd_kernel, grid, dim3(256), 0, st, f1, f2, out, c, hw, 1.0f / sqrtf((float)c));
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int er_pool2_launch(const float* in, float* out, long planes, int h, int w, hipStream_t st) {
    const long n = planes * (h / 2) * (w / 2);
    if (n == 0) return EEM_OK;
    hipLaunchKernelGGL(pool2_kernel, dim3(blocks(n)), dim3(256), 0, st, in, out, planes, h, w);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

// levels 1 .. 3 of a pyramid over `planes` planes of h x w in one launch (pool2x3_kernel); EEM_POOL_CHAIN=1 (read per call: the equality
// test flips it) or a level-3 map that would be empty: three er_pool2_launch
int er_pool2x3_launch(const float* in, float* o1, float* o2, float* o3, long planes, int h, int w, hipStream_t st) {
    const int h1 = h / 2, w1 = w / 2, h2 = h1 / 2, w2 = w1 / 2, h3 = h2 / 2, w3 = w2 / 2;
    const char* e = getenv("EEM_POOL_CHAIN");
    const long nblk = planes * ceil_div(h1, 4);
    if ((e && e[0] == '1') || nblk >= (1L << 31) || h3 < 1 || w3 < 1) {
        int rc;
        if ((rc = er_pool2_launch(in, o1, planes, h, w, st)) != EEM_OK || (rc = er_pool2_launch(o1, o2, planes, h1, w1, st)) != EEM_OK) return rc;
        return er_pool2_launch(o2, o3, planes, h2, w2, st);
    }
    if (planes == 0) return EEM_OK;
    hipLaunchKernelGGL(pool2x3_kernel, dim3((unsigned)nblk), dim3(128), (4 * w1 + 2 * w2) * sizeof(float), st, in, o1, o2, o3, h, w, ceil_div(h1, 4));
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

static const long flat_max = [] { const char* e = getenv("EEM_LOOKUP_FLAT_MAX"); return e ? atol(e) : 1024L; }();
// done_ev (optional): signalled when the launch completes - as the launch's own completion signal (hipExtLaunchKernelGGL's stop event),
// not as a separate hipEventRecord: an event record behind a kernel is a barrier packet that costs the recording stream ~6 us before its
// next kernel starts (tools/eraft_timeline.sh: the gap in front of convc1 in every iteration)
int er_lookup_launch(const LookupArgs& a, hipStream_t st, hipEvent_t done_ev) {
    static const bool plain = [] { const char* e = getenv("EEM_LOOKUP_PLAIN"); return e && e[0] == '1'; }();
    const long blocks64 = (long)ceil_div(a.h * a.w, 64) * 4 * a.batch;
    // EEM_LOOKUP_PX=64|32|16 (read per call: the equality test flips it): the block's pixels; default 16 below EEM_LOOKUP_FLAT_MAX blocks of 64
    const char* epx = getenv("EEM_LOOKUP_PX");
    const int px = epx ? atoi(epx) : (blocks64 < flat_max ? 16 : 64);
    const dim3 gt(ceil_div(a.h * a.w, px == 16 ? 16 : px == 32 ? 32 : 64), 4, a.batch);
    if (plain) hipExtLaunchKernelGGL(lookup_kernel, dim3(blocks((long)a.batch * 324 * a.h * a.w)), dim3(256), 0, st, nullptr, done_ev, 0, a);
    else if (blocks64 < flat_max) {
        if (px == 16) hipExtLaunchKernelGGL((lookup_tiled_kernel<true, 16>), gt, dim3(256), 0, st, nullptr, done_ev, 0, a);
        else if (px == 32) hipExtLaunchKernelGGL((lookup_tiled_kernel<true, 32>), gt, dim3(256), 0, st, nullptr, done_ev, 0, a);
        else hipExtLaunchKernelGGL((lookup_tiled_kernel<true, 64>), gt, dim3(256), 0, st, nullptr, done_ev, 0, a);
    } else {
        if (px == 16) hipExtLaunchKernelGGL((lookup_tiled_kernel<false, 16>), gt, dim3(256), 0, st, nullptr, done_ev, 0, a);
        else if (px == 32) hipExtLaunchKernelGGL((lookup_tiled_kernel<false, 32>), gt, dim3(256), 0, st, nullptr, done_ev, 0, a);
        else hipExtLaunchKernelGGL((lookup_tiled_kernel<false, 64>), gt, dim3(256), 0, st, nullptr, done_ev, 0, a);
    }
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int er_altcorr_launch(const AltCorrArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(altcorr_kernel, dim3(a.h * a.w, 4, a.batch), dim3(128), (a.c + 100) * sizeof(float), st, a);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int er_coords_init_launch(float* c0, float* c1, const float* init, int batch, int h, int w, hipStream_t st) {
    hipLaunchKernelGGL(coords_init_kernel, dim3(blocks((long)batch * 2 * h * w)), dim3(256), 0, st, c0, c1, init, batch, h, w);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int er_flow_launch(const float* c0, const float* c1, float* dst, int dst_ctotal, int dst_coff, int batch, int hw, hipStream_t st) {
    hipLaunchKernelGGL(flow_kernel, dim3(blocks((long)batch * 2 * hw)), dim3(256), 0, st, c0, c1, dst, dst_ctotal, dst_coff, batch, hw);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int er_axpy_launch(float* y, const float* x, long n, hipStream_t st) {
    hipLaunchKernelGGL(axpy_kernel, dim3(blocks(n)), dim3(256), 0, st, y, x, n);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int er_sum_launch(float* out, const float* a, const float* b, long n, hipStream_t st) {
    hipLaunchKernelGGL(sum_kernel, dim3(blocks(n)), dim3(256), 0, st, out, a, b, n);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int er_mul_channels_launch(float* out, const float* a, int a_ctotal, int a_coff, const float* b, int batch, int c, long hw, hipStream_t st) {
    const long n = (long)batch * c * hw;
    hipLaunchKernelGGL(mul_channels_kernel, dim3(blocks(n)), dim3(256), 0, st, out, a, a_ctotal, a_coff, b, c, hw, n);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int er_convex_up_launch(const float* c0, const float* c1, const float* mask, float* out, int batch, int h, int w, int top,
                        int left, int oh, int ow, hipStream_t st, const float* delta, float* c1n) {
    EEM_REQUIRE((delta == nullptr) == (c1n == nullptr) && (c1n == nullptr || c1n != c1), "er_convex_up_launch: delta and coords1_next come together");
    hipLaunchKernelGGL(convex_up_kernel, dim3(blocks((long)batch * oh * ow + (c1n ? (long)batch * 2 * h * w : 0L))), dim3(256), 0, st, c0, c1,
                       mask, out, batch, h, w, top, left, oh, ow, delta, c1n);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}
