// Kernels of the EEMFlow training step (reference: train_mvsec.py:178-183,201-227,241-258 and the autograd
// transposes of model/EEMFlow/EEMFlow.py): loss + its gradient, bilinear-upsample / pooling / local-correlation
// backward, weight and bias gradients, gradient clipping + AdamW, weight re-packing.
// Data gradients of the convolutions reuse gconv (transposed weights, transposed stride, LeakyReLU' gate).
#include "train.h"

namespace {

inline unsigned nblocks(long n) { return (unsigned)((n + 255) / 256); }

// sumsq / nskip (optimizer step only): the launch behind adamw_kernel also advances the count of skipped steps (every adamw thread has
// read the old one by then) and leaves the sum of squares cleared for the next step
__global__ __launch_bounds__(256) void repack_kernel(const float* __restrict__ flat, const int* __restrict__ idx,
                                                     float* __restrict__ arena, long n, double* __restrict__ sumsq,
                                                     int* __restrict__ nskip) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const int k = idx[i]; arena[i] = k ? flat[k - 1] : 0.f; }
    if (sumsq && i == 0) {
        if (!(*sumsq <= 1.7976931348623157e308)) *nskip += 1;
        *sumsq = 0.0;
    }
}

// ---- sequence_loss for one prediction (train_mvsec.py:201-227): loss = mean(valid * |flow - gt|) over B*2*H*W,
// valid = (valid >= 0.5) & (|gt| < 400); dflow = sign(flow - gt) * valid / (B*2*H*W); epe statistics of :218-226.
__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ flow, const float* __restrict__ gt,
                                                   const float* __restrict__ valid, float* __restrict__ dflow, int batch, int hw,
                                                   float weight, double* __restrict__ stats) {
    __shared__ double sh[4][6];
    double l = 0, e = 0, cnt = 0, c1 = 0, c3 = 0, c5 = 0;
    // grid-stride: the five f64 atomics at the end of a block land on one cache line and serialise (~14 ns each)
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < (long)batch * hw; idx += (long)gridDim.x * 256) {
        const int b = idx / hw, p = idx - (long)b * hw;
        const size_t o = (size_t)b * 2 * hw + p;
        const float gx = gt[o], gy = gt[o + hw], fx = flow[o], fy = flow[o + hw];
        const bool ok = (valid[idx] >= 0.5f) && (sqrtf(gx * gx + gy * gy) < 400.f);
        const float dx = fx - gx, dy = fy - gy;
        const float s = ok ? weight / ((float)batch * 2.f * (float)hw) : 0.f;
        dflow[o] = dx > 0.f ? s : (dx < 0.f ? -s : 0.f);
        dflow[o + hw] = dy > 0.f ? s : (dy < 0.f ? -s : 0.f);
        if (ok) {
            l += (double)fabsf(dx) + (double)fabsf(dy);
            const float ep = sqrtf(dx * dx + dy * dy);
            e += ep; cnt += 1; c1 += ep < 1.f; c3 += ep < 3.f; c5 += ep < 5.f;
        }
    }
    double v[6] = {l, e, cnt, c1, c3, c5};
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v[k] += __shfl_xor(v[k], d);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int k = 0; k < 6; ++k) sh[wave][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 6) atomicAdd(&stats[threadIdx.x], sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// ---- adjoint of F.interpolate(bilinear, align_corners=False), separable and deterministic:
// pass X: tmp[nc][Y][xc] = sum_X wx(X, xc) * d[nc][Y][X];  pass Y: out[nc][yc][xc] = sum_Y wy(Y, yc) * tmp[nc][Y][xc]
__device__ __forceinline__ float up_weight(float scale, int dst, int in_size, int c) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    const int i0 = (int)s;
    const int i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    const float l1 = s - (float)i0;
    return (i0 == c ? 1.f - l1 : 0.f) + (i1 == c ? l1 : 0.f);
}

__global__ __launch_bounds__(256) void upbwd_x_kernel(const float* __restrict__ d, float* __restrict__ tmp, long nc_oh, int ow, int w) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nc_oh * w) return;
    const int xc = idx % w;
    const long row = idx / w;
    const float scale = (float)w / (float)ow, inv = (float)ow / (float)w;
    int lo = (int)floorf(((float)xc - 1.f + 0.5f) * inv - 0.5f) - 1, hi = (int)ceilf(((float)xc + 1.f + 0.5f) * inv - 0.5f) + 1;
    lo = lo < 0 ? 0 : lo; hi = hi > ow - 1 ? ow - 1 : hi;
    if (xc == 0) lo = 0;                                      // negative source positions clamp to column 0
    const float* r = d + row * ow;
    float s = 0.f;
    for (int X = lo; X <= hi; ++X) s += up_weight(scale, X, w, xc) * r[X];
    tmp[idx] = s;
}

// The same sums with a WAVE per source row (round 6): the row enters LDS by coalesced loads, then every target column's footprint is
// summed by the 64 lanes together.  The thread-per-target form above reads each row with six far-apart walkers (46 us for 23 MB at
// 346x260 batch 32: 0.5 TB/s).
__global__ __launch_bounds__(256) void upbwd_x_rows_kernel(const float* __restrict__ d, float* __restrict__ tmp, long nc_oh, int ow, int w) {
    extern __shared__ float rowbuf[];                          // 4 waves x ow floats
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= nc_oh) return;
    float* rb = rowbuf + wave * ow;
    const float* r = d + row * ow;
    for (int X = lane; X < ow; X += 64) rb[X] = r[X];
    __builtin_amdgcn_wave_barrier();                           // (a wave's own LDS writes: ordered by the waitcnt the compiler inserts)
    const float scale = (float)w / (float)ow, inv = (float)ow / (float)w;
    for (int xc = 0; xc < w; ++xc) {
        int lo = (int)floorf(((float)xc - 1.f + 0.5f) * inv - 0.5f) - 1, hi = (int)ceilf(((float)xc + 1.f + 0.5f) * inv - 0.5f) + 1;
        lo = lo < 0 ? 0 : lo; hi = hi > ow - 1 ? ow - 1 : hi;
        if (xc == 0) lo = 0;
        float s = 0.f;
        for (int X = lo + lane; X <= hi; X += 64) s += up_weight(scale, X, w, xc) * rb[X];
        s = lane_group_sum<16>(s);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (lane == 0) tmp[row * w + xc] = s;
    }
}

__global__ __launch_bounds__(256) void upbwd_y_kernel(const float* __restrict__ tmp, float* __restrict__ out, int nc, int oh, int h, int w) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)nc * h * w) return;
    const int xc = idx % w, yc = (idx / w) % h;
    const int c = idx / ((long)w * h);
    const float scale = (float)h / (float)oh, inv = (float)oh / (float)h;
    int lo = (int)floorf(((float)yc - 1.f + 0.5f) * inv - 0.5f) - 1, hi = (int)ceilf(((float)yc + 1.f + 0.5f) * inv - 0.5f) + 1;
    lo = lo < 0 ? 0 : lo; hi = hi > oh - 1 ? oh - 1 : hi;
    if (yc == 0) lo = 0;
    float s = 0.f;
    for (int Y = lo; Y <= hi; ++Y) s += up_weight(scale, Y, h, yc) * tmp[((size_t)c * oh + Y) * w + xc];
    out[idx] = s;
}

// ---- avg_pool2d(k) backward: g[nc][y][x] (+)= dpool[nc][y/k][x/k] / k^2 inside the pooled region
__global__ __launch_bounds__(256) void poolbwd_kernel(const float* __restrict__ dpool, float* __restrict__ g, long nc, int h, int w,
                                                      int k, int gh, int gw, int accumulate, const float* __restrict__ gate) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nc * h * w) return;
    const int x = idx % w, y = (idx / w) % h;
    const long c = idx / ((long)w * h);
    const int py = y / k, px = x / k;
    const float v = (py < gh && px < gw) ? dpool[(c * gh + py) * gw + px] / (float)(k * k) : 0.f;
    float r = accumulate ? g[idx] + v : v;
    if (gate) r *= gate[idx] > 0.f ? 1.f : 0.1f;               // gradient w.r.t. the producing conv's pre-activation
    g[idx] = r;
}

// four pixels per thread (w % 4 == 0, k % 4 == 0: the four share one pooled cell), 32-bit index arithmetic
__global__ __launch_bounds__(256) void poolbwd4_kernel(const float* __restrict__ dpool, float* __restrict__ g, long nc, int h, int w,
                                                       int k, int gh, int gw, int accumulate, const float* __restrict__ gate) {
    const int w4 = w >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nc * h * w4) return;
    const int x4 = (int)(idx % w4);
    const long row = idx / w4;
    const int y = (int)(row % h);
    const long c = row / h;
    const int py = y / k, px = (x4 * 4) / k;
    const float v = (py < gh && px < gw) ? dpool[(c * gh + py) * gw + px] / (float)(k * k) : 0.f;
    f32x4* g4 = reinterpret_cast<f32x4*>(g);
    f32x4 r = accumulate ? g4[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] += v;
    if (gate) {
        const f32x4 gt = reinterpret_cast<const f32x4*>(gate)[idx];
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] *= gt[e] > 0.f ? 1.f : 0.1f;
    }
    g4[idx] = r;
}

// ---- local correlation backward (the transpose of corr_kernel in tail.hip):
// dx[b][c][p] += (1/C) sum_t dcv[b][t][p] * y[b][c][p + d_t];   dy[b][c][q] = (1/C) sum_t dcv[b][t][q - d_t] * x[b][c][q - d_t]
// Up to three correlations of one grid shape (the three stages of EEMFlow.py:160-173) as ONE launch (blockIdx.y = job) - round 6: three
// launches of 32 - 39 us each sat in the tail's chain of the training step.  Every load is unconditional from a clamped address with the
// condition applied to the value: a load inside a lane-dependent branch is followed by the compiler's s_waitcnt vmcnt(0) - 106
// dependent round trips per thread in the form this replaces.
struct CorrBwdJobs { CorrBwdJob job[3]; };
__global__ __launch_bounds__(256) void corrbwd_kernel(CorrBwdJobs jobs, int batch, int h, int w, const int* __restrict__ taps, int ntaps) {
    const CorrBwdJob& jb = jobs.job[blockIdx.y];
    const int c = jb.c;
    const int hw = h * w;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)batch * c * hw) return;
    const int p = idx % hw, ch = (idx / hw) % c, b = idx / ((long)hw * c);
    const int y = p / w, x = p - y * w;
    const float* __restrict__ dc = jb.dcv + (size_t)b * jb.dcv_ctotal * hw;
    const float* __restrict__ a1 = jb.f1 + ((size_t)b * c + ch) * hw;
    const float* __restrict__ a2 = jb.f2 + ((size_t)b * c + ch) * hw;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
    for (int t = 0; t < ntaps; ++t) {
        const int tp = taps[t];
        const int dy = tp / 9 - 4, dx = tp % 9 - 4;
        const int yy = y + dy, xx = x + dx;
        const bool in1 = yy >= 0 && yy < h && xx >= 0 && xx < w;
        const float v2 = a2[min(max(yy, 0), h - 1) * w + min(max(xx, 0), w - 1)];
        s1 = fmaf(dc[(size_t)t * hw + p], in1 ? v2 : 0.f, s1);
        const int y2 = y - dy, x2 = x - dx;
        const bool in2 = y2 >= 0 && y2 < h && x2 >= 0 && x2 < w;
        const int q2 = min(max(y2, 0), h - 1) * w + min(max(x2, 0), w - 1);
        const float d2v = dc[(size_t)t * hw + q2], v1 = a1[q2];
        s2 = fmaf(in2 ? d2v : 0.f, v1, s2);
    }
    jb.d1[idx] += s1 / (float)c;
    jb.d2[idx] = s2 / (float)c;
}

// ---- bias gradient: db[co] = sum_{n,p} g[n][co'][p] * LeakyReLU'(gate); grid (chunks, max cout, jobs)
struct BiasBatch { BiasJob job[WGRAD_MAX_JOBS]; };

__global__ __launch_bounds__(256) void biasgrad_kernel(BiasBatch batch) {
    __shared__ float sh[4];
    const BiasJob& b = batch.job[blockIdx.z];
    const int co = blockIdx.y;
    if (co >= b.cout) return;
    const float* __restrict__ g = b.g;
    const float* __restrict__ gate = b.gate;
    const int ch = b.g_coff + co * b.g_cmul;
    const int hw = b.hw;
    const long total = (long)b.n * hw;
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = i / hw, p = i - (long)n * hw;
        const size_t o = ((size_t)n * b.g_ctotal + ch) * hw + p;
        float v = g[o];
        if (gate) v *= gate[o] > 0.f ? 1.f : 0.1f;
        s += v;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&b.db[co], sh[0] + sh[1] + sh[2] + sh[3]);
}

// ---- weight gradient on v_mfma_f32_32x32x2_f32: dW[co][(ci,tap)] = sum_pixels G[co][pixel] * X[(ci,tap)][pixel].
// M = cout tiles, N = (32 input channels of this block's chunk) x taps, K = output pixels.  A block walks 4x32
// pixel tiles; per tile G (with the LeakyReLU' gate folded in) and the haloed X tile go to LDS (G rows with an
// odd pitch -> conflict-free column reads), each wave owns up to MAXT (cout-tile, tap-tile) accumulators and adds
// them to dW with fp32 atomics at the end (one atomic per weight and block).
constexpr int WG_TH = 4, WG_TW = 32, WG_PX = WG_TH * WG_TW, WG_GP = WG_PX + 1, WG_CI = 32, WG_MAXT = 9;

struct WgradBatch { WgradArgs job[WGRAD_MAX_JOBS]; };

template <int KH, int KW, int S>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradBatch batch) {
    const WgradArgs& a = batch.job[blockIdx.z];
    constexpr int KK = KH * KW;
    extern __shared__ __attribute__((aligned(16))) float lds[];     // MT*32*WG_GP + cin_here*XR*XC floats
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int ci0 = blockIdx.y * WG_CI;
    if (ci0 >= a.cin) return;                                        // a job with fewer channel chunks than the launch's grid
    const int cin_here = min(WG_CI, a.cin - ci0);
    const int MT = (a.cout + 31) >> 5;
    const int NT = (cin_here * KK + 31) >> 5;
    constexpr int XR = (WG_TH - 1) * S + KH, XC = (WG_TW - 1) * S + KW;   // compile-time: the staging index math is mul/shift
    float* Gs = lds;
    float* Xs = lds + MT * 32 * WG_GP;

    // this wave's tiles
    int tmt[WG_MAXT], boff[WG_MAXT];
    bool tok[WG_MAXT];
    int ntile = 0;
#pragma unroll
    for (int i = 0; i < WG_MAXT; ++i) {
        const int t = wave + i * 4;
        tok[i] = t < MT * NT;
        const int mt = tok[i] ? t / NT : 0, nt = tok[i] ? t - mt * NT : 0;
        tmt[i] = mt;
        const int nidx = nt * 32 + j;
        const int ci_l = nidx / KK, tap = nidx - ci_l * KK;
        const bool nv = nidx < cin_here * KK;
        boff[i] = nv ? ci_l * XR * XC + (tap / KW) * XC + (tap % KW) : 0;
        if (tok[i]) ntile = i + 1;
    }
    f32x16 acc[WG_MAXT];
#pragma unroll
    for (int i = 0; i < WG_MAXT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int tiles_x = (a.wout + WG_TW - 1) / WG_TW, tiles_y = (a.hout + WG_TH - 1) / WG_TH;
    const int total = tiles_x * tiles_y * a.n;
    const size_t ghw = (size_t)a.hout * a.wout, xhw = (size_t)a.hin * a.win;
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int bx = tile % tiles_x, by = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int oy0 = by * WG_TH, ox0 = bx * WG_TW;
        __syncthreads();
        // staging is branch-free (clamped addresses + select) so that the unrolled loads are all in flight together
#pragma unroll 8
        for (int e = threadIdx.x; e < MT * 32 * WG_PX; e += 256) {
            const int co = e / WG_PX, p = e - co * WG_PX;
            const int oy = oy0 + (p >> 5), ox = ox0 + (p & 31);
            const bool ok = co < a.cout && oy < a.hout && ox < a.wout;
            const int coc = min(co, a.cout - 1), oyc = min(oy, a.hout - 1), oxc = min(ox, a.wout - 1);
            const size_t o = ((size_t)n * a.g_ctotal + a.g_coff + (size_t)coc * a.g_cmul) * ghw + (size_t)oyc * a.wout + oxc;
            float v = a.g[o];
            if (a.gate) v *= a.gate[o] > 0.f ? 1.f : 0.1f;
            Gs[co * WG_GP + p] = ok ? v : 0.f;
        }
        const int gy0 = oy0 * S - (a.kh ? a.ph : a.pad), gx0 = ox0 * S - (a.kh ? a.pw : a.pad);
#pragma unroll 8
        for (int e = threadIdx.x; e < cin_here * XR * XC; e += 256) {
            const int ci_l = e / (XR * XC), rem = e - ci_l * (XR * XC);
            const int ry = rem / XC, rx = rem - ry * XC;
            const int iy = gy0 + ry, ix = gx0 + rx;
            const bool ok = iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win;
            const int iyc = min(max(iy, 0), a.hin - 1), ixc = min(max(ix, 0), a.win - 1);
            const float v = a.x[((size_t)n * a.x_ctotal + a.x_coff + ci0 + ci_l) * xhw + (size_t)iyc * a.win + ixc];
            Xs[e] = ok ? v : 0.f;
        }
        __syncthreads();
#pragma unroll 2
        for (int p = 0; p < WG_PX; p += 2) {
            const int pp = p + h;
            const int xoff = ((pp >> 5) * S) * XC + (pp & 31) * S;
#pragma unroll
            for (int i = 0; i < WG_MAXT; ++i) {
                if (i < ntile) {
                    const float av = Gs[(tmt[i] * 32 + j) * WG_GP + pp];
                    const float bv = Xs[boff[i] + xoff];
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < WG_MAXT; ++i) {
        if (!tok[i]) continue;
        const int t = wave + i * 4;
        const int mt = t / NT, nt = t - mt * NT;
        const int nidx = nt * 32 + j;
        if (nidx >= cin_here * KK) continue;
        const int ci_l = nidx / KK, tap = nidx - ci_l * KK;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (co < a.cout) atomicAdd(&a.dw[((size_t)co * (a.dw_cin ? a.dw_cin : a.cin) + a.dw_coff + ci0 + ci_l) * KK + tap], acc[i][r]);
        }
    }
}

// ---- the same GEMM for small feature maps (the 1/64-grid tail: 5x6 pixels at MVSEC size).  A 4x32-pixel tile of a 5x6 map is
// 81 % padding and every block ends with one atomic per weight, so the tail's eight launches cost ~100 us each.  Here a
// "tile" is `ipt` whole images (ipt * hout * wout <= 128 pixels): G is staged as [co][image, pixel], X as one zero-padded
// plane per (channel, image), and the B-operand offset of a pixel comes from a 128-entry table in LDS.
template <int K>
__global__ __launch_bounds__(256) void wgrad_small_kernel(WgradBatch batch, int ipt) {
    const WgradArgs& a = batch.job[blockIdx.z];
    constexpr int KK = K * K, PADK = K / 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];     // MT*32*WG_GP + cin_here*ipt*XR*XC floats + 128 ints
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int ci0 = blockIdx.y * WG_CI;
    if (ci0 >= a.cin) return;
    const int cin_here = min(WG_CI, a.cin - ci0);
    const int MT = (a.cout + 31) >> 5;
    const int NT = (cin_here * KK + 31) >> 5;
    const int XR = a.hin + 2 * PADK, XC = a.win + 2 * PADK, PL = XR * XC;
    const int hw = a.hout * a.wout;
    float* Gs = lds;
    float* Xs = lds + MT * 32 * WG_GP;
    int* xtab = reinterpret_cast<int*>(Xs + cin_here * ipt * PL);
    if (threadIdx.x < WG_PX) {
        const int p = threadIdx.x, img = p / hw, q = p - img * hw;
        const int y = q / a.wout, x = q - y * a.wout;
        xtab[p] = img < ipt ? img * PL + y * XC + x : 0;            // pixels past ipt*hw carry G = 0
    }
    int tmt[WG_MAXT], boff[WG_MAXT];
    bool tok[WG_MAXT];
    int ntile = 0;
#pragma unroll
    for (int i = 0; i < WG_MAXT; ++i) {
        const int t = wave + i * 4;
        tok[i] = t < MT * NT;
        const int mt = tok[i] ? t / NT : 0, nt = tok[i] ? t - mt * NT : 0;
        tmt[i] = mt;
        const int nidx = nt * 32 + j;
        const int ci_l = nidx / KK, tap = nidx - ci_l * KK;
        const bool nv = nidx < cin_here * KK;
        boff[i] = nv ? ci_l * ipt * PL + (tap / K) * XC + (tap % K) : 0;
        if (tok[i]) ntile = i + 1;
    }
    f32x16 acc[WG_MAXT];
#pragma unroll
    for (int i = 0; i < WG_MAXT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int total = (a.n + ipt - 1) / ipt;
    const size_t xhw = (size_t)a.hin * a.win;
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int n0 = tile * ipt;
        __syncthreads();
#pragma unroll 4
        for (int e = threadIdx.x; e < MT * 32 * WG_PX; e += 256) {
            const int co = e / WG_PX, p = e - co * WG_PX;
            const int img = p / hw, q = p - img * hw;
            const bool ok = co < a.cout && img < ipt && n0 + img < a.n;
            const int coc = min(co, a.cout - 1), nn = min(n0 + min(img, ipt - 1), a.n - 1);
            const size_t o = ((size_t)nn * a.g_ctotal + a.g_coff + (size_t)coc * a.g_cmul) * hw + q;
            float v = a.g[o];
            if (a.gate) v *= a.gate[o] > 0.f ? 1.f : 0.1f;
            Gs[co * WG_GP + p] = ok ? v : 0.f;
        }
#pragma unroll 4
        for (int e = threadIdx.x; e < cin_here * ipt * PL; e += 256) {
            const int ci_l = e / (ipt * PL), rem = e - ci_l * (ipt * PL);
            const int img = rem / PL, r2 = rem - img * PL;
            const int ry = r2 / XC, rx = r2 - ry * XC;
            const int iy = ry - PADK, ix = rx - PADK;
            const bool ok = iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win && n0 + img < a.n;
            const int iyc = min(max(iy, 0), a.hin - 1), ixc = min(max(ix, 0), a.win - 1), nn = min(n0 + img, a.n - 1);
            const float v = a.x[((size_t)nn * a.x_ctotal + a.x_coff + ci0 + ci_l) * xhw + (size_t)iyc * a.win + ixc];
            Xs[e] = ok ? v : 0.f;
        }
        __syncthreads();
        const int kend = min(WG_PX, ((ipt * hw + 1) & ~1));
        for (int p = 0; p < kend; p += 2) {
            const int pp = p + h;
            const int xoff = xtab[pp];
#pragma unroll
            for (int i = 0; i < WG_MAXT; ++i) {
                if (i < ntile) {
                    const float av = Gs[(tmt[i] * 32 + j) * WG_GP + pp];
                    const float bv = Xs[boff[i] + xoff];
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < WG_MAXT; ++i) {
        if (!tok[i]) continue;
        const int t = wave + i * 4;
        const int mt = t / NT, nt = t - mt * NT;
        const int nidx = nt * 32 + j;
        if (nidx >= cin_here * KK) continue;
        const int ci_l = nidx / KK, tap = nidx - ci_l * KK;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (co < a.cout) atomicAdd(&a.dw[((size_t)co * (a.dw_cin ? a.dw_cin : a.cin) + a.dw_coff + ci0 + ci_l) * KK + tap], acc[i][r]);
        }
    }
}

// ---- clip_grad_norm_ + AdamW (train_mvsec.py:178-183,255-256): sum of squares, then the fused update
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, double* __restrict__ out) {
    __shared__ double sh[4];
    double s = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += (double)g[i] * (double)g[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, sh[0] + sh[1] + sh[2] + sh[3]);
}

// A gradient with an inf or a NaN in it (sum of squares not finite) skips the step, as GradScaler.step does for the reference
// (train_mvsec.py:257: no parameter, no moment changes, and the optimizer's own step count - the bias corrections - does not advance).
// `nskip` counts the skipped steps so far; it is advanced by the re-packing launch AFTER this one (every thread reads the old value).
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, const double* __restrict__ sumsq, float clip,
                                                    float lr, float wd, float eps, float b1, float b2, long step,
                                                    const int* __restrict__ nskip) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double ss = *sumsq;
    if (!(ss <= 1.7976931348623157e308)) return;             // inf or NaN
    const float eff = fmaxf((float)(step - (long)*nskip), 1.f);           // optimizer steps taken, this one included
    const float bc1 = 1.f - powf(b1, eff), bc2 = 1.f - powf(b2, eff);
    float coef = 1.f;
    if (clip > 0.f) {
        const float norm = (float)sqrt(ss);
        coef = fminf(clip / (norm + 1e-6f), 1.f);            // torch.nn.utils.clip_grad_norm_
    }
    const float gi = g[i] * coef;
    float pi = p[i] * (1.f - lr * wd);                        // decoupled weight decay
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
}

}  // namespace

int repack_launch(const float* flat, const int* idx, float* arena, long n, hipStream_t st) {
    hipLaunchKernelGGL(repack_kernel, dim3(nblocks(n)), dim3(256), 0, st, flat, idx, arena, n, (double*)nullptr, (int*)nullptr);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int tr_repack_after_step_launch(const float* flat, const int* idx, float* arena, long n, double* sumsq, int* nskip, hipStream_t st) {
    hipLaunchKernelGGL(repack_kernel, dim3(nblocks(n)), dim3(256), 0, st, flat, idx, arena, n, sumsq, nskip);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int tr_loss_launch(const float* flow, const float* gt, const float* valid, float* dflow, int batch, int hw, float weight,
                   double* stats, hipStream_t st) {
    hipLaunchKernelGGL(loss_kernel, dim3(std::min<long>(nblocks((long)batch * hw), 512)), dim3(256), 0, st, flow, gt, valid, dflow, batch, hw, weight, stats);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int tr_upsample_bwd_launch(const float* d, float* tmp, float* out, int nc, int oh, int ow, int h, int w, hipStream_t st) {
    const char* ex = getenv("EEM_UPBWD_THREADS");                     // (=1, read per call: the thread-per-target form, for the equality test)
    if (ow >= 64 && ow <= 8192 && !(ex && ex[0] == '1'))
        hipLaunchKernelGGL(upbwd_x_rows_kernel, dim3((unsigned)(((long)nc * oh + 3) / 4)), dim3(256), 4 * ow * sizeof(float), st, d, tmp, (long)nc * oh, ow, w);
    else
        hipLaunchKernelGGL(upbwd_x_kernel, dim3(nblocks((long)nc * oh * w)), dim3(256), 0, st, d, tmp, (long)nc * oh, ow, w);
    hipLaunchKernelGGL(upbwd_y_kernel, dim3(nblocks((long)nc * h * w)), dim3(256), 0, st, tmp, out, nc, oh, h, w);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int tr_pool_bwd_launch(const float* dpool, float* g, long nc, int h, int w, int k, int gh, int gw, int accumulate, const float* gate,
                       hipStream_t st) {
    if ((w & 3) == 0 && (k & 3) == 0 && (((uintptr_t)g | (uintptr_t)gate) & 15) == 0)
        hipLaunchKernelGGL(poolbwd4_kernel, dim3(nblocks(nc * h * (w / 4))), dim3(256), 0, st, dpool, g, nc, h, w, k, gh, gw, accumulate, gate);
    else
        hipLaunchKernelGGL(poolbwd_kernel, dim3(nblocks(nc * h * w)), dim3(256), 0, st, dpool, g, nc, h, w, k, gh, gw, accumulate, gate);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int tr_corr_bwd_launch_jobs(const CorrBwdJob* jobs, int njobs, int batch, int h, int w, const int* taps, int ntaps, hipStream_t st) {
    EEM_REQUIRE(jobs && njobs >= 1 && njobs <= 3, "tr_corr_bwd_launch_jobs: njobs=%d", njobs);
    CorrBwdJobs jb;
    int cmax = 0;
    for (int i = 0; i < njobs; ++i) { jb.job[i] = jobs[i]; cmax = std::max(cmax, jobs[i].c); }
    hipLaunchKernelGGL(corrbwd_kernel, dim3(nblocks((long)batch * cmax * h * w), njobs), dim3(256), 0, st, jb, batch, h, w, taps, ntaps);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int tr_corr_bwd_launch(const float* dcv, int dcv_ctotal, const float* f1, const float* f2, float* d1, float* d2, int batch, int c,
                       int h, int w, const int* taps, int ntaps, hipStream_t st) {
    CorrBwdJob j{dcv, f1, f2, d1, d2, dcv_ctotal, c};
    return tr_corr_bwd_launch_jobs(&j, 1, batch, h, w, taps, ntaps, st);
}

int tr_bias_grad_launch_batch(const BiasJob* jobs, int njobs, hipStream_t st) {
    EEM_REQUIRE(jobs && njobs >= 1 && njobs <= WGRAD_MAX_JOBS, "tr_bias_grad_launch_batch: njobs=%d", njobs);
    BiasBatch b;
    long total = 0;
    int cmax = 0;
    for (int i = 0; i < njobs; ++i) {
        b.job[i] = jobs[i];
        total = std::max(total, (long)jobs[i].n * jobs[i].hw);
        cmax = std::max(cmax, jobs[i].cout);
    }
    int chunks = (int)((total + 256 * 64 - 1) / (256 * 64));
    chunks = chunks < 1 ? 1 : (chunks > 256 ? 256 : chunks);
    hipLaunchKernelGGL(biasgrad_kernel, dim3(chunks, cmax, njobs), dim3(256), 0, st, b);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int tr_bias_grad_launch(const float* g, const float* gate, int g_ctotal, int g_coff, int g_cmul, int cout, int n, int hw, float* db,
                        hipStream_t st) {
    BiasJob j;
    j.g = g; j.gate = gate; j.db = db; j.g_ctotal = g_ctotal; j.g_coff = g_coff; j.g_cmul = g_cmul; j.cout = cout; j.n = n; j.hw = hw;
    return tr_bias_grad_launch_batch(&j, 1, st);
}

int tr_wgrad_launch_batch(const WgradArgs* jobs, int njobs, hipStream_t st) {
    EEM_REQUIRE(jobs && njobs >= 1 && njobs <= WGRAD_MAX_JOBS, "tr_wgrad_launch_batch: njobs=%d", njobs);
    WgradBatch b;
    int nchunk = 0, workers = 0;
    size_t lds_bytes = 0;
    for (int i = 0; i < njobs; ++i) {
        const WgradArgs& a = jobs[i];
        const int akh = a.kh ? a.kh : a.k, akw = a.kh ? a.kw : a.k;
        EEM_REQUIRE(akh >= 1 && akw >= 1 && a.cout >= 1 && a.cout <= 128 && a.cin >= 1, "tr_wgrad_launch: unsupported conv");
        EEM_REQUIRE(a.stride == 1 || a.stride == 2, "tr_wgrad_launch: stride %d", a.stride);
        EEM_REQUIRE(a.k == jobs[0].k && a.kh == jobs[0].kh && a.kw == jobs[0].kw && a.stride == jobs[0].stride,
                    "tr_wgrad_launch_batch: jobs differ in kernel size / stride");
        const int nc = (a.cin + WG_CI - 1) / WG_CI;
        const int mt = (a.cout + 31) / 32;
        const int nt = ((a.cin < WG_CI ? a.cin : WG_CI) * akh * akw + 31) / 32;
        EEM_REQUIRE((mt * nt + 3) / 4 <= WG_MAXT, "tr_wgrad_launch: %d x %d tiles exceed the per-wave budget", mt, nt);
        const int tiles = ((a.wout + WG_TW - 1) / WG_TW) * ((a.hout + WG_TH - 1) / WG_TH) * a.n;
        int w = 1024 / (nc * njobs);
        w = w < 1 ? 1 : w;
        w = tiles < w ? tiles : w;
        const int xr = (WG_TH - 1) * a.stride + akh, xc = (WG_TW - 1) * a.stride + akw;
        const size_t lb = ((size_t)mt * 32 * WG_GP + (size_t)(a.cin < WG_CI ? a.cin : WG_CI) * xr * xc) * sizeof(float);
        nchunk = nc > nchunk ? nc : nchunk;
        workers = w > workers ? w : workers;
        lds_bytes = lb > lds_bytes ? lb : lds_bytes;
        b.job[i] = a;
    }
    // small maps (stride 1, same shape in every job, >= 2 images per 128-pixel tile): whole images per tile
    {
        const WgradArgs& a0 = jobs[0];
        const int hw0 = a0.hout * a0.wout;
        const char* es = getenv("EEM_NO_WGRAD_SMALL");
        bool small = !(es && es[0] == '1') && a0.kh == 0 && a0.dw_cin == 0 && a0.stride == 1 && hw0 * 2 <= WG_PX && a0.hin == a0.hout && a0.win == a0.wout;
        for (int i = 1; i < njobs; ++i)
            small = small && jobs[i].hout == a0.hout && jobs[i].wout == a0.wout && jobs[i].n == a0.n && jobs[i].hin == a0.hin &&
                    jobs[i].win == a0.win;
        if (small) {
            const int ipt = WG_PX / hw0;
            const int pl = (a0.hin + 2 * (a0.k / 2)) * (a0.win + 2 * (a0.k / 2));
            size_t lb = 0;
            for (int i = 0; i < njobs; ++i) {
                const int mt = (jobs[i].cout + 31) / 32, ch = jobs[i].cin < WG_CI ? jobs[i].cin : WG_CI;
                lb = std::max(lb, ((size_t)mt * 32 * WG_GP + (size_t)ch * ipt * pl + WG_PX) * sizeof(float));
            }
            if (lb <= 160 * 1024) {
                const int tiles = (a0.n + ipt - 1) / ipt;
                int w = 1024 / (nchunk * njobs);
                w = w < 1 ? 1 : (w > tiles ? tiles : w);
                static bool small_attr = false;
                if (!small_attr) {
                    EEM_HIP_CHECK(hipFuncSetAttribute((const void*)wgrad_small_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    EEM_HIP_CHECK(hipFuncSetAttribute((const void*)wgrad_small_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    small_attr = true;
                }
                if (a0.k == 3) hipLaunchKernelGGL((wgrad_small_kernel<3>), dim3(w, nchunk, njobs), dim3(256), lb, st, b, ipt);
                else hipLaunchKernelGGL((wgrad_small_kernel<1>), dim3(w, nchunk, njobs), dim3(256), lb, st, b, ipt);
                EEM_HIP_CHECK(hipGetLastError());
                return EEM_OK;
            }
        }
    }
    dim3 grid(workers, nchunk, njobs);
    const WgradArgs& a = jobs[0];
    const int kh = a.kh ? a.kh : a.k, kw = a.kh ? a.kw : a.k;
#define WG_CASE(KH_, KW_, S_)                                                                                                   \
    if (kh == KH_ && kw == KW_ && a.stride == S_) {                                                                             \
        static bool attr = false;                                                                                               \
        if (!attr) {                                                                                                            \
            EEM_HIP_CHECK(hipFuncSetAttribute((const void*)wgrad_kernel<KH_, KW_, S_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                              160 * 1024));                                                                     \
            attr = true;                                                                                                        \
        }                                                                                                                       \
        hipLaunchKernelGGL((wgrad_kernel<KH_, KW_, S_>), grid, dim3(256), lds_bytes, st, b);                                    \
        EEM_HIP_CHECK(hipGetLastError());                                                                                       \
        return EEM_OK;                                                                                                          \
    }
    WG_CASE(3, 3, 1) WG_CASE(3, 3, 2) WG_CASE(1, 1, 1) WG_CASE(1, 1, 2) WG_CASE(1, 5, 1) WG_CASE(5, 1, 1) WG_CASE(7, 7, 1) WG_CASE(7, 7, 2)
#undef WG_CASE
    eem_set_error("tr_wgrad_launch: kernel %dx%d stride %d is not built", kh, kw, a.stride);
    return EEM_ERR_ARG;
}

// ---- weight gradient of a 3x3 stride-1 conv with at most 8 couts (E-RAFT's flow head 256 -> 2, model/update.py:10; EEMFlow+'s
// 32 -> 2 flow convs) on the vector pipe: on the matrix cores a 2-cout layer is 94 % padding (the generic kernel: 212 us for 88 MFLOP).
// A block = one input channel x a range of pixels; a thread accumulates cout x 9 products for its pixels (nine neighbours of X - clamped
// addresses, the condition on the value - and cout values of G), the block sums them (DPP inside a wave, LDS across the four) and
// leaves with one atomic per weight.  The bias gradient rides in the blocks of input channel 0.
template <int COUT>
__global__ __launch_bounds__(256) void wgrad_few_kernel(WgradArgs a, int per_block) {
    __shared__ float sh[4][COUT * 9 + COUT];
    const int ci = blockIdx.x;
    const int hw = a.hout * a.wout;
    const long total = (long)a.n * hw;
    const long p0 = (long)blockIdx.y * per_block, p1 = p0 + per_block < total ? p0 + per_block : total;
    float acc[COUT][9], bs[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
        bs[c] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[c][t] = 0.f;
    }
    const float* __restrict__ g = a.g;
    const float* __restrict__ x = a.x;
    for (long p = p0 + threadIdx.x; p < p1; p += 256) {
        const int n = (int)(p / hw), q = (int)(p - (long)n * hw);
        const int y = q / a.wout, xx = q - y * a.wout;
        float gv[COUT];
#pragma unroll
        for (int c = 0; c < COUT; ++c) gv[c] = c < a.cout ? g[((size_t)n * a.g_ctotal + a.g_coff + c) * hw + q] : 0.f;
        const float* xp = x + ((size_t)n * a.x_ctotal + a.x_coff + ci) * hw;
        float xv[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = y + ky - 1, ix = xx + kx - 1;
                const bool in = iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win;
                const float v = xp[(size_t)min(max(iy, 0), a.hin - 1) * a.win + min(max(ix, 0), a.win - 1)];
                xv[ky * 3 + kx] = in ? v : 0.f;
            }
#pragma unroll
        for (int c = 0; c < COUT; ++c) {
            bs[c] += gv[c];
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[c][t] += gv[c] * xv[t];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float v = lane_group_sum<16>(acc[c][t]);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane == 0) sh[wave][c * 9 + t] = v;
        }
        float v = lane_group_sum<16>(bs[c]);
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane == 0) sh[wave][COUT * 9 + c] = v;
    }
    __syncthreads();
    const int dwcin = a.dw_cin ? a.dw_cin : a.cin;
    if ((int)threadIdx.x < a.cout * 9) {
        const int c = threadIdx.x / 9, t = threadIdx.x - c * 9;
        atomicAdd(&a.dw[((size_t)c * dwcin + a.dw_coff + ci) * 9 + t], sh[0][c * 9 + t] + sh[1][c * 9 + t] + sh[2][c * 9 + t] + sh[3][c * 9 + t]);
    }
    if (a.db && ci == 0 && (int)threadIdx.x < a.cout) {
        const int k = COUT * 9 + threadIdx.x;
        atomicAdd(&a.db[threadIdx.x], sh[0][k] + sh[1][k] + sh[2][k] + sh[3][k]);
    }
}

bool wgrad_few_supported(const WgradArgs& a) {
    const char* e = getenv("EEM_NO_WGRAD_FEW");                       // read per call: a test flips it inside one process
    if (e && e[0] == '1') return false;
    const int kh = a.kh ? a.kh : a.k, kw = a.kh ? a.kw : a.k, ph = a.kh ? a.ph : a.pad, pw = a.kh ? a.pw : a.pad;
    return kh == 3 && kw == 3 && ph == 1 && pw == 1 && a.stride == 1 && a.cout >= 1 && a.cout <= 8 && a.gate == nullptr && a.g_cmul == 1 &&
           a.nxseg == 0 && a.hin == a.hout && a.win == a.wout;
}

int wgrad_few_launch(const WgradArgs& a, hipStream_t st) {
    const long total = (long)a.n * a.hout * a.wout;
    int splits = (512 + a.cin - 1) / a.cin;                            // ~2 blocks per CU
    const long min_px = 2048;
    if ((long)splits * min_px > total) splits = (int)((total + min_px - 1) / min_px);
    if (splits < 1) splits = 1;
    const int per_block = (int)((total + splits - 1) / splits);
    if (a.cout <= 2) hipLaunchKernelGGL((wgrad_few_kernel<2>), dim3(a.cin, splits), dim3(256), 0, st, a, per_block);
    else if (a.cout <= 4) hipLaunchKernelGGL((wgrad_few_kernel<4>), dim3(a.cin, splits), dim3(256), 0, st, a, per_block);
    else hipLaunchKernelGGL((wgrad_few_kernel<8>), dim3(a.cin, splits), dim3(256), 0, st, a, per_block);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int tr_wgrad_launch(const WgradArgs& a, hipStream_t st) {
    if (wgrad_ring_supported(a) && wgrad_ring_preferred(a)) return wgrad_ring_launch(a, st);
    if (wgrad_enc_supported(a)) return wgrad_enc_launch(a, st);
    return tr_wgrad_launch_batch(&a, 1, st);
}

int tr_sumsq_launch(const float* g, long n, double* out, hipStream_t st) {
    hipLaunchKernelGGL(sumsq_kernel, dim3(512), dim3(256), 0, st, g, n, out);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int tr_adamw_launch(float* p, const float* g, float* m, float* v, long n, const double* sumsq, float clip, float lr, float wd,
                    float eps, float b1, float b2, long step, int* nskip, hipStream_t st) {
    hipLaunchKernelGGL(adamw_kernel, dim3(nblocks(n)), dim3(256), 0, st, p, g, m, v, n, sumsq, clip, lr, wd, eps, b1, b2, step, nskip);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}
