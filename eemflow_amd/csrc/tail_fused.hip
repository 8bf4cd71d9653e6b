// Two fused launches of EEMFlow's 1/64-grid tail (EEMFlow.py:144-181).  The tail is latency-bound: 13 launches of 3-8 us
// each for 0.3 GFLOP, every one paying a kernel boundary plus a memory round trip.  These two take five of them away:
//
//   tail_head_kernel   stage pooling finish + 9x9 local correlation (53 taps) + rconv_k   (was: pool_finalize, corr, rconv)
//       Nothing in it waits for anything else in it: the correlation and the rconv input gathers read the conv epilogues'
//       pooling PARTIAL sums directly (2-4 loads per pooled value, all in flight) instead of a finished pooled map; the
//       pooled maps themselves are still written, by extra blocks of the same launch (parity tests and the training
//       backward read them).
//   tail_up_kernel     out_conv 1x1 (6 -> 2) + bilinear upsample   (was: out_conv, upsample)
//       A block owns an output tile no larger than one coarse cell, so at most 3x3 coarse pixels reach it; it computes
//       their out_conv values itself, keeps them in LDS and interpolates (x weights once per thread, rows walked);
//       `coarse` is written as a side output.  (Measured and dropped: conv7 in the same launch, 4 lanes per value on
//       plain FMAs - 13.7 us against 12.3 us for conv7 + out_conv + upsample as three launches.)
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ pooled operand
__device__ __forceinline__ float pooled_load(const PooledSrc& s, const float* chan_base, int y, int x) {
    const float* p = chan_base + (size_t)y * s.ystride + x;
    float v = p[0];
    for (int i = 1; i < s.rows; ++i) v += p[(size_t)i * s.rstride];
    return v * s.scale;
}

template <int CG>
__device__ __forceinline__ void rconv_role(const TailHeadArgs& a, int blk, f32x4 (*part)[64]) {
    // one block = 16 pixels x 16 couts of one (scale, sample); wave t = filter tap t (see tail_conv_kernel)
    const int lane = threadIdx.x & 63, t = threadIdx.x >> 6;
    const int g = a.gh * a.gw;
    const int ptiles = ceil_div(g, 16);
    const int pt = blk % ptiles;
    const int kb = blk / ptiles;
    const int k = kb % 3, b = kb / 3;
    const PooledSrc src = a.src[k];
    const int cin = a.c[k];
    const int j = lane & 15, gq = lane >> 4;
    const int p = pt * 16 + j;
    const bool pvalid = p < g;
    const int y = p / a.gw, x = p - y * a.gw;
    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
    const bool valid = pvalid && yy >= 0 && yy < a.gh && xx >= 0 && xx < a.gw;
    const float* wp = a.rw[k] + (size_t)t * CG * 64 + lane;
    const float* img = src.base + (size_t)b * src.nstride;                 // events1 half: image b
    float av[CG], bv[CG];
#pragma unroll
    for (int q = 0; q < CG; ++q) {
        const int c = q * 4 + gq;
        av[q] = wp[(size_t)q * 64];
        bv[q] = (valid && c < cin) ? pooled_load(src, img + (size_t)c * src.cstride, yy, xx) : 0.f;
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < CG; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], acc, 0, 0, 0);
    part[t][lane] = acc;
    __syncthreads();
    if (t != 0) return;
    acc = part[0][lane];
#pragma unroll
    for (int q = 1; q < 9; ++q) acc += part[q][lane];                       // fixed order: bitwise repeatable
    if (pvalid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = gq * 4 + r;
            float v = acc[r] + a.rb[k][co];
            v = v > 0.f ? v : 0.1f * v;
            a.cat[k][((size_t)b * a.cat_ctotal + a.ntaps + co) * g + p] = v;
        }
    }
}

__global__ __launch_bounds__(576) void tail_head_kernel(TailHeadArgs a) {
    __shared__ f32x4 part[9][64];
    const int blk = blockIdx.x;
    const int g = a.gh * a.gw;
    if (blk < a.nblk_rconv) {
        const int ptiles = ceil_div(g, 16);
        const int k = (blk / ptiles) % 3;
        if (a.c[k] <= 16) rconv_role<4>(a, blk, part);
        else if (a.c[k] <= 32) rconv_role<8>(a, blk, part);
        else rconv_role<16>(a, blk, part);
        return;
    }
    if (blk < a.nblk_rconv + a.nblk_corr) {
        // four adjacent lanes share one output (scale, sample, tap, pixel) and split its channels (see corr_kernel)
        const int per_job = a.batch * a.ntaps * g;
        const int gid = (blk - a.nblk_rconv) * 576 + threadIdx.x;
        int idx = gid >> 2;
        const int sub = gid & 3;
        const bool live = idx < per_job * 3;
        if (!live) idx = 0;
        const int k = idx / per_job;
        idx -= k * per_job;
        const PooledSrc src = a.src[k];
        const int cin = a.c[k];
        const int p = idx % g; idx /= g;
        const int ti = idx % a.ntaps;
        const int b = idx / a.ntaps;
        const int y = p / a.gw, x = p - y * a.gw;
        const int tap = a.taps[ti];
        const int yy = y + tap / 9 - 4, xx = x + tap % 9 - 4;
        float s = 0.f;
        if (live && yy >= 0 && yy < a.gh && xx >= 0 && xx < a.gw) {
            const float* i1 = src.base + (size_t)b * src.nstride;
            const float* i2 = src.base + (size_t)(a.batch + b) * src.nstride;
#pragma unroll 4
            for (int c = sub; c < cin; c += 4)
                s = fmaf(pooled_load(src, i1 + (size_t)c * src.cstride, y, x), pooled_load(src, i2 + (size_t)c * src.cstride, yy, xx), s);
        }
        s = dpp_add<0xB1>(s);
        s = dpp_add<0x4E>(s);
        if (live && sub == 0) a.cat[k][((size_t)b * a.cat_ctotal + ti) * g + p] = s / (float)cin;
        return;
    }
    // pooled maps [2B][C][gh][gw] as a side output
    int idx = (blk - a.nblk_rconv - a.nblk_corr) * 576 + threadIdx.x;
    for (int k = 0; k < 3; ++k) {
        const int total = 2 * a.batch * a.c[k] * g;
        if (idx < total) {
            if (a.pool_out[k] == nullptr) return;
            const int x = idx % a.gw, y = (idx / a.gw) % a.gh, nc = idx / g;
            const int n = nc / a.c[k], c = nc - n * a.c[k];
            a.pool_out[k][idx] = pooled_load(a.src[k], a.src[k].base + (size_t)n * a.src[k].nstride + (size_t)c * a.src[k].cstride, y, x);
            return;
        }
        idx -= total;
    }
}

// ------------------------------------------------------------------------------------------------ conv7 + out_conv + upsample
// F.interpolate(mode='bilinear', align_corners=False): src = max(scale*(dst+0.5)-0.5, 0)   (same arithmetic as upsample_kernel)
__device__ __forceinline__ void src_index2(float scale, int dst, int in_size, int& i0, int& i1, float& l1) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

__global__ __launch_bounds__(256) void tail_up_kernel(TailUpArgs a) {
    __shared__ float coarse_s[2][9];        // out_conv output at the block's <= 3x3 coarse pixels
    float* __restrict__ out = a.io ? (float*)a.io[2] : a.out;
    const int tid = threadIdx.x;
    const int txn = ceil_div(a.ow, a.tx), tyn = ceil_div(a.oh, a.ty);
    int blk = blockIdx.x;
    const int bx = blk % txn; blk /= txn;
    const int by = blk % tyn;
    const int b = blk / tyn;
    const int y0 = by * a.ty, x0 = bx * a.tx;
    const int y1 = min(y0 + a.ty, a.oh), x1 = min(x0 + a.tx, a.ow);
    const float sy = (float)a.gh / (float)a.oh, sx = (float)a.gw / (float)a.ow;
    int cy0, cx0, t0; float tl;
    src_index2(sy, y0, a.gh, cy0, t0, tl);
    src_index2(sx, x0, a.gw, cx0, t0, tl);
    const int g = a.gh * a.gw;
    // ---- out_conv 1x1 over the six decoder flow channels at the 3x3 coarse pixels from (cy0, cx0)
    if (tid < 18) {
        const int c = tid / 9, cp = tid - c * 9;
        const int cy = cy0 + cp / 3, cx = cx0 + cp % 3;
        float s = 0.f;
        if (cy < a.gh && cx < a.gw) {
            const float* f = a.flowcat + (size_t)b * 6 * g + cy * a.gw + cx;
            float v[6];
#pragma unroll
            for (int m = 0; m < 6; ++m) v[m] = f[(size_t)m * g];
            s = a.bo[c];
#pragma unroll
            for (int m = 0; m < 6; ++m) s = fmaf(a.wo[c * 6 + m], v[m], s);
            if (a.coarse) a.coarse[((size_t)b * 2 + c) * g + cy * a.gw + cx] = s;   // every block that holds it writes the same value
        }
        coarse_s[c][cp] = s;
    }
    __syncthreads();
    // ---- bilinear upsample of the tile: a thread keeps one quad of columns (its four x weights are computed once) and
    // walks rows; (rows x channels) are dealt over the thread groups
    const int xq = (x1 - x0 + 3) >> 2;                       // column quads of the tile (<= 64 for tiles up to 256 wide)
    const int rows = y1 - y0;
    const bool vec = (a.ow & 3) == 0 && (a.tx & 3) == 0 && a.out_aligned16;
    const int q = tid % xq, grp = tid / xq, ngrp = 256 / xq;
    if (grp >= ngrp) return;
    int xa[4], xb[4]; float lx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ox = x0 + q * 4 + i;
        src_index2(sx, ox < a.ow ? ox : a.ow - 1, a.gw, xa[i], xb[i], lx[i]);
        xa[i] -= cx0; xb[i] -= cx0;
    }
    for (int e = grp; e < 2 * rows; e += ngrp) {
        const int c = e / rows, r = e - c * rows;
        const int oy = y0 + r;
        int ya, yb; float ly;
        src_index2(sy, oy, a.gh, ya, yb, ly);
        const float* ra = coarse_s[c] + (ya - cy0) * 3;
        const float* rb = coarse_s[c] + (yb - cy0) * 3;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float top = (1.f - lx[i]) * ra[xa[i]] + lx[i] * ra[xb[i]];
            const float bot = (1.f - lx[i]) * rb[xa[i]] + lx[i] * rb[xb[i]];
            v[i] = (1.f - ly) * top + ly * bot;
        }
        float* dst = out + (((size_t)b * 2 + c) * a.oh + oy) * a.ow + x0 + q * 4;
        if (vec && x0 + q * 4 + 3 < x1) {
            *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (x0 + q * 4 + i < x1) dst[i] = v[i];
        }
    }
}

}  // namespace

int tail_head_launch(const TailHeadArgs& a0, hipStream_t stream) {
    TailHeadArgs a = a0;
    const int g = a.gh * a.gw;
    a.nblk_rconv = ceil_div(g, 16) * 3 * a.batch;
    a.nblk_corr = (int)(((long)3 * a.batch * a.ntaps * g * 4 + 575) / 576);
    long pool_elems = 0;
    for (int k = 0; k < 3; ++k) pool_elems += (long)2 * a.batch * a.c[k] * g;
    const int nblk_pool = (int)((pool_elems + 575) / 576);
    hipLaunchKernelGGL(tail_head_kernel, dim3(a.nblk_rconv + a.nblk_corr + nblk_pool), dim3(576), 0, stream, a);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

bool tail_up_supported(int gh, int gw, int oh, int ow) { return oh >= 4 * gh && ow >= 4 * gw && ow / gw <= 1024; }

int tail_up_launch(const TailUpArgs& a0, hipStream_t stream) {
    TailUpArgs a = a0;
    EEM_REQUIRE(tail_up_supported(a.gh, a.gw, a.oh, a.ow), "tail_up_launch: output %dx%d too small for grid %dx%d", a.oh, a.ow, a.gh, a.gw);
    a.ty = a.oh / a.gh;                                   // a tile no taller / wider than one coarse cell: (ty - 1) * gh / oh < 1,
    a.tx = (a.ow / a.gw) & ~3;                            // so at most 3 coarse rows / columns reach it
    const int blocks = ceil_div(a.oh, a.ty) * ceil_div(a.ow, a.tx) * a.batch;
    hipLaunchKernelGGL(tail_up_kernel, dim3(blocks), dim3(256), 0, stream, a);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}
