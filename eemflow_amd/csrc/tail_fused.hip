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
// ROWS > 0: the row count is a compile-time constant and a caller's loads of several values are all in flight at once (the loop
// over a run-time row count is one memory round trip per row); ROWS = 0: any row count.  Same sum order either way.
template <int ROWS>
__device__ __forceinline__ float pooled_load(const PooledSrc& s, const float* chan_base, int y, int x) {
    const float* p = chan_base + (size_t)y * s.ystride + x;
    float v = p[0];
    if (ROWS > 0) {
#pragma unroll
        for (int i = 1; i < ROWS; ++i) v += p[(size_t)i * s.rstride];
    } else {
        for (int i = 1; i < s.rows; ++i) v += p[(size_t)i * s.rstride];
    }
    return v * s.scale;
}

// Roles are picked by blockIdx.z (0-2 rconv of stage k, 3-5 correlation of stage k, 6 pooled maps) and a correlation block's tap
// by blockIdx.y, so everything a block needs from the launch arguments - its stage's source, its tap - is ONE batch of scalar
// loads issued at its first instruction, and its operand loads are the second and last round trip before the stores (see
// tail_conv_kernel in tail.hip for what a round trip costs beside other frames' kernels).
template <int CG, int ROWS>
__device__ __forceinline__ void rconv_role(const TailHeadArgs& a, int k, int blk, f32x4 (*part)[64]) {
    // one block = 16 pixels x 16 couts of one (scale, sample); wave t = filter tap t (see tail_conv_kernel)
    const int lane = threadIdx.x & 63;
    const int t = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const PooledSrc src = a.src[k];
    const float* rw = a.rw[k];
    const float* rb = a.rb[k];
    float* cat = a.cat[k];
    const int gw = a.gw, gh = a.gh, cin = a.c[k];
    asm volatile("" ::"s"(src.base), "s"(src.rows), "s"(src.scale), "s"(rw), "s"(rb), "s"(cat), "s"(gw), "s"(gh), "s"(cin));
    const int g = gh * gw;
    const int ptiles = ceil_div(g, 16);
    if (blk >= ptiles * a.batch) return;
    const int b = blk / ptiles, pt = blk - b * ptiles;
    const int j = lane & 15, gq = lane >> 4;
    const int p = pt * 16 + j;
    const bool pvalid = p < g;
    const int y = p / gw, x = p - y * gw;
    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
    const bool valid = pvalid && yy >= 0 && yy < gh && xx >= 0 && xx < gw;
    const float* wp = rw + (size_t)t * CG * 64 + lane;
    const float* img = src.base + (size_t)b * src.nstride;                 // events1 half: image b
    float av[CG], bv[CG];
#pragma unroll
    for (int q = 0; q < CG; ++q) {
        const int c = q * 4 + gq;
        av[q] = wp[(size_t)q * 64];
        bv[q] = (valid && c < cin) ? pooled_load<ROWS>(src, img + (size_t)c * src.cstride, yy, xx) : 0.f;
    }
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    if (t == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bs[r] = rb[gq * 4 + r];
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
            float v = acc[r] + bs[r];
            v = v > 0.f ? v : 0.1f * v;
            cat[((size_t)b * a.cat_ctotal + a.ntaps + co) * g + p] = v;
        }
    }
}

// a wave = one (sample, tap, 16-pixel tile): four adjacent lanes share one output and split its channels (see corr_kernel);
// NC = channels per lane (cin / 4): all 2 * NC * ROWS loads of a lane are issued before the first product
template <int NC, int ROWS>
__device__ __forceinline__ void corr_role(const TailHeadArgs& a, int k) {
    const PooledSrc src = a.src[k];
    const int tap = a.tap[blockIdx.y];
    float* cat = a.cat[k];
    const int gw = a.gw, gh = a.gh, cin = a.c[k], batch = a.batch;
    asm volatile("" ::"s"(src.base), "s"(src.rows), "s"(src.scale), "s"(tap), "s"(cat), "s"(gw), "s"(gh), "s"(cin), "s"(batch));
    const int g = gh * gw;
    const int ptiles = ceil_div(g, 16);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tq = blockIdx.x * 9 + wave;
    if (tq >= ptiles * batch) return;
    const int b = tq / ptiles, pt = tq - b * ptiles;
    const int lane = threadIdx.x & 63;
    const int sub = lane & 3;
    const int p = pt * 16 + (lane >> 2);
    const bool live = p < g;
    const int y = p / gw, x = p - y * gw;
    const int yy = y + tap / 9 - 4, xx = x + tap % 9 - 4;
    float s = 0.f;
    if (live && yy >= 0 && yy < gh && xx >= 0 && xx < gw) {
        const float* i1 = src.base + (size_t)b * src.nstride;
        const float* i2 = src.base + (size_t)(batch + b) * src.nstride;
        if (cin == NC * 4) {
            float u[NC], v[NC];
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                u[i] = pooled_load<ROWS>(src, i1 + (size_t)(sub + 4 * i) * src.cstride, y, x);
                v[i] = pooled_load<ROWS>(src, i2 + (size_t)(sub + 4 * i) * src.cstride, yy, xx);
            }
#pragma unroll
            for (int i = 0; i < NC; ++i) s = fmaf(u[i], v[i], s);
        } else {
            for (int c = sub; c < cin; c += 4)
                s = fmaf(pooled_load<0>(src, i1 + (size_t)c * src.cstride, y, x), pooled_load<0>(src, i2 + (size_t)c * src.cstride, yy, xx), s);
        }
    }
    s = dpp_add<0xB1>(s);
    s = dpp_add<0x4E>(s);
    if (live && sub == 0) cat[((size_t)b * a.cat_ctotal + blockIdx.y) * g + p] = s / (float)cin;
}

__global__ __launch_bounds__(576) void tail_head_kernel(TailHeadArgs a) {
    __shared__ f32x4 part[9][64];
    const int role = blockIdx.z;
    const int blk = blockIdx.y * a.grid_x + blockIdx.x;
    if (role < 3) {
        const int rows = a.src[role].rows;                                // 16, 32, 64 input channels (EEMFlow.py:96-98)
        if (role == 0) { if (rows == 4) rconv_role<4, 4>(a, 0, blk, part); else if (rows == 1) rconv_role<4, 1>(a, 0, blk, part); else rconv_role<4, 0>(a, 0, blk, part); }
        else if (role == 1) { if (rows == 2) rconv_role<8, 2>(a, 1, blk, part); else if (rows == 1) rconv_role<8, 1>(a, 1, blk, part); else rconv_role<8, 0>(a, 1, blk, part); }
        else { if (rows == 1) rconv_role<16, 1>(a, 2, blk, part); else rconv_role<16, 0>(a, 2, blk, part); }
        return;
    }
    if (role < 6) {
        const int k = role - 3, rows = a.src[k].rows;
        if (k == 0) { if (rows == 4) corr_role<4, 4>(a, 0); else if (rows == 1) corr_role<4, 1>(a, 0); else corr_role<4, 0>(a, 0); }
        else if (k == 1) { if (rows == 2) corr_role<8, 2>(a, 1); else if (rows == 1) corr_role<8, 1>(a, 1); else corr_role<8, 0>(a, 1); }
        else { if (rows == 1) corr_role<16, 1>(a, 2); else corr_role<16, 0>(a, 2); }
        return;
    }
    // pooled maps [2B][C][gh][gw] as a side output
    const int g = a.gh * a.gw;
    int idx = blk * 576 + threadIdx.x;
    for (int k = 0; k < 3; ++k) {
        const int total = 2 * a.batch * a.c[k] * g;
        if (idx < total) {
            if (a.pool_out[k] == nullptr) return;
            const int nc = idx / g;
            const int n = nc / a.c[k], c = nc - n * a.c[k];
            const int pp = idx - nc * g;
            const int y = pp / a.gw, x = pp - y * a.gw;
            const float* cb = a.src[k].base + (size_t)n * a.src[k].nstride + (size_t)c * a.src[k].cstride;
            const int rows = a.src[k].rows;
            a.pool_out[k][idx] = rows == 4 ? pooled_load<4>(a.src[k], cb, y, x) : rows == 2 ? pooled_load<2>(a.src[k], cb, y, x)
                               : rows == 1 ? pooled_load<1>(a.src[k], cb, y, x) : pooled_load<0>(a.src[k], cb, y, x);
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
    const int tid = threadIdx.x;
    const int txn = ceil_div(a.ow, a.tx), tyn = ceil_div(a.oh, a.ty);
    int blk = blockIdx.x;
    const int bx = blk % txn; blk /= txn;
    const int by = blk % tyn;
    const int b = blk / tyn;
    // caller's flow tensor: the argument, the graph's io table, or - eemflow_forward_many - frame b's own buffer from the table's triples
    float* __restrict__ out = a.io ? (float*)a.io[a.io_frames ? 3 * b + 2 : 2] : a.out;
    const int bo = a.io_frames ? 0 : b;                      // the sample's index inside `out`
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
        float* dst = out + (((size_t)bo * 2 + c) * a.oh + oy) * a.ow + x0 + q * 4;
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

int tail_head_launch(const TailHeadArgs& a0, const int* taps_host, hipStream_t stream) {
    TailHeadArgs a = a0;
    EEM_REQUIRE(a.ntaps >= 1 && a.ntaps <= TAIL_HEAD_MAX_TAPS, "tail_head_launch: ntaps=%d", a.ntaps);
    for (int i = 0; i < a.ntaps; ++i) a.tap[i] = taps_host[i];
    const int g = a.gh * a.gw;
    const int nblk_rconv = ceil_div(g, 16) * a.batch;                     // per stage
    const int nblk_corr = ceil_div(ceil_div(g, 16) * a.batch, 9);         // per stage and tap: nine (sample, pixel tile) waves per block
    long pool_elems = 0;
    for (int k = 0; k < 3; ++k) pool_elems += (long)2 * a.batch * a.c[k] * g;
    const int nblk_pool = (int)((pool_elems + 575) / 576);
    // a box of grid_x x ntaps blocks per role: correlation uses (x, tap), the other roles count it row by row
    a.grid_x = nblk_corr;
    if (ceil_div(nblk_rconv, a.ntaps) > a.grid_x) a.grid_x = ceil_div(nblk_rconv, a.ntaps);
    if (ceil_div(nblk_pool, a.ntaps) > a.grid_x) a.grid_x = ceil_div(nblk_pool, a.ntaps);
    EEM_NOTE_GRID(a.grid_x * a.ntaps * 7, 576);
    hipLaunchKernelGGL(tail_head_kernel, dim3(a.grid_x, a.ntaps, 7), dim3(576), 0, stream, a);
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
    EEM_NOTE_GRID(blocks, 256);
    hipLaunchKernelGGL(tail_up_kernel, dim3(blocks), dim3(256), 0, stream, a);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}
