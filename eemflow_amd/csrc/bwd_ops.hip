// Backward passes of the non-convolution operators of the E-RAFT / EEMFlow+ parts of the path (SURVEY.md section 8b:
// corr_lookup_bwd, allpairs_corr_pyramid_bwd, convex_upsample_bwd, warp_bilinear_bwd).  Each is the exact adjoint of
// the forward kernel in eraft_kernels.hip / plus_kernels.hip (same coordinate arithmetic, same masks) and is checked
// against torch autograd through the oracle's restatement of the reference op (tests/test_gpu_bwd_ops.py).
//   reference forwards: model/corr.py:13-60, model/model_utils.py:7-21, model/eraft.py:83-94,
//                       model/EEMFlow/EEMFlow+.py:137-149, model/EEMFlow/cdc_utils.py:50-78, utils_luo/tools.py:2262-2306
// Scatter-type adjoints use fp32 atomics (HBM-side, ~1.3 TB/s; contention is low: every (pixel, level) owns its slice
// of the correlation pyramid); the two GEMMs of the all-pairs adjoint run on v_mfma_f32_32x32x2_f32.
#include "../../include/eemflow_hip.h"
#include "common.h"

namespace {

inline unsigned nblocks(long n) { return (unsigned)((n + 255) / 256); }

// ------------------------------------------------------------------------------------------------ corr lookup
struct LookupBwdArgs {
    float* dpyr[4];
    int ph[4], pw[4];
    const float* coords;   // [B][2][H][W]
    const float* dout;     // [B][324][H][W]
    int batch, h, w;
};

// adjoint of lookup_kernel: the 4 bilinear taps of (level, window position) scatter dout * weight
__global__ __launch_bounds__(256) void lookup_bwd_kernel(LookupBwdArgs a) {
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
    const float x = cx / sc + (float)(i - 4), y = cy / sc + (float)(jj - 4);
    const int h = a.ph[lvl], w = a.pw[lvl];
    const float xn = 2.f * x / (float)(w - 1) - 1.f, yn = 2.f * y / (float)(h - 1) - 1.f;
    const float ix = ((xn + 1.f) * 0.5f) * (float)(w - 1), iy = ((yn + 1.f) * 0.5f) * (float)(h - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    // the float -> int conversion must not overflow for wild coordinates
    const int x0 = (int)fminf(fmaxf(fx, -2.f), (float)w + 1.f), y0 = (int)fminf(fmaxf(fy, -2.f), (float)h + 1.f);
    const float tx = ix - fx, ty = iy - fy;
    const float g = a.dout[idx];
    float* img = a.dpyr[lvl] + ((size_t)b * hw + p) * h * w;
    auto add = [&](int yy, int xx, float wgt) {
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) atomicAdd(img + (size_t)yy * w + xx, g * wgt);
    };
    add(y0, x0, (1.f - tx) * (1.f - ty));
    add(y0, x0 + 1, tx * (1.f - ty));
    add(y0 + 1, x0, (1.f - tx) * ty);
    add(y0 + 1, x0 + 1, tx * ty);
}

// The same adjoint as a gather.  Every output pixel p has a correlation map of its own per level, and its 81 taps (one fractional
// offset, 9 x 9 integer offsets) touch a 10 x 10 window of that map: a thread owns one row of the window and adds, cell by cell, the
// <= 4 taps whose bilinear corners fall on the cell - plain read-modify-writes (no two threads share a cell), 10 consecutive floats
// per thread, and dout read as full-width runs over p (the scatter form issues 4 atomics per (tap, pixel), each on another map:
// 25 M atomics on distinct lines per call at batch 4, 0.95 ms).  A tap's corner and weight are evaluated with lookup_kernel's own
// arithmetic; only a corner that rounding moved by one cell beside an (almost) integer coordinate is not found - its weight is ~1e-7.
__global__ __launch_bounds__(256) void lookup_bwd_rows_kernel(LookupBwdArgs a) {
    const int hw = a.h * a.w;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)a.batch * 40 * hw) return;
    const int p = idx % hw;
    const int r = (idx / hw) % 10;
    const int lvl = (idx / ((long)hw * 10)) % 4;
    const int b = idx / ((long)hw * 40);
    const float cx = a.coords[((size_t)b * 2 + 0) * hw + p], cy = a.coords[((size_t)b * 2 + 1) * hw + p];
    const float sc = (float)(1 << lvl);
    const int h = a.ph[lvl], w = a.pw[lvl];
    if (h <= 0 || w <= 0) return;
    auto tap = [](float c, float scv, int off, int size, int& i0, float& t) {
        const float v = c / scv + (float)(off - 4);
        const float vn = 2.f * v / (float)(size - 1) - 1.f;
        const float iv = ((vn + 1.f) * 0.5f) * (float)(size - 1);
        const float f = floorf(iv);
        // clamped for the float -> int conversion only, 16 cells outside the map: nine taps span 9 cells, so a window whose first
        // corner is clamped has no cell inside the map, and every corner that can reach the map is exact
        i0 = (int)fminf(fmaxf(f, -16.f), (float)size + 16.f);
        t = iv - f;
    };
    // channel k = i * 9 + jj samples x + (i - 4), y + (jj - 4) (the reference's transposed window)
    int yb; float tyb;
    tap(cy, sc, 0, h, yb, tyb);
    const int Y = yb + r;
    if (Y < 0 || Y >= h) return;
    float wy[2]; int jy[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int jj = r - 1 + q;
        jy[q] = jj;
        wy[q] = 0.f;
        if (jj >= 0 && jj <= 8) {
            int y0; float ty;
            tap(cy, sc, jj, h, y0, ty);
            wy[q] = y0 == Y ? 1.f - ty : (y0 + 1 == Y ? ty : 0.f);
        }
    }
    int x0[9]; float tx[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) tap(cx, sc, i, w, x0[i], tx[i]);
    const float* dch = a.dout + ((size_t)b * 324 + lvl * 81) * hw + p;
    float* row = a.dpyr[lvl] + ((size_t)b * hw + p) * h * w + (size_t)Y * w;
    // dout of the two candidate rows for every x tap: d[q][i]
    // (every load unconditional, from a clamped index: a load inside a lane-dependent branch is followed by the compiler's
    // s_waitcnt vmcnt(0) - eighteen dout reads and ten read-modify-writes were 28 dependent round trips per thread)
    float d[2][9];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int jc = min(max(jy[q], 0), 8);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const float v = dch[(size_t)(i * 9 + jc) * hw];
            d[q][i] = (wy[q] != 0.f) ? v * wy[q] : 0.f;
        }
    }
    const int xb = x0[0];
    float old[10];
#pragma unroll
    for (int c = 0; c < 10; ++c) old[c] = row[min(max(xb + c, 0), w - 1)];
#pragma unroll
    for (int c = 0; c < 10; ++c) {
        const int X = xb + c;
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = c - 1 + q;
            if (i < 0 || i > 8) continue;
            const float wx = x0[i] == X ? 1.f - tx[i] : (x0[i] + 1 == X ? tx[i] : 0.f);
            sum += wx * (d[0][i] + d[1][i]);
        }
        if (X >= 0 && X < w) row[X] = old[c] + sum;
    }
}

// adjoint of pool2_kernel (avg_pool2d(2, 2), floor): fine[2y+dy][2x+dx] += 0.25 * coarse[y][x]
__global__ __launch_bounds__(256) void pool2_bwd_kernel(const float* __restrict__ dcoarse, float* __restrict__ dfine, long planes,
                                                        int h, int w) {
    const int oh = h / 2, ow = w / 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * oh * ow) return;
    const int x = idx % ow, y = (idx / ow) % oh;
    const long p = idx / ((long)ow * oh);
    const float g = 0.25f * dcoarse[idx];
    float* d = dfine + (p * h + 2 * y) * w + 2 * x;
    d[0] += g; d[1] += g; d[w] += g; d[w + 1] += g;
}

// ------------------------------------------------------------------------------------------------ all-pairs GEMMs
// D[c][p] = scale * sum_k A[c][k] * Bop[k][p],  A = fmap [C][HW] (row per c),  k over pixels.
//   BT = true : Bop[k][p] = dcorr[p][k]   (d fmap1: k = p2 runs along a dcorr row -> every lane streams its own row,
//                                          16 bytes per load with the k order  k = 8*(s/4) + 4*slot + s%4)
//   BT = false: Bop[k][p] = dcorr[k][p]   (d fmap2: k = p1 is the row index, lanes p are unit-stride)
// One wave per 64 c x 32 p tile (2 MFMA tiles sharing the B operand).
template <bool BT>
__global__ __launch_bounds__(256) void allpairs_bwd_kernel(const float* __restrict__ fmap, const float* __restrict__ dcorr,
                                                           float* __restrict__ dout, int c, int hw, float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int b = blockIdx.z;
    const int p0 = (blockIdx.x * 4 + wave) * 32;
    const int c0 = blockIdx.y * 64;
    if (p0 >= hw) return;
    const float* A = fmap + (size_t)b * c * hw;
    const float* G = dcorr + (size_t)b * hw * hw;
    const int p = min(p0 + j, hw - 1);
    const int ca = min(c0 + j, c - 1), cb = min(c0 + 32 + j, c - 1);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int k8 = hw & ~7;
    for (int k0 = 0; k0 < k8; k0 += 8) {
        const int kk = k0 + 4 * h;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(A + (size_t)ca * hw + kk);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(A + (size_t)cb * hw + kk);
        f32x4 bv;
        if (BT) {
            bv = *reinterpret_cast<const f32x4*>(G + (size_t)p * hw + kk);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[q] = G[(size_t)(kk + q) * hw + p];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], bv[q], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], bv[q], acc1, 0, 0, 0);
        }
    }
    for (int k = k8 + h; k < hw; k += 2) {                                   // tail: hw % 8 == 4 (hw % 4 == 0 is required)
        const float a0 = A[(size_t)ca * hw + k], a1 = A[(size_t)cb * hw + k];
        const float bv = BT ? G[(size_t)p * hw + k] : G[(size_t)k * hw + p];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc1, 0, 0, 0);
    }
    if (p0 + j >= hw) return;
    float* o = dout + (size_t)b * c * hw + p0 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (c0 + row < c) o[(size_t)(c0 + row) * hw] = acc0[r] * scale;
        if (c0 + 32 + row < c) o[(size_t)(c0 + 32 + row) * hw] = acc1[r] * scale;
    }
}

// ------------------------------------------------------------------------------------------------ convex upsample
// adjoint of convex_up_kernel.  A thread owns one coarse pixel and one sub-row sy and walks its 8 sub-columns: lanes run along the
// coarse x, so the 9 mask logits of a sub-position (channel k*64 + sy*8 + sx) are read, and their gradients written, as contiguous
// runs; the flow gradient of the 9 neighbours is summed over the 8 sub-columns in registers and leaves as 18 atomics per thread
// (one thread per FINE pixel meant 18 atomics per fine pixel - 22 M per call at batch 4, 1.7 ms - and 8 channel planes per 8 lanes).
__global__ __launch_bounds__(256) void convex_up_bwd_kernel(const float* __restrict__ flow, const float* __restrict__ mask,
                                                            const float* __restrict__ dout, float* __restrict__ dflow,
                                                            float* __restrict__ dmask, int batch, int h, int w) {
    const int oh = 8 * h, ow = 8 * w;
    const int hw = h * w;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)batch * 8 * hw) return;
    const int x = idx % w, y = (idx / w) % h;
    const int sy = (idx / hw) % 8, b = idx / ((long)8 * hw);
    float fu[9], fv[9];                                              // 8 * flow of the 3x3 neighbours (0 outside: F.unfold pads)
    bool in[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        in[k] = yy >= 0 && yy < h && xx >= 0 && xx < w;
        const size_t q = (size_t)b * 2 * hw + (in[k] ? yy * w + xx : 0);
        const float f0 = flow[q], f1 = flow[q + hw];                  // (q is clamped: loads unconditional, the bounds applied to the values)
        fu[k] = in[k] ? 8.f * f0 : 0.f;
        fv[k] = in[k] ? 8.f * f1 : 0.f;
    }
    float gu[9], gv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) gu[k] = gv[k] = 0.f;
    const int Y = 8 * y + sy;
#pragma unroll 2
    for (int sx = 0; sx < 8; ++sx) {
        const size_t mo = ((size_t)b * 576 + sy * 8 + sx) * hw + y * w + x;
        float lg[9], mx = -3.4e38f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { lg[k] = mask[mo + (size_t)k * 64 * hw]; mx = fmaxf(mx, lg[k]); }
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { lg[k] = expf(lg[k] - mx); den += lg[k]; }
        const size_t fo = (size_t)b * 2 * oh * ow + (size_t)Y * ow + 8 * x + sx;
        const float du = dout[fo], dv = dout[fo + (size_t)oh * ow];
        float dw[9], dot = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float wk = lg[k] / den;
            lg[k] = wk;
            dw[k] = du * fu[k] + dv * fv[k];
            gu[k] += wk * du;
            gv[k] += wk * dv;
            dot += wk * dw[k];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) dmask[mo + (size_t)k * 64 * hw] = lg[k] * (dw[k] - dot);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k)
        if (in[k]) {
            const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
            const size_t q = (size_t)b * 2 * hw + yy * w + xx;
            atomicAdd(dflow + q, 8.f * gu[k]);
            atomicAdd(dflow + q + hw, 8.f * gv[k]);
        }
}

// ------------------------------------------------------------------------------------------------ warp
// adjoint of warp_kernel (plus_kernels.hip): d x by scattering the 4 taps, d flow through the tap weights
// (grid_sample backward w.r.t. the grid times d grid / d flow = 2/(size-1) * (size-1)/2 or size/2).  mode 2: the
// `>= 1` mask is a constant of the backward pass (no gradient), as in the reference.
__global__ __launch_bounds__(256) void warp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ flow,
                                                       const float* __restrict__ dout, float* __restrict__ dx,
                                                       float* __restrict__ dflow, int batch, int c, int h, int w, int mode) {
    const int hw = h * w;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)batch * hw) return;
    const int p = idx % hw, b = idx / hw;
    const int py = p / w, px = p - py * w;
    const float vx = (float)px + flow[((size_t)b * 2 + 0) * hw + p];
    const float vy = (float)py + flow[((size_t)b * 2 + 1) * hw + p];
    const float xn = 2.0f * vx / (float)max(w - 1, 1) - 1.0f;
    const float yn = 2.0f * vy / (float)max(h - 1, 1) - 1.0f;
    float ix, iy, gx_mult, gy_mult;                          // d ix / d vx, d iy / d vy
    if (mode == 0) {
        ix = (xn + 1.f) * ((float)(w - 1) / 2.f);
        iy = (yn + 1.f) * ((float)(h - 1) / 2.f);
        gx_mult = (2.0f / (float)max(w - 1, 1)) * ((float)(w - 1) / 2.f);
        gy_mult = (2.0f / (float)max(h - 1, 1)) * ((float)(h - 1) / 2.f);
    } else {
        ix = (xn + 1.f) * ((float)w / 2.f) - 0.5f;
        iy = (yn + 1.f) * ((float)h / 2.f) - 0.5f;
        gx_mult = (2.0f / (float)max(w - 1, 1)) * ((float)w / 2.f);
        gy_mult = (2.0f / (float)max(h - 1, 1)) * ((float)h / 2.f);
    }
    const float xw = floorf(ix), yn0 = floorf(iy);
    const float wgt_w = ix - xw, wgt_e = 1.f - wgt_w, wgt_n = iy - yn0, wgt_s = 1.f - wgt_n;
    const float nw = wgt_s * wgt_e, ne = wgt_s * wgt_w, sw = wgt_n * wgt_e, se = wgt_n * wgt_w;
    const float cx = fminf(fmaxf(xw, -2.f), (float)w + 1.f), cy = fminf(fmaxf(yn0, -2.f), (float)h + 1.f);
    const int x0 = (int)cx, y0 = (int)cy;
    const bool in_w = x0 >= 0 && x0 < w, in_e = x0 + 1 >= 0 && x0 + 1 < w;
    const bool in_n = y0 >= 0 && y0 < h, in_s = y0 + 1 >= 0 && y0 + 1 < h;
    float m = 1.f;
    if (mode == 2) {
        const float ones = (((in_n && in_w ? 1.f : 0.f) * nw + (in_n && in_e ? 1.f : 0.f) * ne) + (in_s && in_w ? 1.f : 0.f) * sw) +
                           (in_s && in_e ? 1.f : 0.f) * se;
        m = ones >= 1.0f ? 1.f : 0.f;
    }
    float gix = 0.f, giy = 0.f;
    for (int ch = 0; ch < c; ++ch) {
        const size_t plane = ((size_t)b * c + ch) * hw;
        const float g = dout[plane + p] * m;
        const float* s = x + plane;
        float* d = dx + plane;
        // (unconditional loads from clamped cells, the bounds applied to the values)
        const int ya = min(max(y0, 0), h - 1), yb = min(max(y0 + 1, 0), h - 1), xa = min(max(x0, 0), w - 1), xb = min(max(x0 + 1, 0), w - 1);
        const float r_nw = s[ya * w + xa], r_ne = s[ya * w + xb], r_sw = s[yb * w + xa], r_se = s[yb * w + xb];
        const float v_nw = (in_n && in_w) ? r_nw : 0.f;
        const float v_ne = (in_n && in_e) ? r_ne : 0.f;
        const float v_sw = (in_s && in_w) ? r_sw : 0.f;
        const float v_se = (in_s && in_e) ? r_se : 0.f;
        if (in_n && in_w) atomicAdd(d + y0 * w + x0, g * nw);
        if (in_n && in_e) atomicAdd(d + y0 * w + x0 + 1, g * ne);
        if (in_s && in_w) atomicAdd(d + (y0 + 1) * w + x0, g * sw);
        if (in_s && in_e) atomicAdd(d + (y0 + 1) * w + x0 + 1, g * se);
        // ATen grid_sampler_2d_backward: d/d ix and d/d iy of the bilinear blend
        gix += g * ((v_ne - v_nw) * wgt_s + (v_se - v_sw) * wgt_n);
        giy += g * ((v_sw - v_nw) * wgt_e + (v_se - v_ne) * wgt_w);
    }
    dflow[((size_t)b * 2 + 0) * hw + p] = gix * gx_mult;
    dflow[((size_t)b * 2 + 1) * hw + p] = giy * gy_mult;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" int eraft_corr_lookup_bwd(const float* coords, const float* dout, int batch, int h, int w, float* dpyr0, float* dpyr1,
                                     float* dpyr2, float* dpyr3, void* stream) {
    EEM_REQUIRE(coords && dout && dpyr0 && dpyr1 && dpyr2 && dpyr3 && batch >= 1 && h >= 8 && w >= 8, "eraft_corr_lookup_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    LookupBwdArgs a;
    float* d[4] = {dpyr0, dpyr1, dpyr2, dpyr3};
    int ph = h, pw = w;
    for (int l = 0; l < 4; ++l) {
        a.dpyr[l] = d[l]; a.ph[l] = ph; a.pw[l] = pw;
        EEM_HIP_CHECK(hipMemsetAsync(d[l], 0, (size_t)batch * h * w * ph * pw * sizeof(float), st));
        ph /= 2; pw /= 2;
    }
    a.coords = coords; a.dout = dout; a.batch = batch; a.h = h; a.w = w;
    const char* sc = getenv("EEM_LOOKUP_BWD_SCATTER");            // read per call: the tests run both forms
    if (sc && sc[0] == '1') hipLaunchKernelGGL(lookup_bwd_kernel, dim3(nblocks((long)batch * 324 * h * w)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(lookup_bwd_rows_kernel, dim3(nblocks((long)batch * 40 * h * w)), dim3(256), 0, st, a);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

extern "C" int eraft_corr_pyramid_bwd(const float* fmap1, const float* fmap2, float* dpyr0, float* dpyr1, float* dpyr2,
                                      const float* dpyr3, int batch, int c, int h, int w, float* dfmap1, float* dfmap2,
                                      void* stream) {
    EEM_REQUIRE(fmap1 && fmap2 && dpyr0 && dpyr1 && dpyr2 && dpyr3 && dfmap1 && dfmap2, "eraft_corr_pyramid_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int hw = h * w;
    const long planes = (long)batch * hw;
    // avg_pool2d chain (model/corr.py:24-27): level 3 -> 2 -> 1 -> 0, accumulated in place
    const int hs[4] = {h, h / 2, h / 4, h / 8}, ws[4] = {w, w / 2, w / 4, w / 8};
    const float* coarse[3] = {dpyr3, dpyr2, dpyr1};
    float* fine[3] = {dpyr2, dpyr1, dpyr0};
    for (int s = 0; s < 3; ++s) {
        const int l = 2 - s;                                  // fine level
        const long n = planes * (hs[l] / 2) * (ws[l] / 2);
        if (n > 0) hipLaunchKernelGGL(pool2_bwd_kernel, dim3(nblocks(n)), dim3(256), 0, st, coarse[s], fine[s], planes, hs[l], ws[l]);
    }
    EEM_HIP_CHECK(hipGetLastError());
    const float scale = 1.0f / sqrtf((float)c);
    dim3 grid(ceil_div(hw, 128), ceil_div(c, 64), batch);
    // d fmap1[c][p1] = scale * sum_p2 dcorr[p1][p2] fmap2[c][p2];   d fmap2[c][p2] = scale * sum_p1 dcorr[p1][p2] fmap1[c][p1]
    if ((hw & 3) == 0) {
        hipLaunchKernelGGL((allpairs_bwd_kernel<true>), grid, dim3(256), 0, st, fmap2, dpyr0, dfmap1, c, hw, scale);
        hipLaunchKernelGGL((allpairs_bwd_kernel<false>), grid, dim3(256), 0, st, fmap1, dpyr0, dfmap2, c, hw, scale);
    } else {
        eem_set_error("eraft_corr_pyramid_bwd: h*w must be a multiple of 4 (16-byte rows)");
        return EEM_ERR_ARG;
    }
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

extern "C" int eraft_convex_upsample_bwd(const float* flow, const float* mask, const float* dout, int batch, int h, int w,
                                         float* dflow, float* dmask, void* stream) {
    EEM_REQUIRE(flow && mask && dout && dflow && dmask && batch >= 1 && h >= 1 && w >= 1, "eraft_convex_upsample_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    EEM_HIP_CHECK(hipMemsetAsync(dflow, 0, (size_t)batch * 2 * h * w * sizeof(float), st));
    hipLaunchKernelGGL(convex_up_bwd_kernel, dim3(nblocks((long)batch * 8 * h * w)), dim3(256), 0, st, flow, mask, dout, dflow, dmask,
                       batch, h, w);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

extern "C" int eemplus_warp_bwd(const float* x, const float* flow, const float* dout, int batch, int c, int h, int w, int mode,
                                float* dx, float* dflow, void* stream) {
    EEM_REQUIRE(x && flow && dout && dx && dflow && batch >= 1 && c >= 1 && h >= 1 && w >= 1 && mode >= 0 && mode <= 2,
                "eemplus_warp_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    EEM_HIP_CHECK(hipMemsetAsync(dx, 0, (size_t)batch * c * h * w * sizeof(float), st));
    hipLaunchKernelGGL(warp_bwd_kernel, dim3(nblocks((long)batch * h * w)), dim3(256), 0, st, x, flow, dout, dx, dflow, batch, c, h, w, mode);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}
