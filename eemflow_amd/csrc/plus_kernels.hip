// Warp family and flow resampling of EEMFlow+ (reference: model/EEMFlow/EEMFlow+.py:137-149,
// model/EEMFlow/cdc_utils.py:50-103,156-174, utils_luo/tools.py:2262-2306).
// Built with -ffp-contract=off: the reference's `grid_sample(ones) >= 1.0` mask depends on the last bit of
// nw + ne + sw + se, so the coordinate and weight arithmetic follows ATen's CPU grid sampler operation by
// operation (separate multiplies and adds, same association).
#include "plus_kernels.h"

namespace {

inline unsigned nblocks(long n) { return (unsigned)((n + 255) / 256); }

// mode 0: align_corners=True (EEMFlow_cdc.warp); 1: align_corners=False (torch_warp);
// 2: align_corners=False + `grid_sample(ones) >= 1` mask (WarpingLayer_no_div)
constexpr int kWarpCh = 4;
// one pixel of a warp: the flow (fx, fy) of pixel p, the channels [ch0, ch0 + kWarpCh) of x
__device__ __forceinline__ void warp_px(const float* __restrict__ x, float fx, float fy, float* __restrict__ out, int out_ctotal, int out_coff,
                                        int b, int c, int h, int w, int p, int mode, int ch0) {
    const int hw = h * w;
    const int py = p / w, px = p - py * w;
    const float vx = (float)px + fx;
    const float vy = (float)py + fy;
    const float xn = 2.0f * vx / (float)max(w - 1, 1) - 1.0f;
    const float yn = 2.0f * vy / (float)max(h - 1, 1) - 1.0f;
    float ix, iy;
    if (mode == 0) {
        ix = (xn + 1.f) * ((float)(w - 1) / 2.f);
        iy = (yn + 1.f) * ((float)(h - 1) / 2.f);
    } else {
        ix = (xn + 1.f) * ((float)w / 2.f) - 0.5f;
        iy = (yn + 1.f) * ((float)h / 2.f) - 0.5f;
    }
    const float xw = floorf(ix), yn0 = floorf(iy);
    const float wgt_w = ix - xw, wgt_e = 1.f - wgt_w, wgt_n = iy - yn0, wgt_s = 1.f - wgt_n;
    const float nw = wgt_s * wgt_e, ne = wgt_s * wgt_w, sw = wgt_n * wgt_e, se = wgt_n * wgt_w;
    // the float -> int conversion must not overflow for wild flows
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
    const int o_nw = (in_n && in_w) ? y0 * w + x0 : -1, o_ne = (in_n && in_e) ? y0 * w + x0 + 1 : -1;
    const int o_sw = (in_s && in_w) ? (y0 + 1) * w + x0 : -1, o_se = (in_s && in_e) ? (y0 + 1) * w + x0 + 1 : -1;
    float v[kWarpCh][4];
#pragma unroll
    for (int i = 0; i < kWarpCh; ++i) {
        const float* s = x + ((size_t)b * c + min(ch0 + i, c - 1)) * hw;
        // (unconditional loads from clamped offsets, the bounds applied to the values: a load in one arm of a lane-dependent
        // conditional is a branch followed by s_waitcnt vmcnt(0) - sixteen dependent round trips instead of one)
        const float a0 = s[max(o_nw, 0)], a1 = s[max(o_ne, 0)], a2 = s[max(o_sw, 0)], a3 = s[max(o_se, 0)];
        v[i][0] = o_nw >= 0 ? a0 : 0.f;
        v[i][1] = o_ne >= 0 ? a1 : 0.f;
        v[i][2] = o_sw >= 0 ? a2 : 0.f;
        v[i][3] = o_se >= 0 ? a3 : 0.f;
    }
#pragma unroll
    for (int i = 0; i < kWarpCh; ++i) {
        if (ch0 + i >= c) break;
        const float r = ((v[i][0] * nw + v[i][1] * ne) + v[i][2] * sw) + v[i][3] * se;
        out[((size_t)b * out_ctotal + out_coff + ch0 + i) * hw + p] = r * m;
    }
}

// blockIdx.y = group of kWarpCh channels: the coarse levels are a handful of pixel blocks, and a thread that walks all 32 / 64
// channels pays their gather latencies one after the other (23 x 40 x 64 channels: 22 us); all loads of a group go out together
__global__ __launch_bounds__(256) void warp_kernel(const float* __restrict__ x, const float* __restrict__ flow, int flow_ctotal,
                                                   float* __restrict__ out, int out_ctotal, int out_coff, int batch, int c, int h, int w,
                                                   int mode) {
    const int hw = h * w;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)batch * hw) return;
    const int p = idx % hw, b = idx / hw;
    warp_px(x, flow[((size_t)b * flow_ctotal + 0) * hw + p], flow[((size_t)b * flow_ctotal + 1) * hw + p], out, out_ctotal, out_coff, b, c, h, w,
            p, mode, blockIdx.y * kWarpCh);
}

// bilinear value of a flow plane s [h][w] at output (Y, X) of an [oh][ow] grid, align_corners=True - upflow_kernel's expression
__device__ __forceinline__ float upflow_px(const float* __restrict__ s, int h, int w, int oh, int ow, int Y, int X) {
    const float sy = oh > 1 ? (float)(h - 1) / (float)(oh - 1) : 0.f, sx = ow > 1 ? (float)(w - 1) / (float)(ow - 1) : 0.f;
    const float fy = sy * (float)Y, fx = sx * (float)X;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    return (1.f - ly) * ((1.f - lx) * s[y0 * w + x0] + lx * s[y0 * w + x1]) + ly * ((1.f - lx) * s[y1 * w + x0] + lx * s[y1 * w + x1]);
}

// cdc_model's first three steps as ONE launch (round 6; cdc_utils.py:156-162): flow_init = upsample2d_flow_as(flow_coarse, if_rate=True)
// (upflow_kernel's arithmetic, to the bit), the in-place doubling of the coarse flow that call leaves behind (:85-86; written to a SECOND
// buffer - other threads of this launch still interpolate from the unscaled one - which the host then takes for the coarse flow), and
// WarpingLayer_no_div(x, flow_init) (warp_kernel's mode 2) from the flow value the thread has just formed.  Three launches of ~4.8 us
// each at the coarse levels before.
__global__ __launch_bounds__(256) void upflow_warp_kernel(const float* __restrict__ fc, float* __restrict__ fc_scaled, int hc, int wc,
                                                          float* __restrict__ fi, const float* __restrict__ x, float* __restrict__ out,
                                                          int out_ctotal, int out_coff, int batch, int c, int h, int w) {
    const int hw = h * w;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.y == 0 && idx < (long)batch * 2 * hc * wc)
        fc_scaled[idx] = fc[idx] * (((idx / (hc * wc)) & 1) ? ((float)h / (float)hc) : ((float)w / (float)wc));
    if (idx >= (long)batch * hw) return;
    const int p = idx % hw, b = idx / hw;
    const int py = p / w, px = p - py * w;
    float f[2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        float v = upflow_px(fc + ((size_t)b * 2 + ch) * hc * wc, hc, wc, h, w, py, px);
        v *= ch ? ((float)h / (float)hc) : ((float)w / (float)wc);
        f[ch] = v;
        if (blockIdx.y == 0) fi[((size_t)b * 2 + ch) * hw + p] = v;
    }
    warp_px(x, f[0], f[1], out, out_ctotal, out_coff, b, c, h, w, p, 2, blockIdx.y * kWarpCh);
}

// F.interpolate(bilinear, align_corners=True) of a flow [b][2][h][w] -> [b][2][oh][ow], optionally scaled by
// (ow/w, oh/h) per channel (upsample2d_flow_as, if_rate=True; cdc_utils.py:80-103)
__global__ __launch_bounds__(256) void upflow_kernel(const float* __restrict__ in, float* __restrict__ out, int batch, int h, int w,
                                                     int oh, int ow, int rate) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)batch * 2 * oh * ow) return;
    const int X = idx % ow, Y = (idx / ow) % oh;
    const int bc = idx / ((long)ow * oh);
    float v = upflow_px(in + (size_t)bc * h * w, h, w, oh, ow, Y, X);
    if (rate) v *= (bc & 1) ? ((float)oh / (float)h) : ((float)ow / (float)w);
    out[idx] = v;
}

// The five full-resolution predictions of a forward (EEMFlow+.py:231-232) as ONE launch: job = blockIdx.y, a thread = four
// neighbouring outputs of a row and one 16-byte store (the arithmetic per output is upflow_kernel's, to the bit)
struct UpflowJobs { const float* in[5]; float* out[5]; int h[5], w[5]; };
__global__ __launch_bounds__(256) void upflow_multi_kernel(UpflowJobs J, int batch, int oh, int ow, int rate) {
    const int job = blockIdx.y;
    const int h = J.h[job], w = J.w[job];
    const float* __restrict__ in = J.in[job];
    float* __restrict__ out = J.out[job];
    const int owq = ow >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)batch * 2 * oh * owq) return;
    const int Xq = idx % owq, Y = (idx / owq) % oh;
    const int bc = idx / ((long)owq * oh);
    const float sy = oh > 1 ? (float)(h - 1) / (float)(oh - 1) : 0.f, sx = ow > 1 ? (float)(w - 1) / (float)(ow - 1) : 0.f;
    const float fy = sy * (float)Y;
    const int y0 = (int)fy;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
    const float ly = fy - (float)y0;
    const float* s = in + (size_t)bc * h * w;
    const float* r0 = s + y0 * w;
    const float* r1 = s + y1 * w;
    const float mul = rate ? ((bc & 1) ? ((float)oh / (float)h) : ((float)ow / (float)w)) : 1.f;
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int X = Xq * 4 + i;
        const float fx = sx * (float)X;
        const int x0 = (int)fx;
        const int x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float lx = fx - (float)x0;
        float v = (1.f - ly) * ((1.f - lx) * r0[x0] + lx * r0[x1]) + ly * ((1.f - lx) * r1[x0] + lx * r1[x1]);
        if (rate) v *= mul;
        o[i] = v;
    }
    *reinterpret_cast<f32x4*>(out + ((size_t)bc * oh + Y) * ow + Xq * 4) = o;
}

// the in-place side effect of upsample2d_flow_as(if_rate=True): inputs[:,0] *= ow/w; inputs[:,1] *= oh/h
__global__ __launch_bounds__(256) void scale_flow_kernel(float* __restrict__ f, int batch, int hw, float su, float sv) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)batch * 2 * hw) return;
    f[idx] *= ((idx / hw) & 1) ? sv : su;
}

// flow_up = torch_warp(flow_init, inter_flow) * (1 - sigmoid(m)) + flow_init * sigmoid(m)   (cdc_utils.py:163-173)
__global__ __launch_bounds__(256) void blend_kernel(const float* __restrict__ warped, const float* __restrict__ flow_init,
                                                    const float* __restrict__ xout, float* __restrict__ out, int batch, int hw) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)batch * 2 * hw) return;
    const int p = idx % hw, b = idx / (2L * hw);
    const float mk = 1.f / (1.f + expf(-xout[((size_t)b * 3 + 2) * hw + p]));
    out[idx] = warped[idx] * (1.f - mk) + flow_init[idx] * mk;
}

// cdc_model's last three steps as one launch (cdc_utils.py:163-173 and the copy of flow_up into the decoder's input):
//   warped = torch_warp(flow_init, xout[:, 0:2])  (warp_kernel's mode 1, operation by operation)
//   flow_up = warped * (1 - sigmoid(xout[:, 2])) + flow_init * sigmoid(xout[:, 2])  (blend_kernel's expression)
//   flow_up -> `out` [b][2][hw] and -> channels [cat_coff, cat_coff + 2) of `cat` [b][cat_ctotal][hw]
// With f2 != NULL (round 6) the launch also does the step behind it, EEMFlow_cdc.warp(feature_2, flow_up) (EEMFlow+.py:189; warp_kernel's
// mode 0) from the flow_up value the thread has just formed: blockIdx.y = channel group of f2, group 0 writes flow_up.
__global__ __launch_bounds__(256) void warp_blend_kernel(const float* __restrict__ fi, const float* __restrict__ xout, float* __restrict__ out,
                                                         float* __restrict__ cat, int cat_ctotal, int cat_coff, int batch, int h, int w,
                                                         const float* __restrict__ f2, float* __restrict__ fw, int c2) {
    const int hw = h * w;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)batch * hw) return;
    const int p = idx % hw, b = idx / hw;
    const int py = p / w, px = p - py * w;
    const float vx = (float)px + xout[((size_t)b * 3 + 0) * hw + p];
    const float vy = (float)py + xout[((size_t)b * 3 + 1) * hw + p];
    const float xn = 2.0f * vx / (float)max(w - 1, 1) - 1.0f;
    const float yn = 2.0f * vy / (float)max(h - 1, 1) - 1.0f;
    const float ix = (xn + 1.f) * ((float)w / 2.f) - 0.5f;
    const float iy = (yn + 1.f) * ((float)h / 2.f) - 0.5f;
    const float xw = floorf(ix), yn0 = floorf(iy);
    const float wgt_w = ix - xw, wgt_e = 1.f - wgt_w, wgt_n = iy - yn0, wgt_s = 1.f - wgt_n;
    const float nw = wgt_s * wgt_e, ne = wgt_s * wgt_w, sw = wgt_n * wgt_e, se = wgt_n * wgt_w;
    const float cx = fminf(fmaxf(xw, -2.f), (float)w + 1.f), cy = fminf(fmaxf(yn0, -2.f), (float)h + 1.f);
    const int x0 = (int)cx, y0 = (int)cy;
    const bool in_w = x0 >= 0 && x0 < w, in_e = x0 + 1 >= 0 && x0 + 1 < w;
    const bool in_n = y0 >= 0 && y0 < h, in_s = y0 + 1 >= 0 && y0 + 1 < h;
    const int ya = min(max(y0, 0), h - 1), yb = min(max(y0 + 1, 0), h - 1), xa = min(max(x0, 0), w - 1), xb = min(max(x0 + 1, 0), w - 1);
    const float mk = 1.f / (1.f + expf(-xout[((size_t)b * 3 + 2) * hw + p]));
    float r[2], f0[2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        const float* s = fi + ((size_t)b * 2 + ch) * hw;
        const float a0 = s[ya * w + xa], a1 = s[ya * w + xb], a2 = s[yb * w + xa], a3 = s[yb * w + xb];
        f0[ch] = s[p];
        const float v0 = (in_n && in_w) ? a0 : 0.f, v1 = (in_n && in_e) ? a1 : 0.f, v2 = (in_s && in_w) ? a2 : 0.f, v3 = (in_s && in_e) ? a3 : 0.f;
        r[ch] = ((v0 * nw + v1 * ne) + v2 * sw) + v3 * se;
    }
    float fu[2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        const float v = r[ch] * (1.f - mk) + f0[ch] * mk;
        fu[ch] = v;
        if (blockIdx.y == 0) {
            out[((size_t)b * 2 + ch) * hw + p] = v;
            cat[((size_t)b * cat_ctotal + cat_coff + ch) * hw + p] = v;
        }
    }
    if (f2) warp_px(f2, fu[0], fu[1], fw, c2, 0, b, c2, h, w, p, 0, blockIdx.y * kWarpCh);
}

__global__ __launch_bounds__(256) void copy_channels_kernel(const float* __restrict__ src, int s_ctotal, int s_coff, float* __restrict__ dst,
                                                            int d_ctotal, int d_coff, int c, int batch, int hw) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)batch * c * hw) return;
    const int p = idx % hw, ch = (idx / hw) % c, b = idx / ((long)hw * c);
    dst[((size_t)b * d_ctotal + d_coff + ch) * hw + p] = src ? src[((size_t)b * s_ctotal + s_coff + ch) * hw + p] : 0.f;
}

}  // namespace

int pl_warp_launch(const float* x, const float* flow, int flow_ctotal, float* out, int out_ctotal, int out_coff, int batch, int c, int h,
                   int w, int mode, hipStream_t st) {
    hipLaunchKernelGGL(warp_kernel, dim3(nblocks((long)batch * h * w), (c + kWarpCh - 1) / kWarpCh), dim3(256), 0, st, x, flow, flow_ctotal, out, out_ctotal, out_coff,
                       batch, c, h, w, mode);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int pl_upflow_launch(const float* in, float* out, int batch, int h, int w, int oh, int ow, int rate, hipStream_t st) {
    hipLaunchKernelGGL(upflow_kernel, dim3(nblocks((long)batch * 2 * oh * ow)), dim3(256), 0, st, in, out, batch, h, w, oh, ow, rate);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int pl_upflow_multi_launch(const float* const* in, float* const* out, const int* h, const int* w, int njobs, int batch, int oh, int ow,
                           int rate, hipStream_t st) {
    bool vec = njobs >= 1 && njobs <= 5 && (ow & 3) == 0;
    for (int i = 0; i < njobs && vec; ++i) vec = ((uintptr_t)out[i] & 15) == 0;
    if (!vec) {
        for (int i = 0; i < njobs; ++i) {
            const int rc = pl_upflow_launch(in[i], out[i], batch, h[i], w[i], oh, ow, rate, st);
            if (rc != EEM_OK) return rc;
        }
        return EEM_OK;
    }
    UpflowJobs J;
    for (int i = 0; i < 5; ++i) { const int k = i < njobs ? i : 0; J.in[i] = in[k]; J.out[i] = out[k]; J.h[i] = h[k]; J.w[i] = w[k]; }
    hipLaunchKernelGGL(upflow_multi_kernel, dim3(nblocks((long)batch * 2 * oh * (ow >> 2)), njobs), dim3(256), 0, st, J, batch, oh, ow, rate);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int pl_scale_flow_launch(float* f, int batch, int hw, float su, float sv, hipStream_t st) {
    hipLaunchKernelGGL(scale_flow_kernel, dim3(nblocks((long)batch * 2 * hw)), dim3(256), 0, st, f, batch, hw, su, sv);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int pl_blend_launch(const float* warped, const float* flow_init, const float* xout, float* out, int batch, int hw, hipStream_t st) {
    hipLaunchKernelGGL(blend_kernel, dim3(nblocks((long)batch * 2 * hw)), dim3(256), 0, st, warped, flow_init, xout, out, batch, hw);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int pl_warp_blend_launch(const float* flow_init, const float* xout, float* out, float* cat, int cat_ctotal, int cat_coff, int batch, int h, int w,
                         hipStream_t st) {
    hipLaunchKernelGGL(warp_blend_kernel, dim3(nblocks((long)batch * h * w)), dim3(256), 0, st, flow_init, xout, out, cat, cat_ctotal, cat_coff,
                       batch, h, w, (const float*)nullptr, (float*)nullptr, 0);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int pl_warp_blend_warp_launch(const float* flow_init, const float* xout, float* out, float* cat, int cat_ctotal, int cat_coff, const float* f2,
                              float* fw, int c2, int batch, int h, int w, hipStream_t st) {
    hipLaunchKernelGGL(warp_blend_kernel, dim3(nblocks((long)batch * h * w), (c2 + kWarpCh - 1) / kWarpCh), dim3(256), 0, st, flow_init, xout, out,
                       cat, cat_ctotal, cat_coff, batch, h, w, f2, fw, c2);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int pl_upflow_warp_launch(const float* fc, float* fc_scaled, int hc, int wc, float* fi, const float* x, float* out, int out_ctotal, int out_coff,
                          int batch, int c, int h, int w, hipStream_t st) {
    hipLaunchKernelGGL(upflow_warp_kernel, dim3(nblocks((long)batch * h * w), (c + kWarpCh - 1) / kWarpCh), dim3(256), 0, st, fc, fc_scaled, hc, wc,
                       fi, x, out, out_ctotal, out_coff, batch, c, h, w);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int pl_copy_channels_launch(const float* src, int s_ctotal, int s_coff, float* dst, int d_ctotal, int d_coff, int c, int batch, int hw,
                            hipStream_t st) {
    hipLaunchKernelGGL(copy_channels_kernel, dim3(nblocks((long)batch * c * hw)), dim3(256), 0, st, src, s_ctotal, s_coff, dst, d_ctotal,
                       d_coff, c, batch, hw);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}
