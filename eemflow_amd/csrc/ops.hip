// Operator-level C ABI (eemop_*): the differentiable building blocks of the E-RAFT training graph (model/eraft.py,
// model/update.py, model/extractor.py, model/corr.py under autograd; SURVEY 8a rows A9 train-mode BatchNorm, A12 update-block
// backward).  Each op has a forward and a backward entry point on caller-owned NCHW fp32 tensors; eemflow_amd/ops.py wraps
// them as torch.autograd.Functions, so the recurrence (12 update iterations through one set of weights), the detach of the
// coordinates and the accumulation of weight gradients are autograd bookkeeping while all arithmetic runs here.
//
// Convolutions run on the generic MFMA conv (gconv.hip): forward with the weights as they are, data gradient with the
// transposed + flipped filter (transposed stride for the stride-2 layers), weight gradient on the generalised MFMA
// weight-gradient kernel of train.hip.  The packed weight layouts are pure gathers of the caller's OIHW tensor: the gather
// table of a layer shape is built once (host, cached) and applied on the device in front of every launch.
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/eemflow_hip.h"
#include "eraft_kernels.h"
#include "gconv.h"
#include "train.h"

namespace {

// ------------------------------------------------------------------------------------------------ elementwise kernels
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                      float* __restrict__ out, long n, int kind, float scale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = y[i];
    float d;
    if (kind == GACT_RELU) d = v > 0.f ? 1.f : 0.f;
    else if (kind == GACT_SIGMOID) d = v * (1.f - v);
    else if (kind == GACT_TANH) d = 1.f - v * v;
    else if (kind == GACT_LEAKY) d = v > 0.f ? 1.f : 0.1f;
    else d = 1.f;
    out[i] = dy[i] * d * scale;
}

// kind 0: a + b, 1: a - b, 2: a * b, 3: relu(a + b), 4: alpha * a
__global__ __launch_bounds__(256) void binary_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     float* __restrict__ out, long n, int kind, float alpha) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = a[i];
    float r;
    if (kind == 4) r = alpha * x;
    else {
        const float y = b[i];
        r = kind == 0 ? x + y : kind == 1 ? x - y : kind == 2 ? x * y : fmaxf(x + y, 0.f);
    }
    out[i] = r;
}

// out = sum of up to eight tensors (the gradients a tensor with several consumers receives, in ONE launch: ops.py's FanOut)
struct SumNArgs { const float* in[8]; int n; };
__global__ __launch_bounds__(256) void sum_n_kernel(SumNArgs a, float* __restrict__ out, long n4, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        f32x4 r = reinterpret_cast<const f32x4*>(a.in[0])[i];
#pragma unroll
        for (int k = 1; k < 8; ++k)
            if (k < a.n) r += reinterpret_cast<const f32x4*>(a.in[k])[i];
        reinterpret_cast<f32x4*>(out)[i] = r;
    } else if (i == n4) {                                       // the (< 4) values past the last 16-byte piece
        for (long j = n4 * 4; j < n; ++j) {
            float r = a.in[0][j];
            for (int k = 1; k < a.n; ++k) r += a.in[k][j];
            out[j] = r;
        }
    }
}

// h' = (1 - z) h + z q and its adjoint (model/update.py:48,57)
__global__ __launch_bounds__(256) void gru_blend_kernel(const float* __restrict__ z, const float* __restrict__ h,
                                                        const float* __restrict__ q, float* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (1.f - z[i]) * h[i] + z[i] * q[i];
}
__global__ __launch_bounds__(256) void gru_blend_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ z,
                                                            const float* __restrict__ h, const float* __restrict__ q,
                                                            float* __restrict__ dz, float* __restrict__ dh, float* __restrict__ dq, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float g = dout[i], zz = z[i];
    dz[i] = g * (q[i] - h[i]);
    dh[i] = g * (1.f - zz);
    dq[i] = g * zz;
}

// dst[n][dst_coff + c][hw] = src[n][src_coff + c][hw], c < cc
__global__ __launch_bounds__(256) void copy_channels_kernel(const float* __restrict__ src, int src_ctotal, int src_coff,
                                                            float* __restrict__ dst, int dst_ctotal, int dst_coff, int cc, int hw, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int p = i % hw;
    const int c = (i / hw) % cc;
    const long n = i / ((long)hw * cc);
    dst[((size_t)n * dst_ctotal + dst_coff + c) * hw + p] = src[((size_t)n * src_ctotal + src_coff + c) * hw + p];
}

// out = act(x): kind 1 ReLU, 2 sigmoid, 3 tanh, 4 LeakyReLU(0.1)
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, long n, int kind) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    out[i] = kind == GACT_RELU ? fmaxf(v, 0.f) : kind == GACT_SIGMOID ? 1.f / (1.f + expf(-v)) : kind == GACT_TANH ? tanhf(v)
           : kind == GACT_LEAKY ? (v > 0.f ? v : 0.1f * v) : v;
}

// channel_shuffle (EEMFlow+.py:52-58): dst channel j * groups + g = src channel g * per + j; inverse: the other way round
__global__ __launch_bounds__(256) void shuffle_kernel(const float* __restrict__ src, float* __restrict__ dst, int c, int groups, int hw,
                                                      long total, int inverse) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int p = i % hw;
    const int ch = (i / hw) % c;
    const long n = i / ((long)hw * c);
    const int per = c / groups;
    const int g = ch / per, j = ch - g * per;             // ch as a source channel of the forward shuffle
    const int sh = j * groups + g;
    if (!inverse) dst[((size_t)n * c + sh) * hw + p] = src[i];
    else dst[i] = src[((size_t)n * c + sh) * hw + p];
}

// flow channel scale: out[:, 0] = su * x[:, 0], out[:, 1] = sv * x[:, 1]   (upsample2d_flow_as's rate, cdc_utils.py:85-93)
__global__ __launch_bounds__(256) void scale_flow2_kernel(const float* __restrict__ x, float* __restrict__ out, int hw, long total, float su,
                                                          float sv) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) out[i] = x[i] * (((i / hw) & 1) ? sv : su);
}

// avg_pool2d(2, 2) backward: dx[y][x] = dy[y/2][x/2] / 4 inside the pooled extent, 0 in an odd last row / column
__global__ __launch_bounds__(256) void pool2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int h, int w, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = i % w, y = (i / w) % h;
    const long pl = i / ((long)w * h);
    const int oh = h / 2, ow = w / 2;
    dx[i] = (y / 2 < oh && x / 2 < ow) ? 0.25f * dy[(pl * oh + y / 2) * ow + x / 2] : 0.f;
}

// F.interpolate(bilinear, align_corners=True) of [nc][h][w] -> [nc][oh][ow] and its adjoint (dx zeroed by the caller, atomics)
__device__ __forceinline__ void ac_taps(int Y, int X, int h, int w, int oh, int ow, int& y0, int& y1, int& x0, int& x1, float& ly, float& lx) {
    const float sy = oh > 1 ? (float)(h - 1) / (float)(oh - 1) : 0.f, sx = ow > 1 ? (float)(w - 1) / (float)(ow - 1) : 0.f;
    const float fy = sy * (float)Y, fx = sx * (float)X;
    y0 = (int)fy; x0 = (int)fx;
    y1 = y0 + (y0 < h - 1 ? 1 : 0); x1 = x0 + (x0 < w - 1 ? 1 : 0);
    ly = fy - (float)y0; lx = fx - (float)x0;
}
__global__ __launch_bounds__(256) void resize_ac_kernel(const float* __restrict__ in, float* __restrict__ out, int h, int w, int oh, int ow,
                                                        long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int X = i % ow, Y = (i / ow) % oh;
    const long pl = i / ((long)ow * oh);
    int y0, y1, x0, x1; float ly, lx;
    ac_taps(Y, X, h, w, oh, ow, y0, y1, x0, x1, ly, lx);
    const float* s = in + pl * h * w;
    out[i] = (1.f - ly) * ((1.f - lx) * s[y0 * w + x0] + lx * s[y0 * w + x1]) + ly * ((1.f - lx) * s[y1 * w + x0] + lx * s[y1 * w + x1]);
}
__global__ __launch_bounds__(256) void resize_ac_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, int h, int w, int oh, int ow,
                                                            long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int X = i % ow, Y = (i / ow) % oh;
    const long pl = i / ((long)ow * oh);
    int y0, y1, x0, x1; float ly, lx;
    ac_taps(Y, X, h, w, oh, ow, y0, y1, x0, x1, ly, lx);
    float* d = dx + pl * h * w;
    const float g = dout[i];
    atomicAdd(d + y0 * w + x0, g * (1.f - ly) * (1.f - lx));
    atomicAdd(d + y0 * w + x1, g * (1.f - ly) * lx);
    atomicAdd(d + y1 * w + x0, g * ly * (1.f - lx));
    atomicAdd(d + y1 * w + x1, g * ly * lx);
}

// Gather form of the same adjoint: one wave per (plane, source pixel).  The outputs whose taps touch source row y are those with
// y0 in {y - 1, y}, a contiguous range of Y; the lanes walk the (Y, X) footprint with the forward's own tap arithmetic (ac_taps) and
// add in registers, a DPP / shuffle sum and ONE store finish the pixel.  The scatter above is 4 atomics per OUTPUT pixel: upsampling
// a 6x8 level to 512x768 (EEMFlow+.py:231-232) lands 16 000 of them on every source pixel - 1.8 ms per call, half of the EEMFlow+
// training step.
__global__ __launch_bounds__(256) void resize_ac_bwd_gather_kernel(const float* __restrict__ dout, float* __restrict__ dx, int h, int w, int oh,
                                                                   int ow, long npix) {
    const int lane = threadIdx.x & 63;
    const long pid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);          // (plane, y, x)
    if (pid >= npix) return;
    const int x = pid % w, y = (pid / w) % h;
    const long pl = pid / ((long)w * h);
    const float sy = oh > 1 ? (float)(h - 1) / (float)(oh - 1) : 0.f, sx = ow > 1 ? (float)(w - 1) / (float)(ow - 1) : 0.f;
    // candidate ranges, widened by two so that the float rounding of sy * Y cannot lose an output; ac_taps decides
    int Ya = 0, Yb = oh - 1, Xa = 0, Xb = ow - 1;
    if (sy > 0.f) { Ya = max(0, (int)floorf((float)(y - 1) / sy) - 2); Yb = min(oh - 1, (int)ceilf((float)(y + 1) / sy) + 2); }
    if (sx > 0.f) { Xa = max(0, (int)floorf((float)(x - 1) / sx) - 2); Xb = min(ow - 1, (int)ceilf((float)(x + 1) / sx) + 2); }
    const int ny = Yb - Ya + 1, nx = Xb - Xa + 1;
    const float* g = dout + pl * (long)oh * ow;
    float acc = 0.f;
    for (int e = lane; e < ny * nx; e += 64) {
        const int Y = Ya + e / nx, X = Xa + e % nx;
        int y0, y1, x0, x1; float ly, lx;
        ac_taps(Y, X, h, w, oh, ow, y0, y1, x0, x1, ly, lx);
        const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);     // y0 == y1 == y on the last row: (1 - ly) + ly
        const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
        if (wy != 0.f && wx != 0.f) acc += g[(long)Y * ow + X] * wy * wx;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0) dx[pid] = acc;
}

// ------------------------------------------------------------------------------------------------ norms
__device__ __forceinline__ void block_sum2(double& a, double& b, double* sh) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { a += __shfl_xor(a, d); b += __shfl_xor(b, d); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { sh[wave * 2] = a; sh[wave * 2 + 1] = b; }
    __syncthreads();
    a = b = 0.0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) { a += sh[k * 2]; b += sh[k * 2 + 1]; }
}

// InstanceNorm2d(affine=False, eps) [+ ReLU] backward, one block per (n, c) plane; x is the layer input (the statistics are
// recomputed), y the layer output (ReLU gate):  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * [y > 0]
__global__ __launch_bounds__(1024) void instnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ dy, float* __restrict__ dx, int hw, int relu, float eps) {
    __shared__ double sh[32];
    const size_t base = (size_t)blockIdx.x * hw;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < hw; i += blockDim.x) { const double v = x[base + i]; s += v; q += v * v; }
    block_sum2(s, q, sh);
    const double mean = s / hw;
    double var = q / hw - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps)), meanf = (float)mean;
    double sg = 0.0, sgx = 0.0;
    for (int i = threadIdx.x; i < hw; i += blockDim.x) {
        float g = dy[base + i];
        if (relu && !(y[base + i] > 0.f)) g = 0.f;
        const float xh = (x[base + i] - meanf) * rstd;
        sg += g; sgx += (double)g * xh;
    }
    block_sum2(sg, sgx, sh);
    const float mg = (float)(sg / hw), mgx = (float)(sgx / hw);
    for (int i = threadIdx.x; i < hw; i += blockDim.x) {
        float g = dy[base + i];
        if (relu && !(y[base + i] > 0.f)) g = 0.f;
        const float xh = (x[base + i] - meanf) * rstd;
        dx[base + i] = rstd * (g - mg - xh * mgx);
    }
}

// BatchNorm2d in training mode (model/extractor.py:31-35 with the module in train()): one block per channel (1024 threads for large
// planes: with 64-128 channels the grid is a quarter of the chip, so the threads per block carry the memory parallelism).
// forward: batch mean / biased variance over (N, H, W); y = (x - mean) * rstd * w + b [ReLU]; running statistics updated in place
// with momentum (unbiased variance), save_mean / save_rstd kept for the backward.
__global__ __launch_bounds__(1024) void bn_train_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, float* __restrict__ y,
                                                           float* __restrict__ save_mean, float* __restrict__ save_rstd, int n, int c, int hw,
                                                           float momentum, float eps, int relu) {
    __shared__ double sh[32];
    const int ch = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int img = 0; img < n; ++img) {
        const float* p = x + ((size_t)img * c + ch) * hw;
        for (int i = threadIdx.x; i < hw; i += blockDim.x) { const double v = p[i]; s += v; q += v * v; }
    }
    block_sum2(s, q, sh);
    const double cnt = (double)n * hw;
    const double mean = s / cnt;
    double var = q / cnt - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps)), meanf = (float)mean;
    const float ww = w[ch], bb = b[ch];
    for (int img = 0; img < n; ++img) {
        const size_t o = ((size_t)img * c + ch) * hw;
        for (int i = threadIdx.x; i < hw; i += blockDim.x) {
            float v = (x[o + i] - meanf) * rstd * ww + bb;
            if (relu) v = fmaxf(v, 0.f);
            y[o + i] = v;
        }
    }
    if (threadIdx.x == 0) {
        save_mean[ch] = meanf;
        save_rstd[ch] = rstd;
        const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
        running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * meanf;
        running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unbiased;
    }
}
// backward: g = dy * [y > 0]; dw = sum g * xhat, db = sum g, dx = w * rstd * (g - mean(g) - xhat * mean(g * xhat))
__global__ __launch_bounds__(1024) void bn_train_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ dy, const float* __restrict__ w,
                                                           const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
                                                           float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db, int n, int c,
                                                           int hw, int relu) {
    __shared__ double sh[32];
    const int ch = blockIdx.x;
    const float meanf = save_mean[ch], rstd = save_rstd[ch];
    double sg = 0.0, sgx = 0.0;
    for (int img = 0; img < n; ++img) {
        const size_t o = ((size_t)img * c + ch) * hw;
        for (int i = threadIdx.x; i < hw; i += blockDim.x) {
            float g = dy[o + i];
            if (relu && !(y[o + i] > 0.f)) g = 0.f;
            sg += g; sgx += (double)g * ((x[o + i] - meanf) * rstd);
        }
    }
    block_sum2(sg, sgx, sh);
    const double cnt = (double)n * hw;
    const float mg = (float)(sg / cnt), mgx = (float)(sgx / cnt), k = w[ch] * rstd;
    for (int img = 0; img < n; ++img) {
        const size_t o = ((size_t)img * c + ch) * hw;
        for (int i = threadIdx.x; i < hw; i += blockDim.x) {
            float g = dy[o + i];
            if (relu && !(y[o + i] > 0.f)) g = 0.f;
            dx[o + i] = k * (g - mg - (x[o + i] - meanf) * rstd * mgx);
        }
    }
    if (threadIdx.x == 0) { dw[ch] = (float)sgx; db[ch] = (float)sg; }
}

// ---- BatchNorm2d in EVAL mode (frozen statistics: ERAFT.freeze_bn, model/eraft.py:69-72) - an affine map per channel:
// y = relu?((x - running_mean) * rstd * w + b), rstd = 1 / sqrt(running_var + eps).  One block per channel.
__global__ __launch_bounds__(1024) void bn_eval_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                          const float* __restrict__ rm, const float* __restrict__ rv, float* __restrict__ y,
                                                          int n, int c, int hw, float eps, int relu) {
    const int ch = blockIdx.x;
    const float mean = rm[ch], k = w[ch] * (1.f / sqrtf(rv[ch] + eps)), bias = b[ch];
    for (int img = 0; img < n; ++img) {
        const size_t o = ((size_t)img * c + ch) * hw;
        for (int i = threadIdx.x; i < hw; i += blockDim.x) {
            const float v = (x[o + i] - mean) * k + bias;
            y[o + i] = relu ? fmaxf(v, 0.f) : v;
        }
    }
}
// backward: g = dy * [y > 0]; dx = g * w * rstd; dw = sum g * (x - mean) * rstd; db = sum g   (the statistics are constants)
__global__ __launch_bounds__(1024) void bn_eval_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                                          const float* __restrict__ w, const float* __restrict__ rm, const float* __restrict__ rv,
                                                          float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db, int n, int c,
                                                          int hw, float eps, int relu) {
    __shared__ double sh[32];
    const int ch = blockIdx.x;
    const float mean = rm[ch], rstd = 1.f / sqrtf(rv[ch] + eps), k = w[ch] * rstd;
    double sg = 0.0, sgx = 0.0;
    for (int img = 0; img < n; ++img) {
        const size_t o = ((size_t)img * c + ch) * hw;
        for (int i = threadIdx.x; i < hw; i += blockDim.x) {
            float g = dy[o + i];
            if (relu && !(y[o + i] > 0.f)) g = 0.f;
            sg += g; sgx += (double)g * ((x[o + i] - mean) * rstd);
            dx[o + i] = g * k;
        }
    }
    block_sum2(sg, sgx, sh);
    if (threadIdx.x == 0) { dw[ch] = (float)sgx; db[ch] = (float)sg; }
}

inline unsigned nblk(long n) { return (unsigned)((n + 255) / 256); }

// ------------------------------------------------------------------------------------------------ packed-weight plans
struct Plan {
    int* idx = nullptr;        // device: packed[i] = idx[i] ? w[idx[i] - 1] : 0
    size_t n = 0;
    int* idx16 = nullptr;      // the same for gconv16's fragment-order stream (stride-1 convs with 16-aligned channel counts), or NULL
    size_t n16 = 0;
    int* idxfew = nullptr;     // the same for the few-output direct kernel ([cin][tap][8], <= 8 couts, 3x3, one segment), or NULL
    size_t nfew = 0;
};
std::map<std::string, Plan> g_plans;              // per process; the tables depend on layer shapes only
std::mutex g_plans_mutex;                         // autograd runs backward ops on its own thread
struct Scratch { float* p = nullptr; size_t cap = 0; int dev = -1; };
thread_local Scratch g_scratch[4];                // [0] packed weights, [1] zero bias / zero page, [2] the bf16-piece packing, [3] store sink

int scratch_get(Scratch& s, size_t floats, float** out) {
    int dev = 0;
    EEM_HIP_CHECK(hipGetDevice(&dev));
    if (s.p == nullptr || s.dev != dev || s.cap < floats) {
        if (s.p && s.dev == dev) EEM_HIP_CHECK(hipFree(s.p));          // synchronises with the launches that read it
        s.p = nullptr;
        s.cap = floats + floats / 2 + 1024;
        EEM_HIP_CHECK(hipMalloc(&s.p, s.cap * sizeof(float)));
        EEM_HIP_CHECK(hipMemset(s.p, 0, s.cap * sizeof(float)));
        s.dev = dev;
    }
    *out = s.p;
    return EEM_OK;
}

// index table of gconv16's packing for the same (index-valued) weights, when the shape qualifies
void plan_add16(Plan& p, const float* iw, int cout, const int* cs, int nseg, int kh, int kw) {
    if (!gconv16_shape(cout, cs, nseg, kh, kw, 1)) return;
    std::vector<float> pk(gconv16_packed_floats(cout, cs, nseg, kh, kw), 0.f);
    gconv16_pack(iw, cout, cs, nseg, kh, kw, pk.data());
    std::vector<int> idx(pk.size());
    for (size_t i = 0; i < pk.size(); ++i) idx[i] = (int)pk[i];
    p.n16 = idx.size();
    if (hipMalloc(&p.idx16, p.n16 * sizeof(int)) != hipSuccess ||
        hipMemcpy(p.idx16, idx.data(), p.n16 * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
        p.idx16 = nullptr;                       // the generic packing still serves the layer
        p.n16 = 0;
    }
}

void plan_addfew(Plan& p, const float* iw, int cout, const int* cs, int nseg, int kh, int kw) {
    if (nseg != 1 || cout > 8 || kh != 3 || kw != 3) return;
    std::vector<float> pk(fewout_packed_floats(cs[0], kh, kw), 0.f);
    fewout_pack(iw, cout, cs[0], kh, kw, pk.data());
    std::vector<int> idx(pk.size());
    for (size_t i = 0; i < pk.size(); ++i) idx[i] = (int)pk[i];
    p.nfew = idx.size();
    if (hipMalloc(&p.idxfew, p.nfew * sizeof(int)) != hipSuccess ||
        hipMemcpy(p.idxfew, idx.data(), p.nfew * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
        p.idxfew = nullptr;
        p.nfew = 0;
    }
}

// ---- packed weights kept per (parameter, version): a recurrent model applies the same weights a dozen times per step and the data
// gradient needs them again - the caller names the weight tensor of the next launches (eemop_pack_hint: a token that is never reused
// + the tensor's version counter), and a packing is redone only when that version changes.  Without a hint (token 0) every launch packs
// into the shared scratch, as before.
thread_local long long t_pack_token = 0, t_pack_version = -1;
struct PackEntry { float* p = nullptr; size_t cap = 0; long long version = -1; int dev = -1; };
struct PackKey {
    long long token; const float* w; const int* idx; hipStream_t st;
    bool operator<(const PackKey& o) const { return std::tie(token, w, idx, st) < std::tie(o.token, o.w, o.idx, o.st); }
};
std::map<PackKey, PackEntry> g_pack_cache;
std::mutex g_pack_mutex;                          // autograd runs backward ops on its own thread

// `fill(dst)` writes the n floats of the packing; kind distinguishes packings derived from the same (weights, table)
template <class F>
int packed_cached(const float* w, const int* idx, int kind, size_t n, hipStream_t st, float** out, int scratch_slot, F&& fill) {
    int rc;
    if (t_pack_token == 0) {
        if ((rc = scratch_get(g_scratch[scratch_slot], n, out)) != EEM_OK) return rc;
        return fill(*out);
    }
    int dev = 0;
    EEM_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_pack_mutex);
    const int* kidx = idx + kind;                     // (a key, never dereferenced)
    const PackKey key{t_pack_token, w, kidx, st};
    if (g_pack_cache.find(key) == g_pack_cache.end()) {
        // a packing is kept for ONE stream at a time: a caller that moves to a new stream every step must not grow the cache
        auto it = g_pack_cache.lower_bound(PackKey{t_pack_token, w, kidx, nullptr});
        while (it != g_pack_cache.end() && it->first.token == t_pack_token && it->first.w == w && it->first.idx == kidx) {
            if (it->second.p) (void)hipFree(it->second.p);      // synchronises with the launches that read it
            it = g_pack_cache.erase(it);
        }
    }
    PackEntry& e = g_pack_cache[key];
    if (e.p == nullptr || e.cap < n || e.dev != dev) {
        if (e.p) (void)hipFree(e.p);
        e.p = nullptr; e.cap = 0; e.version = -1; e.dev = dev;
        EEM_HIP_CHECK(hipMalloc(&e.p, n * sizeof(float)));
        e.cap = n;
    }
    if (e.version != t_pack_version) {
        if ((rc = fill(e.p)) != EEM_OK) return rc;
        e.version = t_pack_version;
    }
    *out = e.p;
    return EEM_OK;
}

int packed_weights(const float* w, const int* idx, size_t n, hipStream_t st, float** out) {
    return packed_cached(w, idx, 0, n, st, out, 0, [&](float* dst) { return repack_launch(w, idx, dst, (long)n, st); });
}

// Packs the weights for the launch described by `a` (everything but the weight pointers filled in): the few-output kernel's layout
// or the LDS-tiled kernel's stream when the launch qualifies for them, else the generic kernel's.
int pack_for(const Plan* pl, const float* w, GConvArgs& a, hipStream_t st) {
    int rc;
    float* pk = nullptr;
    float* probe = nullptr;                       // the eligibility tests only look at "is there a packing"
    if ((rc = scratch_get(g_scratch[1], 1024, &probe)) != EEM_OK) return rc;
    if (pl->idxfew) {
        a.wfew = probe;
        if (fewout_supported(a)) {
            if ((rc = packed_weights(w, pl->idxfew, pl->nfew, st, &pk)) != EEM_OK) return rc;
            a.wfew = pk;
            return EEM_OK;
        }
        a.wfew = nullptr;
    }
    if (pl->idx16) {
        a.wpk16 = probe;
        a.zero_page = probe;
        a.wpkb = probe;
        const bool onb = gconvb_supported(a);         // the bf16-piece kernel (gconvb.hip): its packing is derived from gconv16's, on the device
        a.wpkb = nullptr;
        if (onb || gconv16_supported(a)) {
            if ((rc = packed_weights(w, pl->idx16, pl->n16, st, &pk)) != EEM_OK) return rc;
            a.wpk16 = pk;
            if (onb) {
                int cin = 0;
                for (int sgi = 0; sgi < a.nseg; ++sgi) cin += a.seg[sgi].c;
                const int cs1[1] = {cin};
                float* pkb = nullptr;
                const float* src16 = pk;
                const int cout = a.cout, taps = a.kh * a.kw;
                // (scratch slot 2 when no hint names the weights: the fp32 stream sits in slot 0)
                if ((rc = packed_cached(w, pl->idx16, 1, gconvb_packed_floats(cout, cs1, 1, a.kh, a.kw), st, &pkb, 2,
                                        [&](float* dst) { return gconvb_from16_launch(src16, cout, cin, taps, dst, st); })) != EEM_OK) return rc;
                a.wpkb = pkb;
            }
            return EEM_OK;
        }
        a.wpk16 = nullptr;
    }
    if ((rc = packed_weights(w, pl->idx, pl->n, st, &pk)) != EEM_OK) return rc;
    a.wpk = pk;
    return EEM_OK;
}

// forward plan: gconv_pack of w [cout][cin][kh][kw] read as `nseg` input segments
int plan_fwd(int cout, const int* cs, int nseg, int kh, int kw, Plan** out) {
    int dev = 0;
    EEM_HIP_CHECK(hipGetDevice(&dev));
    char key[160];
    snprintf(key, sizeof(key), "f:%d:%d:%d,%d,%d:%d:%dx%d", dev, cout, cs[0], nseg > 1 ? cs[1] : 0, nseg > 2 ? cs[2] : 0, nseg, kh, kw);
    std::lock_guard<std::mutex> lock(g_plans_mutex);
    auto it = g_plans.find(key);
    if (it == g_plans.end()) {
        int cin = 0;
        for (int s = 0; s < nseg; ++s) cin += cs[s];
        const size_t nw = (size_t)cout * cin * kh * kw;
        EEM_REQUIRE(nw < (size_t)16000000, "conv weights with %zu elements exceed the exact-index range of the pack table", nw);
        std::vector<float> iw(nw);
        for (size_t i = 0; i < nw; ++i) iw[i] = (float)(i + 1);
        std::vector<float> pk(gconv_packed_floats(cout, cs, nseg, kh, kw), 0.f);
        gconv_pack(iw.data(), cout, cs, nseg, kh, kw, pk.data());
        std::vector<int> idx(pk.size());
        for (size_t i = 0; i < pk.size(); ++i) idx[i] = (int)pk[i];
        Plan p;
        p.n = idx.size();
        EEM_HIP_CHECK(hipMalloc(&p.idx, p.n * sizeof(int)));
        EEM_HIP_CHECK(hipMemcpy(p.idx, idx.data(), p.n * sizeof(int), hipMemcpyHostToDevice));
        plan_add16(p, iw.data(), cout, cs, nseg, kh, kw);
        plan_addfew(p, iw.data(), cout, cs, nseg, kh, kw);
        it = g_plans.emplace(key, p).first;
    }
    *out = &it->second;
    return EEM_OK;
}

// data-gradient plan: T[ci][co][kh-1-ky][kw-1-kx] = w[co][ci0 + ci][ky][kx] for ci < cic, packed as a conv with `cic` outputs
int plan_bwd(int cout, int cin, int ci0, int cic, int kh, int kw, Plan** out) {
    int dev = 0;
    EEM_HIP_CHECK(hipGetDevice(&dev));
    char key[160];
    snprintf(key, sizeof(key), "b:%d:%d:%d:%d:%d:%dx%d", dev, cout, cin, ci0, cic, kh, kw);
    std::lock_guard<std::mutex> lock(g_plans_mutex);
    auto it = g_plans.find(key);
    if (it == g_plans.end()) {
        const size_t nw = (size_t)cout * cin * kh * kw;
        EEM_REQUIRE(nw < (size_t)16000000, "conv weights with %zu elements exceed the exact-index range of the pack table", nw);
        std::vector<float> T((size_t)cic * cout * kh * kw);
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cic; ++ci)
                for (int ky = 0; ky < kh; ++ky)
                    for (int kx = 0; kx < kw; ++kx)
                        T[(((size_t)ci * cout + co) * kh + (kh - 1 - ky)) * kw + (kw - 1 - kx)] =
                            (float)((((size_t)co * cin + ci0 + ci) * kh + ky) * kw + kx + 1);
        const int cs[1] = {cout};
        std::vector<float> pk(gconv_packed_floats(cic, cs, 1, kh, kw), 0.f);
        gconv_pack(T.data(), cic, cs, 1, kh, kw, pk.data());
        std::vector<int> idx(pk.size());
        for (size_t i = 0; i < pk.size(); ++i) idx[i] = (int)pk[i];
        Plan p;
        p.n = idx.size();
        EEM_HIP_CHECK(hipMalloc(&p.idx, p.n * sizeof(int)));
        EEM_HIP_CHECK(hipMemcpy(p.idx, idx.data(), p.n * sizeof(int), hipMemcpyHostToDevice));
        plan_add16(p, T.data(), cic, cs, 1, kh, kw);
        plan_addfew(p, T.data(), cic, cs, 1, kh, kw);
        it = g_plans.emplace(key, p).first;
    }
    *out = &it->second;
    return EEM_OK;
}

}  // namespace

// ================================================================================================ packed-weight cache
extern "C" int eemop_pack_hint(long long token, long long version) {
    t_pack_token = token;
    t_pack_version = version;
    return EEM_OK;
}

extern "C" long long eemop_pack_cache_bytes() {
    std::lock_guard<std::mutex> lock(g_pack_mutex);
    long long n = 0;
    for (const auto& kv : g_pack_cache) n += (long long)kv.second.cap * (long long)sizeof(float);
    return n;
}

extern "C" int eemop_pack_forget(long long token) {
    std::lock_guard<std::mutex> lock(g_pack_mutex);
    for (auto it = g_pack_cache.begin(); it != g_pack_cache.end();) {
        if (it->first.token == token) {
            if (it->second.p) (void)hipFree(it->second.p);
            it = g_pack_cache.erase(it);
        } else {
            ++it;
        }
    }
    return EEM_OK;
}

// ================================================================================================ convolution
extern "C" int eemop_conv2d_fwd(const float* x0, int c0, const float* x1, int c1, const float* x2, int c2, const float* w,
                                const float* bias, int n, int hin, int win, int cout, int kh, int kw, int stride, int ph, int pw, int act,
                                float out_scale, float* out, int out_ctotal, int out_coff, void* stream) {
    EEM_REQUIRE(x0 && w && out && c0 >= 1 && n >= 1 && cout >= 1, "eemop_conv2d_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int nseg = 1 + (x1 != nullptr) + (x2 != nullptr);
    EEM_REQUIRE(!(x2 && !x1), "eemop_conv2d_fwd: segment 2 without segment 1");
    const int cs[3] = {c0, c1, c2};
    Plan* pl = nullptr;
    int rc = plan_fwd(cout, cs, nseg, kh, kw, &pl);
    if (rc != EEM_OK) return rc;
    GConvArgs a;
    memset(&a, 0, sizeof(a));
    a.nseg = nseg;
    const float* xs[3] = {x0, x1, x2};
    for (int s = 0; s < nseg; ++s) { a.seg[s].ptr = xs[s]; a.seg[s].c = cs[s]; a.seg[s].ctotal = cs[s]; a.seg[s].coff = 0; }
    a.shift = bias;
    a.out = out; a.out_ctotal = out_ctotal; a.out_coff = out_coff;
    a.n = n; a.hin = hin; a.win = win;
    a.hout = (hin + 2 * ph - kh) / stride + 1; a.wout = (win + 2 * pw - kw) / stride + 1;
    a.cout = cout; a.kh = kh; a.kw = kw; a.stride = stride; a.pad_h = ph; a.pad_w = pw;
    a.act = act; a.epi = GEPI_PLAIN; a.out_scale = out_scale;
    if ((rc = pack_for(pl, w, a, st)) != EEM_OK) return rc;
    return gconv_launch(a, st);
}

// dx [n][cic][hin][win] = conv^T(dy [n][cout][hout][wout], w[:, ci0:ci0+cic])
extern "C" int eemop_conv2d_bwd_data(const float* dy, const float* w, int n, int hin, int win, int cin, int ci0, int cic, int cout,
                                     int kh, int kw, int stride, int ph, int pw, float* dx, void* stream) {
    EEM_REQUIRE(dy && w && dx && cic >= 1 && ci0 >= 0 && ci0 + cic <= cin, "eemop_conv2d_bwd_data: bad arguments");
    EEM_REQUIRE(stride == 1 || stride == 2, "eemop_conv2d_bwd_data: stride %d", stride);
    hipStream_t st = (hipStream_t)stream;
    const int hout = (hin + 2 * ph - kh) / stride + 1, wout = (win + 2 * pw - kw) / stride + 1;
    int rc;
    if (stride == 2 && ci0 == 0 && cic == cin && kh == kw && ((kh == 3 && ph == 1 && pw == 1) || (kh == 1 && ph == 0 && pw == 0))) {
        // the encoders' downsampling convs (model/extractor.py:13,33-36): four parity classes of dX as dense convs of dY (dgrad_s2.hip)
        DgradS2Args d;
        memset(&d, 0, sizeof(d));
        d.dy = dy; d.w = w; d.dx = dx;
        d.n = n; d.cin = cin; d.cout = cout; d.hin = hin; d.win = win; d.hout = hout; d.wout = wout;
        float *zp = nullptr, *sink = nullptr;
        if ((rc = scratch_get(g_scratch[1], 1024, &zp)) != EEM_OK || (rc = scratch_get(g_scratch[3], 1024, &sink)) != EEM_OK) return rc;
        d.zero_page = zp; d.trash = sink;
        if (dgrad_s2w_supported(d, kh)) return dgrad_s2w_launch(d, kh, st);
    }
    Plan* pl = nullptr;
    rc = plan_bwd(cout, cin, ci0, cic, kh, kw, &pl);
    if (rc != EEM_OK) return rc;
    GConvArgs a;
    memset(&a, 0, sizeof(a));
    a.nseg = 1;
    a.seg[0].ptr = dy; a.seg[0].c = cout; a.seg[0].ctotal = cout; a.seg[0].coff = 0;
    a.out = dx; a.out_ctotal = cic; a.out_coff = 0;
    a.n = n; a.hin = hout; a.win = wout; a.hout = hin; a.wout = win; a.cout = cic;
    a.kh = kh; a.kw = kw; a.stride = 1; a.pad_h = kh - 1 - ph; a.pad_w = kw - 1 - pw;
    a.tstride = stride;
    a.act = GACT_NONE; a.epi = GEPI_PLAIN; a.out_scale = 1.f;
    if ((rc = pack_for(pl, w, a, st)) != EEM_OK) return rc;
    return gconv_launch(a, st);
}

// dw [cout][cin][kh][kw] += dy (x) x for the input-channel slice [ci0, ci0 + cic) (x is that slice: [n][cic][hin][win]);
// db [cout] += sum dy when not NULL.  The caller zeroes dw / db once per backward.
extern "C" int eemop_conv2d_bwd_weight(const float* x, const float* dy, int n, int hin, int win, int cin, int ci0, int cic, int cout,
                                       int kh, int kw, int stride, int ph, int pw, float* dw, float* db, void* stream) {
    EEM_REQUIRE(x && dy && dw && cic >= 1 && ci0 >= 0 && ci0 + cic <= cin, "eemop_conv2d_bwd_weight: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int hout = (hin + 2 * ph - kh) / stride + 1, wout = (win + 2 * pw - kw) / stride + 1;
    int rc;
    {
        static const bool log_calls = [] { const char* e = getenv("EEM_WGRAD_LOG"); return e && e[0] == '1'; }();
        if (log_calls)                                               // (measurement: the shapes a training step asks for, tools/wgrad_shapes.sh)
            fprintf(stderr, "WGRAD cic=%d cout=%d k=%dx%d s=%d n=%d hin=%d win=%d cin=%d ci0=%d db=%d\n", cic, cout, kh, kw, stride, n, hin, win, cin,
                    ci0, db != nullptr);
    }
    {
        // LDS-tiled kernel (wgrad_enc.hip: G and the haloed X tile by LDS-DMA, K split over the waves, 65-95 TFLOP/s) where the shape
        // allows; it also leaves the bias gradient
        WgradArgs a;
        a.x = x; a.x_ctotal = cic; a.x_coff = 0; a.cin = cic;
        a.g = dy; a.gate = nullptr; a.g_ctotal = cout; a.g_coff = 0; a.g_cmul = 1; a.cout = cout;
        a.dw = dw;
        a.n = n; a.hin = hin; a.win = win; a.hout = hout; a.wout = wout; a.k = kh; a.stride = stride; a.pad = ph;
        a.db = db;
        a.kh = kh; a.kw = kw; a.ph = ph; a.pw = pw; a.dw_cin = cin; a.dw_coff = ci0;
        float* zp = nullptr;
        if ((rc = scratch_get(g_scratch[1], 1024, &zp)) != EEM_OK) return rc;
        a.zero_page = zp;
        if (wgrad_ring_supported(a) && wgrad_ring_preferred(a)) return wgrad_ring_launch(a, st);
        if (wgrad_few_supported(a)) return wgrad_few_launch(a, st);                          // (<= 8 couts: the flow heads)
        if (kh == 3 && kw == 3 && wgrad_enc_supported(a)) return wgrad_enc_launch(a, st);   // (16 / 32 / 64 couts on blocks of their own height)
        if (wgrad_wide_supported(a)) return wgrad_wide_launch(a, st);
    }
    for (int c0 = 0; c0 < cout; c0 += 128) {                     // the kernel holds at most 128 couts per block
        WgradArgs a;
        a.x = x; a.x_ctotal = cic; a.x_coff = 0; a.cin = cic;
        a.g = dy; a.gate = nullptr; a.g_ctotal = cout; a.g_coff = c0; a.g_cmul = 1; a.cout = cout - c0 < 128 ? cout - c0 : 128;
        a.dw = dw + (size_t)c0 * cin * kh * kw;
        a.n = n; a.hin = hin; a.win = win; a.hout = hout; a.wout = wout; a.k = kh; a.stride = stride; a.pad = ph;
        a.zero_page = nullptr; a.db = nullptr;
        a.kh = kh; a.kw = kw; a.ph = ph; a.pw = pw; a.dw_cin = cin; a.dw_coff = ci0;
        if ((rc = tr_wgrad_launch_batch(&a, 1, st)) != EEM_OK) return rc;
    }
    if (db) return tr_bias_grad_launch(dy, nullptr, cout, 0, 1, cout, n, hout * wout, db, st);
    return EEM_OK;
}

// The same for a conv over up to three channel-concatenated inputs (x_s [n][c_s][hin][win]; x1 / x2 may be NULL) in ONE call:
// dw [cout][c0 + c1 + c2][kh][kw].  Where the ring kernel is the faster one (wgrad_ring_preferred) the segments ride one launch - the
// GRU's (1, 5) / (5, 1) convs over [h | inp | motion] are 106 us as one launch against 3 x 48; elsewhere one call per segment as before.
extern "C" int eemop_conv2d_bwd_weight_cat(const float* x0, int c0, const float* x1, int c1, const float* x2, int c2, const float* dy, int n,
                                           int hin, int win, int cout, int kh, int kw, int stride, int ph, int pw, float* dw, float* db,
                                           void* stream) {
    EEM_REQUIRE(x0 && dy && dw && c0 >= 1 && c1 >= 0 && c2 >= 0 && (c1 == 0 || x1) && (c2 == 0 || (x2 && c1 > 0)),
                "eemop_conv2d_bwd_weight_cat: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int cin = c0 + c1 + c2;
    const int hout = (hin + 2 * ph - kh) / stride + 1, wout = (win + 2 * pw - kw) / stride + 1;
    const float* xs[3] = {x0, x1, x2};
    const int cs[3] = {c0, c1, c2};
    const int nseg = c2 > 0 ? 3 : (c1 > 0 ? 2 : 1);
    if (nseg > 1) {
        WgradArgs a;
        a.x = x0; a.x_ctotal = c0; a.x_coff = 0; a.cin = cin;
        a.g = dy; a.gate = nullptr; a.g_ctotal = cout; a.g_coff = 0; a.g_cmul = 1; a.cout = cout;
        a.dw = dw;
        a.n = n; a.hin = hin; a.win = win; a.hout = hout; a.wout = wout; a.k = kh; a.stride = stride; a.pad = ph;
        a.db = db;
        a.kh = kh; a.kw = kw; a.ph = ph; a.pw = pw; a.dw_cin = cin; a.dw_coff = 0;
        a.nxseg = nseg;
        for (int s = 0; s < nseg; ++s) { a.xs[s] = xs[s]; a.xsc[s] = cs[s]; }
        float* zp = nullptr;
        int rc = scratch_get(g_scratch[1], 1024, &zp);
        if (rc != EEM_OK) return rc;
        a.zero_page = zp;
        if (wgrad_ring_supported(a) && wgrad_ring_preferred(a)) return wgrad_ring_launch(a, st);
    }
    int ci0 = 0;
    for (int s = 0; s < nseg; ++s) {
        const int rc = eemop_conv2d_bwd_weight(xs[s], dy, n, hin, win, cin, ci0, cs[s], cout, kh, kw, stride, ph, pw, dw, s == 0 ? db : nullptr, stream);
        if (rc != EEM_OK) return rc;
        ci0 += cs[s];
    }
    return EEM_OK;
}

// ================================================================================================ elementwise
extern "C" int eemop_act_bwd(const float* dy, const float* y, long long n, int kind, float scale, float* out, void* stream) {
    EEM_REQUIRE(dy && y && out && n >= 0, "eemop_act_bwd: bad arguments");
    if (n == 0) return EEM_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, dy, y, out, (long)n, kind, scale);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

extern "C" int eemop_binary(int kind, const float* a, const float* b, float alpha, long long n, float* out, void* stream) {
    EEM_REQUIRE(a && out && (b || kind == 4) && kind >= 0 && kind <= 4, "eemop_binary: bad arguments");
    if (n == 0) return EEM_OK;
    hipLaunchKernelGGL(binary_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, (long)n, kind, alpha);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

extern "C" int eemop_sum_n(const float* const* in, int count, long long n, float* out, void* stream) {
    EEM_REQUIRE(in && out && count >= 1 && count <= 8, "eemop_sum_n: 1..8 inputs, got %d", count);
    if (n == 0) return EEM_OK;
    SumNArgs a;
    bool aligned = ((uintptr_t)out & 15) == 0;
    for (int k = 0; k < 8; ++k) {
        a.in[k] = k < count ? in[k] : in[0];
        EEM_REQUIRE(a.in[k], "eemop_sum_n: NULL input %d", k);
        aligned = aligned && ((uintptr_t)a.in[k] & 15) == 0;
    }
    a.n = count;
    const long n4 = aligned ? (long)(n / 4) : 0;
    hipLaunchKernelGGL(sum_n_kernel, dim3(nblk(n4 + 1)), dim3(256), 0, (hipStream_t)stream, a, out, n4, (long)n);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

extern "C" int eemop_gru_blend(const float* z, const float* h, const float* q, long long n, float* out, void* stream) {
    EEM_REQUIRE(z && h && q && out, "eemop_gru_blend: NULL argument");
    hipLaunchKernelGGL(gru_blend_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, z, h, q, out, (long)n);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

extern "C" int eemop_gru_blend_bwd(const float* dout, const float* z, const float* h, const float* q, long long n, float* dz, float* dh,
                                   float* dq, void* stream) {
    EEM_REQUIRE(dout && z && h && q && dz && dh && dq, "eemop_gru_blend_bwd: NULL argument");
    hipLaunchKernelGGL(gru_blend_bwd_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, dout, z, h, q, dz, dh, dq, (long)n);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

extern "C" int eemop_copy_channels(const float* src, int src_ctotal, int src_coff, float* dst, int dst_ctotal, int dst_coff, int cc, int n,
                                   int hw, void* stream) {
    EEM_REQUIRE(src && dst && cc >= 1 && n >= 1 && hw >= 1, "eemop_copy_channels: bad arguments");
    const long total = (long)n * cc * hw;
    hipLaunchKernelGGL(copy_channels_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, src, src_ctotal, src_coff, dst,
                       dst_ctotal, dst_coff, cc, hw, total);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

extern "C" int eemop_coords_init(float* coords0, float* coords1, const float* flow_init, int batch, int h, int w, void* stream) {
    EEM_REQUIRE(coords0 && coords1, "eemop_coords_init: NULL argument");
    return er_coords_init_launch(coords0, coords1, flow_init, batch, h, w, (hipStream_t)stream);
}

extern "C" int eemop_replicate_pad(const float* in, float* out, int nc, int h, int w, int left, int right, int top, int bottom, void* stream) {
    EEM_REQUIRE(in && out, "eemop_replicate_pad: NULL argument");
    return er_pad_launch(in, out, nc, h, w, left, right, top, bottom, (hipStream_t)stream);
}

// ================================================================================================ norms
extern "C" int eemop_instnorm_fwd(const float* x, const float* res, int planes, int hw, int relu, float* out, void* stream) {
    EEM_REQUIRE(x && out, "eemop_instnorm_fwd: NULL argument");
    return er_instnorm_launch(x, out, res, planes, hw, relu, (hipStream_t)stream);
}

extern "C" int eemop_instnorm_bwd(const float* x, const float* y, const float* dy, int planes, int hw, int relu, float* dx, void* stream) {
    EEM_REQUIRE(x && y && dy && dx, "eemop_instnorm_bwd: NULL argument");
    hipLaunchKernelGGL(instnorm_bwd_kernel, dim3(planes), dim3(hw >= 16384 ? 1024 : 256), 0, (hipStream_t)stream, x, y, dy, dx, hw, relu, 1e-5f);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

extern "C" int eemop_batchnorm_train_fwd(const float* x, const float* weight, const float* bias, float* running_mean, float* running_var,
                                         int n, int c, int hw, float momentum, float eps, int relu, float* y, float* save_mean,
                                         float* save_rstd, void* stream) {
    EEM_REQUIRE(x && weight && bias && running_mean && running_var && y && save_mean && save_rstd, "eemop_batchnorm_train_fwd: NULL argument");
    hipLaunchKernelGGL(bn_train_fwd_kernel, dim3(c), dim3((long)n * hw >= 16384 ? 1024 : 256), 0, (hipStream_t)stream, x, weight, bias, running_mean, running_var, y,
                       save_mean, save_rstd, n, c, hw, momentum, eps, relu);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

extern "C" int eemop_batchnorm_train_bwd(const float* x, const float* y, const float* dy, const float* weight, const float* save_mean,
                                         const float* save_rstd, int n, int c, int hw, int relu, float* dx, float* dweight, float* dbias,
                                         void* stream) {
    EEM_REQUIRE(x && y && dy && weight && save_mean && save_rstd && dx && dweight && dbias, "eemop_batchnorm_train_bwd: NULL argument");
    hipLaunchKernelGGL(bn_train_bwd_kernel, dim3(c), dim3((long)n * hw >= 16384 ? 1024 : 256), 0, (hipStream_t)stream, x, y, dy, weight, save_mean, save_rstd, dx, dweight,
                       dbias, n, c, hw, relu);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

extern "C" int eemop_batchnorm_eval_fwd(const float* x, const float* weight, const float* bias, const float* running_mean,
                                        const float* running_var, int n, int c, int hw, float eps, int relu, float* y, void* stream) {
    EEM_REQUIRE(x && weight && bias && running_mean && running_var && y, "eemop_batchnorm_eval_fwd: NULL argument");
    EEM_REQUIRE(n >= 1 && c >= 1 && hw >= 1, "eemop_batchnorm_eval_fwd: bad sizes");
    hipLaunchKernelGGL(bn_eval_fwd_kernel, dim3(c), dim3((long)n * hw >= 16384 ? 1024 : 256), 0, (hipStream_t)stream, x, weight, bias, running_mean,
                       running_var, y, n, c, hw, eps, relu);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

extern "C" int eemop_batchnorm_eval_bwd(const float* x, const float* y, const float* dy, const float* weight, const float* running_mean,
                                        const float* running_var, int n, int c, int hw, float eps, int relu, float* dx, float* dweight,
                                        float* dbias, void* stream) {
    EEM_REQUIRE(x && y && dy && weight && running_mean && running_var && dx && dweight && dbias, "eemop_batchnorm_eval_bwd: NULL argument");
    hipLaunchKernelGGL(bn_eval_bwd_kernel, dim3(c), dim3((long)n * hw >= 16384 ? 1024 : 256), 0, (hipStream_t)stream, x, y, dy, weight, running_mean,
                       running_var, dx, dweight, dbias, n, c, hw, eps, relu);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

// ================================================================================================ correlation / upsampling
// CorrBlock.__init__ on caller tensors (model/corr.py:13-27,53-60): pyr_l [batch*h*w][h >> l][w >> l]
extern "C" int eemop_corr_pyramid_fwd(const float* fmap1, const float* fmap2, int batch, int c, int h, int w, float* pyr0, float* pyr1,
                                      float* pyr2, float* pyr3, void* stream) {
    EEM_REQUIRE(fmap1 && fmap2 && pyr0 && pyr1 && pyr2 && pyr3, "eemop_corr_pyramid_fwd: NULL argument");
    EEM_REQUIRE((h >> 3) >= 1 && (w >> 3) >= 1, "eemop_corr_pyramid_fwd: a %dx%d map has no fourth pyramid level", h, w);
    hipStream_t st = (hipStream_t)stream;
    const long planes = (long)batch * h * w;
    int rc = er_allpairs_launch(fmap1, fmap2, pyr0, batch, c, h * w, st);
    if (rc != EEM_OK) return rc;
    return er_pool2x3_launch(pyr0, pyr1, pyr2, pyr3, planes, h, w, st);
}

// CorrBlock.__call__ on caller tensors (model/corr.py:29-50): out [batch][324][h][w]
extern "C" int eemop_corr_lookup_fwd(const float* pyr0, const float* pyr1, const float* pyr2, const float* pyr3, const float* coords,
                                     int batch, int h, int w, float* out, void* stream) {
    EEM_REQUIRE(pyr0 && pyr1 && pyr2 && pyr3 && coords && out, "eemop_corr_lookup_fwd: NULL argument");
    LookupArgs la;
    const float* lv[4] = {pyr0, pyr1, pyr2, pyr3};
    int ph = h, pw = w;
    for (int l = 0; l < 4; ++l) { la.pyr[l] = lv[l]; la.ph[l] = ph; la.pw[l] = pw; ph /= 2; pw /= 2; }
    la.coords = coords; la.out = out; la.batch = batch; la.h = h; la.w = w; la.out_ctotal = 324;
    return er_lookup_launch(la, (hipStream_t)stream);
}

// ERAFT.upsample_flow on caller tensors (model/eraft.py:83-94): out [batch][2][8h][8w]; `zeros` [batch][2][h][w] all zero
extern "C" int eemop_convex_upsample_fwd(const float* zeros, const float* flow, const float* mask, int batch, int h, int w, float* out,
                                         void* stream) {
    EEM_REQUIRE(zeros && flow && mask && out, "eemop_convex_upsample_fwd: NULL argument");
    return er_convex_up_launch(zeros, flow, mask, out, batch, h, w, 0, 0, 8 * h, 8 * w, (hipStream_t)stream);
}

// ================================================================================================ EEMFlow+ under autograd
extern "C" int eemop_act_fwd(const float* x, long long n, int kind, float* out, void* stream) {
    EEM_REQUIRE(x && out && n >= 0, "eemop_act_fwd: bad arguments");
    if (n == 0) return EEM_OK;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, x, out, (long)n, kind);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

extern "C" int eemop_shuffle_channels(const float* src, float* dst, int n, int c, int groups, int hw, int inverse, void* stream) {
    EEM_REQUIRE(src && dst && groups >= 1 && c % groups == 0, "eemop_shuffle_channels: bad arguments");
    const long total = (long)n * c * hw;
    hipLaunchKernelGGL(shuffle_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, src, dst, c, groups, hw, total, inverse);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

extern "C" int eemop_scale_flow(const float* x, int n, int hw, float su, float sv, float* out, void* stream) {
    EEM_REQUIRE(x && out, "eemop_scale_flow: NULL argument");
    const long total = (long)n * 2 * hw;
    hipLaunchKernelGGL(scale_flow2_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, out, hw, total, su, sv);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

extern "C" int eemop_pool2_fwd(const float* x, float* out, long long planes, int h, int w, void* stream) {
    EEM_REQUIRE(x && out && h >= 2 && w >= 2, "eemop_pool2_fwd: bad arguments");
    return er_pool2_launch(x, out, (long)planes, h, w, (hipStream_t)stream);
}

extern "C" int eemop_pool2_bwd(const float* dy, float* dx, long long planes, int h, int w, void* stream) {
    EEM_REQUIRE(dy && dx, "eemop_pool2_bwd: NULL argument");
    const long total = (long)planes * h * w;
    hipLaunchKernelGGL(pool2_bwd_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, dy, dx, h, w, total);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

extern "C" int eemop_resize_ac_fwd(const float* in, float* out, int nc, int h, int w, int oh, int ow, void* stream) {
    EEM_REQUIRE(in && out && nc >= 1, "eemop_resize_ac_fwd: bad arguments");
    const long total = (long)nc * oh * ow;
    hipLaunchKernelGGL(resize_ac_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, in, out, h, w, oh, ow, total);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

extern "C" int eemop_resize_ac_bwd(const float* dout, float* dx, int nc, int h, int w, int oh, int ow, void* stream) {
    EEM_REQUIRE(dout && dx && nc >= 1, "eemop_resize_ac_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const char* e = getenv("EEM_RESIZE_BWD_SCATTER");              // read per call: the tests compare both forms in one process
    if (!(e && e[0] == '1')) {
        const long npix = (long)nc * h * w;
        hipLaunchKernelGGL(resize_ac_bwd_gather_kernel, dim3((unsigned)((npix + 3) / 4)), dim3(256), 0, st, dout, dx, h, w, oh, ow, npix);
        EEM_HIP_CHECK(hipGetLastError());
        return EEM_OK;
    }
    EEM_HIP_CHECK(hipMemsetAsync(dx, 0, (size_t)nc * h * w * sizeof(float), st));
    const long total = (long)nc * oh * ow;
    hipLaunchKernelGGL(resize_ac_bwd_kernel, dim3(nblk(total)), dim3(256), 0, st, dout, dx, h, w, oh, ow, total);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

// adjoint of eemflow_local_corr53: dcv [batch][53][h][w] -> df1, df2 [batch][c][h][w]
extern "C" int eemop_local_corr53_bwd(const float* dcv, const float* f1, const float* f2, int batch, int c, int h, int w, float* df1,
                                      float* df2, void* stream) {
    EEM_REQUIRE(dcv && f1 && f2 && df1 && df2, "eemop_local_corr53_bwd: NULL argument");
    static const int kTaps[53] = {0,  2,  4,  6,  8,  10, 12, 14, 16, 18, 20, 21, 22, 23, 24, 26, 28, 29, 30, 31, 32, 33, 34, 36, 38, 39, 40,
                                  41, 42, 44, 46, 47, 48, 49, 50, 51, 52, 54, 56, 57, 58, 59, 60, 62, 64, 66, 68, 70, 72, 74, 76, 78, 80};
    static thread_local int* taps = nullptr;
    static thread_local int taps_dev = -1;
    int dev = 0;
    EEM_HIP_CHECK(hipGetDevice(&dev));
    if (taps == nullptr || taps_dev != dev) {
        EEM_HIP_CHECK(hipMalloc(&taps, sizeof(kTaps)));
        EEM_HIP_CHECK(hipMemcpy(taps, kTaps, sizeof(kTaps), hipMemcpyHostToDevice));
        taps_dev = dev;
    }
    hipStream_t st = (hipStream_t)stream;
    EEM_HIP_CHECK(hipMemsetAsync(df1, 0, (size_t)batch * c * h * w * sizeof(float), st));
    return tr_corr_bwd_launch(dcv, 53, f1, f2, df1, df2, batch, c, h, w, taps, 53, st);
}
