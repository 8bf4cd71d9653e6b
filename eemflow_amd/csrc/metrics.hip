// Evaluation metrics on the device: Test.flow_error of the reference
// (test_mvsec.py:291-346) - AEE, %EE<1px, %(EE<3px or EE<10% of |gt|) over the pixels where the ground truth is
// finite and non-zero (and, for the 'sparse' evaluation, where the event-count image is positive; rows >= 190 are
// dropped for the MVSEC 'is_car' crop).  One pass over 5 planes (HBM-bound: 20 B per pixel); per-block partial sums
// in double, combined with double atomics (5 per block).
#include <string.h>

#include "common.h"

namespace {

// VEC = 4: four pixels per thread and step by 16-byte loads (plane size and pointers 16-byte aligned); few, fat blocks because each
// block ends with five f64 atomics on one cache line (~14 ns apiece, serialised at the memory side)
// blockIdx.y = sample: eemflow_flow_error_many scores the samples of a coalesced call by one launch (their pointers in `many`)
struct FlowErrMany { const float* gt[16]; const float* pred[16]; const float* ev[16]; };

template <int VEC>
__global__ __launch_bounds__(1024) void flow_error_kernel(const float* __restrict__ gt, const float* __restrict__ pred,
                                                          const float* __restrict__ ev, int h, int w, int max_row,
                                                          double* __restrict__ out, FlowErrMany many, int nmany) {
    if (nmany > 0) {
        gt = many.gt[blockIdx.y]; pred = many.pred[blockIdx.y]; ev = many.ev[blockIdx.y];
        out += 5 * blockIdx.y;
    }
    const long plane = (long)h * w, npix = (long)min(max_row, h) * w;
    double s_ee = 0, s_gt = 0, n = 0, n1 = 0, n3 = 0;
    auto pixel = [&](float gx, float gy, float px, float py, float e) {
        // numpy: ~isinf(gx) & ~isinf(gy) & (norm > 0); a NaN ground truth passes the reference's mask as well
        const float ng = sqrtf(gx * gx + gy * gy);
        const bool m = !isinf(gx) && !isinf(gy) && (ng > 0.f) && e > 0.f;
        if (!m) return;
        const float dx = gx - px, dy = gy - py;
        const float ee = sqrtf(dx * dx + dy * dy);
        s_ee += ee; s_gt += ng; n += 1.0;
        n1 += ee < 1.0f ? 1.0 : 0.0;
        n3 += (ee < 3.0f || ee < 0.1f * ng) ? 1.0 : 0.0;
    };
    const long step = (long)gridDim.x * blockDim.x * VEC;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC; i < npix; i += step) {
        if (VEC == 4) {
            const f32x4 gx = *reinterpret_cast<const f32x4*>(gt + i), gy = *reinterpret_cast<const f32x4*>(gt + plane + i);
            const f32x4 px = *reinterpret_cast<const f32x4*>(pred + i), py = *reinterpret_cast<const f32x4*>(pred + plane + i);
            const f32x4 e = ev ? *reinterpret_cast<const f32x4*>(ev + i) : f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) pixel(gx[k], gy[k], px[k], py[k], e[k]);
        } else {
            pixel(gt[i], gt[plane + i], pred[i], pred[plane + i], ev ? ev[i] : 1.f);
        }
    }
    __shared__ double red[5][16];
    double v[5] = {s_ee, s_gt, n, n1, n3};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        double s = 0.0;
        for (int q = 0; q < (int)(blockDim.x >> 6); ++q) s += red[threadIdx.x][q];
        atomicAdd(out + threadIdx.x, s);
    }
}

}  // namespace

// out5 (device, 5 doubles): sum EE, sum |gt|, n_points, count(EE < 1), count(EE < 3 or EE < 0.1 |gt|)
extern "C" int eemflow_flow_error(const float* flow_gt, const float* flow_pred, const float* event_img, int h, int w,
                                  int max_row, double* out5, void* stream) {
    EEM_REQUIRE(flow_gt && flow_pred && out5 && h >= 1 && w >= 1 && max_row >= 1, "eemflow_flow_error: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    EEM_HIP_CHECK(hipMemsetAsync(out5, 0, 5 * sizeof(double), st));
    const long npix = (long)(max_row < h ? max_row : h) * w;
    const bool vec = (npix & 3) == 0 && (((long)h * w) & 3) == 0 && ((((uintptr_t)flow_gt | (uintptr_t)flow_pred | (uintptr_t)event_img) & 15) == 0);
    int blocks = (int)((npix + 4095) / 4096);
    if (blocks > 64) blocks = 64;
    FlowErrMany none;
    memset(&none, 0, sizeof(none));
    if (vec) hipLaunchKernelGGL(flow_error_kernel<4>, dim3(blocks), dim3(1024), 0, st, flow_gt, flow_pred, event_img, h, w, max_row, out5, none, 0);
    else hipLaunchKernelGGL(flow_error_kernel<1>, dim3(blocks), dim3(1024), 0, st, flow_gt, flow_pred, event_img, h, w, max_row, out5, none, 0);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

// n samples (1..16) of one image size by ONE launch: flow_gt[i], flow_pred[i] (and event_img[i], or event_img == NULL for the 'dense'
// evaluation) are host arrays of device pointers; out5n (device) receives n x 5 doubles, row i = sample i's five sums - the same values as
// n eemflow_flow_error calls (per sample the same blocks add the same partial sums; the f64 atomics of a row commute only up to
// rounding, as in the single call)
extern "C" int eemflow_flow_error_many(int n, const float* const* flow_gt, const float* const* flow_pred, const float* const* event_img,
                                       int h, int w, int max_row, double* out5n, void* stream) {
    EEM_REQUIRE(n >= 1 && n <= 16 && flow_gt && flow_pred && out5n && h >= 1 && w >= 1 && max_row >= 1, "eemflow_flow_error_many: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    FlowErrMany m;
    memset(&m, 0, sizeof(m));
    uintptr_t bits = 0;
    for (int i = 0; i < n; ++i) {
        EEM_REQUIRE(flow_gt[i] && flow_pred[i], "eemflow_flow_error_many: sample %d has a NULL tensor", i);
        m.gt[i] = flow_gt[i]; m.pred[i] = flow_pred[i]; m.ev[i] = event_img ? event_img[i] : nullptr;
        bits |= (uintptr_t)flow_gt[i] | (uintptr_t)flow_pred[i] | (uintptr_t)m.ev[i];
    }
    EEM_HIP_CHECK(hipMemsetAsync(out5n, 0, (size_t)n * 5 * sizeof(double), st));
    const long npix = (long)(max_row < h ? max_row : h) * w;
    const bool vec = (npix & 3) == 0 && (((long)h * w) & 3) == 0 && (bits & 15) == 0;
    int blocks = (int)((npix + 4095) / 4096);
    if (blocks > 64) blocks = 64;
    if (vec) hipLaunchKernelGGL(flow_error_kernel<4>, dim3(blocks, n), dim3(1024), 0, st, nullptr, nullptr, nullptr, h, w, max_row, out5n, m, n);
    else hipLaunchKernelGGL(flow_error_kernel<1>, dim3(blocks, n), dim3(1024), 0, st, nullptr, nullptr, nullptr, h, w, max_row, out5n, m, n);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}
