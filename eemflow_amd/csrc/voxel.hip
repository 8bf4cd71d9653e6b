// Event voxelization (reference: loader/loader_utils.py:447-537, EventSequenceToVoxelGrid_Pytorch):
// temporal-bilinear voting of N events (t, x, y, p) into a (bins, H, W) fp32 grid, then mean /
// unbiased-std normalisation over the non-zero voxels.
//
// Integer work is bit-exact with the reference: the f64 time scaling is evaluated in the same
// order ((bins-1)*(t-t0))/dT with IEEE mul/div (this file is built with -ffp-contract=off), floor,
// truncating casts and the flat index x + y*W + bin*W*H are int64.  The two votes are fp32 atomic
// adds straight to HBM (memory-side atomics on gfx950; events arrive time-sorted, so neighbouring
// lanes hit unrelated voxels).
#include "common.h"

namespace {

struct VoxStats {
    double sum;
    double ss;
    unsigned long long count;
    unsigned long long pad;
};

__global__ __launch_bounds__(256) void voxel_scatter_kernel(const double* __restrict__ ev, long n, int bins, int h,
                                                            int w, float* __restrict__ grid,
                                                            long long* __restrict__ idx_left,
                                                            long long* __restrict__ idx_right) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double t0 = ev[0];
    const double t1 = ev[(n - 1) * 4];
    double dT = t1 - t0;
    if (dT == 0.0) dT = 1.0;                                       // loader_utils.py:485-486
    const double t = ev[i * 4 + 0];
    const double ts = ((double)(bins - 1) * (t - t0)) / dT;        // :488
    const long long xs = (long long)ev[i * 4 + 1];                 // .long() truncates, :490-491
    const long long ys = (long long)ev[i * 4 + 2];
    float pol = (float)ev[i * 4 + 3];
    if (pol == 0.f) pol = -1.f;                                    // :493
    const double tis = floor(ts);
    const long long tl = (long long)tis;
    const float dts = (float)(ts - tis);
    const float vleft = pol * (1.0f - dts);
    const float vright = pol * dts;
    const long long plane = (long long)w * h;
    const bool okl = (tis < (double)bins) && (tis >= 0.0);         // :502-503
    const bool okr = ((tis + 1.0) < (double)bins) && (tis >= 0.0); // :517-518
    const long long il = xs + ys * w + tl * plane;
    const long long ir = xs + ys * w + (tl + 1) * plane;
    if (okl) atomicAdd(grid + il, vleft);
    if (okr) atomicAdd(grid + ir, vright);
    if (idx_left) idx_left[i] = okl ? il : -1;
    if (idx_right) idx_right[i] = okr ? ir : -1;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// pass 1: count and sum of non-zero voxels; pass 2: sum of squared deviations from the fp32 mean
template <int PASS>
__global__ __launch_bounds__(256) void voxel_stats_kernel(const float* __restrict__ grid, long total,
                                                          VoxStats* __restrict__ st) {
    __shared__ double sh[4];
    __shared__ unsigned long long shc[4];
    float mean = 0.f;
    if (PASS == 2) {
        const unsigned long long c = st->count;
        mean = c ? (float)(st->sum / (double)c) : 0.f;
    }
    double acc = 0.0;
    unsigned long long cnt = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const float v = grid[i];
        if (v != 0.f) {
            if (PASS == 1) { acc += (double)v; ++cnt; }
            else { const double d = (double)v - (double)mean; acc += d * d; }
        }
    }
    acc = wave_sum(acc);
    if (PASS == 1) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sh[wave] = acc; shc[wave] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s = sh[0] + sh[1] + sh[2] + sh[3];
        if (PASS == 1) {
            atomicAdd(&st->sum, s);
            atomicAdd(&st->count, shc[0] + shc[1] + shc[2] + shc[3]);
        } else {
            atomicAdd(&st->ss, s);
        }
    }
}

__global__ __launch_bounds__(256) void voxel_norm_kernel(float* __restrict__ grid, long total,
                                                         const VoxStats* __restrict__ st) {
    const unsigned long long c = st->count;
    if (c == 0) return;                                            // :529
    const float mean = (float)(st->sum / (double)c);
    const float sd = (float)sqrt(st->ss / (double)(c - 1));        // unbiased; NaN when c == 1
    const bool scale = sd > 0.f;                                   // :532 (false for NaN)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const float v = grid[i];
        if (v != 0.f) grid[i] = scale ? (v - mean) / sd : (v - mean);
    }
}

}  // namespace

size_t voxel_scratch_bytes() { return sizeof(VoxStats); }

int voxel_launch(const double* events, int64_t n, int bins, int h, int w, int normalize, float* grid,
                 int64_t* idx_left, int64_t* idx_right, void* scratch, hipStream_t stream) {
    EEM_REQUIRE(n >= 1, "voxelize: need at least one event (the reference indexes events[-1], "
                        "loader_utils.py:476), got n=%ld", (long)n);
    EEM_REQUIRE(bins > 0 && h > 0 && w > 0, "voxelize: bad shape bins=%d h=%d w=%d", bins, h, w);
    const long total = (long)bins * h * w;
    EEM_HIP_CHECK(hipMemsetAsync(grid, 0, total * sizeof(float), stream));
    hipLaunchKernelGGL(voxel_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, events,
                       (long)n, bins, h, w, grid, (long long*)idx_left, (long long*)idx_right);
    EEM_HIP_CHECK(hipGetLastError());
    if (normalize) {
        VoxStats* st = (VoxStats*)scratch;
        EEM_HIP_CHECK(hipMemsetAsync(st, 0, sizeof(VoxStats), stream));
        const unsigned blocks = (unsigned)((total / 4 + 255) / 256 < 2048 ? (total / 4 + 255) / 256 : 2048);
        hipLaunchKernelGGL(voxel_stats_kernel<1>, dim3(blocks ? blocks : 1), dim3(256), 0, stream, grid, total, st);
        hipLaunchKernelGGL(voxel_stats_kernel<2>, dim3(blocks ? blocks : 1), dim3(256), 0, stream, grid, total, st);
        hipLaunchKernelGGL(voxel_norm_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, stream, grid, total, st);
        EEM_HIP_CHECK(hipGetLastError());
    }
    return EEM_OK;
}
