// Event voxelization (reference: loader/loader_utils.py:447-537, EventSequenceToVoxelGrid_Pytorch):
// temporal-bilinear voting of N events (t, x, y, p) into a (bins, H, W) fp32 grid, then mean /
// unbiased-std normalisation over the non-zero voxels.
//
// Integer work is bit-exact with the reference: the f64 time scaling is evaluated in the same
// order ((bins-1)*(t-t0))/dT with IEEE mul/div (this file is built with -ffp-contract=off), floor,
// truncating casts and the flat index x + y*W + bin*W*H are int64.
//
// Events arrive time-sorted, so neighbouring lanes vote into unrelated voxels: two fp32 atomics per event
// straight to HBM run at ~20 G atomics/s whatever the kernel does (0.2 ms for 2e6 events).  The default path
// therefore bins first and adds in LDS.  Up to VOX_MAX_JOBS event sets of one grid shape share each launch (blockIdx.y = job:
// eemflow_voxelize_pair / eemflow_voxelize_many - at 2e5 events per set the launches sit at their fixed costs, so a call per set pays
// them per set):
//   1. vox_bin_kernel     every block of 1024 threads derives the votes of its 1024 x EPT events (EPT = 1, 2, 4, 8 by event count), ranks
//                         them per band of `band_px` consecutive pixels with LDS integer atomics (the return value is the vote's rank
//                         in its run), scans the counts and builds its slab of 12-BYTE records {pixel-in-band | bin << 16, left
//                         vote, right vote} sorted by band IN LDS, from where the slab leaves as whole 16-byte pieces of
//                         consecutive threads, plus a row of run offsets (run_start[blk][band]): no global atomics, no counting
//                         pass; also the optional int64 index outputs (event order);
//   2. vox_band_kernel    one block per band (BT = 256 / 512 / 1024 threads by the band's LDS size; 9216 cells = 72 KB of fp64 by
//                         default -> 1024 threads, two blocks per CU; 57 VGPRs, no scratch).  A run = the band's records of one
//                         binning block; 2^sgs adjacent lanes share a run (1.7 x the mean run length), eight runs per thread in
//                         flight: their table entries as one batch of buffer loads, then their first records as one batch of
//                         12-byte buffer loads (out-of-range lanes read zeros: no branches), the rare longer runs in a clean-up
//                         loop.  The band accumulates in fp64 LDS cells (ds_add_f64: ds_add_f32 runs at 0.33 lanes per clock and CU
//                         on this chip, 8 - 20 x slower, profiles/r04_lds_atomics.txt).  One pass then rounds every cell to fp32
//                         once, stores it (zeros included: no memset of the grid) and leaves the f64 (count, sum, sum of squares)
//                         of the band's non-zero voxels (a thread's few dozen cells as an integer count and fp32 partial sums, widened
//                         once: round 6).  The band is zeroed under the first table round trip (LDS-only block wait);
//   3. normalisation (loader_utils.py:527-535), one of
//        vox_norm_kernel   (normalize = 1) adds the band sums in a fixed order, mean / unbiased sd, rewrites the non-zero voxels;
//        vox_stats_kernel  (normalize = 2, "deferred") writes {mean, sd, scale, any} into the four floats BEHIND the grid and leaves
//                          the grid raw: the first encoder layer normalises as it reads (conv_enc1.hip, NORM) - at 2e5 events per
//                          volume the rewrite is half of a sample's voxelization time (37 -> 18 us per sample, ten per call).
// Optional (EEM_VOX_TWOPASS=<r>, off by default): with fewer than one event per r voxels the band kernel runs twice instead of
// step 3 - a moments-only launch, then a launch that accumulates the bands again and stores them already normalised.  Measured at
// 2e5 events per 4.6 M voxels: no gain.
// HBM traffic: 32 B (event) + 12 B written + 12 B read (record) per event + 4 B per voxel (normalize = 1: + 8 B per voxel).
// An event with x >= W lands in a neighbouring row exactly as the reference's flat index_add_ puts it; votes whose
// flat pixel index x + y*W falls outside the image (the reference raises) are dropped on this path, the index
// outputs still report them.  Grid values: every voxel's votes are summed exactly (fp64) and rounded once - within 2e-5 of the
// reference's fp32 index_add_, whose order is not defined either; the integer indices are bit-exact.
// Grids too large for the LDS band layout (band_px * bins > 19 200 cells, bins > 64) and more than 33.5 M events
// take the direct atomic kernel (one job at a time).
#include <string.h>

#include "common.h"

#include <algorithm>
#include <cstdlib>

namespace {

// ------------------------------------------------------------------------------------------------ per-event arithmetic
struct VoxVote {
    long long il, ir;        // flat int64 indices of the two votes (valid where okl / okr)
    long long pix;           // x + y*W
    int tl;                  // left bin
    float vl, vr;
    bool okl, okr;
};

struct VoxTime {
    double t0, dT;
};

__device__ __forceinline__ VoxTime vox_time(const double* __restrict__ ev, long n) {
    VoxTime tm;
    tm.t0 = ev[0];
    tm.dT = ev[(n - 1) * 4] - tm.t0;
    if (tm.dT == 0.0) tm.dT = 1.0;                                 // loader_utils.py:485-486
    return tm;
}

__device__ __forceinline__ VoxVote vox_vote(double t, double x, double y, double p, const VoxTime& tm, int bins, int h,
                                            int w) {
    VoxVote v;
    const double ts = ((double)(bins - 1) * (t - tm.t0)) / tm.dT;  // :488
    const long long xs = (long long)x;                             // .long() truncates, :490-491
    const long long ys = (long long)y;
    float pol = (float)p;
    if (pol == 0.f) pol = -1.f;                                    // :493
    const double tis = floor(ts);
    const long long tl = (long long)tis;
    const float dts = (float)(ts - tis);
    v.vl = pol * (1.0f - dts);
    v.vr = pol * dts;
    const long long plane = (long long)w * h;
    v.okl = (tis < (double)bins) && (tis >= 0.0);                  // :502-503
    v.okr = ((tis + 1.0) < (double)bins) && (tis >= 0.0);          // :517-518
    v.pix = xs + ys * w;
    v.il = v.pix + tl * plane;
    v.ir = v.pix + (tl + 1) * plane;
    v.tl = (int)tl;
    return v;
}

struct VoxSums {             // (count, sum, sum of squares) of one block's non-zero voxels
    double count, sum, sumsq, pad;
};

// ------------------------------------------------------------------------------------------------ direct atomic path
__global__ __launch_bounds__(256) void voxel_scatter_kernel(const double* __restrict__ ev, long n, int bins, int h,
                                                            int w, float* __restrict__ grid,
                                                            long long* __restrict__ idx_left,
                                                            long long* __restrict__ idx_right) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const VoxTime tm = vox_time(ev, n);
    const VoxVote v = vox_vote(ev[i * 4 + 0], ev[i * 4 + 1], ev[i * 4 + 2], ev[i * 4 + 3], tm, bins, h, w);
    // the reference adds at the FLAT index (index_add_ on the flattened grid, loader_utils.py:505-521): a pixel index outside
    // [0, h*w) lands in a neighbouring bin's plane as long as the flat index stays inside the grid; outside the grid the
    // reference raises - those votes are dropped here (writing past the allocation is not an option)
    const long long total = (long long)bins * h * w;
    if (v.okl && v.il >= 0 && v.il < total) atomicAdd(grid + v.il, v.vl);
    if (v.okr && v.ir >= 0 && v.ir < total) atomicAdd(grid + v.ir, v.vr);
    if (idx_left) idx_left[i] = v.okl ? v.il : -1;
    if (idx_right) idx_right[i] = v.okr ? v.ir : -1;
}

// ------------------------------------------------------------------------------------------------ block helpers
constexpr int VT = 1024;                 // threads per block of the binned path (also the maximum number of bands)

// sum over the wave, in every lane: the four steps inside a 16-lane row are DPP moves (VALU), only the two across rows go through
// ds_bpermute - with all six as __shfl_xor a 3-value reduction was 36 dependent LDS round trips (~1.5 us in front of every
// moments store and every normalisation)
template <int CTRL>
__device__ __forceinline__ double dpp_add64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return v + __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double wave_sum(double v) {
    v = dpp_add64<0xB1>(v);                     // quad_perm [1,0,3,2]
    v = dpp_add64<0x4E>(v);                     // quad_perm [2,3,0,1]
    v = dpp_add64<0x141>(v);                    // row_half_mirror
    v = dpp_add64<0x140>(v);                    // row_mirror
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// exclusive prefix sum over the block's NT threads (sh: NT / 64 words)
template <int NT = 1024>
__device__ __forceinline__ unsigned block_exscan(unsigned v, unsigned* sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned t = __shfl_up(x, d);
        if (lane >= d) x += t;
    }
    if (lane == 63) sh[wave] = x;
    __syncthreads();
    unsigned before = 0;
#pragma unroll
    for (int k = 0; k < NT / 64; ++k) before += k < wave ? sh[k] : 0u;
    return before + x - v;
}

struct VoxPlan {
    int band_px;             // pixels per band (multiple of 4 unless it covers the whole image)
    int nb;                  // bands
    unsigned hw;             // H * W
};
struct VoxSums;
// One voxelization of a launch: blockIdx.y picks the job, so the two event sets of a sample (loader/HREM.py:226-232) share each of
// the three launches (same grid shape and plan, their own events, slabs, run tables, moments and grid)
struct VoxJob {
    const double* ev;
    long n;
    unsigned* run_start;
    unsigned* recs;          // 12-byte records {pixel-in-band | bin << 16, left vote, right vote}, a slab of VT * EPT per binning block
    long long* idx_left;
    long long* idx_right;
    float* grid;
    VoxSums* acc;
    int nblk;                // binning blocks (slabs) of this job
};
struct VoxJobs { VoxJob j[VOX_MAX_JOBS]; };      // (72 bytes each: 2.3 KB of kernel arguments at 32 jobs)

// ------------------------------------------------------------------------------------------------ 1. binning into slabs
// Block `blk` owns the slab recs[blk * E .. (blk + 1) * E) (E = 1024 * EPT events per block): its votes, sorted by band, and
// row `blk` of the run table: run_start[blk][b] = offset of band b's run inside the slab, run_start[blk][nb] = its end.
template <int EPT>
__global__ __launch_bounds__(VT) void vox_bin_kernel(VoxJobs jobs, int bins, int h, int w, VoxPlan pl) {
    const VoxJob& J = jobs.j[blockIdx.y];
    if ((int)blockIdx.x >= J.nblk) return;                         // the shorter event set of a pair
    const double* __restrict__ ev = J.ev;
    const long n = J.n;
    unsigned* __restrict__ run_start = J.run_start;
    long long* __restrict__ idx_left = J.idx_left;
    long long* __restrict__ idx_right = J.idx_right;
    __shared__ unsigned hist[VT];
    __shared__ unsigned lpos[VT];
    __shared__ unsigned sh[VT / 64];
    extern __shared__ __attribute__((aligned(16))) unsigned stage[];   // the slab as it goes to memory: 3 words per record
    const int tid = threadIdx.x;
    hist[tid] = 0;
    __syncthreads();
    const VoxTime tm = vox_time(ev, n);
    const f64x2* e2 = reinterpret_cast<const f64x2*>(ev);
    unsigned key[EPT], band[EPT], rank[EPT];
    float vl[EPT], vr[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const long i = ((long)blockIdx.x * EPT + k) * VT + tid;
        band[k] = 0xffffffffu;
        if (i < n) {
            const f64x2 a = e2[i * 2], b = e2[i * 2 + 1];
            const VoxVote v = vox_vote(a[0], a[1], b[0], b[1], tm, bins, h, w);
            if (idx_left) idx_left[i] = v.okl ? v.il : -1;
            if (idx_right) idx_right[i] = v.okr ? v.ir : -1;
            long long pix = v.pix;
            int tl = v.tl;
            float wl = v.vl, wr_ = v.okr ? v.vr : 0.f;             // a masked right vote stays masked wherever the pair lands
            if (v.okl && (unsigned long long)pix >= (unsigned long long)pl.hw) {
                // flat-index semantics of the reference (see voxel_scatter_kernel): a pixel index outside the plane moves the
                // pair of votes by whole planes; each vote survives while ITS flat index stays inside the grid
                long long q = pix / (long long)pl.hw;
                if (pix - q * (long long)pl.hw < 0) --q;           // floor division
                pix -= q * (long long)pl.hw;
                const long long t2 = (long long)tl + q;
                if (t2 == -1) { tl = 0; wl = wr_; wr_ = 0.f; }     // left vote in front of the grid, right vote in plane 0
                else tl = (t2 >= 0 && t2 < bins) ? (int)t2 : -1;
            }
            if (v.okl && tl >= 0) {
                const unsigned bd = (unsigned)pix / (unsigned)pl.band_px;
                band[k] = bd;
                key[k] = ((unsigned)pix - bd * (unsigned)pl.band_px) | ((unsigned)tl << 16);
                vl[k] = wl;
                vr[k] = wr_;
                rank[k] = atomicAdd(&hist[bd], 1u);                // LDS: the returned count is the rank inside the run
            }
        }
    }
    __syncthreads();
    const unsigned start = block_exscan(hist[tid], sh);            // bands >= nb hold 0: thread nb gets the slab's fill
    lpos[tid] = start;
    if (tid <= pl.nb) run_start[(size_t)blockIdx.x * (pl.nb + 1) + tid] = start;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        if (band[k] != 0xffffffffu) {
            unsigned* r = stage + 3u * (lpos[band[k]] + rank[k]);
            r[0] = key[k];
            r[1] = __float_as_uint(vl[k]);
            r[2] = __float_as_uint(vr[k]);
        }
    }
    __syncthreads();
    // the sorted slab leaves as whole 16-byte pieces of consecutive threads (scattered 12-byte stores ran at half the rate)
    const unsigned words = 3u * lpos[pl.nb];
    u32x4* slab = reinterpret_cast<u32x4*>(J.recs + (size_t)blockIdx.x * (VT * EPT) * 3);
    const u32x4* st4 = reinterpret_cast<const u32x4*>(stage);
    for (unsigned i = tid; i * 4 < words; i += VT) slab[i] = st4[i];   // (the last piece may carry up to three stale words: nobody reads them)
}

// ------------------------------------------------------------------------------------------------ moments
// Every block stores the f64 (count, sum, sum of squares) of its non-zero voxels in its own slot; the normalisation
// kernel's waves each add the slots up again (<= 1023 x 24 B from L2, no barrier).  Measured alternatives: f64 atomics
// into shared accumulators serialise (376 blocks on one line: +16 us, on 64 lines: +5 us), a ticket with the last block
// merging costs ~20 us of serial latency.
__device__ __forceinline__ void vox_store_sums(double c, double s, double q, VoxSums* __restrict__ mine, double* sh3) {
    c = wave_sum(c);
    s = wave_sum(s);
    q = wave_sum(q);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
    if (lane == 0) { sh3[wave * 3] = c; sh3[wave * 3 + 1] = s; sh3[wave * 3 + 2] = q; }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (not __syncthreads: that would wait for the band's stores in flight too)
    if (threadIdx.x == 0) {
        c = s = q = 0.0;
        for (int k = 0; k < nw; ++k) { c += sh3[k * 3]; s += sh3[k * 3 + 1]; q += sh3[k * 3 + 2]; }
        mine->count = c;
        mine->sum = s;
        mine->sumsq = q;
    }
}

// mean / unbiased sd of the non-zero voxels from the per-block slots: slots strided over the threads, wave sums, then every
// thread adds the wave results in the same order (all NT threads of the block call this)
struct VoxNorm {
    float mean, sd;
    bool any, scale;
};

template <int NT = 1024>
__device__ __forceinline__ VoxNorm vox_final(const VoxSums* __restrict__ acc, int nsums, double* sh3) {
    const int tid = threadIdx.x;
    double c = 0.0, sm = 0.0, sq = 0.0;
    for (int k = tid; k < nsums; k += NT) { c += acc[k].count; sm += acc[k].sum; sq += acc[k].sumsq; }
    c = wave_sum(c);
    sm = wave_sum(sm);
    sq = wave_sum(sq);
    if ((tid & 63) == 0) { sh3[(tid >> 6) * 3] = c; sh3[(tid >> 6) * 3 + 1] = sm; sh3[(tid >> 6) * 3 + 2] = sq; }
    __syncthreads();
    c = sm = sq = 0.0;
#pragma unroll
    for (int k = 0; k < NT / 64; ++k) { c += sh3[k * 3]; sm += sh3[k * 3 + 1]; sq += sh3[k * 3 + 2]; }
    VoxNorm nm;
    nm.any = c != 0.0;                                             // :529
    double m2 = nm.any ? sq - sm * sm / c : 0.0;                   // sum (v - mean)^2
    if (m2 < 0.0) m2 = 0.0;                                        // equal voxels: rounding only
    nm.mean = nm.any ? (float)(sm / c) : 0.f;
    nm.sd = (float)sqrt(m2 / (c - 1.0));                           // unbiased; NaN when c == 1
    nm.scale = nm.sd > 0.f;                                        // :532 (false for NaN)
    return nm;
}

// ------------------------------------------------------------------------------------------------ 3. bands in LDS
// what a band block does with its accumulated band: store it (and, with `acc`, leave its moments for vox_norm_kernel); leave the
// moments only; or read every block's moments and store the band normalised (after a VOX_BAND_MOMENTS launch)
enum { VOX_BAND_RAW = 0, VOX_BAND_MOMENTS = 1, VOX_BAND_NORMALISED = 2 };

#ifdef EEM_VOX_STAMPS
// stamps build only (EEM_EXTRA_FLAGS=-DEEM_VOX_STAMPS, tools/vox_stamps.py): s_memtime of wave 0 of the first 1024 band blocks at the phase
// boundaries: [0] start [1] band zeroed [2] table entries in [3] first records in [4] adds done, block through [5] read out [6] end
__device__ unsigned long long g_vox_stamps[1024 * 8];
#define VSTAMP(i) { __builtin_amdgcn_sched_barrier(0); vst[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define VSTAMP(i)
#endif

template <int BT>
__global__ __launch_bounds__(BT, BT / 128) void vox_band_kernel(VoxJobs jobs, int slab, int bins, VoxPlan pl, int vec4, int with_moments, int mode, int sgs) {
    const VoxJob& J = jobs.j[blockIdx.y];
    const unsigned* __restrict__ recs = J.recs;
    const unsigned* __restrict__ run_start = J.run_start;
    const int nblk = J.nblk;
    float* __restrict__ grid = J.grid;
    VoxSums* __restrict__ acc = with_moments ? J.acc : nullptr;
    // [bins][band_px] fp64 cells: the LDS fp32 add runs at a third of a lane per clock and CU on this chip, the fp64 add 8 - 20 times faster
    // (profiles/r04_lds_atomics.txt); the sum of a cell's votes is rounded to fp32 once, when the band leaves
    extern __shared__ __attribute__((aligned(16))) double band[];
    __shared__ double sh3[BT / 64 * 3];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int bpx = pl.band_px;
    const unsigned p0 = (unsigned)b * (unsigned)bpx;
    const int npx = min(bpx, (int)(pl.hw - p0));
    const int nfl = bins * bpx;
#ifdef EEM_VOX_STAMPS
    unsigned long long vst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    VSTAMP(0)
    // A run = this band's votes of one binning block: `len` consecutive 12-byte records.  2^sgs adjacent lanes share a run (sized so
    // that nearly every run is one record per lane): a wave's load then touches a few runs' lines instead of 64 unrelated ones - with a
    // thread per run (and, before, a thread per vote with a binary search for its run) the address unit's one line per clock was the
    // kernel: 24 of 35 us.  Eight runs per thread are in flight: all their table entries, then all their first records, then the adds.
    constexpr int NP = 8;
    typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
    const int sub = tid & ((1 << sgs) - 1);
    const int rpp = BT >> sgs;                                     // runs per pass of the block
    // both streams as buffer loads with 32-bit offsets (a run past the last binning block and a lane past its run's end read zeros:
    // no branches around the loads, all of a phase in flight together)
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(run_start), (short)0, nblk * (pl.nb + 1) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(recs), (short)0, (unsigned)nblk * (unsigned)slab * 12u, 0x00020000);
    constexpr int kOut = 0x7ffffff0;
    for (int g0 = 0; g0 < nblk; g0 += NP * rpp) {
        unsigned st[NP];
        int ln[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int sidx = g0 + q * rpp + (tid >> sgs);
            const int off = sidx < nblk ? (sidx * (pl.nb + 1) + b) * 4 : kOut;
            st[q] = __builtin_amdgcn_raw_buffer_load_b32(trs, off, 0, 0);
            ln[q] = (int)__builtin_amdgcn_raw_buffer_load_b32(trs, off, 4, 0);
        }
        if (g0 == 0) {
            // the band is zeroed UNDER the first table round trip (round 6; in-kernel stamps, tools/vox_stamps.py: zeroing + the block's
            // start-up skew 2 000 cycles, table entries 2 700, first records 1 600, adds 3 700, read-out 4 800 of a block's 14 800): an
            // LDS-only block wait - __syncthreads would drain the table loads first
            for (int i = tid * 2; i < nfl; i += BT * 2) *reinterpret_cast<f64x2*>(band + i) = f64x2{0.0, 0.0};
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            VSTAMP(1)
        }
        u32x3 r0[NP];
#ifdef EEM_VOX_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        VSTAMP(2)
#endif
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int sidx = g0 + q * rpp + (tid >> sgs);
            ln[q] -= (int)st[q];
            st[q] = ((unsigned)sidx * (unsigned)slab + st[q] + (unsigned)sub) * 12u;     // byte offset of this lane's first record
            r0[q] = __builtin_amdgcn_raw_buffer_load_b96(rrs, sub < ln[q] ? (int)st[q] : kOut, 0, 0);
        }
#ifdef EEM_VOX_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        VSTAMP(3)
#endif
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (sub < ln[q]) {
                const int tl = (int)(r0[q][0] >> 16);
                double* cell = band + tl * bpx + (int)(r0[q][0] & 0xffffu);
                atomicAdd(cell, (double)__uint_as_float(r0[q][1]));
                if (tl + 1 < bins) atomicAdd(cell + bpx, (double)__uint_as_float(r0[q][2]));
            }
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {                             // the rare longer runs
            for (int k = sub + (1 << sgs); k < ln[q]; k += 1 << sgs) {
                const u32x3 r = __builtin_amdgcn_raw_buffer_load_b96(rrs, (int)(st[q] + (unsigned)(k - sub) * 12u), 0, 0);
                const int tl = (int)(r[0] >> 16);
                double* cell = band + tl * bpx + (int)(r[0] & 0xffffu);
                atomicAdd(cell, (double)__uint_as_float(r[1]));
                if (tl + 1 < bins) atomicAdd(cell + bpx, (double)__uint_as_float(r[2]));
            }
        }
    }
    __syncthreads();
    VSTAMP(4)
    float mean = 0.f, sd = 1.f;
    bool scale = false, shift = false;
    if (mode == VOX_BAND_NORMALISED) {
        const VoxNorm nm = vox_final<BT>(acc, pl.nb, sh3);
        mean = nm.mean; sd = nm.sd; scale = nm.scale; shift = nm.any;
    }
    auto fin = [&](float v) { return (shift && v != 0.f) ? (scale ? (v - mean) / sd : (v - mean)) : v; };
    // one pass over the band: every cell is rounded to fp32 once, counted into the moments and stored
    const bool want_sums = acc && mode != VOX_BAND_NORMALISED;
    const bool store = mode != VOX_BAND_MOMENTS;
    // a thread's share of the band's moments (at most a few dozen cells) in fp32 and an integer count, widened once below: zeros add
    // nothing, so no branch - the fp64 form (compare, branch, convert, two adds and an fma per cell, 9 216 cells per band) made the
    // read-out the longest phase of a band block (stamps: 4 800 of 14 800 cycles, vector-pipe bound: fp64 runs at half rate)
    int cnz = 0;
    float sm32 = 0.f, sq32 = 0.f;
    auto note = [&](float v) { cnz += v != 0.f ? 1 : 0; sm32 += v; sq32 = __builtin_fmaf(v, v, sq32); };
    for (int bin = 0; bin < bins; ++bin) {
        const double* src = band + bin * bpx;
        float* dst = grid + (size_t)bin * pl.hw + p0;
        if (vec4) {
            for (int i = tid * 4; i < npx; i += BT * 4) {
                const f64x2 lo = *reinterpret_cast<const f64x2*>(src + i), hi = *reinterpret_cast<const f64x2*>(src + i + 2);
                f32x4 v = {(float)lo[0], (float)lo[1], (float)hi[0], (float)hi[1]};
#pragma unroll
                for (int q = 0; q < 4; ++q) { note(v[q]); v[q] = fin(v[q]); }
                if (store) *reinterpret_cast<f32x4*>(dst + i) = v;
            }
        } else {
            for (int i = tid; i < npx; i += BT) {
                const float v = (float)src[i];
                note(v);
                if (store) dst[i] = fin(v);
            }
        }
    }
    VSTAMP(5)
    if (want_sums) vox_store_sums((double)cnz, (double)sm32, (double)sq32, acc + b, sh3);
#ifdef EEM_VOX_STAMPS
    VSTAMP(6)
    if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 1024)
        for (int i = 0; i < 8; ++i) g_vox_stamps[blockIdx.x * 8 + i] = vst[i];
#endif
}

// ------------------------------------------------------------------------------------------------ 4. normalisation
__global__ __launch_bounds__(VT) void vox_norm_kernel(VoxJobs jobs, long total, int nsums) {
    float* __restrict__ grid = jobs.j[blockIdx.y].grid;
    const VoxSums* __restrict__ acc = jobs.j[blockIdx.y].acc;
    __shared__ double sh3[VT / 64 * 3];
    const int tid = threadIdx.x;
    // the first NPRE 16-byte pieces of this thread are requested before the sums are known (5 cover 1280x720x5 and
    // 640x480x15 entirely with 256 blocks)
    constexpr int NPRE = 5;
    const bool vec = (total & 3) == 0 && ((uintptr_t)grid & 15) == 0;
    f32x4* g4 = reinterpret_cast<f32x4*>(grid);
    const long n4 = total / 4, stride = (long)gridDim.x * VT, first = (long)blockIdx.x * VT + tid;
    f32x4 pre[NPRE];
    if (vec) {
#pragma unroll
        for (int k = 0; k < NPRE; ++k)
            if (first + k * stride < n4) pre[k] = g4[first + k * stride];
    }
    const VoxNorm nm = vox_final(acc, nsums, sh3);
    if (!nm.any) return;                                           // :529
    const float mean = nm.mean, sd = nm.sd;
    const bool scale = nm.scale;
    if (vec) {
        auto apply = [&](f32x4 v, long i) {
            bool any = false;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (v[q] != 0.f) { v[q] = scale ? (v[q] - mean) / sd : (v[q] - mean); any = true; }
            if (any) g4[i] = v;
        };
#pragma unroll
        for (int k = 0; k < NPRE; ++k)
            if (first + k * stride < n4) apply(pre[k], first + k * stride);
        for (long i = first + NPRE * stride; i < n4; i += stride) apply(g4[i], i);
    } else {
        for (long i = (long)blockIdx.x * VT + tid; i < total; i += (long)gridDim.x * VT) {
            const float v = grid[i];
            if (v != 0.f) grid[i] = scale ? (v - mean) / sd : (v - mean);
        }
    }
}

// ------------------------------------------------------------------------------------------------ 4b. normalisation left to the consumer
// normalize == 2 ("deferred"): the grid stays RAW and the four floats behind it - grid[total .. total + 4) - receive the record
// {mean, sd, scale ? 1 : 0, any ? 1 : 0} of loader_utils.py:527-535; the first encoder layer applies (v - mean) / sd to the non-zero voxels
// as it reads them (conv_enc1.hip), so the 2 x 18.4 MB read-modify-write of vox_norm_kernel never happens.  One block per job.
__global__ __launch_bounds__(VT) void vox_stats_kernel(VoxJobs jobs, long total, int nsums) {
    __shared__ double sh3[VT / 64 * 3];
    const VoxNorm nm = vox_final(jobs.j[blockIdx.x].acc, nsums, sh3);
    if (threadIdx.x == 0) {
        float* rec = jobs.j[blockIdx.x].grid + total;
        rec[0] = nm.mean; rec[1] = nm.scale ? nm.sd : 1.f; rec[2] = nm.scale ? 1.f : 0.f; rec[3] = nm.any ? 1.f : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ direct path: moments
__global__ __launch_bounds__(VT) void vox_moments_kernel(const float* __restrict__ grid, long total, VoxSums* __restrict__ acc) {
    __shared__ double sh3[VT / 64 * 3];
    double c = 0.0, sm = 0.0, q = 0.0;
    for (long i = (long)blockIdx.x * VT + threadIdx.x; i < total; i += (long)gridDim.x * VT) {
        const float v = grid[i];
        if (v != 0.f) { c += 1.0; sm += (double)v; q += (double)v * (double)v; }
    }
    vox_store_sums(c, sm, q, acc + blockIdx.x, sh3);
}

struct VoxScratch {
    VoxSums acc[VT];
};

// blocks of the binning kernel for n events at EPT events per thread; the run table has (nb + 1) <= 1024 words per block
inline long vox_blocks(long n, int ept) { return (n + (long)VT * ept - 1) / ((long)VT * ept); }
inline int vox_ept(long n) {
    int ept = 1;
    while (ept < 8 && vox_blocks(n, ept) > 512) ept *= 2;
    return ept;
}
constexpr long VOX_MAX_BLOCKS = 4096;                              // 33.5 M events at 8 per thread; beyond: direct kernel

template <int EPT>
void launch_bin(const VoxJobs& jobs, int njobs, long nblk_max, int bins, int h, int w, const VoxPlan& pl, hipStream_t stream) {
    constexpr int stage_bytes = VT * EPT * 12;
    if (stage_bytes > 48 * 1024) {
        static bool raised = false;
        if (!raised) {
            (void)hipFuncSetAttribute((const void*)vox_bin_kernel<EPT>, hipFuncAttributeMaxDynamicSharedMemorySize, stage_bytes);
            raised = true;
        }
    }
    hipLaunchKernelGGL((vox_bin_kernel<EPT>), dim3((unsigned)nblk_max, njobs), dim3(VT), stage_bytes, stream, jobs, bins, h, w, pl);
}

bool make_plan(int64_t n, int bins, int h, int w, const double* events, VoxPlan* pl, int* lds_bytes) {
    const long hw = (long)h * w;
    if (bins > 64 || vox_blocks(n, 8) > VOX_MAX_BLOCKS || hw >= (1L << 31) || ((uintptr_t)events & 15)) return false;
    const char* e = getenv("EEM_VOX_DIRECT");
    if (e && e[0] == '1') return false;
    // 9216 fp64 cells (72 KB) per band by default -> vox_band_kernel<1024>, two blocks per CU; EEM_VOX_BAND_FLOATS=<cells> picks another
    // band size (<= 3072 cells: 256-thread blocks, <= 5120: 512) - 2304 .. 18432 cells measure the same within 3 % at 2e5 and 2e6 events
    static const long band_floats = [] { const char* b = getenv("EEM_VOX_BAND_FLOATS"); const long v = b ? atol(b) : 0; return v >= 256 ? v : 9216L; }();
    long band_px = (band_floats / bins) & ~3L;
    if (band_px < 64) band_px = 64;
    const long spread = ((hw + 511) / 512 + 3) & ~3L;              // small images: still a few hundred bands
    if (band_px > spread) band_px = spread < 64 ? 64 : spread;
    if ((hw + band_px - 1) / band_px > VT - 1) band_px = ((hw + VT - 2) / (VT - 1) + 3) & ~3L;   // thread nb holds the record count
    if (band_px * bins > 19200 || band_px > 65535) return false;   // 150 KiB of LDS (fp64 cells) beside the static tables, 16-bit pixel-in-band
    pl->band_px = (int)band_px;
    pl->nb = (int)((hw + band_px - 1) / band_px);
    pl->hw = (unsigned)hw;
    *lds_bytes = (int)(band_px * bins * 8);
    return true;
}

}  // namespace

#ifdef EEM_VOX_STAMPS
extern "C" __attribute__((visibility("default"))) int eemflow_debug_read_vox_stamps(unsigned long long* dst, size_t n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_vox_stamps), n * sizeof(unsigned long long)) == hipSuccess ? 0 : 2;
}
#endif


// sums + one 16-byte record slot per event (slabs are whole blocks: round up) + the run table
size_t voxel_scratch_bytes(int64_t n) {
    const long m = n > 0 ? (long)n : 0;
    const long slots = (m + 8191) / 8192 * 8192 + 8192;
    const long blocks = std::max(512L, std::min(VOX_MAX_BLOCKS, vox_blocks(m, 8) + 1));
    return sizeof(VoxScratch) + (size_t)slots * 16 + (size_t)blocks * VT * sizeof(unsigned);
}

// njobs (1 or 2) voxelizations of the same grid shape in one launch sequence; scratch[k] >= voxel_scratch_bytes(n[k]) each
int voxel_launch_jobs(int njobs, const double* const* events, const int64_t* n, int bins, int h, int w, int normalize, float* const* grid,
                      int64_t* const* idx_left, int64_t* const* idx_right, void* const* scratch, hipStream_t stream) {
    EEM_REQUIRE(njobs >= 1 && njobs <= VOX_MAX_JOBS, "voxelize: %d jobs (1..%d per launch sequence)", njobs, VOX_MAX_JOBS);
    for (int k = 0; k < njobs; ++k)
        EEM_REQUIRE(n[k] >= 1, "voxelize: need at least one event (the reference indexes events[-1], "
                               "loader_utils.py:476), got n=%ld", (long)n[k]);
    EEM_REQUIRE(bins > 0 && h > 0 && w > 0, "voxelize: bad shape bins=%d h=%d w=%d", bins, h, w);
    EEM_REQUIRE(normalize >= 0 && normalize <= 2, "voxelize: normalize is 0 (raw), 1 (normalised grid) or 2 (raw grid + record); got %d", normalize);
    const long total = (long)bins * h * w;
    VoxPlan pl;
    int lds = 0;
    int nsums = 0;
    int64_t nmax = n[0];
    for (int k = 1; k < njobs; ++k) nmax = n[k] > nmax ? n[k] : nmax;
    bool planned = true;
    for (int k = 0; k < njobs; ++k) planned = planned && make_plan(n[k], bins, h, w, events[k], &pl, &lds);
    if (!planned && njobs >= 2) {                                      // the direct kernel has no multi-job form: one after the other
        for (int k = 0; k < njobs; ++k) {
            const int rc = voxel_launch_jobs(1, events + k, n + k, bins, h, w, normalize, grid + k, idx_left + k, idx_right + k, scratch + k, stream);
            if (rc != EEM_OK) return rc;
        }
        return EEM_OK;
    }
    VoxJobs jobs;
    memset(&jobs, 0, sizeof(jobs));
    if (planned) {
        static const int ept_env = [] { const char* e = getenv("EEM_VOX_EPT"); return e ? atoi(e) : 0; }();
        int ept = ept_env;
        if ((ept != 1 && ept != 2 && ept != 4 && ept != 8) || vox_blocks((long)nmax, ept) > std::max(512L, vox_blocks((long)nmax, 8) + 1))
            ept = vox_ept((long)nmax);                                 // the run table is sized for these block counts
        // one EPT for both jobs (the longer set's): the shorter one's slabs are the same size, it just fills fewer of them
        long nblk_max = 0;
        for (int k = 0; k < njobs; ++k) {
            VoxScratch* sc = (VoxScratch*)scratch[k];
            const long slots = ((long)n[k] + 8191) / 8192 * 8192 + 8192;
            VoxJob& J = jobs.j[k];
            J.ev = events[k]; J.n = (long)n[k];
            J.recs = reinterpret_cast<unsigned*>(sc + 1);
            J.run_start = J.recs + slots * 3;                          // (the scratch is sized at 16 bytes per slot)
            J.idx_left = (long long*)idx_left[k]; J.idx_right = (long long*)idx_right[k];
            J.grid = grid[k]; J.acc = sc->acc;
            J.nblk = (int)vox_blocks((long)n[k], ept);
            EEM_REQUIRE((long)J.nblk * VT * ept <= slots, "voxelize: slab budget (n=%ld, ept=%d)", (long)n[k], ept);
            nblk_max = std::max(nblk_max, (long)J.nblk);
        }
        switch (ept) {
            case 1: launch_bin<1>(jobs, njobs, nblk_max, bins, h, w, pl, stream); break;
            case 2: launch_bin<2>(jobs, njobs, nblk_max, bins, h, w, pl, stream); break;
            case 4: launch_bin<4>(jobs, njobs, nblk_max, bins, h, w, pl, stream); break;
            default: launch_bin<8>(jobs, njobs, nblk_max, bins, h, w, pl, stream); break;
        }
        if (lds > 64 * 1024 - 12 * 1024) {
            static thread_local int raised = 0;
            if (raised < lds) {
                EEM_HIP_CHECK(hipFuncSetAttribute((const void*)vox_band_kernel<VT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                raised = lds;
            }
        }
        bool aligned = true;
        for (int k = 0; k < njobs; ++k) aligned = aligned && ((uintptr_t)grid[k] & 15) == 0;
        const int vec4 = (pl.band_px % 4 == 0 && pl.hw % 4 == 0 && aligned) ? 1 : 0;
        // Few events per voxel: accumulating the bands twice (16 B per vote from L2 / HBM each time) is cheaper than the normalisation's
        // read + write of the whole grid - a moments-only launch, then a launch that stores the bands already normalised
        const char* tp = getenv("EEM_VOX_TWOPASS");                  // read per call: the tests run both forms in one process
        const long two_pass_ratio = tp ? atol(tp) : 0L;
        // lanes per run: the smallest power of two >= EEM_VOX_RUN_LANES_X10 / 10 (default 4.0 for sparse sets, else 1.7 - the rule through round 5) x the mean run length
        // (slab events / bands), 4 .. 64: a run longer than its lanes costs its block a dependent round trip in the clean-up loop, and with
        // 196 runs of mean length 2 per band and four lanes each nearly every block had one (stamps: 3 700 cycles in the adds)
        // (sparse sets only - mean run below 4 records: at 2e6 events per grid the wider runs idle more lanes than they save, 53.0 against
        // 50.6 us per sample)
        static const long lanes_env = [] { const char* e = getenv("EEM_VOX_RUN_LANES_X10"); const long v = e ? atol(e) : 0; return v >= 10 ? v : 0L; }();
        const long lanes_x10 = lanes_env ? lanes_env : ((long)VT * ept < 4L * pl.nb ? 40L : 17L);
        int sgs = 2;
        while (sgs < 6 && (1 << sgs) * 10L * pl.nb < lanes_x10 * VT * ept) ++sgs;
        auto band = [&](int with_moments, int mode) {
            if (lds <= 24 * 1024)
                hipLaunchKernelGGL(vox_band_kernel<256>, dim3(pl.nb, njobs), dim3(256), lds, stream, jobs, VT * ept, bins, pl, vec4, with_moments, mode, sgs);
            else if (lds <= 40 * 1024)
                hipLaunchKernelGGL(vox_band_kernel<512>, dim3(pl.nb, njobs), dim3(512), lds, stream, jobs, VT * ept, bins, pl, vec4, with_moments, mode, sgs);
            else
                hipLaunchKernelGGL(vox_band_kernel<VT>, dim3(pl.nb, njobs), dim3(VT), lds, stream, jobs, VT * ept, bins, pl, vec4, with_moments, mode, sgs);
        };
        if (normalize == 1 && two_pass_ratio > 0 && (long)nmax * two_pass_ratio <= total) {
            band(1, (int)VOX_BAND_MOMENTS);
            band(1, (int)VOX_BAND_NORMALISED);
            EEM_HIP_CHECK(hipGetLastError());
            return EEM_OK;
        }
        band(normalize ? 1 : 0, (int)VOX_BAND_RAW);
        nsums = pl.nb;
    } else {
        VoxScratch* sc = (VoxScratch*)scratch[0];
        jobs.j[0].grid = grid[0]; jobs.j[0].acc = sc->acc;
        EEM_HIP_CHECK(hipMemsetAsync(grid[0], 0, total * sizeof(float), stream));
        hipLaunchKernelGGL(voxel_scatter_kernel, dim3((unsigned)((n[0] + 255) / 256)), dim3(256), 0, stream, events[0],
                           (long)n[0], bins, h, w, grid[0], (long long*)idx_left[0], (long long*)idx_right[0]);
        if (normalize) {
            nsums = (int)((total + 4095) / 4096 < 512 ? (total + 4095) / 4096 : 512);
            hipLaunchKernelGGL(vox_moments_kernel, dim3(nsums), dim3(VT), 0, stream, grid[0], total, sc->acc);
        }
    }
    EEM_HIP_CHECK(hipGetLastError());
    if (normalize == 2) {                                              // deferred: the record behind each grid, no pass over the grids
        hipLaunchKernelGGL(vox_stats_kernel, dim3(njobs), dim3(VT), 0, stream, jobs, total, nsums);
        EEM_HIP_CHECK(hipGetLastError());
        return EEM_OK;
    }
    if (normalize) {
        long blocks = (total / 4 + VT - 1) / VT;
        blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);   // one block per CU: the sums are re-added once per block
        if (njobs >= 2 && blocks > 256 / njobs) blocks = std::max(16L, 256L / njobs);     // the jobs share the chip
        hipLaunchKernelGGL(vox_norm_kernel, dim3((unsigned)blocks, njobs), dim3(VT), 0, stream, jobs, total, nsums);
        EEM_HIP_CHECK(hipGetLastError());
    }
    return EEM_OK;
}

int voxel_launch(const double* events, int64_t n, int bins, int h, int w, int normalize, float* grid,
                 int64_t* idx_left, int64_t* idx_right, void* scratch, hipStream_t stream) {
    return voxel_launch_jobs(1, &events, &n, bins, h, w, normalize, &grid, &idx_left, &idx_right, &scratch, stream);
}
