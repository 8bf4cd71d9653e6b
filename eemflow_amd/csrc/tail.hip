// Everything that runs on EEMFlow's 1/64-resolution grid (reference: model/EEMFlow/EEMFlow.py):
//   stage pooling (:144-154), 9x9 local correlation + tap select (:14-23,160), rconv_k and the
//   three decoders with grouped convs + channel shuffle (:37-69,96-102,161-176), out_conv (:104,180)
//   and the bilinear upsample back to the input size (:118-120,181).
// The grid is tiny (12x20 at 1280x720) so these kernels are latency-, not throughput-bound: each
// conv layer is one launch that covers all decoders/groups/samples ("jobs"), spreads 16x16 MFMA
// tiles over the chip and splits K over the four waves of a block.
#include "common.h"
#include <cstring>

namespace {

// ------------------------------------------------------------------------------- pooling
struct PoolArgs {
    PoolJob job[3];
    int first_wave[4];      // prefix of output counts per job
    int njobs, nimg;
};

__global__ __launch_bounds__(256) void pool_kernel(PoolArgs a) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);      // one wave per pooled output value
    if (gw >= a.first_wave[a.njobs]) return;
    int ji = 0;
    while (ji + 1 < a.njobs && gw >= a.first_wave[ji + 1]) ++ji;
    const PoolJob jb = a.job[ji];
    const int oh = jb.h / jb.k, ow = jb.w / jb.k;
    int o = gw - a.first_wave[ji];
    const int ox = o % ow; o /= ow;
    const int oy = o % oh; o /= oh;                          // o = n*c + channel
    const float* src = jb.in + ((size_t)o * jb.h + (size_t)oy * jb.k) * jb.w + (size_t)ox * jb.k;
    const int kk = jb.k * jb.k;
    float s = 0.f;
    for (int i = lane; i < kk; i += 64) {
        const int ry = i / jb.k, rx = i - ry * jb.k;
        s += src[(size_t)ry * jb.w + rx];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) jb.out[((size_t)o * oh + oy) * ow + ox] = s / (float)kk;
}

struct PoolFinArgs {
    PoolFinJob job[3];
    int njobs, nimg, gh, gw;
};

__global__ __launch_bounds__(256) void pool_finalize_kernel(PoolFinArgs a) {
    int idx = blockIdx.x * 256 + threadIdx.x;
    const int g = a.gh * a.gw;
    for (int ji = 0; ji < a.njobs; ++ji) {
        const PoolFinJob jb = a.job[ji];
        const int total = a.nimg * jb.c * g;
        if (idx < total) {
            const int gx = idx % a.gw, gy = (idx / a.gw) % a.gh, nc = idx / g;
            const float* src = jb.partial + ((size_t)nc * jb.prows + (size_t)gy * jb.rows) * jb.pcols + gx;
            float s = 0.f;
            for (int i = 0; i < jb.rows; ++i) s += src[(size_t)i * jb.pcols];
            jb.out[idx] = s / (float)(jb.k * jb.k);
            return;
        }
        idx -= total;
    }
}

// ------------------------------------------------------------------------------- local correlation
struct CorrArgs {
    CorrJob job[3];
    int njobs, batch, h, w, ntaps;
    const int* taps;
};

// Four adjacent lanes share one output (tap, pixel) and split its channels (c = sub, sub+4, ...): the grid is tiny and
// L2-resident, so the kernel is one chain of dependent-latency loads per thread - a quarter of the channels per thread
// with all of them in flight cuts that chain (18 -> ~6 us inside the forward graph); the partial sums meet by DPP.
__global__ __launch_bounds__(256) void corr_kernel(CorrArgs a) {
    const int hw = a.h * a.w;
    const int per_job = a.batch * a.ntaps * hw;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    int idx = gid >> 2;
    const int sub = gid & 3;
    const bool live = idx < per_job * a.njobs;
    if (!live) idx = 0;                                   // keep the quad together for the cross-lane sum
    const int ji = idx / per_job;
    idx -= ji * per_job;
    const CorrJob jb = a.job[ji];
    const int p = idx % hw; idx /= hw;
    const int ti = idx % a.ntaps;
    const int b = idx / a.ntaps;
    const int y = p / a.w, x = p - y * a.w;
    const int tap = a.taps[ti];                 // dy-major index into the 9x9 window
    const int yy = y + tap / 9 - 4, xx = x + tap % 9 - 4;
    float s = 0.f;
    if (live && yy >= 0 && yy < a.h && xx >= 0 && xx < a.w) {
        const float* p1 = jb.f1 + (size_t)b * jb.c * hw + p;
        const float* p2 = jb.f2 + (size_t)b * jb.c * hw + yy * a.w + xx;
#pragma unroll 16
        for (int c = sub; c < jb.c; c += 4) s = fmaf(p1[(size_t)c * hw], p2[(size_t)c * hw], s);
    }
    s = dpp_add<0xB1>(s);                       // quad_perm [1,0,3,2]
    s = dpp_add<0x4E>(s);                       // quad_perm [2,3,0,1]
    if (live && sub == 0) jb.out[((size_t)b * jb.out_ctotal + ti) * hw + p] = s / (float)jb.c;
}

// The same correlation for larger maps (EEMFlow+'s pyramid levels from 45x80 up): a block of two waves owns an 8x16-pixel tile, the
// f2 patch with its 4-pixel halo goes through LDS eight channels at a time (zeros outside the image), every thread keeps its
// pixel's 53 tap sums in registers and reads f1 once per channel - instead of re-reading f1 for every tap and spending four lanes
// per output.  The next chunk's patch and f1 values are in flight (registers) while the current one is consumed, one barrier per
// chunk; the tap list is a compile-time table (the launcher checks the caller's list against it), so every LDS read is one
// ds_read_b32 with an immediate offset from one base register.  What bounds it then is the LDS port: 53 reads per pixel and
// channel, 212 bytes, against 128 bytes per clock and CU - and, with one or two waves per SIMD, the chunk chain itself: 19 us at
// 180x320 and at 90x160 alike (64 channels; 53.6 us with 16x16 tiles and one load - wait - compute chain per chunk, 49 us on the
// four-lanes-per-output kernel above).
__device__ constexpr int kCorrTaps53[53] = {0,  2,  4,  6,  8,  10, 12, 14, 16, 18, 20, 21, 22, 23, 24, 26, 28, 29, 30, 31, 32, 33, 34, 36, 38, 39, 40,
                                            41, 42, 44, 46, 47, 48, 49, 50, 51, 52, 54, 56, 57, 58, 59, 60, 62, 64, 66, 68, 70, 72, 74, 76, 78, 80};

__global__ __launch_bounds__(128) void corr_tiled53_kernel(CorrArgs a, int tiles_x) {
    constexpr int NT = 53, CK = 8, TY = 8, TX = 16, HY = TY + 8, HX = TX + 8, PITCH = 48, PLANE = HY * PITCH, NP = HY * HX / 128;   // pitch 48: a wave's four rows of 16 lanes fall on the four quarters of the 64 banks
    static_assert(HY * HX == NP * 128, "three patch positions per thread and channel");
    __shared__ float tile[2][CK * PLANE];
    __shared__ float f1s[2][CK * 128];
    const int ji = blockIdx.z / a.batch, b = blockIdx.z % a.batch;
    const CorrJob jb = a.job[ji];
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int tid = threadIdx.x;
    const int lx = tid & 15, ly = tid >> 4;
    const int y0 = ty * TY, x0 = tx * TX;
    const int y = y0 + ly, x = x0 + lx;
    const bool inside = y < a.h && x < a.w;
    const int hw = a.h * a.w;
    // both operands as buffer loads: a position outside the image, a pixel outside it and a channel past c carry an offset beyond
    // the descriptor's range and read as zero - no branch, so a chunk's 32 loads per thread are all in flight at once
    const int bytes = jb.c * hw * 4;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(jb.f1) + (size_t)b * jb.c * hw, (short)0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(jb.f2) + (size_t)b * jb.c * hw, (short)0, bytes, 0x00020000);
    constexpr int kOut = 0x40000000;
    const int p1 = inside ? (y * a.w + x) * 4 : kOut;
    // the thread's three positions of the 16 x 24 patch (the same for every channel)
    int gpos[NP], lpos[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int pos = tid + 128 * k;
        const int ry = pos / HX, rx = pos - ry * HX;
        const int gy = y0 - 4 + ry, gx = x0 - 4 + rx;
        gpos[k] = (gy >= 0 && gy < a.h && gx >= 0 && gx < a.w) ? (gy * a.w + gx) * 4 : kOut;
        lpos[k] = ry * PITCH + rx;
    }
    float s[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) s[t] = 0.f;
    float pf[CK][NP], v1[CK];
    auto fetch = [&](int c0) {                           // (the launcher sends channel counts that are not multiples of CK elsewhere)
#pragma unroll
        for (int c = 0; c < CK; ++c) {
            const int co = __builtin_amdgcn_readfirstlane((c0 + c) * hw * 4);      // the channel's offset rides in the scalar operand
#pragma unroll
            for (int k = 0; k < NP; ++k) pf[c][k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r2, gpos[k], co, 0));
            v1[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1, p1, co, 0));
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int c = 0; c < CK; ++c) {
#pragma unroll
            for (int k = 0; k < NP; ++k) tile[buf][c * PLANE + lpos[k]] = pf[c][k];
            f1s[buf][c * 128 + tid] = v1[c];
        }
    };
    fetch(0);
    stage(0);
    __syncthreads();
    const int nch = jb.c / CK;
    for (int ci = 0; ci < nch; ++ci) {
        if (ci + 1 < nch) fetch((ci + 1) * CK);
        const float* base = &tile[ci & 1][ly * PITCH + lx];
        const float* vb = &f1s[ci & 1][tid];
        // two channels per trip (fully unrolled, the scheduler hoists all 424 LDS reads of a chunk: 512 VGPRs and scratch)
#pragma unroll 2
        for (int c = 0; c < CK; ++c) {
            const float v = vb[c * 128];
#pragma unroll
            for (int t = 0; t < NT; ++t) s[t] = fmaf(v, base[c * PLANE + (kCorrTaps53[t] / 9) * PITCH + kCorrTaps53[t] % 9], s[t]);   // window origin = (-4, -4)
        }
        if (ci + 1 < nch) stage((ci + 1) & 1);
        __syncthreads();
    }
    if (!inside) return;
    const float inv = 1.f / (float)jb.c;
    const int p = y * a.w + x;
#pragma unroll
    for (int t = 0; t < NT; ++t) jb.out[((size_t)b * jb.out_ctotal + t) * hw + p] = s[t] * inv;
}

// ------------------------------------------------------------------------------- small-grid conv
// D[cout 16][pixel 16] tiles on v_mfma_f32_16x16x4_f32; pixels are the flattened (y*w+x) index.
// One wave per filter tap (9 waves for 3x3): a wave's K range is its tap x all channel groups, so the
// tap displacement and the bounds test are per-wave constants and ALL of its operand loads (<= MAXCG
// weight fragments + MAXCG input gathers) are issued before the first MFMA - the grid is tiny and
// L2-resident, so the kernel is one memory round trip + <= 25 MFMAs + a 9-way LDS reduction.
//
// Built for what a launch costs BESIDE other frames' encoder kernels (four frames in flight; profiles/r03_tailcost.txt):
//   * the seven decoder launches as empty kernels cost the frame 2.5 us, as 315 sleeping blocks of 3 us each 5 us, as one
//     sleeping wave of 6 us each (42 us of chain, no footprint) 2.7 us - neither the launches nor the chain's length is the price;
//   * the real launches cost ~1.2x their own duration above an empty launch's: a block holds a CU slot an encoder block wants for
//     as long as it lives, and it lives as long as its chain of memory round trips, each several times its idle length there;
//   * 16 KB of straight-line code in front of the sleeping blocks costs 17 us more (instruction fetch is shared by the CU's waves).
// So: ONE batch of scalar loads (launch header + the job, picked by blockIdx.z), ONE batch of operand loads, and small code - both
// operand streams are buffer loads whose range check returns the zeros of padded channel groups, of channels past cin and of
// taps outside the grid (clamps and selects on 64-bit addresses before: 19 KB -> 2.7 KB).  1280x720: the seven launches 38.2 ->
// 27.4 us alone, frame 126.4 -> 122.2 us with four in flight.
template <int KS, int MAXCG, bool GATE>
__global__ __launch_bounds__(64 * KS * KS) void tail_conv_kernel(TailConvLaunch L) {
    constexpr int KK = KS * KS;
    __shared__ f32x4 part[KK][64];
    const int lane = threadIdx.x & 63;
    const int t = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // grid = (pixel tiles x batch, cout tiles, jobs): the job index is a launch register, so the job's fields and the launch
    // header are ONE batch of scalar loads (the kernel's first memory round trip; its second and last before the stores is the
    // operand loads below).  Beside other frames' kernels a round trip is several times its idle length and the block holds its
    // CU slot throughout - the chain of five dependent scalar loads this replaces was most of a decoder launch's price there.
    const TailConvJob jb = L.job[blockIdx.z];
    const int hw = L.h * L.w, lw = L.w, lh = L.h;
    // (all of them wanted in registers here: the compiler otherwise leaves some of the loads behind the first branch)
    asm volatile("" ::"s"(jb.in), "s"(jb.wpk), "s"(jb.bias), "s"(jb.out), "s"(jb.add), "s"(jb.gate), "s"(jb.cin), "s"(jb.cout), "s"(jb.in_cmul),
                 "s"(jb.out_cmul), "s"(jb.act), "s"(lw), "s"(lh));
    const int cot = blockIdx.y;
    const bool live = cot * 16 < jb.cout;               // a launch's jobs may differ in cout: surplus blocks load nothing (range 0) and leave
    const int ptiles = (hw + 15) >> 4;
    const int b = blockIdx.x / ptiles;
    const int j = lane & 15, g = lane >> 4;
    const int p = (blockIdx.x - b * ptiles) * 16 + j;
    const int y = p / lw, x = p - y * lw;
    const int yy = y + (KS == 3 ? t / 3 - 1 : 0), xx = x + (KS == 3 ? t % 3 - 1 : 0);
    const bool valid = p < hw && yy >= 0 && yy < lh && xx >= 0 && xx < lw;
    const int cg = (jb.cin + 3) >> 2;
    const int cmul = jb.in_cmul > 1 ? jb.in_cmul : 1;
    // weights: cg fragments of 64 floats for (cout tile, tap); a fragment past cg reads as zero
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(jb.wpk) + ((size_t)cot * KK + t) * cg * 64, (short)0, live ? cg * 256 : 0, 0x00020000);
    // input: channel c at c * cmul * hw floats from (b, in_coff); a channel past cin and a tap outside the grid read as zero
    const size_t ibase = ((size_t)b * jb.in_ctotal + jb.in_coff) * hw;
    const int ibytes = live ? (((jb.cin - 1) * cmul + 1) * hw) * 4 : 0;
    const __amdgpu_buffer_rsrc_t ir = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(jb.in) + ibase, (short)0, ibytes, 0x00020000);
    __amdgpu_buffer_rsrc_t gr = ir;
    if (GATE) gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(jb.gate) + ibase, (short)0, ibytes, 0x00020000);
    const int cstep = 4 * cmul * hw * 4;                                   // bytes from one channel group to the next
    int voff = valid ? (g * cmul * hw + yy * lw + xx) * 4 : 0x40000000;

    float av[MAXCG], bv[MAXCG];
#pragma unroll
    for (int q = 0; q < MAXCG; ++q) {
        av[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wr, lane * 4 + q * 256, 0, 0));    // offsets in voffset: range-checked
        bv[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ir, voff, 0, 0));
        if (GATE) bv[q] *= __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(gr, voff, 0, 0)) > 0.f ? 1.f : 0.1f;
        voff += cstep;
    }
    // the finishing wave's bias and residual ride in the same round trip
    const int co0 = cot * 16 + g * 4;
    const size_t o0 = ((size_t)b * jb.out_ctotal + co0 * jb.out_cmul + jb.out_coff) * hw + p;
    const size_t ostep = (size_t)jb.out_cmul * hw;
    float bs[4] = {0.f, 0.f, 0.f, 0.f}, ad[4] = {0.f, 0.f, 0.f, 0.f};
    if (t == 0 && p < hw && live) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (co0 + r < jb.cout) {
                if (jb.bias) bs[r] = jb.bias[co0 + r];
                if (jb.add) ad[r] = jb.add[o0 + r * ostep];                // residual at the output's own index (EEMFlow+ decoders)
            }
    }
    __builtin_amdgcn_sched_barrier(0);                  // every load in flight before the first MFMA waits
    if (!live) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < MAXCG; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], acc, 0, 0, 0);

    if (KK > 1) {
        part[t][lane] = acc;
        __syncthreads();
        if (t != 0) return;
        acc = part[0][lane];
#pragma unroll
        for (int k = 1; k < KK; ++k) acc += part[k][lane];          // fixed order: bitwise repeatable
    }
    if (p < hw) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (co0 + r < jb.cout) {
                float v = acc[r] + bs[r];
                if (jb.act) v = v > 0.f ? v : 0.1f * v;
                if (jb.add) v += ad[r];
                jb.out[o0 + r * ostep] = v;
            }
    }
}

// The same convolution for NARROW layers of a BATCHED chain (eemflow_forward_many: ten frames = 150 pixel tiles per decoder): a
// block of the kernel above carries CG <= 5 MFMAs per wave - the grouped 20 -> 20 decoder convs - behind the same two memory round
// trips, so its time is the round trips'.  Here a block takes PT consecutive pixel tiles of its (cout tile, job): the weight
// fragments are loaded once, the PT x CG input gathers leave as one batch, PT x CG MFMAs, PT reductions - a fifth of the blocks for
// the same arithmetic in the same order (bitwise the kernel above: per tile the same fragments meet in the same sequence).  Wider
// layers take it with PT = 2 or 3 (weights once per block, half the blocks' round trips).
template <int CG, int PT>
__global__ __launch_bounds__(576) void tail_conv_multi_kernel(TailConvLaunch L) {
    constexpr int KK = 9;
    __shared__ f32x4 part[KK][PT][64];
    const int lane = threadIdx.x & 63;
    const int t = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const TailConvJob jb = L.job[blockIdx.z];
    const int hw = L.h * L.w, lw = L.w, lh = L.h;
    asm volatile("" ::"s"(jb.in), "s"(jb.wpk), "s"(jb.bias), "s"(jb.out), "s"(jb.add), "s"(jb.cin), "s"(jb.cout), "s"(jb.in_cmul),
                 "s"(jb.out_cmul), "s"(jb.act), "s"(lw), "s"(lh));
    const int cot = blockIdx.y;
    const bool live = cot * 16 < jb.cout;
    const int ptiles = (hw + 15) >> 4, ntile = ptiles * L.batch;
    const int j = lane & 15, g = lane >> 4;
    const int cg = (jb.cin + 3) >> 2;
    const int cmul = jb.in_cmul > 1 ? jb.in_cmul : 1;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(jb.wpk) + ((size_t)cot * KK + t) * cg * 64, (short)0, live ? cg * 256 : 0, 0x00020000);
    // one descriptor over the whole batch of the job's input: a tile's sample is part of the lane's offset
    const long ibytes_l = live ? ((long)(L.batch - 1) * jb.in_ctotal * hw + (long)(jb.in_coff + (jb.cin - 1) * cmul + 1) * hw) * 4 : 0;
    const __amdgpu_buffer_rsrc_t ir = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(jb.in), (short)0, (int)ibytes_l, 0x00020000);
    const int cstep = 4 * cmul * hw * 4;
    float av[CG], bv[PT][CG];
#pragma unroll
    for (int q = 0; q < CG; ++q) av[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wr, lane * 4 + q * 256, 0, 0));
    int pp[PT], bb[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int tix = blockIdx.x * PT + i;
        const int b = tix / ptiles;
        const int p = (tix - b * ptiles) * 16 + j;
        const int y = p / lw, x = p - y * lw;
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        const bool valid = tix < ntile && p < hw && yy >= 0 && yy < lh && xx >= 0 && xx < lw;
        // (one descriptor spans the batch, so a channel past cin must be sent out of range by hand: inside the span it would read the
        // next sample's data)
        int voff = valid ? (int)((((long)b * jb.in_ctotal + jb.in_coff + g * cmul) * hw + yy * lw + xx) * 4) : 0x7f000000;
        pp[i] = p; bb[i] = tix < ntile ? b : -1;
#pragma unroll
        for (int q = 0; q < CG; ++q) {
            bv[i][q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ir, (4 * q + g < jb.cin) ? voff : 0x7f000000, 0, 0));
            voff += valid ? cstep : 0;
        }
    }
    const int co0 = cot * 16 + g * 4;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    if (t < PT && live && jb.bias) {                     // the finishing waves' bias rides in the same round trip
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (co0 + r < jb.cout) bs[r] = jb.bias[co0 + r];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!live) return;
    f32x4 acc[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < CG; ++q) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[i][q], acc[i], 0, 0, 0);
        part[t][i][lane] = acc[i];
    }
    __syncthreads();
    if (t >= PT) return;                                 // wave i finishes tile i (PT <= 9)
    const int i = t;
    f32x4 a = part[0][i][lane];
#pragma unroll
    for (int k = 1; k < KK; ++k) a += part[k][i][lane];  // fixed order: bitwise the single-tile kernel
    // tile i's coordinates again for this wave (pp / bb above are per-iteration registers of the unrolled loop: index them by a
    // compile-time switch)
    int p = 0, b = -1;
#pragma unroll
    for (int q = 0; q < PT; ++q)
        if (q == i) { p = pp[q]; b = bb[q]; }
    if (b < 0 || p >= hw) return;
    const size_t o0 = ((size_t)b * jb.out_ctotal + co0 * jb.out_cmul + jb.out_coff) * hw + p;
    const size_t ostep = (size_t)jb.out_cmul * hw;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (co0 + r < jb.cout) {
            float v = a[r] + bs[r];
            if (jb.act) v = v > 0.f ? v : 0.1f * v;
            if (jb.add) v += jb.add[o0 + r * ostep];
            jb.out[o0 + r * ostep] = v;
        }
}

// ------------------------------------------------------------------------------- bilinear resize
// F.interpolate(mode='bilinear', align_corners=False): src = max(scale*(dst+0.5)-0.5, 0)
__device__ __forceinline__ void src_index(float scale, int dst, int in_size, int& i0, int& i1, float& l1) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

__global__ void io_table_kernel(const void** table, const void* e1, const void* e2, void* out) {
    table[0] = e1; table[1] = e2; table[2] = out;
}
struct IoTriples { const void* p[3 * EEM_MAX_COALESCE]; };
__global__ void io_table_many_kernel(const void** table, IoTriples t, int n) {
    for (int i = threadIdx.x; i < 3 * n; i += 64) table[i] = t.p[i];
}

__global__ __launch_bounds__(256) void upsample_kernel(const float* __restrict__ in, float* __restrict__ out_arg,
                                                       int nc, int h, int w, int oh, int ow, const void* const* io, int io_frames) {
    float* __restrict__ out = io ? (float*)io[2] : out_arg;
    const int xq = ceil_div(ow, 4);
    const long total = (long)nc * oh * xq;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int q = idx % xq;
    const int oy = (idx / xq) % oh;
    const int c = idx / ((long)xq * oh);
    int co = c;                                           // channel index inside `out`
    if (io_frames) {                                      // per-frame flow buffers [1][2][oh][ow] (eemflow_forward_many)
        out = (float*)io[3 * (c >> 1) + 2];
        co = c & 1;
    }
    const float sy = (float)h / (float)oh, sx = (float)w / (float)ow;
    int y0, y1; float ly;
    src_index(sy, oy, h, y0, y1, ly);
    const float* r0 = in + ((size_t)c * h + y0) * w;
    const float* r1 = in + ((size_t)c * h + y1) * w;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ox = q * 4 + i;
        int x0, x1; float lx;
        src_index(sx, ox < ow ? ox : ow - 1, w, x0, x1, lx);
        const float top = (1.f - lx) * r0[x0] + lx * r0[x1];
        const float bot = (1.f - lx) * r1[x0] + lx * r1[x1];
        v[i] = (1.f - ly) * top + ly * bot;
    }
    float* dst = out + ((size_t)co * oh + oy) * ow + q * 4;
    if ((ow & 3) == 0) {
        *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (q * 4 + i < ow) dst[i] = v[i];
    }
}

}  // namespace

// ------------------------------------------------------------------------------- host side
int pool_launch(const PoolJob* jobs, int njobs, int nimg, hipStream_t stream) {
    EEM_REQUIRE(njobs >= 1 && njobs <= 3, "pool_launch: njobs=%d", njobs);
    PoolArgs a;
    a.njobs = njobs;
    a.nimg = nimg;
    int total = 0;
    for (int i = 0; i < njobs; ++i) {
        a.job[i] = jobs[i];
        a.first_wave[i] = total;
        total += nimg * jobs[i].c * (jobs[i].h / jobs[i].k) * (jobs[i].w / jobs[i].k);
    }
    for (int i = njobs; i < 4; ++i) a.first_wave[i] = total;
    if (total == 0) return EEM_OK;
    hipLaunchKernelGGL(pool_kernel, dim3(ceil_div(total, 4)), dim3(256), 0, stream, a);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int pool_finalize_launch(const PoolFinJob* jobs, int njobs, int nimg, int gh, int gw, hipStream_t stream) {
    EEM_REQUIRE(njobs >= 1 && njobs <= 3, "pool_finalize_launch: njobs=%d", njobs);
    PoolFinArgs a;
    a.njobs = njobs; a.nimg = nimg; a.gh = gh; a.gw = gw;
    int total = 0;
    for (int i = 0; i < njobs; ++i) { a.job[i] = jobs[i]; total += nimg * jobs[i].c * gh * gw; }
    if (total == 0) return EEM_OK;
    hipLaunchKernelGGL(pool_finalize_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, a);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int corr_launch(const CorrJob* jobs, int njobs, int batch, int h, int w, const int* taps_dev, int ntaps,
                hipStream_t stream) {
    EEM_REQUIRE(njobs >= 1 && njobs <= 3, "corr_launch: njobs=%d", njobs);
    CorrArgs a;
    for (int i = 0; i < njobs; ++i) a.job[i] = jobs[i];
    a.njobs = njobs; a.batch = batch; a.h = h; a.w = w; a.ntaps = ntaps; a.taps = taps_dev;
    const long total = (long)njobs * batch * ntaps * h * w;
    if (total == 0) return EEM_OK;
    static const bool plain = [] { const char* e = getenv("EEM_CORR_PLAIN"); return e && e[0] == '1'; }();
    bool c8 = true;
    for (int i = 0; i < njobs; ++i) c8 = c8 && jobs[i].c % 8 == 0;
    if (!plain && ntaps == 53 && c8 && (long)h * w >= 8192) {                  // below: the tile's chain of eight chunks (~18 us whatever the map) loses to the direct kernel
        // (ntaps == 53 is EEMFlow's list, EEMFlow.py:14-23: every caller passes a device copy of that table - api.hip, plus_api.hip,
        // ops.hip - and the tiled kernel has it compiled in)
        const int tiles_x = ceil_div(w, 16);
        hipLaunchKernelGGL(corr_tiled53_kernel, dim3(tiles_x * ceil_div(h, 8), 1, njobs * batch), dim3(128), 0, stream, a, tiles_x);
        EEM_HIP_CHECK(hipGetLastError());
        return EEM_OK;
    }
    hipLaunchKernelGGL(corr_kernel, dim3((unsigned)((total * 4 + 255) / 256)), dim3(256), 0, stream, a);   // 4 lanes per output
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

size_t tail_packed_floats(int cin, int cout, int ksize) {
    return (size_t)ceil_div(cout, 16) * ksize * ksize * ceil_div(cin, 4) * 64;
}

void tail_pack_weights(const float* w, int cin, int cout, int ksize, float* packed) {
    const int cg = ceil_div(cin, 4), kk = ksize * ksize, ksteps = kk * cg;
    for (int cot = 0; cot < ceil_div(cout, 16); ++cot)
        for (int s = 0; s < ksteps; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int co = cot * 16 + (lane & 15);
                const int t = s / cg, c = (s % cg) * 4 + (lane >> 4);
                float v = 0.f;
                if (co < cout && c < cin) v = w[((size_t)co * cin + c) * kk + t];
                packed[((size_t)cot * ksteps + s) * 64 + lane] = v;
            }
}

template <int KS, int MAXCG>
static void tail_launch_t(const TailConvLaunch& l, dim3 grid, hipStream_t stream) {
    bool gate = false;
    for (int i = 0; i < l.njobs; ++i) gate |= l.job[i].gate != nullptr;
    for (int i = 0; i < l.njobs && gate; ++i)
        if (!l.job[i].gate) { fprintf(stderr, "tail_conv_launch: gated and plain jobs in one launch\n"); abort(); }
    if (gate) hipLaunchKernelGGL((tail_conv_kernel<KS, MAXCG, true>), grid, dim3(64 * KS * KS), 0, stream, l);
    else hipLaunchKernelGGL((tail_conv_kernel<KS, MAXCG, false>), grid, dim3(64 * KS * KS), 0, stream, l);
}

int tail_conv_launch(const TailConvLaunch& l, hipStream_t stream) {
    EEM_REQUIRE(l.njobs >= 1 && l.njobs <= TAIL_MAX_JOBS, "tail_conv_launch: njobs=%d", l.njobs);
    EEM_REQUIRE(l.ksize == 3 || l.ksize == 1, "tail_conv_launch: ksize=%d", l.ksize);
    int max_cout = 0, max_cg = 0;
    for (int i = 0; i < l.njobs; ++i) {
        max_cout = l.job[i].cout > max_cout ? l.job[i].cout : max_cout;
        const int cg = ceil_div(l.job[i].cin, 4);
        max_cg = cg > max_cg ? cg : max_cg;
    }
    EEM_REQUIRE(max_cg <= 46, "tail_conv_launch: cin > 184 is not built");
    if (max_cg > 25) {
        // EEMFlow+'s dense estimator on the coarse pyramid levels (cdc_utils.py: 128 / 160 / 176 / 184 input channels): 46 fragments and
        // 46 gathers per wave in one round trip - 4 - 5 us where the LDS-tiled kernel and fewout_kernel took 12 - 14 on 23 x 40 cells
        EEM_REQUIRE(l.ksize == 3, "tail_conv_launch: cin > 100 needs a 3x3 layer");
        dim3 gw(ceil_div(l.h * l.w, 16) * l.batch, ceil_div(max_cout, 16), l.njobs);
        EEM_NOTE_GRID(gw.x * gw.y * gw.z, 576);
        tail_launch_t<3, 46>(l, gw, stream);
        EEM_HIP_CHECK(hipGetLastError());
        return EEM_OK;
    }
    // narrow layers of a batched chain: five pixel tiles per block (tail_conv_multi_kernel; EEM_NO_TAIL_MULTI=1: the one-tile kernel)
    // (EEM_TAIL_MULTI_MAXCG: the widest layer, in 4-channel groups, that takes it.  5 = the grouped convs: 19.2 -> 13.4 us per launch of
    // ten frames; the wider layers with two or three tiles per block measured 25.1 -> 23.5 (conv1), 19.9 -> 21.7 (conv5), 3.7 -> 4.7 (conv7))
    static const int multi_maxcg = [] { const char* e = getenv("EEM_TAIL_MULTI_MAXCG"); return e ? atoi(e) : 5; }();
    bool multi_ok = l.ksize == 3 && max_cg <= multi_maxcg && ceil_div(l.h * l.w, 16) * l.batch >= 40;
    for (int i = 0; i < l.njobs && multi_ok; ++i) {
        const TailConvJob& jb = l.job[i];
        const long span = ((long)(l.batch - 1) * jb.in_ctotal + jb.in_coff + (jb.cin - 1) * (jb.in_cmul > 1 ? jb.in_cmul : 1) + 1) * l.h * l.w * 4;
        multi_ok = jb.gate == nullptr && span < 0x7f000000L;
    }
    { const char* e = getenv("EEM_NO_TAIL_MULTI"); if (e && e[0] == '1') multi_ok = false; }
    dim3 grid(ceil_div(l.h * l.w, 16) * l.batch, ceil_div(max_cout, 16), l.njobs);
    EEM_NOTE_GRID(grid.x * grid.y * grid.z, 64 * l.ksize * l.ksize);
    if (l.ksize == 1) {
        if (max_cg <= 2) tail_launch_t<1, 2>(l, grid, stream);
        else tail_launch_t<1, 25>(l, grid, stream);
    } else if (multi_ok) {
        auto go = [&](auto cg_tag, auto pt_tag) {
            constexpr int CG = decltype(cg_tag)::value, PT = decltype(pt_tag)::value;
            dim3 gm(ceil_div((int)grid.x, PT), grid.y, grid.z);
            EEM_NOTE_GRID(gm.x * gm.y * gm.z, 576);
            hipLaunchKernelGGL((tail_conv_multi_kernel<CG, PT>), gm, dim3(576), 0, stream, l);
        };
        using std::integral_constant;
        if (max_cg <= 5) go(integral_constant<int, 5>{}, integral_constant<int, 5>{});
        else if (max_cg <= 8) go(integral_constant<int, 8>{}, integral_constant<int, 3>{});
        else if (max_cg <= 16) go(integral_constant<int, 16>{}, integral_constant<int, 2>{});
        else if (max_cg <= 18) go(integral_constant<int, 18>{}, integral_constant<int, 2>{});
        else go(integral_constant<int, 25>{}, integral_constant<int, 2>{});
    } else if (max_cg <= 5) tail_launch_t<3, 5>(l, grid, stream);
    else if (max_cg <= 8) tail_launch_t<3, 8>(l, grid, stream);
    else if (max_cg <= 16) tail_launch_t<3, 16>(l, grid, stream);
    else if (max_cg <= 18) tail_launch_t<3, 18>(l, grid, stream);
    else tail_launch_t<3, 25>(l, grid, stream);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int upsample_launch(const float* in, float* out, int nc, int h, int w, int oh, int ow, hipStream_t stream, const void* const* io,
                    int io_frames) {
    const long total = (long)nc * oh * ceil_div(ow, 4);
    if (total == 0) return EEM_OK;
    hipLaunchKernelGGL(upsample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, in, out, nc,
                       h, w, oh, ow, io, io_frames);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

#ifdef EEM_DIAG
static __global__ void spin_kernel(long ticks) {                 // wall_clock64: the 100 MHz constant clock
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

static __global__ void spin_code_kernel(long ticks) {            // the same wait behind 16 KB of straight-line code (instruction-cache footprint)
    const long t0 = wall_clock64();
    asm volatile(".rept 4096\n s_nop 0\n .endr\n" ::: "memory");
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

int spin_launch(float us, hipStream_t stream) {
    static const int blocks = [] { const char* e = getenv("EEM_SKIP_SPIN_BLOCKS"); return e ? atoi(e) : 1; }();    // x 576 threads
    static const bool code = [] { const char* e = getenv("EEM_SKIP_SPIN_CODE"); return e && e[0] == '1'; }();
    const long ticks = (long)(us * 100.f);
    const dim3 g(blocks), b(blocks > 1 ? 576 : 64);
    if (code) hipLaunchKernelGGL(spin_code_kernel, g, b, 0, stream, ticks);
    else hipLaunchKernelGGL(spin_kernel, g, b, 0, stream, ticks);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

#endif

int io_table_many_launch(const void** table, int n, const float* const* e1, const float* const* e2, float* const* out, hipStream_t stream) {
    EEM_REQUIRE(n >= 1 && n <= EEM_MAX_COALESCE, "io_table_many_launch: n=%d", n);
    IoTriples t;
    for (int i = 0; i < 3 * EEM_MAX_COALESCE; ++i) t.p[i] = nullptr;
    for (int i = 0; i < n; ++i) { t.p[3 * i] = e1[i]; t.p[3 * i + 1] = e2[i]; t.p[3 * i + 2] = out[i]; }
    hipLaunchKernelGGL(io_table_many_kernel, dim3(1), dim3(64), 0, stream, table, t, n);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int io_table_launch(const void** table, const void* e1, const void* e2, void* out, hipStream_t stream) {
    hipLaunchKernelGGL(io_table_kernel, dim3(1), dim3(1), 0, stream, table, e1, e2, out);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}
