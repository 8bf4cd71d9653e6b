// Winograd F(2x2,3x3) for EEMFlow+'s mid-size 3x3 stride-1 layers at the fine pyramid levels (EEMFlow+.py:38-71 decoders,
// cdc_utils.py:105-151 dense mask estimator, the rconv layers): inputs of 32 .. 256 channels that are a channel RANGE of a wider
// tensor (the dense estimator's growing concatenation, the decoder's 96-channel buffer, a group of a grouped conv), outputs in
// slices of 32 channels written with a channel stride (channel_shuffle is an output stride of `groups`).
//
// Same algebra and fragment layout as conv_wino32.hip (v_mfma_f32_32x32x2_f32: M = 32 couts, N = 32 2x2-pixel tiles, K = 2 cin;
// the 16 Winograd positions split by ROWS over the four waves of a team, V = B^T d B computed in the lane that feeds it to the
// MFMA, the four waves exchange u_xi = M_xi A through LDS and each finishes 8 cout rows), re-cut around the input depth:
//   * the input is consumed in chunks of 32 channels: the chunk's haloed tile [32][6][72] goes L2/HBM -> LDS by 16-byte LDS-DMA,
//     double-buffered; the accumulators live across the chunks of a tile and the output transform runs after the last one;
//   * a chunk's weights (64 KB for the four waves of a team) cannot stay in registers across chunks: they stream from L2 in halves
//     of a chunk (8 k-steps = 32 VGPRs) through a ring of three buffers, each half requested two halves before it multiplies - as asm
//     loads with counted s_waitcnt vmcnt, the way gconv16.hip does it (the compiler's own bookkeeping of loads in flight across a
//     loop's back edge drains them all);
//   * the requests are issued UNDER the MFMAs: every k-step sends one weight load of the half after next and, in a chunk's first half,
//     one piece (two for the 16-cout form) of the next chunk's tile DMA.  First form: all 16 requests of a chunk in one block at its top
//     - 128 instructions of 1 KB per CU at the address unit's 16 cycles each, the matrix pipe idle: 17 - 19 % of a wave's cycles by the
//     in-kernel stamps (tools/wnc_stamps.py), 4.6 % now (the decoder's first conv 59.8 -> 55.0 us);
//   * a launch takes up to eight JOBS (32-cout slices of one layer, or the groups of a grouped layer): persistent blocks walk
//     (job, image, tile) triples, so the three groups of a decoder layer or the three slices of its first conv fill the chip as one
//     launch.
// Measured: three weight buffers with two halves of lead but the requests still in one block per chunk changed no layer by more than
// 0.5 us - what cost time was that block's issue at the address unit, not the round trip; job parameters fetched once per tile instead
// of indexed out of the kernel arguments per chunk: 19 -> 17 % in that block.  Where a wave's cycles go now (decoder's first conv, 3 jobs x
// 3 chunks, 9 chunk-iterations per block): the two halves 78 % (the matrix pipe's own time for two waves per SIMD is 68 %), exchange +
// epilogue 10 %, barrier 4 %, planning 5 %, waits 3 %.
// Input depths that are not a multiple of 32 (the estimator's 176 and 184) end in a chunk that overlaps the one before it; the
// repeated channels' weights are packed as zeros (wnc_pack) - what they multiply is a finite activation.
#include <type_traits>
#include <vector>

#include "wnc.h"

namespace {

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// TEAMS = 2, NGH = 1: block tile 4 x 64 pixels - two groups of 32 tiles (2 x 64 pixels each), one per team of four waves, one block
// per CU (the maps of >= 30000 pixels).  TEAMS = 1, NGH = 2: block tile 4 x 32 pixels - one group of 16 x 2 tiles - four waves, two
// blocks per CU: 96 x 160 maps are 120 such tiles per image and job (72 of the wide ones, with 17 % of their columns outside).
template <int TEAMS_, int NGH_>
struct WncCfg {
    static constexpr int TEAMS = TEAMS_, NGH = NGH_, WAVES = 4 * TEAMS;
    static constexpr int TH = 4, TW = NGH == 1 ? 64 : 32;
    static_assert((TEAMS == 2 && NGH == 1) || (TEAMS == 1 && NGH == 2), "block tile = TEAMS groups of 32 tiles");
    static constexpr int IN_ROWS = TH + 2, ROWP = TW + 8, PPR = ROWP / 4, PLANE = IN_ROWS * ROWP, PC = IN_ROWS * PPR;
    // one DMA instruction of the block moves CPI whole channels; lane slot = wave * 64 + lane holds the same (channel in group,
    // row, 16-byte piece) for every instruction (conv_wino32.hip: W32Cfg)
    static constexpr int SLOTS_I = WAVES * 64, CPI = (SLOTS_I / PC) & ~1, NI = 32 / CPI, STAGE = NI * WAVES * 256;
    __host__ __device__ static constexpr int chan_off(int c) { return ((c / CPI) * SLOTS_I + (c % CPI) * PC) * 4; }   // floats from the stage base
    static_assert(CPI >= 2 && 32 % CPI == 0 && STAGE >= WAVES * 64 * 32, "stage >= exchange buffer");
    static_assert((chan_off(30) + PLANE) * 4 + 3 * ROWP * 4 + 64 < 65536, "ds_read immediate range");
    static_assert(2 * STAGE * 4 * (TEAMS == 1 ? 2 : 1) <= 160 * 1024, "LDS budget");
};

#ifdef EEM_WNC_STAMPS
#ifndef EEM_WNC_STAMPS_JOBS
#define EEM_WNC_STAMPS_JOBS 3                              // which launches leave their stamps: the decoder's first conv (3 slices x 3 chunks)
#define EEM_WNC_STAMPS_CHUNKS 3
#endif
// stamps build only (-DEEM_WNC_STAMPS, tools/wnc_stamps.py): per wave of the first 256 blocks, cycles (s_memtime) summed over the block's
// iterations: [0] top wait [1] top barrier [2] requests issued [3] first half [4] mid wait [5] second half [6] exchange + epilogue [7] total
__device__ unsigned long long g_wnc_stamps[256 * 8 * 8];
#define WSTAMP(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); wst[i] += now_ - wprev; wprev = now_; __builtin_amdgcn_sched_barrier(0); }
#else
#define WSTAMP(i)
#endif

template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int I, int N, class F>
__device__ __forceinline__ void unroll_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        unroll_for<I + 1, N>(f);
    }
}

// M16: the layers of at most 16 couts (cdc_utils.py:149-151: 160 -> 16, 176 -> 8, 184 -> 3) on v_mfma_f32_16x16x4_f32 - M = 16 couts,
// N = 16 tiles, K = 4 cin: a wave multiplies the two halves of its 32-tile group by the same weight fragment, half the matrix-pipe time
// of the 32-cout form with its upper 16 rows idle.  A k-step is four channels, a half-chunk four k-steps (4 weight loads of 16 bytes);
// accumulator register r of lane l is cout 4 (l / 16) + r, and wave xi finishes register xi: one cout per lane, four stores per tile.
template <class C, bool M16>
__global__ __launch_bounds__(C::WAVES * 64, C::TEAMS == 1 ? 2 : 1) void wnc_kernel(WncArgs a) {
    constexpr int NW = M16 ? 4 : 8;                          // weight loads (k-steps) per half-chunk
    constexpr int NST = M16 ? 4 : 8;                         // stores per wave and tile (always issued: the waits count them)
    constexpr int TH = C::TH, TW = C::TW, WAVES = C::WAVES, ROWP = C::ROWP, PLANE = C::PLANE, PC = C::PC, PPR = C::PPR, CPI = C::CPI, NI = C::NI,
                  STAGE = C::STAGE;
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 31, kk = lane >> 5;
    const int n16 = lane & 15, kq = lane >> 4;               // (M16: tile of a half-group, channel of a k-step)
    const int xi = wave & 3;                                 // Winograd row of this wave
    // tile of this lane inside the block tile: row pair `tr`, column pair `txb`
    const int tr = C::NGH == 1 ? (wave >> 2) : (nl >> 4), txb = C::NGH == 1 ? nl : (nl & 15);
    const int tiles_x = ceil_div(a.w, TW), tiles_y = ceil_div(a.h, TH);
    const TileRange range = block_tile_range(tiles_x * tiles_y * a.njobs * a.n, blockIdx.x, gridDim.x);
    const int ntile = range.count;
    if (ntile == 0) return;
    const int nchunks = a.nchunks, niter = ntile * nchunks;
    const int plane = a.h * a.w;
    TileCoord cur = tile_coord(range.first, tiles_x, tiles_y), nxt = cur;      // .n = job * batch + image
    int cur_ch = 0, nxt_ch = 0;

    // ---- what a (job, image) pair means for this wave, fetched ONCE per tile (round 6, in-kernel stamps - tools/wnc_stamps.py: with
    // the job table indexed and the image / job split divided out in every chunk's request block, that block took 2 600 of a chunk's
    // 13 800 cycles, the matrix pipe idle: 19 % of the decoder's first conv)
    struct Tile {
        const float* in;       // channel 0 of the job's input range in this image
        const char* w;         // this wave's (Winograd row's) weight stream
        const char* bias;
        float* out;            // channel out_coff of this image
        const float* res;
        int out_cmul, cout;
    };
    auto fetch = [&](const TileCoord& tc) __attribute__((always_inline)) {
        const int j = tc.n / a.n, n = tc.n - j * a.n;
        const WncJob& J = a.job[j];
        Tile t;
        t.in = J.in + ((size_t)n * J.in_ctotal + J.in_coff) * plane;
        t.w = reinterpret_cast<const char*>(J.w) + (size_t)xi * 2 * nchunks * NW * 1024;
        t.bias = reinterpret_cast<const char*>(J.bias);
        const size_t ob = ((size_t)n * J.out_ctotal + J.out_coff) * plane;
        t.out = J.out + ob;
        t.res = J.res ? J.res + ob : nullptr;
        t.out_cmul = J.out_cmul; t.cout = J.cout;
        return t;
    };
    Tile tcur = fetch(cur), tnxt = tcur;
    const int cin = a.cin;                                   // chunk k starts at channel min(32 k, cin - 32) (wnc_chunks)

    // ---- DMA plan (fixed for the kernel): slot -> (channel in group, tile row, 16-byte piece)
    const int dslot = wave * 64 + lane;
    const int dcl = dslot / PC, drem = dslot - dcl * PC, dry = drem / PPR, dq = drem - dry * PPR;
    // a chunk's tile request = this lane's first piece address + the distance to the same piece of the next channel group (`plan`), then
    // NI pieces (`piece`): all at once in the prologue (`issue`), one or two per k-step under the MFMAs afterwards
    struct Plan { const char* gp; unsigned step; };
    auto plan = [&](const TileCoord& tc, const Tile& t, int ch) __attribute__((always_inline)) {
        const int choff = min(32 * ch, cin - 32);
        const float* base = t.in + (size_t)choff * plane;
        const int gy = tc.by * TH - 1 + dry, gx = tc.bx * TW - 4 + dq * 4;
        const bool ok = dcl < CPI && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
        Plan p;
        p.gp = ok ? reinterpret_cast<const char*>(base + ((dcl * a.h + gy) * a.w + gx)) : reinterpret_cast<const char*>(a.zero_page);
        p.step = ok ? (unsigned)(CPI * plane) * 4u : 0u;
        return p;
    };
    auto piece = [&](auto k_tag, int stage, const Plan& p) __attribute__((always_inline)) {
        constexpr int k = decltype(k_tag)::value;
        __builtin_amdgcn_global_load_lds(GLB_PTR(p.gp + (size_t)k * p.step), LDS_PTR(lds + stage * STAGE + (wave + k * WAVES) * 256), 16, 0, 0);
    };
    auto issue = [&](int stage, const TileCoord& tc, const Tile& t, int ch) __attribute__((always_inline)) {
        const Plan p = plan(tc, t, ch);
        unroll_for<0, NI>([&](auto k_tag) __attribute__((always_inline)) { piece(k_tag, stage, p); });
    };
    // ---- weights: [xi][half-chunk][k-step of the half (8)][lane][nu] floats per job; a half = 8 asm loads of 16 bytes per lane
    const unsigned vo0 = lane * 16u, vo1 = vo0 + 4096u;
    auto load_w = [&](f32x4 (&dst)[8], const Tile& t, int hf) __attribute__((always_inline)) {
        const char* wb = t.w + (size_t)hf * NW * 1024;
#pragma unroll
        for (int s = 0; s < NW; ++s)
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst[s]) : "v"(s < 4 ? vo0 : vo1), "s"(wb), "n"((s & 3) * 1024) : "memory");
    };
    auto load_w1 = [](auto s_tag, f32x4& dst, const char* wb, unsigned lane_off) __attribute__((always_inline)) {
        constexpr int s = decltype(s_tag)::value;
        const unsigned vo = lane_off + (s < 4 ? 0u : 4096u);
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(vo), "s"(wb), "n"((s & 3) * 1024) : "memory");
    };
    auto landed = [&](f32x4 (&w)[8]) __attribute__((always_inline)) {      // tells the compiler the registers are ready (gconv16.hip)
#pragma unroll
        for (int s = 0; s < NW; ++s) asm volatile("" : "+v"(w[s]));
    };
    // bias of the cout rows this wave finishes: co = r + 8 xi + 4 kk, r = 0..3 (M16: the one cout 4 kq + xi, in biasv[0])
    f32x4 biasv = {0.f, 0.f, 0.f, 0.f};
    const unsigned bo = M16 ? (unsigned)(4 * kq + xi) * 4u : (unsigned)(8 * xi + 4 * kk) * 4u;
    auto load_bias = [&](const Tile& t) __attribute__((always_inline)) {
        const char* bb = t.bias;
        if constexpr (M16) asm volatile("global_load_dword %0, %1, %2" : "=v"(biasv[0]) : "v"(bo), "s"(bb) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(biasv) : "v"(bo), "s"(bb) : "memory");
    };

    // patch rows of this wave: t_xi = e_a + sgn * e_b with (a, b, sgn) = (0,2,-) (1,2,+) (2,1,-) (1,3,-)
    const int ra = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
    const int rb = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
    const float sgn = xi == 1 ? 1.f : -1.f;
    const int lbase = kk * PLANE + 2 * tr * ROWP + 2 * txb + 2;          // patch column -1 sits at the odd half of an aligned pair
    // M16: the lane's tile in the group's first half, and the distance to the one in its second half (16 tile columns on, or the next tile row)
    const int tr0 = C::NGH == 1 ? (wave >> 2) : 0, lbase16 = kq * PLANE + 2 * tr0 * ROWP + 2 * n16 + 2;
    constexpr int DNB = C::NGH == 1 ? 32 : 2 * ROWP;

    f32x16 acc[4];
    f32x4 acc16[4][2];
    f32x4 wr[3][8];
    // one half of a chunk: 8 k-steps (2 channels each) x 4 MFMAs; req(s) issues the k-step's share of the requests under its MFMAs
    auto half = [&](auto hl_tag, const f32x4 (&w)[8], const float* stage, auto&& req) __attribute__((always_inline)) {
        constexpr int HL = decltype(hl_tag)::value;
        if constexpr (!M16) {
            const float* pa = stage + lbase + ra * ROWP;
            const float* pb = stage + lbase + rb * ROWP;
            f32x2 na[3], nb[3];
            auto load_patch = [&](auto sc_tag) __attribute__((always_inline)) {
                constexpr int off = C::chan_off(2 * decltype(sc_tag)::value);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    na[q] = *reinterpret_cast<const f32x2*>(pa + off + 2 * q);
                    nb[q] = *reinterpret_cast<const f32x2*>(pb + off + 2 * q);
                }
            };
            load_patch(std::integral_constant<int, HL * 8>{});
            unroll_for<0, 8>([&](auto s_tag) __attribute__((always_inline)) {
                constexpr int s = decltype(s_tag)::value;
                const float ea[4] = {na[0][1], na[1][0], na[1][1], na[2][0]};
                const float eb[4] = {nb[0][1], nb[1][0], nb[1][1], nb[2][0]};
                if constexpr (s + 1 < 8) {
                    load_patch(std::integral_constant<int, HL * 8 + s + 1>{});
                    __builtin_amdgcn_sched_barrier(0);       // the next k-step's reads stay ahead of this one's transform and MFMAs
                }
                float t[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) t[b] = __builtin_fmaf(sgn, eb[b], ea[b]);
                const float v[4] = {t[0] - t[2], t[1] + t[2], t[2] - t[1], t[1] - t[3]};
#pragma unroll
                for (int nu = 0; nu < 4; ++nu) acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[s][nu], v[nu], acc[nu], 0, 0, 0);
                req(s_tag);
            });
        } else {
            // four k-steps of four channels (this lane: channel 4 s + kq), two half-groups of 16 tiles per weight fragment
            const float* pa = stage + lbase16 + ra * ROWP;
            const float* pb = stage + lbase16 + rb * ROWP;
            f32x2 na[2][3], nb[2][3];
            auto load_patch = [&](auto sc_tag) __attribute__((always_inline)) {
                constexpr int off = C::chan_off(4 * decltype(sc_tag)::value);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        na[h2][q] = *reinterpret_cast<const f32x2*>(pa + off + h2 * DNB + 2 * q);
                        nb[h2][q] = *reinterpret_cast<const f32x2*>(pb + off + h2 * DNB + 2 * q);
                    }
            };
            load_patch(std::integral_constant<int, HL * 4>{});
            unroll_for<0, 4>([&](auto s_tag) __attribute__((always_inline)) {
                constexpr int s = decltype(s_tag)::value;
                float v[2][4];
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const float ea[4] = {na[h2][0][1], na[h2][1][0], na[h2][1][1], na[h2][2][0]};
                    const float eb[4] = {nb[h2][0][1], nb[h2][1][0], nb[h2][1][1], nb[h2][2][0]};
                    float t[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) t[b] = __builtin_fmaf(sgn, eb[b], ea[b]);
                    v[h2][0] = t[0] - t[2]; v[h2][1] = t[1] + t[2]; v[h2][2] = t[2] - t[1]; v[h2][3] = t[1] - t[3];
                }
                if constexpr (s + 1 < 4) {
                    load_patch(std::integral_constant<int, HL * 4 + s + 1>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int nu = 0; nu < 4; ++nu)
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2)
                        acc16[nu][h2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s][nu], v[h2][nu], acc16[nu][h2], 0, 0, 0);
                req(s_tag);
            });
        }
    };

    // ---- requests (second form, round 6; in-kernel stamps, tools/wnc_stamps.py): with a chunk's 16 requests per wave issued in one block at
    // its top - 128 instructions of 1 KB per CU, 16 cycles each at the address unit - that block was 17 - 19 % of a wave's cycles with the
    // matrix pipe idle.  Now every k-step issues its share under its MFMAs: one weight load (half g + 2's k-step s: three half-chunk buffers,
    // half g lives in buffer g % 3, so the loop body comes in three copies - P = chunk mod 3) and, in a chunk's first half, one or two pieces
    // of the next chunk's tile.  Issue order: prologue [W(0)] [tile 0] [W(1)]; chunk it: [bias] | half 0: W(2it+2)[s], tile(it+1)[s] ... |
    // half 1: W(2it+3)[s] ... | [stores].  Waits: at the top everything up to tile(it)'s last piece - younger are W(2it+1) and the previous
    // tile's stores; in the middle W(2it+1) - younger are the first half's NW + NI requests (the bias and any stores in between are waited
    // for as well: vmcnt counts from the youngest).
    load_w(wr[0], tcur, 0);
    issue(0, cur, tcur, 0);
    load_w(wr[1], tcur, 1);
    bool stored = false;                                     // the previous iteration ended in a tile's stores (the youngest requests)
#ifdef EEM_WNC_STAMPS
    unsigned long long wst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long wprev = __builtin_amdgcn_s_memtime();
    const unsigned long long wstart = wprev;
#endif
    auto chunk = [&](auto p_tag, int it) __attribute__((always_inline)) {
        constexpr int P = decltype(p_tag)::value;
        f32x4(&B0)[8] = wr[(2 * P) % 3];
        f32x4(&B1)[8] = wr[(2 * P + 1) % 3];
        f32x4(&B2)[8] = wr[(2 * P + 2) % 3];
        const bool first = cur_ch == 0, last = cur_ch == nchunks - 1, more = it + 1 < niter;
        // this chunk's tile and its first half of weights have landed; every wave is through with the other stage
        if (stored) wait_vm<NW + NST>(); else wait_vm<NW>();
        landed(B0);
        WSTAMP(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        WSTAMP(1)
        if (first) {
            load_bias(tcur);
            if constexpr (M16) {
#pragma unroll
                for (int nu = 0; nu < 4; ++nu) { acc16[nu][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc16[nu][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            } else {
#pragma unroll
                for (int nu = 0; nu < 4; ++nu)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[nu][r] = 0.f;
            }
        }
        Plan pn = {reinterpret_cast<const char*>(a.zero_page), 0u};
        const char* wn = tcur.w;
        if (more) {
            if (++nxt_ch == nchunks) {
                nxt_ch = 0;
                const int n_before = nxt.n;
                tile_advance(nxt, tiles_x, tiles_y);
                if (nxt.n != n_before) tnxt = fetch(nxt);    // (a block's tiles are consecutive: the pair changes at most a few times)
            }
            pn = plan(nxt, tnxt, nxt_ch);
            wn = tnxt.w + (size_t)(2 * nxt_ch) * NW * 1024;
        }
        const int nstage = (it + 1) & 1;
        __builtin_amdgcn_sched_barrier(0);
        WSTAMP(2)
        const float* stage = lds + (it & 1) * STAGE;
        half(std::integral_constant<int, 0>{}, B0, stage, [&](auto s_tag) __attribute__((always_inline)) {
            constexpr int s = decltype(s_tag)::value;
            if (more) {
                load_w1(s_tag, B2[s], wn, vo0);
                constexpr int PPS = NI / NW;                 // tile pieces per k-step
                unroll_for<0, PPS>([&](auto q_tag) __attribute__((always_inline)) {
                    piece(std::integral_constant<int, s * PPS + decltype(q_tag)::value>{}, nstage, pn);
                });
            }
        });
        WSTAMP(3)
        // the second half's weights (and the bias): only the first half's requests are younger
        if (more) wait_vm<NW + NI>(); else wait_vm<0>();
        landed(B1);
        asm volatile("" : "+v"(biasv));
        WSTAMP(4)
        __builtin_amdgcn_sched_barrier(0);
        half(std::integral_constant<int, 1>{}, B1, stage, [&](auto s_tag) __attribute__((always_inline)) {
            constexpr int s = decltype(s_tag)::value;
            if (more) load_w1(s_tag, B0[s], wn + (size_t)NW * 1024, vo0);
        });
        WSTAMP(5)
        stored = false;
        if (last && M16) {
            // ---- M16: u of (half-group h2, cout register r) at [wave][h2 * 4 + r][lane][2]; wave xi finishes register xi of both halves
            __builtin_amdgcn_s_barrier();
            float* xst = lds + (it & 1) * STAGE;
            f32x2* xw = reinterpret_cast<f32x2*>(xst) + (wave * 8) * 64 + lane;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    f32x2 u;
                    u[0] = acc16[0][h2][r] + acc16[1][h2][r] + acc16[2][h2][r];
                    u[1] = acc16[1][h2][r] - acc16[2][h2][r] - acc16[3][h2][r];
                    xw[(h2 * 4 + r) * 64] = u;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const f32x2* xr = reinterpret_cast<const f32x2*>(xst) + ((wave - xi) * 8 + xi) * 64 + lane;
            const int co = 4 * kq + xi;
            const bool okc = co < tcur.cout;
            float* sink = a.trash + lane * 2;
            float* pc = tcur.out + (size_t)co * tcur.out_cmul * plane;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const f32x2 u0 = xr[(0 * 8 + h2 * 4) * 64], u1 = xr[(1 * 8 + h2 * 4) * 64], u2 = xr[(2 * 8 + h2 * 4) * 64], u3 = xr[(3 * 8 + h2 * 4) * 64];
                float y00 = u0[0] + u1[0] + u2[0] + biasv[0], y01 = u0[1] + u1[1] + u2[1] + biasv[0];
                float y10 = u1[0] - u2[0] - u3[0] + biasv[0], y11 = u1[1] - u2[1] - u3[1] + biasv[0];
                if (a.act) {
                    const float sl = a.act == 1 ? 0.1f : 0.f;
                    y00 = fmaxf(y00, sl * y00); y01 = fmaxf(y01, sl * y01);
                    y10 = fmaxf(y10, sl * y10); y11 = fmaxf(y11, sl * y11);
                }
                const int oy = cur.by * TH + 2 * (C::NGH == 1 ? tr0 : h2), ox = cur.bx * TW + 2 * (C::NGH == 1 ? h2 * 16 + n16 : n16);
                const bool in0 = oy < a.h && ox < a.w, in1 = oy + 1 < a.h && ox < a.w;
                float* p = pc + (size_t)oy * a.w + ox;
                *reinterpret_cast<f32x2*>(in0 && okc ? p : sink) = f32x2{y00, y01};
                *reinterpret_cast<f32x2*>(in1 && okc ? p + a.w : sink) = f32x2{y10, y11};
            }
            stored = true;
        }
        if (last && !M16) {
            // ---- u = M_xi A, exchanged through the stage this tile has finished with: [wave][r][lane][2]
            __builtin_amdgcn_s_barrier();                    // every wave is done reading the stage
            float* xst = lds + (it & 1) * STAGE;
            f32x2* xw = reinterpret_cast<f32x2*>(xst) + (wave * 16) * 64 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                f32x2 u;
                u[0] = acc[0][r] + acc[1][r] + acc[2][r];
                u[1] = acc[1][r] - acc[2][r] - acc[3][r];
                xw[r * 64] = u;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // wave xi finishes accumulator rows 4 xi .. 4 xi + 3 of its team: couts r + 8 xi + 4 kk
            const f32x2* xr = reinterpret_cast<const f32x2*>(xst) + ((wave - xi) * 16 + 4 * xi) * 64 + lane;
            const int oy = cur.by * TH + 2 * tr, ox = cur.bx * TW + 2 * txb;
            const bool in0 = oy < a.h && ox < a.w, in1 = oy + 1 < a.h && ox < a.w;
            const int co0 = 8 * xi + 4 * kk;
            float* sink = a.trash + lane * 2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x2 u0 = xr[(0 * 16 + r) * 64], u1 = xr[(1 * 16 + r) * 64], u2 = xr[(2 * 16 + r) * 64], u3 = xr[(3 * 16 + r) * 64];
                float y00 = u0[0] + u1[0] + u2[0] + biasv[r], y01 = u0[1] + u1[1] + u2[1] + biasv[r];
                float y10 = u1[0] - u2[0] - u3[0] + biasv[r], y11 = u1[1] - u2[1] - u3[1] + biasv[r];
                if (a.act) {
                    const float sl = a.act == 1 ? 0.1f : 0.f;
                    y00 = fmaxf(y00, sl * y00); y01 = fmaxf(y01, sl * y01);
                    y10 = fmaxf(y10, sl * y10); y11 = fmaxf(y11, sl * y11);
                }
                const int co = co0 + r;
                const bool okc = co < tcur.cout;
                const size_t po = (size_t)co * tcur.out_cmul * plane + (size_t)oy * a.w + ox;
                float* p = tcur.out + po;
                if (tcur.res) {                              // (wave-uniform) residual block: relu(res + .); outside lanes read the zero page
                    const f32x2 r0 = *reinterpret_cast<const f32x2*>(in0 && okc ? tcur.res + po : a.zero_page);
                    const f32x2 r1 = *reinterpret_cast<const f32x2*>(in1 && okc ? tcur.res + po + a.w : a.zero_page);
                    y00 = fmaxf(y00 + r0[0], 0.f); y01 = fmaxf(y01 + r0[1], 0.f);
                    y10 = fmaxf(y10 + r1[0], 0.f); y11 = fmaxf(y11 + r1[1], 0.f);
                }
                // every lane stores (lanes outside the image or beyond cout into a scratch page): exactly NST stores per wave and tile
                *reinterpret_cast<f32x2*>(in0 && okc ? p : sink) = f32x2{y00, y01};
                *reinterpret_cast<f32x2*>(in1 && okc ? p + a.w : sink) = f32x2{y10, y11};
            }
            stored = true;
        }
        WSTAMP(6)
        cur = nxt;
        cur_ch = nxt_ch;
        tcur = tnxt;
    };
    {
        int it = 0;
#pragma unroll 1
        while (it < niter) {
            chunk(std::integral_constant<int, 0>{}, it);
            if (++it >= niter) break;
            chunk(std::integral_constant<int, 1>{}, it);
            if (++it >= niter) break;
            chunk(std::integral_constant<int, 2>{}, it);
            ++it;
        }
    }
#ifdef EEM_WNC_STAMPS
    wst[7] = __builtin_amdgcn_s_memtime() - wstart;
    if (lane == 0 && blockIdx.x < 256 && wave < 8 && a.njobs == EEM_WNC_STAMPS_JOBS && a.nchunks == EEM_WNC_STAMPS_CHUNKS)
        for (int i = 0; i < 8; ++i) g_wnc_stamps[(blockIdx.x * 8 + wave) * 8 + i] = wst[i];
#endif
}

}  // namespace

#ifdef EEM_WNC_STAMPS
extern "C" __attribute__((visibility("default"))) int eemflow_debug_read_wnc_stamps(unsigned long long* dst, size_t n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wnc_stamps), n * sizeof(unsigned long long)) == hipSuccess ? 0 : 2;
}
#endif

int wnc_chunks(int cin, int* chunk_off) {
    int n = 0;
    for (int c = 0; c + 32 <= cin; c += 32) chunk_off[n++] = c;
    if (cin % 32) chunk_off[n++] = cin - 32;
    return n;
}

size_t wnc_packed_floats(int cin, int m16) {
    int off[WNC_MAX_CHUNKS + 2];
    const int nch = cin >= 32 && cin <= 32 * WNC_MAX_CHUNKS ? wnc_chunks(cin, off) : 0;
    return (size_t)4 * 2 * nch * (m16 ? 4 : 8) * 64 * 4;
}

// four floats of the stream: quad q = ((xi * 2 nch + hf) * nw + s) * 64 + lane (host and device: wnc_pack, wnc_pack_kernel)
__host__ __device__ static inline void wnc_pack_quad(const float* w, int cout, int cin, int co0, int m16, int q, float* o) {
    const int nch = (cin + 31) / 32;
    const int rep = cin % 32 ? 32 - cin % 32 : 0;            // the last chunk's first `rep` channels repeat the chunk before it
    const int nw = m16 ? 4 : 8;                              // k-steps per half-chunk (of 4 / 2 channels)
    const int lane = q & 63, s = (q >> 6) % nw, hf = (q >> 6) / nw % (2 * nch), xi = (q >> 6) / nw / (2 * nch);
    const int k = hf >> 1;
    const int l = (hf & 1) * 16 + (m16 ? 4 * s + (lane >> 4) : 2 * s + (lane >> 5));
    const int co = co0 + (m16 ? (lane & 15) : (lane & 31));
    const int ci = (k == nch - 1 && cin % 32 ? cin - 32 : 32 * k) + l;            // (wnc_chunks' offsets)
    if (co >= cout || (k == nch - 1 && l < rep)) { o[0] = o[1] = o[2] = o[3] = 0.f; return; }
    const float* g = w + ((size_t)co * cin + ci) * 9;
    float m[3];                                              // row xi of G g
    for (int kx = 0; kx < 3; ++kx) {
        const float g0 = g[kx], g1 = g[3 + kx], g2 = g[6 + kx];
        m[kx] = xi == 0 ? g0 : (xi == 1 ? 0.5f * (g0 + g1 + g2) : (xi == 2 ? 0.5f * (g0 - g1 + g2) : g2));
    }
    o[0] = m[0];
    o[1] = 0.5f * (m[0] + m[1] + m[2]);
    o[2] = 0.5f * (m[0] - m[1] + m[2]);
    o[3] = m[2];
}

void wnc_pack(const float* w, int cout, int cin, int co0, int m16, float* packed) {
    const int quads = 4 * 2 * ((cin + 31) / 32) * (m16 ? 4 : 8) * 64;
    for (int q = 0; q < quads; ++q) wnc_pack_quad(w, cout, cin, co0, m16, q, packed + (size_t)q * 4);
}

namespace {
__global__ __launch_bounds__(256) void wnc_pack_kernel(WncPackArgs a) {
    const WncPackJob J = a.job[blockIdx.y];
    const int quads = 4 * 2 * ((J.cin + 31) / 32) * (J.m16 ? 4 : 8) * 64;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < (J.m16 ? 16 : 32) && J.bias_out)
        J.bias_out[threadIdx.x] = J.co0 + (int)threadIdx.x < J.cout ? J.bias[J.co0 + threadIdx.x] : 0.f;
    if (q >= quads) return;
    float o[4];
    wnc_pack_quad(J.w, J.cout, J.cin, J.co0, J.m16, q, o);
    *reinterpret_cast<f32x4*>(J.packed + (size_t)q * 4) = f32x4{o[0], o[1], o[2], o[3]};
}
}  // namespace

int wnc_pack_device_launch(const WncPackArgs& a, hipStream_t st) {
    EEM_REQUIRE(a.njobs >= 1 && a.njobs <= WNC_PACK_MAX_JOBS, "wnc_pack_device_launch: njobs = %d", a.njobs);
    int maxq = 0;
    for (int j = 0; j < a.njobs; ++j) {
        const WncPackJob& J = a.job[j];
        EEM_REQUIRE(J.w && J.packed && J.cin >= 32 && J.cin <= 32 * WNC_MAX_CHUNKS && !((uintptr_t)J.packed & 15), "wnc_pack_device_launch: job %d", j);
        const int quads = 4 * 2 * ((J.cin + 31) / 32) * (J.m16 ? 4 : 8) * 64;
        maxq = quads > maxq ? quads : maxq;
    }
    hipLaunchKernelGGL(wnc_pack_kernel, dim3((maxq + 255) / 256, a.njobs), dim3(256), 0, st, a);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

bool wnc_supported(const WncArgs& a) {
    const char* e = getenv("EEM_NO_WNC");                   // read per call: a test runs both forms in one process
    if (e && e[0] == '1') return false;
    if (a.njobs < 1 || a.njobs > WNC_MAX_JOBS || a.nchunks < 1 || a.nchunks > WNC_MAX_CHUNKS || a.n < 1) return false;
    if (a.cin < 32 || (a.cin + 31) / 32 != a.nchunks) return false;
    if (a.w % 4 || a.h < 1 || !a.zero_page || !a.trash || ((uintptr_t)a.zero_page & 15) || ((uintptr_t)a.trash & 7)) return false;
    if ((size_t)a.h * a.w * 32 * 4 >= (1u << 31)) return false;                  // 32-bit byte offsets inside a chunk
    for (int j = 0; j < a.njobs; ++j) {
        const WncJob& J = a.job[j];
        if (!J.in || !J.w || !J.bias || !J.out || (J.res && (a.m16 || ((uintptr_t)J.res & 15))) || J.cout < 1 || J.cout > (a.m16 ? 16 : 32) || J.out_cmul < 1) return false;
        if (((uintptr_t)J.in & 15) || ((uintptr_t)J.w & 15) || ((uintptr_t)J.bias & 15) || ((uintptr_t)J.out & 15)) return false;
    }
    return true;
}

int wnc_launch(const WncArgs& a, hipStream_t st) {
    // EEM_WNC_SMALL_MAXPX (read per call; 30000): maps below it take the 4 x 32 block tile, two blocks per CU
    const char* m = getenv("EEM_WNC_SMALL_MAXPX");
    const bool small = (long)a.h * a.w < (m ? atol(m) : 30000L);
    if (small) {
        using C = WncCfg<1, 2>;
        const int T = ceil_div(a.w, C::TW) * ceil_div(a.h, C::TH) * a.njobs * a.n;
        int per_xcd = ceil_div(T, 8);
        if (per_xcd > 64) per_xcd = 64;                      // two resident blocks per CU
        if (a.m16) hipLaunchKernelGGL((wnc_kernel<C, true>), dim3(per_xcd * 8), dim3(C::WAVES * 64), 0, st, a);
        else hipLaunchKernelGGL((wnc_kernel<C, false>), dim3(per_xcd * 8), dim3(C::WAVES * 64), 0, st, a);
    } else {
        using C = WncCfg<2, 1>;
        const int T = ceil_div(a.w, C::TW) * ceil_div(a.h, C::TH) * a.njobs * a.n;
        int per_xcd = ceil_div(T, 8);
        if (per_xcd > 32) per_xcd = 32;                      // one resident block per CU
        if (a.m16) hipLaunchKernelGGL((wnc_kernel<C, true>), dim3(per_xcd * 8), dim3(C::WAVES * 64), 0, st, a);
        else hipLaunchKernelGGL((wnc_kernel<C, false>), dim3(per_xcd * 8), dim3(C::WAVES * 64), 0, st, a);
    }
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}
