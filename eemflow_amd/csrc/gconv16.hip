// Stride-1 convolutions with 16-channel-aligned inputs (E-RAFT's update block and residual stacks, EEMFlow+'s decoders:
// 1x1, 3x3, 1x5, 5x1 kernels, 32..384 -> 16..576 channels, inputs that are the concatenation of up to three tensors) as an
// LDS-tiled implicit GEMM on v_mfma_f32_16x16x4_f32.  The generic kernel of gconv.hip reads both operands of every MFMA
// straight from L2 (two 256-byte loads per MFMA, 44 TFLOP/s at best); here
//   * a block = 4 waves = one 64-cout chunk x a 4x16-pixel tile (2x16 for launches of few blocks, whose run time is quantised
//     in whole blocks per CU); wave w owns the 16 couts of M-tile w and all pixel rows
//     (N-tiles), so every weight fragment feeds four MFMAs and every input fragment is read once per wave from LDS (layers
//     of <= 32 couts: two M-tiles x two row groups per block, so no wave idles);
//   * the input is consumed in chunks of 16 channels: the chunk's haloed tile [16][4 + KH - 1][24] (columns x0 - 4 .. x0 + 19; 16 columns
//     for the one-column-wide filters)
//     goes HBM/L2 -> LDS by 16-byte LDS-DMA, double-buffered, out-of-image pieces as out-of-range buffer offsets (zeros); the plane pitch is padded to
//     16 mod 32 floats so the four channel lanes of a k-step fall on different bank halves;
//   * the chunk's weight fragments (taps x 4 k-steps) are loaded into registers from a stream packed in fragment order
//     (256 bytes per wave and k-step, the four waves' fragments adjacent) one chunk ahead of their use;
//   * one barrier per chunk; epilogue as gconv's (scale/shift, activation, GRU / residual combinations).
#include "gconv.h"

namespace {

#ifdef EEM_G16_STAMPS
// per wave: [0] start [1] prologue done [2] sum(wait vmcnt) [3] sum(barrier) [4] sum(requests) [5] sum(compute) [6] loop done [7] end
__device__ unsigned long long g_g16_stamps[1024 * 4 * 8];
#define G16_T() __builtin_amdgcn_s_memtime()
#endif

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float g16_act(float v, int act) {
    switch (act) {
        case GACT_RELU: return v > 0.f ? v : 0.f;
        case GACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case GACT_TANH: return tanhf(v);
        case GACT_LEAKY: return v > 0.f ? v : 0.1f * v;
        default: return v;
    }
}

// S = stride (1, or 2 for the encoders' downsampling 3x3 / 1x1 convs: the tile's input rows and columns are twice as far apart, the
// staged patch (TH - 1) * 2 + KH rows of 40 columns, a lane's B fragments two floats apart - a 2-way bank conflict it can afford)
template <int KH, int KW, int THT, int S = 1>
struct G16Cfg {
    static constexpr int TH = THT, TW = 16;
    static constexpr int ROWS = (TH - 1) * S + KH;
    // columns S * x0 - XM .. S * x0 + 16 S + XM - 1: a four-column margin either side for filters wider than one column, none for the
    // 1x1 and 5x1 filters (round 6: they staged - and fetched - 24 columns for 16 too: a third of a 1x1 conv's input traffic)
    static constexpr int XM = KW > 1 ? 4 : 0;
    static constexpr int COLS = 16 * S + 2 * XM;
    static constexpr int PCS = COLS / 4;                         // 16-byte pieces per staged row
    static constexpr int PL0 = ROWS * COLS;
    static constexpr int PL = PL0 % 32 == 16 ? PL0 : PL0 + ((48 - PL0 % 32) % 32);   // plane pitch = 16 mod 32 floats
    static constexpr int PQ = PL / 4;                            // 16-byte slots per plane (ROWS * PCS real ones)
    static constexpr int SLOTS = 16 * PQ;
    static constexpr int NI = (SLOTS + 255) / 256;               // DMA instructions per wave and chunk
    static constexpr int STAGE = NI * 256 * 4;                   // floats
    static constexpr int TAPS = KH * KW;
    static constexpr int KS = TAPS * 4;                          // k-steps (weight fragments) per chunk
    static_assert(PL % 4 == 0 && PL % 32 == 16, "plane pitch");
};

// base address of input chunk `ch` (16 channels) of an input that is the concatenation of up to three tensors
__device__ __forceinline__ const char* g16_chunk_base(int ch, const char* sp0, const char* sp1, const char* sp2, int sc0, int sc1, int hw) {
    const int c0 = ch * 16, c1 = c0 - sc0, c2 = c1 - sc1;
    const char* b = c0 < sc0 ? sp0 : (c1 < sc1 ? sp1 : sp2);
    const int c = c0 < sc0 ? c0 : (c1 < sc1 ? c1 : c2);
    return b + (size_t)c * hw * 4;
}

// One chunk's tile pieces, HBM/L2 -> LDS, as buffer loads: the descriptor's base is the chunk's first channel (scalar), a lane's
// offset inside the chunk is fixed for the whole kernel, and out-of-image pieces carry an offset beyond the descriptor's range - the
// hardware writes zeros for them.  No vector instruction per chunk: with several waves on a SIMD every VALU instruction of the
// request phase waited for a partner's MFMA to leave the shared pipe (stamps: 20 % of a full launch's wave time was spent issuing
// requests, 64-bit pointer selects against a zero page included).
constexpr unsigned G16_RANGE = 0x7fffff00u;                     // bytes a descriptor covers; offsets at or above it read as zero
template <class C, int NI>
__device__ __forceinline__ void g16_issue(float* sb, const char* xb, const unsigned (&off)[NI], int wave) {
#if __HIP_DEVICE_COMPILE__                                      // (the buffer-resource type does not exist in the host pass)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xb), (short)0, (int)G16_RANGE, 0x00020000);
#pragma unroll
    for (int k = 0; k < NI; ++k)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(sb + (wave + 4 * k) * 256), 16, off[k], 0, 0, 0);
#endif
}

// s_waitcnt vmcnt(N): the oldest requests have landed, the N youngest may still fly
template <int N>
__device__ __forceinline__ void g16_wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// KG = groups of four waves that split the channel chunks of one tile (group g takes chunks g, g + KG, ...; partial sums meet in LDS).
// 1 for launches that fill the chip.  2 for launches of at most one block per CU (E-RAFT's 60 x 80 update block at batch 1: 150-300
// blocks), where a SIMD holds ONE wave and nothing overlaps that wave's own instruction stream: in-kernel stamps (EEM_G16_STAMPS,
// tools/g16_stamps.py; 384 -> 128 1x5, 200 blocks of 3-row tiles) read 84 k cycles per wave for 46 k of MFMAs - 10 k issuing the next
// chunk's requests (a lone wave issues one instruction per 4 cycles, ~100 per chunk), 12 k of LDS round trips inside the MFMA stream,
// 8.5 k in the epilogue (bias loads and stores one after the other), 5 k waiting.  A second wave on the SIMD fills those gaps with its
// MFMAs, which is what a second frame in flight did (+ 50 %).
// Measured and not kept: a four-stage ring (three chunks of LDS-DMA and weight fragments in flight, s_waitcnt vmcnt(2 x requests per
// chunk)) and two chunks per barrier - identical or worse kernel times, those launches are not waiting for memory; B fragments by
// counted asm ds_read2_b32 one tap ahead of the MFMAs - the same at one wave per SIMD, 5 % slower on full launches.
// tools/micro/dispatch_map.hip: the dispatcher does spread 200 blocks over 200 CUs.
// S = stride (1, or 2 for the encoders' downsampling convs: the staged tile is (TH - 1) S + KH rows of 16 S + 8 columns, a B fragment
// reads every S-th of them).  Its occupancy bound stays at 2 blocks: at 3 (168 VGPRs) the compiler spilled, and a spill is fatal here -
// the weight fragments arrive by asm loads the compiler takes for complete, so it parks a register in scratch and hands it to an
// address computation while the load is still in flight (seen as a memory fault at a wild address; no instance may use scratch).
template <int KH, int KW, int THT, int WM, int KG, int S = 1>
__global__ __launch_bounds__(256 * KG, (KG > 1 || S > 1 || (THT != 2 && THT != 4)) ? 2 : ((KH * KW == 9 && THT == 4 && WM >= 2) ? 3 : 4)) void gconv16_kernel(GConvArgs a, const float* __restrict__ wpk16, const float* __restrict__ zero_page,
                                                      int tiles_x, int nchunks) {
    using C = G16Cfg<KH, KW, THT, S>;
#ifdef EEM_G16_STAMPS
    unsigned long long st[8] = {G16_T(), 0, 0, 0, 0, 0, 0, 0};
#endif
    constexpr int WP = 4 / WM, NR = C::TH / WP;                  // pixel-row groups, rows per wave
    static_assert(NR >= 1, "a wave needs a row");
    constexpr int RED = KG > 1 ? 4 * NR * 256 : 0;               // floats of the partial sums one group hands over
    constexpr int LDS_FLOATS = KG * 2 * C::STAGE > RED ? KG * 2 * C::STAGE : RED;
    __shared__ __attribute__((aligned(16))) float lds_all[LDS_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3);      // within the K group
    const int kg = KG > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0;
    float* const lds = lds_all + kg * 2 * C::STAGE;
    const int j = lane & 15, g = lane >> 4;
    // WM = waves along the couts (16 each): 4 -> a block covers 64 couts and every wave all TH rows; 2 -> 32 couts (layers of
    // <= 32 couts), the two wave pairs split the rows; 1 -> 16 couts (EEMFlow+'s 160 -> 16 estimator layer), a row per wave
    const int wm = wave % WM, wp = wave / WM;
    int grp = 0, by = blockIdx.y;                                // grouped launch: blockIdx.y = (group, cout tile of the group)
    if (a.groups > 1) { const int tpg = gridDim.y / a.groups; grp = by / tpg; by -= grp * tpg; }
    wpk16 += (size_t)grp * a.g_wstride16;
    const int mtg = by * WM + wm;                                // 16-cout tile of this wave (within its group)
    const int n = blockIdx.z, cc = mtg >> 2, mt = mtg & 3;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * C::TH, x0 = tx * C::TW;
    const int hw = a.hin * a.win;                                // input plane; "same" padding
    const int hwo = S == 1 ? hw : a.hout * a.wout;               // output plane (= the input's at stride 1)
    constexpr int PH = KH / 2, PW = KW / 2;

    // ---- DMA plan: slot f -> (channel, row, 16-byte piece) of the chunk's tile
    unsigned off[C::NI];
#pragma unroll
    for (int k = 0; k < C::NI; ++k) {
        const int f = (wave + 4 * k) * 64 + lane;
        const int ci = f / C::PQ, q = f - ci * C::PQ;
        const int row = q / C::PCS, pc = q - row * C::PCS;
        const bool ok = f < C::SLOTS && q < C::ROWS * C::PCS;
        const int gy = y0 * S - PH + row, gx = x0 * S - C::XM + 4 * pc;
        const bool in = ok && gy >= 0 && gy < a.hin && gx >= 0 && gx + 4 <= a.win;
        off[k] = in ? (unsigned)(((size_t)ci * hw + (size_t)gy * a.win + gx) * 4) : G16_RANGE;
    }
    // chunk index -> (segment, first channel).  The three descriptors are separate scalars on purpose: as arrays the compiler turned
    // the chain of selects into an indexed read of a private (scratch) copy - a scratch load and an `s_waitcnt vmcnt(0)` in front of
    // every chunk's DMA (kernel arguments indexed dynamically were a scalar load + wait in the same place)
#define G16_SEG_PTR(S) (reinterpret_cast<const char*>(a.seg[S].ptr + ((size_t)n * a.seg[S].ctotal + a.seg[S].coff) * hw))
    const char* const sp0 = G16_SEG_PTR(0) + (size_t)grp * a.seg[0].c * hw * 4;
    const char* const sp1 = a.nseg > 1 ? G16_SEG_PTR(1) : nullptr;
    const char* const sp2 = a.nseg > 2 ? G16_SEG_PTR(2) : nullptr;
#undef G16_SEG_PTR
    const int sc0 = a.seg[0].c, sc1 = a.nseg > 1 ? a.seg[1].c : 0;
    // (free functions with by-value scalars, not lambdas: a closure that another closure captures stays in memory across the loop's
    // `memory`-clobbering waits, and its fields were re-read from scratch in front of every chunk's DMA)
    auto issue = [=](int stage, int ch) __attribute__((always_inline)) {
        g16_issue<C>(lds + stage * C::STAGE, g16_chunk_base(ch, sp0, sp1, sp2, sc0, sc1, hw), off, wave);
    };
    // weight fragments of chunk `ch` for this wave's M-tile: stream[((((cc * nchunks + ch) * 4 + mt) * TAPS + tap) * 64 + lane] is the
    // float4 of the tap's four k-steps (channel groups) - one 16-byte load per tap, a wave's taps 1 KB apart: scalar base of the
    // (chunk, M-tile) + the lane's fixed offset + an immediate, no vector arithmetic per chunk
    // (as asm: the compiler's own bookkeeping of loads in flight across the loop's back edge drained them all - vmcnt(0) - where the
    // explicit waits of the chunk loop meant to keep requests in flight; requests it does not see are synchronised by those waits alone)
    const unsigned wl0 = lane * 16u, wl1 = wl0 + 4096u, wl2 = wl0 + 8192u;
    auto load_w = [&](int ch, f32x4 (&wr)[C::TAPS]) __attribute__((always_inline)) {
        const char* wb = reinterpret_cast<const char*>(wpk16) + ((((size_t)cc * nchunks + ch) * 4 + mt) * C::TAPS) * 1024;
#pragma unroll
        for (int t = 0; t < C::TAPS; ++t) {
            const unsigned vo = t < 4 ? wl0 : (t < 8 ? wl1 : wl2);
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(wr[t]) : "v"(vo), "s"(wb), "n"((t & 3) * 1024) : "memory");
        }
    };

    f32x4 acc[NR];
#pragma unroll
    for (int t = 0; t < NR; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int bbase = g * C::PL + S * j + C::XM - PW;            // B fragment: channel g of a group, pixel column j

    f32x4 wr[2][C::TAPS];
    auto compute = [&](int stage, f32x4 (&w)[C::TAPS]) __attribute__((always_inline)) {
        const float* sb = lds + stage * C::STAGE;
#pragma unroll
        for (int ky = 0; ky < KH; ++ky)
#pragma unroll
            for (int kx = 0; kx < KW; ++kx)
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) {
                    float bv[NR];
#pragma unroll
                    for (int t = 0; t < NR; ++t) bv[t] = sb[bbase + cg * 4 * C::PL + (S * (wp * NR + t) + ky) * C::COLS + kx];
#pragma unroll
                    for (int t = 0; t < NR; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ky * KW + kx][cg], bv[t], acc[t], 0, 0, 0);
                }
    };
    // After the explicit wait the fragments of the current chunk have landed; passing them through an empty asm tells the
    // compiler so - otherwise it guards their first use with its own s_waitcnt vmcnt(0), which also waits for the younger chunks'
    // requests and serialises the whole prefetch.
    auto landed = [&](f32x4 (&w)[C::TAPS]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < C::TAPS; ++t) asm volatile("" : "+v"(w[t]));
    };
    // bias / scale of this lane's four couts: requested now, they arrive under the chunk loop (as part of the epilogue they were two
    // dependent round trips in front of the stores)
    float e_scale[4], e_shift[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = min(mtg * 16 + 4 * g + r, a.cout - 1);
        e_scale[r] = a.scale ? a.scale[co + grp * a.g_pstride] : 1.f;
        e_shift[r] = a.shift ? a.shift[co + grp * a.g_pstride] : 0.f;
    }
    // group kg takes the chunks kg, kg + KG, ...; every group runs the same number of steps (the barrier counts all waves)
    const int nmine = (nchunks - kg + KG - 1) / KG, nsteps = (nchunks + KG - 1) / KG;
    if (nmine > 0) { issue(0, kg); load_w(kg, wr[0]); }
#ifdef EEM_G16_STAMPS
    st[1] = G16_T();
#define G16_ACC(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = G16_T(); st[i] += now_ - tprev; tprev = now_; __builtin_amdgcn_sched_barrier(0); }
    unsigned long long tprev = st[1];
#else
#define G16_ACC(i)
#endif
#pragma unroll 1
    for (int step0 = 0; step0 < nsteps; step0 += 2) {
#pragma unroll
        for (int st_ = 0; st_ < 2; ++st_) {
            const int step = step0 + st_;
            if (step >= nsteps) break;
            g16_wait_vm<0>();
            landed(wr[st_]);
            G16_ACC(2)
            __builtin_amdgcn_s_barrier();                        // every wave is done with the other stage
            asm volatile("" ::: "memory");
            G16_ACC(3)
            if (step + 1 < nmine) { issue(1 - st_, kg + (step + 1) * KG); load_w(kg + (step + 1) * KG, wr[1 - st_]); }
            __builtin_amdgcn_sched_barrier(0);
            G16_ACC(4)
            if (KG == 1 || step < nmine) compute(st_, wr[st_]);
            G16_ACC(5)
        }
    }
#ifdef EEM_G16_STAMPS
    st[6] = G16_T();
#endif
    // ---- K groups: group 1 hands its partial sums over, group 0 finishes (the groups' LDS stages are dead by then)
    if (KG > 1) {
        __syncthreads();
        f32x4* red = reinterpret_cast<f32x4*>(lds_all);          // [wave][row][lane]
        if (kg == 1) {
#pragma unroll
            for (int t = 0; t < NR; ++t) red[(wave * NR + t) * 64 + lane] = acc[t];
        }
        __syncthreads();
        if (kg == 1) return;
#pragma unroll
        for (int t = 0; t < NR; ++t) acc[t] += red[(wave * NR + t) * 64 + lane];
    }

    // ---- epilogue: D[cout 4g + r][pixel j].  Its operands (GRU state and gate, residual, the per-pixel addend) are loaded
    // unconditionally from clamped indices, one operand's NR x 4 values as one batch: a load inside a lane-dependent branch is followed
    // by the compiler's s_waitcnt vmcnt(0), which made the GRU epilogue up to 2 x NR x 4 dependent round trips (8.5 k of a wave's 84 k
    // cycles in the stamps above)
    const int x = x0 + j;
    if (a.epi == GEPI_PLAIN && a.pre == nullptr) {               // (wave-uniform) nothing to load: the plain form
        if (x < a.wout) {
#pragma unroll
            for (int t = 0; t < NR; ++t) {
                const int y = y0 + wp * NR + t;
                if (y >= a.hout) continue;
                const int p = y * a.wout + x;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = mtg * 16 + 4 * g + r;
                    if (co >= a.cout) continue;
                    const float v = g16_act(acc[t][r] * e_scale[r] + e_shift[r], a.act);
                    const int oc = a.out_coff + grp * a.g_ocoff + co * (a.out_cmul > 1 ? a.out_cmul : 1);
                    a.out[((size_t)n * a.out_ctotal + oc) * hwo + p] = v * a.out_scale;
                }
            }
        }
    } else {
        unsigned ip[NR][4];                                      // (co, pixel) part of an operand's index (gconv16_supported: < 2^31); ~0u = outside
#pragma unroll
        for (int t = 0; t < NR; ++t) {
            const int y = y0 + wp * NR + t;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = mtg * 16 + 4 * g + r;
                const bool in = x < a.wout && y < a.hout && co < a.cout;
                ip[t][r] = in ? (unsigned)(co * hwo + y * a.wout + x) : ~0u;
            }
        }
        float e0v[NR][4], e1v[NR][4], prv[NR][4];
#pragma unroll
        for (int t = 0; t < NR; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { e0v[t][r] = 0.f; e1v[t][r] = 0.f; prv[t][r] = 0.f; }
        if (a.pre) {
            const float* b = a.pre + ((size_t)n * a.pre_ctotal + a.pre_coff) * hwo;
#pragma unroll
            for (int t = 0; t < NR; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) prv[t][r] = b[ip[t][r] == ~0u ? 0u : ip[t][r]];
        }
        const unsigned zsplit = a.epi == GEPI_ZR ? (unsigned)(a.split * hwo) : 0u;   // GEPI_ZR: e0 is indexed by co - split, from split on
        if (a.epi != GEPI_PLAIN) {
            const float* b = a.e0 + ((size_t)n * a.e0_ctotal + a.e0_coff) * hwo;
#pragma unroll
            for (int t = 0; t < NR; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) e0v[t][r] = b[(ip[t][r] == ~0u || ip[t][r] < zsplit) ? 0u : ip[t][r] - zsplit];
        }
        if (a.epi == GEPI_GRU) {
            const float* b = a.e1 + ((size_t)n * a.e1_ctotal + a.e1_coff) * hwo;
#pragma unroll
            for (int t = 0; t < NR; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) e1v[t][r] = b[ip[t][r] == ~0u ? 0u : ip[t][r]];
        }
#pragma unroll
        for (int t = 0; t < NR; ++t) {
            const int p = (y0 + wp * NR + t) * a.wout + x;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (ip[t][r] == ~0u) continue;
                const int co = mtg * 16 + 4 * g + r;
                float v = acc[t][r] * e_scale[r] + e_shift[r] + prv[t][r];
                v = g16_act(v, a.act);
                if (a.epi == GEPI_MUL) {
                    v *= e0v[t][r];
                } else if (a.epi == GEPI_GRU) {
                    v = (1.f - e1v[t][r]) * e0v[t][r] + e1v[t][r] * v;
                } else if (a.epi == GEPI_ADD_RELU) {
                    v += e0v[t][r];
                    v = v > 0.f ? v : 0.f;
                } else if (a.epi == GEPI_ADD) {
                    v += e0v[t][r];
                }
                const int oc = a.out_coff + grp * a.g_ocoff + co * (a.out_cmul > 1 ? a.out_cmul : 1);
                float* dst = a.out + ((size_t)n * a.out_ctotal + oc) * hwo + p;
                // GEPI_ZR: r leaves as r * h, to the second output.  One store through a selected pointer: written as a second store in the
                // else-if chain above (followed by `continue`), hipcc's code stored through `out2` on the other epilogues' paths too (memory
                // fault in <3,3,5,4,1> with GEPI_ADD_RELU; round 3's GEPI_SPLIT_MUL attempt had died the same way)
                if (a.epi == GEPI_ZR && co >= a.split) {
                    v *= e0v[t][r];
                    dst = a.out2 + ((size_t)n * a.out2_ctotal + (co - a.split)) * hwo + p;
                }
                *dst = v * a.out_scale;
            }
        }
    }
#ifdef EEM_G16_STAMPS
    if (lane == 0) {
        st[7] = G16_T();
        const unsigned bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (bid < 1024)
            for (int i = 0; i < 8; ++i) g_g16_stamps[(bid * 4 + wave) * 8 + i] = st[i];
    }
#endif
}

template <int KH, int KW, int THT, int WM>
int launch_wm(const GConvArgs& a, const float* wpk16, const float* zero_page, hipStream_t stream) {
    using C = G16Cfg<KH, KW, THT>;
    int cin = 0;
    for (int s = 0; s < a.nseg; ++s) cin += a.seg[s].c;
    const int tiles_x = ceil_div(a.wout, C::TW), tiles_y = ceil_div(a.hout, C::TH);
    const int ng = a.groups > 1 ? a.groups : 1;
    dim3 grid(tiles_x * tiles_y, ceil_div(a.cout, 16 * WM) * ng, a.n);
    // at most one block per CU: two K groups of four waves per tile (see the kernel's KG)
    static const int kg_env = [] { const char* e = getenv("EEM_G16_KG"); return e ? atoi(e) : 0; }();
    static const int cus = [] { int d = 0, n = 256; hipDeviceProp_t p; if (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) n = p.multiProcessorCount; return n > 0 ? n : 256; }();
    const int nchunks = cin / 16;
    const long blocks = (long)grid.x * grid.y * grid.z;
    const bool two = nchunks >= 4 && (kg_env ? kg_env == 2 : (blocks <= cus && a.in_flight < 3));
    if (two) hipLaunchKernelGGL((gconv16_kernel<KH, KW, THT, WM, 2>), grid, dim3(512), 0, stream, a, wpk16, zero_page, tiles_x, nchunks);
    else hipLaunchKernelGGL((gconv16_kernel<KH, KW, THT, WM, 1>), grid, dim3(256), 0, stream, a, wpk16, zero_page, tiles_x, nchunks);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

template <int KH, int KW, int THT>
int launch_th(const GConvArgs& a, const float* wpk16, const float* zero_page, hipStream_t stream) {
    if (a.cout <= 16) return launch_wm<KH, KW, 4, 1>(a, wpk16, zero_page, stream);
    if (a.cout <= 32) return launch_wm<KH, KW, THT, 2>(a, wpk16, zero_page, stream);
    return launch_wm<KH, KW, THT, 4>(a, wpk16, zero_page, stream);
}

// stride 2 (3x3 pad 1, 1x1 pad 0): 64-cout chunks of 4-row tiles, one K group
template <int KH, int KW>
int launch_s2(const GConvArgs& a, const float* wpk16, const float* zero_page, hipStream_t stream) {
    using C = G16Cfg<KH, KW, 4, 2>;
    int cin = 0;
    for (int s = 0; s < a.nseg; ++s) cin += a.seg[s].c;
    const int tiles_x = ceil_div(a.wout, C::TW), tiles_y = ceil_div(a.hout, C::TH);
    dim3 grid(tiles_x * tiles_y, ceil_div(a.cout, 64), a.n);
    hipLaunchKernelGGL((gconv16_kernel<KH, KW, 4, 4, 1, 2>), grid, dim3(256), 0, stream, a, wpk16, zero_page, tiles_x, cin / 16);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

template <int KH, int KW>
int launch(const GConvArgs& a, const float* wpk16, const float* zero_page, hipStream_t stream) {
    static const int th_env = [] { const char* e = getenv("EEM_G16_TH"); return e ? atoi(e) : 0; }();
    const auto blocks_of = [&](int th) { return (long)ceil_div(a.wout, 16) * ceil_div(a.hout, th) * ceil_div(a.cout, 64) * a.n * (a.groups > 1 ? a.groups : 1); };
    int th;
    if (th_env) {
        th = th_env;
    } else if (a.in_flight >= 3) {
        // with several frames in flight the other frames fill the CUs a short launch leaves idle, and what counts is CU time: 4-row tiles
        // read every weight fragment half as often (E-RAFT batch 4, three in flight: 168 -> 175 frames/s; one at a time 144 -> 138)
        th = blocks_of(4) < 512 ? 2 : 4;
    } else if (blocks_of(4) >= 2048) {
        th = 4;
    } else if (a.cout <= 32) {
        th = 2;
    } else {
        // A launch of a few blocks per CU runs as long as the CU with the most blocks: E-RAFT's update block at batch 1 (60 x 80 x 128
        // couts) is 300 blocks of 2-row tiles - 44 CUs take two and the other 212 wait for them; 200 blocks of 3-row tiles are one round
        // of 1.5x the work.  Rows per tile = the one with the least (rounds x rows), the per-block overhead counted as 0.6 rows.
        static const int cus = [] { int d = 0, n = 256; hipDeviceProp_t p; if (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) n = p.multiProcessorCount; return n > 0 ? n : 256; }();
        float best = 1e30f;
        th = 2;
        for (int t = 2; t <= 6; ++t) {
            const float cost = (float)ceil_div((int)blocks_of(t), cus) * ((float)t + 0.6f);
            if (cost < best - 1e-3f) { best = cost; th = t; }
        }
    }
    if (a.cout <= 32 && th != 2) th = 4;                         // the 32- and 16-cout forms split the rows over the waves: even tiles
    switch (th) {
        case 2: return launch_th<KH, KW, 2>(a, wpk16, zero_page, stream);
        case 3: return launch_wm<KH, KW, 3, 4>(a, wpk16, zero_page, stream);
        case 5: return launch_wm<KH, KW, 5, 4>(a, wpk16, zero_page, stream);
        case 6: return launch_wm<KH, KW, 6, 4>(a, wpk16, zero_page, stream);
        default: return launch_th<KH, KW, 4>(a, wpk16, zero_page, stream);
    }
}

}  // namespace

#ifdef EEM_G16_STAMPS
extern "C" __attribute__((visibility("default"))) int eemflow_debug_read_g16_stamps(unsigned long long* dst, size_t n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_g16_stamps), n * sizeof(unsigned long long)) == hipSuccess ? 0 : 2;
}
#endif

bool gconv16_shape(int cout, const int* cs, int nseg, int kh, int kw, int stride) {
    if ((stride != 1 && stride != 2) || cout < 16) return false;
    if (stride == 2 && !(((kh == 1 && kw == 1) || (kh == 3 && kw == 3)) && nseg == 1 && cout > 32)) return false;   // the encoders' downsampling convs
    if (!((kh == 1 && kw == 1) || (kh == 3 && kw == 3) || (kh == 1 && kw == 5) || (kh == 5 && kw == 1))) return false;
    for (int s = 0; s < nseg; ++s)
        if (cs[s] <= 0 || cs[s] % 16) return false;
    return true;
}

size_t gconv16_packed_floats(int cout, const int* cs, int nseg, int kh, int kw) {
    int cin = 0;
    for (int s = 0; s < nseg; ++s) cin += cs[s];
    return (size_t)ceil_div(cout, 64) * (cin / 16) * kh * kw * 4 * 4 * 64;
}

// stream[((((cc * nchunks + ch) * 4 + mt) * taps + tap) * 64 + lane) * 4 + cg] = W[cc*64 + mt*16 + lane%16][ch*16 + 4*cg + lane/16][tap]
void gconv16_pack(const float* w, int cout, const int* cs, int nseg, int kh, int kw, float* packed) {
    int cin = 0;
    for (int s = 0; s < nseg; ++s) cin += cs[s];
    const int taps = kh * kw, nch = cin / 16;
    for (int cc = 0; cc < ceil_div(cout, 64); ++cc)
        for (int ch = 0; ch < nch; ++ch)
            for (int tap = 0; tap < taps; ++tap)
                for (int cg = 0; cg < 4; ++cg)
                    for (int mt = 0; mt < 4; ++mt)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int co = cc * 64 + mt * 16 + (lane & 15), c = ch * 16 + 4 * cg + (lane >> 4);
                            packed[((((((size_t)cc * nch + ch) * 4 + mt) * taps + tap) * 64) + lane) * 4 + cg] =
                                co < cout ? w[((size_t)co * cin + c) * taps + tap] : 0.f;
                        }
}

bool gconv16_supported(const GConvArgs& a) {
    const char* e = getenv("EEM_NO_GCONV16");                    // read per call: a test flips it inside one process
    if (e && e[0] == '1') return false;
    if (!a.wpk16 || !a.zero_page || a.tstride > 1 || a.pad_h != a.kh / 2 || a.pad_w != a.kw / 2) return false;
    int cs[3];
    for (int s = 0; s < a.nseg; ++s) {
        cs[s] = a.seg[s].c;
        if (a.seg[s].cmul > 1 || a.seg[s].gate || ((uintptr_t)a.seg[s].ptr & 15)) return false;
    }
    int cin = 0;
    for (int s = 0; s < a.nseg; ++s) cin += cs[s];
    // measured on E-RAFT (640x480, batch 1 / 4) and EEMFlow+ (1280x720): shallow inputs and launches of a few dozen blocks
    // stay on the generic kernel's split-K form (128 blocks before the K groups and the tile-row choice, 12 with them)
    static const int min_cin = [] { const char* m = getenv("EEM_G16_MINCIN"); return m ? atoi(m) : 32; }();
    static const long min_blk = [] { const char* m = getenv("EEM_G16_MINBLK"); return m ? atol(m) : 12L; }();
    const long blocks = (long)ceil_div(a.wout, 16) * ceil_div(a.hout, 4) * ceil_div(a.cout, 64) * a.n * (a.groups > 1 ? a.groups : 1);
    if (cin < min_cin || blocks < min_blk) return false;
    if (a.groups > 1 && (a.nseg != 1 || a.epi != GEPI_PLAIN)) return false;
    if (a.stride == 2) {                                         // downsampling form: plain epilogues, one segment, no groups
        const char* m = getenv("EEM_NO_G16_S2");                 // read per call, like EEM_NO_GCONV16
        if ((m && m[0] == '1') || a.groups > 1 || a.pre || (a.epi != GEPI_PLAIN && a.epi != GEPI_ADD_RELU && a.epi != GEPI_ADD)) return false;
        return gconv16_shape(a.cout, cs, a.nseg, a.kh, a.kw, 2) && a.win % 4 == 0 && a.hout == (a.hin + 2 * a.pad_h - a.kh) / 2 + 1 &&
               a.wout == (a.win + 2 * a.pad_w - a.kw) / 2 + 1 && (size_t)16 * a.hin * a.win * 4 < (1u << 31) &&
               (size_t)a.cout * a.hout * a.wout < (1u << 31);
    }
    if ((a.epi != GEPI_PLAIN || a.pre) && (size_t)a.cout * a.hin * a.win >= (1u << 31)) return false;   // 32-bit operand indices in the epilogue
    return gconv16_shape(a.cout, cs, a.nseg, a.kh, a.kw, a.stride) && a.win % 4 == 0 && a.hout == a.hin && a.wout == a.win &&
           (size_t)16 * a.hin * a.win * 4 < (1u << 31);
}

int gconv16_launch(const GConvArgs& a, hipStream_t stream) {
    if (a.stride == 2) return a.kh == 1 ? launch_s2<1, 1>(a, a.wpk16, a.zero_page, stream) : launch_s2<3, 3>(a, a.wpk16, a.zero_page, stream);
    if (a.kh == 1 && a.kw == 1) return launch<1, 1>(a, a.wpk16, a.zero_page, stream);
    if (a.kh == 3) return launch<3, 3>(a, a.wpk16, a.zero_page, stream);
    if (a.kh == 1) return launch<1, 5>(a, a.wpk16, a.zero_page, stream);
    return launch<5, 1>(a, a.wpk16, a.zero_page, stream);
}
