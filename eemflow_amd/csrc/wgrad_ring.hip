// Weight gradients of 3x3 / 1x5 / 5x1 convs (the transposes of EEMFlow.py:26-30,75-82, model/update.py:33-60, model/extractor.py:7-57
// under autograd; train_mvsec.py:253-258 runs them), rebuilt around OPERAND DELIVERY (round 6):
//     dW[co][ci][ky][kx] = sum_{n, oy, ox} G[n][co][oy][ox] * X[n][ci][oy*S + ky - PH][ox*S + kx - PW]
// = a GEMM  dW[M = cout][N = (ci, tap)] = G[M][K = pixels] * X[K][N]  whose K runs over every output pixel of the batch.
//
// What wgrad_enc.hip measured (profiles/r05_wgrad_nomfma.txt): a 128-pixel tile per block and iteration behind vmcnt(0) + barrier keeps
// 60-70 % of its time with the MFMAs removed - 37 FLOP per staged byte at 16 input channels per block asks 4.2 TB/s of LDS-DMA at the fp32
// MFMA peak, G is staged once per 16-channel input chunk and the 3x3 halo of a 4-row tile re-reads X 1.9 x.  Here:
//   * ONE 8-wave block per CU owns a SEGMENT: a run of output rows of one image (strip of TW columns, TW chosen per launch; whole rows
//     where they fit) x up to 64 couts x up to 64 input channels.  It walks the segment row by row ("slice" = one output row of TW pixels):
//     the G row of the slice and the S new X rows enter LDS once and are used by every (cout, cin, tap) of the block - no vertical halo
//     inside a segment (KH - S rows at its start), horizontal halo (S TW + 8) / (S TW), G staged once for all input chunks of the block:
//     at 64 x 64 channels 140 FLOP per staged byte (1.1 TB/s at the MFMA peak instead of 4.2);
//   * G slices and X rows live in two LDS RINGS filled by 16-byte LDS-DMA (buffer_load ... lds) D - 1 slices ahead of the MFMAs:
//     counted s_waitcnt vmcnt(N) (N = the DMA instructions THIS wave issued for the slices that may stay in flight), one raw
//     s_barrier per slice, the X ring indexed by input row modulo (D - 1) S + KH so rows are shared by consecutive slices;
//   * wave roles: wave = (16-channel input chunk, cout half, K part).  A wave keeps MW x TAPS accumulator tiles (16x16x4 fp32 MFMA:
//     M = 16 couts, N = 16 (ci, tap) columns, K = 4 pixels) for the whole segment; K parts (narrow layers) split the slice's pixels and
//     meet in LDS once, at the end; one fp32 atomic per weight and block then (chip-wide atomic rate 1.3 TB/s: MI355X_MICROARCH.md -
//     which is why a block's (cout, cin) extent is a launch choice, not always 64 x 64);
//   * out-of-image rows / columns (the conv's zero padding, ragged strips, channel remainders) are DMA pieces whose offset lies beyond
//     the buffer descriptor's range - the hardware writes zeros, no branches; a row outside the image is a descriptor of zero records.
// Bias gradient (row sums of G) rides in the waves of input chunk 0.
#include "common.h"
#include "train.h"

namespace {

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

struct RingGeom {
    int tw;              // output pixels per slice (multiple of 4 * WK)
    int gp, gq;          // G row pitch in LDS (floats; gp / 4 odd: conflict-free A reads) and 16-byte pieces per row
    int xp, xq;          // X channel-row pitch (floats) = S tw + 8, pieces per channel row
    int rp;              // X ring row pitch (floats) = CW xp + pad
    int gslot;           // floats per G slot (whole DMA instructions)
    int xbase;           // float offset of the X ring
    int nstrips, rows, segs_y, nseg;
    int chunks, pairs;   // input-channel chunks of a block's CW channels; chunk pairs = chunks x cout chunks
};

template <int MT_, int CT_, int S_, int KH_, int KW_, int D_, int WM_, int WK_, int TWMAX_>
struct RingCfg {
    static constexpr int MT = MT_, CT = CT_, S = S_, KH = KH_, KW = KW_, D = D_, WM = WM_, WK = WK_, TWMAX = TWMAX_;
    static constexpr int TAPS = KH * KW, PH = KH / 2, PW = KW / 2;
    static constexpr int WN = CT;
    static_assert(WN * WM * WK == 8, "eight waves");
    static_assert(MT % WM == 0, "cout tiles split evenly");
    static constexpr int MW = MT / WM;                         // cout tiles per wave
    static constexpr int NT = TAPS;                            // 16 input channels x TAPS = TAPS N-tiles of 16
    static constexpr int NR = (D - 1) * S + KH;                // X ring rows
    static constexpr int CW = CT * 16;
    static constexpr int GQMAX = (TWMAX + 4) / 4, XQMAX = (S * TWMAX + 8) / 4;
    static constexpr int NGI = (MT * 16 * GQMAX + 511) / 512;  // DMA instructions per wave: G slice, one X row
    static constexpr int NXI = (CW * XQMAX + 511) / 512;
    static constexpr int RED = WK > 1 ? WN * WM * MW * NT * 256 + 64 : 64;   // floats of the K-part reduction buffer (+ the bias sums)
};

// s_waitcnt vmcnt(n) for a run-time n (the instruction takes an immediate): n is wave-uniform, <= 24 here
__device__ __forceinline__ void wait_vm(int n) {
    switch (n) {
#define WVM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        WVM(0) WVM(1) WVM(2) WVM(3) WVM(4) WVM(5) WVM(6) WVM(7) WVM(8) WVM(9) WVM(10) WVM(11) WVM(12) WVM(13) WVM(14) WVM(15) WVM(16)
        WVM(17) WVM(18) WVM(19) WVM(20) WVM(21) WVM(22) WVM(23) WVM(24) WVM(25) WVM(26) WVM(27) WVM(28) WVM(29) WVM(30) WVM(31) WVM(32)
#undef WVM
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;       // waiting for more than asked is always safe
    }
}

#ifdef EEM_DIAG
__device__ int g_wr_dbg;          // diagnostic builds: EEM_WG_DBG bit 0 leaves the compute phase out, bit 1 the atomics, bit 2 the DMA, bit 3 the
                                  // operand reads (MFMAs alone), bit 4 the MFMAs (operand reads alone)
#endif

template <class C>
__global__ __launch_bounds__(512) void wgrad_ring_kernel(WgradArgs a, RingGeom q) {
    constexpr int MT = C::MT, S = C::S, KH = C::KH, KW = C::KW, D = C::D, WM = C::WM, WK = C::WK, WN = C::WN;
    constexpr int TAPS = C::TAPS, PH = C::PH, PW = C::PW, MW = C::MW, NT = C::NT, NR = C::NR, CW = C::CW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int wn = wave % WN, wm = (wave / WN) % WM, wk = wave / (WN * WM);

    // one-dimensional grid over (segment, chunk pair), dealt evenly over the XCDs: XCD x owns the contiguous logical range
    // [x cpx, (x + 1) cpx) - the chunk pairs of a segment (they share its G rows / X rows) and neighbouring segments (halo rows) meet in
    // one L2, and no XCD gets more blocks than CUs while another has idle ones (a (segments, cin chunks, cout chunks) grid with
    // segments padded to a multiple of 8 left four XCDs with 36 blocks for 32 CUs: two rounds)
    const int lt = (int)xcd_logical_block(blockIdx.x, gridDim.x);
    if (lt >= q.nseg * q.pairs) return;
    const int seg = lt / q.pairs, pair = lt - seg * q.pairs;
    const int chunk_ci = pair % q.chunks, chunk_co = pair / q.chunks;
    const int sy = seg % q.segs_y;
    const int t0 = seg / q.segs_y;
    const int strip = t0 % q.nstrips, n = t0 / q.nstrips;
    const int y0 = sy * q.rows;
    const int nsl = min(q.rows, a.hout - y0);                     // slices (output rows) of this segment
    const int x0 = strip * q.tw;
    // the block's input chunk: chunk index -> (input segment, chunk inside it); a chunk never straddles two segments
    const float* xptr = a.x;
    int x_ct = a.x_ctotal, x_co = a.x_coff, seg_c = a.cin, seg_start = 0, local = chunk_ci;
    if (a.nxseg > 0) {
        x_co = 0;
        int rem = chunk_ci;
        bool found = false;
#pragma unroll
        for (int sgi = 0; sgi < 3; ++sgi) {
            if (sgi < a.nxseg && !found) {
                const int nch = (a.xsc[sgi] + CW - 1) / CW;
                if (rem < nch) { xptr = a.xs[sgi]; x_ct = seg_c = a.xsc[sgi]; local = rem; found = true; }
                else { rem -= nch; seg_start += a.xsc[sgi]; }
            }
        }
    }
    (void)xptr; (void)x_ct; (void)x_co;                          // (read by the device pass only)
    const int ci_loc = local * CW;                                // first channel of the chunk inside its segment
    const int ci_base = seg_start + ci_loc, co_base = chunk_co * MT * 16;      // ... and inside the concatenated input (dW's column)
    const int cin_here = min(CW, seg_c - ci_loc), cout_here = min(MT * 16, a.cout - co_base);
    const size_t ghw = (size_t)a.hout * a.wout, xhw = (size_t)a.hin * a.win;

    // ---- DMA plan: instruction k of this wave moves pieces [(8 k + wave) 64, +64); a lane's byte offset from the row's base pointer is
    // fixed for the whole segment (the base moves, a scalar); pieces that must read as zero carry an offset beyond the descriptor's range
    constexpr unsigned RANGE = 0x7fffff00u;
    unsigned goff[C::NGI], xoff[C::NXI];
    int gcnt = 0, xcnt = 0;
    {
        const int ngp = MT * 16 * q.gq, nxp = CW * q.xq;
#pragma unroll
        for (int k = 0; k < C::NGI; ++k) {
            const int f0 = (8 * k + wave) * 64, f = f0 + lane;
            if (f0 < ngp) gcnt = k + 1;
            const int co = f / q.gq, qq = f - co * q.gq;
            const bool ok = f < ngp && 4 * qq < q.tw && co < cout_here && x0 + 4 * qq < a.wout;
            goff[k] = ok ? (unsigned)(((size_t)co * ghw + 4 * qq) * 4) : RANGE;
        }
#pragma unroll
        for (int k = 0; k < C::NXI; ++k) {
            const int f0 = (8 * k + wave) * 64, f = f0 + lane;
            if (f0 < nxp) xcnt = k + 1;
            const int ci = f / q.xq, qq = f - ci * q.xq;
            const int ix = x0 * S - 4 + 4 * qq;
            const bool ok = f < nxp && ci < cin_here && ix >= 0 && ix + 4 <= a.win;
            xoff[k] = ok ? (unsigned)(((size_t)ci * xhw + 4 * qq) * 4) : RANGE;
        }
    }
    const int nw = gcnt + S * xcnt;                               // this wave's DMA instructions per regular slice

#if __HIP_DEVICE_COMPILE__
    const char* gimg = reinterpret_cast<const char*>(a.g + ((size_t)n * a.g_ctotal + a.g_coff + co_base) * ghw) + (long)x0 * 4;
    const char* ximg = reinterpret_cast<const char*>(xptr + ((size_t)n * x_ct + x_co + ci_loc) * xhw) + ((long)x0 * S - 4) * 4;
    auto issue_g = [&](int slot, int y) {                         // y past the segment: a slice of zeros (keeps every slice's count equal)
        const bool valid = y < y0 + nsl;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(gimg + (long)y * a.wout * 4), (short)0,
                                                                           valid ? (int)RANGE : 0, 0x00020000);
        float* dst = lds + slot * q.gslot;
#pragma unroll
        for (int k = 0; k < C::NGI; ++k)
            if (k < gcnt) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(dst + (8 * k + wave) * 256), 16, goff[k], 0, 0, 0);
    };
    auto issue_x = [&](int rslot, int iy, bool live) {
        const bool valid = live && iy >= 0 && iy < a.hin;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ximg + (long)iy * a.win * 4), (short)0,
                                                                           valid ? (int)RANGE : 0, 0x00020000);
        float* dst = lds + q.xbase + rslot * q.rp;
#pragma unroll
        for (int k = 0; k < C::NXI; ++k)
            if (k < xcnt) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(dst + (8 * k + wave) * 256), 16, xoff[k], 0, 0, 0);
    };
#else
    auto issue_g = [&](int, int) {};
    auto issue_x = [&](int, int, bool) {};
#endif

    // ---- operand addresses (bytes).  A: lane (j, g) reads G[cout tile row j][pixel 4 s + g]; B: X[(ci, tap) column j][pixel 4 s + g]
    const int lds_base = (int)(unsigned)(uintptr_t)LDS_PTR(lds);   // (0 unless static LDS sits in front)
    const int p0 = wk * (q.tw / WK);                              // this wave's pixels of a slice: [p0, p0 + tw / WK)
    const int ns = q.tw / (4 * WK);                               // k-steps per slice and wave
    int a_st[MW];
#pragma unroll
    for (int m = 0; m < MW; ++m) a_st[m] = (((wm * MW + m) * 16 + j) * q.gp + p0 + g) * 4;
    int b_st[NT], b_ky[NT];
    const int nvalid = max(0, min(16, cin_here - wn * 16)) * TAPS;   // columns of this wave that exist
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        int nn = nt * 16 + j;
        nn = nn < nvalid ? nn : 0;                                // columns past the channels are never written back
        const int ci = nn / TAPS, tap = nn - ci * TAPS;
        const int ky = tap / KW, kx = tap - ky * KW;
        b_st[nt] = ((wn * 16 + ci) * q.xp + (p0 + g) * S + kx + 4 - PW + q.xbase) * 4 + lds_base;
        b_ky[nt] = ky * q.rp * 4;                                 // byte offset of tap row ky from the slice's first ring row
    }
    const int ring_bytes = NR * q.rp * 4, step_bytes = S * q.rp * 4;

    f32x4 acc[MW][NT];
    float bsum[MW];
#pragma unroll
    for (int m = 0; m < MW; ++m) {
        bsum[m] = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool idle = nvalid == 0;                                // a wave whose 16 input channels lie past the layer's: DMA and barriers only

#ifdef EEM_DIAG
    const int dbg = g_wr_dbg;
#else
    constexpr int dbg = 0;
#endif

    // ---- prologue: slices 0 .. D - 2 (slice 0 brings KH rows, every later one S)
    const int iy_first = y0 * S - PH;
    int next_row = 0;                                             // rows issued so far (relative to iy_first); ring slot = row mod NR
    int next_rslot = 0;
    auto issue_rows = [&](int count, bool live) {
        for (int r = 0; r < count; ++r) {
            issue_x(next_rslot, iy_first + next_row, live);
            ++next_row;
            if (++next_rslot == NR) next_rslot = 0;
        }
    };
    if (!(dbg & 4)) {
        issue_g(0, y0);
        issue_rows(KH, true);
#pragma unroll
        for (int d = 1; d < D - 1; ++d) {
            issue_g(d, y0 + d);
            issue_rows(S, d < nsl);
        }
    }
    int gs = 0;                                                   // G slot of slice i
    int gn = (D - 1) % D;                                         // G slot slice i + D - 1 goes to
    int rb = 0;                                                   // byte offset of the ring row of slice i's first input row
#pragma unroll 1
    for (int i = 0; i < nsl; ++i) {
        wait_vm((D - 2) * nw);                                    // this wave's pieces of slice i have landed
        __builtin_amdgcn_s_barrier();                             // everyone's have; everyone is done with slice i - 1
        asm volatile("" ::: "memory");
        if (!(dbg & 4)) {
            issue_g(gn, y0 + i + D - 1);
            issue_rows(S, i + D - 1 < nsl);
        }
        if (++gn == D) gn = 0;
        if (!idle && !(dbg & 1)) {
            int ao[MW], bo[NT];
#pragma unroll
            for (int m = 0; m < MW; ++m) ao[m] = a_st[m] + gs * q.gslot * 4 + lds_base;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                int t = rb + b_ky[nt];                            // (rb: byte offset of the slice's first ring row)
                t = t >= ring_bytes ? t - ring_bytes : t;
                bo[nt] = b_st[nt] + t;
            }
            float av[2][MW] = {}, bv[2][NT] = {};
            typedef const __attribute__((address_space(3))) float* lds_f;       // (32-bit LDS addresses: no generic-pointer add per read)
            auto load = [&](int buf, int off_a, int off_b) __attribute__((always_inline)) {
                if (dbg & 8) return;                              // (diagnostic builds: the MFMAs on whatever the registers hold)
#pragma unroll
                for (int m = 0; m < MW; ++m) av[buf][m] = *reinterpret_cast<lds_f>((unsigned)(ao[m] + off_a));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[buf][nt] = *reinterpret_cast<lds_f>((unsigned)(bo[nt] + off_b));
            };
            // "these operands have arrived": an empty asm that reads them.  The compiler's wait for values loaded in the previous turn of
            // the loop is lgkmcnt(0); placed here - BEFORE the next step's reads are issued, a whole MFMA phase after these were - it
            // costs nothing, and the MFMAs below then start without a wait that would also cover the reads just issued
            auto arrived = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
                for (int m = 0; m < MW; ++m) asm volatile("" ::"v"(av[buf][m]));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) asm volatile("" ::"v"(bv[buf][nt]));
            };
            // four k-steps per turn (immediate offsets 0 / 16 / 32 / 48 S bytes, one address advance per turn), the next step's operands
            // requested before this step's MFMAs (reads past the wave's last step fetch unused words inside the rings).  sched_barriers:
            // left alone, the scheduler sinks every read down to its MFMA - one exposed LDS round trip per k-step; the empty asm keeps
            // the addresses as registers that advance, instead of one add per read.  The row sums of G (bias gradient) ride in the
            // waves of input chunk 0 only: a VALU instruction beside fp32 MFMAs costs its full issue time
            auto kloop = [&](auto with_bias) __attribute__((always_inline)) {
                auto mul = [&](int buf) __attribute__((always_inline)) {
                    if (dbg & 16) return;                         // (diagnostic builds: the operand reads without the MFMAs)
#pragma unroll
                    for (int m = 0; m < MW; ++m) {
                        if (decltype(with_bias)::value) bsum[m] += av[buf][m];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][m], bv[buf][nt], acc[m][nt], 0, 0, 0);
                    }
                };
                auto phase = [&](int cur, int off_next) __attribute__((always_inline)) {
                    arrived(cur);
                    __builtin_amdgcn_sched_barrier(0);
                    load(cur ^ 1, off_next, off_next * S);
                    __builtin_amdgcn_sched_barrier(0);
                    mul(cur);
                    __builtin_amdgcn_sched_barrier(0);
                };
                load(0, 0, 0);
                int s = 0;
                while (true) {
                    phase(0, 16);
                    if (++s == ns) break;
                    phase(1, 32);
                    if (++s == ns) break;
                    phase(0, 48);
                    if (++s == ns) break;
#pragma unroll
                    for (int m = 0; m < MW; ++m) { ao[m] += 64; asm volatile("" : "+v"(ao[m])); }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) { bo[nt] += 64 * S; asm volatile("" : "+v"(bo[nt])); }
                    phase(1, 0);
                    if (++s == ns) break;
                }
            };
            if (wn == 0) kloop(std::true_type{});
            else kloop(std::false_type{});
        }
        if (++gs == D) gs = 0;
        rb += step_bytes;
        if (rb >= ring_bytes) rb -= ring_bytes;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (the zero slices issued past the end)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#ifdef EEM_DIAG
    if (dbg & 2) return;
#endif

    // ---- bias gradient (blocks of the first input chunk; the waves of input chunk 0): over the 4 pixel slots of a k-step, over the K parts
    // in LDS, then ONE atomic per cout and block (atomics of 8 waves x 256 blocks on 16 addresses serialise: 25 us at 16 couts)
    float* bred = lds + (WK > 1 ? WN * WM * MW * NT * 256 : 0);
    const bool want_bias = a.db && chunk_ci == 0;
    if (want_bias) {
        if (threadIdx.x < 64) bred[threadIdx.x] = 0.f;
        __syncthreads();
        if (wn == 0 && !idle) {
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                float v = bsum[m];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (g == 0) atomicAdd(&bred[(wm * MW + m) * 16 + j], v);     // (LDS: WK adders per cell)
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < cout_here) atomicAdd(&a.db[co_base + threadIdx.x], bred[threadIdx.x]);
    }
    // ---- weights: K parts meet in LDS (part 0 stores, the others add in turn: plain 16-byte read-modify-writes), then one atomic per
    // weight and block
    const int dwcin = a.dw_cin ? a.dw_cin : a.cin;
    auto flush = [&](const f32x4& v, int mrow, int nt, int l, int wn_) __attribute__((always_inline)) {
        const int nn = nt * 16 + (l & 15);
        const int nv = max(0, min(16, cin_here - wn_ * 16)) * TAPS;
        if (nn >= nv) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) {                             // D[co = 4 (l / 16) + r][n = l % 16]
            const int co = mrow * 16 + 4 * (l >> 4) + r;
            if (co < cout_here)
                atomicAdd(&a.dw[((size_t)(co_base + co) * dwcin + a.dw_coff + ci_base + wn_ * 16) * TAPS + nn], v[r]);
        }
    };
    if (WK == 1) {
        if (!idle) {
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) flush(acc[m][nt], wm * MW + m, nt, lane, wn);
        }
        return;
    }
    f32x4* red = reinterpret_cast<f32x4*>(lds);                   // (the bias cells lie behind this buffer)
    const int grp = wn * WM + wm;
#pragma unroll 1
    for (int w = 0; w < WK; ++w) {
        if (wk == w && !idle) {
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    f32x4* cell = red + ((grp * MW + m) * NT + nt) * 64 + lane;
                    *cell = w == 0 ? acc[m][nt] : *cell + acc[m][nt];
                }
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < WN * WM * MW * NT * 64; e += 512) {
        const int l = e & 63, tile = e >> 6;
        const int nt = tile % NT, gm = tile / NT;
        const int m = gm % MW, gr = gm / MW;
        const int wm_ = gr % WM, wn_ = gr / WM;
        if (min(16, cin_here - wn_ * 16) <= 0) continue;
        flush(red[e], wm_ * MW + m, nt, l, wn_);
    }
}

// ------------------------------------------------------------------------------------------------ host side
template <class C>
bool geometry(const WgradArgs& a, RingGeom* out, int* lds_bytes, int* gx, int* chunks, int* cochunks) {
    constexpr int S = C::S, WK = C::WK, D = C::D, NR = C::NR, CW = C::CW, MT = C::MT;
    const int step = 4 * WK;
    // strip width: fewest padded pixels, then the widest (fewer strips = less horizontal halo); LDS must hold the rings
    int best = 0, best_cost = 0;
    RingGeom bg{};
    int best_lds = 0;
    for (int tw = step; tw <= C::TWMAX; tw += step) {
        RingGeom q{};
        q.tw = tw;
        q.gq = tw / 4 + ((tw / 4) % 2 == 0 ? 1 : 0);                      // odd piece count per row: A-operand reads hit every bank twice
        q.gp = q.gq * 4;
        q.xp = S * tw + 8;
        q.xq = q.xp / 4;
        q.rp = (CW * q.xp + 255) / 256 * 256 + 16;                        // whole DMA instructions per row, rows 16 banks apart
        q.gslot = (MT * 16 * q.gq * 4 + 255) / 256 * 256;
        q.xbase = D * q.gslot;
        const int bytes = (q.xbase + NR * q.rp) * 4;
        if (bytes > 160 * 1024) continue;
        const int ns = ceil_div(a.wout, tw);
        const int cost = ns * tw;
        if (best == 0 || cost < best_cost || (cost == best_cost && tw > best)) { best = tw; best_cost = cost; bg = q; best_lds = bytes; }
    }
    if (best == 0) return false;
    bg.nstrips = ceil_div(a.wout, best);
    *chunks = 0;
    if (a.nxseg > 0) for (int sgi = 0; sgi < a.nxseg; ++sgi) *chunks += ceil_div(a.xsc[sgi], CW);
    else *chunks = ceil_div(a.cin, CW);
    *cochunks = ceil_div(a.cout, MT * 16);
    const int pairs = *chunks * *cochunks;
    const int units = a.n * bg.nstrips;
    // one block per CU (the rings take most of its LDS): segments per image column by a small cost model - rounds of 256 blocks x
    // (rows of a segment + its KH - S halo rows + ~3 rows' worth of prologue, K-part reduction and atomics)
    int best_sy = 1;
    double best_t = 0;
    for (int sy = 1; sy <= a.hout; ++sy) {
        const int rows = ceil_div(a.hout, sy);
        if (ceil_div(a.hout, rows) != sy) continue;                       // (the same segmentation as a smaller sy)
        const long blocks = (long)pairs * units * sy;
        const double t = (double)ceil_div((int)blocks, 256) * (rows + C::KH - S + 3);
        if (sy == 1 || t < best_t) { best_t = t; best_sy = sy; }
        if (blocks >= 1024) break;
    }
    bg.rows = ceil_div(a.hout, best_sy);
    bg.segs_y = ceil_div(a.hout, bg.rows);
    bg.nseg = units * bg.segs_y;
    bg.chunks = *chunks;
    bg.pairs = pairs;
    *gx = (bg.nseg * pairs + 7) & ~7;
    *out = bg;
    *lds_bytes = best_lds > C::RED * 4 ? best_lds : C::RED * 4;            // (the epilogue's reduction buffer lies over the rings)
    return true;
}

template <class C>
int launch_ring(const WgradArgs& a, hipStream_t st) {
    RingGeom q;
    int lds_bytes = 0, gx = 0, chunks = 0, cochunks = 0;
    if (!geometry<C>(a, &q, &lds_bytes, &gx, &chunks, &cochunks)) {
        eem_set_error("wgrad_ring: no strip width fits LDS (wout %d)", a.wout);
        return EEM_ERR_ARG;
    }
#ifdef EEM_DIAG
    { static int once = [] { const char* e = getenv("EEM_WG_DBG"); int v = e ? atoi(e) : 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wr_dbg), &v, sizeof v); return v; }(); (void)once; }
#endif
    static bool raised = false;
    if (!raised) {
        EEM_HIP_CHECK(hipFuncSetAttribute((const void*)wgrad_ring_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised = true;
    }
    hipLaunchKernelGGL((wgrad_ring_kernel<C>), dim3(gx), dim3(512), lds_bytes, st, a, q);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

//                      MT CT S KH KW D WM WK TWMAX
using Cfg64x64 = RingCfg<4, 4, 1, 3, 3, 2, 2, 1, 96>;        // pconv3_2 / pconv3_3; E-RAFT's 3x3 layers in 64 x 64 chunks
using Cfg64x32s2 = RingCfg<4, 2, 2, 3, 3, 2, 2, 2, 80>;      // pconv3_1
using Cfg32x32 = RingCfg<2, 2, 1, 3, 3, 3, 1, 4, 144>;       // pconv2_2 / pconv2_3
using Cfg32x16s2 = RingCfg<2, 1, 2, 3, 3, 3, 1, 8, 96>;      // pconv2_1
using Cfg16x16 = RingCfg<1, 1, 1, 3, 3, 4, 1, 8, 224>;       // pconv1_2
using Cfg64x64r15 = RingCfg<4, 4, 1, 1, 5, 2, 2, 1, 128>;    // SepConvGRU's (1, 5) convs
using Cfg64x64r51 = RingCfg<4, 4, 1, 5, 1, 2, 2, 1, 64>;     // ... and its (5, 1) ones
using Cfg64x32r51 = RingCfg<4, 2, 1, 5, 1, 2, 2, 2, 96>;     // ... (5, 1) with whole rows of 80 (six ring rows of 32 channels fit LDS)
using Cfg64x64s2 = RingCfg<4, 4, 2, 3, 3, 2, 2, 1, 48>;      // the encoders' downsampling convs (model/extractor.py layer2 / layer3)

bool common_ok(const WgradArgs& a) {
    const char* e = getenv("EEM_NO_WGRAD_RING");                      // read per call: a test flips it inside one process
    if (e && e[0] == '1') return false;
    if (a.nxseg > 0) {
        for (int sgi = 0; sgi < a.nxseg; ++sgi)
            if (!a.xs[sgi] || ((uintptr_t)a.xs[sgi] & 15) || a.xsc[sgi] < 1) return false;
    } else if (((uintptr_t)a.x & 15) != 0) {
        return false;
    }
    return a.zero_page && a.gate == nullptr && a.g_cmul == 1 && a.wout % 4 == 0 && a.win % 4 == 0 && ((uintptr_t)a.g & 15) == 0 && ((size_t)a.hout * a.wout) % 4 == 0 && ((size_t)a.hin * a.win) % 4 == 0 &&
           (size_t)64 * a.hout * a.wout * 4 < (1u << 31) && (size_t)64 * a.hin * a.win * 4 < (1u << 31) && a.hout >= 1;
}

}  // namespace

// 3x3 (stride 1 / 2), 1x5 and 5x1 (stride 1) convs with "same" padding and at least 16 input channels: the encoder layers of the fused
// EEMFlow training step (16 -> 16, 16 -> 32 s2, 32 -> 32, 32 -> 64 s2, 64 -> 64) on blocks of their own extent, everything wider
// (E-RAFT's update block, heads and encoders, EEMFlow+ under autograd) in 64 x 64 chunks
bool wgrad_ring_supported(const WgradArgs& a) {
    if (!common_ok(a)) return false;
    const int kh = a.kh ? a.kh : a.k, kw = a.kh ? a.kw : a.k;
    const int ph = a.kh ? a.ph : a.pad, pw = a.kh ? a.pw : a.pad;
    if (ph != kh / 2 || pw != kw / 2 || a.cin < 16 || a.cout < 16) return false;
    if (a.stride == 1) return (kh == 3 && kw == 3) || (kh == 1 && kw == 5) || (kh == 5 && kw == 1);
    return a.stride == 2 && kh == 3 && kw == 3;
}

// Where the rings are the faster kernel (tools/wgrad_bench.py on the shapes a training step asks for, profiles/r06_wgrad_bench.txt): the
// encoders' stride-2 convs (3 x: the other path is the generic kernel).  For the stride-1 shapes the tile kernel of wgrad_enc.hip with its
// products as bf16 pieces (later in round 6) is faster alone - 73 against 120 us at EEMFlow's 64 -> 64 layer, 89 against 125 at E-RAFT's
// heads, 34 against 50 at a GRU conv's input segment - and in the E-RAFT step (81.4 ms against 82.7 with every wide layer on the rings);
// against the tile kernel's fp32 form the rings had won that step (89.5 against 93.8 ms).  What the rings buy - a third of the staged
// bytes per product - is what the bf16-piece multiply would need next: that kernel is now bound by its operands' way in.
// EEM_WGRAD_RING=all / none overrides (read per call).
bool wgrad_ring_preferred(const WgradArgs& a) {
    if (const char* e = getenv("EEM_WGRAD_RING")) {
        if (e[0] == 'a') return true;
        if (e[0] == 'n') return false;
    }
    return a.stride == 2 && a.cin >= 64;
}

int wgrad_ring_launch(const WgradArgs& a, hipStream_t st) {
    const int kh = a.kh ? a.kh : a.k;
    // EEM_WGRAD_RING_BLOCK=<couts><cins> (read per call; measurement): the block extent of stride-1 3x3 layers - 6464, 3232 or 1616
    if (const char* e = getenv("EEM_WGRAD_RING_BLOCK")) {
        const int v = atoi(e);
        if (a.stride == 1 && kh == 3 && (a.kh ? a.kw : a.k) == 3) {
            if (v == 6464) return launch_ring<Cfg64x64>(a, st);
            if (v == 3232) return launch_ring<Cfg32x32>(a, st);
            if (v == 1616) return launch_ring<Cfg16x16>(a, st);
        }
    }
    if (a.stride == 2) {
        if (a.cout <= 32 && a.cin <= 16) return launch_ring<Cfg32x16s2>(a, st);
        if (a.cout <= 64 && a.cin <= 32) return launch_ring<Cfg64x32s2>(a, st);
        return launch_ring<Cfg64x64s2>(a, st);
    }
    if (kh == 5) {
        const char* e = getenv("EEM_WGRAD_RING_51");                 // (measurement: 6464 = 64 input channels per block, strips of <= 64)
        if (e && atoi(e) == 6464) return launch_ring<Cfg64x64r51>(a, st);
        return a.wout > 64 ? launch_ring<Cfg64x32r51>(a, st) : launch_ring<Cfg64x64r51>(a, st);
    }
    if (kh == 1) return launch_ring<Cfg64x64r15>(a, st);
    if (a.cout <= 16 && a.cin <= 16) return launch_ring<Cfg16x16>(a, st);
    if (a.cout <= 32 && a.cin <= 32) return launch_ring<Cfg32x32>(a, st);
    return launch_ring<Cfg64x64>(a, st);
}
