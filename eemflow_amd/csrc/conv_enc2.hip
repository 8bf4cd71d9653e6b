// Fast path of the encoder convolutions (layers 2..8 of EEMFlow.py:75-82) for feature maps whose
// width is a multiple of 4 (always true for inputs padded to a multiple of 64).
//
// Same implicit GEMM as conv_enc.hip (D[cout][pixel] on the fp32 matrix cores), but the operands
// are streamed through LDS by the DMA path instead of through registers:
//   * the input is consumed in chunks of CK channels; a chunk's input tile [CK][rows][ROWP] (16-byte
//     aligned row segments incl. halo) AND its packed weight fragments are copied HBM/L2 -> LDS with
//     global_load_lds_dwordx4 (1 KiB per wave-instruction, per-lane source address, out-of-image
//     pieces read a zero page);
//   * two LDS stages: chunk c+1 is in flight while chunk c feeds the MFMAs (counted vmcnt + raw
//     s_barrier, one __shared__ array - see cdna_hip_programming.md "Pipelining across barriers");
//   * weights are read back as ds_read_b128 (4 k-steps per lane), input taps as ds_read_b32 with
//     compile-time immediates for every (tap, channel) displacement.
#include "common.h"

namespace {

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int CIN, int COUT, int STRIDE, int TH, int TWT, int WAVES, int WM, int CK>
struct Cfg2 {
    static constexpr bool M16 = (COUT == 16);
    static constexpr int NPIX = M16 ? 16 : 32;
    static constexpr int KPS = M16 ? 4 : 2;
    static constexpr int MT = M16 ? 1 : COUT / 32;
    static constexpr int ACC = M16 ? 4 : 16;
    static constexpr int TW = TWT * NPIX;
    static constexpr int IN_ROWS = (TH - 1) * STRIDE + 3;
    static constexpr int ROWP = ((TW - 1) * STRIDE + 6 + 3) / 4 * 4;     // floats per staged row (starts at x0*S-4)
    static constexpr int PPR = ROWP / 4;                                   // 16-byte pieces per row
    static constexpr int PLANE = IN_ROWS * ROWP;
    static constexpr int NCHUNK = CIN / CK;
    static constexpr int CG = CK / KPS;                                    // k-steps per tap in a chunk
    static constexpr int STEPS = 9 * CG;                                   // k-steps per chunk
    static constexpr int B_PIECES = CK * IN_ROWS * PPR;
    static constexpr int NB = (B_PIECES + 63) / 64;                        // wave-instructions for the input tile
    static constexpr int NA = MT * STEPS / 4;                              // wave-instructions for the weights
    static constexpr int NI = (NB + NA + WAVES - 1) / WAVES;               // per wave, per chunk (incl. dummies)
    static constexpr int STAGE = (NI * WAVES) * 256;                       // floats per LDS stage
    static constexpr int NSTAGE = NCHUNK > 1 ? 2 : 1;
    static constexpr int WP = WAVES / WM;                                  // wave groups along pixels
    static constexpr int UPW = TH * TWT / WP;                              // pixel tiles per wave
    static constexpr int MTW = MT / WM;                                    // cout tiles per wave
    static_assert(CIN % CK == 0 && CK % KPS == 0 && STEPS % 4 == 0, "chunking");
    static_assert((TH * TWT) % WP == 0 && MT % WM == 0 && WAVES % WM == 0, "wave split");
    static_assert(NSTAGE * STAGE * 4 <= 160 * 1024, "LDS budget");
};

template <bool M16> struct AccT2 { using type = f32x16; };
template <> struct AccT2<true> { using type = f32x4; };

template <int CIN, int COUT, int STRIDE, int TH, int TWT, int WAVES, int WM, int CK, int POOLK>
__global__ __launch_bounds__(WAVES * 64) void enc_conv2_kernel(EncConvArgs a) {
    ENC_ARGS_NOW(a);
    using C = Cfg2<CIN, COUT, STRIDE, TH, TWT, WAVES, WM, CK>;
    using acc_t = typename AccT2<C::M16>::type;
    __shared__ __attribute__((aligned(16))) float lds[C::NSTAGE * C::STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const unsigned lid = xcd_logical_block(blockIdx.x, gridDim.x);
    if (lid >= (unsigned)(a.tiles_x * a.tiles_y * a.nimg)) return;
    const int bx = lid % a.tiles_x, by = (lid / a.tiles_x) % a.tiles_y;
    const int n = lid / (a.tiles_x * a.tiles_y);
    const int oy0 = by * TH;
    const int ox0 = bx * C::TW;
    const int gy0 = oy0 * STRIDE - 1;
    const int gxa = ox0 * STRIDE - 4;                       // 16-byte aligned start column
    const float* src = a.in0 + (size_t)n * CIN * a.hin * a.win;
    const float* zero_page = a.zero_page;

    // ---- DMA plan: wave-instruction i = wave + k*WAVES of a chunk moves 64 16-byte pieces.  Which piece a
    // lane moves (channel c, tile row ry, piece q) never changes, so its offset inside a chunk and its
    // in-image test are computed once; per chunk only the chunk base is added.
    int poff[C::NI];                                        // float offset inside the chunk, or -1: zero page
#pragma unroll
    for (int k = 0; k < C::NI; ++k) {
        const int i = wave + k * WAVES;
        const int p = i * 64 + lane;
        const int c = p / (C::IN_ROWS * C::PPR);
        const int rem = p - c * (C::IN_ROWS * C::PPR);
        const int ry = rem / C::PPR;
        const int q = rem - ry * C::PPR;
        const int gy = gy0 + ry, gx = gxa + q * 4;
        const bool ok = (i < C::NB) && (p < C::B_PIECES) && gy >= 0 && gy < a.hin && gx >= 0 && gx < a.win;
        poff[k] = ok ? (c * a.hin + gy) * a.win + gx : -1;
    }
    const size_t chunk_stride = (size_t)CK * a.hin * a.win;
    auto issue = [&](int ch, int stage) {
        float* sbase = lds + stage * C::STAGE;
        const float* cbase = src + ch * chunk_stride;
        const float* wbase = a.wpk2 + (size_t)ch * C::NA * 256 + lane * 4;
#pragma unroll
        for (int k = 0; k < C::NI; ++k) {
            const int i = wave + k * WAVES;                 // wave-uniform instruction index
            const float* g;
            if (i < C::NB) g = poff[k] >= 0 ? cbase + poff[k] : zero_page;
            else if (i < C::NB + C::NA) g = wbase + (i - C::NB) * 256;
            else g = zero_page;                             // padding instruction keeps vmcnt uniform
            __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(sbase + i * 256), 16, 0, 0);
        }
    };

    // ---- per-lane constants
    const int j = lane & (C::NPIX - 1);
    const int g = lane / C::NPIX;
    const int wm = wave % WM, wp = wave / WM;

    acc_t acc[C::UPW][C::MTW];
#pragma unroll
    for (int u = 0; u < C::UPW; ++u)
#pragma unroll
        for (int m = 0; m < C::MTW; ++m)
#pragma unroll
            for (int r = 0; r < C::ACC; ++r) {
                const int mt = wm * C::MTW + m;
                const int co = C::M16 ? (g * 4 + r) : (mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g);
                acc[u][m][r] = a.bias[co];
            }

    int ubase[C::UPW];
#pragma unroll
    for (int u = 0; u < C::UPW; ++u) {
        const int unit = wp * C::UPW + u;
        const int row = unit / TWT, ct = unit % TWT;
        ubase[u] = g * C::PLANE + row * STRIDE * C::ROWP + (ct * C::NPIX + j) * STRIDE + 3;
    }

    // Chunk ch+1 is requested early in the compute phase of chunk ch (its stage was last read in chunk
    // ch-1, and every wave has passed this iteration's barrier since): the DMA address arithmetic hides
    // between MFMAs, and one barrier per chunk both publishes chunk ch and retires chunk ch-1's stage.
    constexpr int ISSUE_AT = 1;                              // s4 group after which the next chunk is requested
    issue(0, 0);
#pragma unroll 1
    for (int ch = 0; ch < C::NCHUNK; ++ch) {
        const int cur = (C::NSTAGE == 2) ? (ch & 1) : 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        const float* tb = lds + cur * C::STAGE;
        const f32x4* ta = reinterpret_cast<const f32x4*>(tb + C::NB * 256) + lane;
#pragma unroll
        for (int s4 = 0; s4 < C::STEPS / 4; ++s4) {
            if (C::NSTAGE == 2 && s4 == ISSUE_AT && ch + 1 < C::NCHUNK) issue(ch + 1, cur ^ 1);
            f32x4 av[C::MTW];
#pragma unroll
            for (int m = 0; m < C::MTW; ++m) av[m] = ta[((wm * C::MTW + m) * (C::STEPS / 4) + s4) * 64];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = s4 * 4 + q;
                const int t = s / C::CG, cg = s % C::CG;
                const int off = cg * C::KPS * C::PLANE + (t / 3) * C::ROWP + (t % 3);
#pragma unroll
                for (int u = 0; u < C::UPW; ++u) {
                    const float b = tb[ubase[u] + off];
#pragma unroll
                    for (int m = 0; m < C::MTW; ++m) {
                        if constexpr (C::M16)
                            acc[u][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][q], b, acc[u][m], 0, 0, 0);
                        else
                            acc[u][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][q], b, acc[u][m], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue: LeakyReLU, [pooling], NCHW store; with POOLK also the stage pooling (EEMFlow.py:144-154) as
    // per-block partial sums: lanes reduce their pixels inside a window by shuffles, rows / tiles of the
    // block are summed through LDS in a fixed order (bitwise repeatable, no atomics), and one value per
    // (channel, window) and block row-group goes to a small buffer that pool_finalize sums.
    constexpr int SW = (POOLK > 0 && POOLK < C::NPIX) ? POOLK : C::NPIX;   // pixels per reduction slot
    constexpr int SLOTS = C::TW / SW;                                       // slots per block row
    float* red = lds;                                                       // [TH][COUT][SLOTS]
#pragma unroll
    for (int u = 0; u < C::UPW; ++u)
#pragma unroll
        for (int m = 0; m < C::MTW; ++m)
#pragma unroll
            for (int r = 0; r < C::ACC; ++r) {
                const float v = acc[u][m][r];
                if (a.act) acc[u][m][r] = v > 0.f ? v : 0.1f * v;
            }
    if constexpr (POOLK > 0) {
        // pooling goes first so that its barriers never wait for the feature-map stores below
        constexpr int NWX = C::TW / POOLK, SPW = POOLK / SW;
        static_assert(C::TW % POOLK == 0 && POOLK % TH == 0 && POOLK % SW == 0, "pool windows must tile the block");
        static_assert(TH * COUT * SLOTS <= C::NSTAGE * C::STAGE, "reduction scratch fits the staging LDS");
        __builtin_amdgcn_s_barrier();                                       // staged operands are dead now
#pragma unroll
        for (int u = 0; u < C::UPW; ++u) {
            const int unit = wp * C::UPW + u;
            const int row = unit / TWT, ct = unit % TWT;
#pragma unroll
            for (int m = 0; m < C::MTW; ++m)
#pragma unroll
                for (int r = 0; r < C::ACC; ++r) {
                    const int mt = wm * C::MTW + m;
                    const int co = C::M16 ? (g * 4 + r) : (mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g);
                    // windows that leave the image are never read back
                    const float sred = lane_group_sum<SW>(acc[u][m][r]);
                    if ((j & (SW - 1)) == 0) red[(row * COUT + co) * SLOTS + ct * (C::NPIX / SW) + j / SW] = sred;
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int idx = tid; idx < COUT * NWX; idx += WAVES * 64) {
            const int co = idx / NWX, wx = idx - co * NWX;
            float s = 0.f;
#pragma unroll
            for (int row = 0; row < TH; ++row)
#pragma unroll
                for (int q = 0; q < SPW; ++q) s += red[(row * COUT + co) * SLOTS + wx * SPW + q];
            a.pool_partial[(((size_t)n * COUT + co) * a.tiles_y + by) * (a.tiles_x * NWX) + bx * NWX + wx] = s;
        }
    }
    float* dst = a.out + (size_t)n * COUT * a.hout * a.wout;
#pragma unroll
    for (int u = 0; u < C::UPW; ++u) {
        const int unit = wp * C::UPW + u;
        const int row = unit / TWT, ct = unit % TWT;
        const int oy = oy0 + row;
        const int ox = ox0 + ct * C::NPIX + j;
        if (oy < a.hout && ox < a.wout) {
#pragma unroll
            for (int m = 0; m < C::MTW; ++m)
#pragma unroll
                for (int r = 0; r < C::ACC; ++r) {
                    const int mt = wm * C::MTW + m;
                    const int co = C::M16 ? (g * 4 + r) : (mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g);
                    const size_t o = ((size_t)co * a.hout + oy) * a.wout + ox;
                    float v = acc[u][m][r];
                    if (a.gate) v *= a.gate[(size_t)n * COUT * a.hout * a.wout + o] > 0.f ? 1.f : 0.1f;
                    dst[o] = v;
                }
        }
    }
}

// ------------------------------------------------------------------------------- persistent variant
// Same math and operand path as enc_conv2_kernel, but a block walks a list of tiles and treats
// (tile, chunk) pairs as one stream of work items over an NST-deep LDS ring: the DMA of item i+NST-1 is
// in flight while item i feeds the MFMAs, across tile boundaries too, so neither a tile's first chunk nor
// its epilogue stalls the matrix pipe.  With NST = 3 there is one barrier per item (it both publishes
// item i and proves everybody left item i-1's stage, which item i+2 then overwrites).
// vmcnt bookkeeping: loads retire in order; the feature-map stores of an epilogue are younger than the
// loads they follow, so the counted wait simply leaves them (NS per wave) in flight as well.
template <int CIN, int COUT, int STRIDE, int TH, int TWT, int WAVES, int WM, int CK, int POOLK, int NST>
__global__ __launch_bounds__(WAVES * 64) void enc_conv3_kernel(EncConvArgs a) {
    using C = Cfg2<CIN, COUT, STRIDE, TH, TWT, WAVES, WM, CK>;
    using acc_t = typename AccT2<C::M16>::type;
    constexpr int SW = (POOLK > 0 && POOLK < C::NPIX) ? POOLK : C::NPIX;
    constexpr int SLOTS = C::TW / SW;
    constexpr int RED = POOLK > 0 ? TH * COUT * SLOTS : 0;              // pooling scratch (floats)
    constexpr int NS = C::UPW * C::MTW * C::ACC;                         // feature-map stores per wave per tile
    static_assert(NST == 2 || NST == 3, "ring depth");
    static_assert((NST * C::STAGE + RED) * 4 <= 160 * 1024, "LDS budget");
    static_assert((NST - 1) * C::NI + NS <= 63, "vmcnt immediate");
    __shared__ __attribute__((aligned(16))) float lds[NST * C::STAGE + RED];
    float* red = lds + NST * C::STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // ---- this block's tiles: XCD x owns the contiguous logical range [x*cpx, (x+1)*cpx); its resident
    // blocks sweep it together, so neighbouring tiles are in flight on the same L2 at the same time
    const int T = a.tiles_x * a.tiles_y * a.nimg;
    const int cpx = (T + 7) >> 3;
    const int xcd = blockIdx.x & 7, kb = blockIdx.x >> 3, gb = gridDim.x >> 3;
    const int r0 = xcd * cpx, r1 = min(r0 + cpx, T);
    const int ntile = (r0 + kb < r1) ? (r1 - r0 - kb + gb - 1) / gb : 0;
    const int nitem = ntile * C::NCHUNK;
    if (nitem == 0) return;
    const float* zero_page = a.zero_page;

    auto issue = [&](int item) {
        const int ti = item / C::NCHUNK, ch = item - ti * C::NCHUNK;
        const int lt = r0 + kb + ti * gb;
        const int bx = lt % a.tiles_x, by = (lt / a.tiles_x) % a.tiles_y, n = lt / (a.tiles_x * a.tiles_y);
        const int gy0 = by * TH * STRIDE - 1, gxa = bx * C::TW * STRIDE - 4;
        const float* src = a.in0 + ((size_t)n * CIN + ch * CK) * a.hin * a.win;
        float* sbase = lds + (item % NST) * C::STAGE;
#pragma unroll
        for (int k = 0; k < C::NI; ++k) {
            const int i = wave + k * WAVES;
            const float* g;
            if (i < C::NB) {
                const int p = i * 64 + lane;
                const int c = p / (C::IN_ROWS * C::PPR);
                const int rem = p - c * (C::IN_ROWS * C::PPR);
                const int ry = rem / C::PPR;
                const int q = rem - ry * C::PPR;
                const int gy = gy0 + ry, gx = gxa + q * 4;
                const bool ok = (p < C::B_PIECES) && gy >= 0 && gy < a.hin && gx >= 0 && gx < a.win;
                g = ok ? src + ((size_t)c * a.hin + gy) * a.win + gx : zero_page;
            } else if (i < C::NB + C::NA) {
                g = a.wpk2 + ((size_t)ch * C::NA + (i - C::NB)) * 256 + lane * 4;
            } else {
                g = zero_page;
            }
            __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(sbase + i * 256), 16, 0, 0);
        }
    };

    const int j = lane & (C::NPIX - 1);
    const int g = lane / C::NPIX;
    const int wm = wave % WM, wp = wave / WM;
    int ubase[C::UPW];
#pragma unroll
    for (int u = 0; u < C::UPW; ++u) {
        const int unit = wp * C::UPW + u;
        const int row = unit / TWT, ct = unit % TWT;
        ubase[u] = g * C::PLANE + row * STRIDE * C::ROWP + (ct * C::NPIX + j) * STRIDE + 3;
    }
    float biasv[C::MTW][C::ACC];
#pragma unroll
    for (int m = 0; m < C::MTW; ++m)
#pragma unroll
        for (int r = 0; r < C::ACC; ++r) {
            const int mt = wm * C::MTW + m;
            biasv[m][r] = a.bias[C::M16 ? (g * 4 + r) : (mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g)];
        }
    acc_t acc[C::UPW][C::MTW];

#pragma unroll
    for (int pre = 0; pre < NST - 1; ++pre)
        if (pre < nitem) issue(pre);

    bool stores_pending = false;                 // did the previous item end with an epilogue?
#pragma unroll 1
    for (int item = 0; item < nitem; ++item) {
        const int ti = item / C::NCHUNK, ch = item - ti * C::NCHUNK;
        // ---- wait for item's DMA; younger ops that may stay in flight: the (NST-2) or fewer items issued
        // after it plus the previous epilogue's stores
        const int ahead = min(NST - 2, nitem - 1 - item);     // items already issued beyond `item` (0 or 1)
        if (ahead == 1) {
            if (stores_pending) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NI + NS) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NI) : "memory");
        } else {
            if (stores_pending) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (item + NST - 1 < nitem) issue(item + NST - 1);    // its stage was last read by item-1: free now
        stores_pending = false;

        if (ch == 0) {
#pragma unroll
            for (int u = 0; u < C::UPW; ++u)
#pragma unroll
                for (int m = 0; m < C::MTW; ++m)
#pragma unroll
                    for (int r = 0; r < C::ACC; ++r) acc[u][m][r] = biasv[m][r];
        }
        const float* tb = lds + (item % NST) * C::STAGE;
        const f32x4* ta = reinterpret_cast<const f32x4*>(tb + C::NB * 256) + lane;
#pragma unroll
        for (int s4 = 0; s4 < C::STEPS / 4; ++s4) {
            f32x4 av[C::MTW];
#pragma unroll
            for (int m = 0; m < C::MTW; ++m) av[m] = ta[((wm * C::MTW + m) * (C::STEPS / 4) + s4) * 64];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = s4 * 4 + q;
                const int t = s / C::CG, cg = s % C::CG;
                const int off = cg * C::KPS * C::PLANE + (t / 3) * C::ROWP + (t % 3);
#pragma unroll
                for (int u = 0; u < C::UPW; ++u) {
                    const float b = tb[ubase[u] + off];
#pragma unroll
                    for (int m = 0; m < C::MTW; ++m) {
                        if constexpr (C::M16)
                            acc[u][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][q], b, acc[u][m], 0, 0, 0);
                        else
                            acc[u][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][q], b, acc[u][m], 0, 0, 0);
                    }
                }
            }
        }

        if (ch == C::NCHUNK - 1) {
            // ---- epilogue of this tile: LeakyReLU, [pooling partial sums], NCHW stores
            const int lt = r0 + kb + ti * gb;
            const int bx = lt % a.tiles_x, by = (lt / a.tiles_x) % a.tiles_y, n = lt / (a.tiles_x * a.tiles_y);
            const int oy0 = by * TH, ox0 = bx * C::TW;
#pragma unroll
            for (int u = 0; u < C::UPW; ++u)
#pragma unroll
                for (int m = 0; m < C::MTW; ++m)
#pragma unroll
                    for (int r = 0; r < C::ACC; ++r) {
                        const float v = acc[u][m][r];
                        if (a.act) acc[u][m][r] = v > 0.f ? v : 0.1f * v;
                    }
            if constexpr (POOLK > 0) {
                constexpr int NWX = C::TW / POOLK, SPW = POOLK / SW;
                static_assert(C::TW % POOLK == 0 && POOLK % TH == 0 && POOLK % SW == 0, "pool windows must tile the block");
#pragma unroll
                for (int u = 0; u < C::UPW; ++u) {
                    const int unit = wp * C::UPW + u;
                    const int row = unit / TWT, ct = unit % TWT;
#pragma unroll
                    for (int m = 0; m < C::MTW; ++m)
#pragma unroll
                        for (int r = 0; r < C::ACC; ++r) {
                            const int mt = wm * C::MTW + m;
                            const int co = C::M16 ? (g * 4 + r) : (mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g);
                            const float sred = lane_group_sum<SW>(acc[u][m][r]);
                            if ((j & (SW - 1)) == 0) red[(row * COUT + co) * SLOTS + ct * (C::NPIX / SW) + j / SW] = sred;
                        }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                for (int idx = tid; idx < COUT * NWX; idx += WAVES * 64) {
                    const int co = idx / NWX, wx = idx - co * NWX;
                    float s = 0.f;
#pragma unroll
                    for (int row = 0; row < TH; ++row)
#pragma unroll
                        for (int q = 0; q < SPW; ++q) s += red[(row * COUT + co) * SLOTS + wx * SPW + q];
                    a.pool_partial[(((size_t)n * COUT + co) * a.tiles_y + by) * (a.tiles_x * NWX) + bx * NWX + wx] = s;
                }
                // `red` is rewritten one tile later at the earliest: every wave passes a ring barrier before that
            }
            float* dst = a.out + (size_t)n * COUT * a.hout * a.wout;
#pragma unroll
            for (int u = 0; u < C::UPW; ++u) {
                const int unit = wp * C::UPW + u;
                const int row = unit / TWT, ct = unit % TWT;
                const int oy = oy0 + row;
                const int ox = ox0 + ct * C::NPIX + j;
                const bool inside = oy < a.hout && ox < a.wout;
#pragma unroll
                for (int m = 0; m < C::MTW; ++m)
#pragma unroll
                    for (int r = 0; r < C::ACC; ++r) {
                        const int mt = wm * C::MTW + m;
                        const int co = C::M16 ? (g * 4 + r) : (mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g);
                        // every lane stores (outside lanes into a scratch page): each wave issues exactly NS
                        // store instructions, which the counted vmcnt waits above rely on
                        float* p = inside ? dst + ((size_t)co * a.hout + oy) * a.wout + ox : a.trash + lane;
                        *p = acc[u][m][r];
                    }
            }
            stores_pending = true;
        }
    }
}

template <int CIN, int COUT, int STRIDE, int TH, int TWT, int WAVES, int WM, int CK, int POOLK>
int launch2(const EncConvArgs& a0, hipStream_t stream) {
    using C = Cfg2<CIN, COUT, STRIDE, TH, TWT, WAVES, WM, CK>;
    EncConvArgs a = a0;
    a.tiles_x = ceil_div(a.wout, C::TW);
    a.tiles_y = ceil_div(a.hout, TH);
    dim3 grid((unsigned)ceil_div(a.tiles_x * a.tiles_y * a.nimg, 8) * 8);
    EEM_NOTE_GRID(grid.x, WAVES * 64);
    if (POOLK > 0 && a.pool_partial != nullptr && a.pool_k == POOLK)
        hipLaunchKernelGGL((enc_conv2_kernel<CIN, COUT, STRIDE, TH, TWT, WAVES, WM, CK, POOLK>), grid, dim3(WAVES * 64),
                           0, stream, a);
    else if (a.pool_partial == nullptr)
        hipLaunchKernelGGL((enc_conv2_kernel<CIN, COUT, STRIDE, TH, TWT, WAVES, WM, CK, 0>), grid, dim3(WAVES * 64), 0,
                           stream, a);
    else {
        eem_set_error("enc_conv2: fused pooling with k=%d is not built for this layer", a.pool_k);
        return EEM_ERR_ARG;
    }
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

}  // namespace

// ------------------------------------------------------------------------------- host side
// Weight layout of the fast path: [chunk][cout tile][k-step/4][lane][4], k-step s of a chunk = tap-major
// (t = s / CG), then channel group (cg = s % CG); lane supplies channel cg*KPS + lane/NPIX.
static int enc2_ck(int cin, int cout);

bool enc2_supported(int cin, int cout, int stride, int win) {
    if ((win & 3) != 0 || cin < 16) return false;
    return (cin == 16 && cout == 16 && stride == 1) || (cin == 16 && cout == 32 && stride == 2) ||
           (cin == 32 && cout == 32 && stride == 1) || (cin == 32 && cout == 64 && stride == 2) ||
           (cin == 64 && cout == 64 && stride == 1);
}

size_t enc2_packed_floats(int cin, int cout) { return (size_t)cout * cin * 9; }

void enc2_pack_weights(const float* w, int cin, int cout, float* packed) {
    const bool m16 = cout == 16;
    const int npix = m16 ? 16 : 32, kps = m16 ? 4 : 2, mt = m16 ? 1 : cout / 32;
    const int ck = enc2_ck(cin, cout), cg = ck / kps, steps = 9 * cg, nchunk = cin / ck;
    for (int ch = 0; ch < nchunk; ++ch)
        for (int m = 0; m < mt; ++m)
            for (int s = 0; s < steps; ++s)
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = m * 32 + (lane & (npix - 1));
                    const int c = ch * ck + (s % cg) * kps + lane / npix;
                    const int t = s / cg;
                    const size_t idx = ((((size_t)ch * mt + m) * (steps / 4) + s / 4) * 64 + lane) * 4 + (s & 3);
                    packed[idx] = w[((size_t)co * cin + c) * 9 + t];
                }
}

template <int CIN, int COUT, int STRIDE, int TH, int TWT, int WAVES, int WM, int CK, int POOLK, int NST>
int launch3(const EncConvArgs& a0, hipStream_t stream) {
    using C = Cfg2<CIN, COUT, STRIDE, TH, TWT, WAVES, WM, CK>;
    EncConvArgs a = a0;
    a.tiles_x = ceil_div(a.wout, C::TW);
    a.tiles_y = ceil_div(a.hout, TH);
    const int T = a.tiles_x * a.tiles_y * a.nimg;
    constexpr int SW = (POOLK > 0 && POOLK < C::NPIX) ? POOLK : C::NPIX;
    constexpr int LDS_BYTES = (NST * C::STAGE + (POOLK > 0 ? TH * COUT * (C::TW / SW) : 0)) * 4;
    constexpr int BY_LDS = (160 * 1024) / LDS_BYTES, BY_WAVES = 32 / WAVES;
    constexpr int RES = BY_LDS < BY_WAVES ? (BY_LDS < 1 ? 1 : BY_LDS) : BY_WAVES;   // resident blocks per CU
    int per_xcd = ceil_div(T, 8);
    static const int cap = enc_blocks_per_xcd(COUT == 32 ? "P32" : (COUT == 64 ? "P64" : "P"), 32 * RES);
    if (per_xcd > cap) per_xcd = cap;
    if (a.pool_partial != nullptr && !(POOLK > 0 && a.pool_k == POOLK)) {
        eem_set_error("enc_conv3: fused pooling with k=%d is not built for this layer", a.pool_k);
        return EEM_ERR_ARG;
    }
    if (POOLK > 0 && a.pool_partial != nullptr)
        hipLaunchKernelGGL((enc_conv3_kernel<CIN, COUT, STRIDE, TH, TWT, WAVES, WM, CK, POOLK, NST>), dim3(per_xcd * 8),
                           dim3(WAVES * 64), 0, stream, a);
    else
        hipLaunchKernelGGL((enc_conv3_kernel<CIN, COUT, STRIDE, TH, TWT, WAVES, WM, CK, 0, NST>), dim3(per_xcd * 8),
                           dim3(WAVES * 64), 0, stream, a);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

// ---- variant table.  Each layer shape has a few tilings; index 0 is the production choice, the others are
// kept for tuning on hardware (EEM_V<cin>_<cout>=<index> in the environment selects one - a tuning knob,
// read once).
struct Variant {
    int th, tw, poolk, ck;
    int (*launch)(const EncConvArgs&, hipStream_t);
};

//                      CIN COUT S TH TWT WAVES WM CK POOLK
#define V(CIN, COUT, S, TH, TWT, WAVES, WM, CK, POOLK) \
    Variant{TH, TWT * (COUT == 16 ? 16 : 32), POOLK, CK, &launch2<CIN, COUT, S, TH, TWT, WAVES, WM, CK, POOLK>}

// (measured at 1280x720 b1, profiles/r01_variant_sweep.txt: index 0 is the fastest of each family)
static const Variant kV16_16[] = {V(16, 16, 1, 4, 4, 8, 1, 16, 32), V(16, 16, 1, 4, 4, 4, 1, 16, 32),
                                  V(16, 16, 1, 8, 4, 4, 1, 16, 32), V(16, 16, 1, 8, 4, 8, 1, 16, 32)};
static const Variant kV16_32[] = {V(16, 32, 2, 4, 1, 4, 1, 8, 0), V(16, 32, 2, 2, 1, 2, 1, 16, 0),
                                  V(16, 32, 2, 4, 1, 4, 1, 16, 0), V(16, 32, 2, 8, 1, 8, 1, 8, 0)};
static const Variant kV32_32[] = {V(32, 32, 1, 4, 2, 8, 1, 8, 16), V(32, 32, 1, 4, 2, 8, 1, 16, 16),
                                  V(32, 32, 1, 4, 1, 4, 1, 8, 16), V(32, 32, 1, 2, 2, 4, 1, 8, 16)};
static const Variant kV32_64[] = {V(32, 64, 2, 4, 1, 8, 2, 8, 0), V(32, 64, 2, 2, 1, 4, 2, 8, 0),
                                  V(32, 64, 2, 4, 1, 4, 1, 8, 0)};
static const Variant kV64_64[] = {V(64, 64, 1, 4, 1, 8, 2, 8, 8), V(64, 64, 1, 2, 1, 4, 2, 8, 8),
                                  V(64, 64, 1, 4, 1, 8, 2, 16, 8), V(64, 64, 1, 4, 1, 4, 1, 8, 8)};
#define P(CIN, COUT, S, TH, TWT, WAVES, WM, CK, POOLK, NST) \
    Variant{TH, TWT * (COUT == 16 ? 16 : 32), POOLK, CK, &launch3<CIN, COUT, S, TH, TWT, WAVES, WM, CK, POOLK, NST>}
static const Variant kP16_16[] = {P(16, 16, 1, 4, 4, 8, 1, 16, 32, 2), P(16, 16, 1, 4, 4, 8, 1, 16, 32, 3),
                                  P(16, 16, 1, 8, 4, 8, 1, 16, 32, 2), P(16, 16, 1, 4, 4, 4, 1, 16, 32, 3)};
static const Variant kP16_32[] = {P(16, 32, 2, 4, 1, 4, 1, 8, 0, 3), P(16, 32, 2, 4, 1, 4, 1, 16, 0, 2),
                                  P(16, 32, 2, 8, 1, 8, 1, 8, 0, 3)};
static const Variant kP32_32[] = {P(32, 32, 1, 4, 2, 8, 1, 8, 16, 3), P(32, 32, 1, 4, 2, 8, 1, 16, 16, 2),
                                  P(32, 32, 1, 4, 2, 8, 1, 8, 16, 2)};
static const Variant kP32_64[] = {P(32, 64, 2, 4, 1, 8, 2, 8, 0, 3), P(32, 64, 2, 4, 1, 8, 2, 8, 0, 2)};
static const Variant kP64_64[] = {P(64, 64, 1, 4, 1, 8, 2, 8, 8, 3), P(64, 64, 1, 4, 1, 8, 2, 8, 8, 2),
                                  P(64, 64, 1, 4, 1, 8, 2, 16, 8, 2)};
#undef P
#undef V

static const Variant* pick_variant(int cin, int cout) {
    const Variant* tab = nullptr;
    int n = 0;
    if (cin == 16 && cout == 16) { tab = kV16_16; n = sizeof(kV16_16) / sizeof(Variant); }
    else if (cin == 16 && cout == 32) { tab = kV16_32; n = sizeof(kV16_32) / sizeof(Variant); }
    else if (cin == 32 && cout == 32) { tab = kV32_32; n = sizeof(kV32_32) / sizeof(Variant); }
    else if (cin == 32 && cout == 64) { tab = kV32_64; n = sizeof(kV32_64) / sizeof(Variant); }
    else if (cin == 64 && cout == 64) { tab = kV64_64; n = sizeof(kV64_64) / sizeof(Variant); }
    if (!tab) return nullptr;
    char name[32];
    snprintf(name, sizeof(name), "EEM_V%d_%d", cin, cout);
    const char* e = getenv(name);
    int idx = e ? atoi(e) : 0;
    if (idx >= 100) {                            // 100 + i selects persistent variant i
        const Variant* pt = nullptr;
        int pn = 0;
        if (cin == 16 && cout == 16) { pt = kP16_16; pn = sizeof(kP16_16) / sizeof(Variant); }
        else if (cin == 16 && cout == 32) { pt = kP16_32; pn = sizeof(kP16_32) / sizeof(Variant); }
        else if (cin == 32 && cout == 32) { pt = kP32_32; pn = sizeof(kP32_32) / sizeof(Variant); }
        else if (cin == 32 && cout == 64) { pt = kP32_64; pn = sizeof(kP32_64) / sizeof(Variant); }
        else { pt = kP64_64; pn = sizeof(kP64_64) / sizeof(Variant); }
        if (idx - 100 < pn) return pt + (idx - 100);
        idx = 0;
    }
    if (idx < 0 || idx >= n) idx = 0;
    return tab + idx;
}

static int enc2_ck(int cin, int cout) {
    const Variant* v = pick_variant(cin, cout);
    return v ? v->ck : 16;
}

// Block tile (rows, cols) of the fast path and the pooling window it can fuse (0 = none), per layer.
void enc2_tile(int cin, int cout, int* th, int* tw, int* poolk) {
    const Variant* v = pick_variant(cin, cout);
    *th = v ? v->th : 4; *tw = v ? v->tw : 32; *poolk = v ? v->poolk : 0;
}

int enc_conv2_launch(int cin, int cout, int stride, const EncConvArgs& a, hipStream_t stream) {
    const Variant* v = pick_variant(cin, cout);
    if (!v) {
        eem_set_error("enc_conv2_launch: unsupported layer cin=%d cout=%d stride=%d", cin, cout, stride);
        return EEM_ERR_ARG;
    }
    if (v->ck == 8 && s2_supported(cin, cout, stride, a)) return s2_launch(cin, a, stream);
    return v->launch(a, stream);
}
