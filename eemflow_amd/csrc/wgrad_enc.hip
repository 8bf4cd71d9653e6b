// Weight gradients of the encoder's 3x3 convs (the transposes of EEMFlow.py:26-30,75-82 under autograd):
//     dW[co][ci][ky][kx] = sum_{n, oy, ox} G[n][co][oy][ox] * X[n][ci][oy*S + ky - 1][ox*S + kx - 1]
// with G already the gradient w.r.t. the conv's pre-activation (train_api.hip stores the encoder gradients
// pre-gated), so this is a plain GEMM  dW[M = cout][N = (ci, tap)] = G[M][K = pixels] * X[K][N]  whose K runs over
// every output pixel of the batch.
//
// Layout of the work:
//   * a block owns 16 input channels (blockIdx.y) x all couts and a contiguous range of 128-pixel tiles (TH x TW,
//     4x32 or 8x16); N = 16 ci x 9 taps = 144 = nine 16-wide MFMA tiles exactly, M = cout / 16 tiles: no padding
//     anywhere, v_mfma_f32_16x16x4_f32 throughout (the 32x32x2 shape wastes half of a 16-cout layer);
//   * K is split over the four waves: wave w takes pixels [32w, 32w+32) of the tile (8 k-steps of 4 pixels) against
//     ALL (M, N) tiles - 36 / 72 / 144 accumulator registers for 16 / 32 / 64 couts - so the waves share no
//     operand, read each LDS word once, and are perfectly balanced; the four partial dW meet in LDS at the very end
//     (plain read-modify-writes, wave after wave) and leave with one fp32 atomic per weight and block; the bias
//     gradient (row sums of G) rides along in the first channel chunk;
//   * G and the haloed X tile stream HBM/L2 -> LDS by 16-byte LDS-DMA, double-buffered: tile i+1 is requested
//     right after the barrier that hands tile i to the MFMAs (one barrier per tile, no VALU staging).  G rows sit at
//     a pitch of 132 floats (33 DMA pieces, the last one a dummy) so that the A-operand column reads hit every
//     bank exactly twice; X rows start 4 columns left of the tile so every piece is 16-byte aligned, and pieces
//     outside the image (the conv's zero padding, ragged tiles) come from a zero page.
// Per k-step a wave issues MT + 9 ds_read_b32 for 9*MT MFMAs.
#include "common.h"
#include "train.h"

namespace {

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int MT, int S, int TW, int CP, int KH = 3, int KW = 3>
struct WgCfg {
    static constexpr int TAPS = KH * KW, PH = KH / 2, PW = KW / 2;
    static constexpr int NT = (CP * TAPS + 15) / 16;             // 16-wide N tiles: 9 for 16 channels x 9 taps, 3 for the 5 of pconv1_1
    static constexpr int TH = 128 / TW;
    static constexpr int GP = 132;                               // G row pitch (floats): 33 pieces
    static constexpr int GSLOTS = MT * 16 * 33;
    static constexpr int XR = (TH - 1) * S + KH;
    static constexpr int XC = ((TW - 1) * S + KW - PW + 4 + 3) & ~3;   // staged columns, from ox0*S - 4 (3x3: 40 / 72 at stride 1 / 2)
    static constexpr int XQ = XC / 4;
    static constexpr int PC = XR * XC;                           // channel plane
    static constexpr int XSLOTS = CP * XR * XQ;
    static constexpr int NGI = (GSLOTS + 255) / 256;             // DMA instructions per wave and tile
    static constexpr int NXI = (XSLOTS + 255) / 256;
    static constexpr int GFL = NGI * 256 * 4;                    // floats per stage (whole instructions)
    static constexpr int XFL = NXI * 256 * 4;
    static constexpr int STAGE = GFL + XFL;
    static constexpr int RED = MT * NT * 256;                    // epilogue buffer (floats), reuses the stages
    static_assert(2 * STAGE >= RED, "epilogue buffer fits the stages");
    static_assert(2 * STAGE * 4 <= 160 * 1024, "LDS budget");
};

// ---- fp32 products as bf16 pieces (round 6; the arithmetic of conv_bx3.hip / gconvb.hip): every operand = three bf16 pieces that sum to it
// exactly (8 significand bits each, by truncation), a product = six piece products accumulated in fp32 by v_mfma_f32_16x16x32_bf16,
// small terms first.  K = 32 PIXELS per MFMA: a lane holds 8 consecutive pixels of its cout row (A) / of its (ci, tap) column (B).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 wg_bf(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ void wg_split8(const float (&x)[8], u32x4& p0, u32x4& p1, u32x4& p2) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const float xa = x[2 * d], xb = x[2 * d + 1];
        const float ra = xa - __uint_as_float(__float_as_uint(xa) & 0xffff0000u), rb = xb - __uint_as_float(__float_as_uint(xb) & 0xffff0000u);
        const float sa = ra - __uint_as_float(__float_as_uint(ra) & 0xffff0000u), sb = rb - __uint_as_float(__float_as_uint(rb) & 0xffff0000u);
        p0[d] = __builtin_amdgcn_perm(__float_as_uint(xb), __float_as_uint(xa), 0x07060302u);      // (hi16(xb) << 16) | hi16(xa)
        p1[d] = __builtin_amdgcn_perm(__float_as_uint(rb), __float_as_uint(ra), 0x07060302u);
        p2[d] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
    }
}

#ifdef EEM_DIAG
__device__ int g_wg_dbg;                                 // diagnostic builds: EEM_WG_DBG=1 leaves the MFMAs out (what is left is the operands' way in)
#endif

// BX: the tile's products on the bf16 matrix pipe (exact three-piece operands, above) - wave w's 32 pixels of the tile are ONE K = 32 step:
// (MT + 9) operand splits and 6 x 9 MT MFMAs of 16 cycles where the fp32 form issues 8 x 9 MT of 32
template <int MT, int S, int TW, int CP, int KH = 3, int KW = 3, bool BX = false>
__global__ __launch_bounds__(256) void wgrad_enc_kernel(WgradArgs a, int tiles_x, int tiles_y, int flags) {
    const int walk = flags & 1;
    const bool burst = (flags & 2) != 0;
    using C = WgCfg<MT, S, TW, CP, KH, KW>;
    constexpr int NT = C::NT, TAPS = C::TAPS;
    // blockIdx.z: chunk of MT * 16 couts (layers wider than 64 couts: E-RAFT's update block and heads)
    a.g_coff += blockIdx.z * MT * 16;
    const int co_base = blockIdx.z * MT * 16;
    const int cout_here = min(MT * 16, a.cout - co_base);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int ci0 = blockIdx.y * CP;
    const int cin_here = min(CP, a.cin - ci0);
    const int nvalid = cin_here * TAPS;

    // tiles: an XCD owns a contiguous range; its resident blocks take every gb-th tile of it (interleaved, round 6: the tiles whose halo
    // rows / columns overlap are then in flight together and meet in the XCD's L2 - this kernel is bound by its operands' way in since its
    // products run as bf16 pieces).  An EXPERIMENT (EEM_WGRAD_WALK=1): measured at no difference; default each block a contiguous sub-range
    TileRange tr_ = block_tile_range(tiles_x * tiles_y * a.n, blockIdx.x, gridDim.x);
    int walk_stride = 1;
    if (walk) {
        const int T = tiles_x * tiles_y * a.n;
        const int cpx = (T + 7) >> 3, xcd = blockIdx.x & 7, kb = blockIdx.x >> 3, gb = gridDim.x >> 3;
        const int r0 = xcd * cpx, r1 = min(r0 + cpx, T);
        const int have = r1 - r0 - kb;
        tr_.first = r0 + kb;
        tr_.count = have > 0 ? (have + gb - 1) / gb : 0;
        walk_stride = gb;
    }
    if (tr_.count == 0) return;
    TileCoord cur = tile_coord(tr_.first, tiles_x, tiles_y);
    auto advance = [&](TileCoord& t) {
        t.bx += walk_stride;
        while (t.bx >= tiles_x) { t.bx -= tiles_x; if (++t.by == tiles_y) { t.by = 0; ++t.n; } }
    };
    const size_t ghw = (size_t)a.hout * a.wout, xhw = (size_t)a.hin * a.win;

    // ---- DMA plan: per instruction and lane a byte offset from the tile's base pointer and the piece's
    // (row, column) inside the tile for the bounds test
    constexpr unsigned RANGE = 0x7fffff00u;                     // bytes a buffer descriptor covers; offsets at or above it read as zero
    unsigned goff[C::NGI], xoff[C::NXI];
    int grc[C::NGI], xrc[C::NXI];
#pragma unroll
    for (int k = 0; k < C::NGI; ++k) {
        const int f = (wave + 4 * k) * 64 + lane;
        const int co = f / 33, q = f - co * 33;
        const int p = 4 * q, py = p / TW, px = p - py * TW;
        const bool ok = f < C::GSLOTS && q < 32 && co < cout_here;
        goff[k] = ok ? (unsigned)(((size_t)co * ghw + (size_t)py * a.wout + px) * 4) : RANGE;
        grc[k] = ok ? (py | (px << 8)) : -1;
    }
#pragma unroll
    for (int k = 0; k < C::NXI; ++k) {
        const int f = (wave + 4 * k) * 64 + lane;
        const int ci = f / (C::XR * C::XQ), rem = f - ci * (C::XR * C::XQ);
        const int ry = rem / C::XQ, qx = rem - ry * C::XQ;
        const bool ok = f < C::XSLOTS && ci < cin_here;
        xoff[k] = ok ? (unsigned)(((size_t)ci * xhw + (size_t)ry * a.win + 4 * qx) * 4) : RANGE;
        xrc[k] = ok ? (ry | ((4 * qx) << 8)) : -1;
    }
    // Both operands travel as buffer loads: the descriptor's base is the tile's corner (a scalar), a lane's offset inside the tile is
    // fixed for the whole kernel, and pieces outside the image (or past the block's channels) carry an offset beyond the descriptor's
    // range - the hardware writes zeros.  Interior tiles need no vector instruction per piece; edge tiles one compare + select.
    // (round 6: a tile's NGI + NXI requests are no longer one block behind the barrier - 13 instructions of 1 KB per wave at the address
    // unit's 16 cycles each, the matrix pipe idle meanwhile, as conv_wnc.hip's stamps showed for the same pattern - but go out one or two
    // per column tile / k-step of the running tile's multiply loop: `prep` builds the two descriptors, `piece` sends request k)
#if __HIP_DEVICE_COMPILE__
    struct Req {
        __amdgpu_buffer_rsrc_t gr, xr;
        float* sg;
        float* sx;
        int oy0, ox0, gy0, gx0;
        bool g_inside, x_inside;
    };
    auto prep = [&](int stage, const TileCoord& tc) {
        Req r;
        r.oy0 = tc.by * C::TH; r.ox0 = tc.bx * TW;
        r.gy0 = r.oy0 * S - C::PH; r.gx0 = r.ox0 * S - 4;
        const char* gb = reinterpret_cast<const char*>(a.g + ((size_t)tc.n * a.g_ctotal + a.g_coff) * ghw) +
                         ((long)r.oy0 * a.wout + r.ox0) * 4;
        const char* xb = reinterpret_cast<const char*>(a.x + ((size_t)tc.n * a.x_ctotal + a.x_coff + ci0) * xhw) +
                         ((long)r.gy0 * a.win + r.gx0) * 4;
        r.sg = lds + stage * C::STAGE;
        r.sx = r.sg + C::GFL;
        r.g_inside = r.oy0 + C::TH <= a.hout && r.ox0 + TW <= a.wout;
        r.x_inside = r.gy0 >= 0 && r.gy0 + C::XR <= a.hin && r.gx0 >= 0 && r.gx0 + 4 * C::XQ <= a.win;
        r.gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(gb), (short)0, (int)RANGE, 0x00020000);
        r.xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xb), (short)0, (int)RANGE, 0x00020000);
        return r;
    };
    // request k of a tile: 0 .. NGI - 1 the G pieces, NGI .. NGI + NXI - 1 the X pieces (k is a compile-time value at every call)
    auto piece = [&](const Req& r, int k) __attribute__((always_inline)) {
        if (k < C::NGI) {
            unsigned vo = goff[k];
            if (!r.g_inside) {
                const int py = grc[k] & 255, px = grc[k] >> 8;
                vo = (grc[k] >= 0 && r.oy0 + py < a.hout && r.ox0 + px < a.wout) ? vo : RANGE;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r.gr, LDS_PTR(r.sg + (wave + 4 * k) * 256), 16, vo, 0, 0, 0);
        } else if (k < C::NGI + C::NXI) {
            const int kx = k - C::NGI;
            unsigned vo = xoff[kx];
            if (!r.x_inside) {
                const int iy = r.gy0 + (xrc[kx] & 255), ix = r.gx0 + (xrc[kx] >> 8);
                vo = (xrc[kx] >= 0 && iy >= 0 && iy < a.hin && ix >= 0 && ix + 4 <= a.win) ? vo : RANGE;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r.xr, LDS_PTR(r.sx + (wave + 4 * kx) * 256), 16, vo, 0, 0, 0);
        }
    };
    auto issue = [&](int stage, const TileCoord& tc) {
        const Req r = prep(stage, tc);
#pragma unroll
        for (int k = 0; k < C::NGI + C::NXI; ++k) piece(r, k);
    };
#else
    struct Req { int unused; };
    auto prep = [&](int, const TileCoord&) { return Req{0}; };
    auto piece = [&](const Req&, int) {};
    auto issue = [&](int, const TileCoord&) {};
#endif

    // ---- operand offsets.  k-step s of this wave covers pixels 32*wave + 4s + g
    int boff[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        int n = nt * 16 + j;
        n = n < nvalid ? n : 0;                                  // columns past cin_here*9 are never written back
        const int ci = n / TAPS, tap = n - ci * TAPS;
        boff[nt] = ci * C::PC + (tap / KW) * C::XC + (tap % KW) + 4 - C::PW;   // the stage starts 4 columns left of the tile
    }
    int aoff[8], xo[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int p = wave * 32 + 4 * s + g;
        const int py = p / TW, px = p - py * TW;
        aoff[s] = j * C::GP + p;
        xo[s] = py * S * C::XC + px * S;
    }

    f32x4 acc[MT][NT];
    float bsum[MT];                                              // bias gradient: row sums of G, lane (j, g) -> cout mt*16 + j
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        bsum[mt] = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    issue(0, cur);
    TileCoord nxt = cur;
#pragma unroll 1
    for (int it = 0; it < tr_.count; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's pieces of tile `it` have landed
        __builtin_amdgcn_s_barrier();                            // everyone's have; everyone is done with tile it-1
        asm volatile("" ::: "memory");
        const bool more = it + 1 < tr_.count;
        constexpr int NREQ = C::NGI + C::NXI;
        if (more) advance(nxt);
        const Req rq = prep((it + 1) & 1, nxt);                  // (of the running tile when there is no next one: never sent)
        if (more && burst) {
#pragma unroll
            for (int k = 0; k < C::NGI + C::NXI; ++k) piece(rq, k);
        }
        const bool spread = more && !burst;
        const float* sg = lds + (it & 1) * C::STAGE;
        const float* sx = sg + C::GFL;
#ifdef EEM_DIAG
        if (g_wg_dbg & 1) {                                      // (no compute phase: the requests it would have carried, at once)
            if (spread) {
#pragma unroll
                for (int k = 0; k < NREQ; ++k) piece(rq, k);
            }
            continue;
        }
#endif
        if constexpr (BX) {
            // this lane's eight pixels: p = 32 wave + 8 g + e (one row of the tile: 8 divides TW)
            const int p8 = wave * 32 + 8 * g;
            const int py8 = p8 / TW, px8 = p8 - py8 * TW;
            u32x4 ap[MT][3];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float x[8];
                const f32x4 lo = *reinterpret_cast<const f32x4*>(sg + (mt * 16 + j) * C::GP + p8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(sg + (mt * 16 + j) * C::GP + p8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { x[e] = lo[e]; x[4 + e] = hi[e]; }
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[mt] += x[e];
                wg_split8(x, ap[mt][0], ap[mt][1], ap[mt][2]);
            }
            const int xo8 = py8 * S * C::XC + px8 * S;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = sx[boff[nt] + xo8 + e * S];
                u32x4 bp[3];
                wg_split8(x, bp[0], bp[1], bp[2]);
                if (spread) {                                                // this column tile's share of the next tile's requests
                    constexpr int PP = (NREQ + NT - 1) / NT;
#pragma unroll
                    for (int q = 0; q < PP; ++q) piece(rq, nt * PP + q);
                }
#pragma unroll
                for (int i = 0; i < 6; ++i) {                            // small terms first
                    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wg_bf(ap[mt][PA[i]]), wg_bf(bp[PB[i]]), acc[mt][nt], 0, 0, 0);
                }
            }
            continue;
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (spread) {
                constexpr int PP = (NREQ + 7) / 8;
#pragma unroll
                for (int q = 0; q < PP; ++q) piece(rq, s * PP + q);
            }
            float av[MT], bv[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt] = sg[aoff[s] + mt * 16 * C::GP];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = sx[boff[nt] + xo[s]];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                bsum[mt] += av[mt];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
            }
        }
    }

    // ---- the four waves' partial sums meet in LDS: wave 0 stores, waves 1..3 add in turn (plain 16-byte
    // read-modify-writes at [tile][lane] - LDS float atomics took 43 us here), then one global atomic per weight
    f32x4* red = reinterpret_cast<f32x4*>(lds);
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    f32x4* cell = red + (mt * NT + nt) * 64 + lane;
                    *cell = w == 0 ? acc[mt][nt] : *cell + acc[mt][nt];
                }
        }
    }
    __syncthreads();
#ifdef EEM_DIAG
    if (g_wg_dbg & 2) return;                            // EEM_WG_DBG=2: no global atomics (what the one-atomic-per-weight-and-block epilogue costs)
#endif
    for (int e = threadIdx.x; e < MT * NT * 64; e += 256) {
        const int tile = e >> 6, l = e & 63;
        const int mt = tile / NT, nt = tile - mt * NT;
        const int n = nt * 16 + (l & 15);
        if (n >= nvalid) continue;
        const f32x4 v = red[e];
#pragma unroll
        for (int r = 0; r < 4; ++r) {                            // D[co = 4 * (l / 16) + r][n = l % 16]
            const int co = mt * 16 + 4 * (l >> 4) + r;
            if (co < cout_here)
                atomicAdd(&a.dw[((size_t)(co_base + co) * (a.dw_cin ? a.dw_cin : a.cin) + a.dw_coff + ci0) * TAPS + n], v[r]);
        }
    }
    // ---- bias gradient (first channel chunk only): sum over the 4 pixel slots of a k-step, the waves, the blocks
    if (a.db && blockIdx.y == 0) {
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float v = bsum[mt];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (g == 0) lds[wave * 64 + mt * 16 + j] = v;
        }
        __syncthreads();
        if ((int)threadIdx.x < cout_here)
            atomicAdd(&a.db[co_base + threadIdx.x], lds[threadIdx.x] + lds[64 + threadIdx.x] + lds[128 + threadIdx.x] + lds[192 + threadIdx.x]);
    }
}

template <int MT, int S, int TW, int CP, int KH = 3, int KW = 3, bool BX = false>
int launch(const WgradArgs& a, hipStream_t st) {
    using C = WgCfg<MT, S, TW, CP, KH, KW>;
    const int tiles_x = ceil_div(a.wout, TW), tiles_y = ceil_div(a.hout, C::TH);
    const int T = tiles_x * tiles_y * a.n;
    const int chunks = ceil_div(a.cin, CP), cochunks = ceil_div(a.cout, MT * 16);
    const int lds_bytes = 2 * C::STAGE * 4;
    int per_cu = (160 * 1024) / lds_bytes;
    per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
    int gx = (256 * per_cu) / (chunks * cochunks);               // ~per_cu resident blocks per CU over all chunks
    gx = (gx + 7) & ~7;
    if (gx < 8) gx = 8;
    const int need = (ceil_div(T, 8)) * 8;
    if (gx > need) gx = need;
#ifdef EEM_DIAG
    { static int once = [] { const char* e = getenv("EEM_WG_DBG"); int v = e ? atoi(e) : 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wg_dbg), &v, sizeof v); return v; }(); (void)once; }
#endif
    static bool raised = false;
    if (!raised) {
        EEM_HIP_CHECK(hipFuncSetAttribute((const void*)wgrad_enc_kernel<MT, S, TW, CP, KH, KW, BX>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          160 * 1024));
        raised = true;
    }
    static const int walk = [] { const char* e = getenv("EEM_WGRAD_WALK"); return e ? atoi(e) : 0; }();     // (measured: no difference, profiles/r06_wgwalk.txt - its operands come out of L2 / MALL either way)
    const char* eb = getenv("EEM_WGRAD_BURST");                 // (=1, read per call: a tile's requests in one block behind the barrier, the form through round 6's first half)
    const int flags = (walk & 1) | ((eb && eb[0] == '1') ? 2 : 0);
    hipLaunchKernelGGL((wgrad_enc_kernel<MT, S, TW, CP, KH, KW, BX>), dim3(gx, chunks, cochunks), dim3(256), lds_bytes, st, a, tiles_x, tiles_y, flags);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

// EEM_NO_WGRAD_BX3=1 (read per call): the fp32 MFMA form of every launch
static bool use_bx3() {
    const char* e = getenv("EEM_NO_WGRAD_BX3");
    return !(e && e[0] == '1');
}

// stride-1 layers of any width with 3x3, 1x5, 5x1 or 1x1 filters (E-RAFT's residual stacks, update block and heads): 64-cout chunks
template <int KH, int KW>
int launch_wide(const WgradArgs& a, hipStream_t st) {
    if (use_bx3()) {
        if (a.wout % 32 == 0 || a.wout >= 256) return launch<4, 1, 32, 16, KH, KW, true>(a, st);
        return launch<4, 1, 16, 16, KH, KW, true>(a, st);
    }
    if (a.wout % 32 == 0 || a.wout >= 256) return launch<4, 1, 32, 16, KH, KW>(a, st);
    return launch<4, 1, 16, 16, KH, KW>(a, st);
}

template <int MT, int S>
int launch_tw(const WgradArgs& a, hipStream_t st) {
    static const bool mt1 = [] { const char* e = getenv("EEM_WGRAD_BX3_MT1"); return e && e[0] == '1'; }();     // (measurement)
    if ((MT >= 2 || mt1) && use_bx3()) {                             // (one 16-cout tile per wave: the splits cost what the multiplies save)
        if (a.wout % 32 == 0 || a.wout >= 256) return launch<MT, S, 32, 16, 3, 3, true>(a, st);
        return launch<MT, S, 16, 16, 3, 3, true>(a, st);
    }
    if (a.wout % 32 == 0 || a.wout >= 256) return launch<MT, S, 32, 16>(a, st);
    return launch<MT, S, 16, 16>(a, st);
}

}  // namespace

bool wgrad_enc_supported(const WgradArgs& a) {
    const char* e = getenv("EEM_NO_WGRAD_ENC");                      // read per call: a test flips it inside one process
    if (e && e[0] == '1') return false;
    return a.zero_page && a.gate == nullptr && a.k == 3 && a.pad == 1 && (a.stride == 1 || a.stride == 2) &&
           (a.cout == 16 || a.cout == 32 || a.cout == 64) && a.g_cmul == 1 && (a.cin <= 16 || a.cin % 16 == 0) &&
           a.wout % 4 == 0 && a.win % 4 == 0 && ((uintptr_t)a.g & 15) == 0 && ((uintptr_t)a.x & 15) == 0 &&
           ((size_t)a.hout * a.wout) % 4 == 0 && ((size_t)a.hin * a.win) % 4 == 0 &&
           (size_t)a.cout * a.hout * a.wout * 4 < (1u << 31) && (size_t)16 * a.hin * a.win * 4 < (1u << 31);
}

// a.kh / a.kw (0: a.k), a.ph / a.pw, a.dw_cin / a.dw_coff as in train.h
bool wgrad_wide_supported(const WgradArgs& a) {
    const char* e = getenv("EEM_NO_WGRAD_WIDE");                     // read per call: a test flips it inside one process
    if (e && e[0] == '1') return false;
    const int kh = a.kh ? a.kh : a.k, kw = a.kh ? a.kw : a.k;
    const int ph = a.kh ? a.ph : a.pad, pw = a.kh ? a.pw : a.pad;
    const bool shape = (kh == 3 && kw == 3) || (kh == 1 && kw == 5) || (kh == 5 && kw == 1) || (kh == 1 && kw == 1);
    return shape && ph == kh / 2 && pw == kw / 2 && a.stride == 1 && a.zero_page && a.gate == nullptr && a.g_cmul == 1 && a.cout >= 16 &&
           a.cin >= 16 && a.wout % 4 == 0 && a.win % 4 == 0 && ((uintptr_t)a.g & 15) == 0 && ((uintptr_t)a.x & 15) == 0 &&
           ((size_t)a.hout * a.wout) % 4 == 0 && ((size_t)a.hin * a.win) % 4 == 0 &&
           (size_t)a.cout * a.hout * a.wout * 4 < (1u << 31) && (size_t)16 * a.hin * a.win * 4 < (1u << 31);
}

int wgrad_wide_launch(const WgradArgs& a, hipStream_t st) {
    const int kh = a.kh ? a.kh : a.k, kw = a.kh ? a.kw : a.k;
    if (kh == 3) return launch_wide<3, 3>(a, st);
    if (kh == 5) return launch_wide<5, 1>(a, st);
    if (kw == 5) return launch_wide<1, 5>(a, st);
    return launch_wide<1, 1>(a, st);
}

int wgrad_enc_launch(const WgradArgs& a, hipStream_t st) {
    const int mt = a.cout / 16;
    if (a.stride == 1) {
        if (mt == 1) return launch_tw<1, 1>(a, st);
        if (mt == 2) return launch_tw<2, 1>(a, st);
        return launch_tw<4, 1>(a, st);
    }
    if (mt == 1 && a.cin <= 5 && (a.wout % 32 == 0 || a.wout >= 256)) return launch<1, 2, 32, 5>(a, st);   // pconv1_1
    if (mt == 1) return launch_tw<1, 2>(a, st);
    if (mt == 2) return launch_tw<2, 2>(a, st);
    return launch_tw<4, 2>(a, st);
}
