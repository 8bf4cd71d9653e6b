// Data gradient of the encoder's stride-2 3x3 convs (pconv2_1 16->32, pconv3_1 32->64; EEMFlow.py:77,80 under autograd):
//     dX[n][ci][iy][ix] = sum_{co, ky, kx} W[co][ci][ky][kx] * dY[n][co][(iy + 1 - ky) / 2][(ix + 1 - kx) / 2]
// over the taps whose numerators are even.  The generic conv kernel evaluates that parity test per tap and lane (10 TFLOP/s);
// here the four parity classes of (iy, ix) are four small dense convs of dY - 1, 2, 2 and 4 taps - that write interleaved
// pixels of dX:  iy even: ky = 1, oy = iy/2;   iy odd: ky = 0 -> oy = (iy+1)/2, ky = 2 -> oy = (iy-1)/2  (same in x).
//
//   * a block walks 16x32 tiles of dX; the 9x17 patch of dY it needs (all couts) is copied to LDS by 16-byte LDS-DMA
//     (plane pitch 208 floats: the four couts of a k-step fall on different bank halves), the weights once per block as
//     [tap][ci tile][co][16 ci];
//   * v_mfma_f32_16x16x4_f32 with M = ci, N = 16 pixels of one dX row and parity, K = 4 couts: per k-step a wave reads
//     8 B fragments (2 row x 2 column offsets x its 2 pixel rows) and 9*MT weight fragments for 18*MT MFMAs; wave w owns
//     class rows w and w+4, all four classes - every wave does all nine taps, no imbalance, no cross-wave sums;
//   * epilogue: the even/odd-column classes of a lane are neighbours in dX -> 8-byte stores; the pooling branch of the
//     stage output (EEMFlow.py:141-146 backwards: + dpool / k^2) and the LeakyReLU' gate of the producing layer are applied
//     here instead of in a separate pass over dX.
#include "common.h"
#include "train.h"

namespace {

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

constexpr int D2_TH = 16, D2_TW = 32;            // dX tile
constexpr int D2_RY = D2_TH / 2 + 1;             // dY rows of a tile
constexpr int D2_CX = 20;                        // dY columns staged (17 needed, 5 pieces)
constexpr int D2_PL = 208;                       // plane pitch: 52 pieces, 45 real; 208 % 32 == 16
constexpr int D2_PQ = D2_PL / 4;

template <int CO, int CI, int NST>
struct D2Cfg {
    static constexpr int MT = CI / 16;
    static constexpr int WFL = 9 * CO * CI;                      // weights in LDS
    static constexpr int SLOTS = CO * D2_PQ;
    static constexpr int NI = (SLOTS + 255) / 256;               // DMA instructions per wave and tile
    static constexpr int STAGE = NI * 256 * 4;
    static_assert((WFL + NST * STAGE) * 4 <= 160 * 1024, "LDS budget");
};

template <int CO, int CI, int NST>
__global__ __launch_bounds__(256) void dgrad_s2_kernel(DgradS2Args a, int tiles_x, int tiles_y) {
    using C = D2Cfg<CO, CI, NST>;
    constexpr int MT = C::MT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;
    float* st0 = lds + C::WFL;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;

    const TileRange tr_ = block_tile_range(tiles_x * tiles_y * a.n, blockIdx.x, gridDim.x);
    if (tr_.count == 0) return;
    TileCoord cur = tile_coord(tr_.first, tiles_x, tiles_y);
    const size_t yhw = (size_t)a.hout * a.wout, xhw = (size_t)a.hin * a.win;

    // weights: wl[((tap * MT + mt) * CO + co) * 16 + i] = W[co][mt*16 + i][tap]
    for (int e = threadIdx.x; e < C::WFL; e += 256) {
        const int i = e & 15, co = (e >> 4) % CO, tm = (e >> 4) / CO;
        const int mt = tm % MT, tap = tm / MT;
        wl[e] = a.w[((size_t)co * CI + mt * 16 + i) * 9 + tap];
    }

    // DMA plan
    int off[C::NI], rc[C::NI];
#pragma unroll
    for (int k = 0; k < C::NI; ++k) {
        const int f = (wave + 4 * k) * 64 + lane;
        const int co = f / D2_PQ, q = f - co * D2_PQ;
        const int row = q / 5, pc = q - row * 5;
        const bool ok = f < C::SLOTS && q < D2_RY * 5;
        off[k] = (int)(((size_t)co * yhw + (size_t)row * a.wout + 4 * pc) * 4);
        rc[k] = ok ? (row | ((4 * pc) << 8)) : -1;
    }
    const char* zero = reinterpret_cast<const char*>(a.zero_page);
    const float* __restrict__ gate = a.gate;
    const float* __restrict__ dpool = a.dpool;
    float* __restrict__ dx = a.dx;
    const float pool_scale = a.dpool ? 1.f / (float)(a.pool_k * a.pool_k) : 0.f;
    auto issue = [&](int stage, const TileCoord& tc) {
        const int oy0 = tc.by * (D2_TH / 2), ox0 = tc.bx * (D2_TW / 2);
        const char* yb = reinterpret_cast<const char*>(a.dy + (size_t)tc.n * CO * yhw) + ((size_t)oy0 * a.wout + ox0) * 4;
        float* sb = st0 + stage * C::STAGE;
#pragma unroll
        for (int k = 0; k < C::NI; ++k) {
            const bool ok = rc[k] >= 0 && oy0 + (rc[k] & 255) < a.hout && ox0 + (rc[k] >> 8) < a.wout;
            const char* p = ok ? yb + off[k] : zero;
            __builtin_amdgcn_global_load_lds(GLB_PTR(p), LDS_PTR(sb + (wave + 4 * k) * 256), 16, 0, 0);
        }
    };

    issue(0, cur);
    TileCoord nxt = cur;
    const int aoff = g * 16 + j;                                 // weight fragment: co = 4kk + g, ci = j
    const int boff = g * D2_PL + j;                              // dY fragment: co = 4kk + g, column j

#pragma unroll 1
    for (int it = 0; it < tr_.count; ++it) {
        // tile `it` must have landed.  With two stages its DMA was issued before the previous tile's 16*MT stores, which
        // may stay in flight (vmcnt retires in order); with one stage the DMA is the youngest operation
        if (NST == 2 && it > 0) wait_vmcnt<16 * MT>();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                         // tile `it` landed (first pass: the weights too)
        const float* sb = st0 + (NST == 2 ? (it & 1) : 0) * C::STAGE;
        if (NST == 2 && it + 1 < tr_.count) {
            tile_advance(nxt, tiles_x, tiles_y);
            issue((it + 1) & 1, nxt);
        }

        // gate and pooling-branch values of this tile are requested before the MFMA loop (their latency hides behind it);
        // a 16-row tile lies inside one pooling row (pool_k % 16 == 0) and a lane's two columns inside one pooling cell
        const int iy0 = cur.by * D2_TH, ix0 = cur.bx * D2_TW;
        const int ix = ix0 + 2 * j;
        const bool colok = ix < a.win;
        f32x2 gt[2][2][MT][4];
        float pv[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = mt * 16 + 4 * g + r;
                const size_t plane = ((size_t)cur.n * CI + ci);
                // every load unconditional, from a clamped (always valid) address, the condition applied to the value: a load inside a
                // branch is followed by the compiler's s_waitcnt vmcnt(0) - twenty dependent round trips per tile, each also waiting for
                // the next tile's DMA (346x260 batch 32: 291 -> 100 us for the 32 -> 16 layer)
                {
                    const int py = iy0 / a.pool_k, px = ix / a.pool_k;
                    const bool pok = dpool && py < a.gh && px < a.gw;
                    const float* pp = dpool ? dpool + (plane * a.gh + min(py, a.gh - 1)) * a.gw + min(px, a.gw - 1) : a.zero_page;
                    const float pvv = *pp;
                    pv[mt][r] = pok ? pvv * pool_scale : 0.f;
                }
#pragma unroll
                for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int iy = iy0 + 2 * (wave + 4 * t) + pr;
                        const bool gok = gate && colok && iy < a.hin;
                        const float* gp = gate ? gate + plane * xhw + (size_t)min(iy, a.hin - 1) * a.win + (colok ? ix : 0) : a.zero_page;
                        const f32x2 gv = *reinterpret_cast<const f32x2*>(gp);
                        gt[pr][t][mt][r] = gok ? gv : f32x2{1.f, 1.f};
                    }
            }

        f32x4 acc[2][2][2][MT];                                  // [row parity][column parity][pixel row of the wave][ci tile]
#pragma unroll
        for (int e = 0; e < 8 * MT; ++e) (&acc[0][0][0][0])[e] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 2
        for (int kk = 0; kk < CO / 4; ++kk) {
            float bv[2][2][2];                                   // [row offset][column offset][pixel row]
#pragma unroll
            for (int ro = 0; ro < 2; ++ro)
#pragma unroll
                for (int cofs = 0; cofs < 2; ++cofs)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        bv[ro][cofs][t] = sb[boff + kk * 4 * D2_PL + (wave + 4 * t + ro) * D2_CX + cofs];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int pr = ky != 1, pcn = kx != 1;       // parity class this tap feeds
                    const int ro = ky == 0, cofs = kx == 0;      // dY offset: +1 for tap 0, 0 for taps 1 and 2
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const float av = wl[((ky * 3 + kx) * MT + mt) * CO * 16 + kk * 64 + aoff];
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            acc[pr][pcn][t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[ro][cofs][t], acc[pr][pcn][t][mt], 0, 0, 0);
                    }
                }
        }

        // ---- epilogue: D[ci = 4g + r][pixel j]; the two column parities of a lane are neighbours in dX
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int iy = iy0 + 2 * (wave + 4 * t) + pr;
                const bool inside = iy < a.hin && colok;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ci = mt * 16 + 4 * g + r;
                        const size_t o = ((size_t)cur.n * CI + ci) * xhw + (size_t)iy * a.win + ix;
                        f32x2 v = {acc[pr][0][t][mt][r] + pv[mt][r], acc[pr][1][t][mt][r] + pv[mt][r]};
                        v[0] *= gt[pr][t][mt][r][0] > 0.f ? 1.f : 0.1f;
                        v[1] *= gt[pr][t][mt][r][1] > 0.f ? 1.f : 0.1f;
                        float* q = inside ? dx + o : a.trash + lane * 2;            // uniform store count for the counted wait
                        *reinterpret_cast<f32x2*>(q) = v;
                    }
            }
        tile_advance(cur, tiles_x, tiles_y);
        if (NST == 1 && it + 1 < tr_.count) {
            __syncthreads();                                     // everyone is done reading the single stage
            issue(0, cur);
        }
    }
}

template <int CO, int CI, int NST>
int launch(const DgradS2Args& a, hipStream_t st) {
    using C = D2Cfg<CO, CI, NST>;
    const int tiles_x = ceil_div(a.win, D2_TW), tiles_y = ceil_div(a.hin, D2_TH);
    const int T = tiles_x * tiles_y * a.n;
    const int lds_bytes = (C::WFL + NST * C::STAGE) * 4;
    int per_cu = (160 * 1024) / lds_bytes;
    per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
    int gx = 256 * per_cu;
    const int need = ceil_div(T, 8) * 8;
    if (gx > need) gx = need;
    static bool raised = false;
    if (!raised) {
        EEM_HIP_CHECK(hipFuncSetAttribute((const void*)dgrad_s2_kernel<CO, CI, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised = true;
    }
    hipLaunchKernelGGL((dgrad_s2_kernel<CO, CI, NST>), dim3(gx), dim3(256), lds_bytes, st, a, tiles_x, tiles_y);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}


// ------------------------------------------------------------------------------------------------ wide layers (E-RAFT's encoders)
// The same four parity classes for model/extractor.py's downsampling convs (layer2: 64 <- 96, layer3: 96 <- 128 channels; the 3x3 conv of
// the residual block and its 1x1 stride-2 shortcut, :13,:33-36 under autograd) - round 6; before, these ran the generic kernel's per-tap
// parity test (830 us x 8 per training step, 20 x their MACs' time).  Differences to the kernel above: a block owns ONE 16-channel tile
// of dX (blockIdx.y) with the weights of ALL couts resident in LDS ([tap][co][16 ci]: 55 / 74 KB), and walks its tiles with the dY patch
// arriving in chunks of 32 couts through two LDS stages (the next chunk - or the next tile's first - is requested right after the barrier
// that hands over the current one); K1 = the 1x1 shortcut: one tap, one class, the other three classes of dX are zeros.
constexpr int D2W_KC = 32;                                   // couts per staged chunk
template <int CO, bool K1>
struct D2WCfg {
    static constexpr int TAPS = K1 ? 1 : 9;
    static constexpr int NCH = CO / D2W_KC;
    static constexpr int WFL = TAPS * CO * 16;
    static constexpr int SLOTS = D2W_KC * D2_PQ;
    static constexpr int NI = (SLOTS + 255) / 256;
    static constexpr int STAGE = NI * 256 * 4;
    static_assert(CO % D2W_KC == 0, "cout chunks");
    static_assert((WFL + 2 * STAGE) * 4 <= 160 * 1024, "LDS budget");
};

template <int CO, bool K1>
__global__ __launch_bounds__(256) void dgrad_s2w_kernel(DgradS2Args a, int tiles_x, int tiles_y) {
    using C = D2WCfg<CO, K1>;
    constexpr int TAPS = C::TAPS, NCH = C::NCH;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;
    float* st0 = lds + C::WFL;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int ci0 = blockIdx.y * 16;

    const TileRange tr_ = block_tile_range(tiles_x * tiles_y * a.n, blockIdx.x, gridDim.x);
    if (tr_.count == 0) return;
    TileCoord cur = tile_coord(tr_.first, tiles_x, tiles_y);
    const size_t yhw = (size_t)a.hout * a.wout, xhw = (size_t)a.hin * a.win;

    // weights: wl[(tap * CO + co) * 16 + i] = W[co][ci0 + i][tap]   (channels past cin read as zero)
    for (int e = threadIdx.x; e < C::WFL; e += 256) {
        const int i = e & 15, co = (e >> 4) % CO, tap = (e >> 4) / CO;
        wl[e] = ci0 + i < a.cin ? a.w[((size_t)co * a.cin + ci0 + i) * TAPS + tap] : 0.f;
    }

    int off[C::NI], rc[C::NI];
#pragma unroll
    for (int k = 0; k < C::NI; ++k) {
        const int f = (wave + 4 * k) * 64 + lane;
        const int co = f / D2_PQ, q = f - co * D2_PQ;
        const int row = q / 5, pc = q - row * 5;
        const bool ok = f < C::SLOTS && q < D2_RY * 5;
        off[k] = (int)(((size_t)co * yhw + (size_t)row * a.wout + 4 * pc) * 4);
        rc[k] = ok ? (row | ((4 * pc) << 8)) : -1;
    }
    const char* zero = reinterpret_cast<const char*>(a.zero_page);
    float* __restrict__ dx = a.dx;
    auto issue = [&](int stage, const TileCoord& tc, int ch) {
        const int oy0 = tc.by * (D2_TH / 2), ox0 = tc.bx * (D2_TW / 2);
        const char* yb = reinterpret_cast<const char*>(a.dy + ((size_t)tc.n * CO + ch * D2W_KC) * yhw) + ((size_t)oy0 * a.wout + ox0) * 4;
        float* sb = st0 + stage * C::STAGE;
#pragma unroll
        for (int k = 0; k < C::NI; ++k) {
            const bool ok = rc[k] >= 0 && oy0 + (rc[k] & 255) < a.hout && ox0 + (rc[k] >> 8) < a.wout;
            const char* p = ok ? yb + off[k] : zero;
            __builtin_amdgcn_global_load_lds(GLB_PTR(p), LDS_PTR(sb + (wave + 4 * k) * 256), 16, 0, 0);
        }
    };

    issue(0, cur, 0);
    TileCoord nxt = cur;
    int nch_next = 0;                                            // chunk the NEXT step stages
    const int aoff = g * 16 + j;                                 // weight fragment: co = 4kk + g, ci = j
    const int boff = g * D2_PL + j;                              // dY fragment: co = 4kk + g, column j
    f32x4 acc[2][2][2];                                          // [row parity][column parity][pixel row of the wave]
    bool stored = false;                                         // the previous step ended with the tile's stores (they may stay in flight)
    const int steps = tr_.count * NCH;
    int ch = 0;
#pragma unroll 1
    for (int t = 0; t < steps; ++t) {
        if (stored) wait_vmcnt<16>();                            // the step's DMA is older than the 16 stores behind it: it has landed
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                         // (first pass: the weights too)
        const float* sb = st0 + (t & 1) * C::STAGE;
        if (t + 1 < steps) {
            if (++nch_next == NCH) { nch_next = 0; tile_advance(nxt, tiles_x, tiles_y); }
            issue((t + 1) & 1, nxt, nch_next);
        }
        if (ch == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) (&acc[0][0][0])[e] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const float* wc = wl + ch * D2W_KC * 16;
#pragma unroll 2
        for (int kk = 0; kk < D2W_KC / 4; ++kk) {
            float bv[2][2][2];                                   // [row offset][column offset][pixel row]
#pragma unroll
            for (int ro = 0; ro < (K1 ? 1 : 2); ++ro)
#pragma unroll
                for (int cofs = 0; cofs < (K1 ? 1 : 2); ++cofs)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
                        bv[ro][cofs][tt] = sb[boff + kk * 4 * D2_PL + (wave + 4 * tt + ro) * D2_CX + cofs];
            if (K1) {
                const float av = wc[kk * 64 + aoff];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
                    acc[0][0][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[0][0][tt], acc[0][0][tt], 0, 0, 0);
            } else {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int pr = ky != 1, pcn = kx != 1;   // parity class this tap feeds
                        const int ro = ky == 0, cofs = kx == 0;  // dY offset: +1 for tap 0, 0 for taps 1 and 2
                        const float av = wc[(ky * 3 + kx) * CO * 16 + kk * 64 + aoff];
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt)
                            acc[pr][pcn][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[ro][cofs][tt], acc[pr][pcn][tt], 0, 0, 0);
                    }
            }
        }
        stored = false;
        if (++ch == NCH) {
            ch = 0;
            // ---- epilogue: D[ci = 4g + r][pixel j]; the two column parities of a lane are neighbours in dX
            const int iy0 = cur.by * D2_TH, ix0 = cur.bx * D2_TW;
            const int ix = ix0 + 2 * j;
            const bool colok = ix < a.win;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int iy = iy0 + 2 * (wave + 4 * tt) + pr;
                    const bool inside = iy < a.hin && colok;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ci = ci0 + 4 * g + r;
                        const size_t o = ((size_t)cur.n * a.cin + ci) * xhw + (size_t)iy * a.win + ix;
                        const f32x2 v = {acc[pr][0][tt][r], acc[pr][1][tt][r]};
                        float* q = (inside && ci < a.cin) ? dx + o : a.trash + lane * 2;       // uniform store count for the counted wait
                        *reinterpret_cast<f32x2*>(q) = v;
                    }
                }
            stored = true;
            tile_advance(cur, tiles_x, tiles_y);
        }
    }
}

template <int CO, bool K1>
int launch_wide(const DgradS2Args& a, hipStream_t st) {
    using C = D2WCfg<CO, K1>;
    const int tiles_x = ceil_div(a.win, D2_TW), tiles_y = ceil_div(a.hin, D2_TH);
    const int T = tiles_x * tiles_y * a.n;
    const int lds_bytes = (C::WFL + 2 * C::STAGE) * 4;
    const int citiles = ceil_div(a.cin, 16);
    int per_cu = (160 * 1024) / lds_bytes;
    per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
    int gx = (256 * per_cu) / citiles;                           // ~per_cu resident blocks per CU over all channel tiles
    gx = (gx + 7) & ~7;
    if (gx < 8) gx = 8;
    const int need = ceil_div(T, 8) * 8;
    if (gx > need) gx = need;
    static bool raised = false;
    if (!raised) {
        EEM_HIP_CHECK(hipFuncSetAttribute((const void*)dgrad_s2w_kernel<CO, K1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised = true;
    }
    hipLaunchKernelGGL((dgrad_s2w_kernel<CO, K1>), dim3(gx, citiles), dim3(256), lds_bytes, st, a, tiles_x, tiles_y);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

}  // namespace

bool dgrad_s2_supported(const DgradS2Args& a) {
    const char* e = getenv("EEM_NO_DGRAD_S2");                      // read per call: a test flips it inside one process
    if (e && e[0] == '1') return false;
    const bool shape = (a.cout == 32 && a.cin == 16) || (a.cout == 64 && a.cin == 32);
    return shape && a.zero_page && a.wout % 4 == 0 && a.win % 2 == 0 && a.hout == (a.hin + 1) / 2 && a.wout == (a.win + 1) / 2 &&
           ((uintptr_t)a.dy & 15) == 0 && ((uintptr_t)a.dx & 7) == 0 && (a.gate == nullptr || ((uintptr_t)a.gate & 7) == 0) &&
           (size_t)a.cout * a.hout * a.wout * 4 < (1u << 31) && (!a.dpool || (a.pool_k >= 16 && a.pool_k % 16 == 0));
}

int dgrad_s2_launch(const DgradS2Args& a, hipStream_t st) {
    if (a.cout == 32) return launch<32, 16, 2>(a, st);
    return launch<64, 32, 1>(a, st);
}

// wide layers: cout 96 or 128, any cin; k = 3 (pad 1) or k = 1 (pad 0): DgradS2Args::pool_k carries the kernel size here (no pooling branch)
bool dgrad_s2w_supported(const DgradS2Args& a, int ksize) {
    const char* e = getenv("EEM_NO_DGRAD_S2W");                     // read per call: a test flips it inside one process
    if (e && e[0] == '1') return false;
    return (a.cout == 96 || a.cout == 128) && (ksize == 3 || ksize == 1) && a.zero_page && a.trash && a.gate == nullptr && a.dpool == nullptr &&
           a.wout % 4 == 0 && a.win % 2 == 0 && a.hout == (a.hin + 1) / 2 && a.wout == (a.win + 1) / 2 && ((uintptr_t)a.dy & 15) == 0 &&
           ((uintptr_t)a.dx & 7) == 0 && (size_t)a.cout * a.hout * a.wout * 4 < (1u << 31);
}

int dgrad_s2w_launch(const DgradS2Args& a, int ksize, hipStream_t st) {
    if (ksize == 1) return a.cout == 96 ? launch_wide<96, true>(a, st) : launch_wide<128, true>(a, st);
    return a.cout == 96 ? launch_wide<96, false>(a, st) : launch_wide<128, false>(a, st);
}
