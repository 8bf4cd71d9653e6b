// Winograd F(2x2,3x3) for the C = 32 and C = 64 stride-1 encoder layers (pconv2_2/2_3, pconv3_2/3_3,
// EEMFlow.py:78-79,81-82) on v_mfma_f32_32x32x2_f32.  Same algebra and data path as conv_wino.hip (weights
// stationary in registers, input tile by LDS-DMA, V = B^T d B computed in the lane that feeds it to the MFMA,
// output transform on the accumulators), re-cut so that the matrix pipe, not VALU issue, is the limit:
//   * M = 32 couts per MFMA: one transformed input value feeds 32 output channels, so the transform costs
//     8 VALU + 6 LDS reads per 4 MFMAs of 64 cycles (the 16x16x4 form pays twice that per MFMA cycle);
//   * the 16 Winograd positions are split by ROWS over 4 waves (xi = 0..3): a wave needs two patch rows per
//     k-step (t_xi = e_a + sgn * e_b), owns 4 positions (64 accumulator registers) and their weights for all
//     cin (C/2 k-steps x 4 = 64 / 128 VGPRs);
//   * the four waves of a team exchange u_xi = M_xi A (2 values per cout and tile) through the LDS stage they
//     have just finished reading; wave xi then finishes 4 of the lane's 16 cout rows: Y0 = u0+u1+u2,
//     Y1 = u1-u2-u3, bias, LeakyReLU, [gate], float2 NCHW stores, pooling partial sums;
//   * patch reads are three aligned ds_read_b64 per row (columns -1..4), no lane-dependent selects.
#include <type_traits>

#include "common.h"

namespace {

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int C, int TH, int TW, int NGH, int WAVES>
struct W32Cfg {
    static constexpr int COG = C / 32;                   // 32-cout groups
    static constexpr int KS = C / 2;                     // k-steps (2 cin each)
    static constexpr int TEAM = COG * 4;                 // waves sharing one group of 32 tiles
    static constexpr int SLOTS = WAVES / TEAM;
    static constexpr int NGW = NGH == 1 ? 64 : 32;       // pixel width of a tile group (32 tiles = 64x2 or 32x4 pixels)
    static constexpr int NGX = TW / NGW, NGY = TH / (2 * NGH);
    static constexpr int IN_ROWS = TH + 2;
    static constexpr int ROWP = TW + 8;
    static constexpr int PPR = ROWP / 4;
    static constexpr int PLANE = IN_ROWS * ROWP;
    static constexpr int PC = IN_ROWS * PPR;             // 16-byte pieces per channel
    // One DMA instruction of the whole block (WAVES x 64 lanes) moves CPI whole channels; lane slot = wave * 64 + lane always holds
    // the same (channel-in-group, tile row, piece) - instruction k only adds k * CPI channels to the address, so a tile's DMA costs
    // one decomposition + 2 VALU per instruction instead of a decomposition and a bounds test per piece (~200 -> ~45 VALU per wave
    // and tile, which the matrix pipe pays for: f32 MFMA and VALU issue add up on gfx950).  Slots >= CPI * PC are padding.
    static constexpr int SLOTS_I = WAVES * 64;
    static constexpr int CPI = (SLOTS_I / PC) & ~1;      // even: the two channels of a k-step share an instruction
    static constexpr int NI = C / CPI;
    static constexpr int XCH = WAVES * 64 * 32;          // exchange floats (lives in the finished stage)
    static constexpr int STAGE = NI * WAVES * 256;
    static constexpr int chan_off(int c) { return ((c / CPI) * SLOTS_I + (c % CPI) * PC) * 4; }   // floats from the stage base
    // tile 0 of a block starts its k-loop before the whole tile has landed: DMA instruction k of every wave (channels
    // [k * CPI, (k + 1) * CPI)) must have landed before k-step s reads channels 2s, 2s+1
    static constexpr int kq(int s) {
        int need = (2 * s + 1) / CPI;
        need |= 1;                                       // sync points after instructions 1, 3, 5, ...
        return need < NI ? need : NI - 1;
    }
    static_assert(CPI >= 2 && C % CPI == 0 && STAGE >= XCH, "channel groups per DMA instruction");
    static_assert(WAVES % TEAM == 0 && NGX * NGY == SLOTS, "one tile group per wave and block tile");
    static_assert(TW % NGW == 0 && TH % (2 * NGH) == 0, "block tile");
    static_assert((chan_off(C - 2) + PLANE) * 4 + 3 * ROWP * 4 + 64 < 65536, "ds_read immediate range");
};

#ifdef EEM_STAMPS
// diagnostic build only: per-wave s_memtime stamps (kept in SGPRs, written once at the very end)
__device__ unsigned long long g_stamps[2048 * 8 * 8];
#define STAMP(i) st[i] = __builtin_amdgcn_s_memtime()
#else
#define STAMP(i)
#endif

// KEEP = false: the pooling partial sums are the launch's only output (EncConvArgs::no_store), no feature-map stores
template <int C, int TH, int TW, int NGH, int WAVES, bool STREAM, int POOLK, bool KEEP = true>
__global__ __launch_bounds__(WAVES * 64) void wino32_kernel(EncConvArgs a) {
    ENC_ARGS_NOW(a);
#ifdef EEM_STAMPS
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    st[7] = __builtin_amdgcn_s_memrealtime();
#endif
    STAMP(0);
    using K = W32Cfg<C, TH, TW, NGH, WAVES>;
    constexpr int NWX = POOLK > 0 ? TW / POOLK : 1;
    constexpr int RED = POOLK > 0 ? (TH / 2) * C * NWX : 0;
    constexpr int NS = (KEEP ? 8 : 0) + (POOLK > 0 ? 1 : 0);             // stores per wave and tile
    static_assert(KEEP || POOLK > 0, "a store-free launch needs the pooling output");
    static_assert((2 * K::STAGE + RED) * 4 <= 160 * 1024, "LDS budget");
    static_assert(POOLK == 0 || (POOLK % TH == 0 && TW % POOLK == 0 && C * NWX <= WAVES * 64), "pool windows");
    __shared__ __attribute__((aligned(16))) float lds[2 * K::STAGE + RED];
    float* red = lds + 2 * K::STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 31, kk = lane >> 5;
    const int slot = wave / K::TEAM;
    const int tw_ = wave % K::TEAM;
    const int cog = tw_ >> 2, xi = tw_ & 3;
    const int tx = NGH == 1 ? nl : (nl & 15), ty = NGH == 1 ? 0 : (nl >> 4);
    const int tr = (slot / K::NGX) * NGH + ty;                           // tile row inside the block tile
    const int txb = (slot % K::NGX) * (K::NGW / 2) + tx;                 // tile column inside the block tile

    const TileRange tr_ = block_tile_range(a.tiles_x * a.tiles_y * a.nimg, blockIdx.x, gridDim.x);
    const int ntile = tr_.count;
    if (ntile == 0) return;
    TileCoord cur = tile_coord(tr_.first, a.tiles_x, a.tiles_y), nxt = cur;      // tile being computed / next to request
    const float* zero_page = a.zero_page;

    // ---- DMA: see W32Cfg - lane slot = (channel-in-group cl, tile row ry, 16-byte column q) for every instruction
    auto issue = [&](int it, const TileCoord& tc) {
        const int bx = tc.bx, by = tc.by, n = tc.n;
        const int gy0 = by * TH - 1, gxa = bx * TW - 4;
        const int slot = wave * 64 + lane;
        const int cl = slot / K::PC;
        const int rem = slot - cl * K::PC;
        const int ry = rem / K::PPR;
        const int q = rem - ry * K::PPR;
        const int gy = gy0 + ry, gx = gxa + q * 4;
        const bool ok = cl < K::CPI && gy >= 0 && gy < a.hin && gx >= 0 && gx < a.win;
        const int plane = a.hin * a.win;
        const char* gp = ok ? reinterpret_cast<const char*>(a.in0 + (size_t)n * C * plane + (gy0 * a.win + gxa) + ((cl * a.hin + ry) * a.win + q * 4))
                            : reinterpret_cast<const char*>(zero_page);
        const unsigned step = ok ? (unsigned)(K::CPI * plane) * 4u : 0u;     // bytes to the same piece of the next channel group
        float* sbase = lds + (it & 1) * K::STAGE;
#pragma unroll
        for (int k = 0; k < K::NI; ++k) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(gp), LDS_PTR(sbase + (wave + k * WAVES) * 256), 16, 0, 0);
            gp += step;
        }
    };

    // bias of the 4 cout rows this wave finishes: co = cog*32 + r' + 8*xi + 4*kk
    float biasv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) biasv[r] = a.bias[cog * 32 + r + 8 * xi + 4 * kk];
    issue(0, nxt);
    // Weights: one float4 (nu = 0..3) per k-step.  They are NOT requested here: a CU serves its waves' requests in
    // issue order, so 8 x 32 KB of weight loads queued behind the first waves' DMA would hold back the other waves'
    // input pieces (measured: tile 0 landed after 10-13k cycles).  They are requested inside the k-loop, H k-steps
    // ahead of use; the compiler counts vmcnt for them.
    //   STREAM = false (C = 32): H = WD for tile 0 only; the KS registers then stay stationary for the block's life.
    //   STREAM = true  (C = 64): a ring of H = KS/2 k-steps, refilled in every tile - 128 stationary registers would
    //   leave the k-loop no room (spills, LDS reads serialised against their use).
    constexpr int WD = 8;
    constexpr int H = STREAM ? K::KS / 2 : WD;       // request distance in k-steps
    constexpr int NWR = STREAM ? H : K::KS;          // weight registers (float4)
    f32x4 wr[NWR];
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(a.wwino) + (size_t)(cog * 4 + xi) * K::KS * 64 + lane;
    // patch rows of this wave: t_xi = e_a + sgn * e_b with (a, b, sgn) = (0,2,-) (1,2,+) (2,1,-) (1,3,-)
    const int ra = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
    const int rb = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
    const float sgn = xi == 1 ? 1.f : -1.f;
    const int lbase = kk * K::PLANE + 2 * tr * K::ROWP + 2 * txb + 2;    // patch column -1 (8-byte aligned)

    auto tile = [&](int it, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        if constexpr (!FIRST) {
            if (it + 1 < ntile) {                    // its stage was last read by tile it-1: free since the barrier
                tile_advance(nxt, a.tiles_x, a.tiles_y);
                issue(it + 1, nxt);
            }
        }
        const int bx = cur.bx, by = cur.by, n = cur.n;
        float* stage = lds + (it & 1) * K::STAGE;
        const float* pa = stage + lbase + ra * K::ROWP;
        const float* pb = stage + lbase + rb * K::ROWP;

        f32x16 acc[4];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nu][r] = 0.f;

        f32x2 na[3], nb[3];
        auto load_patch = [&](int s) {
            const int off = K::chan_off(2 * s);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                na[q] = *reinterpret_cast<const f32x2*>(pa + off + 2 * q);
                nb[q] = *reinterpret_cast<const f32x2*>(pb + off + 2 * q);
            }
        };
        if constexpr (FIRST) {
            // younger than the needed DMA instructions at this point: the rest of this tile's DMA (weights come later)
            wait_vmcnt<K::NI - 1 - K::kq(0)>();
            __builtin_amdgcn_s_barrier();
        }
        load_patch(0);
        constexpr bool LOADW = FIRST || STREAM;      // this tile requests weights
        if constexpr (LOADW) {
#pragma unroll
            for (int s = 0; s < H && s < K::KS; ++s) wr[s % NWR] = wsrc[s * 64];
        }
        static_for<0, K::KS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            const float ea[4] = {na[0][1], na[1][0], na[1][1], na[2][0]};
            const float eb[4] = {nb[0][1], nb[1][0], nb[1][1], nb[2][0]};
            if constexpr (s + 1 < K::KS) {
                if constexpr (FIRST && K::kq(s + 1) > K::kq(s)) {
                    // the next k-step's channels are in a later DMA group: younger ops = remaining DMA + weights so far
                    constexpr int WL = s + H < K::KS ? s + H : K::KS;       // weight loads requested so far
                    wait_vmcnt<K::NI - 1 - K::kq(s + 1) + WL>();
                    __builtin_amdgcn_s_barrier();
                }
                load_patch(s + 1);
                // keep these reads ahead of this k-step's transform and MFMAs (the scheduler otherwise sinks them
                // next to their use and exposes the LDS latency)
                __builtin_amdgcn_sched_barrier(0);
            }
            float t[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) t[b] = __builtin_fmaf(sgn, eb[b], ea[b]);
            const float v[4] = {t[0] - t[2], t[1] + t[2], t[2] - t[1], t[1] - t[3]};
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[s % NWR][nu], v[nu], acc[nu], 0, 0, 0);
            if constexpr (LOADW && s + H < K::KS) wr[(s + H) % NWR] = wsrc[(s + H) * 64];   // slot free: its MFMAs have issued
        });
        if constexpr (FIRST) {
            if (ntile > 1) {                         // after tile 0's last counted wait (keeps those counts exact)
                tile_advance(nxt, a.tiles_x, a.tiles_y);
                issue(1, nxt);
            }
        }

        if constexpr (FIRST) STAMP(3);
        // ---- u = M_xi A, exchanged through the stage this tile has finished with: [wave][r][lane][2]
        __builtin_amdgcn_s_barrier();                                    // every wave is done reading the stage
        f32x2* xw = reinterpret_cast<f32x2*>(stage) + (wave * 16) * 64 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            f32x2 u;
            u[0] = acc[0][r] + acc[1][r] + acc[2][r];
            u[1] = acc[1][r] - acc[2][r] - acc[3][r];
            xw[r * 64] = u;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (FIRST) STAMP(4);
        // wave xi finishes accumulator rows 4*xi .. 4*xi+3 of its team
        const f32x2* xr = reinterpret_cast<const f32x2*>(stage) + ((wave - xi) * 16 + 4 * xi) * 64 + lane;
        const int oy = by * TH + 2 * tr, ox = bx * TW + 2 * txb;
        const bool in0 = oy < a.hout && ox < a.wout, in1 = oy + 1 < a.hout && ox < a.wout;
        const int hw = a.hout * a.wout;
        const int co0 = cog * 32 + 8 * xi + 4 * kk;
        float* dst = a.out + (size_t)n * C * hw;
        const float* gsrc = a.gate ? a.gate + (size_t)n * C * hw : nullptr;
        const int o0 = (co0 * a.hout + oy) * a.wout + ox;
        const bool full = by * TH + TH <= a.hout && bx * TW + TW <= a.wout;         // wave-uniform
        const unsigned lane_bo = (unsigned)o0 * 4u;
        float psum[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x2 u0 = xr[(0 * 16 + r) * 64], u1 = xr[(1 * 16 + r) * 64], u2 = xr[(2 * 16 + r) * 64], u3 = xr[(3 * 16 + r) * 64];
            float y00 = u0[0] + u1[0] + u2[0] + biasv[r], y01 = u0[1] + u1[1] + u2[1] + biasv[r];
            float y10 = u1[0] - u2[0] - u3[0] + biasv[r], y11 = u1[1] - u2[1] - u3[1] + biasv[r];
            if (a.act) {
                y00 = fmaxf(y00, 0.1f * y00); y01 = fmaxf(y01, 0.1f * y01);
                y10 = fmaxf(y10, 0.1f * y10); y11 = fmaxf(y11, 0.1f * y11);
            }
            const int o = o0 + r * hw;
            if (gsrc) {
                if (in0) {
                    const f32x2 gt = *reinterpret_cast<const f32x2*>(gsrc + o);
                    y00 *= gt[0] > 0.f ? 1.f : 0.1f; y01 *= gt[1] > 0.f ? 1.f : 0.1f;
                }
                if (in1) {
                    const f32x2 gt = *reinterpret_cast<const f32x2*>(gsrc + o + a.wout);
                    y10 *= gt[0] > 0.f ? 1.f : 0.1f; y11 *= gt[1] > 0.f ? 1.f : 0.1f;
                }
            }
            psum[r] = (y00 + y01) + (y10 + y11);
            if constexpr (!KEEP) {
                // (no store)
            } else if (full) {
                // tile inside the image: a scalar base per (cout register, row) + ONE 32-bit lane offset - no 64-bit vector
                // arithmetic, no selects
                char* rb = reinterpret_cast<char*>(dst) + (size_t)r * hw * 4;
                *reinterpret_cast<f32x2*>(rb + lane_bo) = f32x2{y00, y01};
                *reinterpret_cast<f32x2*>(rb + (size_t)a.wout * 4 + lane_bo) = f32x2{y10, y11};
            } else {
                // every lane stores (outside lanes into a scratch page): exactly 8 stores per wave and tile
                float* p0 = in0 ? dst + o : a.trash + lane * 2;
                float* p1 = in1 ? dst + o + a.wout : a.trash + lane * 2;
                *reinterpret_cast<f32x2*>(p0) = f32x2{y00, y01};
                *reinterpret_cast<f32x2*>(p1) = f32x2{y10, y11};
            }
        }
        if constexpr (POOLK > 0) {
            constexpr int SW = POOLK / 2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sred = lane_group_sum<SW>(psum[r]);
                if ((tx & (SW - 1)) == 0) red[(tr * C + co0 + r) * NWX + (2 * txb) / POOLK] = sred;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            float s = 0.f;
            const bool act = tid < C * NWX;
            const int co = act ? tid / NWX : 0, wx = act ? tid - co * NWX : 0;
#pragma unroll
            for (int q = 0; q < TH / 2; ++q) s += red[(q * C + co) * NWX + wx];
            float* p = act ? a.pool_partial + (((size_t)n * C + co) * a.tiles_y + by) * (a.tiles_x * NWX) + bx * NWX + wx
                           : a.trash + lane * 2;
            *p = s;
        }
    };

    STAMP(1);
    STAMP(2);
    tile(0, std::true_type{});
    tile_advance(cur, a.tiles_x, a.tiles_y);
    STAMP(5);
    if constexpr (!STREAM) {                         // from here on the weights are plain register values
#pragma unroll
        for (int s = 0; s < K::KS; ++s) asm volatile("" : "+v"(wr[s]));
    }
#pragma unroll 1
    for (int it = 1; it < ntile; ++it) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS) : "memory");
        __builtin_amdgcn_s_barrier();
        tile(it, std::false_type{});
        tile_advance(cur, a.tiles_x, a.tiles_y);
    }
#ifdef EEM_STAMPS
    STAMP(6);
    st[7] = __builtin_amdgcn_s_memrealtime() - st[7];
    if (lane == 0 && blockIdx.x < 2048) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int i = 0; i < 8; ++i) g_stamps[(blockIdx.x * 8 + wave) * 8 + i] = st[i];
    }
#endif
}

// U = G g G^T in the fragment order above: [cog][xi][s][lane = (cout % 32) + 32 * (cin % 2)][nu]
__global__ void wino32_wt_kernel(const float* __restrict__ w, int c, int transpose_flip, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= c * c) return;
    const int co = t / c, ci = t - co * c;
    float gk[3][3];
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx)
            gk[ky][kx] = transpose_flip ? w[((size_t)ci * c + co) * 9 + (2 - ky) * 3 + (2 - kx)]
                                        : w[((size_t)co * c + ci) * 9 + ky * 3 + kx];
    float m[4][3];
    for (int kx = 0; kx < 3; ++kx) {
        m[0][kx] = gk[0][kx];
        m[1][kx] = 0.5f * (gk[0][kx] + gk[1][kx] + gk[2][kx]);
        m[2][kx] = 0.5f * (gk[0][kx] - gk[1][kx] + gk[2][kx]);
        m[3][kx] = gk[2][kx];
    }
    const int ks = c / 2;
    const int cog = co >> 5, lane = (co & 31) + 32 * (ci & 1), s = ci >> 1;
    for (int xi = 0; xi < 4; ++xi) {
        float* o = out + ((((size_t)(cog * 4 + xi)) * ks + s) * 64 + lane) * 4;
        o[0] = m[xi][0];
        o[1] = 0.5f * (m[xi][0] + m[xi][1] + m[xi][2]);
        o[2] = 0.5f * (m[xi][0] - m[xi][1] + m[xi][2]);
        o[3] = m[xi][2];
    }
}

template <int C> struct W32Tile;
//                                             TH  TW  NGH WAVES POOLK
template <> struct W32Tile<32> { static constexpr int TH = 4, TW = 64, NGH = 1, WAVES = 8, POOLK = 16; static constexpr bool STREAM = false; };
template <> struct W32Tile<64> { static constexpr int TH = 4, TW = 32, NGH = 2, WAVES = 8, POOLK = 8; static constexpr bool STREAM = true; };

template <int C>
int launch_c(const EncConvArgs& a0, hipStream_t stream) {
    using W = W32Tile<C>;
    EncConvArgs a = a0;
    a.tiles_x = ceil_div(a.wout, W::TW);
    a.tiles_y = ceil_div(a.hout, W::TH);
    const int T = a.tiles_x * a.tiles_y * a.nimg;
    int per_xcd = ceil_div(T, 8);
    static const int env_cap = enc_blocks_per_xcd(C == 32 ? "W32" : "W64", 0);     // tuning override
    const int cap = env_cap > 0 ? env_cap : (a.blocks_per_xcd > 0 ? a.blocks_per_xcd : 32);   // default: one resident block per CU
    if (per_xcd > cap) per_xcd = cap;
    if (a.pool_partial != nullptr && a.pool_k != W::POOLK) {
        eem_set_error("wino32: fused pooling with k=%d is not built for C=%d", a.pool_k, C);
        return EEM_ERR_ARG;
    }
    EEM_NOTE_GRID(per_xcd * 8, W::WAVES * 64);
    EEM_NOTE_PIPE(3);                                    // F(2x2,3x3): 16 products per 4 outputs against the direct form's 36
    if (a.pool_partial != nullptr && a.no_store && C == 64 && a.gate == nullptr)
        hipLaunchKernelGGL((wino32_kernel<C, W::TH, W::TW, W::NGH, W::WAVES, W::STREAM, W::POOLK, C != 64>), dim3(per_xcd * 8),
                           dim3(W::WAVES * 64), 0, stream, a);
    else if (a.pool_partial != nullptr)
        hipLaunchKernelGGL((wino32_kernel<C, W::TH, W::TW, W::NGH, W::WAVES, W::STREAM, W::POOLK>), dim3(per_xcd * 8), dim3(W::WAVES * 64),
                           0, stream, a);
    else
        hipLaunchKernelGGL((wino32_kernel<C, W::TH, W::TW, W::NGH, W::WAVES, W::STREAM, 0>), dim3(per_xcd * 8), dim3(W::WAVES * 64), 0,
                           stream, a);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

}  // namespace

#ifdef EEM_STAMPS
extern "C" __attribute__((visibility("default"))) int eemflow_debug_read_stamps(unsigned long long* dst, size_t n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), n * sizeof(unsigned long long)) == hipSuccess ? 0 : 2;
}
#endif

int wino32_transform_launch(const float* w, int c, int transpose_flip, float* packed, hipStream_t stream) {
    hipLaunchKernelGGL(wino32_wt_kernel, dim3(ceil_div(c * c, 256)), dim3(256), 0, stream, w, c, transpose_flip, packed);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

void wino32_tile(int c, int* th, int* tw, int* poolk) {
    if (c == 32) { *th = W32Tile<32>::TH; *tw = W32Tile<32>::TW; *poolk = W32Tile<32>::POOLK; }
    else { *th = W32Tile<64>::TH; *tw = W32Tile<64>::TW; *poolk = W32Tile<64>::POOLK; }
}

int wino32_launch(int c, const EncConvArgs& a, hipStream_t stream) {
    if (c == 32) return launch_c<32>(a, stream);
    if (c == 64) return launch_c<64>(a, stream);
    eem_set_error("wino32_launch: unsupported C=%d", c);
    return EEM_ERR_ARG;
}
