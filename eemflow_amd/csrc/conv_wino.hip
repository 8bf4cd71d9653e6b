// Winograd F(2x2,3x3) form of the encoder's stride-1 C -> C convolutions (pconv1_2, pconv2_2/2_3, pconv3_2/3_3:
// EEMFlow.py:76,78-79,81-82; 11.3 of the frame's 14.6 GFLOP), fp32 throughout:
//
//     Y = A^T [ sum_cin (G g G^T) (.) (B^T d B) ] A          (2x2 outputs from a 4x4 input patch, 16 products
//                                                             per (cin, cout) instead of 36: 2.25x fewer MFMAs)
//
// One MFMA (v_mfma_f32_16x16x4_f32) per Winograd position p = (xi, nu) and k-step: D_p[16 cout][16 tiles] +=
// U_p[16 cout][4 cin] V_p[4 cin][16 tiles].  How the operands reach the matrix cores:
//   * U (transformed weights) is STATIONARY IN REGISTERS: a wave owns one 16-cout group
//     and keeps its fragments for all positions and cin (64 VGPRs) for its whole life; the kernel is persistent, so they
//     are fetched once per CU.
//   * the input tile (all C channels, rows with halo, 16-byte pieces) is copied HBM/L2 -> LDS by LDS-DMA
//     (global_load_lds_dwordx4), double buffered across the tiles a block walks.
//   * V = B^T d B is computed by each lane for its own B-fragment slot (tile j = lane % 16, channel 4s + lane/16)
//     from 16 LDS reads per k-step and goes straight into the MFMA - it never touches LDS or HBM.
//     LDS reads are conflict-free: patch columns -1..4 are three aligned ds_read_b64 per row, and the channel
//     slots of one half-wave are two planes apart (PLANE % 64 in {16, 48} -> 32 banks apart for b64).
//   * the output transform runs on the accumulators in registers; bias, LeakyReLU, [gate], NCHW stores (float2 per
//     lane, 128-B segments) and the fused stage pooling (DPP lane sums + LDS, as in conv_enc2.hip) follow.
//   This file holds the C = 16 form (pconv1_2: one wave owns all 16 positions of a 16-tile group); C = 32 / 64 run
//   on 32x32x2 MFMAs with the Winograd rows split over four waves (conv_wino32.hip), reached through wino_*.
#include <type_traits>

#include "common.h"

namespace {

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int C, int TH, int TW, int WAVES>
struct WCfg {
    static constexpr int COG = C / 16;                   // 16-cout groups = waves that share one group of 16 tiles
    static constexpr int KS = C / 4;                     // k-steps (4 cin each)
    static constexpr int SLOTS = WAVES / COG;            // tile groups in flight per block
    static constexpr int IN_ROWS = TH + 2;
    static constexpr int ROWP = TW + 8;                  // staged row: x0-4 .. x0+TW+3 (16-byte aligned start)
    static constexpr int PPR = ROWP / 4;
    static constexpr int PLANE = IN_ROWS * ROWP;
    static constexpr int PIECES = C * IN_ROWS * PPR;
    static constexpr int NB = (PIECES + 63) / 64;
    static constexpr int NI = (NB + WAVES - 1) / WAVES;  // DMA wave-instructions per wave per tile
    static constexpr int STAGE = NI * WAVES * 256;       // floats per LDS stage
    static constexpr int NGX = TW / 32, NGY = TH / 2;    // tile groups (16 tiles = 32 x 2 pixels) per block tile
    static constexpr int NG = NGX * NGY;
    static constexpr int NGW = NG / SLOTS;               // tile groups per wave per block tile
    static constexpr int NW4 = 16 * KS / 4;              // float4 weight registers per lane
    static_assert(WAVES % COG == 0 && NG % SLOTS == 0, "wave roles");
    static_assert(TW % 32 == 0 && TH % 2 == 0, "block tile");
    static_assert(PLANE % 64 == 16 || PLANE % 64 == 48, "plane stride must put paired channel slots 32 banks apart");
    static_assert((4 * (KS - 1) + 3) * PLANE * 4 + 3 * ROWP * 4 + 32 < 65536, "ds_read immediate range");
};

// channel of k-step slot g (lane / 16): slots 0,1 (one half-wave) take channels 0 and 2 of the step
__device__ __forceinline__ int cperm(int g) { return ((g & 1) << 1) | (g >> 1); }

#ifdef EEM_STAMPS
// diagnostic build only: per-wave s_memtime stamps of the block's SECOND tile (steady state), written at the end
__device__ unsigned long long g_stamps16[2048 * 8 * 8];
#define STAMP16(i) if (it == 1) st[i] = __builtin_amdgcn_s_memtime()
#else
#define STAMP16(i)
#endif

template <int C, int TH, int TW, int WAVES, int POOLK>
__global__ __launch_bounds__(WAVES * 64) void wino_kernel(EncConvArgs a) {
    ENC_ARGS_NOW(a);
#ifdef EEM_STAMPS
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    st[6] = __builtin_amdgcn_s_memtime();
    st[7] = __builtin_amdgcn_s_memrealtime();
#endif
    using K = WCfg<C, TH, TW, WAVES>;
    constexpr int NWX = POOLK > 0 ? TW / POOLK : 1;                      // pooling windows per block tile (x)
    constexpr int RED = POOLK > 0 ? K::NGY * C * NWX : 0;                // pooling scratch (floats)
    constexpr int NSF = K::NGW * 8;                                      // feature-map stores per wave per tile
    constexpr int NS = NSF;                                              // younger than the next tile's DMA
    static_assert((2 * K::STAGE + 2 * RED) * 4 <= 160 * 1024, "LDS budget");
    static_assert(NS <= 63, "vmcnt immediate");
    static_assert(POOLK == 0 || (POOLK % TH == 0 && TW % POOLK == 0 && C * NWX <= WAVES * 64), "pool windows");
    __shared__ __attribute__((aligned(16))) float lds[2 * K::STAGE + 2 * RED];
    float* red0 = lds + 2 * K::STAGE;          // pooling scratch, double-buffered by tile parity

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: roles below are scalar
    const int j = lane & 15, g = lane >> 4;
    const int slot = wave / K::COG;
    const int cog = wave % K::COG;

    const TileRange tr_ = block_tile_range(a.tiles_x * a.tiles_y * a.nimg, blockIdx.x, gridDim.x);
    const int ntile = tr_.count;
    if (ntile == 0) return;
    TileCoord cur = tile_coord(tr_.first, a.tiles_x, a.tiles_y), nxt = cur, prv = cur;   // computed / next to request / previous
    const float* zero_page = a.zero_page;

    // ---- DMA plan: the piece a lane moves in wave-instruction k never changes; its offset from the tile's first
    // staged element is kept, and tiles that lie inside the image skip the per-piece bounds tests
    int poff[K::NI];
#pragma unroll
    for (int k = 0; k < K::NI; ++k) {
        int p = (wave + k * WAVES) * 64 + lane;
        p = p < K::PIECES ? p : K::PIECES - 1;                           // padding lanes re-copy the last piece
        const int c = p / (K::IN_ROWS * K::PPR);
        const int rem = p - c * (K::IN_ROWS * K::PPR);
        const int ry = rem / K::PPR;
        const int q = rem - ry * K::PPR;
        poff[k] = (c * a.hin + ry) * a.win + q * 4;
    }
    struct DmaTile { const float* src; float* sbase; int gy0, gxa; bool interior; };
    auto dma_prep = [&](int it, const TileCoord& tc) {
        DmaTile d;
        d.gy0 = tc.by * TH - 1; d.gxa = tc.bx * TW - 4;
        d.src = a.in0 + (size_t)tc.n * C * a.hin * a.win + (d.gy0 * a.win + d.gxa);
        d.sbase = lds + (it & 1) * K::STAGE;
        d.interior = d.gy0 >= 0 && d.gy0 + K::IN_ROWS <= a.hin && d.gxa >= 0 && d.gxa + K::ROWP <= a.win;
        return d;
    };
    auto dma_piece = [&](const DmaTile& d, int k) {
        const float* gp = d.src + poff[k];
        if (!d.interior) {
            int p = (wave + k * WAVES) * 64 + lane;
            p = p < K::PIECES ? p : K::PIECES - 1;
            const int rem = p % (K::IN_ROWS * K::PPR);
            const int ry = rem / K::PPR;
            const int q = rem - ry * K::PPR;
            const int gy = d.gy0 + ry, gx = d.gxa + q * 4;
            gp = (gy >= 0 && gy < a.hin && gx >= 0 && gx < a.win) ? gp : zero_page;
        }
        __builtin_amdgcn_global_load_lds(GLB_PTR(gp), LDS_PTR(d.sbase + (wave + k * WAVES) * 256), 16, 0, 0);
    };

    float biasv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) biasv[r] = a.bias[cog * 16 + g * 4 + r];
    // pooling finish (one value per thread of the first C * NWX threads): the lane's share of the address
    const bool pf_act = tid < C * NWX;
    const int pf_co = pf_act ? tid / NWX : 0, pf_wx = pf_act ? tid - pf_co * NWX : 0;
    const unsigned pf_off = (unsigned)pf_co * (unsigned)(a.tiles_y * a.tiles_x * NWX) + (unsigned)pf_wx;
    {
        const DmaTile d0 = dma_prep(0, nxt);
#pragma unroll
        for (int k = 0; k < K::NI; ++k) dma_piece(d0, k);
    }
    // stationary weights (k-step-major float4s); requested only after tile 0's input has landed - see conv_wino32.hip
    f32x4 wr[K::NW4];
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(a.wwino) + (size_t)cog * K::NW4 * 64 + lane;

    auto tile = [&](int it, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        STAMP16(1);
        // The next tile's DMA (its stage was last read by tile it-1: free since the barrier) is issued piecewise
        // inside the k-loop, behind each k-step's MFMAs: an LDS-DMA instruction costs the issuing wave 100-200
        // cycles, which then overlap MFMAs in flight instead of stalling every wave of the block at once.
        const bool have_next = it + 1 < ntile;
        DmaTile dnext = {};
        if (have_next) {
            tile_advance(nxt, a.tiles_x, a.tiles_y);
            dnext = dma_prep(it + 1, nxt);
        }
        if constexpr (POOLK > 0 && !FIRST) {
            // finish the previous tile's pooling partial sums (its barrier is the ring barrier just passed);
            // one store per lane (idle lanes into the scratch page)
            const float* redp = red0 + ((it - 1) & 1) * RED;
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < K::NGY; ++q) s += redp[(q * C + pf_co) * NWX + pf_wx];
            // scalar part of the address per tile, lane part (pf_off) fixed for the block's life
            float* pb = a.pool_partial + ((size_t)prv.n * C * a.tiles_y + prv.by) * (a.tiles_x * NWX) + prv.bx * NWX;
            float* p = pf_act ? pb + pf_off : a.trash + lane * 2;
            *p = s;
        }
        float* red = red0 + (it & 1) * RED;
        STAMP16(2);
        if constexpr (FIRST) {
#pragma unroll
            for (int q = 0; q < K::NW4; ++q) wr[q] = wsrc[q * 64];
        }
        const int bx = cur.bx, by = cur.by, n = cur.n;
        const float* tb = lds + (it & 1) * K::STAGE;
        const int hw = a.hout * a.wout;
        float* dst = a.out + (size_t)n * C * hw;
        const float* gsrc = a.gate ? a.gate + (size_t)n * C * hw : nullptr;

        // The tile's work comes in two forms: PIPE - the next tile lies inside the image, so its DMA pieces need no bounds test and are
        // issued branch-free behind the k-steps; and the border form - the next tile's pieces (with their tests and branches) are
        // requested up front and the k-loop carries none.  (The per-piece `have_next` / `interior` tests inside the k-loop cost a
        // vector compare, a select and a branch per k-step.)
        auto eloop = [&](auto pipe_tag) {
        constexpr bool PIPE = decltype(pipe_tag)::value;
#pragma unroll
        for (int e = 0; e < K::NGW; ++e) {
            const int ng = slot * K::NGW + e;
            const int tr = ng / K::NGX, xg = ng % K::NGX;
            // patch columns -1..4 as three aligned ds_read_b64 per row (conflict-free: the two channel slots of a
            // half-wave are two planes = 32 banks apart)
            const float* pl = tb + cperm(g) * K::PLANE + 2 * tr * K::ROWP + 2 * (xg * 16 + j) + 2;

            // bias rides in the accumulators: with A^T = [[1,1,1,0],[0,1,-1,-1]] a constant b at position (0,0), -b at (0,3) and (3,0)
            // and +b at (3,3) reaches each of the four outputs exactly once (saves the 16 bias adds of the output transform)
            f32x4 acc[16];
#pragma unroll
            for (int p = 0; p < 16; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[0][r] = biasv[r]; acc[3][r] = -biasv[r]; acc[12][r] = -biasv[r]; acc[15][r] = biasv[r]; }

            f32x2 dn[4][3];
            auto load_patch = [&](int s) {
                const int off = 4 * s * K::PLANE;
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int q = 0; q < 3; ++q) dn[r][q] = *reinterpret_cast<const f32x2*>(pl + off + r * K::ROWP + 2 * q);
            };
            load_patch(0);
#pragma unroll
            for (int s = 0; s < K::KS; ++s) {
                float d[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { d[r][0] = dn[r][0][1]; d[r][1] = dn[r][1][0]; d[r][2] = dn[r][1][1]; d[r][3] = dn[r][2][0]; }
                if (s + 1 < K::KS) load_patch(s + 1);
                __builtin_amdgcn_sched_barrier(0);   // keep the next k-step's LDS reads ahead of this k-step's work
                float t[4][4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    t[0][b] = d[0][b] - d[2][b];
                    t[1][b] = d[1][b] + d[2][b];
                    t[2][b] = d[2][b] - d[1][b];
                    t[3][b] = d[1][b] - d[3][b];
                }
                // all 16 B operands of the k-step first, then the 16 MFMAs back to back: computed one MFMA ahead, every MFMA waited
                // on its own operand's write (an s_nop in front of each: 48 per tile)
                float v[16];
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    v[x * 4 + 0] = t[x][0] - t[x][2]; v[x * 4 + 1] = t[x][1] + t[x][2];
                    v[x * 4 + 2] = t[x][2] - t[x][1]; v[x * 4 + 3] = t[x][1] - t[x][3];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    const int q = s * 16 + p;
                    acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[q >> 2][q & 3], v[p], acc[p], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (PIPE) {   // this k-step's share of the next tile's DMA
                    constexpr int STEPS = K::KS * K::NGW, PER = (K::NI + STEPS - 1) / STEPS;
#pragma unroll
                    for (int q = 0; q < PER; ++q)
                        if ((e * K::KS + s) * PER + q < K::NI) {
                            const int k = (e * K::KS + s) * PER + q;
                            __builtin_amdgcn_global_load_lds(GLB_PTR(dnext.src + poff[k]), LDS_PTR(dnext.sbase + (wave + k * WAVES) * 256), 16, 0, 0);
                        }
                }
            }

            STAMP16(3);
            // ---- output transform, bias, LeakyReLU, [gate], stores, pooling partial sums
            const int oy = by * TH + 2 * tr, ox = bx * TW + 2 * (xg * 16 + j);
            const bool in0 = oy < a.hout && ox < a.wout, in1 = oy + 1 < a.hout && ox < a.wout;
            const int o0 = ((cog * 16 + g * 4) * a.hout + oy) * a.wout + ox;
            const bool full = by * TH + TH <= a.hout && bx * TW + TW <= a.wout;       // wave-uniform
            const unsigned lane_bo = (unsigned)o0 * 4u;
            float psum[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u[4][2];
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    u[x][0] = acc[x * 4 + 0][r] + acc[x * 4 + 1][r] + acc[x * 4 + 2][r];
                    u[x][1] = acc[x * 4 + 1][r] - acc[x * 4 + 2][r] - acc[x * 4 + 3][r];
                }
                float y00 = u[0][0] + u[1][0] + u[2][0], y01 = u[0][1] + u[1][1] + u[2][1];      // (bias: in the accumulators)
                float y10 = u[1][0] - u[2][0] - u[3][0], y11 = u[1][1] - u[2][1] - u[3][1];
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
                if (full) {
                    // tile inside the image (every tile at sizes that are multiples of the block tile): the address is a scalar base
                    // per (cout register, row) + ONE 32-bit lane offset - no 64-bit vector arithmetic, no selects
                    char* rb = reinterpret_cast<char*>(dst) + (size_t)r * hw * 4;
                    *reinterpret_cast<f32x2*>(rb + lane_bo) = f32x2{y00, y01};
                    *reinterpret_cast<f32x2*>(rb + (size_t)a.wout * 4 + lane_bo) = f32x2{y10, y11};
                } else {
                    // every lane stores (outside lanes into a scratch page): exactly NSF stores per wave and tile
                    float* p0 = in0 ? dst + o : a.trash + lane * 2;
                    float* p1 = in1 ? dst + o + a.wout : a.trash + lane * 2;
                    *reinterpret_cast<f32x2*>(p0) = f32x2{y00, y01};
                    *reinterpret_cast<f32x2*>(p1) = f32x2{y10, y11};
                }
            }
            if constexpr (POOLK > 0) {
                constexpr int SW = POOLK / 2;                   // tiles (lanes) per pooling window
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sred = lane_group_sum<SW>(psum[r]);
                    if ((j & (SW - 1)) == 0)
                        red[(tr * C + cog * 16 + g * 4 + r) * NWX + xg * (32 / POOLK) + j / SW] = sred;
                }
            }
        }
        };
        if (have_next && dnext.interior) {
            eloop(std::true_type{});
        } else {
            if (have_next) {
#pragma unroll
                for (int k = 0; k < K::NI; ++k) dma_piece(dnext, k);
            }
            eloop(std::false_type{});
        }
        STAMP16(4);
        // the pooling partial sums in `red` are finished after the next ring barrier (top of the next tile / after
        // the loop): one barrier per tile instead of two
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // tile 0's input has landed
    __builtin_amdgcn_s_barrier();
    tile(0, std::true_type{});
    prv = cur;
    tile_advance(cur, a.tiles_x, a.tiles_y);
    // from here on the weights are plain register values for the compiler (no vmcnt bookkeeping in the loop)
#pragma unroll
    for (int q = 0; q < K::NW4; ++q) asm volatile("" : "+v"(wr[q]));
#pragma unroll 1
    for (int it = 1; it < ntile; ++it) {
        STAMP16(0);
#ifdef EEM_WAIT0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS) : "memory");           // all but the previous tile's stores
#endif
        if constexpr (POOLK > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // its pooling sums are in LDS
        __builtin_amdgcn_s_barrier();
        tile(it, std::false_type{});
        prv = cur;
        tile_advance(cur, a.tiles_x, a.tiles_y);
        STAMP16(5);
    }
    if constexpr (POOLK > 0) {                       // last tile's pooling partial sums
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const float* redp = red0 + ((ntile - 1) & 1) * RED;
        if (tid < C * NWX) {
            const int co = tid / NWX, wx = tid - co * NWX;
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < K::NGY; ++q) s += redp[(q * C + co) * NWX + wx];
            a.pool_partial[(((size_t)prv.n * C + co) * a.tiles_y + prv.by) * (a.tiles_x * NWX) + prv.bx * NWX + wx] = s;
        }
    }
#ifdef EEM_STAMPS
    st[6] = __builtin_amdgcn_s_memtime() - st[6];
    st[7] = __builtin_amdgcn_s_memrealtime() - st[7];
    if (lane == 0 && blockIdx.x < 2048) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int i = 0; i < 8; ++i) g_stamps16[(blockIdx.x * 8 + wave) * 8 + i] = st[i];
    }
#endif
}

// ---- weight transform U = G g G^T into the fragment order read above:
//   [cog][q4 = (s * 16 + p) / 4][lane][4],  lane = (cout % 16) + 16 * slot, slot holding cin 4s + cperm(slot)
__global__ void wino_wt_kernel(const float* __restrict__ w, int c, int transpose_flip, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= c * c) return;
    const int co = t / c, ci = t - co * c;
    float gk[3][3];
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx)
            gk[ky][kx] = transpose_flip ? w[((size_t)ci * c + co) * 9 + (2 - ky) * 3 + (2 - kx)]
                                        : w[((size_t)co * c + ci) * 9 + ky * 3 + kx];
    float m[4][3];                                   // G g
    for (int kx = 0; kx < 3; ++kx) {
        m[0][kx] = gk[0][kx];
        m[1][kx] = 0.5f * (gk[0][kx] + gk[1][kx] + gk[2][kx]);
        m[2][kx] = 0.5f * (gk[0][kx] - gk[1][kx] + gk[2][kx]);
        m[3][kx] = gk[2][kx];
    }
    const int ks = c / 4, nw4 = 16 * ks / 4;
    const int cog = co >> 4, i = co & 15, s = ci >> 2, kk = ci & 3;
    const int slot = ((kk & 1) << 1) | (kk >> 1);    // cperm is its own inverse
    const int lane = i + 16 * slot;
    for (int xi = 0; xi < 4; ++xi) {
        const float u[4] = {m[xi][0], 0.5f * (m[xi][0] + m[xi][1] + m[xi][2]), 0.5f * (m[xi][0] - m[xi][1] + m[xi][2]),
                            m[xi][2]};
        for (int nu = 0; nu < 4; ++nu) {
            const int q = s * 16 + xi * 4 + nu;
            out[(((size_t)cog * nw4 + (q >> 2)) * 64 + lane) * 4 + (q & 3)] = u[nu];
        }
    }
}

template <int C> struct WTile;
//                                           TH  TW WAVES POOLK
template <> struct WTile<16> { static constexpr int TH = 8, TW = 64, WAVES = 8, POOLK = 32; };

template <int C>
int wino_launch_c(const EncConvArgs& a0, hipStream_t stream) {
    using W = WTile<C>;
    EncConvArgs a = a0;
    a.tiles_x = ceil_div(a.wout, W::TW);
    a.tiles_y = ceil_div(a.hout, W::TH);
    const int T = a.tiles_x * a.tiles_y * a.nimg;
    int per_xcd = ceil_div(T, 8);
    static const int env_cap = enc_blocks_per_xcd("W16", 0);     // tuning override
    const int cap = env_cap > 0 ? env_cap : (a.blocks_per_xcd > 0 ? a.blocks_per_xcd : 32);   // default: one resident block per CU
    if (per_xcd > cap) per_xcd = cap;
    if (a.pool_partial != nullptr && a.pool_k != W::POOLK) {
        eem_set_error("wino: fused pooling with k=%d is not built for C=%d", a.pool_k, C);
        return EEM_ERR_ARG;
    }
    EEM_NOTE_GRID(per_xcd * 8, W::WAVES * 64);
    EEM_NOTE_PIPE(3);                                    // F(2x2,3x3): 16 products per 4 outputs against the direct form's 36
    if (a.pool_partial != nullptr)
        hipLaunchKernelGGL((wino_kernel<C, W::TH, W::TW, W::WAVES, W::POOLK>), dim3(per_xcd * 8), dim3(W::WAVES * 64), 0,
                           stream, a);
    else
        hipLaunchKernelGGL((wino_kernel<C, W::TH, W::TW, W::WAVES, 0>), dim3(per_xcd * 8), dim3(W::WAVES * 64), 0,
                           stream, a);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

}  // namespace

bool wino_supported(int cin, int cout, int stride, int win) {
    return stride == 1 && cin == cout && (cin == 16 || cin == 32 || cin == 64) && (win & 3) == 0;
}

size_t wino_packed_floats(int c) { return wino4_packed_floats(c); }      // 36 c^2 >= 16 c^2: either form fits

#ifdef EEM_STAMPS
extern "C" __attribute__((visibility("default"))) int eemflow_debug_read_stamps16(unsigned long long* dst, size_t n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps16), n * sizeof(unsigned long long)) == hipSuccess ? 0 : 2;
}
#endif

int wino_transform_launch(const float* w, int c, int transpose_flip, float* packed, hipStream_t stream, int f4) {
    EEM_REQUIRE(c == 16 || c == 32 || c == 64, "wino_transform_launch: C=%d", c);
    if (f4) return wino4_transform_launch(w, c, transpose_flip, packed, stream);
    if (c >= 32) return wino32_transform_launch(w, c, transpose_flip, packed, stream);
    hipLaunchKernelGGL(wino_wt_kernel, dim3(ceil_div(c * c, 256)), dim3(256), 0, stream, w, c, transpose_flip, packed);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

void wino_tile(int c, int f4, int* th, int* tw, int* poolk) {
    if (f4) { wino4_tile(c, th, tw, poolk); return; }
    if (c >= 32) { wino32_tile(c, th, tw, poolk); return; }
    *th = WTile<16>::TH; *tw = WTile<16>::TW; *poolk = WTile<16>::POOLK;
}

int wino_launch(int c, const EncConvArgs& a, hipStream_t stream) {
    EEM_REQUIRE(a.wwino && a.zero_page && a.trash, "wino_launch: NULL operand");
    if (a.wino_f4) return wino4_launch(c, a, stream);
    if (c >= 32) return wino32_launch(c, a, stream);
    if (c == 16) return wino_launch_c<16>(a, stream);
    eem_set_error("wino_launch: unsupported C=%d", c);
    return EEM_ERR_ARG;
}
