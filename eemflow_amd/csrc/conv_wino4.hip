// Winograd F(4x4,3x3) form of the encoder's stride-1 C -> C convolutions (pconv1_2, pconv2_2/2_3, pconv3_2/3_3:
// EEMFlow.py:76,78-79,81-82), fp32 throughout:
//
//     Y = A^T [ sum_cin (G g G^T) (.) (B^T d B) ] A          (4x4 outputs from a 6x6 input patch: 36 products per
//                                                             (cin, cout) and 16 outputs - 2.25 per output, where
//                                                             F(2x2,3x3) spends 4 and the direct form 9)
//
// Why this form: f32 MFMA and f32 VALU are one pipe on gfx950 (profiles/r02_mfma_valu_overlap.txt), so a kernel's time is
// MFMA cycles + VALU instructions x their issue cost.  F(4x4) issues 0.5625 of F(2x2)'s MFMAs; the transforms run on PACKED
// f32 pairs (v_pk_fma_f32 ...: the cost of a scalar instruction for two values, tools/micro/pk_valu.hip), and two waves per SIMD
// take turns at the pipe (a VALU instruction beside MFMAs costs a lone wave 7.4 cycles, two resident waves 3.5 each).
//
// One wave owns a GROUP of 16 tiles (8 x 2 tiles = 32 x 8 pixels), one 16-cout group and ALL 36 Winograd positions:
// v_mfma_f32_16x16x4_f32 per position p = (xi, nu) and k-step, D_p[16 cout][16 tiles] += U_p[16 cout][4 cin] V_p[4 cin][16 tiles];
// a 16x16 accumulator is 4 registers, so the 36 of them are 144 (AGPRs) - the output transform runs in registers with no
// exchange between waves - and the rest of the wave lives in 112 VGPRs: 8 waves per block, two per SIMD.
//   * input AND weights reach LDS by LDS-DMA (global_load_lds_dwordx4) in K-STEP slices - 4 channels x (TH + 2) rows x (TW + 8)
//     columns of the input, then U = G g G^T of those 4 channels for every cout group ([cog][q][lane][4], 9 KB per group) -
//     through a ring of R slices that runs ahead of the k-loop ACROSS tile boundaries (a block walks a contiguous tile range);
//   * V = B^T d B is computed by the lane that feeds it to the MFMA (tile = lane % 16, channel = lane / 16) from a
//     ds_read2_b32 + ds_read_b128 per patch row (columns -1 | 4 and 0..3: the pairs the packed column pass works on);
//   * the A operands (U) are ds_read_b128 of 4 positions at a time, next to their MFMAs;
//   * bias rides in the accumulator of position (1,1) (column 1 of A^T is all ones), the first k-step of a tile takes C = 0 as an
//     inline operand (no accumulator clears);
//   * output transform on pairs of cout registers, LeakyReLU, [gate], float4 NCHW stores (a wave writes 128-byte row segments),
//     stage-pooling partial sums (DPP lane sums + a small LDS table finished behind the next tile's first barrier).
#include <type_traits>

#include "common.h"

namespace {

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int C, int NGX, int NGY>
struct W4Cfg {
    static constexpr int WAVES = 8;
    static constexpr int COG = C / 16;                   // 16-cout groups: waves that share one group of 16 tiles
    static constexpr int NG = WAVES / COG;               // tile groups per block tile
    static constexpr int TW = NGX * 32, TH = NGY * 8;    // block tile (pixels)
    static constexpr int KS = C / 4;                     // k-steps (4 cin each)
    static constexpr int IN_ROWS = TH + 2;
    static constexpr int ROWP = TW + 8;                  // staged row: x0-4 .. x0+TW+3 (16-byte aligned start)
    static constexpr int PPR = ROWP / 4;
    static constexpr int PC = IN_ROWS * PPR;             // 16-byte pieces of one channel
    static constexpr int PLANE_P = (PC + 15) / 16 * 16;  // channel planes start on 256-byte boundaries: the four channel slots of a
    static constexpr int PLANE = PLANE_P * 4;            // ds_read_b128 lane group then cover the 64 banks exactly once
    static constexpr int INP = 4 * PLANE_P;              // pieces of a slice's input part
    static constexpr int UP = COG * 9 * 64;              // pieces of its weight part ([cog][q][lane] float4s)
    static constexpr int NI = (INP + UP + 511) / 512;    // DMA wave-instructions per wave and k-step
    static constexpr int STAGE = NI * 2048;              // floats per ring slot
    static constexpr int R = 3;                          // ring slots
    static_assert(NG * COG == WAVES && NGX * NGY == NG, "wave roles");
    static_assert((4 * ROWP) % 64 == 32, "the two tile rows of a group must sit 128 bytes apart modulo 256 (ds_read_b128 banks)");
    static_assert((5 * ROWP + 8) * 4 < 65536, "ds_read immediate range");
};

#include "wino4_xf.h"

#ifdef EEM_STAMPS
// diagnostic build only (-DEEM_STAMPS=<C>: the layers with that channel count): s_memtime stamps of every wave (slot 0: start,
// 1: prologue done, 2 + 2L: k-step L has passed its barrier, 3 + 2L: its MFMAs are issued, 60: after the last output phase,
// 61: s_memrealtime ticks of the whole kernel), written as they are taken
__device__ unsigned long long g_stamps4[1024 * 8 * 64];
#define STAMP4(i) do { if (C == EEM_STAMPS && lane == 0 && blockIdx.x < 1024 && (i) < 64) g_stamps4[(blockIdx.x * 8 + wave) * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP4(i)
#endif

// EPI: 0 plain, 1 the result is multiplied by LeakyReLU'(gate) (data gradients), 2 relu(residual + result) (E-RAFT's residual blocks),
// 3 pooling partial sums only, no feature-map stores (pconv3_3 in inference: f13 is read through its pooled map alone) -
// separate instantiations, so the forward kernels' register allocation (252 of 256 VGPRs, no scratch) does not carry them
template <int C, int NGX, int NGY, int POOLK, int EPI>
__global__ __launch_bounds__(512, 2) void wino4_kernel(EncConvArgs a) {
    ENC_ARGS_NOW(a);
    using K = W4Cfg<C, NGX, NGY>;
    constexpr int R = K::R, NI = K::NI, KS = K::KS;
    constexpr int NWX = POOLK > 0 ? K::TW / POOLK : 1;                   // pooling windows per block tile (x)
    constexpr int RED1 = POOLK > 0 ? NGY * C * NWX : 0;                  // one tile's pooling scratch (floats): [group row][cout][window]
    constexpr int NSTORE = EPI == 3 ? 0 : 16;                            // feature-map stores per wave and tile
    static_assert(EPI != 3 || POOLK > 0, "a store-free launch needs the pooling output");
    constexpr int NPOOL = POOLK > 0 ? 1 : 0;
    static_assert((R * K::STAGE + 2 * RED1) * 4 <= 160 * 1024, "LDS budget");
    static_assert(POOLK == 0 || (POOLK % 8 == 0 && K::TW % POOLK == 0 && C * NWX * NGY <= 512 && POOLK <= 32), "pool windows");
    static_assert(NI + NSTORE + NPOOL <= 63, "vmcnt immediate");
    __shared__ __attribute__((aligned(256))) float lds[R * K::STAGE + 2 * RED1];
    float* red0 = lds + R * K::STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int tx = j & 7, ty = j >> 3;
    const int cog = wave % K::COG, grp = wave / K::COG;
    const int gx = grp % NGX, gy = grp / NGX;

    STAMP4(0);
#ifdef EEM_STAMPS
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    TileRange tr_ = block_tile_range(a.tiles_x * a.tiles_y * a.nimg, blockIdx.x, gridDim.x);
    int walk_stride = 1;
    if (a.reverse == 3) {
        // INTERLEAVED walk (round 6): the XCD still owns a contiguous range of tiles, but its G resident blocks take tiles kb, kb + G, kb + 2 G ...
        // of it - at any moment the XCD works on G CONSECUTIVE tiles, so the tiles that share partial cache lines (x-neighbours) and halo
        // rows (y-neighbours: G = 32 tiles are six rows of five) are in flight together and meet in the XCD's L2, instead of a tile time
        // apart in one block (342 MB per frame of HBM-side traffic at ten frames per launch against 296 of inputs + outputs)
        const int T = a.tiles_x * a.tiles_y * a.nimg;
        const int cpx = (T + 7) >> 3, xcd = blockIdx.x & 7, kb = blockIdx.x >> 3, gb = gridDim.x >> 3;
        const int r0 = xcd * cpx, r1 = min(r0 + cpx, T);
        const int have = r1 - r0 - kb;
        tr_.first = r0 + kb;
        tr_.count = have > 0 ? (have + gb - 1) / gb : 0;
        walk_stride = gb;
    }
    const int ntile = tr_.count;
    if (ntile == 0) return;
    const int total = ntile * KS;                                        // k-steps this block walks
    const int T_all = a.tiles_x * a.tiles_y * a.nimg;
    // a.reverse: 0 rows of tiles front to back (x fastest), 1 the same walk back to front, 2 COLUMNS of tiles (y fastest): the block's
    // consecutive tiles then share halo ROWS and the tiles that share the partial cache lines at their left / right edges - two of the
    // three to six 128-byte lines a tile row touches - belong to neighbouring blocks of the XCD, which reach them at about the same time
    auto coord_of = [&](int lt) {
        if (a.reverse == 2) return TileCoord{(lt / a.tiles_y) % a.tiles_x, lt % a.tiles_y, lt / (a.tiles_x * a.tiles_y)};
        return tile_coord(a.reverse == 1 ? T_all - 1 - lt : lt, a.tiles_x, a.tiles_y);
    };
    TileCoord cur = coord_of(tr_.first), nxt = cur, prv = cur;   // computed / being requested / previous
    auto step_tile = [&](TileCoord& t) {
        if (a.reverse == 3) {
            t.bx += walk_stride;
            while (t.bx >= a.tiles_x) { t.bx -= a.tiles_x; if (++t.by == a.tiles_y) { t.by = 0; ++t.n; } }
        } else if (a.reverse == 2) {
            if (++t.by == a.tiles_y) { t.by = 0; if (++t.bx == a.tiles_x) { t.bx = 0; ++t.n; } }
        } else if (a.reverse == 1) tile_retreat(t, a.tiles_x, a.tiles_y);
        else tile_advance(t, a.tiles_x, a.tiles_y);
    };
    const char* zero_page = reinterpret_cast<const char*>(a.zero_page);
    const int plane_in = a.hin * a.win;

    // ---- DMA plan: slot p = (k * 8 + wave) * 64 + lane of a slice is always the same piece: p < INP - input channel p / PLANE_P
    // of the k-step, row ry, 16-byte column q (padding slots re-copy a valid piece); else weight piece p - INP of the k-step
    unsigned poff[NI];                                                   // bytes from the slice's input origin / weight origin
    int pryq[NI];                                                        // input: ry | q << 8 (| 1 << 16 padding); weights: 1 << 17
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int p = (k * 8 + wave) * 64 + lane;
        if (p < K::INP) {
            const int c = p / K::PLANE_P;
            int rem = p - c * K::PLANE_P;
            const bool pad = rem >= K::PC;
            rem = pad ? K::PC - 1 : rem;
            const int ry = rem / K::PPR, q = rem - ry * K::PPR;
            poff[k] = (unsigned)((c * a.hin + ry) * a.win + q * 4) * 4u;
            pryq[k] = ry | (q << 8) | (pad ? 1 << 16 : 0);
        } else {
            const int up = p - K::INP;
            poff[k] = (unsigned)(up < K::UP ? up : K::UP - 1) * 16u;
            pryq[k] = 1 << 17;
        }
    }
    // weights of k-step s: [s][cog][q][lane] float4s, contiguous per k-step
    const char* wbase = reinterpret_cast<const char*>(a.wwino);
    int dma_s = 0;                                                       // k-step (channel group) of the next slice to request
    int dma_slot = 0;
    unsigned okm = 0;                                                    // per piece k: inside the image (input pieces of tile `nxt`)
    bool dma_interior = false;
    const char* dsrc = nullptr;                                          // input origin of the next slice
    auto dma_tile = [&]() {                                              // per tile: origin, border tests
        const int gy0 = nxt.by * K::TH - 1, gxa = nxt.bx * K::TW - 4;
        dma_interior = gy0 >= 0 && gy0 + K::IN_ROWS <= a.hin && gxa >= 0 && gxa + K::ROWP <= a.win;
        dsrc = reinterpret_cast<const char*>(a.in0 + (size_t)nxt.n * C * plane_in) + ((long)gy0 * a.win + gxa) * 4;
        if (!dma_interior) {
            // rows [rymin, rymax) and 16-byte columns [qmin, qmax) of the slice lie inside the image; the rest is zero padding
            const int rymin = gy0 < 0 ? -gy0 : 0, rymax = min(K::IN_ROWS, a.hin - gy0);
            const int qmin = gxa < 0 ? (-gxa) >> 2 : 0, qmax = min(K::PPR, (a.win - gxa) >> 2);
            okm = 0;
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int ry = pryq[k] & 0xFF, q = (pryq[k] >> 8) & 0xFF;
                const bool ok = pryq[k] >= (1 << 17) || (pryq[k] < (1 << 16) && ry >= rymin && ry < rymax && q >= qmin && q < qmax);
                okm |= ok ? 1u << k : 0u;
            }
        }
    };
    auto dma_issue = [&]() {
        float* sbase = lds + dma_slot * K::STAGE;
#ifdef EEM_DIAG
        const char* usrc = wbase + (size_t)(a.nt_store & 2 ? 0 : dma_s) * (K::UP * 16);    // EEM_NT_STORE bit 1 (diag): every k-step reads slice 0's weights
#else
        const char* usrc = wbase + (size_t)dma_s * (K::UP * 16);
#endif
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const bool all_in = (k + 1) * 512 <= K::INP, all_u = k * 512 >= K::INP;          // compile-time per k
            const char* gp;
            if (all_in) gp = dsrc + poff[k];
            else if (all_u) gp = usrc + poff[k];
            else gp = (pryq[k] >= (1 << 17) ? usrc : dsrc) + poff[k];
            if (!all_u && !dma_interior) gp = (okm >> k) & 1 ? gp : zero_page;
            __builtin_amdgcn_global_load_lds(GLB_PTR(gp), LDS_PTR(sbase + (k * 8 + wave) * 256), 16, 0, 0);
        }
        dsrc += (size_t)plane_in * 16;                                   // the next four channels
        if (++dma_s == KS) { dma_s = 0; step_tile(nxt); dma_tile(); }
        if (++dma_slot == R) dma_slot = 0;
    };

    f32x4 biasq;
#pragma unroll
    for (int r = 0; r < 4; ++r) biasq[r] = a.bias[cog * 16 + g * 4 + r];

    // ---- patch reads: columns -1..4 of the lane's tile, 6 rows, channel slot g of the k-step; weight reads: [cog][q][lane]
    const int lbase = g * K::PLANE + (gy * 8 + ty * 4) * K::ROWP + (gx * 8 + tx) * 4 + 2;     // floats; +1: column -1, +2: column 0 (16-byte aligned)
    const int wlbase = K::INP * 4 + (cog * 9 * 64 + lane) * 4;

    f32x4 acc[36];
    int L = 0;                                                           // k-steps done by this block
    int slot = 0;                                                        // ring slot of k-step L

    // pooling finish: one value per thread of the first NGY * C * NWX threads; partial sums are laid out in rows of 8 pixels (a
    // tile group's height): [n][cout][prow = ceil(hout / 8)][tiles_x * NWX]
    const int prow = (a.hout + 7) >> 3;
    const bool pf_act = tid < NGY * C * NWX;
    const int pf_gy = pf_act ? tid / (C * NWX) : 0;
    const int pf_co = pf_act ? (tid - pf_gy * C * NWX) / NWX : 0, pf_wx = pf_act ? tid % NWX : 0;

    // One k-step.  S0: first k-step of a tile (accumulators start from the inline 0 / the bias).
    // WAITN: vmcnt that leaves exactly the operations younger than the DMA of k-step L in flight (see the table below the kernel).
    auto step = [&](auto s0_tag, auto waitn_tag) {
        constexpr bool S0 = decltype(s0_tag)::value;
        constexpr int WAITN = decltype(waitn_tag)::value;
        // (a) this k-step's slice has landed (this wave's pieces; the barrier covers the other waves' and frees slot L-1)
        if (L + 1 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (S0 && POOLK > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the output phase's pooling sums are in LDS
        __builtin_amdgcn_s_barrier();
        STAMP4(2 + 2 * L);
        if constexpr (S0 && POOLK > 0) {
            // (b) finish the previous tile's pooling partial sums (written before this barrier); every lane stores (idle ones into
            // the scratch page) so that the wave's store count is fixed
            if (L > 0) {
                const float* redp = red0 + (((L / KS) - 1) & 1) * RED1;
                const float s = redp[(pf_gy * C + pf_co) * NWX + pf_wx];
                float* pb = a.pool_partial + ((size_t)prv.n * C * prow + prv.by * NGY) * (a.tiles_x * NWX) + prv.bx * NWX;
                float* p = pf_act && prv.by * NGY + pf_gy < prow ? pb + ((size_t)pf_co * prow + pf_gy) * (a.tiles_x * NWX) + pf_wx : a.trash + lane * 4;
                *p = s;
            }
        }
        // (c) request k-step L+R-1 into the slot k-step L-1 has left
        if (L + R - 1 < total) dma_issue();

        // (d) the patch, then its column transform (rows -> xi) on the column pairs (-1,4) (0,1) (2,3) the reads deliver
        const float* sl = lds + slot * K::STAGE;
        f32x2 t05[6], t12[6], t34[6];                                    // [xi]: columns (0,5) (1,2) (3,4) of the patch
        {
            const float* p = sl + lbase;
            f32x2 pout[6];
            f32x4 pmid[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                pout[r] = f32x2{p[r * K::ROWP + 1], p[r * K::ROWP + 6]};
                pmid[r] = *reinterpret_cast<const f32x4*>(p + r * K::ROWP + 2);
            }
            f32x2 x[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) x[r] = pout[r];
            bt6(x, t05);
#pragma unroll
            for (int r = 0; r < 6; ++r) x[r] = f32x2{pmid[r][0], pmid[r][1]};
            bt6(x, t12);
#pragma unroll
            for (int r = 0; r < 6; ++r) x[r] = f32x2{pmid[r][2], pmid[r][3]};
            bt6(x, t34);
        }
        // (e) row transform (columns -> nu), then the 36 MFMAs back to back, A operands from LDS four positions at a time:
        // rows in blocks of RB - their transforms (independent chains: instruction-level parallelism for the lone issue slot), then
        // their MFMAs back to back with the A operands read two quads ahead
        const f32x4* wl = reinterpret_cast<const f32x4*>(sl + wlbase);
        constexpr int RB = 3;
        f32x4 wq0 = wl[0], wq1 = wl[64];
#pragma unroll
        for (int xb = 0; xb < 6; xb += RB) {
            float v[RB * 6];
#pragma unroll
            for (int xi = 0; xi < RB; ++xi) bt6_row(t05[xb + xi], t12[xb + xi], t34[xb + xi], v + xi * 6);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pp = 0; pp < RB * 6; ++pp) {
                const int p = xb * 6 + pp;
                if (pp > 0 && (p & 3) == 0) {
                    wq0 = wq1;
                    if ((p >> 2) + 1 < 9) wq1 = wl[((p >> 2) + 1) * 64];
                }
                if constexpr (S0) {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq0[p & 3], v[pp], p == 7 ? biasq : z, 0, 0, 0);
                } else {
                    acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq0[p & 3], v[pp], acc[p], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP4(3 + 2 * L);
        ++L;
        slot = slot + 1 == R ? 0 : slot + 1;
    };

    // ---- output phase of the tile in `cur`: A^T M A on pairs of cout registers, LeakyReLU, [gate], stores, pooling partial sums
    auto output = [&](int it) {
        const int hw = a.hout * a.wout;
        const int oy = cur.by * K::TH + gy * 8 + ty * 4, ox = cur.bx * K::TW + (gx * 8 + tx) * 4;
        const int co0 = cog * 16 + g * 4;
        float* dst = a.out + (size_t)cur.n * C * hw;
        const float* gsrc = EPI == 1 ? a.gate + (size_t)cur.n * C * hw : nullptr;
        const float* rsrc = EPI == 2 ? a.res + (size_t)cur.n * C * hw : nullptr;
        const bool full = cur.by * K::TH + K::TH <= a.hout && cur.bx * K::TW + K::TW <= a.wout;    // wave-uniform
        const bool inx = ox < a.wout;                                    // widths are multiples of 4: a tile row is in or out
        const unsigned lane_bo = (unsigned)((co0 * a.hout + oy) * a.wout + ox) * 4u;
        float psum[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {                                    // cout registers (2h, 2h+1) as one packed pair
            f32x2 u[6][4];
#pragma unroll
            for (int xi = 0; xi < 6; ++xi) {
                f32x2 m[6];
#pragma unroll
                for (int nu = 0; nu < 6; ++nu) m[nu] = f32x2{acc[xi * 6 + nu][2 * h], acc[xi * 6 + nu][2 * h + 1]};
                at6(m, u[xi]);
            }
            f32x2 y[4][4];                                               // [row][x]
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                f32x2 m[6], o[4];
#pragma unroll
                for (int xi = 0; xi < 6; ++xi) m[xi] = u[xi][x];
                at6(m, o);
#pragma unroll
                for (int yy = 0; yy < 4; ++yy) y[yy][x] = o[yy];
            }
            if (a.act) {
                const float slope = a.act == 2 ? 0.f : 0.1f;
#pragma unroll
                for (int yy = 0; yy < 4; ++yy)
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        const f32x2 sc = slope * y[yy][x];
                        y[yy][x] = f32x2{fmaxf(y[yy][x][0], sc[0]), fmaxf(y[yy][x][1], sc[1])};
                    }
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = 2 * h + e;
                char* rb = reinterpret_cast<char*>(dst) + (size_t)r * hw * 4;
                f32x4 o[4];
#pragma unroll
                for (int yy = 0; yy < 4; ++yy) o[yy] = f32x4{y[yy][0][e], y[yy][1][e], y[yy][2][e], y[yy][3][e]};
                // gate / residual: the four rows' loads unconditional (an out-of-image lane reads the plane's first pixels) and issued
                // together, the condition applied to the values - a load inside a lane-dependent branch is followed by the compiler's
                // s_waitcnt vmcnt(0): sixteen dependent round trips per tile before, each also waiting for the ring's DMA in flight
                if constexpr (EPI == 1) {
                    const char* gb = reinterpret_cast<const char*>(gsrc) + (size_t)r * hw * 4;
                    f32x4 gt[4];
#pragma unroll
                    for (int yy = 0; yy < 4; ++yy) {
                        const bool in = inx && oy + yy < a.hout;
                        gt[yy] = *reinterpret_cast<const f32x4*>(gb + (in ? lane_bo + (unsigned)yy * (unsigned)a.wout * 4u : 0u));
                    }
#pragma unroll
                    for (int yy = 0; yy < 4; ++yy)
#pragma unroll
                        for (int x = 0; x < 4; ++x) o[yy][x] *= gt[yy][x] > 0.f ? 1.f : 0.1f;
                }
                if constexpr (EPI == 2) {                               // relu(res + act(conv)); waited for here: the ring's counted waits only wait longer
                    const char* qb = reinterpret_cast<const char*>(rsrc) + (size_t)r * hw * 4;
                    f32x4 rv[4];
#pragma unroll
                    for (int yy = 0; yy < 4; ++yy) {
                        const bool in = inx && oy + yy < a.hout;
                        rv[yy] = *reinterpret_cast<const f32x4*>(qb + (in ? lane_bo + (unsigned)yy * (unsigned)a.wout * 4u : 0u));
                    }
#pragma unroll
                    for (int yy = 0; yy < 4; ++yy)
#pragma unroll
                        for (int x = 0; x < 4; ++x) o[yy][x] = fmaxf(o[yy][x] + rv[yy][x], 0.f);
                }
                float ps = 0.f;
#pragma unroll
                for (int yy = 0; yy < 4; ++yy) {
                    const bool in = inx && oy + yy < a.hout;
                    if constexpr (EPI == 3) {
                        // (no store)
                    } else if (full) {
                        if (a.nt_store) __builtin_nontemporal_store(o[yy], reinterpret_cast<f32x4*>(rb + lane_bo + (size_t)yy * a.wout * 4));
                        else *reinterpret_cast<f32x4*>(rb + lane_bo + (size_t)yy * a.wout * 4) = o[yy];
                    } else {
                        // every lane stores (outside lanes into a scratch page): exactly NSTORE stores per wave and tile
                        float* p = in ? reinterpret_cast<float*>(rb + lane_bo + (size_t)yy * a.wout * 4) : a.trash + lane * 4;
                        *reinterpret_cast<f32x4*>(p) = o[yy];
                    }
                    const float rs = (o[yy][0] + o[yy][1]) + (o[yy][2] + o[yy][3]);
                    ps += (full || in) ? rs : 0.f;
                }
                psum[r] = ps;
            }
        }
        if constexpr (POOLK > 0) {
            constexpr int SW = POOLK / 4;                                // tiles (lanes) per pooling window
            float* red = red0 + (it & 1) * RED1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sred = window_sum<SW>(psum[r]);
                if ((tx & (SW - 1)) == 0 && ty == 0) red[(gy * C + co0 + r) * NWX + (gx * 8 + tx) / SW] = sred;
            }
        }
    };

    // ---- prologue: R-1 slices in flight
    dma_tile();
#pragma unroll
    for (int q = 0; q < R - 1; ++q)
        if (q < total) dma_issue();
    STAMP4(1);

    // vmcnt table (operations younger than the DMA of k-step L when k-step L starts; R = 3, per wave): that DMA was issued in k-step
    // L-2 (or the prologue), k-step L-1 issued the next one (NI pieces)
    //   tile 0 (s = 0, 1) and every s >= 2:                                                                  -> NI
    //   later tiles, s = 0: + the NSTORE stores of the output phase before it                                -> NI + NSTORE
    //   later tiles, s = 1: + those stores + the pooling store of k-step (s = 0)                             -> NI + NSTORE + NPOOL
    // (the last k-step of a block, behind which nothing was requested, waits for everything)
    using T = std::true_type;
    using F = std::false_type;
    for (int it = 0; it < ntile; ++it) {
        if (it == 0) {
            step(T{}, std::integral_constant<int, NI>{});
            step(F{}, std::integral_constant<int, NI>{});
        } else {
            step(T{}, std::integral_constant<int, NI + NSTORE>{});
            step(F{}, std::integral_constant<int, NI + NSTORE + NPOOL>{});
        }
#pragma unroll 1
        for (int s = 2; s < KS; ++s) step(F{}, std::integral_constant<int, NI>{});
        output(it);
        prv = cur;
        step_tile(cur);
    }
    STAMP4(60);
#ifdef EEM_STAMPS
    if (C == EEM_STAMPS && lane == 0 && blockIdx.x < 1024) g_stamps4[(blockIdx.x * 8 + wave) * 64 + 61] = __builtin_amdgcn_s_memrealtime() - rt0;
#endif
    if constexpr (POOLK > 0) {                       // last tile's pooling partial sums
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const float* redp = red0 + ((ntile - 1) & 1) * RED1;
        if (pf_act && prv.by * NGY + pf_gy < prow)
            a.pool_partial[(((size_t)prv.n * C + pf_co) * prow + prv.by * NGY + pf_gy) * (a.tiles_x * NWX) + prv.bx * NWX + pf_wx] =
                redp[(pf_gy * C + pf_co) * NWX + pf_wx];
    }
}

// ---- weight transform U = G g G^T (6x6 per (cout, cin)) into the fragment order read above:
//   [s = cin / 4][cog = cout / 16][q = p / 4][lane = (cout % 16) + 16 * (cin % 4)][e = p % 4],  p = xi * 6 + nu
//   (a k-step's weights for every cout group are one contiguous run: the weight part of that k-step's DMA slice)
// (up to W4_WT_JOBS transforms per launch, job = blockIdx.y: a training step refreshes two forms of five layers after every weight
// update, ten 7-us launches in front of its forward before)
struct W4WtJobs { const float* w[W4_WT_JOBS]; float* out[W4_WT_JOBS]; int c[W4_WT_JOBS]; int flip[W4_WT_JOBS]; };
__global__ void wino4_wt_kernel(W4WtJobs jobs) {
    const float* __restrict__ w = jobs.w[blockIdx.y];
    float* __restrict__ out = jobs.out[blockIdx.y];
    const int c = jobs.c[blockIdx.y], transpose_flip = jobs.flip[blockIdx.y];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= c * c) return;
    const int co = t / c, ci = t - co * c;
    float gk[3][3];
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx)
            gk[ky][kx] = transpose_flip ? w[((size_t)ci * c + co) * 9 + (2 - ky) * 3 + (2 - kx)]
                                        : w[((size_t)co * c + ci) * 9 + ky * 3 + kx];
    const float G[6][3] = {{0.25f, 0.f, 0.f},
                           {-1.f / 6.f, -1.f / 6.f, -1.f / 6.f},
                           {-1.f / 6.f, 1.f / 6.f, -1.f / 6.f},
                           {1.f / 24.f, 1.f / 12.f, 1.f / 6.f},
                           {1.f / 24.f, -1.f / 12.f, 1.f / 6.f},
                           {0.f, 0.f, 1.f}};
    double m[6][3];                                  // G g (the transform runs once per weight update: double keeps U at fp32 round-off)
    for (int xi = 0; xi < 6; ++xi)
        for (int kx = 0; kx < 3; ++kx)
            m[xi][kx] = (double)G[xi][0] * gk[0][kx] + (double)G[xi][1] * gk[1][kx] + (double)G[xi][2] * gk[2][kx];
    const int ncog = c / 16;
    const int cog = co >> 4, s = ci >> 2, lane = (co & 15) + 16 * (ci & 3);
    for (int xi = 0; xi < 6; ++xi)
        for (int nu = 0; nu < 6; ++nu) {
            const double u = m[xi][0] * G[nu][0] + m[xi][1] * G[nu][1] + m[xi][2] * G[nu][2];
            const int p = xi * 6 + nu;
            out[((((size_t)s * ncog + cog) * 9 + (p >> 2)) * 64 + lane) * 4 + (p & 3)] = (float)u;
        }
}

template <int C> struct W4Tile;
//                                             NGX NGY POOLK
template <> struct W4Tile<16> { static constexpr int NGX = 4, NGY = 2, POOLK = 32; };      // 8 groups x 1 cout group: 128 x 16 pixels
template <> struct W4Tile<32> { static constexpr int NGX = 2, NGY = 2, POOLK = 16; };      // 4 groups x 2:  64 x 16
template <> struct W4Tile<64> { static constexpr int NGX = 1, NGY = 2, POOLK = 8; };       // 2 groups x 4:  32 x 16

template <int C>
int launch4_c(const EncConvArgs& a0, hipStream_t stream) {
    using W = W4Tile<C>;
    using K = W4Cfg<C, W::NGX, W::NGY>;
    EncConvArgs a = a0;
    a.tiles_x = ceil_div(a.wout, K::TW);
    a.tiles_y = ceil_div(a.hout, K::TH);
    const int T = a.tiles_x * a.tiles_y * a.nimg;
    int per_xcd = ceil_div(T, 8);
    static const int env_cap = enc_blocks_per_xcd(C == 16 ? "F16" : (C == 32 ? "F32" : "F64"), 0);     // tuning override
    // one resident block per CU at most; the frames-in-flight grid hint (EncConvArgs::blocks_per_xcd) is not taken as a number: these
    // tiles are 4x the F(2x2) kernels' (240 / 120 / 60 per frame at 1280x720), an arbitrary smaller grid would only unbalance them
    int cap = env_cap > 0 ? env_cap : 32;
    // several frames in flight (the hint is set): pconv1_2's 240 tiles go two to a block - one prologue per two tiles (+1.4 % frames/s)
    if (env_cap <= 0 && a.blocks_per_xcd > 0 && C == 16 && per_xcd > 16) cap = (per_xcd + 1) / 2;
    if (per_xcd > cap) per_xcd = cap;
    if (a.pool_partial != nullptr && a.pool_k != W::POOLK) {
        eem_set_error("wino4: fused pooling with k=%d is not built for C=%d", a.pool_k, C);
        return EEM_ERR_ARG;
    }
    EEM_NOTE_GRID(per_xcd * 8, 512);
    EEM_NOTE_PIPE(2);                                    // F(4x4,3x3) on the fp32 MFMA: a quarter of the direct form's multiplies
    if ((a.gate != nullptr || a.res != nullptr) && (a.pool_partial != nullptr || (a.gate != nullptr && a.res != nullptr))) {
        eem_set_error("wino4: gate / residual epilogues come without pooling and one at a time");
        return EEM_ERR_ARG;
    }
    if (a.pool_partial != nullptr && a.no_store && C == 64)
        hipLaunchKernelGGL((wino4_kernel<C, W::NGX, W::NGY, W::POOLK, (C == 64 ? 3 : 0)>), dim3(per_xcd * 8), dim3(512), 0, stream, a);
    else if (a.pool_partial != nullptr)
        hipLaunchKernelGGL((wino4_kernel<C, W::NGX, W::NGY, W::POOLK, 0>), dim3(per_xcd * 8), dim3(512), 0, stream, a);
    else if (a.gate != nullptr)
        hipLaunchKernelGGL((wino4_kernel<C, W::NGX, W::NGY, 0, 1>), dim3(per_xcd * 8), dim3(512), 0, stream, a);
    else if (a.res != nullptr)
        hipLaunchKernelGGL((wino4_kernel<C, W::NGX, W::NGY, 0, 2>), dim3(per_xcd * 8), dim3(512), 0, stream, a);
    else
        hipLaunchKernelGGL((wino4_kernel<C, W::NGX, W::NGY, 0, 0>), dim3(per_xcd * 8), dim3(512), 0, stream, a);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

}  // namespace

#ifdef EEM_STAMPS
extern "C" __attribute__((visibility("default"))) int eemflow_debug_read_stamps4(unsigned long long* dst, size_t n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps4), n * sizeof(unsigned long long)) == hipSuccess ? 0 : 2;
}
extern "C" __attribute__((visibility("default"))) int eemflow_debug_clear_stamps4() {
    static unsigned long long zero[1024 * 8 * 64];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps4), zero, sizeof zero) == hipSuccess ? 0 : 2;
}
#endif

size_t wino4_packed_floats(int c) { return (size_t)36 * c * c; }

int wino4_transform_multi_launch(const float* const* w, const int* c, const int* transpose_flip, float* const* packed, int njobs,
                                 hipStream_t stream) {
    EEM_REQUIRE(njobs >= 1 && njobs <= W4_WT_JOBS, "wino4_transform_multi_launch: njobs=%d", njobs);
    W4WtJobs J;
    int cmax = 0;
    for (int i = 0; i < W4_WT_JOBS; ++i) {
        const int k = i < njobs ? i : 0;
        EEM_REQUIRE(c[k] == 16 || c[k] == 32 || c[k] == 64, "wino4_transform_multi_launch: C=%d", c[k]);
        J.w[i] = w[k]; J.out[i] = packed[k]; J.c[i] = c[k]; J.flip[i] = transpose_flip[k];
        if (c[k] > cmax) cmax = c[k];
    }
    hipLaunchKernelGGL(wino4_wt_kernel, dim3(ceil_div(cmax * cmax, 256), njobs), dim3(256), 0, stream, J);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int wino4_transform_launch(const float* w, int c, int transpose_flip, float* packed, hipStream_t stream) {
    return wino4_transform_multi_launch(&w, &c, &transpose_flip, &packed, 1, stream);
}

// what the pooling protocol needs to know: partial sums come in rows of `th` = 8 pixels (a tile group's height, whatever the block
// tile's) and `tw`-wide block columns
void wino4_tile(int c, int* th, int* tw, int* poolk) {
    *th = 8;
    if (c == 16) { *tw = W4Cfg<16, W4Tile<16>::NGX, W4Tile<16>::NGY>::TW; *poolk = W4Tile<16>::POOLK; }
    else if (c == 32) { *tw = W4Cfg<32, W4Tile<32>::NGX, W4Tile<32>::NGY>::TW; *poolk = W4Tile<32>::POOLK; }
    else { *tw = W4Cfg<64, W4Tile<64>::NGX, W4Tile<64>::NGY>::TW; *poolk = W4Tile<64>::POOLK; }
}

int wino4_launch(int c, const EncConvArgs& a, hipStream_t stream) {
    EEM_REQUIRE(a.wwino && a.zero_page && a.trash, "wino4_launch: NULL operand");
    if (c == 16) return launch4_c<16>(a, stream);
    if (c == 32) return launch4_c<32>(a, stream);
    if (c == 64) return launch4_c<64>(a, stream);
    eem_set_error("wino4_launch: unsupported C=%d", c);
    return EEM_ERR_ARG;
}
