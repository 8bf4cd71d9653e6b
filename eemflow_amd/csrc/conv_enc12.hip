// pconv1_1 computed INSIDE pconv1_2's block (EEMFlow.py:75-76,135-136: 3x3 stride-2 conv 5 -> 16 + LeakyReLU on the replicate-padded event
// volumes, then 3x3 conv 16 -> 16 + LeakyReLU): the intermediate map `a1` (31.5 MB per frame pair at 1280x720, written once and read back
// 1.1x by the layer-by-layer schedule) never becomes a tensor in HBM.
//
// A persistent block of 8 waves owns one F(4x4,3x3) tile of pconv1_2 at a time - 128 x 16 output pixels, all 16 couts, the block tile of
// conv_wino4.hip at C = 16 - and alternates two phases on it:
//   A. the haloed a1 tile the Winograd phase reads (18 rows x 136 columns x 16 channels) is computed from the event volumes in BANDS of two
//      a1 rows: the five source rows x five bins a band needs arrive by 16-byte LDS-DMA (row clamp = replicate pad, zero page = conv pad;
//      four band buffers, three bands in flight ahead of the one being multiplied), K = 45 -> 48 on v_mfma_f32_16x16x4_f32 with pixels on M
//      (conv_enc1.hip's arithmetic: same operand order, same k order, bit-identical a1), bias as the accumulator's initial value, LeakyReLU,
//      zeros outside the image (pconv1_2's own padding).  The tile goes to a block-private scratch in the layout the Winograd ring's slots
//      have - four k-step slices of 4 channels x 18 rows x 136 columns - 156 KB per block that are rewritten for every tile and read back
//      within microseconds: L2 traffic, not HBM traffic (the whole grid's scratch is 256 x 156 KB = 40 MB; what matters is that a line is
//      re-read before the streaming source / f11 lines push it out of its XCD's 4 MB);
//   B. conv_wino4.hip's k-step loop, unchanged in its arithmetic: slices of a1 (from the scratch) and of U = G g G^T stream through the
//      three-slot LDS ring by LDS-DMA, V = B^T d B in the consuming lane, 36 MFMAs per k-step and wave, output transform, LeakyReLU, float4
//      stores of f11, 32 x 32 stage-pooling partial sums.
// The two phases share the block's LDS (A's band buffers lie over B's ring), so they do not overlap inside a block; a CU's MFMA pipe is
// busy in both (A: 162 units x 12 MFMAs, B: 8 waves x 4 k-steps x 36).
#include <type_traits>

#include "common.h"

namespace {

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

#include "wino4_xf.h"

// ---- phase B: conv_wino4.hip's W4Cfg<16, 4, 2>
constexpr int C = 16, NGX = 4, NGY = 2, KS = 4;
constexpr int TW = 128, TH = 16, IN_ROWS = TH + 2, ROWP = TW + 8, PPR = ROWP / 4;
constexpr int PC = IN_ROWS * PPR, PLANE_P = (PC + 15) / 16 * 16, PLANE = PLANE_P * 4;
constexpr int INP = 4 * PLANE_P, UP = 9 * 64, NI = (INP + UP + 511) / 512, STAGE = NI * 2048, R = 3;
constexpr int SLICE = INP * 4;                           // floats of one k-step's a1 slice in slot format
constexpr int POOLK = 32, NWX = TW / POOLK, RED1 = NGY * C * NWX;
// ---- phase A: bands of two a1 rows
constexpr int CIN = 5;
constexpr int SROW = 2 * ROWP + 4;                       // staged source row: padded columns 2 x0 - 12 .. 2 x0 + 263 (16-byte aligned start)
constexpr int SPPR = SROW / 4;
constexpr int BROWS = 5;                                 // source rows of a band: 2 (2 b + rr) + ky, rr in {0, 1}, ky in {0, 1, 2}
constexpr int BPL = BROWS * SROW;                        // one bin's plane in a band buffer
constexpr int BPIECES = CIN * BROWS * SPPR;
constexpr int NIB = (BPIECES + 511) / 512;
constexpr int BSTAGE = NIB * 2048;                       // floats per band buffer
constexpr int NBUF = 4, NBAND = IN_ROWS / 2, UPR = 9;    // 9 units of 16 columns per a1 row: 0, 16, ..., 112, 120 (the last overlaps)
static_assert(NBUF * BSTAGE <= R * STAGE, "the band buffers lie over the Winograd ring");
static_assert((R * STAGE + 2 * RED1 + 2 * C + 3 * 256 + 512 * NIB) * 4 <= 160 * 1024, "LDS budget");
static_assert(SPPR == 69 && BPIECES == 1725 && NIB == 4 && NI == 6, "piece plans below");

}  // namespace

size_t enc12_scratch_floats(int blocks) { return (size_t)blocks * KS * SLICE; }

namespace {

__global__ __launch_bounds__(512, 2) void enc12_kernel(Enc12Args a) {
    __shared__ __attribute__((aligned(256))) float lds[R * STAGE + 2 * RED1 + 2 * C + 3 * 256 + 512 * NIB];
    float* red0 = lds + R * STAGE;
    float* bias2_s = red0 + 2 * RED1;                    // pconv1_2's bias: read at a tile's first k-step, not held in registers
    float* bias1_s = bias2_s + C;                        // pconv1_1's bias and its packed weights ([q][lane] float4s): phase A's stationary
    float* wr_s = bias1_s + C;                           // operands come back from LDS every tile (no VMEM wait in front of the bands)
    int* plan_s = reinterpret_cast<int*>(wr_s + 3 * 256); // band DMA plan [thread][k]: bin << 16 | band row << 8 | 16-byte column of piece
                                                         // (k * 8 + wave) * 64 + lane - two divisions per piece, made once per block

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gx = wave % NGX, gy = wave / NGX;

    const TileRange tr_ = block_tile_range(a.tiles_x * a.tiles_y * a.nimg, blockIdx.x, gridDim.x);
    const int ntile = tr_.count;
    if (ntile == 0) return;
    // tiles in COLUMN order (by fastest, then bx, then image): a block walks down a strip, so the five source rows a tile shares with the
    // one below it were fetched by the same CU microseconds before (L2 hits instead of the row order's second HBM read)
    auto coord_of = [&](int lt) {
        if (a.row_order) return tile_coord(lt, a.tiles_x, a.tiles_y);
        return TileCoord{(lt / a.tiles_y) % a.tiles_x, lt % a.tiles_y, lt / (a.tiles_x * a.tiles_y)};
    };
    auto advance = [&](TileCoord& t) {
        if (a.row_order) { tile_advance(t, a.tiles_x, a.tiles_y); return; }
        if (++t.by == a.tiles_y) {
            t.by = 0;
            if (++t.bx == a.tiles_x) { t.bx = 0; ++t.n; }
        }
    };
    TileCoord cur = coord_of(tr_.first), prv = cur, nxt = cur;
    bool have_next = false;
    const float* zero_page = a.zero_page;
    float* scr = a.scratch + (size_t)blockIdx.x * (KS * SLICE);
    const char* wbase = reinterpret_cast<const char*>(a.u2);
    const float* in0 = (a.io && !a.io_frames) ? (const float*)a.io[0] : a.in0;
    const float* in1 = (a.io && !a.io_frames) ? (const float*)a.io[1] : a.in1;

    // pooling finish of a tile: one value per thread of the first NGY * C * NWX threads (partial sums in rows of 8 pixels:
    // [n][cout][prow = ceil(h1 / 8)][tiles_x * NWX])
    auto pool_finish = [&](int it_done, const TileCoord& t) {
        int tq = tid;
        asm volatile("" : "+v"(tq));                     // (made here, once per tile: nothing of it stays live across the k-loop)
        const int prow = (a.h1 + 7) >> 3;
        const bool pf_act = tq < NGY * C * NWX;
        const int pf_gy = pf_act ? tq / (C * NWX) : 0;
        const int pf_co = pf_act ? (tq - pf_gy * C * NWX) / NWX : 0, pf_wx = pf_act ? tq % NWX : 0;
        const float* redp = red0 + (it_done & 1) * RED1;
        if (pf_act && t.by * NGY + pf_gy < prow)
            a.pool_partial[(((size_t)t.n * C + pf_co) * prow + t.by * NGY + pf_gy) * (a.tiles_x * NWX) + t.bx * NWX + pf_wx] =
                redp[(pf_gy * C + pf_co) * NWX + pf_wx];
    };

    // ------------------------------------------------------------------------------------------------ phase A
    // band b of tile t (two a1 rows: source rows 2 y0 + 4 b - 3 .. + 1, five bins, padded columns 2 x0 - 12 .. 2 x0 + 263) into band
    // buffer (b + 2) % 4: bands 0 and 1 land in the two buffers that lie over ring slots 1 and 2, which the Winograd phase's LAST k-step
    // (slice 3, slot 0) no longer reads - they are requested there, a whole k-step and the output phase ahead of their use
    // the image a tile reads: found ONCE per tile (with per-frame buffers it is a load from the io table, and a load inside the band
    // loop drags a full s_waitcnt vmcnt(0) in front of every band - the compiler's wait for it - which is the end of any prefetch)
    auto frame_src = [&](const TileCoord& t) -> const float* {
        const int n = __builtin_amdgcn_readfirstlane(t.n);
        const float* p;
        if (a.io_frames) p = (const float*)(n < a.nimg0 ? a.io[3 * n] : a.io[3 * (n - a.nimg0) + 1]);
        else p = n < a.nimg0 ? in0 + (size_t)n * CIN * a.hraw * a.wraw : in1 + (size_t)(n - a.nimg0) * CIN * a.hraw * a.wraw;
        const uintptr_t v = (uintptr_t)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const float*)(((uintptr_t)hi << 32) | lo);
    };
    const float* src_cur = frame_src(cur);
    const float* src_nxt = src_cur;
    auto band_issue = [&](const float* src, const TileCoord& t, int b) {
        int lq = lane;
        asm volatile("" : "+v"(lq));
        const int x0 = t.bx * TW, y0 = t.by * TH;
        float* sb = lds + ((b + 2) & (NBUF - 1)) * BSTAGE;
        const u32x4 plan = *reinterpret_cast<const u32x4*>(plan_s + (wave * 64 + lq) * NIB);
#pragma unroll
        for (int k = 0; k < NIB; ++k) {
            const int c = plan[k] >> 16, r5 = (plan[k] >> 8) & 0xFF, q = plan[k] & 0xFF;
            const int sy_p = 2 * y0 + 4 * b - 3 + r5, sx_p = 2 * x0 - 12 + 4 * q;          // padded-image coordinates
            const bool ok = sy_p >= 0 && sy_p < a.hin && sx_p >= 0 && sx_p < a.win;        // else: pconv1_1's zero padding
            const int sy = min(max(sy_p - a.pad_top, 0), a.hraw - 1);                      // replicate rows of the pad band
            const unsigned off = (unsigned)((c * a.hraw + sy) * a.wraw + sx_p);            // < 2^31 floats per frame
            const float* gp = ok ? src + off : zero_page;
            __builtin_amdgcn_global_load_lds(GLB_PTR(gp), LDS_PTR(sb + (k * 8 + wave) * 256), 16, 0, 0);
        }
    };

    // NU = units of 16 pixels x 16 couts this wave multiplies per band (18 per band over 8 waves: waves 0, 1 take three).
    // FIRST: the block's first tile (its bands 0, 1 were requested by the prologue, nothing else is in flight); later tiles find bands
    // 0, 1 requested in front of the previous tile's NOUT output stores
    auto phase_a = [&](auto nu_tag) {
        constexpr int NU = decltype(nu_tag)::value;
        constexpr int NOUT = 16;
        int ln = lane;
        asm volatile("" : "+v"(ln));                     // per-tile constants stay inside the phase (phase B needs every register)
        const int jj = ln & 15, gg = ln >> 4;
        const int x0 = cur.bx * TW, y0 = cur.by * TH;
        int koff[12];
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            int k = s * 4 + gg;
            k = k < 45 ? k : 0;
            const int c = k / 9, t = k - c * 9;
            koff[s] = c * BPL + (t / 3) * SROW + (t % 3);
        }
        int ubase[NU], soff[NU], urr[NU], ucol[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int id = wave + 8 * u;                 // 0 .. 17
            urr[u] = id / UPR;
            const int cu = id - urr[u] * UPR;
            ucol[u] = cu * 16 < ROWP - 16 ? cu * 16 : ROWP - 16;
            ubase[u] = urr[u] * 2 * SROW + 2 * (ucol[u] + jj) + 3;
            soff[u] = (jj >> 2) * SLICE + (jj & 3) * PLANE + urr[u] * ROWP + ucol[u] + 4 * gg;
        }
        // the previous tile's Winograd phase has read its last slot (and written its pooling sums) before band 2 lands in buffer 0
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        band_issue(src_cur, cur, 2);
        f32x4 wr[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) wr[q] = (reinterpret_cast<const f32x4*>(wr_s) + ln)[q * 64];
        const float b1 = bias1_s[jj];
        static_for<0, NBAND>([&](auto bt) {
            constexpr int B = decltype(bt)::value;
            // band B has landed.  VMEM operations younger than its DMA that may stay in flight (vmcnt retires in order), per wave -
            // the request order is  D0 D1 | NOUT output stores | D2 | D3 S0 | D4 S1 | ... (Sb: the NU scratch stores of band b):
            //   B = 0: D1, the output stores, D2;  B = 1: the stores, D2, D3, S0;  B = 2: D3 S0 D4 S1;
            //   B >= 3: S(B-3) D(B+1) S(B-2) D(B+2) S(B-1), as far as those bands exist
            // (a block's first tile has no output stores in front: its prologue waits for everything once instead)
            constexpr int AHEAD = B + 2 < NBAND ? 2 : (B + 1 < NBAND ? 1 : 0);
            constexpr int YOUNGER = B == 0 ? 2 * NIB + NOUT : (B == 1 ? 2 * NIB + NOUT + NU : (B == 2 ? 2 * NIB + 2 * NU : 3 * NU + AHEAD * NIB));
            wait_vmcnt<YOUNGER>();
            __builtin_amdgcn_s_barrier();                 // everyone's pieces of band B; everyone is done with band B - 1's buffer
            if constexpr (B + 3 < NBAND) band_issue(src_cur, cur, B + 3);
            const float* tb = lds + ((B + 2) & (NBUF - 1)) * BSTAGE;
            f32x4 acc[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) acc[u] = f32x4{b1, b1, b1, b1};
#ifdef EEM_DIAG
            if (a.dbg & 16) __builtin_amdgcn_s_sleep(36);     // ~1 us of nothing in the multiplies' place: does the DMA run underneath?
            if (!(a.dbg & 8))
#endif
            {
                // all operands of the band first, then the MFMAs back to back (left alone, the scheduler pairs every ds_read with its
                // MFMA and waits for it: 36 LDS round trips per band and wave)
                float av[12][NU];
#pragma unroll
                for (int s = 0; s < 12; ++s)
#pragma unroll
                    for (int u = 0; u < NU; ++u) av[s][u] = tb[ubase[u] + koff[s]];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 12; ++s)
#pragma unroll
                    for (int u = 0; u < NU; ++u)
                        acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][u], wr[s >> 2][s & 3], acc[u], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int ay = y0 - 1 + 2 * B + urr[u], ax = x0 - 4 + ucol[u] + 4 * gg;    // a1 coordinates (widths are multiples of 4)
                const bool in = ay >= 0 && ay < a.h1 && ax >= 0 && ax < a.w1;              // outside: pconv1_2's zero padding
                f32x4 v = acc[u];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = in ? fmaxf(v[r], 0.1f * v[r]) : 0.f;
#ifdef EEM_DIAG
                if (a.dbg & 4) *reinterpret_cast<f32x4*>(a.trash + ln * 4) = v; else
#endif
                *reinterpret_cast<f32x4*>(scr + soff[u] + 2 * B * ROWP) = v;
            }
        });
        // the tile is in the scratch: stores acknowledged, every wave's.  Writer and readers are waves of ONE workgroup on one CU, whose
        // vector L1 is write-through and kept coherent with the stores that pass through it: workgroup-scope ordering (wait + barrier) is
        // all the memory model asks for - an agent-scope `buffer_inv sc1` here also drops the XCD's non-coherent L2 lines, 256 blocks x
        // every tile: measured 2.3x on the whole launch
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    // ------------------------------------------------------------------------------------------------ phase B
    if (tid < C) { bias2_s[tid] = a.bias2[tid]; bias1_s[tid] = a.bias1[tid]; }      // visible after phase A's first barrier
    if (tid < 3 * 64) reinterpret_cast<f32x4*>(wr_s)[tid] = reinterpret_cast<const f32x4*>(a.wpk1)[tid];
#pragma unroll
    for (int k = 0; k < NIB; ++k) {
        int p = (k * 8 + wave) * 64 + lane;
        p = p < BPIECES ? p : BPIECES - 1;
        const int c = p / (BROWS * SPPR);
        const int rem = p - c * (BROWS * SPPR);
        const int r5 = rem / SPPR;
        plan_s[tid * NIB + k] = (c << 16) | (r5 << 8) | (rem - r5 * SPPR);     // read back by the same thread only
    }
    band_issue(src_cur, cur, 0);
    band_issue(src_cur, cur, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the first tile's counted waits assume a previous tile's output stores in flight)
    f32x4 acc[36];
    int L = 0, slot = 0, dma_s = 0, dma_slot = 0;

    auto dma_issue = [&]() {                             // slice dma_s of the tile in the scratch + its U, into ring slot dma_slot
        float* sbase = lds + dma_slot * STAGE;
        const char* ssrc = reinterpret_cast<const char*>(scr + dma_s * SLICE);
        const char* usrc = wbase + (size_t)dma_s * (UP * 16);
        int lq = lane;
        asm volatile("" : "+v"(lq));                     // addresses are made here, per call: 24 per-lane pointers kept across the k-loop spill
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int p = (k * 8 + wave) * 64 + lq;
            const bool all_in = (k + 1) * 512 <= INP, all_u = k * 512 >= INP;              // compile-time per k
            const char* gp;
            if (all_in) gp = ssrc + p * 16;
            else if (all_u) gp = usrc + min(p - INP, UP - 1) * 16;
            else gp = p < INP ? ssrc + p * 16 : usrc + min(p - INP, UP - 1) * 16;
            __builtin_amdgcn_global_load_lds(GLB_PTR(gp), LDS_PTR(sbase + (k * 8 + wave) * 256), 16, 0, 0);
        }
        ++dma_s;
        if (++dma_slot == R) dma_slot = 0;
    };

    // one k-step of conv_wino4.hip (S0: the tile's first; WAITN: VMEM operations younger than this k-step's DMA)
    auto step = [&](auto s0_tag, auto waitn_tag) {
        constexpr bool S0 = decltype(s0_tag)::value;
        constexpr int WAITN = decltype(waitn_tag)::value;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
        __builtin_amdgcn_s_barrier();
        if (L + R - 1 < KS) dma_issue();
        if (L == KS - 1 && have_next) {                  // slots 1, 2 are free from here on: the next tile's first two bands
            src_nxt = frame_src(nxt);
            band_issue(src_nxt, nxt, 0);
            band_issue(src_nxt, nxt, 1);
        }
        int lq = lane;
        asm volatile("" : "+v"(lq));                     // the lane's LDS offsets are made per k-step (a handful of VALU): only `lane` stays live
        const int qj = lq & 15, qg = lq >> 4;
        const int lbase = qg * PLANE + (gy * 8 + (qj >> 3) * 4) * ROWP + (gx * 8 + (qj & 7)) * 4 + 2;
        const int wlbase = INP * 4 + lq * 4;
        const float* sl = lds + slot * STAGE;
        f32x2 t05[6], t12[6], t34[6];
        {
            const float* p = sl + lbase;
            f32x2 pout[6];
            f32x4 pmid[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                pout[r] = f32x2{p[r * ROWP + 1], p[r * ROWP + 6]};
                pmid[r] = *reinterpret_cast<const f32x4*>(p + r * ROWP + 2);
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
                    if (p == 7) {
                        const f32x4 biasq = *reinterpret_cast<const f32x4*>(bias2_s + qg * 4);
                        acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq0[p & 3], v[pp], biasq, 0, 0, 0);
                    } else
                        acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq0[p & 3], v[pp], z, 0, 0, 0);
                } else {
                    acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq0[p & 3], v[pp], acc[p], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        ++L;
        slot = slot + 1 == R ? 0 : slot + 1;
    };

    auto output = [&](int it) {
        int lq = lane;
        asm volatile("" : "+v"(lq));
        const int j = lq & 15, g = lq >> 4, tx = j & 7, ty = j >> 3;
        const int hw = a.h1 * a.w1;
        const int oy = cur.by * TH + gy * 8 + ty * 4, ox = cur.bx * TW + (gx * 8 + tx) * 4;
        const int co0 = g * 4;
        float* dst = a.out + (size_t)cur.n * C * hw;
        const bool full = cur.by * TH + TH <= a.h1 && cur.bx * TW + TW <= a.w1;          // wave-uniform
        const bool inx = ox < a.w1;
        const unsigned lane_bo = (unsigned)((co0 * a.h1 + oy) * a.w1 + ox) * 4u;
        float psum[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x2 u[6][4];
#pragma unroll
            for (int xi = 0; xi < 6; ++xi) {
                f32x2 m[6];
#pragma unroll
                for (int nu = 0; nu < 6; ++nu) m[nu] = f32x2{acc[xi * 6 + nu][2 * h], acc[xi * 6 + nu][2 * h + 1]};
                at6(m, u[xi]);
            }
            f32x2 y[4][4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                f32x2 m[6], o[4];
#pragma unroll
                for (int xi = 0; xi < 6; ++xi) m[xi] = u[xi][x];
                at6(m, o);
#pragma unroll
                for (int yy = 0; yy < 4; ++yy) y[yy][x] = o[yy];
            }
#pragma unroll
            for (int yy = 0; yy < 4; ++yy)
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const f32x2 sc = 0.1f * y[yy][x];
                    y[yy][x] = f32x2{fmaxf(y[yy][x][0], sc[0]), fmaxf(y[yy][x][1], sc[1])};
                }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = 2 * h + e;
                char* rb = reinterpret_cast<char*>(dst) + (size_t)r * hw * 4;
                float ps = 0.f;
#pragma unroll
                for (int yy = 0; yy < 4; ++yy) {
                    const f32x4 o = f32x4{y[yy][0][e], y[yy][1][e], y[yy][2][e], y[yy][3][e]};
                    const bool in = inx && oy + yy < a.h1;
                    if (full) {
                        *reinterpret_cast<f32x4*>(rb + lane_bo + (size_t)yy * a.w1 * 4) = o;
                    } else {
                        float* p = in ? reinterpret_cast<float*>(rb + lane_bo + (size_t)yy * a.w1 * 4) : a.trash + lq * 4;
                        *reinterpret_cast<f32x4*>(p) = o;
                    }
                    const float rs = (o[0] + o[1]) + (o[2] + o[3]);
                    ps += (full || in) ? rs : 0.f;
                }
                psum[r] = ps;
            }
        }
        constexpr int SW = POOLK / 4;
        float* red = red0 + (it & 1) * RED1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sred = window_sum<SW>(psum[r]);
            if ((tx & (SW - 1)) == 0 && ty == 0) red[(gy * C + co0 + r) * NWX + (gx * 8 + tx) / SW] = sred;
        }
    };

    using T = std::true_type;
    using F = std::false_type;
    for (int it = 0; it < ntile; ++it) {
        nxt = cur;
        advance(nxt);
        have_next = it + 1 < ntile;
#ifdef EEM_DIAG
        if (!(a.dbg & 1)) {
#endif
        if (wave < 2) phase_a(std::integral_constant<int, 3>{});
        else phase_a(std::integral_constant<int, 2>{});
#ifdef EEM_DIAG
        }
        if (a.dbg & 2) { prv = cur; advance(cur); src_cur = frame_src(cur); continue; }
#endif
        if (it > 0) pool_finish(it - 1, prv);            // written before phase A's first barrier; one more store in flight below
        L = 0; slot = 0; dma_s = 0; dma_slot = 0;
        dma_issue();
        dma_issue();
        // VMEM operations younger than k-step s's DMA: the next slice's pieces (+ the pooling store issued in front of them)
        step(T{}, std::integral_constant<int, NI>{});
        step(F{}, std::integral_constant<int, NI>{});
        step(F{}, std::integral_constant<int, NI>{});
        step(F{}, std::integral_constant<int, 0>{});
        output(it);
        prv = cur;
        advance(cur);
        src_cur = src_nxt;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    pool_finish(ntile - 1, prv);
}

}  // namespace

bool enc12_supported(const Enc12Args& a) {
    return (a.wraw & 3) == 0 && a.win == a.wraw && (a.w1 & 3) == 0 && ((uintptr_t)a.in0 & 15) == 0 && ((uintptr_t)a.in1 & 15) == 0 &&
           ((uintptr_t)a.out & 15) == 0 && a.pool_partial != nullptr && a.h1 == (a.hin - 1) / 2 + 1 && a.w1 == (a.win - 1) / 2 + 1;
}

int enc12_blocks(int nimg, int h1, int w1, int blocks_per_xcd) {
    const int T = ceil_div(w1, TW) * ceil_div(h1, TH) * nimg;
    int per_xcd = ceil_div(T, 8);
    static const int env_cap = enc_blocks_per_xcd("E12", 0);
    const int cap = env_cap > 0 ? env_cap : (blocks_per_xcd > 0 ? blocks_per_xcd : 32);
    if (per_xcd > cap) per_xcd = cap;
    return per_xcd * 8;
}

int enc12_launch(const Enc12Args& a0, int blocks, hipStream_t stream) {
    EEM_REQUIRE(a0.wpk1 && a0.bias1 && a0.u2 && a0.bias2 && a0.zero_page && a0.trash && a0.scratch && a0.out && a0.pool_partial,
                "enc12_launch: NULL operand");
    Enc12Args a = a0;
#ifdef EEM_DIAG
    { const char* e = getenv("EEM_E12_DBG"); a.dbg = e ? atoi(e) : 0; }      // 1: no phase A, 2: no phase B, 4: A's stores to the trash page, 8: A without MFMAs
#endif
    a.tiles_x = ceil_div(a.w1, TW);
    a.tiles_y = ceil_div(a.h1, TH);
    { static const bool ro = [] { const char* e = getenv("EEM_E12_ROW_ORDER"); return e && e[0] == '1'; }(); a.row_order = ro ? 1 : 0; }
    EEM_NOTE_GRID(blocks, 512);
    hipLaunchKernelGGL(enc12_kernel, dim3(blocks), dim3(512), 0, stream, a);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}
