// The stride-2 encoder layers pconv2_1 (16 -> 32) and pconv3_1 (32 -> 64) (EEMFlow.py:77,80) on the bf16 matrix pipe with fp32 results: every operand is cut into
// three bf16 pieces, a = a0 + a1 + a2 EXACTLY (8 significand bits each, by truncation: a0 = a & 0xffff0000, a1 the same of a - a0, a2
// the rest), and a product a * b is the six bf16 products a0 b0, a0 b1, a1 b0, a0 b2, a1 b1, a2 b0 summed in fp32 by the MFMA (the
// three dropped terms are below 2^-24 |a b|).  tools/micro/bf16x3.hip: over K = 144 the worst error of that form is 1.8e-7 of
// sum |a b| against 1.5e-7 for v_mfma_f32_32x32x2_f32 - the same arithmetic quality, not a reduced precision.
//
// Why: on gfx950 the fp32 MFMA runs at the fp32 VECTOR rate (64 FLOP/clk/SIMD, 1/16 of bf16) and holds the vector issue port while it
// does (VALU work beside it is additive, profiles/r03_pk_valu.txt), the bf16 MFMA is a separate pipe that holds the port for 8 of its
// 32 cycles.  A k-step of 16 (one tap x 16 channels) is eight v_mfma_f32_32x32x2_f32 = 512 cycles, or six v_mfma_f32_32x32x16_bf16 = 192
// cycles with the split's VALU work underneath (same microbenchmark: 12 MFMAs + 24 VALU = 404 cycles against 384 for the MFMAs alone).
//
// Layout: conv_s2.hip's - one 4 x 32-pixel tile per 4-wave block, the whole fp32 input tile (CIN planes of 9 rows x 68 columns)
// by LDS-DMA, one wait, one barrier; a wave = an output row, all couts.  Per k-step (a tap x 16 channels) a lane reads its pixel's 8
// channels (ds_read_b32), splits them (44 VALU instructions) and issues six MFMAs per 32 couts; the weights arrive pre-split
// ([k-step][piece][cout tile][lane][4 dwords], bx3_transform_launch) through a ring of three k-steps of global loads (L1 / L2 hits:
// every block reads the same 27 / 108 KB).  Software pipeline: k-step s + 1's raw values are in registers when k-step s's MFMAs
// issue, and their split sits between those MFMAs (pinned by sched_barrier; the scheduler's own order put the whole split in front).
// pconv2_1 (16 -> 32): 40 KB of LDS, four blocks per CU, 17.0 -> 13.3 us; pconv3_1 (32 -> 64, EEM_NO_BX3_64=1 for conv_enc2.hip's
// kernel): 80 KB, two per CU, 14.5 -> 11.1 us.  What is left is not arithmetic: the tile's way in from HBM and the stores.
// Measured and not kept: pconv2_1 as persistent blocks (weights stationary in 108 registers, two LDS stages, tile t + 1 streaming in under
// tile t's MFMAs and tile t - 1's stores, counted waits; two blocks per CU): 14.1 us at two tiles per block, 17.2 at three on 150 CUs,
// against 13.1 for these one-tile blocks on the same box, and 8 790-8 890 frames/s with four frames in flight against 8 850-8 880.
// Inputs with an infinity give NaN here where the fp32 kernels give an infinity (inf - inf in the split): the output is non-finite at
// exactly the positions where the fp32 kernel's is and unchanged elsewhere (contract, tests/test_gpu_parity.py::
// test_bf16_piece_kernel_and_an_infinity_in_the_input); a guard would add two VALU per value to a split that runs once per USE here.
#include "common.h"

namespace {

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CIN, int COUT>
struct BxCfg {
    static constexpr int TH = 4, NPIX = 32, WAVES = 4;
    static constexpr int MT = COUT / 32;                 // cout tiles, all of them on every wave
    static constexpr int NC16 = CIN / 16;                // 16-channel chunks: a k-step = (tap, chunk)
    static constexpr int KSTEPS = 9 * NC16;
    static constexpr int IN_ROWS = 2 * (TH - 1) + 3;     // 9
    static constexpr int ROWP = 68;                      // staged floats per row: columns 2*x0 - 4 .. 2*x0 + 63
    static constexpr int PPR = ROWP / 4;
    static constexpr int PLANE = IN_ROWS * ROWP;
    static constexpr int PIECES = CIN * IN_ROWS * PPR;
    static constexpr int NI = (PIECES + WAVES * 64 - 1) / (WAVES * 64);
    static constexpr int LDS_FLOATS = NI * WAVES * 256;
    static constexpr int RING = 3;                       // k-steps of weight fragments in flight
};

__device__ __forceinline__ bf16x8 as_bf(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

// eight fp32 values -> three vectors of eight bf16 (element e in the low / high half of dword e / 2)
__device__ __forceinline__ void split8(const float (&x)[8], u32x4& p0, u32x4& p1, u32x4& p2) {
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

// the same for dword d alone (elements 2 d and 2 d + 1): a quarter of the work, to be placed between two MFMAs
__device__ __forceinline__ void split_pair(const float (&x)[8], int d, u32x4& p0, u32x4& p1, u32x4& p2) {
    const float xa = x[2 * d], xb = x[2 * d + 1];
    const float ra = xa - __uint_as_float(__float_as_uint(xa) & 0xffff0000u), rb = xb - __uint_as_float(__float_as_uint(xb) & 0xffff0000u);
    const float sa = ra - __uint_as_float(__float_as_uint(ra) & 0xffff0000u), sb = rb - __uint_as_float(__float_as_uint(rb) & 0xffff0000u);
    p0[d] = __builtin_amdgcn_perm(__float_as_uint(xb), __float_as_uint(xa), 0x07060302u);
    p1[d] = __builtin_amdgcn_perm(__float_as_uint(rb), __float_as_uint(ra), 0x07060302u);
    p2[d] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256, (CIN == 16 ? 4 : 2)) void bx3_s2_kernel(EncConvArgs a, const u32x4* __restrict__ wq) {
    using C = BxCfg<CIN, COUT>;
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    ENC_ARGS_NOW(a);
    const unsigned lid = xcd_logical_block(blockIdx.x, gridDim.x);
    if (lid >= (unsigned)(a.tiles_x * a.tiles_y * a.nimg)) return;
    const int bx = lid % a.tiles_x, by = (lid / a.tiles_x) % a.tiles_y;
    const int n = lid / (a.tiles_x * a.tiles_y);
    const int row = wave;
    const int j = lane & 31, g = lane >> 5;

    // ---- weight fragments of the first k-steps: requested first, they land while the DMA plan is computed
    // wq[((s * 3 + piece) * MT + mt) * 64 + lane], s = tap * NC16 + chunk
    const u32x4* wsrc = wq + lane;
    u32x4 wv[C::RING][3][C::MT];
#pragma unroll
    for (int s = 0; s < C::RING; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int m = 0; m < C::MT; ++m) wv[s][p][m] = wsrc[((s * 3 + p) * C::MT + m) * 64];

    // ---- the whole input tile by LDS-DMA: piece p = (channel, tile row, 16-byte column), channel-major as it lies in LDS
    const int oy0 = by * C::TH, ox0 = bx * C::NPIX;
    const int gy0 = oy0 * 2 - 1, gxa = ox0 * 2 - 4;
    const float* src = a.in0 + (size_t)n * CIN * a.hin * a.win;
#pragma unroll
    for (int k = 0; k < C::NI; ++k) {
        int p = (wave + k * C::WAVES) * 64 + lane;
        const bool real = p < C::PIECES;
        p = real ? p : 0;
        const int c = p / (C::IN_ROWS * C::PPR);
        const int rem = p - c * (C::IN_ROWS * C::PPR);
        const int ry = rem / C::PPR;
        const int q = rem - ry * C::PPR;
        const int gy = gy0 + ry, gx = gxa + q * 4;
        const bool ok = real && gy >= 0 && gy < a.hin && gx >= 0 && gx < a.win;
        const float* gp = ok ? src + ((size_t)(c * a.hin + gy) * a.win + gx) : a.zero_page;
        __builtin_amdgcn_global_load_lds(GLB_PTR(gp), LDS_PTR(lds + (wave + k * C::WAVES) * 256), 16, 0, 0);
    }

    f32x16 acc[C::MT];
#pragma unroll
    for (int m = 0; m < C::MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = a.bias[m * 32 + (r & 3) + 8 * (r >> 2) + 4 * g];

    // B operand of k-step (tap t, chunk c16): channels c16 * 16 + 8 g + e, input row 2 * row + ky, column 2 * j + kx + 3
    const float* bl = lds + g * 8 * C::PLANE + 2 * row * C::ROWP + 2 * j + 3;
    auto read_x = [&](int s, float (&x)[8]) __attribute__((always_inline)) {
        const int t = s / C::NC16, c16 = s % C::NC16;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = bl[(c16 * 16 + e) * C::PLANE + (t / 3) * C::ROWP + (t % 3)];
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // software pipeline: k-step s + 1's raw values are in registers when k-step s's MFMAs issue, and their split goes between those
    float x[2][8];
    u32x4 b[2][3];
    read_x(0, x[0]);
    if (C::KSTEPS > 1) read_x(1, x[1]);
    split8(x[0], b[0][0], b[0][1], b[0][2]);
#pragma unroll
    for (int s = 0; s < C::KSTEPS; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        u32x4(&w)[3][C::MT] = wv[s % C::RING];
        if (s + 2 < C::KSTEPS) read_x(s + 2, x[cur]);               // (x[cur] was split during the previous k-step)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 6; ++q) {                                // small terms first
            constexpr int PW[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int m = 0; m < C::MT; ++m)
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(w[PW[q]][m]), as_bf(b[cur][PB[q]]), acc[m], 0, 0, 0);
            if (q < 4 && s + 1 < C::KSTEPS) split_pair(x[nxt], q, b[nxt][0], b[nxt][1], b[nxt][2]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ring: this slot's MFMAs have issued - request the fragments of the k-step that uses the slot next
        if (s + C::RING < C::KSTEPS) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int m = 0; m < C::MT; ++m) w[p][m] = wsrc[(((s + C::RING) * 3 + p) * C::MT + m) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: LeakyReLU, NCHW stores (a wave's 32 lanes of a cout row are 128 consecutive bytes)
    const int oy = oy0 + row, ox = ox0 + j;
    const int hw = a.hout * a.wout;
    float* dst = a.out + (size_t)n * COUT * hw;
    const bool full = oy0 + C::TH <= a.hout && ox0 + C::NPIX <= a.wout;           // block-uniform
#pragma unroll
    for (int m = 0; m < C::MT; ++m) {
        const int co0 = m * 32 + 4 * g;
        const unsigned lane_bo = (unsigned)((co0 * a.hout + oy) * a.wout + ox) * 4u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[m][r];
            if (a.act) v = fmaxf(v, 0.1f * v);
            const int dco = (r & 3) + 8 * (r >> 2);
            if (full) {
                char* rb = reinterpret_cast<char*>(dst) + (size_t)dco * hw * 4;   // scalar base per register, one 32-bit lane offset
                *reinterpret_cast<float*>(rb + lane_bo) = v;
            } else if (oy < a.hout && ox < a.wout) {
                dst[(size_t)(co0 + dco) * hw + oy * a.wout + ox] = v;
            }
        }
    }
}

// OIHW fp32 weights -> the kernel's stream of pre-split A fragments: dword ((((s * 3 + piece) * MT + mt) * 64 + lane) * 4 + d) holds
// the piece of W[mt * 32 + lane % 32][chunk * 16 + 8 (lane / 32) + 2 d (+ 1)][tap] in its low (high) half, s = tap * NC16 + chunk
__global__ void bx3_wt_kernel(const float* __restrict__ w, int cin, int cout, unsigned* __restrict__ out, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int nc16 = cin / 16, mt_n = cout / 32;
    const int d = i & 3, lane = (i >> 2) & 63;
    int rest = i >> 8;
    const int mt = rest % mt_n; rest /= mt_n;
    const int s = rest;                                   // tap * nc16 + chunk
    const int t = s / nc16, c16 = s % nc16;
    const int co = mt * 32 + (lane & 31), c = c16 * 16 + 8 * (lane >> 5) + 2 * d;
    unsigned lo[3], hi[3];
    for (int h = 0; h < 2; ++h) {
        const float x = w[((size_t)co * cin + c + h) * 9 + t];
        const float x0 = __uint_as_float(__float_as_uint(x) & 0xffff0000u), r = x - x0;
        const float x1 = __uint_as_float(__float_as_uint(r) & 0xffff0000u), q = r - x1;
        unsigned* dst = h ? hi : lo;
        dst[0] = __float_as_uint(x0) >> 16; dst[1] = __float_as_uint(x1) >> 16; dst[2] = __float_as_uint(q) >> 16;
    }
    for (int p = 0; p < 3; ++p) out[((((size_t)s * 3 + p) * mt_n + mt) * 64 + lane) * 4 + d] = (hi[p] << 16) | lo[p];
}

// ---- the stride-1 layers pconv2_2 (32 -> 32) and pconv3_2 (64 -> 64) (EEMFlow.py:78,80) in the same arithmetic.  One 4 x 32-pixel tile
// and all couts per block; a wave = an output row.  Splitting a B operand where it is used costs 44 VALU instructions per six MFMAs and
// every input value is used by nine taps of up to four rows (measured in that form: 23 us for 32 -> 32 against 22.5 us for F(4x4) on
// 120 CUs - VALU-bound).  Here the tile is split ONCE, on its way in: a thread loads 8 channels x 4 columns (eight 16-byte global loads),
// splits the 32 values and writes twelve 16-byte LDS entries [piece][8-channel group][row][column] = 8 bf16; the main loop is three
// ds_read_b128 for B, 3 MT for A and 6 MT MFMAs per k-step, no VALU.
// A k-step's weight fragments (3 pieces x MT cout tiles x 1 KB) go L2 -> LDS once per block by LDS-DMA into a ring, one barrier per
// k-step (through L1 per wave they would be the CU's whole 64 B/clk at 64 -> 64).
// NOT the default (EEM_BX3_S1 = mask, 1: 32 -> 32, 2: 64 -> 64): correct (tests/test_gpu_parity.py), and at 1280x720 20.3 / 16.7 us
// per launch against 22.6 / 36.3 us for the F(4x4) kernels - but those run on 120 / 64 CUs (10.6 / 9.1 us of chip time), these hold
// every CU, and with four frames in flight the frame rate DROPS 3.4 % / 3 % (8 500 -> 8 210 / 8 240).  Where the time goes (launches
// with parts switched off, 32 -> 32): launch + weight DMA 3.0 us, staging 3, the k loop 9 (two rounds of 2 blocks per CU; 5.8 us of
// MFMA in all), stores 5.7 - one after the other, because a block's phases only overlap with ONE other block's.  What would pay is
// the F(4x4) kernels' shape: a persistent block per CU with the weights stationary in LDS and the next tile's staging under this
// tile's MFMAs (not built).
//   32 -> 32: four waves, 63 KB of LDS, two blocks per CU (900 blocks at 180 x 320).
//   64 -> 64: EIGHT waves - two groups of four split the input channels (group kg: channels 32 kg .. 32 kg + 31) and exchange halves
//   of their partial sums through LDS at the end, each finishing 32 couts; 138 KB of LDS, one block per CU (230 blocks at 90 x 160).
template <int CIN, int COUT, int KGT>
struct B1Cfg {
    static constexpr int TH = 4, NPIX = 32, KG = KGT, WAVES = 4 * KG, THREADS = 64 * WAVES;
    static constexpr int MT = COUT / 32;
    static constexpr int NC16 = CIN / 16, NCG = NC16 / KG;          // 16-channel chunks, per group
    static constexpr int KLOC = 9 * NCG;                            // k-steps of a group
    static constexpr int NG = CIN / 8;                              // 8-channel groups: one 16-byte entry per pixel, piece and group
    static constexpr int IN_ROWS = TH + 2, COLS = 40, QPR = COLS / 4;   // columns x0 - 4 .. x0 + 35
    static constexpr int ITEMS = NG * IN_ROWS * QPR;                // (group, row, 4-column piece): one per thread
    static constexpr int GPLANE = IN_ROWS * COLS;                   // entries per (piece, group)
    static constexpr int TILE_U4 = 3 * NG * GPLANE;
    static constexpr int SLOT_U4 = 3 * MT * 64;                     // 16-byte fragments per k-step
    static constexpr int NQ = 3 * MT;                               // 1 KB wave-instructions per slot: instruction q comes from row q % 4
    static constexpr int RING = KG == 1 ? 6 : 4, AHEAD = RING - 1;  // slots; k-steps issued ahead of the one whose MFMAs run
    static constexpr int LDS_U4 = TILE_U4 + KG * RING * SLOT_U4;
    static_assert(ITEMS <= THREADS && NC16 % KG == 0 && NQ <= 8 && (KG == 1 || MT == 2), "one staging item per thread; two groups finish one cout tile each");
    static_assert(KG == 1 || MT * 4 * 64 * 4 <= TILE_U4, "partial-sum exchange fits the dead input tile");
    static_assert(LDS_U4 * 16 <= 160 * 1024, "LDS");
};

template <int N>
__device__ __forceinline__ void bx_wait_vm(int nrow) {            // s_waitcnt vmcnt(nrow * N), nrow in 0..2 (wave-uniform)
    if (nrow == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * N) : "memory");
    else if (nrow == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int CIN, int COUT, int KGT>
__global__ __launch_bounds__((64 * 4 * KGT), (KGT == 1 ? 2 : 1)) void bx3_s1_kernel(EncConvArgs a, const u32x4* __restrict__ wq) {
    using C = B1Cfg<CIN, COUT, KGT>;
    __shared__ __attribute__((aligned(16))) u32x4 lds[C::LDS_U4];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = C::KG > 1 ? wave >> 2 : 0, row = wave & 3;
    ENC_ARGS_NOW(a);
    const unsigned lid = xcd_logical_block(blockIdx.x, gridDim.x);
    if (lid >= (unsigned)(a.tiles_x * a.tiles_y * a.nimg)) return;
    const int bx = lid % a.tiles_x, by = (lid / a.tiles_x) % a.tiles_y;
    const int n = lid / (a.tiles_x * a.tiles_y);
    const int j = lane & 31, g = lane >> 5;
    const int oy0 = by * C::TH, ox0 = bx * C::NPIX;

    // ---- weight ring of this group: local k-step i = tap * NCG + c  ->  global k-step tap * NC16 + kg * NCG + c
    u32x4* const ring = lds + C::TILE_U4 + kg * C::RING * C::SLOT_U4;
    const char* const wbase = reinterpret_cast<const char*>(wq) + lane * 16;
    const int nrow = (C::NQ + 3 - row) / 4;                        // this wave's instructions per slot (wave-uniform)
    auto issue_a = [&](int i) __attribute__((always_inline)) {
        const int ic = i < C::KLOC ? i : C::KLOC - 1;              // past the end: a harmless reload into a dead slot keeps the counts uniform
        const int sg = (ic / C::NCG) * C::NC16 + kg * C::NCG + ic % C::NCG;
        const char* sp = wbase + (size_t)sg * C::SLOT_U4 * 16;
        u32x4* slot = ring + (i % C::RING) * C::SLOT_U4;
        if (row < C::NQ) __builtin_amdgcn_global_load_lds(GLB_PTR(sp + row * 1024), LDS_PTR(slot + row * 64), 16, 0, 0);
        if (row + 4 < C::NQ) __builtin_amdgcn_global_load_lds(GLB_PTR(sp + (row + 4) * 1024), LDS_PTR(slot + (row + 4) * 64), 16, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < C::AHEAD; ++i) issue_a(i);

    // ---- the input tile, split on its way in: this thread's item = (8-channel group cg, tile row r, 4-column piece q)
    {
        const int item = tid < C::ITEMS ? tid : 0;
        const int cg = item / (C::IN_ROWS * C::QPR), rq = item - cg * (C::IN_ROWS * C::QPR);
        const int r = rq / C::QPR, q = rq - r * C::QPR;
        const int gy = oy0 - 1 + r, gx = ox0 - 4 + 4 * q;
        const bool in = gy >= 0 && gy < a.hin && gx >= 0 && gx < a.win;     // a piece is inside or outside as a whole (win % 4 == 0)
        const float* sp = a.in0 + ((size_t)(n * CIN + cg * 8) * a.hin + (in ? gy : 0)) * a.win + (in ? gx : 0);
        f32x4 v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const f32x4*>(sp + (size_t)e * a.hin * a.win);
        u32x4* dst = lds + (cg * C::IN_ROWS + r) * C::COLS + 4 * q;            // + piece * NG * GPLANE + column
        if (tid < C::ITEMS) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = in ? v[e][k] : 0.f;
                u32x4 p0, p1, p2;
                split8(x, p0, p1, p2);
                dst[k] = p0;
                dst[C::NG * C::GPLANE + k] = p1;
                dst[2 * C::NG * C::GPLANE + k] = p2;
            }
        }
    }

    f32x16 acc[C::MT];
#pragma unroll
    for (int m = 0; m < C::MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    // B operand of local k-step (tap t, chunk c), piece p: entry [p][group 2 (kg NCG + c) + g][row + ky][j + kx + 3]
    const u32x4* bl = lds + ((kg * C::NCG * 2 + g) * C::IN_ROWS + row) * C::COLS + j + 3;
    const u32x4* const ring4 = ring + lane;
    auto read_b = [&](int i, u32x4 (&b)[3]) __attribute__((always_inline)) {
        const int t = i / C::NCG, c = i % C::NCG;
#pragma unroll
        for (int p = 0; p < 3; ++p) b[p] = bl[(p * C::NG + 2 * c) * C::GPLANE + (t / 3) * C::COLS + (t % 3)];
    };
    auto read_w = [&](int i, u32x4 (&w)[3][C::MT]) __attribute__((always_inline)) {
        const u32x4* sl = ring4 + (i % C::RING) * C::SLOT_U4;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int m = 0; m < C::MT; ++m) w[p][m] = sl[(p * C::MT + m) * 64];
    };
    // slot 0 has landed and the tile is written (for every wave: barrier); k-step 0's operands
    bx_wait_vm<C::AHEAD - 1>(nrow);
    __syncthreads();
    u32x4 w[2][3][C::MT], b[2][3];
    read_w(0, w[0]);
    read_b(0, b[0]);
#pragma unroll
    for (int i = 0; i < C::KLOC; ++i) {
        const int cur = i & 1, nxt = cur ^ 1;
        // slot i + 1 has landed (k-steps i + 2 .. i + AHEAD - 1 may still fly), for every wave; and every wave has consumed slot i - 1
        bx_wait_vm<C::AHEAD - 2>(nrow);
        __builtin_amdgcn_s_barrier();
        issue_a(i + C::AHEAD);
        if (i + 1 < C::KLOC) { read_w(i + 1, w[nxt]); read_b(i + 1, b[nxt]); }
        __builtin_amdgcn_sched_barrier(0);                           // (the next k-step's requests first, then this one's MFMAs)
#pragma unroll
        for (int m = 0; m < C::MT; ++m) {                            // small terms first
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(w[cur][2][m]), as_bf(b[cur][0]), acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(w[cur][1][m]), as_bf(b[cur][1]), acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(w[cur][0][m]), as_bf(b[cur][2]), acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(w[cur][1][m]), as_bf(b[cur][0]), acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(w[cur][0][m]), as_bf(b[cur][1]), acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(w[cur][0][m]), as_bf(b[cur][0]), acc[m], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // (the clamped reloads still write LDS)

    const int oy = oy0 + row, ox = ox0 + j;
    const int hw = a.hout * a.wout;
    float* dst = a.out + (size_t)n * COUT * hw;
    if constexpr (C::KG == 2) {
        // ---- the groups exchange halves: group kg keeps cout tile kg and hands the other one over (the input tile is dead by now)
        __builtin_amdgcn_s_barrier();
        f32x16* xch = reinterpret_cast<f32x16*>(lds);               // [group that reads it][row][lane]
        xch[((1 - kg) * 4 + row) * 64 + lane] = kg == 0 ? acc[1] : acc[0];
        __syncthreads();
        f32x16 mine = kg == 0 ? acc[0] : acc[1];
        const f32x16 other = xch[(kg * 4 + row) * 64 + lane];
        const int co0 = kg * 32 + 4 * g;
        if (oy < a.hout && ox < a.wout) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (r & 3) + 8 * (r >> 2);
                float v = mine[r] + other[r] + a.bias[co];
                if (a.act) v = fmaxf(v, 0.1f * v);
                dst[(size_t)co * hw + oy * a.wout + ox] = v;
            }
        }
    } else {
        if (oy < a.hout && ox < a.wout) {
#pragma unroll
            for (int m = 0; m < C::MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = m * 32 + 4 * g + (r & 3) + 8 * (r >> 2);
                    float v = acc[m][r] + a.bias[co];
                    if (a.act) v = fmaxf(v, 0.1f * v);
                    dst[(size_t)co * hw + oy * a.wout + ox] = v;
                }
        }
    }
}

template <int CIN, int COUT, int KGT>
int bx3_s1_launch_t(const EncConvArgs& a0, hipStream_t stream) {
    using C = B1Cfg<CIN, COUT, KGT>;
    EncConvArgs a = a0;
    a.tiles_x = ceil_div(a.wout, C::NPIX);
    a.tiles_y = ceil_div(a.hout, C::TH);
    dim3 grid((unsigned)ceil_div(a.tiles_x * a.tiles_y * a.nimg, 8) * 8);
    EEM_NOTE_GRID(grid.x, C::THREADS);
    EEM_NOTE_PIPE(1);
    hipLaunchKernelGGL((bx3_s1_kernel<CIN, COUT, KGT>), grid, dim3(C::THREADS), 0, stream, a, reinterpret_cast<const u32x4*>(a.wbx3));
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

template <int CIN, int COUT>
int bx3_launch_t(const EncConvArgs& a0, hipStream_t stream) {
    using C = BxCfg<CIN, COUT>;
    EncConvArgs a = a0;
    a.tiles_x = ceil_div(a.wout, C::NPIX);
    a.tiles_y = ceil_div(a.hout, C::TH);
    dim3 grid((unsigned)ceil_div(a.tiles_x * a.tiles_y * a.nimg, 8) * 8);
    EEM_NOTE_GRID(grid.x, 256);
    EEM_NOTE_PIPE(1);
    hipLaunchKernelGGL((bx3_s2_kernel<CIN, COUT>), grid, dim3(256), 0, stream, a, reinterpret_cast<const u32x4*>(a.wbx3));
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

}  // namespace

bool bx3_shape(int cin, int cout, int stride) {
    return (stride == 2 && ((cin == 16 && cout == 32) || (cin == 32 && cout == 64))) || (stride == 1 && cin == cout && (cin == 32 || cin == 64));
}

size_t bx3_packed_floats(int cin, int cout) { return (size_t)9 * (cin / 16) * 3 * (cout / 32) * 64 * 4; }

int bx3_transform_launch(const float* w, int cin, int cout, float* packed, hipStream_t stream) {
    const int total = 9 * (cin / 16) * (cout / 32) * 64 * 4;
    hipLaunchKernelGGL(bx3_wt_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, w, cin, cout, reinterpret_cast<unsigned*>(packed), total);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

bool bx3_supported(int cin, int cout, int stride, const EncConvArgs& a) {
    const char* off = getenv("EEM_NO_BX3");              // read per launch: the tests compare both kernels in one process
    if (off && off[0] == '1') return false;
    if (stride == 2 && cin == 32) {                       // pconv3_1: EEM_NO_BX3_64=1 keeps conv_enc2.hip's chunked kernel
        const char* m = getenv("EEM_NO_BX3_64");
        if (m && m[0] == '1') return false;
    }
    if (stride == 1) {                                    // EEM_BX3_S1 = mask of channel widths: 1 = 32, 2 = 64
        const char* m = getenv("EEM_BX3_S1");
        const int mask = m ? atoi(m) : 0;                 // off by default: see the note at bx3_s1_kernel
        if (!(mask & (cin == 32 ? 1 : 2))) return false;
    }
    return bx3_shape(cin, cout, stride) && a.wbx3 && a.gate == nullptr && a.pool_partial == nullptr && a.res == nullptr && (a.win & 3) == 0 &&
           (a.act == 0 || a.act == 1) && (((uintptr_t)a.in0) & 15) == 0 && (size_t)cin * a.hin * a.win * 4 < (1u << 31);
}

int bx3_launch(int cin, int stride, const EncConvArgs& a, hipStream_t stream) {
    if (cin == 64) return bx3_s1_launch_t<64, 64, 2>(a, stream);
    if (cin == 32 && stride == 2) return bx3_launch_t<32, 64>(a, stream);
    if (cin == 32) return bx3_s1_launch_t<32, 32, 1>(a, stream);
    return bx3_launch_t<16, 32>(a, stream);
}
