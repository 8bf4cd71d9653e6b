// The two stride-2 encoder layers of 16 and 32 input channels (pconv2_1 16 -> 32, pconv3_1 32 -> 64: EEMFlow.py:77,80) as a direct
// convolution on v_mfma_f32_16x16x4_f32 in the form of conv_wino4.hip: 8 waves per block (two per SIMD), input AND weights through one
// LDS-DMA ring of k-step slices, no transform - the matrix pipe is the bound, not the vector pipe:
//   * a wave owns one 16-cout group and a SET of GR output rows x 16 output pixels: D[16 cout][16 pixels] per row, GR x 4 accumulator
//     registers; per k-step (4 input channels) it issues 9 x GR MFMAs, every A operand (a tap's 16 x 4 weights) feeding GR of them;
//   * B operands: the lane's pixel x needs input columns 2x-1, 2x, 2x+1 of rows 2y-1..2y+1 - two aligned ds_read_b64 per input row
//     (columns 2x-2..2x-1 and 2x..2x+1), each input row read once per k-step and used by the one or two output rows it belongs to;
//     channel planes sit 128 bytes apart modulo 256, so the two channel slots of a 32-lane ds_read_b64 group never collide;
//   * A operands: the k-step's 9 taps as three ds_read_b128 ([cog][q][lane][4] beside the input slice, staged by the same DMA);
//   * the slice of k-step L+2 is requested while k-step L computes (ring of three), across tile boundaries;
//   * bias rides in the accumulators (C operand of the first MFMA), LeakyReLU and the NCHW stores (64-byte runs per cout) follow.
// MEASURED (round 3, 1280x720): 18.9 us for pconv2_1 and 15.7 us for pconv3_1 on full grids, against 17.3 / 14.6 us for the light-block
// kernels of conv_s2.hip / conv_enc2.hip, and 7 720 against 7 900 frames/s with four frames in flight - so this kernel is OFF by
// default (EEM_S2R=1, read per launch, turns it on; tests run both).  Software-pipelined B reads and two interleaved accumulator
// streams changed nothing: with 4 / 8 k-steps per tile and one tile per block the launch is its prologue (DMA plan, first slice from
// HBM) plus one slice latency per k-step, not the matrix pipe; the layers are also within 2x of their HBM floor (47 / 24 MB).
#include <type_traits>

#include "common.h"

namespace {

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int CIN, int COUT, int GR>
struct S2RCfg {
    static constexpr int WAVES = 8;
    static constexpr int COG = COUT / 16;                // 16-cout groups
    static constexpr int NS = WAVES / COG;               // pixel sets per block tile, side by side
    static constexpr int TW = NS * 16, TH = GR;          // output pixels of a block tile
    static constexpr int KS = CIN / 4;                   // k-steps (4 input channels each)
    static constexpr int IN_ROWS = 2 * TH + 1;           // input rows 2*oy0 - 1 .. 2*oy0 + 2*TH - 1
    static constexpr int ROWP = 2 * TW + 8;              // staged columns 2*ox0 - 4 .. 2*ox0 + 2*TW + 3 (16-byte aligned start)
    static constexpr int PPR = ROWP / 4;
    static constexpr int PC = IN_ROWS * PPR;             // 16-byte pieces of one channel
    static constexpr int PLANE_P = (PC - 8 + 15) / 16 * 16 + 8;      // the smallest count >= PC that is 8 mod 16
    static constexpr int PLANE = PLANE_P * 4;            // floats: 32 mod 64
    static constexpr int INP = 4 * PLANE_P;              // pieces of a slice's input part
    static constexpr int UP = COG * 3 * 64;              // pieces of its weight part ([cog][q][lane] float4s: taps 4q .. 4q+3)
    static constexpr int NI = (INP + UP + 511) / 512;    // DMA wave-instructions per wave and k-step
    static constexpr int STAGE = NI * 2048;              // floats per ring slot
    static constexpr int R = 3;                          // ring slots
    static constexpr int NSTORE = GR * 4;                // stores per wave and tile
    static_assert(NS * COG == WAVES && KS >= 2, "wave roles");
    static_assert(PLANE_P >= PC && PLANE_P % 16 == 8, "channel planes 128 bytes apart modulo 256");
    static_assert((2 * TH * ROWP + 2 * TW + 8) * 4 < 65536, "ds_read immediate range");
    static_assert(NI + NSTORE <= 63, "vmcnt immediate");
    static_assert(R * STAGE * 4 <= 160 * 1024, "LDS budget");
};

template <int CIN, int COUT, int GR>
__global__ __launch_bounds__(512, 2) void s2r_kernel(EncConvArgs a) {
    using K = S2RCfg<CIN, COUT, GR>;
    constexpr int R = K::R, NI = K::NI, KS = K::KS;
    __shared__ __attribute__((aligned(256))) float lds[R * K::STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int cog = wave % K::COG, sx = wave / K::COG;

    const TileRange tr_ = block_tile_range(a.tiles_x * a.tiles_y * a.nimg, blockIdx.x, gridDim.x);
    const int ntile = tr_.count;
    if (ntile == 0) return;
    const int total = ntile * KS;
    TileCoord cur = tile_coord(tr_.first, a.tiles_x, a.tiles_y), nxt = cur;
    const char* zero_page = reinterpret_cast<const char*>(a.zero_page);
    const int plane_in = a.hin * a.win;

    // ---- DMA plan (as conv_wino4.hip): slot p = (k * 8 + wave) * 64 + lane of a slice is always the same piece
    unsigned poff[NI];
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
    const char* wbase = reinterpret_cast<const char*>(a.ws2r);          // [s][cog][q][lane] float4s, contiguous per k-step
    int dma_s = 0, dma_slot = 0;
    unsigned okm = 0;
    bool dma_interior = false;
    const char* dsrc = nullptr;
    auto dma_tile = [&]() {
        const int gy0 = 2 * nxt.by * K::TH - 1, gxa = 2 * nxt.bx * K::TW - 4;
        dma_interior = gy0 >= 0 && gy0 + K::IN_ROWS <= a.hin && gxa >= 0 && gxa + K::ROWP <= a.win;
        dsrc = reinterpret_cast<const char*>(a.in0 + (size_t)nxt.n * CIN * plane_in) + ((long)gy0 * a.win + gxa) * 4;
        if (!dma_interior) {
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
        const char* usrc = wbase + (size_t)dma_s * (K::UP * 16);
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
        dsrc += (size_t)plane_in * 16;
        if (++dma_s == KS) { dma_s = 0; tile_advance(nxt, a.tiles_x, a.tiles_y); dma_tile(); }
        if (++dma_slot == R) dma_slot = 0;
    };

    f32x4 biasq;
#pragma unroll
    for (int r = 0; r < 4; ++r) biasq[r] = a.bias[cog * 16 + g * 4 + r];

    // the lane's reads inside a slice: channel slot g, staged column 2 * (sx * 16 + j) + 2 (the pair 2x-2, 2x-1; the pair 2x, 2x+1 next to it)
    const int lbase = g * K::PLANE + 2 * (sx * 16 + j) + 2;
    const int wlbase = K::INP * 4 + (cog * 3 * 64 + lane) * 4;

    f32x4 acc[GR];
    int L = 0, slot = 0;

    auto step = [&](auto s0_tag, auto waitn_tag) {
        constexpr bool S0 = decltype(s0_tag)::value;
        constexpr int WAITN = decltype(waitn_tag)::value;
        if (L + 1 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (L + R - 1 < total) dma_issue();
        const float* sl = lds + slot * K::STAGE;
        const f32x4* wl = reinterpret_cast<const f32x4*>(sl + wlbase);
        const f32x4 w0 = wl[0], w1 = wl[64], w2 = wl[128];               // taps 0..3, 4..7, 8
        const float wt[9] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2[0]};
        const float* p = sl + lbase;
        // Input row ri (staged row 2 * gr + ky) is read once: an even row is tap row 0 of output row ri / 2 and tap row 2 of output row
        // ri / 2 - 1, an odd row tap row 1 of output row (ri - 1) / 2 - every accumulator takes its taps in the order ky = 0, 1, 2.
        // Two independent streams - output rows [0, GR/2) from input rows 0..GR, output rows [GR/2, GR) from input rows GR..2GR - are
        // issued alternately, so that consecutive MFMAs never share an accumulator (a dependent v_mfma_f32_16x16x4_f32 waits 40 cycles,
        // an independent one 32), and each row is requested one row ahead of its use.
        constexpr int H = GR / 2;
        static_assert(GR % 2 == 0, "two streams");
        auto load_row = [&](int ri, f32x2& lo, f32x2& hi) {
            lo = *reinterpret_cast<const f32x2*>(p + ri * K::ROWP);
            hi = *reinterpret_cast<const f32x2*>(p + ri * K::ROWP + 2);
        };
        f32x2 loA, hiA, loB, hiB, nloA, nhiA, nloB, nhiB;
        load_row(0, loA, hiA);
        load_row(GR, loB, hiB);
#pragma unroll
        for (int t = 0; t <= GR; ++t) {
            if (t < GR) { load_row(t + 1, nloA, nhiA); load_row(GR + t + 1, nloB, nhiB); }
            __builtin_amdgcn_sched_barrier(0);
            const float bA[3] = {loA[1], hiA[0], hiA[1]}, bB[3] = {loB[1], hiB[0], hiB[1]};      // columns 2x-1, 2x, 2x+1
            // local row t of a stream: odd -> tap row 1 of local output (t-1)/2; even -> tap row 2 of t/2-1 and tap row 0 of t/2
            if (t & 1) {
                const int o = (t - 1) / 2;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    acc[o] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[3 + kx], bA[kx], acc[o], 0, 0, 0);
                    acc[H + o] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[3 + kx], bB[kx], acc[H + o], 0, 0, 0);
                }
            } else {
                if (t >= 2) {
                    const int o = t / 2 - 1;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        acc[o] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[6 + kx], bA[kx], acc[o], 0, 0, 0);
                        acc[H + o] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[6 + kx], bB[kx], acc[H + o], 0, 0, 0);
                    }
                }
                if (t / 2 < H) {
                    const int o = t / 2;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        if (S0 && kx == 0) {
                            acc[o] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[0], bA[0], biasq, 0, 0, 0);
                            acc[H + o] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[0], bB[0], biasq, 0, 0, 0);
                        } else {
                            acc[o] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[kx], bA[kx], acc[o], 0, 0, 0);
                            acc[H + o] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[kx], bB[kx], acc[H + o], 0, 0, 0);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            loA = nloA; hiA = nhiA; loB = nloB; hiB = nhiB;
        }
        ++L;
        slot = slot + 1 == R ? 0 : slot + 1;
    };

    auto output = [&]() {
        const int hw = a.hout * a.wout;
        const int ox = cur.bx * K::TW + sx * 16 + j;
        const int co0 = cog * 16 + g * 4;
        float* dst = a.out + (size_t)cur.n * COUT * hw;
        const bool full = cur.by * K::TH + K::TH <= a.hout && cur.bx * K::TW + K::TW <= a.wout;    // wave-uniform
#pragma unroll
        for (int gr = 0; gr < GR; ++gr) {
            const int oy = cur.by * K::TH + gr;
            const unsigned lane_bo = (unsigned)((co0 * a.hout + oy) * a.wout + ox) * 4u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[gr][r];
                if (a.act) v = fmaxf(v, 0.1f * v);
                char* rb = reinterpret_cast<char*>(dst) + (size_t)r * hw * 4;
                // every lane stores (outside lanes into a scratch page): exactly NSTORE stores per wave and tile
                float* q = (full || (oy < a.hout && ox < a.wout)) ? reinterpret_cast<float*>(rb + lane_bo) : a.trash + lane;
                *q = v;
            }
        }
    };

    // ---- prologue: R-1 slices in flight
    dma_tile();
#pragma unroll
    for (int q = 0; q < R - 1; ++q)
        if (q < total) dma_issue();

    // vmcnt (operations younger than the DMA of k-step L when k-step L starts): the next slice's NI pieces; on the first two k-steps
    // of a later tile the previous tile's NSTORE stores as well
    using T = std::true_type;
    using F = std::false_type;
    for (int it = 0; it < ntile; ++it) {
        if (it == 0) {
            step(T{}, std::integral_constant<int, NI>{});
            step(F{}, std::integral_constant<int, NI>{});
        } else {
            step(T{}, std::integral_constant<int, NI + K::NSTORE>{});
            step(F{}, std::integral_constant<int, NI + K::NSTORE>{});
        }
#pragma unroll 1
        for (int s = 2; s < KS; ++s) step(F{}, std::integral_constant<int, NI>{});
        output();
        tile_advance(cur, a.tiles_x, a.tiles_y);
    }
}

// ---- weights in the order read above: [s = cin / 4][cog = cout / 16][q = tap / 4][lane = (cout % 16) + 16 * (cin % 4)][e = tap % 4]
__global__ void s2r_wt_kernel(const float* __restrict__ w, int cin, int cout, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= cin * cout) return;
    const int co = t / cin, ci = t - co * cin;
    const int ncog = cout / 16;
    const int cog = co >> 4, s = ci >> 2, lane = (co & 15) + 16 * (ci & 3);
    for (int p = 0; p < 12; ++p)
        out[((((size_t)s * ncog + cog) * 3 + (p >> 2)) * 64 + lane) * 4 + (p & 3)] = p < 9 ? w[((size_t)co * cin + ci) * 9 + p] : 0.f;
}

template <int CIN, int COUT, int GR>
int launch_s2r(const EncConvArgs& a0, hipStream_t stream) {
    using K = S2RCfg<CIN, COUT, GR>;
    EncConvArgs a = a0;
    a.tiles_x = ceil_div(a.wout, K::TW);
    a.tiles_y = ceil_div(a.hout, K::TH);
    const int T = a.tiles_x * a.tiles_y * a.nimg;
    int per_xcd = ceil_div(T, 8);
    static const int env_cap = enc_blocks_per_xcd(CIN == 16 ? "S16" : "S32", 0);     // tuning override
    const int cap = env_cap > 0 ? env_cap : 32;                 // one resident block per CU at most
    if (per_xcd > cap) per_xcd = cap;
    EEM_NOTE_GRID(per_xcd * 8, 512);
    hipLaunchKernelGGL((s2r_kernel<CIN, COUT, GR>), dim3(per_xcd * 8), dim3(512), 0, stream, a);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

}  // namespace

size_t s2r_packed_floats(int cin, int cout) { return (size_t)(cin / 4) * (cout / 16) * 3 * 64 * 4; }

bool s2r_shape(int cin, int cout, int stride) { return stride == 2 && ((cin == 16 && cout == 32) || (cin == 32 && cout == 64)); }

bool s2r_supported(int cin, int cout, int stride, const EncConvArgs& a) {
    const char* on = getenv("EEM_S2R");                         // read per launch: the tests compare both kernels in one process
    if (!(on && on[0] == '1')) return false;
    return s2r_shape(cin, cout, stride) && a.ws2r != nullptr && a.gate == nullptr && a.pool_partial == nullptr && (a.win & 3) == 0 &&
           (((uintptr_t)a.in0) & 15) == 0 && (size_t)cin * a.hin * a.win * 4 < (1u << 31);
}

int s2r_transform_launch(const float* w, int cin, int cout, float* packed, hipStream_t stream) {
    hipLaunchKernelGGL(s2r_wt_kernel, dim3(ceil_div(cin * cout, 256)), dim3(256), 0, stream, w, cin, cout, packed);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int s2r_launch(int cin, const EncConvArgs& a, hipStream_t stream) {
    EEM_REQUIRE(a.ws2r && a.zero_page && a.trash, "s2r_launch: NULL operand");
    if (cin == 16) return launch_s2r<16, 32, 8>(a, stream);
    return launch_s2r<32, 64, 4>(a, stream);
}
