// Stride-1 convolutions with 32-channel-aligned inputs (E-RAFT's update block and encoders' residual stacks, EEMFlow+'s decoders: 1x1, 3x3,
// 1x5, 5x1 kernels, 64..384 -> 32..576 channels, inputs that are the concatenation of up to three tensors) as an implicit GEMM on the
// bf16 matrix pipe with fp32 results: the three-piece arithmetic of conv_bx3.hip - every fp32 operand = three bf16 pieces that sum to
// it EXACTLY (8 significand bits each, by truncation), a product = six piece products (a0 b0, a0 b1, a1 b0, a0 b2, a1 b1, a2 b0; the
// dropped three are below 2^-24 of it) accumulated in fp32 by the MFMA, small terms first.  tools/micro/bf16x3.hip: worst error over
// K = 144 is 1.8e-7 of sum |a b| against 1.5e-7 for the fp32 MFMA - the same arithmetic, not a reduced precision.
// Non-finite operands (contract, tested by tests/test_gpu_bwd_ops.py::test_bf16_piece_wide_conv_and_an_infinity_in_the_input): the split of
// an infinity is inf - inf = NaN, so an infinite input leaves this kernel as NaN where gconv16.hip leaves an infinity - the output is
// non-finite at exactly the positions where the fp32 kernel's is, every other value agrees, nothing non-finite ever becomes a number
// (the training step's inf / NaN test sees it either way).  A guard would cost two VALU per value of the split for no caller of the path.
//
// Why here: these are direct convolutions with K = 256 .. 3 456 and no cheaper algorithm (1x5, 5x1, 1x1; the 3x3 ones sit beside them in
// the same loops); gconv16.hip runs them at 0.4-0.55 of the fp32 MFMA's 155 TFLOP/s.  tools/micro/mfma_clock.hip: six
// v_mfma_f32_16x16x32_bf16 per product sustain 2.1 PFLOP/s / 6 = 350 TFLOP/s of fp32 products with every CU issuing.  What the form
// costs is operand traffic - a weight is 6 bytes and a 16x16x32 MFMA consumes 2 KB of operands per 16 cycles - so it is built for
// launches with pixels enough to fill the chip with LARGE tiles (gconvb_supported); smaller ones stay on gconv16.hip.
//
//   * block = 12 waves = TH rows (2 / 4 / 6 / 8, chosen per launch) x 16 pixels x 128 couts: eight MULTIPLYING waves - wave = (row half,
//     cout quarter): TH / 2 pixel tiles (rows) x 2 cout groups of 16, v_mfma_f32_16x16x32_bf16 with M = 16 pixels, N = 16 couts, K = 32
//     channels: 6 TH MFMAs per k-step (tap x 32-channel chunk) and wave against 1.5 TH A-fragment reads (LDS) and 6 B-fragment loads
//     (global) - two of them per SIMD - and four STAGING waves, one per SIMD;
//   * INPUT: the chunk's haloed tile, split once on its way in: a staging thread loads 8 channels x 4 columns (eight 16-byte loads, one
//     chunk ahead, out-of-image pieces from the zero page), splits the 32 values and writes twelve 16-byte LDS entries
//     [piece][8-channel group][row][column] = 8 bf16; double-buffered, one barrier per chunk;
//   * WEIGHTS: pre-split on the host into B fragments [64-cout chunk][32-channel chunk][tap][cout group][piece][lane] (16 bytes),
//     streamed L2 -> registers through a ring of two k-steps per wave - no LDS, no barrier;
//   * D: lane = (cout, 4 consecutive pixels): epilogue operands (GRU state and gate, residual, per-pixel addend) and results move as
//     16-byte loads / stores; every epilogue of gconv16.hip.
#include "gconv.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 gb_bf(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

template <int KH, int KW, int THT>
struct GBCfg {
    static constexpr int TH = THT, TW = 16, PT = THT / 2;        // a multiplying wave's pixel tiles (rows)
    static constexpr int ROWS = TH + KH - 1;
    static constexpr int XOFF = KW > 1 ? 4 : 0;                  // staged columns x0 - XOFF .. (16-byte pieces)
    static constexpr int COLS = TW + 2 * XOFF, QPR = COLS / 4;
    static constexpr int PLANE = ROWS * COLS;                    // entries per (piece, 8-channel group)
    static constexpr int STAGE = 3 * 4 * PLANE;                  // entries per chunk buffer
    static constexpr int ITEMS = 4 * ROWS * QPR;                 // (group, row, piece): one per thread
    static constexpr int TAPS = KH * KW;
    static constexpr int PH = KH / 2, PW = KW / 2;
    static_assert(ITEMS <= 256 && PLANE % 16 == 0, "one staging item per thread; planes are multiples of 256 bytes (ds_read_b128 banks)");
    static_assert((2 * STAGE + 64) * 16 <= 160 * 1024, "LDS");
};

__device__ __forceinline__ float gb_act(float v, int act) {
    switch (act) {
        case GACT_RELU: return v > 0.f ? v : 0.f;
        case GACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case GACT_TANH: return tanhf(v);
        case GACT_LEAKY: return v > 0.f ? v : 0.1f * v;
        default: return v;
    }
}

#ifdef EEM_DIAG
// diagnostic builds, EEM_GB_DBG (tools/gconvb_phases.sh): 1 no MFMAs, 2 no weight loads inside the k-loop, 4 no A-fragment reads inside the
// k-loop, 8 the staging waves only keep the barriers, 16 no epilogue operands / activation (plain store), 32 every wave leaves at once
// (the launch alone), 64 no staging prologue either (with 8), 128 no stores
__device__ int g_gb_dbg;
#endif

template <int KH, int KW, int THT, int RING>
__global__ __launch_bounds__(768, 3) void gconvb_kernel(GConvArgs ka, const u32x4* __restrict__ wq, int tiles_x, int nchunks) {
    // every launch argument the kernel uses, as scalars of its own: closures that reach the argument STRUCT by reference kept a copy of it in
    // scratch (328 bytes stored and re-read per thread)
    const int a_hin = ka.hin, a_win = ka.win, a_hout = ka.hout, a_wout = ka.wout, a_cout = ka.cout, a_act = ka.act, a_epi = ka.epi, a_nseg = ka.nseg;
    const int a_split = ka.split, a_out_ctotal = ka.out_ctotal, a_out_coff = ka.out_coff, a_out2_ctotal = ka.out2_ctotal;
    const int a_pre_ctotal = ka.pre_ctotal, a_pre_coff = ka.pre_coff, a_e0_ctotal = ka.e0_ctotal, a_e0_coff = ka.e0_coff, a_e1_ctotal = ka.e1_ctotal, a_e1_coff = ka.e1_coff;
    const float a_out_scale = ka.out_scale;
    const float* const a_zero_page = ka.zero_page; const float* const a_scale = ka.scale; const float* const a_shift = ka.shift;
    const float* const a_pre = ka.pre; const float* const a_e0 = ka.e0; const float* const a_e1 = ka.e1;
    float* const a_out = ka.out; float* const a_out2 = ka.out2;
    const float* const seg_ptr0 = ka.seg[0].ptr; const float* const seg_ptr1 = ka.seg[1].ptr; const float* const seg_ptr2 = ka.seg[2].ptr;
    const int seg_c0 = ka.seg[0].c, seg_c1 = ka.seg[1].c, seg_ct0 = ka.seg[0].ctotal, seg_ct1 = ka.seg[1].ctotal, seg_ct2 = ka.seg[2].ctotal;
    const int seg_co0 = ka.seg[0].coff, seg_co1 = ka.seg[1].coff, seg_co2 = ka.seg[2].coff;
    using C = GBCfg<KH, KW, THT>;
    constexpr int PLANE = C::PLANE, COLS = C::COLS, TAPS = C::TAPS, PT = C::PT;
    __shared__ __attribute__((aligned(256))) u32x4 lds[2 * C::STAGE + 64];   // two chunk buffers + a sink for threads that stage nothing
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave12 = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef EEM_DIAG
    const int dbg = __builtin_amdgcn_readfirstlane(g_gb_dbg);
#else
    constexpr int dbg = 0;
#endif
    if (dbg & 32) return;
    // Roles: waves 0-7 multiply (ds_read_b128 + weight loads + MFMAs, nothing else in their stream), two per SIMD; waves 8-11 stage the
    // next chunk's tile (global loads, the split, LDS writes), one per SIMD.  Why: the split's VALU work runs on the SIMD's vector pipe
    // BESIDE the MFMAs instead of between them; vmcnt retires in order - a multiplying wave that had issued the staging loads (HBM)
    // waited for them whenever it waited for a weight fragment (L2) issued later (diagnostic builds of the one-role kernel: the weight
    // loads cost 30 of 123 us per launch, the conversion 33, in a kernel whose MFMAs alone take 56); and two multipliers on a SIMD
    // cover each other's operand round trips, so neither needs a double-buffered A operand or a deep weight ring (168 registers).
    const bool stager = wave12 >= 8;
    const int wave = wave12 & 7;
    const int ph = wave & 1, cq = wave >> 1;                     // row half, cout quarter (two groups of 16) of the block's 128 couts
    const int m = lane & 15, kg = lane >> 4;
    const int n = blockIdx.z;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * C::TH, x0 = tx * C::TW;
    const int hw = a_hin * a_win;
    f32x4 acc[PT][2];
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (stager) {
        // ---- input staging: item = (8-channel group cg of the chunk, tile row r, 16-byte column piece q)
        const int st = tid - 512;
        const float* sp0 = seg_ptr0 + ((size_t)n * seg_ct0 + seg_co0) * hw;
        const float* sp1 = a_nseg > 1 ? seg_ptr1 + ((size_t)n * seg_ct1 + seg_co1) * hw : nullptr;
        const float* sp2 = a_nseg > 2 ? seg_ptr2 + ((size_t)n * seg_ct2 + seg_co2) * hw : nullptr;
        const int sc0 = seg_c0, sc1 = a_nseg > 1 ? seg_c1 : 0;
        const bool s_act = st < C::ITEMS;
        const int item = s_act ? st : 0;
        const int s_cg = item / (C::ROWS * C::QPR), s_rq = item - s_cg * (C::ROWS * C::QPR);
        const int s_r = s_rq / C::QPR, s_q = s_rq - s_r * C::QPR;
        const int s_gy = y0 - C::PH + s_r, s_gx = x0 - C::XOFF + 4 * s_q;
        const bool s_in = s_act && s_gy >= 0 && s_gy < a_hin && s_gx >= 0 && s_gx < a_win;      // a piece is inside or outside as a whole (win % 4 == 0)
        const size_t s_off = s_in ? (size_t)(s_cg * 8) * hw + (size_t)s_gy * a_win + s_gx : 0;
        const size_t s_step = s_in ? (size_t)hw : 0;
        const int s_dst = s_act ? (s_cg * C::ROWS + s_r) * COLS + 4 * s_q : 2 * C::STAGE + 4 * (lane & 15);   // (+ piece * 4 * PLANE + column; + buffer); idle threads: the sink
        const int s_pstride = s_act ? 4 * PLANE : 0;
        f32x4 sv[8];
        auto stage_load = [&](int ch) __attribute__((always_inline)) {      // (past the last chunk: the first one again, converted into a dead buffer)
            const int c0 = (ch < nchunks ? ch : 0) * 32, c1 = c0 - sc0, c2 = c1 - sc1;
            // (the three bases pass through an empty asm: the compiler otherwise folds the selects into ONE load at a selected offset of
            // the closure that holds them by reference - an indexed read that keeps every captured variable, the staging registers
            // included, in scratch)
            const float *q0 = sp0, *q1 = sp1, *q2 = sp2;
            asm volatile("" : "+s"(q0), "+s"(q1), "+s"(q2));
            const float* b = c0 < sc0 ? q0 : (c1 < sc1 ? q1 : q2);
            const int cb = c0 < sc0 ? c0 : (c1 < sc1 ? c1 : c2);
            const float* sp = s_in ? b + (size_t)cb * hw + s_off : a_zero_page;
#pragma unroll
            for (int e = 0; e < 8; ++e) sv[e] = *reinterpret_cast<const f32x4*>(sp + e * s_step);
        };
        auto convert = [&](int buf) __attribute__((always_inline)) {       // 32 values -> twelve 16-byte entries
            u32x4* d = lds + (s_act ? buf * C::STAGE : 0) + s_dst;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                u32x4 cp[3];
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {
                    const float xa = sv[2 * dd][k], xb = sv[2 * dd + 1][k];
                    const float ra = xa - __uint_as_float(__float_as_uint(xa) & 0xffff0000u), rb = xb - __uint_as_float(__float_as_uint(xb) & 0xffff0000u);
                    const float sa = ra - __uint_as_float(__float_as_uint(ra) & 0xffff0000u), sb = rb - __uint_as_float(__float_as_uint(rb) & 0xffff0000u);
                    cp[0][dd] = __builtin_amdgcn_perm(__float_as_uint(xb), __float_as_uint(xa), 0x07060302u);      // (hi16(xb) << 16) | hi16(xa)
                    cp[1][dd] = __builtin_amdgcn_perm(__float_as_uint(rb), __float_as_uint(ra), 0x07060302u);
                    cp[2][dd] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
                }
                d[k] = cp[0];
                d[s_pstride + k] = cp[1];
                d[2 * s_pstride + k] = cp[2];
            }
        };
        if (!(dbg & 64)) {
            stage_load(0);
            convert(0);
            stage_load(1);
        }
        __syncthreads();
        for (int ch = 0; ch < nchunks; ++ch) {
            if (!(dbg & 8)) {
                convert((ch + 1) & 1);                                       // chunk ch + 1's tile, while the multipliers read chunk ch's
                stage_load(ch + 2);
            }
            __syncthreads();
        }
        return;
    }

    // ---- multipliers.  Weights: fragment (cc, ch, tap, cog, piece) at wq[((((cc * nchunks + ch) * TAPS + tap) * 4 + cog) * 3 + piece) * 64 + lane],
    // cc = 64-cout chunk; this wave's two groups are 2 cq, 2 cq + 1 of the block's eight
    const int cog8 = 2 * cq;
    const int ccw = blockIdx.y * 2 + (cog8 >> 2);
    const bool wlive = ccw * 64 + (cog8 & 3) * 16 < a_cout;                  // (a cout count that ends inside the block: idle multipliers only keep the barriers)
    const u32x4* wbase = wq + ((size_t)(wlive ? ccw : 0) * nchunks * TAPS * 4 + (cog8 & 3)) * 3 * 64 + lane;
    // the ring: k-step s's fragments live in slot s % RING and are requested RING - 1 k-steps before their MFMAs (RING = 2; three slots
    // for the tiles of up to six rows were measured and are no faster - gb_launch)
    u32x4 bw[RING][2][3];
    auto load_b = [&](auto slot_tag, int s) __attribute__((always_inline)) {      // k-step s = ch * TAPS + tap (clamped past the end: a harmless reload)
        constexpr int SL = decltype(slot_tag)::value;
        const int sc = s < nchunks * TAPS ? s : nchunks * TAPS - 1;
        const u32x4* p = wbase + (size_t)sc * (4 * 3 * 64);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) bw[SL][q][pc] = p[(q * 3 + pc) * 64];
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    load_b(P0{}, 0);
    if constexpr (RING > 2) load_b(P1{}, 1);
    __syncthreads();

    // A fragment of (tile row p, tap (ky, kx), piece pc): entry ((pc * 4 + kg) * ROWS + 4 ph + p + ky) * COLS + XOFF - PW + kx + m
    const unsigned a_lane = (unsigned)((kg * C::ROWS + PT * ph) * COLS + C::XOFF - C::PW + m) * 16u;
    const char* lb = reinterpret_cast<const char*>(lds);

    // one chunk: TAPS k-steps of 6 TH MFMAs; PAR = ring slot of its first k-step's weights = (ch * TAPS) % RING
    u32x4 av[PT][3];
    auto chunk = [&](auto par_tag, int ch) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_tag)::value;
        const unsigned abuf = a_lane + (unsigned)((ch & 1) * C::STAGE) * 16u;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int ky = t / KW, kx = t % KW;
            if (!(dbg & 4) || (ch == 0 && t == 0)) {
#pragma unroll
            for (int p = 0; p < PT; ++p)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
                    av[p][pc] = *reinterpret_cast<const u32x4*>(lb + abuf + ((pc * 4 * PLANE) + (p + ky) * COLS + kx) * 16);
            }
            // the weights of k-step s + RING - 1, into the slot the previous k-step has left
            const int s = ch * TAPS + t;
            const int slot = (PAR + t + RING - 1) % RING;                    // (t is an unrolled loop's counter: folded)
            if (!(dbg & 2)) {
                if (slot == 0) load_b(P0{}, s + RING - 1);
                else if (slot == 1) load_b(P1{}, s + RING - 1);
                else if constexpr (RING > 2) load_b(std::integral_constant<int, 2>{}, s + RING - 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(dbg & 1))
#pragma unroll
            for (int i = 0; i < 6; ++i) {                                    // small terms first
                constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int p = 0; p < PT; ++p)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gb_bf(av[p][PA[i]]), gb_bf(bw[(PAR + t) % RING][q][PB[i]]), acc[p][q], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                                     // this chunk's tile is read, the next one's is written
    };
    if (wlive) {
        // (groups of RING chunks without a branch inside: the weights a chunk's last k-steps request for the next chunk's first ones are
        // used in the same basic block - behind `if (ch + 1 < nchunks)` the compiler sank the loads into the conditional block, past
        // the barrier, and waited for them there: one exposed L2 round trip per pair of chunks)
        int ch = 0;
        for (; ch + RING <= nchunks; ch += RING) {
            chunk(P0{}, ch);
            chunk(std::integral_constant<int, TAPS % RING>{}, ch + 1);
            if constexpr (RING > 2) chunk(std::integral_constant<int, (2 * TAPS) % RING>{}, ch + 2);
        }
        if (ch < nchunks) {
            chunk(P0{}, ch);
            if (RING > 2 && ch + 1 < nchunks) chunk(std::integral_constant<int, TAPS % RING>{}, ch + 1);
        }
    } else {
        for (int ch = 0; ch < nchunks; ++ch) __syncthreads();
        return;
    }

    // ---- epilogue: lane = cout (2 cq + q) * 16 + m of the block's 128, pixels (y0 + 4 ph + p, x0 + 4 kg .. + 3)
    const int hwo = a_hout * a_wout;
    const int x = x0 + 4 * kg;
    float e_scale[2], e_shift[2];
    int co[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        co[q] = ccw * 64 + ((cog8 & 3) + q) * 16 + m;
        const int cl = min(co[q], a_cout - 1);
        e_scale[q] = a_scale ? a_scale[cl] : 1.f;
        e_shift[q] = a_shift ? a_shift[cl] : 0.f;
    }
    // operands: 16-byte loads from clamped indices, all of an operand's eight in flight together (a load inside a lane-dependent branch
    // is followed by the compiler's s_waitcnt vmcnt(0))
    unsigned ip[PT][2];
    bool ok[PT][2];
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int y = y0 + PT * ph + p;
            ok[p][q] = x < a_wout && y < a_hout && co[q] < a_cout;
            ip[p][q] = ok[p][q] ? (unsigned)(co[q] * hwo + y * a_wout + x) : 0u;
        }
    f32x4 e0v[PT][2], e1v[PT][2], prv[PT][2];
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) { e0v[p][q] = z4; e1v[p][q] = z4; prv[p][q] = z4; }
    const int a_epi_ld = (dbg & 16) ? GEPI_PLAIN : a_epi;
    if (a_pre && !(dbg & 16)) {
        const float* b = a_pre + ((size_t)n * a_pre_ctotal + a_pre_coff) * hwo;
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q) prv[p][q] = *reinterpret_cast<const f32x4*>(b + ip[p][q]);
    }
    const unsigned zsplit = a_epi == GEPI_ZR ? (unsigned)(a_split * hwo) : 0u;   // GEPI_ZR: e0 is indexed by co - split, from split on
    if (a_epi_ld != GEPI_PLAIN) {
        const float* b = a_e0 + ((size_t)n * a_e0_ctotal + a_e0_coff) * hwo;
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q) e0v[p][q] = *reinterpret_cast<const f32x4*>(b + (ip[p][q] < zsplit ? 0u : ip[p][q] - zsplit));
    }
    if (a_epi_ld == GEPI_GRU) {
        const float* b = a_e1 + ((size_t)n * a_e1_ctotal + a_e1_coff) * hwo;
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q) e1v[p][q] = *reinterpret_cast<const f32x4*>(b + ip[p][q]);
    }
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (!ok[p][q] || (dbg & 128)) continue;
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = acc[p][q][j] * e_scale[q] + e_shift[q] + prv[p][q][j];
                t = gb_act(t, (dbg & 16) ? GACT_NONE : a_act);
                if (a_epi == GEPI_MUL) t *= e0v[p][q][j];
                else if (a_epi == GEPI_GRU) t = (1.f - e1v[p][q][j]) * e0v[p][q][j] + e1v[p][q][j] * t;
                else if (a_epi == GEPI_ADD_RELU) { t += e0v[p][q][j]; t = t > 0.f ? t : 0.f; }
                else if (a_epi == GEPI_ADD) t += e0v[p][q][j];
                else if (a_epi == GEPI_ZR && co[q] >= a_split) t *= e0v[p][q][j];
                v[j] = t * a_out_scale;
            }
            const int y = y0 + PT * ph + p;
            const size_t pix = (size_t)y * a_wout + x;
            // GEPI_ZR: r leaves as r * h, to the second output.  One store through a selected pointer (gconv16.hip's note on two stores)
            float* d = (a_epi == GEPI_ZR && co[q] >= a_split) ? a_out2 + ((size_t)n * a_out2_ctotal + (co[q] - a_split)) * hwo + pix
                                                             : a_out + ((size_t)n * a_out_ctotal + a_out_coff + co[q]) * hwo + pix;
            *reinterpret_cast<f32x4*>(d) = v;
        }
}

template <int KH, int KW, int THT, int RING>
int gb_launch_t(const GConvArgs& a, hipStream_t stream) {
    using C = GBCfg<KH, KW, THT>;
    int cin = 0;
    for (int s = 0; s < a.nseg; ++s) cin += a.seg[s].c;
    const int tiles_x = ceil_div(a.wout, C::TW), tiles_y = ceil_div(a.hout, C::TH);
    dim3 grid(tiles_x * tiles_y, ceil_div(a.cout, 128), a.n);
#ifdef EEM_DIAG
    { static int once = [] { const char* e = getenv("EEM_GB_DBG"); int v = e ? atoi(e) : 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gb_dbg), &v, sizeof v); return v; }(); (void)once; }
#endif
    hipLaunchKernelGGL((gconvb_kernel<KH, KW, THT, RING>), grid, dim3(768), 0, stream, a, reinterpret_cast<const u32x4*>(a.wpkb), tiles_x, cin / 32);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

// Rows per tile (2, 4, 6 or 8): a launch of a few blocks per CU runs as long as the CU with the most blocks - E-RAFT's z | r conv at batch 4
// is 320 blocks of 8 rows (two rounds, the second a quarter full), 400 of 6 (two rounds of 3/4 the work) or 640 of 4 (three rounds of half
// the work); its 128-cout convs are 160 blocks of 8 rows on 256 CUs or 200 of 6.  The count with the least (rounds x (rows + 2)), the
// per-block prologue and epilogue counted as two rows (anything from one to six rows gives the same frame rates): E-RAFT 640x480 x 12
// batch 4 245 -> 256 frames/s against tiles of 4 / 8 rows only, batch 1 162 -> 165; every forced height is slower than the choice
// (EEM_GCONVB_TH = 2 / 4 / 6 / 8: 208 / 235 / 249 / 243 at batch 4).
template <int KH, int KW>
int gb_launch(const GConvArgs& a, hipStream_t stream) {
    static const int th_env = [] { const char* e = getenv("EEM_GCONVB_TH"); return e ? atoi(e) : 0; }();
    static const int cus = [] { int d = 0, n = 256; hipDeviceProp_t p; if (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) n = p.multiProcessorCount; return n > 0 ? n : 256; }();
    const auto cost = [&](int th) { return (float)ceil_div(ceil_div(a.wout, 16) * ceil_div(a.hout, th) * ceil_div(a.cout, 128) * a.n, cus) * (float)(th + 2); };
    int th = th_env;
    if (th != 2 && th != 4 && th != 6 && th != 8) {
        th = 8;
        for (int t = 6; t >= 2; t -= 2)
            if (cost(t) < cost(th) - 1e-3f) th = t;
    }
#ifdef EEM_DIAG
    // diagnostic builds, EEM_GCONVB_RING=3: the tiles of up to six rows request their weights two k-steps ahead (three slots).  Measured
    // (tools/gconvb_ring.sh): E-RAFT 640x480 x 12 batch 1 178.1 / 179.2 against 179.9 / 180.0 frames/s, batch 4 272.1 / 272.4 against
    // 275.5 / 275.6 - the weights' round trip is not what a k-step waits for
    const char* rg = getenv("EEM_GCONVB_RING");
    if (rg && rg[0] == '3') {
        if (th == 2) return gb_launch_t<KH, KW, 2, 3>(a, stream);
        if (th == 4) return gb_launch_t<KH, KW, 4, 3>(a, stream);
        if (th == 6) return gb_launch_t<KH, KW, 6, 3>(a, stream);
    }
#endif
    switch (th) {
        case 2: return gb_launch_t<KH, KW, 2, 2>(a, stream);
        case 4: return gb_launch_t<KH, KW, 4, 2>(a, stream);
        case 6: return gb_launch_t<KH, KW, 6, 2>(a, stream);
        default: return gb_launch_t<KH, KW, 8, 2>(a, stream);
    }
}

}  // namespace

// gconv16.hip's fp32 fragment stream of the same weights -> this file's pre-split B fragments, on the device: what a training step does
// after every weight update (ops.hip gathers the fp32 stream - forward or transposed-flipped for the data gradient - by an index table;
// the split is arithmetic, not a gather).  gconv16: stream[((((cc * nch16 + ch16) * 4 + mt) * taps + tap) * 64 + lane16) * 4 + cg] =
// W[cc * 64 + mt * 16 + lane16 % 16][ch16 * 16 + 4 cg + lane16 / 16][tap]
namespace {
__global__ void gconvb_from16_kernel(const float* __restrict__ w16, int cin, int taps, unsigned* __restrict__ out, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int nch32 = cin / 32, nch16 = cin / 16;
    const int d = (int)(i & 3), lane = (int)((i >> 2) & 63);
    long rest = i >> 8;
    const int pc = (int)(rest % 3); rest /= 3;
    const int cog = (int)(rest & 3); rest >>= 2;
    const int tap = (int)(rest % taps); rest /= taps;
    const int ch = (int)(rest % nch32);
    const int cc = (int)(rest / nch32);
    const int c = ch * 32 + 8 * (lane >> 4) + 2 * d;
    unsigned pcs[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int cch = c + h, ch16 = cch >> 4, r = cch & 15, cg = r >> 2, lh = r & 3;
        const float x = w16[(((((size_t)cc * nch16 + ch16) * 4 + cog) * taps + tap) * 64 + (lane & 15) + 16 * lh) * 4 + cg];
        const float x0 = __uint_as_float(__float_as_uint(x) & 0xffff0000u), r1 = x - x0;
        const float x1 = __uint_as_float(__float_as_uint(r1) & 0xffff0000u), q = r1 - x1;
        pcs[h] = (pc == 0 ? __float_as_uint(x0) : pc == 1 ? __float_as_uint(x1) : __float_as_uint(q)) >> 16;
    }
    out[i] = (pcs[1] << 16) | pcs[0];
}
}  // namespace

int gconvb_from16_launch(const float* wpk16, int cout, int cin, int taps, float* wpkb, hipStream_t stream) {
    const long total = (long)ceil_div(cout, 64) * (cin / 32) * taps * 4 * 3 * 64 * 4;
    hipLaunchKernelGGL(gconvb_from16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, wpk16, cin, taps,
                       reinterpret_cast<unsigned*>(wpkb), total);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

bool gconvb_shape(int cout, const int* cs, int nseg, int kh, int kw, int stride) {
    // (a block multiplies 128 couts: narrower layers - EEMFlow+'s 96 -> 64 -> 32 decoder convs - would idle most of its multiplying waves
    // while the staging work per tile stays the same; they stay on gconv16.hip)
    if (stride != 1 || cout < 96) return false;
    if (!((kh == 1 && kw == 1) || (kh == 3 && kw == 3) || (kh == 1 && kw == 5) || (kh == 5 && kw == 1))) return false;
    for (int s = 0; s < nseg; ++s)
        if (cs[s] <= 0 || cs[s] % 32) return false;
    return true;
}

size_t gconvb_packed_floats(int cout, const int* cs, int nseg, int kh, int kw) {
    int cin = 0;
    for (int s = 0; s < nseg; ++s) cin += cs[s];
    return (size_t)ceil_div(cout, 64) * (cin / 32) * kh * kw * 4 * 3 * 64 * 4;
}

// u32x4 index ((((cc * nch + ch) * taps + tap) * 4 + cog) * 3 + piece) * 64 + lane; dword d of lane (cout = cc * 64 + cog * 16 + lane % 16,
// channels ch * 32 + 8 (lane / 16) + 2 d (+ 1)) holds the pieces of the two weights in its low (high) half
void gconvb_pack(const float* w, int cout, const int* cs, int nseg, int kh, int kw, float* packed) {
    int cin = 0;
    for (int s = 0; s < nseg; ++s) cin += cs[s];
    const int taps = kh * kw, nch = cin / 32;
    unsigned* out = reinterpret_cast<unsigned*>(packed);
    auto pieces = [](float x, unsigned (&p)[3]) {                   // (unions: the host pass of hipcc sees only the device's memcpy)
        union { float f; unsigned u; } v, h;
        v.f = x;
        h.u = v.u & 0xffff0000u;
        p[0] = h.u >> 16;
        v.f = x - h.f;
        h.u = v.u & 0xffff0000u;
        p[1] = h.u >> 16;
        v.f = v.f - h.f;
        p[2] = v.u >> 16;
    };
    for (int cc = 0; cc < ceil_div(cout, 64); ++cc)
        for (int ch = 0; ch < nch; ++ch)
            for (int tap = 0; tap < taps; ++tap)
                for (int cog = 0; cog < 4; ++cog)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int co = cc * 64 + cog * 16 + (lane & 15);
                        for (int d = 0; d < 4; ++d) {
                            unsigned lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
                            const int c = ch * 32 + 8 * (lane >> 4) + 2 * d;
                            if (co < cout) {
                                pieces(w[((size_t)co * cin + c) * taps + tap], lo);
                                pieces(w[((size_t)co * cin + c + 1) * taps + tap], hi);
                            }
                            for (int pc = 0; pc < 3; ++pc)
                                out[((((((size_t)cc * nch + ch) * taps + tap) * 4 + cog) * 3 + pc) * 64 + lane) * 4 + d] = (hi[pc] << 16) | lo[pc];
                        }
                    }
}

bool gconvb_supported(const GConvArgs& a) {
    const char* e = getenv("EEM_NO_GCONVB");                      // read per call: a test flips it inside one process
    if (e && e[0] == '1') return false;
    if (!a.wpkb || !a.zero_page || a.tstride > 1 || a.stride != 1 || a.pad_h != a.kh / 2 || a.pad_w != a.kw / 2 || a.groups > 1 || a.out_cmul > 1) return false;
    int cs[3];
    for (int s = 0; s < a.nseg; ++s) {
        cs[s] = a.seg[s].c;
        if (a.seg[s].cmul > 1 || a.seg[s].gate || ((uintptr_t)a.seg[s].ptr & 15)) return false;
    }
    if (!gconvb_shape(a.cout, cs, a.nseg, a.kh, a.kw, a.stride)) return false;
    // 1x1: a k-step per chunk - the staging waves (one tile conversion per 48 MFMAs) set the pace: 72 us against gconv16's 45-63 at
    // E-RAFT's batch 4; EEM_GCONVB_1X1=1 sends them here all the same (tests)
    if (a.kh == 1 && a.kw == 1) { const char* o = getenv("EEM_GCONVB_1X1"); if (!(o && o[0] == '1')) return false; }
    if (a.win % 4 || a.wout % 4 || a.hout != a.hin || a.wout != a.win || ((uintptr_t)a.out & 15)) return false;
    if ((a.pre && ((uintptr_t)a.pre & 15)) || (a.epi != GEPI_PLAIN && ((uintptr_t)a.e0 & 15)) || (a.epi == GEPI_GRU && ((uintptr_t)a.e1 & 15)) ||
        (a.epi == GEPI_ZR && (((uintptr_t)a.out2 & 15) || a.split % 16)))
        return false;
    if ((size_t)a.cout * a.hin * a.win >= (1u << 30) || (size_t)32 * a.hin * a.win * 4 >= (1u << 31)) return false;
    // large tiles (128 pixels x 64 couts): launches that fill the chip with them; the others stay on gconv16.hip
    const char* mb = getenv("EEM_GCONVB_MINBLK");                 // (read per call, like the switch above: the tests run small shapes through it)
    const long min_blk = mb ? atol(mb) : 64L;
    const long blocks = (long)ceil_div(a.wout, 16) * ceil_div(a.hout, 8) * ceil_div(a.cout, 128) * a.n;
    return blocks >= min_blk;
}

int gconvb_launch(const GConvArgs& a, hipStream_t stream) {
    if (a.kh == 1 && a.kw == 1) return gb_launch<1, 1>(a, stream);
    if (a.kh == 3) return gb_launch<3, 3>(a, stream);
    if (a.kh == 1) return gb_launch<1, 5>(a, stream);
    return gb_launch<5, 1>(a, stream);
}
