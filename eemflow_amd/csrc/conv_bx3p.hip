// The stride-1 C -> C encoder layers at C = 32 and C = 64 (pconv2_2 / pconv2_3 / pconv3_2 / pconv3_3, EEMFlow.py:78-79,81-82) as DIRECT 3x3
// convolutions on the bf16 matrix pipe with fp32 results - the three-piece arithmetic of conv_bx3.hip (every fp32 operand = three bf16
// pieces that sum to it exactly, a product = six piece products accumulated in fp32 by the MFMA, small terms first) - in the form that
// file's notes ask for: PERSISTENT blocks, weights STATIONARY, the next rows staged under the current rows' MFMAs.
//
//   v_mfma_f32_16x16x32_bf16: M = 16 pixels of an output row, N = 16 couts, K = 32 input channels; a k-step = (filter tap, 32-channel
//   chunk): 9 at C = 32, 18 at C = 64; six MFMAs per k-step, pixel tile and cout group.
//   * WEIGHTS IN REGISTERS: a wave keeps the pre-split B fragments of every k-step for its couts - 9 x 2 cout groups (C = 32: a wave
//     finishes all 32 couts) or 18 x 1 (C = 64: the four waves of a block take 16 couts each) x 3 pieces x 4 dwords = 216 VGPRs, loaded
//     once per block.  One wave per SIMD (512 registers), four waves per block, one block per CU.
//   * INPUT IN AN LDS RING OF ROWS, split once on its way in ([piece][8-channel group][ring row][column] entries of 8 bf16 = 16 bytes:
//     an A fragment is one ds_read_b128, conflict-free).  A block walks DOWN a 32-pixel-wide strip in bands (C = 32: 4 rows, one per
//     wave; C = 64: 1 row, shared by the waves): each band adds 4 / 1 new input rows to the ring - no vertical halo is fetched twice
//     inside a block's segment - and the next band's rows are loaded (global -> registers) when a band starts and split + written to
//     LDS in the middle of its k-loop.  One barrier per band (216 MFMAs = 3 456 cycles per wave).
//   * the main loop is ds_read_b128 + MFMA: 6 reads per k-step against 24 (C = 32) / 12 (C = 64) MFMAs of 16 cycles.
//   * D: lane = (cout, 4 consecutive pixels): 16-byte NCHW stores; bias in the accumulators; LeakyReLU; optional stage-pooling
//     partial sums per (4 rows x POOLK columns) and an optional store-free form (f13 is only read through its pooled map).
// MFMA time of a layer: 2.265 GFLOP x 6 / 2.5 PFLOP/s = 5.4 us on the whole chip whatever the channel count - against 3.6 us of fp32
// MFMA + the transforms' VALU time beside it for F(4x4,3x3), which the measured kernels turn into 9-11 us of chip time.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int C>
struct PCfg {
    static constexpr int WAVES = 4, THREADS = 256;
    static constexpr int NCO = C == 32 ? 2 : 1;          // 16-cout groups per wave
    static constexpr int COG = C / 16;                   // 16-cout groups in all
    static constexpr int BROWS = C == 32 ? 4 : 1;        // output rows per band
    static constexpr int CHUNKS = C / 32, KSTEPS = 9 * CHUNKS;
    static constexpr int NG = C / 8;                     // 8-channel groups
    static constexpr int COLS = 40, QPR = 10;            // staged columns x0 - 4 .. x0 + 35, in 16-byte pieces
    static constexpr int RING = C == 32 ? 10 : 4;        // ring rows: a band's rows + the next band's new ones (even: planes are multiples of 256 bytes)
    static constexpr int PLANE = RING * COLS;            // entries per (piece, group)
    static constexpr int LDS_E = 3 * NG * PLANE;         // 16-byte entries
    static constexpr int FIRST_ROWS = BROWS + 2;
    static constexpr int POOLK = C == 32 ? 16 : 8;       // stage pooling window (EEMFlow.py:147-154)
    static constexpr int PROWS = 4;                      // rows per pooling partial sum
    static constexpr int NWX = 32 / POOLK;               // windows per strip
    static_assert(NG * FIRST_ROWS * QPR <= THREADS && RING % 2 == 0, "one staging item per thread");
    static_assert((2 * NG * PLANE + 4 * (CHUNKS - 1) * PLANE + 40) * 16 < 65536, "ds_read immediate range");
};

__device__ __forceinline__ bf16x8 as_bf(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

// eight fp32 values -> three vectors of eight bf16 (element e in the low / high half of dword e / 2), exact: x = p0 + p1 + p2
__device__ __forceinline__ void split8p(const float (&x)[8], u32x4& p0, u32x4& p1, u32x4& p2) {
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

// POOL: stage-pooling partial sums [n][cout][ceil(H / 4)][strips * NWX]; KEEP: the feature map is stored
template <int C, bool POOL, bool KEEP>
__global__ __launch_bounds__(256, 1) void bx3p_kernel(EncConvArgs a, const u32x4* __restrict__ wq, int seg_bands, int nseg) {
    using K = PCfg<C>;
    constexpr int NG = K::NG, RING = K::RING, COLS = K::COLS, PLANE = K::PLANE, NCO = K::NCO, BROWS = K::BROWS;
    __shared__ __attribute__((aligned(256))) u32x4 lds[K::LDS_E];
    __shared__ float red[POOL ? 4 * 2 * NCO * 64 : 1];                   // pooling: [wave][pixel tile][cout group][lane]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    ENC_ARGS_NOW(a);
    const int H = a.hout, W = a.wout;
    const int strips = a.tiles_x;
    const unsigned lid = xcd_logical_block(blockIdx.x, gridDim.x);
    if (lid >= (unsigned)(strips * nseg * a.nimg)) return;
    const int strip = lid % strips, sg = (lid / strips) % nseg, n = lid / (strips * nseg);
    const int x0 = strip * 32;
    const int nbands_all = (H + BROWS - 1) / BROWS;
    const int b0 = sg * seg_bands;
    const int nb = min(seg_bands, nbands_all - b0);
    if (nb <= 0) return;
    const int yfirst = b0 * BROWS;
    const int m = lane & 15, kg = lane >> 4;
    const int qbase = C == 32 ? 0 : wave;                                 // first cout group of this wave
    const int wrow = C == 32 ? wave : 0;                                  // this wave's row inside a band

    // ---- the wave's weight fragments: wq[((s * COG + cog) * 3 + piece) * 64 + lane]
    u32x4 wv[K::KSTEPS][NCO][3];
    {
        const u32x4* wsrc = wq + lane;
#pragma unroll
        for (int s = 0; s < K::KSTEPS; ++s)
#pragma unroll
            for (int q = 0; q < NCO; ++q)
#pragma unroll
                for (int p = 0; p < 3; ++p) wv[s][q][p] = wsrc[((s * K::COG + qbase + q) * 3 + p) * 64];
    }

    // ---- staging: item = (8-channel group cg, row r of the rows being staged, 16-byte column piece q); ring slot = rel % RING,
    // rel = input row - (yfirst - 1)
    const float* src = a.in0 + (size_t)n * C * a.hin * a.win;
    const size_t cplane = (size_t)a.hin * a.win;
    f32x4 sv[8];
    bool s_in = false, s_act = false;
    int s_dst = 0;
    auto stage_load = [&](int rel0, int nrows) __attribute__((always_inline)) {
        s_act = tid < NG * nrows * K::QPR;
        const int item = s_act ? tid : 0;
        const int cg = item / (nrows * K::QPR), rq = item - cg * (nrows * K::QPR);
        const int r = rq / K::QPR, q = rq - r * K::QPR;
        const int rel = rel0 + r;
        const int gy = yfirst - 1 + rel, gx = x0 - 4 + 4 * q;
        s_in = gy >= 0 && gy < a.hin && gx >= 0 && gx < a.win;               // a piece is inside or outside as a whole (win % 4 == 0)
        const float* sp = src + ((size_t)(cg * 8) * a.hin + (s_in ? gy : 0)) * a.win + (s_in ? gx : 0);
#pragma unroll
        for (int e = 0; e < 8; ++e) sv[e] = *reinterpret_cast<const f32x4*>(sp + e * cplane);
        s_dst = (cg * RING + rel % RING) * COLS + 4 * q;                  // + piece * NG * PLANE + column
    };
    auto stage_convert = [&](int k) __attribute__((always_inline)) {     // column k of the item: 8 channels -> three 16-byte entries
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = s_in ? sv[e][k] : 0.f;
        u32x4 p0, p1, p2;
        split8p(x, p0, p1, p2);
        if (s_act) {
            lds[s_dst + k] = p0;
            lds[NG * PLANE + s_dst + k] = p1;
            lds[2 * NG * PLANE + s_dst + k] = p2;
        }
    };

    stage_load(0, K::FIRST_ROWS);
#pragma unroll
    for (int k = 0; k < 4; ++k) stage_convert(k);
    __syncthreads();

    // bias: the lane's cout is (qbase + q) * 16 + m
    float bias[NCO];
#pragma unroll
    for (int q = 0; q < NCO; ++q) bias[q] = a.bias[(qbase + q) * 16 + m];

    const int lane_e = kg * PLANE + m + 3;                                 // entry index of (group kg, pixel m, column x0 - 1)
    const int hw = H * W;
    float* dst = a.out + (size_t)n * C * hw;
    float psum[2][NCO];                                                    // pooling sums of the rows gathered so far (C = 64: over 4 bands)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < NCO; ++q) psum[p][q] = 0.f;

    for (int b = 0; b < nb; ++b) {
        const int relb = b * BROWS;                                         // rel of the band's first input row (output row - 1)
        // the next band's new rows: loads in flight now, converted in the middle of this band's k-loop
        const bool more = b + 1 < nb;
        if (more) stage_load(relb + BROWS + 2, BROWS);
        // A-fragment addresses of the three filter rows
        unsigned abase[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) abase[ky] = (unsigned)(lane_e + ((relb + wrow + ky) % RING) * COLS) * 16u;
        const char* lb = reinterpret_cast<const char*>(lds);

        f32x4 acc[2][NCO];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < NCO; ++q) acc[p][q] = f32x4{bias[q], bias[q], bias[q], bias[q]};

        auto read_a = [&](int s, u32x4 (&av)[2][3]) __attribute__((always_inline)) {
            const int t = s / K::CHUNKS, c = s % K::CHUNKS;
            const int ky = t / 3, kx = t % 3;
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
                    av[p][pc] = *reinterpret_cast<const u32x4*>(lb + abase[ky] + ((pc * NG + c * 4) * PLANE + kx + 16 * p) * 16);
        };
        u32x4 av[2][2][3];
        read_a(0, av[0]);
#pragma unroll
        for (int s = 0; s < K::KSTEPS; ++s) {
            const int cur = s & 1;
            if (s + 1 < K::KSTEPS) read_a(s + 1, av[cur ^ 1]);
            // the staged rows: one column (a quarter of the item) per k-step from the second on
            constexpr int CONV0 = 1;
            if (more && s >= CONV0 && s < CONV0 + 4) stage_convert(s - CONV0);
#pragma unroll
            for (int i = 0; i < 6; ++i) {                                    // small terms first
                constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int q = 0; q < NCO; ++q)
                        acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(av[cur][p][PA[i]]), as_bf(wv[s][q][PB[i]]), acc[p][q], 0, 0, 0);
            }
        }

        // ---- epilogue: LeakyReLU, 16-byte NCHW stores (lane = cout m of group q, pixels x0 + 16 p + 4 kg .. + 3), pooling sums
        const int y = yfirst + relb + wrow;
        const bool yin = y < H;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int x = x0 + 16 * p + 4 * kg;
#pragma unroll
            for (int q = 0; q < NCO; ++q) {
                f32x4 v = acc[p][q];
                if (a.act) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.1f * v[j]);
                }
                const bool in = yin && x < W;                                 // widths are multiples of 4: a quad is in or out
                if (KEEP && in) *reinterpret_cast<f32x4*>(dst + ((size_t)((qbase + q) * 16 + m) * H + y) * W + x) = v;
                if (POOL) psum[p][q] += in ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f;
            }
        }
        if constexpr (POOL) {
            // a partial sum covers PROWS rows x POOLK columns: C = 32 a band (rows = waves), C = 64 four bands (one row each)
            const bool flush = C == 32 || ((b & 3) == 3) || b + 1 == nb;
            if (flush) {
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int q = 0; q < NCO; ++q) {
                        red[((wave * 2 + p) * NCO + q) * 64 + lane] = psum[p][q];
                        psum[p][q] = 0.f;
                    }
            }
            __syncthreads();
            if (flush) {
                // outputs of the block: C = 32: 32 couts x 2 windows; C = 64: 64 couts x 4 windows (window = 8 pixels = two lane groups)
                constexpr int NOUT = C * K::NWX;
                if (tid < NOUT) {
                    const int co = tid / K::NWX, wx = tid - co * K::NWX;
                    float s = 0.f;
                    if (C == 32) {
                        const int q = co >> 4, mm = co & 15;                  // window wx = pixel tile wx: the four lane groups, the four waves (rows)
#pragma unroll
                        for (int wv_ = 0; wv_ < 4; ++wv_)
#pragma unroll
                            for (int g = 0; g < 4; ++g) s += red[((wv_ * 2 + wx) * NCO + q) * 64 + g * 16 + mm];
                    } else {
                        const int wv_ = co >> 4, mm = co & 15;                // cout group = wave; window wx = pixel tile wx / 2, lane groups 2 (wx % 2) + {0, 1}
                        const int p = wx >> 1, g0 = (wx & 1) * 2;
                        s = red[((wv_ * 2 + p) * NCO) * 64 + g0 * 16 + mm] + red[((wv_ * 2 + p) * NCO) * 64 + (g0 + 1) * 16 + mm];
                    }
                    const int prow = (H + K::PROWS - 1) / K::PROWS;
                    const int py = (yfirst + relb) / K::PROWS;
                    if (py < prow && x0 + wx * K::POOLK < W)
                        a.pool_partial[(((size_t)n * C + co) * prow + py) * (strips * K::NWX) + strip * K::NWX + wx] = s;
                }
            }
        }
        __syncthreads();                                                       // the band's rows are read, the next band's rows are written
    }
}

// OIHW fp32 weights -> pre-split B fragments: u32x4 index ((s * COG + cog) * 3 + piece) * 64 + lane, s = tap * CHUNKS + chunk; dword d of
// lane (cout = cog * 16 + lane % 16, channels chunk * 32 + 8 (lane / 16) + 2 d (+ 1)) holds the pieces in its low (high) half
__global__ void bx3p_wt_kernel(const float* __restrict__ w, int c, unsigned* __restrict__ out, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int chunks = c / 32, cog_n = c / 16;
    const int d = i & 3, lane = (i >> 2) & 63;
    int rest = i >> 8;
    const int cog = rest % cog_n; rest /= cog_n;
    const int s = rest;
    const int t = s / chunks, ch = s % chunks;
    const int co = cog * 16 + (lane & 15), ci = ch * 32 + 8 * (lane >> 4) + 2 * d;
    unsigned lo[3], hi[3];
    for (int h = 0; h < 2; ++h) {
        const float x = w[((size_t)co * c + ci + h) * 9 + t];
        const float x0 = __uint_as_float(__float_as_uint(x) & 0xffff0000u), r = x - x0;
        const float x1 = __uint_as_float(__float_as_uint(r) & 0xffff0000u), q = r - x1;
        unsigned* dd = h ? hi : lo;
        dd[0] = __float_as_uint(x0) >> 16; dd[1] = __float_as_uint(x1) >> 16; dd[2] = __float_as_uint(q) >> 16;
    }
    for (int p = 0; p < 3; ++p) out[((((size_t)s * cog_n + cog) * 3 + p) * 64 + lane) * 4 + d] = (hi[p] << 16) | lo[p];
}

template <int C>
int bx3p_launch_t(const EncConvArgs& a0, hipStream_t stream) {
    using K = PCfg<C>;
    EncConvArgs a = a0;
    a.tiles_x = ceil_div(a.wout, 32);
    const int nbands = ceil_div(a.hout, K::BROWS);
    // bands per block: the segments are multiples of the pooling partials' 4 rows; one block per CU at most, fewer and longer ones when
    // several frames share the chip (EncConvArgs::blocks_per_xcd > 0) - a block's prologue loads 55 KB of weights per wave
    static const int env_seg = [] { const char* e = getenv("EEM_BX3P_SEG"); return e ? atoi(e) : 0; }();
    int seg = (C == 32 ? 1 : 4);
    const int target = a.blocks_per_xcd > 0 ? a.blocks_per_xcd * 8 : 256;
    while (a.tiles_x * ceil_div(nbands, seg) * a.nimg > target) seg += (C == 32 ? 1 : 4);
    if (env_seg > 0) seg = env_seg;
    a.tiles_y = ceil_div(nbands, seg);
    const int blocks = a.tiles_x * a.tiles_y * a.nimg;
    dim3 grid((unsigned)ceil_div(blocks, 8) * 8);
    EEM_NOTE_GRID(blocks, 256);
    EEM_NOTE_PIPE(1);
    const u32x4* wq = reinterpret_cast<const u32x4*>(a.wbx3);
    if (a.pool_partial != nullptr && a.no_store)
        hipLaunchKernelGGL((bx3p_kernel<C, true, false>), grid, dim3(256), 0, stream, a, wq, seg, a.tiles_y);
    else if (a.pool_partial != nullptr)
        hipLaunchKernelGGL((bx3p_kernel<C, true, true>), grid, dim3(256), 0, stream, a, wq, seg, a.tiles_y);
    else
        hipLaunchKernelGGL((bx3p_kernel<C, false, true>), grid, dim3(256), 0, stream, a, wq, seg, a.tiles_y);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

}  // namespace

bool bx3p_shape(int cin, int cout, int stride) { return stride == 1 && cin == cout && (cin == 32 || cin == 64); }

size_t bx3p_packed_floats(int c) { return (size_t)9 * (c / 32) * (c / 16) * 3 * 64 * 4; }

int bx3p_transform_launch(const float* w, int c, float* packed, hipStream_t stream) {
    const int total = 9 * (c / 32) * (c / 16) * 64 * 4;
    hipLaunchKernelGGL(bx3p_wt_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, w, c, reinterpret_cast<unsigned*>(packed), total);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

// the pooling protocol of this kernel: partial sums of 4 rows x POOLK columns, 32-pixel strips
void bx3p_tile(int c, int* th, int* tw, int* poolk) { *th = 4; *tw = 32; *poolk = c == 32 ? 16 : 8; }

bool bx3p_supported(int cin, int cout, int stride, const EncConvArgs& a) {
    return bx3p_shape(cin, cout, stride) && a.wbx3 && a.gate == nullptr && a.res == nullptr && (a.win & 3) == 0 && (a.act == 0 || a.act == 1) &&
           (((uintptr_t)a.in0) & 15) == 0 && (((uintptr_t)a.out) & 15) == 0 && (size_t)cin * a.hin * a.win * 4 < (1u << 31) &&
           (a.pool_partial == nullptr || a.pool_k == (cin == 32 ? 16 : 8));
}

int bx3p_launch(int c, const EncConvArgs& a, hipStream_t stream) {
    return c == 32 ? bx3p_launch_t<32>(a, stream) : bx3p_launch_t<64>(a, stream);
}
