// First encoder layer, pconv1_1 (EEMFlow.py:75,26-30: 3x3 stride-2 conv 5 -> 16 + LeakyReLU) fused with the
// reference's replicate padding (utils/image_utils.py:139-140, EEMFlow.py:134), for raw event volumes whose rows
// are 16-byte aligned and that need no horizontal padding (width % 64 == 0, e.g. HREM 1280x720).  Other shapes
// keep the generic kernel of conv_enc.hip.
//
// The layer is HBM-bound (0.71 GFLOP for 68 MB at 1280x720), so the kernel is built around the copy:
//   * persistent blocks (one per CU) walk their tiles over a 3-deep LDS ring: tiles i+1 and i+2 stream in by
//     16-byte LDS-DMA pieces (global_load_lds_dwordx4, 1 KiB per wave-instruction, 6 per wave and tile instead of
//     the generic kernel's 43 four-byte ones) while tile i feeds the MFMAs; vertical replicate padding is a row
//     clamp on the piece's source address, the conv's own zero padding a redirect to a zero page;
//   * K = 45 (cin*9) padded to 48: 12 x v_mfma_f32_16x16x4_f32 per 16 output pixels, weights (3 registers per
//     lane) stationary;
//   * bias is the accumulator's initial value, LeakyReLU in the epilogue, NCHW stores.
#include "common.h"

namespace {

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int TH, int TWT, int WAVES>
struct E1Cfg {
    static constexpr int CIN = 5, COUT = 16;
    static constexpr int TW = TWT * 16;
    static constexpr int IN_ROWS = 2 * TH + 1;
    static constexpr int ROWP = 2 * TW + 4;              // staged row: x = 2*ox0-4 .. 2*ox0+2*TW-1
    static constexpr int PPR = ROWP / 4;
    static constexpr int PLANE = IN_ROWS * ROWP;
    static constexpr int PIECES = CIN * IN_ROWS * PPR;
    static constexpr int NB = (PIECES + 63) / 64;
    static constexpr int NI = (NB + WAVES - 1) / WAVES;
    static constexpr int STAGE = NI * WAVES * 256;
    static constexpr int UPW = TH * TWT / WAVES;         // (row, 16-pixel tile) units per wave
    static constexpr int KSTEPS = 12;                    // ceil(45 / 4)
    static_assert((TH * TWT) % WAVES == 0, "units split over the waves");
    static constexpr int NST = 3;                        // LDS ring: two tiles in flight behind the one being computed
    static_assert(NST * STAGE * 4 <= 160 * 1024, "LDS budget");
};

// NORM: the event volumes are RAW voxel grids followed by their normalisation record {mean, sd, scale, any} (eemflow_voxelize with
// normalize = 2): loader_utils.py:527-535's (v - mean) / sd on the non-zero voxels is applied to every operand as it is read from LDS
// (zero padding and empty voxels stay 0; replicated rows normalise like the rows they copy)
template <int TH, int TWT, int WAVES, bool NORM>
__global__ __launch_bounds__(WAVES * 64) void enc1_kernel(EncConvArgs a) {
    ENC_ARGS_NOW(a);
    using K = E1Cfg<TH, TWT, WAVES>;
    constexpr int NS = K::UPW;                           // stores per wave and tile
    __shared__ __attribute__((aligned(16))) float lds[K::NST * K::STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    // inside a cached graph the event volumes come through the context's io table (scalar loads; see EncConvArgs::io)
    const float* in0 = a.io ? (const float*)a.io[0] : a.in0;
    const float* in1 = a.io ? (const float*)a.io[1] : a.in1;

    const TileRange tr_ = block_tile_range(a.tiles_x * a.tiles_y * a.nimg, blockIdx.x, gridDim.x);
    const int ntile = tr_.count;
    if (ntile == 0) return;
    TileCoord cur = tile_coord(tr_.first, a.tiles_x, a.tiles_y), nxt = cur;      // tile being computed / next to request
    const float* zero_page = a.zero_page;

    // stationary operands: weights (packed by enc_pack_weights: [k-step / 4][lane][4]) and bias
    f32x4 wr[K::KSTEPS / 4];
#pragma unroll
    for (int q = 0; q < K::KSTEPS / 4; ++q) wr[q] = (reinterpret_cast<const f32x4*>(a.wpk) + lane)[q * 64];
    // pixels sit on the MFMA M dimension, couts on N: lane (cout j = lane % 16, g = lane / 16) ends up with 4
    // consecutive pixels 4g..4g+3 of its cout -> one 16-byte store per lane and 16-pixel unit
    float biasv = a.bias[j];

    // ---- DMA plan: piece -> (channel, tile row, 16-byte column piece), fixed per lane and instruction
    int pc[K::NI], pry[K::NI], pq[K::NI];
#pragma unroll
    for (int k = 0; k < K::NI; ++k) {
        int p = (wave + k * WAVES) * 64 + lane;
        p = p < K::PIECES ? p : K::PIECES - 1;
        pc[k] = p / (K::IN_ROWS * K::PPR);
        const int rem = p - pc[k] * (K::IN_ROWS * K::PPR);
        pry[k] = rem / K::PPR;
        pq[k] = (rem - pry[k] * K::PPR) * 4;
    }
    // eemflow_forward_many: every image is its own buffer, named by the io table's per-frame triples; a block's tiles are a contiguous
    // range, so the image changes once or twice per block and the (dependent, scalar) table read happens only then
    int src_n = -1;
    const float* src_many = nullptr;
    auto issue = [&](int it, const TileCoord& tc) {
        const int bx = tc.bx, by = tc.by, n = tc.n;
        const int gy0 = by * TH * 2 - 1, gx0 = bx * K::TW * 2 - 4;          // padded-image coordinates
        const float* src;
        if (a.io_frames) {
            if (n != src_n) {
                src_many = (const float*)(n < a.nimg0 ? a.io[3 * n] : a.io[3 * (n - a.nimg0) + 1]);
                src_n = n;
            }
            src = src_many;
        } else {
            const size_t img = (size_t)K::CIN * a.hraw * a.wraw + (NORM ? 4 : 0);     // (a batch of raw grids carries a record behind each)
            src = (n < a.nimg0) ? in0 + (size_t)n * img : in1 + (size_t)(n - a.nimg0) * img;
        }
        float* sbase = lds + (it % K::NST) * K::STAGE;
#pragma unroll
        for (int k = 0; k < K::NI; ++k) {
            const int gy = gy0 + pry[k], gx = gx0 + pq[k];
            const bool ok = gy >= 0 && gy < a.hin && gx >= 0 && gx < a.win;  // else: the conv's zero padding
            const int sy = min(max(gy - a.pad_top, 0), a.hraw - 1);          // replicate rows of the pad band
            const float* gp = ok ? src + ((size_t)pc[k] * a.hraw + sy) * a.wraw + gx : zero_page;
            __builtin_amdgcn_global_load_lds(GLB_PTR(gp), LDS_PTR(sbase + (wave + k * WAVES) * 256), 16, 0, 0);
        }
    };

    // LDS offset of tap k = 4s + g (cin = k / 9, ky, kx) for this lane's k-slot; k >= 45 reads tap 0 (weight 0)
    int koff[K::KSTEPS];
#pragma unroll
    for (int s = 0; s < K::KSTEPS; ++s) {
        int k = s * 4 + g;
        k = k < 45 ? k : 0;
        const int c = k / 9, t = k - c * 9;
        koff[s] = c * K::PLANE + (t / 3) * K::ROWP + (t % 3);
    }
    int ubase[K::UPW];
#pragma unroll
    for (int u = 0; u < K::UPW; ++u) {
        const int unit = wave * K::UPW + u;
        const int row = unit / TWT, ct = unit % TWT;
        ubase[u] = row * 2 * K::ROWP + (ct * 16 + j) * 2 + 3;
    }

    // the normalisation record of the image being multiplied (scalar loads when the image changes: once or twice per block)
    int rec_n = -1;
    float n_mean = 0.f, n_inv = 1.f;
    bool n_on = false;
    auto load_record = [&](int n) {
        const float* img;
        if (a.io_frames) img = (const float*)(n < a.nimg0 ? a.io[3 * n] : a.io[3 * (n - a.nimg0) + 1]);
        else img = (n < a.nimg0) ? in0 + (size_t)n * (K::CIN * a.hraw * a.wraw + 4) : in1 + (size_t)(n - a.nimg0) * (K::CIN * a.hraw * a.wraw + 4);
        const float* rec = img + (size_t)K::CIN * a.hraw * a.wraw;
        n_mean = rec[0];
        n_inv = rec[2] != 0.f ? 1.f / rec[1] : 1.f;
        n_on = rec[3] != 0.f;
        rec_n = n;
    };

    issue(0, nxt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < K::KSTEPS / 4; ++q) asm volatile("" : "+v"(wr[q]));
    asm volatile("" : "+v"(biasv));
    if (ntile > 1) {
        tile_advance(nxt, a.tiles_x, a.tiles_y);
        issue(1, nxt);
    }

#pragma unroll 1
    for (int it = 0; it < ntile; ++it) {
        // tile `it` must have landed.  Younger operations that may stay in flight (vmcnt retires in order): the
        // stores of tile it-2, the DMA of tile it+1 (if there is one) and the stores of tile it-1
        if (it > 0) {
            const bool ahead = it + 1 < ntile;
            if (it == 1) {
                if (ahead) wait_vmcnt<K::NI + NS>(); else wait_vmcnt<NS>();
            } else {
                if (ahead) wait_vmcnt<K::NI + 2 * NS>(); else wait_vmcnt<2 * NS>();
            }
        }
        __builtin_amdgcn_s_barrier();
        if (it + 2 < ntile) {                        // its stage was last read by tile it-1: free since the barrier
            tile_advance(nxt, a.tiles_x, a.tiles_y);
            issue(it + 2, nxt);
        }

        const int bx = cur.bx, by = cur.by, n = cur.n;
        const float* tb = lds + (it % K::NST) * K::STAGE;
        if constexpr (NORM) {
            if (n != rec_n) load_record(n);
        }

        f32x4 acc[K::UPW];
#pragma unroll
        for (int u = 0; u < K::UPW; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[u][r] = biasv;
#pragma unroll
        for (int s = 0; s < K::KSTEPS; ++s)
#pragma unroll
            for (int u = 0; u < K::UPW; ++u) {
                float x = tb[ubase[u] + koff[s]];
                if constexpr (NORM) x = (n_on && x != 0.f) ? (x - n_mean) * n_inv : x;
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, wr[s >> 2][s & 3], acc[u], 0, 0, 0);
            }

        const int hw = a.hout * a.wout;
        float* dst = a.out + (size_t)n * K::COUT * hw;
#pragma unroll
        for (int u = 0; u < K::UPW; ++u) {
            const int unit = wave * K::UPW + u;
            const int row = unit / TWT, ct = unit % TWT;
            const int oy = by * TH + row, ox = bx * K::TW + ct * 16 + 4 * g;
            const bool inside = oy < a.hout && ox < a.wout;                  // wout % 4 == 0: all four or none
            f32x4 v = acc[u];
            if (a.act) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.1f * v[r]);
            }
            float* p = inside ? dst + (j * a.hout + oy) * a.wout + ox : a.trash + lane * 4;   // uniform store count
            *reinterpret_cast<f32x4*>(p) = v;
        }
        tile_advance(cur, a.tiles_x, a.tiles_y);
    }
}

}  // namespace

bool enc1_supported(const EncConvArgs& a) {
    return a.pad_left == 0 && a.win == a.wraw && (a.wraw & 3) == 0 && ((uintptr_t)a.in0 & 15) == 0 &&
           ((uintptr_t)a.in1 & 15) == 0 && a.gate == nullptr && (a.wout & 3) == 0 && ((uintptr_t)a.out & 15) == 0;
}

int enc1_launch(const EncConvArgs& a0, hipStream_t stream) {
    constexpr int TH = 8, TWT = 4, WAVES = 8;
    EncConvArgs a = a0;
    a.tiles_x = ceil_div(a.wout, TWT * 16);
    a.tiles_y = ceil_div(a.hout, TH);
    const int T = a.tiles_x * a.tiles_y * a.nimg;
    int per_xcd = ceil_div(T, 8);
    static const int env_cap = enc_blocks_per_xcd("E1", 0);     // tuning override
    const int cap = env_cap > 0 ? env_cap : (a.blocks_per_xcd > 0 ? a.blocks_per_xcd : 32);   // default: one resident block per CU
    if (per_xcd > cap) per_xcd = cap;
    EEM_NOTE_GRID(per_xcd * 8, WAVES * 64);
    if (a.in_norm) hipLaunchKernelGGL((enc1_kernel<TH, TWT, WAVES, true>), dim3(per_xcd * 8), dim3(WAVES * 64), 0, stream, a);
    else hipLaunchKernelGGL((enc1_kernel<TH, TWT, WAVES, false>), dim3(per_xcd * 8), dim3(WAVES * 64), 0, stream, a);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}
