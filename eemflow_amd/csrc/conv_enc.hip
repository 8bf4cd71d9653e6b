// Encoder convolutions of EEMFlow (reference: model/EEMFlow/EEMFlow.py:26-30,75-82,135-140) as
// implicit GEMM on the gfx950 fp32 matrix cores.
//
//   D[cout][pixel] = sum_k  A[cout][k] * B[k][pixel],   k = (tap, cin)
//
// * A (weights) is pre-packed on the host into MFMA fragment order and streamed from L2
//   straight into VGPRs (16 B per lane = 4 k-steps); it never touches LDS.
// * B (input patch) comes from an LDS-staged input tile [cin][rows][cols] with the 1-pixel
//   halo; lane (pixel j, k-slot h) reads plane cin+h at column j -> consecutive lanes hit
//   consecutive banks, all tap/channel displacements are ds_read immediates.
// * pixels sit on the MFMA N dimension, so every accumulator register is a run of
//   consecutive x for one output channel: NCHW stores are 128-B (64-B for cout=16) segments.
// * cout >= 32 uses v_mfma_f32_32x32x2_f32, cout == 16 uses v_mfma_f32_16x16x4_f32 (same
//   64 FLOP/clk/SIMD, no wasted rows).  Exact fp32 (k-ordered fma chain).
// * the first layer folds the reference's replicate padding (utils/image_utils.py:139-140)
//   into the tile loader; bias is the accumulator's initial value; LeakyReLU(0.1) in the
//   epilogue.
#include <stdlib.h>

#include "common.h"

namespace {

template <int CIN, int COUT, int STRIDE, int TH, int TWT>
struct Cfg {
    static constexpr bool M16 = (COUT == 16);
    static constexpr int NPIX = M16 ? 16 : 32;           // pixels per MFMA tile
    static constexpr int KPS = M16 ? 4 : 2;              // k consumed per MFMA
    static constexpr int MT = M16 ? 1 : COUT / 32;       // cout tiles
    static constexpr bool FLATK = (CIN % KPS) != 0;      // first layer: k = cin*9+tap, padded
    static constexpr int K = CIN * 9;
    static constexpr int KSTEPS = FLATK ? ((K + KPS - 1) / KPS + 3) / 4 * 4 : K / KPS;
    static constexpr int TW = TWT * NPIX;
    static constexpr int IN_ROWS = (TH - 1) * STRIDE + 3;
    static constexpr int IN_COLS = (TW - 1) * STRIDE + 3;
    static constexpr int PLANE = IN_ROWS * IN_COLS;
    static constexpr int TILE = CIN * PLANE;
    static constexpr int UNITS = TH * TWT;               // (row, col-tile) units per block
    static constexpr int UPW = UNITS / 4;                // units per wave
    static constexpr int ACC = M16 ? 4 : 16;
    static_assert(UNITS % 4 == 0, "tile units must split over 4 waves");
    static_assert(KSTEPS % 4 == 0, "k-steps are loaded four at a time");
};

template <bool M16> struct AccT { using type = f32x16; };
template <> struct AccT<true> { using type = f32x4; };

template <int CIN, int COUT, int STRIDE, int TH, int TWT, bool PADIN>
__global__ __launch_bounds__(256) void enc_conv_kernel(EncConvArgs a) {
    using C = Cfg<CIN, COUT, STRIDE, TH, TWT>;
    using acc_t = typename AccT<C::M16>::type;
    __shared__ float tile[(C::TILE + 63) / 64 * 64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const unsigned lid = xcd_logical_block(blockIdx.x, gridDim.x);
    if (lid >= (unsigned)(a.tiles_x * a.tiles_y * a.nimg)) return;
    const int bx = lid % a.tiles_x, by = (lid / a.tiles_x) % a.tiles_y;
    const int n = lid / (a.tiles_x * a.tiles_y);
    const int oy0 = by * TH;
    const int ox0 = bx * C::TW;

    // ---- stage the input tile (zero outside the conv input; replicate inside the pad band) with 4-byte
    // LDS-DMA copies: each lane's source address is clamped (replicate) or the zero page (conv padding),
    // all of a wave's copies are in flight at once and no VGPR holds the data.
    {
        const float* src;
        if (PADIN) {
            const float* in0 = a.io ? (const float*)a.io[0] : a.in0;     // cached graph: buffers through the io table
            const float* in1 = a.io ? (const float*)a.io[1] : a.in1;
            if (a.io_frames)                                             // per-frame buffers (eemflow_forward_many)
                src = (const float*)(n < a.nimg0 ? a.io[3 * n] : a.io[3 * (n - a.nimg0) + 1]);
            else
                src = (n < a.nimg0) ? in0 + (size_t)n * CIN * a.hraw * a.wraw
                                    : in1 + (size_t)(n - a.nimg0) * CIN * a.hraw * a.wraw;
        }
        else
            src = a.in0 + (size_t)n * CIN * a.hin * a.win;
        const int gy0 = oy0 * STRIDE - 1;
        const int gx0 = ox0 * STRIDE - 1;
#pragma unroll 4
        for (int e0 = wave * 64; e0 < C::TILE; e0 += 256) {
            const int e = e0 + lane;
            const int c = e / C::PLANE;
            const int rem = e - c * C::PLANE;
            const int ry = rem / C::IN_COLS;
            const int rx = rem - ry * C::IN_COLS;
            const int gy = gy0 + ry, gx = gx0 + rx;
            const bool ok = e < C::TILE && gy >= 0 && gy < a.hin && gx >= 0 && gx < a.win;
            const float* g;
            if (PADIN) {
                const int sy = min(max(gy - a.pad_top, 0), a.hraw - 1);
                const int sx = min(max(gx - a.pad_left, 0), a.wraw - 1);
                g = src + ((size_t)c * a.hraw + sy) * a.wraw + sx;
            } else {
                g = src + ((size_t)c * a.hin + gy) * a.win + gx;
            }
            g = ok ? g : a.zero_page;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(tile + e0), 4, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    // ---- per-lane constants
    const int j = lane & (C::NPIX - 1);        // pixel within the MFMA tile
    const int g = lane / C::NPIX;              // k-slot (0..KPS-1)

    acc_t acc[C::UPW][C::MT];
#pragma unroll
    for (int u = 0; u < C::UPW; ++u)
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
            for (int r = 0; r < C::ACC; ++r) {
                const int co = C::M16 ? (g * 4 + r) : (mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g);
                acc[u][mt][r] = a.bias[co];
            }

    // LDS word offset of this lane's pixel for unit u (row, col-tile), tap (0,0), channel slot g
    int ubase[C::UPW];
#pragma unroll
    for (int u = 0; u < C::UPW; ++u) {
        const int unit = wave * C::UPW + u;
        const int row = unit / TWT, ct = unit % TWT;
        ubase[u] = row * STRIDE * C::IN_COLS + (ct * C::NPIX + j) * STRIDE;
    }

    const f32x4* wp = reinterpret_cast<const f32x4*>(a.wpk) + lane;

    if constexpr (C::FLATK) {
        // k = 4*s + g  ->  (cin = k / 9, tap = k % 9); weights beyond K are packed as zeros
        int koff[C::KSTEPS];
#pragma unroll
        for (int s = 0; s < C::KSTEPS; ++s) {
            int k = s * C::KPS + g;
            k = k < C::K ? k : 0;
            const int c = k / 9, t = k - c * 9;
            koff[s] = c * C::PLANE + (t / 3) * C::IN_COLS + (t % 3);
        }
#pragma unroll
        for (int s4 = 0; s4 < C::KSTEPS / 4; ++s4) {
            const f32x4 av = wp[s4 * 64];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = s4 * 4 + q;
#pragma unroll
                for (int u = 0; u < C::UPW; ++u) {
                    const float b = tile[ubase[u] + koff[s]];
                    acc[u][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], b, acc[u][0], 0, 0, 0);
                }
            }
        }
    } else {
        constexpr int CG = CIN / C::KPS;       // channel groups per tap
        const float* tl = tile + g * C::PLANE;
#pragma unroll
        for (int s4 = 0; s4 < C::KSTEPS / 4; ++s4) {
            f32x4 av[C::MT];
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt) av[mt] = wp[(mt * (C::KSTEPS / 4) + s4) * 64];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = s4 * 4 + q;
                const int t = s / CG, cg = s % CG;
                const int off = cg * C::KPS * C::PLANE + (t / 3) * C::IN_COLS + (t % 3);
#pragma unroll
                for (int u = 0; u < C::UPW; ++u) {
                    const float b = tl[ubase[u] + off];
#pragma unroll
                    for (int mt = 0; mt < C::MT; ++mt) {
                        if constexpr (C::M16)
                            acc[u][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][q], b, acc[u][mt], 0, 0, 0);
                        else
                            acc[u][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt][q], b, acc[u][mt], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue: LeakyReLU + NCHW store (each register = consecutive x of one channel)
    float* dst = a.out + (size_t)n * COUT * a.hout * a.wout;
#pragma unroll
    for (int u = 0; u < C::UPW; ++u) {
        const int unit = wave * C::UPW + u;
        const int row = unit / TWT, ct = unit % TWT;
        const int oy = oy0 + row;
        const int ox = ox0 + ct * C::NPIX + j;
        if (oy < a.hout && ox < a.wout) {
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                for (int r = 0; r < C::ACC; ++r) {
                    const int co = C::M16 ? (g * 4 + r) : (mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g);
                    float v = acc[u][mt][r];
                    if (a.act) v = v > 0.f ? v : 0.1f * v;
                    const size_t o = ((size_t)co * a.hout + oy) * a.wout + ox;
                    if (a.gate) v *= a.gate[(size_t)n * COUT * a.hout * a.wout + o] > 0.f ? 1.f : 0.1f;
                    dst[o] = v;
                }
        }
    }
}

template <int CIN, int COUT, int STRIDE, int TH, int TWT, bool PADIN>
int launch(const EncConvArgs& a0, hipStream_t stream) {
    using C = Cfg<CIN, COUT, STRIDE, TH, TWT>;
    EncConvArgs a = a0;
    a.tiles_x = ceil_div(a.wout, C::TW);
    a.tiles_y = ceil_div(a.hout, TH);
    dim3 grid((unsigned)ceil_div(a.tiles_x * a.tiles_y * a.nimg, 8) * 8);
    hipLaunchKernelGGL((enc_conv_kernel<CIN, COUT, STRIDE, TH, TWT, PADIN>), grid, dim3(256), 0, stream, a);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

}  // namespace

// ------------------------------------------------------------------------------- host side
static void enc_kinfo(int cin, int cout, int* kps, int* ksteps, int* mt, bool* flatk) {
    const bool m16 = (cout == 16);
    *kps = m16 ? 4 : 2;
    *mt = m16 ? 1 : cout / 32;
    *flatk = (cin % *kps) != 0;
    const int K = cin * 9;
    *ksteps = *flatk ? ((K + *kps - 1) / *kps + 3) / 4 * 4 : K / *kps;
}

size_t enc_packed_floats(int cin, int cout) {
    int kps, ksteps, mt;
    bool flatk;
    enc_kinfo(cin, cout, &kps, &ksteps, &mt, &flatk);
    return (size_t)mt * ksteps * 64;
}

void enc_pack_weights(const float* w, int cin, int cout, float* packed) {
    int kps, ksteps, mt;
    bool flatk;
    enc_kinfo(cin, cout, &kps, &ksteps, &mt, &flatk);
    const int npix = (cout == 16) ? 16 : 32;
    const int cg = flatk ? 1 : cin / kps;
    for (int m = 0; m < mt; ++m)
        for (int s = 0; s < ksteps; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int co = m * 32 + (lane & (npix - 1));
                const int g = lane / npix;
                float v = 0.f;
                if (flatk) {
                    const int k = s * kps + g;
                    if (k < cin * 9) v = w[(size_t)co * cin * 9 + k];   // OIHW flat: cin*9 + tap
                } else {
                    const int t = s / cg, c = (s % cg) * kps + g;
                    v = w[((size_t)co * cin + c) * 9 + t];
                }
                packed[((size_t)(m * (ksteps / 4) + s / 4) * 64 + lane) * 4 + (s & 3)] = v;
            }
}

int enc_conv_launch(int cin, int cout, int stride, const EncConvArgs& a, hipStream_t stream) {
    if (!a.zero_page) {
        eem_set_error("enc_conv_launch: zero_page is NULL");
        return EEM_ERR_ARG;
    }
    if (a.in_norm) {                                     // raw voxel grids + normalisation records: only conv_enc1.hip reads that form
        if (cin == 5 && cout == 16 && stride == 2 && enc1_supported(a)) return enc1_launch(a, stream);
        eem_set_error("deferred input normalisation needs the 5-bin first layer on 16-byte-aligned volumes whose width is a multiple of 4 "
                      "and that pad on the bottom only (conv_enc1.hip); normalise in the voxelizer instead");
        return EEM_ERR_ARG;
    }
    if (a.ws2r && s2r_supported(cin, cout, stride, a)) return s2r_launch(cin, a, stream);
    if (a.wbx3 && bx3_supported(cin, cout, stride, a)) return bx3_launch(cin, stride, a, stream);
    if (a.wwino && wino_supported(cin, cout, stride, a.win)) return wino_launch(cin, a, stream);
    if (a.wpk2 && enc2_supported(cin, cout, stride, a.win)) return enc_conv2_launch(cin, cout, stride, a, stream);
    //                                   CIN COUT S  TH TWT PADIN
    if (cin == 5 && cout == 16 && stride == 2 && enc1_supported(a) && !getenv("EEM_NO_ENC1")) return enc1_launch(a, stream);
    if (cin == 5 && cout == 16 && stride == 2) return launch<5, 16, 2, 8, 4, true>(a, stream);
    if (cin == 16 && cout == 16 && stride == 1) return launch<16, 16, 1, 8, 4, false>(a, stream);
    if (cin == 16 && cout == 32 && stride == 2) return launch<16, 32, 2, 4, 1, false>(a, stream);
    if (cin == 32 && cout == 32 && stride == 1) return launch<32, 32, 1, 4, 2, false>(a, stream);
    if (cin == 32 && cout == 64 && stride == 2) return launch<32, 64, 2, 4, 1, false>(a, stream);
    if (cin == 64 && cout == 64 && stride == 1) return launch<64, 64, 1, 4, 1, false>(a, stream);
    eem_set_error("enc_conv_launch: unsupported layer cin=%d cout=%d stride=%d", cin, cout, stride);
    return EEM_ERR_ARG;
}
