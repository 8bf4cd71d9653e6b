// The stride-2 encoder layer pconv2_1 (16 -> 32, EEMFlow.py:77) as many light blocks (pconv3_1, 32 -> 64, can run here too:
// EEM_S2W_64=1; it is 0.6 us slower than on conv_enc2.hip and stays there).
//
// conv_enc2.hip runs these layers as blocks of one 4 x 32-pixel tile that stream the input in 8-channel chunks through two LDS
// stages together with the chunk's weight fragments: 64 KB of LDS and 169 VGPRs allow two blocks per CU, a barrier per chunk,
// and pconv2_1's 900 blocks need two rounds on 512 slots.  Here
//   * the weight fragments never touch LDS: a lane's A operand of k-step s is one float of a 16-byte global load (the packed
//     stream of conv_enc2.hip, [chunk][cout tile][k-step / 4][lane][4], read-only and shared by every block: L2 hits), held in
//     registers for the whole tile (16 -> 32: 72 VGPRs) or as a two-chunk ring refilled behind its use (32 -> 64);
//   * the whole input tile (all channels, 9 rows x 68 columns) is requested by LDS-DMA at once: ONE wait, ONE barrier, then
//     CIN * 9 / 2 MFMAs per wave whose B operands are ds_read_b32 with immediate offsets from one base register;
//   * 40 KB of LDS and 105 VGPRs per 4-wave block: four blocks per CU, the 900 tiles are resident at once.
// Same arithmetic order per output as conv_enc2.hip (k ascending: chunk, tap, channel pair): bitwise the same results.
// Measured (1280x720): 18.1 -> 17.3 us.  What bounds the layer, whatever the form (two-stage chunks 18.1, these blocks 17.3, a
// persistent block with two whole-tile stages and stationary weights 17.7 - built, measured, removed): 900 tiles x 4 rows = 3 600
// wave-tiles of 72 MFMAs on 1 024 SIMDs is 3.5 per SIMD, i.e. four on most (8.8 us of matrix pipe at 2.1 GHz against 7.7 balanced),
// in front of it launch + DMA plan + the first bytes' way from HBM (~5 us) and behind it the stores (~1.5 us); with four frames in
// flight those ends overlap other frames' kernels, which is why none of the three forms moves the 4-in-flight rate.
#include "common.h"

namespace {

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int CIN, int COUT>
struct S2Cfg {
    static constexpr int TH = 4, NPIX = 32;
    static constexpr int MT = COUT / 32;                 // cout tiles
    static constexpr int WAVES = TH * MT;                // wave = (row, cout tile)
    static constexpr int IN_ROWS = 2 * (TH - 1) + 3;     // 9
    static constexpr int ROWP = 68;                      // staged floats per row: columns 2*x0 - 4 .. 2*x0 + 63
    static constexpr int PPR = ROWP / 4;                 // 16-byte pieces per row
    static constexpr int PLANE = IN_ROWS * ROWP;         // floats per channel
    static constexpr int PIECES = CIN * IN_ROWS * PPR;
    static constexpr int NI = (PIECES + WAVES * 64 - 1) / (WAVES * 64);   // DMA instructions per wave
    static constexpr int LDS_FLOATS = NI * WAVES * 256;
    static constexpr int NCH = CIN / 8;                  // weight chunks of 8 channels (36 k-steps, 9 float4 per lane)
    static constexpr int RING = NCH <= 2 ? NCH : 2;      // chunks of weights held in registers
    static constexpr int MINB = COUT == 32 ? 4 : 2;      // blocks per CU the register budget is sized for
    static_assert((14 * PLANE + 2 * ROWP + 2) * 4 < 65536, "ds_read immediate range of a 16-channel half");
};

template <int CIN, int COUT>
__global__ __launch_bounds__((S2Cfg<CIN, COUT>::WAVES * 64), (S2Cfg<CIN, COUT>::MINB * S2Cfg<CIN, COUT>::WAVES / 4))
void s2_kernel(EncConvArgs a) {
    using C = S2Cfg<CIN, COUT>;
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    ENC_ARGS_NOW(a);
    const unsigned lid = xcd_logical_block(blockIdx.x, gridDim.x);
    if (lid >= (unsigned)(a.tiles_x * a.tiles_y * a.nimg)) return;
    const int bx = lid % a.tiles_x, by = (lid / a.tiles_x) % a.tiles_y;
    const int n = lid / (a.tiles_x * a.tiles_y);
    const int row = wave % C::TH, mt = wave / C::TH;
    const int j = lane & 31, g = lane >> 5;

    // ---- weights of chunk 0 (and 1): requested first, they land while the DMA plan is computed
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(a.wpk2) + (size_t)mt * 9 * 64 + lane;      // + (ch * MT * 9 + s4) * 64
    f32x4 wv[C::RING][9];
#pragma unroll
    for (int r = 0; r < C::RING; ++r)
#pragma unroll
        for (int s4 = 0; s4 < 9; ++s4) wv[r][s4] = wsrc[(r * C::MT * 9 + s4) * 64];

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

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = a.bias[mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g];

    // B operand of k-step s = (chunk ch, tap t, channel pair cg): channel ch * 8 + cg * 2 + g, input row 2 * row + ky, column 2 * j + kx + 3
    // (one base register per 16 channels keeps every displacement inside the 16-bit immediate)
    const float* bl = lds + g * C::PLANE + 2 * row * C::ROWP + 2 * j + 3;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int ch = 0; ch < C::NCH; ++ch) {
        f32x4(&w)[9] = wv[ch % C::RING];
        const float* blh = bl + (ch / 2) * 16 * C::PLANE;
#pragma unroll
        for (int s4 = 0; s4 < 9; ++s4) {
            float b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = s4 * 4 + q, t = s / 4, cg = s % 4;
                b[q] = blh[((ch % 2) * 8 + cg * 2) * C::PLANE + (t / 3) * C::ROWP + (t % 3)];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[s4][q], b[q], acc, 0, 0, 0);
            // ring: this slot's MFMAs have issued - request the same float4 of the chunk that will use the slot next
            if (ch + C::RING < C::NCH) w[s4] = wsrc[((ch + C::RING) * C::MT * 9 + s4) * 64];
        }
    }

    // ---- epilogue: LeakyReLU, NCHW stores (a wave's 32 lanes of a cout row are 128 consecutive bytes)
    const int oy = oy0 + row, ox = ox0 + j;
    const int hw = a.hout * a.wout;
    float* dst = a.out + (size_t)n * COUT * hw;
    const int co0 = mt * 32 + 4 * g;
    const bool full = oy0 + C::TH <= a.hout && ox0 + C::NPIX <= a.wout;           // block-uniform
    const unsigned lane_bo = (unsigned)((co0 * a.hout + oy) * a.wout + ox) * 4u;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = acc[r];
        if (a.act) v = fmaxf(v, 0.1f * v);
        const int dco = (r & 3) + 8 * (r >> 2);
        if (full) {
            char* rb = reinterpret_cast<char*>(dst) + (size_t)dco * hw * 4;       // scalar base per register, one 32-bit lane offset
            *reinterpret_cast<float*>(rb + lane_bo) = v;
        } else if (oy < a.hout && ox < a.wout) {
            dst[(size_t)(co0 + dco) * hw + oy * a.wout + ox] = v;
        }
    }
}

template <int CIN, int COUT>
int s2_launch_t(const EncConvArgs& a0, hipStream_t stream) {
    using C = S2Cfg<CIN, COUT>;
    EncConvArgs a = a0;
    a.tiles_x = ceil_div(a.wout, C::NPIX);
    a.tiles_y = ceil_div(a.hout, C::TH);
    dim3 grid((unsigned)ceil_div(a.tiles_x * a.tiles_y * a.nimg, 8) * 8);
    EEM_NOTE_GRID(grid.x, C::WAVES * 64);
    hipLaunchKernelGGL((s2_kernel<CIN, COUT>), grid, dim3(C::WAVES * 64), 0, stream, a);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

}  // namespace

// wpk2 must be conv_enc2.hip's packing with 8-channel chunks (its production variants of these two layers)
bool s2_supported(int cin, int cout, int stride, const EncConvArgs& a) {
    const char* off = getenv("EEM_NO_S2W");              // read per launch: the tests compare both kernels in one process
    if (off && off[0] == '1') return false;
    const char* e64 = getenv("EEM_S2W_64");
    const bool also64 = e64 && e64[0] == '1';
    return stride == 2 && ((cin == 16 && cout == 32) || (also64 && cin == 32 && cout == 64)) && a.wpk2 && a.gate == nullptr &&
           a.pool_partial == nullptr && (a.win & 3) == 0 && (((uintptr_t)a.in0) & 15) == 0 && (size_t)cin * a.hin * a.win * 4 < (1u << 31);
}

int s2_launch(int cin, const EncConvArgs& a, hipStream_t stream) {
    if (cin == 16) return s2_launch_t<16, 32>(a, stream);
    return s2_launch_t<32, 64>(a, stream);
}
