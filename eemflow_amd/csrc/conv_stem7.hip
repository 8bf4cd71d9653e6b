// The encoders' stem (model/extractor.py:136: Conv2d(n_first_channels, 64, 7, stride 2, padding 3)) on event volumes of up to five
// bins, as an implicit GEMM on the fp32 matrix pipe:  out[64 couts][pixels] = W[64][K] x patches[K][pixels],  K = (channel, ky, kx).
//
// Why a kernel of its own: gconv_taps_kernel<7, 7> (gconv.hip) gathers 49 stride-2 taps per channel pair per lane from global memory
// and spends 32x32x2 MFMAs on pairs of channels - five bins are three pairs - at 0.3 of the pipe (E-RAFT 640x480, batch 4: 247 + 493 us
// for the two networks, 5 % of the forward).  Here
//   * a block owns 8 x 32 output pixels x all 64 couts; its input patch (5 channels x 21 rows x 72 columns, 30 KB) is read ONCE, as
//     coalesced 16-byte pieces, into LDS - the next tile's pieces travel into registers while the current tile multiplies;
//   * K is ordered (channel, ky, kx padded to 8): a k-step of v_mfma_f32_16x16x4_f32 is four consecutive kx of one patch row, so a
//     lane's B operand is  patch[base(lane) + compile-time offset]  - one ds_read_b32, no address arithmetic (the eighth tap's weight
//     is zero); 70 k-steps for five bins;
//   * the weights (64 x 280, A fragments of four cout tiles per lane as one float4) sit in LDS for the life of the block (70 KB);
//   * wave w multiplies output rows 2w, 2w + 1 (four 16-pixel tiles) x four cout tiles: 16 MFMAs per k-step from 1 ds_read_b128 + 4
//     ds_read_b32; scale / shift (folded BatchNorm) and the activation in the epilogue, 64-byte row segments out.
#include "gconv.h"

namespace {

constexpr int ST_TH = 8, ST_TW = 32;                    // output tile
constexpr int ST_ROWS = 2 * ST_TH + 5, ST_COLS = 72;    // patch rows; columns from 2 x0 - 4 (16-byte aligned) to 2 x0 + 67
constexpr int ST_PLANE = ST_ROWS * ST_COLS;
constexpr int ST_MAXC = 5;
constexpr int ST_QPR = ST_COLS / 4;                      // 16-byte pieces per patch row

__device__ __forceinline__ float st_act(float v, int act) {
    switch (act) {
        case GACT_RELU: return v > 0.f ? v : 0.f;
        case GACT_LEAKY: return v > 0.f ? v : 0.1f * v;
        case GACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case GACT_TANH: return tanhf(v);
        default: return v;
    }
}

template <int CIN>
__global__ __launch_bounds__(256) void stem7_kernel(GConvArgs a, const f32x4* __restrict__ wst, int tiles_x, int tiles_y) {
    constexpr int KS = CIN * 7 * 2;                      // k-steps: (channel, ky, half of the padded kx row)
    constexpr int PIECES = CIN * ST_ROWS * ST_QPR;       // 16-byte pieces of a patch
    constexpr int PPT = (PIECES + 255) / 256;            // per thread
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x4* wl = reinterpret_cast<f32x4*>(lds);           // [KS][64 lanes] float4: the four cout tiles' A operands
    float* patch = lds + KS * 64 * 4;                    // [CIN][ST_ROWS][ST_COLS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n16 = lane & 15, kg = lane >> 4;
    const int hwi = a.hin * a.win, hwo = a.hout * a.wout;
    for (int e = tid; e < KS * 64; e += 256) wl[e] = wst[e];
    const int ntiles = tiles_x * tiles_y * a.n;

    f32x4 nxt[PPT];
    auto fetch = [&](int tile) __attribute__((always_inline)) {
        const int img = tile / (tiles_x * tiles_y), t2 = tile - img * (tiles_x * tiles_y);
        const int ty = t2 / tiles_x, tx = t2 - ty * tiles_x;
        const int gy0 = 2 * ty * ST_TH - 3, gx0 = 2 * tx * ST_TW - 4;
        const float* src = a.seg[0].ptr + ((size_t)img * a.seg[0].ctotal + a.seg[0].coff) * hwi;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int f = tid + 256 * k;
            const int c = f / (ST_ROWS * ST_QPR), rem = f - c * (ST_ROWS * ST_QPR);
            const int r = rem / ST_QPR, q = rem - r * ST_QPR;
            const int gy = gy0 + r, gx = gx0 + 4 * q;
            // (win % 4 == 0 and gx0 % 4 == 0: a piece is inside or outside the image as a whole)
            const bool ok = f < PIECES && gy >= 0 && gy < a.hin && gx >= 0 && gx < a.win;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            nxt[k] = ok ? *reinterpret_cast<const f32x4*>(src + (size_t)c * hwi + (size_t)gy * a.win + gx) : z;
        }
    };
    auto stash = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int f = tid + 256 * k;
            if (f < PIECES) *reinterpret_cast<f32x4*>(patch + 4 * f) = nxt[k];
        }
    };

    // the lane's 16 couts' scale and shift, once (inside the epilogue they were 32 dependent L2 round trips per tile)
    float sc[4][4], sh[4][4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = 16 * ct + 4 * kg + j;
            sc[ct][j] = a.scale ? a.scale[co] : 1.f;
            sh[ct][j] = a.shift ? a.shift[co] : 0.f;
        }
    int tile = blockIdx.x;
    if (tile < ntiles) fetch(tile);
    // B operand of (row r of the wave, pixel tile t, k-step): patch[c][2 (2 wave + r) + ky][2 (16 t + n16) + 4 half + kg + 1]
    // (column 0 of the patch is image column 2 x0 - 4; tap kx reads image column 2 ox - 3 + kx)
    const int bbase = (2 * (2 * wave)) * ST_COLS + 2 * n16 + kg + 1;
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                                 // (the previous tile is multiplied; the first pass: the weights are in)
        stash();
        __syncthreads();
        const int cur = tile;
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
        f32x4 acc[4][4];                                 // [pixel tile: 2 rows x 2 halves][cout tile]
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[p][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int c = 0; c < CIN; ++c) {
#pragma unroll
            for (int ky = 0; ky < 7; ++ky)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int ks = (c * 7 + ky) * 2 + half;
                    const f32x4 av = wl[ks * 64 + lane];
                    const float* pb = patch + c * ST_PLANE + ky * ST_COLS + 4 * half + bbase;
                    float bv[4];
#pragma unroll
                    for (int p = 0; p < 4; ++p) bv[p] = pb[(p >> 1) * 2 * ST_COLS + (p & 1) * 32];
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) acc[p][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ct], bv[p], acc[p][ct], 0, 0, 0);
                }
        }
        // ---- epilogue: lane (n16, kg) holds couts 16 ct + 4 kg + j of pixel (row 2 wave + p / 2, column 16 (p % 2) + n16)
        const int img = cur / (tiles_x * tiles_y), t2 = cur - img * (tiles_x * tiles_y);
        const int ty = t2 / tiles_x, tx = t2 - ty * tiles_x;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = 16 * ct + 4 * kg + j;
                float* dst = a.out + ((size_t)img * a.out_ctotal + a.out_coff + co) * hwo;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int oy = ty * ST_TH + 2 * wave + (p >> 1), ox = tx * ST_TW + 16 * (p & 1) + n16;
                    if (oy < a.hout && ox < a.wout) dst[(size_t)oy * a.wout + ox] = st_act(acc[p][ct][j] * sc[ct][j] + sh[ct][j], a.act) * a.out_scale;
                }
            }
    }
}

}  // namespace

size_t stem7_packed_floats(int cin) { return (size_t)cin * 14 * 64 * 4; }

// w [64][cin][7][7] -> [k-step = (c, ky, half)][lane = (m, kg)][cout tile ct]: W[16 ct + m][c][ky][4 half + kg] (kx = 7: zero)
void stem7_pack(const float* w, int cin, float* packed) {
    for (int c = 0; c < cin; ++c)
        for (int ky = 0; ky < 7; ++ky)
            for (int half = 0; half < 2; ++half)
                for (int lane = 0; lane < 64; ++lane)
                    for (int ct = 0; ct < 4; ++ct) {
                        const int co = 16 * ct + (lane & 15), kx = 4 * half + (lane >> 4);
                        packed[((((size_t)c * 7 + ky) * 2 + half) * 64 + lane) * 4 + ct] = kx < 7 ? w[(((size_t)co * cin + c) * 7 + ky) * 7 + kx] : 0.f;
                    }
}

bool stem7_supported(const GConvArgs& a) {
    const char* e = getenv("EEM_NO_STEM7");                       // read per call: a test flips it inside one process
    if (e && e[0] == '1') return false;
    return a.wstem && a.nseg == 1 && a.seg[0].c >= 1 && a.seg[0].c <= ST_MAXC && a.cout == 64 && a.kh == 7 && a.kw == 7 && a.stride == 2 &&
           a.tstride <= 1 && a.pad_h == 3 && a.pad_w == 3 && a.groups <= 1 && a.epi == GEPI_PLAIN && a.pre == nullptr &&
           a.seg[0].gate == nullptr && a.seg[0].cmul <= 1 && a.out_cmul <= 1 && a.win % 4 == 0 && ((uintptr_t)a.seg[0].ptr & 15) == 0 &&
           ((size_t)a.hin * a.win) % 4 == 0 && a.hout == (a.hin - 1) / 2 + 1 && a.wout == (a.win - 1) / 2 + 1;
}

template <int CIN>
static int stem7_launch_t(const GConvArgs& a, hipStream_t stream) {
    const int tiles_x = ceil_div(a.wout, ST_TW), tiles_y = ceil_div(a.hout, ST_TH);
    const int ntiles = tiles_x * tiles_y * a.n;
    const int lds_bytes = (CIN * 14 * 64 * 4 + CIN * ST_PLANE) * 4;
    static bool raised = false;
    if (!raised) {
        EEM_HIP_CHECK(hipFuncSetAttribute((const void*)stem7_kernel<CIN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised = true;
    }
    static const int cus = [] { int d = 0, n = 256; hipDeviceProp_t p; if (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) n = p.multiProcessorCount; return n > 0 ? n : 256; }();
    const int grid = ntiles < cus ? ntiles : cus;       // one resident block per CU, each walks tiles blockIdx.x, + grid, ...
    hipLaunchKernelGGL((stem7_kernel<CIN>), dim3(grid), dim3(256), lds_bytes, stream, a, reinterpret_cast<const f32x4*>(a.wstem), tiles_x, tiles_y);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int stem7_launch(const GConvArgs& a, hipStream_t stream) {
    switch (a.seg[0].c) {
        case 1: return stem7_launch_t<1>(a, stream);
        case 2: return stem7_launch_t<2>(a, stream);
        case 3: return stem7_launch_t<3>(a, stream);
        case 4: return stem7_launch_t<4>(a, stream);
        default: return stem7_launch_t<5>(a, stream);
    }
}
