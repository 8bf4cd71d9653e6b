// Shared declarations for the EEMFlow MI355X (gfx950) hot-path library.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#include <stdio.h>

#include <type_traits>

#include "../../include/eemflow_hip.h"   // every definition of an entry point sees its declaration (default visibility; the build hides the rest)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define EEM_OK 0
#define EEM_ERR_ARG 1
#define EEM_ERR_HIP 2
#define EEM_ERR_STATE 3

void eem_set_error(const char* fmt, ...);

#define EEM_HIP_CHECK(expr)                                                            \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            eem_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),       \
                          __FILE__, __LINE__);                                         \
            return EEM_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

#define EEM_REQUIRE(cond, ...)                                                         \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            eem_set_error(__VA_ARGS__);                                                \
            return EEM_ERR_ARG;                                                        \
        }                                                                              \
    } while (0)

__host__ __device__ static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// grid of the encoder launch this thread made last (blocks x threads per block): eemflow_time_kernels reports it beside the
// duration - a kernel that runs on a quarter of the chip by design is not a slow kernel (api.hip defines it)
extern thread_local int eem_last_grid_blocks, eem_last_grid_threads, eem_last_pipe;
#define EEM_NOTE_GRID(blocks, threads) do { eem_last_grid_blocks = (int)(blocks); eem_last_grid_threads = (int)(threads); } while (0)
// which matrix pipe the launch's contraction runs on: 0 fp32 MFMA (157 TFLOP/s), 1 fp32 products as six bf16-piece MFMAs (2.5 PFLOP/s / 6),
// 2 Winograd F(4x4,3x3) on the fp32 MFMA (36 products per 16 outputs where the direct form has 144: its roof in direct-convolution FLOPs
// is 4 x 157), 3 F(2x2,3x3) (16 per 4 outputs against 36: 2.25 x)
#define EEM_NOTE_PIPE(p) do { eem_last_pipe = (int)(p); } while (0)

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N) - the index is usable as a template /
// inline-asm immediate inside f
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    // counts above the 6-bit field are clamped: waiting for more than asked is always safe
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N > 63 ? 63 : N) : "memory");
}

// XCD-aware block remap: the dispatcher deals consecutive block ids round-robin over the 8 XCDs (each with
// its own L2), so id b and b+8 share an L2.  Giving XCD x the contiguous range [x*cpx, (x+1)*cpx) of
// logical tiles makes spatially adjacent tiles (which share halo rows/columns) hit the same L2.  Placement
// is a speed matter only; the grid is padded to a multiple of 8 and surplus blocks exit.
__device__ static inline unsigned xcd_logical_block(unsigned bid, unsigned nblocks_padded) {
    const unsigned cpx = nblocks_padded >> 3;
    return (bid & 7u) * cpx + (bid >> 3);
}

// Sum over groups of SW (4, 8 or 16) adjacent lanes with DPP row operations (VALU cross-lane moves - no
// LDS traffic, unlike __shfl_xor's ds_bpermute): every lane of a group ends up holding the group sum.
template <int CTRL>
__device__ static __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int SW>
__device__ static __forceinline__ float lane_group_sum(float v) {
    static_assert(SW == 4 || SW == 8 || SW == 16, "group width");
    v = dpp_add<0xB1>(v);                       // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);                       // quad_perm [2,3,0,1]
    if (SW >= 8) v = dpp_add<0x141>(v);         // row_half_mirror: lane i <-> 7-i
    if (SW >= 16) v = dpp_add<0x140>(v);        // row_mirror: lane i <-> 15-i
    return v;
}

// Persistent blocks walk a CONTIGUOUS range of logical tiles (x fastest, then y, then image): XCD x owns tiles
// [x*cpx, (x+1)*cpx) and its resident blocks split that range evenly.  Consecutive tiles of a block are
// neighbours in x, so the coordinates advance without integer divisions - the CU's single scalar unit is shared
// by all its waves, and three divisions per tile and wave were ~20 % of a tile's time in the Winograd kernels.
// Persistent encoder kernels launch one block per CU (32 per XCD) by default; EEM_ENC_PER_XCD_<tag> (or EEM_ENC_PER_XCD) lowers the cap:
// fewer, longer blocks amortise the per-block prologue when several frames share the chip.
static inline int enc_blocks_per_xcd(const char* tag, int dflt) {
    char name[48];
    snprintf(name, sizeof name, "EEM_ENC_PER_XCD_%s", tag);
    const char* e = getenv(name);
    if (!e) e = getenv("EEM_ENC_PER_XCD");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : dflt;
}
struct TileCoord { int bx, by, n; };
struct TileRange { int first, count; };
__device__ static inline TileRange block_tile_range(int T, unsigned bid, unsigned nblocks) {
    const int cpx = (T + 7) >> 3;
    const int xcd = bid & 7, kb = bid >> 3, gb = nblocks >> 3;
    const int r0 = xcd * cpx, r1 = min(r0 + cpx, T);
    const int per = (cpx + gb - 1) / gb;
    const int f = r0 + kb * per;
    const int cnt = min(per, r1 - f);
    return TileRange{f, cnt > 0 ? cnt : 0};
}
__device__ static inline TileCoord tile_coord(int lt, int tiles_x, int tiles_y) {
    return TileCoord{lt % tiles_x, (lt / tiles_x) % tiles_y, lt / (tiles_x * tiles_y)};
}
// (the same walk backwards: image order reversed, see EncConvArgs::reverse)
__device__ static inline void tile_retreat(TileCoord& t, int tiles_x, int tiles_y) {
    if (t.bx-- == 0) {
        t.bx = tiles_x - 1;
        if (t.by-- == 0) { t.by = tiles_y - 1; --t.n; }
    }
}
__device__ static inline void tile_advance(TileCoord& t, int tiles_x, int tiles_y) {
    if (++t.bx == tiles_x) {
        t.bx = 0;
        if (++t.by == tiles_y) { t.by = 0; ++t.n; }
    }
}

// ----------------------------------------------------------------------------- encoder conv
// Identifies one of the eight encoder layers (EEMFlow.py:75-82).
enum EncLayer { ENC_1_1 = 0, ENC_1_2, ENC_2_1, ENC_2_2, ENC_2_3, ENC_3_1, ENC_3_2, ENC_3_3, ENC_NUM };

struct EncLayerDesc {
    int cin, cout, stride;
};
static const EncLayerDesc kEncLayers[ENC_NUM] = {
    {5, 16, 2}, {16, 16, 1}, {16, 32, 2}, {32, 32, 1}, {32, 32, 1}, {32, 64, 2}, {64, 64, 1}, {64, 64, 1}};

// Number of packed floats for an encoder layer's MFMA A-fragments (see conv_enc.hip).
size_t enc_packed_floats(int cin, int cout);
// Pack OIHW weights [cout][cin][3][3] into MFMA A-fragment order (host).
void enc_pack_weights(const float* w, int cin, int cout, float* packed);

// Launch a fused {replicate-pad (first layer only)} + conv3x3 + bias + LeakyReLU(0.1).
//  in : layer ENC_1_1: two raw tensors ev1, ev2 of shape [B][cin][hraw][wraw] (images 0..B-1, B..2B-1)
//       other layers : [nimg][cin][hin][win]
//  out: [nimg][cout][hout][wout], hout = (hin-1)/stride+1 (pad 1, k 3)
struct EncConvArgs {
    const float* in0;
    const float* in1;      // only ENC_1_1 (second event volume), else unused
    const float* wpk;      // packed weights, generic kernel (conv_enc.hip)
    const float* wpk2;     // packed weights, LDS-DMA fast path (conv_enc2.hip); may be NULL
    const float* wwino;    // Winograd-domain weights (conv_wino.hip) for the stride-1 C->C layers; may be NULL
    const float* zero_page;// >= 16 zero bytes in device memory (source of out-of-image pieces)
    float* trash;          // >= 1024 writable bytes: sink for the stores of out-of-image lanes (up to 16 B per lane)
    const float* bias;
    float* out;
    int nimg;              // images in this launch (2B)
    int nimg0;             // images taken from in0 (B) - rest from in1
    int hin, win;          // conv input extent (after the fused replicate pad for ENC_1_1)
    int hout, wout;
    int hraw, wraw;        // raw extent for ENC_1_1 (== hin, win elsewhere)
    int pad_top, pad_left; // replicate-pad offsets for ENC_1_1 (0 elsewhere)
    int act;               // 1: LeakyReLU(0.1); 2: ReLU (conv_wino4.hip only: E-RAFT's encoder); 0: none
    const float* gate;     // backward use: [nimg][cout][hout][wout]; the result is multiplied by LeakyReLU'(gate)
    int tiles_x, tiles_y;  // filled by the launcher: block tiles per image (blocks are remapped XCD-aware)
    float* pool_partial;   // fast path only: per-block partial sums of the k x k stage pooling, or NULL
    int pool_k;
    // ENC_1_1 inside a cached HIP graph: device table {events1, events2, flow_out}; when non-NULL the kernel reads in0 / in1
    // from it instead of from the fields above, so the graph does not depend on the caller's buffers (api.hip)
    const void* const* io;
    // 0: the table is {events1, events2, flow_out} of one batch in contiguous tensors.  n >= 1 (eemflow_forward_many): the table holds n
    // triples {events1_i, events2_i, flow_out_i}, one single-frame buffer each - image i < nimg0 is read from io[3 i], image nimg0 + i from
    // io[3 i + 1]: n unrelated frames ride one batch-n chain
    int io_frames = 0;
    // ENC_1_1 only: the event volumes are RAW voxel grids, each followed by its four-float normalisation record (eemflow_voxelize with
    // normalize = 2); the kernel normalises as it reads (conv_enc1.hip) - the other first-layer kernels do not know this form
    int in_norm = 0;
    // walk the tiles from the LAST image to the first (conv_wino4.hip): in a batched chain a layer's input was written by the launch
    // before it, front to back, and is larger than the 256 MB Infinity Cache - read back to front, the part written last is still there
    int reverse = 0;
    // feature-map stores as non-temporal stores (conv_wino4.hip; EEM_NT_STORE=<layer mask>, experiment): a batched chain's outputs stream
    // through each XCD's 4 MB L2 beside the Winograd weights and halo rows the kernel re-reads
    int nt_store = 0;
    // persistent kernels: blocks per XCD (0 = one per CU).  The context lowers it when the application keeps several frames in flight
    // (eemflow_set_frames_in_flight): fewer, longer blocks spend less CU time on per-block prologues
    int blocks_per_xcd = 0;
    int wino_f4 = 0;       // wwino holds F(4x4,3x3) weights (conv_wino4.hip) instead of F(2x2,3x3) ones
    const float* ws2r = nullptr;   // weights of a stride-2 layer in conv_s2r.hip's order (s2r_transform_launch), or NULL
    // conv_wino4.hip only: residual [nimg][C][hout][wout]; the result is relu(res + act(conv)) (ResidualBlock, model/extractor.py:50-57)
    const float* res = nullptr;
    // the stride-2 layer 16 -> 32 as pre-split bf16 weight fragments (conv_bx3.hip, bx3_transform_launch), or NULL
    const float* wbx3 = nullptr;
    // a layer whose output is only read through its fused pooling partial sums (pconv3_3 in inference: f13, EEMFlow.py:152-154) may skip
    // the feature-map stores: a hint - kernels built with a store-free form honour it (the two Winograd forms at C = 64), the others store
    int no_store = 0;
};
// Every launch argument a block uses, wanted in scalar registers at its first instruction: one batch of scalar loads instead of the three or
// four dependent ones the compiler otherwise spreads over the prologue (see tail_conv_kernel in tail.hip for what a memory round trip costs
// beside other frames' kernels)
#define ENC_ARGS_NOW(a)                                                                                                                  \
    asm volatile("" ::"s"((a).in0), "s"((a).in1), "s"((a).wpk), "s"((a).wpk2), "s"((a).wwino), "s"((a).zero_page), "s"((a).trash),        \
                 "s"((a).bias), "s"((a).out), "s"((a).nimg), "s"((a).nimg0), "s"((a).hin), "s"((a).win), "s"((a).hout), "s"((a).wout),      \
                 "s"((a).hraw), "s"((a).wraw), "s"((a).pad_top), "s"((a).pad_left), "s"((a).act), "s"((a).gate), "s"((a).tiles_x),        \
                 "s"((a).tiles_y), "s"((a).pool_partial), "s"((a).io), "s"(gridDim.x))
int enc_conv_launch(int cin, int cout, int stride, const EncConvArgs& a, hipStream_t stream);
// first layer with 16-byte LDS-DMA staging (conv_enc1.hip): raw width % 4 == 0, no horizontal padding
bool enc1_supported(const EncConvArgs& a);
int enc1_launch(const EncConvArgs& a, hipStream_t stream);
// fast path (feature width % 4 == 0, layers 2..8)
bool enc2_supported(int cin, int cout, int stride, int win);
size_t enc2_packed_floats(int cin, int cout);
void enc2_pack_weights(const float* w, int cin, int cout, float* packed);
int enc_conv2_launch(int cin, int cout, int stride, const EncConvArgs& a, hipStream_t stream);
// the two stride-2 layers as light blocks with register-resident weights (conv_s2.hip); reads wpk2 in its 8-channel-chunk packing
bool s2_supported(int cin, int cout, int stride, const EncConvArgs& a);
int s2_launch(int cin, const EncConvArgs& a, hipStream_t stream);
void enc2_tile(int cin, int cout, int* th, int* tw, int* poolk);
// the stride-2 layers 16 -> 32 and 32 -> 64 as 8-wave blocks on an LDS-DMA ring of k-step slices, weights included (conv_s2r.hip)
bool s2r_shape(int cin, int cout, int stride);
size_t s2r_packed_floats(int cin, int cout);
int s2r_transform_launch(const float* w, int cin, int cout, float* packed, hipStream_t stream);     // OIHW device weights -> packed
bool s2r_supported(int cin, int cout, int stride, const EncConvArgs& a);
int s2r_launch(int cin, const EncConvArgs& a, hipStream_t stream);
// pconv2_1 on the bf16 matrix pipe with fp32 results: three bf16 pieces per operand, six MFMAs per product (conv_bx3.hip)
bool bx3_shape(int cin, int cout, int stride);
size_t bx3_packed_floats(int cin, int cout);
int bx3_transform_launch(const float* w, int cin, int cout, float* packed, hipStream_t stream);     // OIHW device weights -> packed
bool bx3_supported(int cin, int cout, int stride, const EncConvArgs& a);
int bx3_launch(int cin, int stride, const EncConvArgs& a, hipStream_t stream);
// Winograd F(2x2,3x3) path for the stride-1 C -> C layers (C = 16, 32, 64), conv_wino.hip
bool wino_supported(int cin, int cout, int stride, int win);
size_t wino_packed_floats(int c);      // room for either form
// U = G g G^T of OIHW weights w [c][c][3][3] (device pointers), written in the register-fragment order of
// wino_kernel (f4 = 0) or wino4_kernel (f4 = 1); transpose_flip = 1 transforms W^T with flipped taps (the data gradient's weights)
int wino_transform_launch(const float* w, int c, int transpose_flip, float* packed, hipStream_t stream, int f4);
int wino_launch(int c, const EncConvArgs& a, hipStream_t stream);          // a.wino_f4 selects the form
void wino_tile(int c, int f4, int* th, int* tw, int* poolk);
// Winograd F(4x4,3x3) on 16x16x4 MFMAs, one wave per SIMD (conv_wino4.hip); reached through wino_*
size_t wino4_packed_floats(int c);
int wino4_transform_launch(const float* w, int c, int transpose_flip, float* packed, hipStream_t stream);
#define W4_WT_JOBS 12
int wino4_transform_multi_launch(const float* const* w, const int* c, const int* transpose_flip, float* const* packed, int njobs,
                                 hipStream_t stream);
int wino4_launch(int c, const EncConvArgs& a, hipStream_t stream);
void wino4_tile(int c, int* th, int* tw, int* poolk);
// C = 32 / 64 on 32x32x2 MFMA with the Winograd rows split over 4 waves (conv_wino32.hip); reached through wino_*
int wino32_transform_launch(const float* w, int c, int transpose_flip, float* packed, hipStream_t stream);
int wino32_launch(int c, const EncConvArgs& a, hipStream_t stream);
void wino32_tile(int c, int* th, int* tw, int* poolk);

// pconv1_1 computed inside pconv1_2's block (conv_enc12.hip): the first two encoder layers as ONE launch, `a1` never written to memory
struct Enc12Args {
    const float* in0;            // events1 [nimg0][5][hraw][wraw] (images 0 .. nimg0-1)
    const float* in1;            // events2 (images nimg0 .. nimg-1)
    const void* const* io;       // graph io table or NULL, as EncConvArgs::io / io_frames
    int io_frames;
    const float* wpk1;           // pconv1_1 weights packed by enc_pack_weights(5, 16)
    const float* bias1;          // pconv1_1 bias [16]
    const float* u2;             // pconv1_2 weights in the Winograd F(4x4,3x3) fragment order (wino4_transform_launch)
    const float* bias2;          // pconv1_2 bias [16]
    const float* zero_page;
    float* trash;
    float* out;                  // f11 [nimg][16][h1][w1]
    float* pool_partial;         // 32 x 32 stage-pooling partial sums (conv_wino4.hip's layout)
    float* scratch;              // enc12_scratch_floats(blocks) floats: a block's haloed a1 tile in ring-slot layout (L2-resident)
    int nimg, nimg0;
    int hraw, wraw, pad_top;     // raw event-volume extent; rows replicated on top (bottom rows follow from hin)
    int hin, win;                // replicate-padded extent = pconv1_1's input
    int h1, w1;                  // a1 / f11 extent
    int tiles_x, tiles_y;        // filled by the launcher
    int dbg;                     // diagnostic builds (-DEEM_DIAG): phases switched off (EEM_E12_DBG)
    int row_order;               // tiles in row order (x fastest) instead of the column walk (EEM_E12_ROW_ORDER=1: measurement)
};
bool enc12_supported(const Enc12Args& a);
int enc12_blocks(int nimg, int h1, int w1, int blocks_per_xcd);
size_t enc12_scratch_floats(int blocks);
int enc12_launch(const Enc12Args& a, int blocks, hipStream_t stream);

// ----------------------------------------------------------------------------- tail kernels
// Generic small-grid 3x3 (or 1x1) conv on MFMA 16x16x4, K split over the 4 waves of a block.
struct TailConvJob {
    const float* in;       // [B][in_ctotal][h][w]; this job reads channels [in_coff, in_coff+cin)
    const float* wpk;      // packed A fragments for this job
    const float* bias;     // [cout]
    float* out;            // [B][out_ctotal][h][w]; writes channel co*out_cmul + out_coff
    int cin, cout;
    int in_ctotal, in_coff;
    int out_ctotal, out_coff, out_cmul;
    int act;
    // backward use (data gradients of the tail convs): input channel i is read at in_coff + i * in_cmul (0 = 1) and
    // multiplied by LeakyReLU'(gate) (gate indexed like the input; NULL = none); bias may be NULL
    const float* gate;
    int in_cmul;
    const float* add;      // optional residual, indexed like `out` (added after the activation); NULL = none
};
#define TAIL_MAX_JOBS 16
struct TailConvLaunch {
    TailConvJob job[TAIL_MAX_JOBS];
    int njobs;
    int batch, h, w;
    int ksize;             // 3 or 1
};
size_t tail_packed_floats(int cin, int cout, int ksize);
void tail_pack_weights(const float* w, int cin, int cout, int ksize, float* packed);
int tail_conv_launch(const TailConvLaunch& l, hipStream_t stream);

// avg-pool k x k (stride k, floor) of [n][c][h][w] -> [n][c][h/k][w/k]
struct PoolJob { const float* in; float* out; int c, h, w, k; };
int pool_launch(const PoolJob* jobs, int njobs, int nimg, hipStream_t stream);
// sums the per-block partial sums written by the fused conv epilogue: out[n][c][gy][gx] =
// (1/k^2) * sum_{i < rows} partial[n][c][gy*rows + i][gx], partial is [n][c][prows][pcols]
struct PoolFinJob { const float* partial; float* out; int c, prows, pcols, rows, k; };
int pool_finalize_launch(const PoolFinJob* jobs, int njobs, int nimg, int gh, int gw, hipStream_t stream);

// 9x9 local correlation, selected taps, scaled by 1/C; writes channels [0,ntaps) of out
struct CorrJob { const float* f1; const float* f2; float* out; int c, out_ctotal; };
int corr_launch(const CorrJob* jobs, int njobs, int batch, int h, int w, const int* taps_dev, int ntaps,
                hipStream_t stream);

// plain bilinear resize (align_corners False) of [n][c][h][w] -> [n][c][oh][ow]
// io: optional device table {events1, events2, flow_out}; when non-NULL the destination is io[2] (see EncConvArgs::io)
int upsample_launch(const float* in, float* out, int nc, int h, int w, int oh, int ow, hipStream_t stream,
                    const void* const* io = nullptr, int io_frames = 0);
// writes {e1, e2, out} into the device table (one tiny launch in front of a cached graph whose buffers changed)
int spin_launch(float us, hipStream_t stream);          // diagnostic builds (-DEEM_DIAG): one wave asleep for <us> (tools/marginal.sh)
int io_table_launch(const void** table, const void* e1, const void* e2, void* out, hipStream_t stream);
// the per-frame form: n triples {e1[i], e2[i], out[i]} (eemflow_forward_many), n <= EEM_MAX_COALESCE
#define EEM_MAX_COALESCE 16
int io_table_many_launch(const void** table, int n, const float* const* e1, const float* const* e2, float* const* out, hipStream_t stream);

// ---- fused launches of the tail (tail_fused.hip)
// A pooled [n][c][gh][gw] map read through the conv epilogues' partial sums: value = scale * sum_{i < rows} base[n*nstride +
// c*cstride + y*ystride + i*rstride + x]  (a finished map: rows 1, scale 1)
struct PooledSrc {
    const float* base;
    int nstride, cstride, ystride, rstride, rows;
    float scale;
};
#define TAIL_HEAD_MAX_TAPS 81
struct TailHeadArgs {
    PooledSrc src[3];            // the three stages (images 0..B-1 = events1, B..2B-1 = events2)
    int c[3];                    // 16, 32, 64
    float* cat[3];               // [B][cat_ctotal][g]: correlation -> channels [0, ntaps), rconv -> [ntaps, ntaps + 16)
    float* pool_out[3];          // finished pooled maps [2B][c][g] (side output) or NULL
    const float* rw[3];          // rconv_k weights packed by tail_pack_weights
    const float* rb[3];          // rconv_k bias
    int batch, gh, gw, ntaps, cat_ctotal;
    int grid_x;                  // filled by the launcher
    int tap[TAIL_HEAD_MAX_TAPS]; // filled by the launcher from taps_host: the selected taps of the 9x9 window (EEMFlow.py:14-23)
};
int tail_head_launch(const TailHeadArgs& a, const int* taps_host, hipStream_t stream);
struct TailUpArgs {
    const float* flowcat;        // the three decoders' flows [B][6][g]
    const float* wo;             // out_conv weight [2][6], bias [2] (state_dict layout)
    const float* bo;
    float* coarse;               // side output [B][2][g] (may be NULL)
    float* out;                  // [B][2][oh][ow]
    const void* const* io;       // graph io table (out = io[2]) or NULL
    int io_frames;               // > 0: per-frame triples, frame b writes io[3 b + 2] (see EncConvArgs::io_frames)
    int batch, gh, gw, oh, ow, out_aligned16;
    int ty, tx;                  // filled by the launcher
};
bool tail_up_supported(int gh, int gw, int oh, int ow);
int tail_up_launch(const TailUpArgs& a, hipStream_t stream);

// arena[i] = idx[i] ? flat[idx[i] - 1] : 0  (weight re-packing after an optimizer step, see api.hip)
int repack_launch(const float* flat, const int* idx, float* arena, long n, hipStream_t stream);

// ----------------------------------------------------------------------------- voxelizer
int voxel_launch(const double* events, int64_t n, int bins, int h, int w, int normalize, float* grid,
                 int64_t* idx_left, int64_t* idx_right, void* scratch, hipStream_t stream);
size_t voxel_scratch_bytes(int64_t n);
// up to VOX_MAX_JOBS voxelizations of the same grid shape in ONE three-launch sequence (blockIdx.y = job)
#define VOX_MAX_JOBS 32
int voxel_launch_jobs(int njobs, const double* const* events, const int64_t* n, int bins, int h, int w, int normalize, float* const* grid,
                      int64_t* const* idx_left, int64_t* const* idx_right, void* const* scratch, hipStream_t stream);
