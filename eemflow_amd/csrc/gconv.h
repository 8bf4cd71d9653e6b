// Generic NCHW fp32 convolution on the fp32 matrix cores (used by the E-RAFT / EEMFlow+ parts of the path,
// where kernel sizes, strides and channel counts vary per layer: 7x7 s2, 3x3, 1x1, 1x5, 5x1; 2..384 -> 2..576).
#pragma once
#include "common.h"

struct GConvSeg {
    const float* ptr;      // [N][ctotal][hin][win]
    int c, ctotal, coff;   // this segment = channels coff + i*cmul, i in [0, c)
    int cmul;              // channel stride (0 is read as 1); >1 undoes a channel shuffle in backward passes
    const float* gate;     // optional, same shape/indexing as ptr: the value is multiplied by LeakyReLU'(gate)
                           // (1 if gate > 0 else 0.1) - backward through convrelu without a separate pass
};

enum { GACT_NONE = 0, GACT_RELU = 1, GACT_SIGMOID = 2, GACT_TANH = 3, GACT_LEAKY = 4 };
// epilogue modes: v = act(acc * scale[co] + shift[co]) then
enum {
    GEPI_PLAIN = 0,        // out = v
    GEPI_MUL = 1,          // out = v * e0                      (r * h of the GRU)
    GEPI_GRU = 2,          // out = (1 - e1) * e0 + e1 * v      (e0 = h, e1 = z)
    GEPI_ADD_RELU = 3,     // out = relu(v + e0)                (residual block tail)
    GEPI_ADD = 4,          // out = v + e0                      (EEMFlow+ decoder: flow residual)
    GEPI_ZR = 5,           // co < split: out = v;  co >= split: out2[co - split] = v * e0[co - split]
    GEPI_SUM2 = 6          // out = v and out2 = e0 + v         (E-RAFT: delta_flow and coords1 + delta_flow; fewout kernel only)
                           // (z | r of a GRU pass as one conv: z stays, r leaves as r * h - model/update.py:46-48,54-56)
};

struct GConvArgs {
    GConvSeg seg[3];
    int nseg;
    const float* wpk;      // packed by gconv_pack
    const float* wpk16;    // packed by gconv16_pack (stride-1 layers with 16-aligned channel counts), or NULL
    const float* zero_page;// >= 16 zero bytes on the device (LDS-DMA source of out-of-image pieces); needed with wpk16
    const float* scale;    // [cout] or NULL (= 1)
    const float* shift;    // [cout] or NULL (= 0)
    float* out;            // [N][out_ctotal][hout][wout], channel co goes to out_coff + co
    int out_ctotal, out_coff;
    int out_cmul;          // channel co goes to out_coff + co * out_cmul (0 is read as 1): channel shuffle of grouped convs
    int n, hin, win, hout, wout, cout;
    int kh, kw, stride, pad_h, pad_w;
    int tstride;           // 0/1: ordinary conv.  2: transposed conv (data gradient of a stride-2 conv): the
                           // source position (o - pad + tap) must be a multiple of tstride and is divided by it
    int act, epi;
    const float* e0; int e0_ctotal, e0_coff;
    const float* e1; int e1_ctotal, e1_coff;
    float out_scale;       // final multiplier (1 = none)
    int in_flight;         // frames the application keeps in flight on this GPU (0 / 1: one): tile choices favour CU time over latency
    const float* wfew;     // packed by fewout_pack (layers of <= 8 couts, 3x3 stride 1, one input segment; its layout depends on cout <= 2), or NULL
    // grouped convolution as ONE launch of the LDS-tiled kernel (EEMFlow+'s decoder: three 32 -> 32 groups + channel shuffle, EEMFlow+.py:52-63);
    // 0 / 1 = none.  All fields above describe group 0; group g reads the channels g * seg[0].c of segment 0 onwards, takes its weights
    // g_wstride16 floats and its scale / shift g_pstride floats further on, and writes channel out_coff + g * g_ocoff + co * out_cmul
    int groups;
    long g_wstride16, g_pstride;
    int g_ocoff;
    // optional per-pixel addend in front of the activation: v = act(acc * scale + shift + pre[n][pre_coff + co][p]) - the part of a
    // convolution over concatenated inputs that belongs to an input which does not change between calls (E-RAFT's GRU: the context
    // features, model/update.py:43-60), computed once
    const float* pre;
    int pre_ctotal, pre_coff;
    float* out2;           // GEPI_ZR / GEPI_SUM2: [N][out2_ctotal][hout][wout]
    int out2_ctotal, split;
    const float* wpkb;     // packed by gconvb_pack (pre-split bf16 B fragments, gconvb.hip), or NULL
    const float* wstem;    // packed by stem7_pack (the encoders' 7x7 stride-2 stem on <= 5 channels, conv_stem7.hip), or NULL
};

// number of packed floats / packing for weights [cout][sum(c_s)][kh][kw] read as segments of sizes cs[0..nseg)
size_t gconv_packed_floats(int cout, const int* cs, int nseg, int kh, int kw);
void gconv_pack(const float* w, int cout, const int* cs, int nseg, int kh, int kw, float* packed);
int gconv_launch(const GConvArgs& a, hipStream_t stream);
// LDS-tiled 16x16x4 path (gconv16.hip); gconv_launch takes it when a.wpk16 is set and the launch qualifies
bool gconv16_shape(int cout, const int* cs, int nseg, int kh, int kw, int stride);
size_t gconv16_packed_floats(int cout, const int* cs, int nseg, int kh, int kw);
void gconv16_pack(const float* w, int cout, const int* cs, int nseg, int kh, int kw, float* packed);
bool gconv16_supported(const GConvArgs& a);
// Layers of <= 8 output channels (EEMFlow+'s mask estimator tail 176 -> 8 -> 3, the 32 -> 2 flow convs): on the matrix cores a
// 32-cout tile is 75-94 % padding, so these run as a direct convolution on the vector pipe - a thread per pixel, the weights of a
// (channel, tap) as one uniform 32-byte load.  [cin][kh*kw][8] floats.
// the encoders' stem: 7x7, stride 2, padding 3, <= 5 input channels, 64 couts (conv_stem7.hip)
size_t stem7_packed_floats(int cin);
void stem7_pack(const float* w_64xcinx7x7, int cin, float* packed);
bool stem7_supported(const GConvArgs& a);
int stem7_launch(const GConvArgs& a, hipStream_t stream);
size_t fewout_packed_floats(int cin, int kh, int kw);
void fewout_pack(const float* w, int cout, int cin, int kh, int kw, float* packed);
bool fewout_supported(const GConvArgs& a);
int fewout_launch(const GConvArgs& a, hipStream_t stream);
int gconv16_launch(const GConvArgs& a, hipStream_t stream);
// the same layers on the bf16 matrix pipe with exact three-piece operands (gconvb.hip): 3x3 / 1x5 / 5x1, cout >= 96, launches of at
// least 64 blocks of 8 rows x 16 pixels x 128 couts; gconv_launch takes it when a.wpkb is set and the launch qualifies
bool gconvb_shape(int cout, const int* cs, int nseg, int kh, int kw, int stride);
size_t gconvb_packed_floats(int cout, const int* cs, int nseg, int kh, int kw);
void gconvb_pack(const float* w, int cout, const int* cs, int nseg, int kh, int kw, float* packed);
bool gconvb_supported(const GConvArgs& a);
int gconvb_launch(const GConvArgs& a, hipStream_t stream);
// device-side: gconv16_pack's stream of the same weights -> gconvb_pack's (ops.hip: packings follow the weights of a training step)
int gconvb_from16_launch(const float* wpk16, int cout, int cin, int taps, float* wpkb, hipStream_t stream);
