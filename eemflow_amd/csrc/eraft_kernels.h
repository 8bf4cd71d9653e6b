// Non-convolution kernels of the E-RAFT part of the path (reference: model/eraft.py, model/corr.py,
// model/extractor.py, model/model_utils.py).
#pragma once
#include "common.h"

// F.pad(mode='replicate') of [nc][h][w] -> [nc][h+top+bottom][w+left+right]   (utils/image_utils.py:139-140)
int er_pad_launch(const float* in, float* out, int nc, int h, int w, int left, int right, int top, int bottom, hipStream_t st);
// both event volumes of a sample (in0 -> planes [0, nc), in1 -> [nc, 2 nc) of out) as one launch
int er_pad2_launch(const float* in0, const float* in1, float* out, int nc, int h, int w, int left, int right, int top, int bottom,
                   hipStream_t st);

// InstanceNorm2d(affine=False, eps=1e-5) per (n,c) plane + options (model/extractor.py:31-35,43-57):
//   v = (x - mean) / sqrt(var + eps);  if relu_inner: v = relu(v);  if res: v = relu(v + res)
// planes too large for the in-register kernel take a two-pass form that needs er_instnorm_scratch_doubles(planes, hw) doubles of scratch
// owned by the caller (one per context / stream); without it they fall back to one block per plane
size_t er_instnorm_scratch_doubles(int planes, int hw);
int er_instnorm_launch(const float* x, float* out, const float* res, int planes, int hw, int relu_inner, hipStream_t st, double* stats = nullptr,
                       size_t stats_cap = 0);

// All-pairs correlation (model/corr.py:53-60): out[b][p1][p2] = sum_c f1[b][c][p1] * f2[b][c][p2] / sqrt(C)
int er_allpairs_launch(const float* f1, const float* f2, float* out, int batch, int c, int hw, hipStream_t st);

// avg_pool2d(2, stride 2) over the last two dims of [planes][h][w] -> [planes][h/2][w/2]   (model/corr.py:24-27)
int er_pool2_launch(const float* in, float* out, long planes, int h, int w, hipStream_t st);
int er_pool2x3_launch(const float* in, float* o1, float* o2, float* o3, long planes, int h, int w, hipStream_t st);   // three levels, one launch

// 4-level 9x9 bilinear lookup (model/corr.py:29-50, model/model_utils.py:7-21), keeping the reference's
// transposed window: channel l*81 + i*9 + j samples (x/2^l + i-4, y/2^l + j-4).
struct LookupArgs {
    const float* pyr[4];
    int ph[4], pw[4];
    const float* coords;   // [B][2][H][W]
    float* out;            // [B][out_ctotal][H][W], channels 0..323 are written
    int batch, h, w, out_ctotal;
    // optional: the level-0 blocks also write flow = coords - coords0 into channels [flow_coff, flow_coff + 2) of flow_dst
    // (model/eraft.py:144 and the motion encoder's cat([out, flow]), model/update.py:81) - saves the separate launch
    const float* coords0 = nullptr;
    float* flow_dst = nullptr;
    int flow_ctotal = 0, flow_coff = 0;
};
int er_lookup_launch(const LookupArgs& a, hipStream_t st, hipEvent_t done_ev = nullptr);

// the same 324 features computed on the fly from fmap1 and the avg-pooled levels of fmap2 (no all-pairs volume)
struct AltCorrArgs {
    const float* f1;       // [B][C][H][W]
    const float* f2[4];    // level l: [B][C][H >> l][W >> l] (avg_pool2d(2) chain of fmap2)
    int ph[4], pw[4];
    const float* coords;   // [B][2][H][W]
    float* out;            // [B][out_ctotal][H][W], channels 0..323 are written
    int batch, c, h, w, out_ctotal;
    float scale;           // 1 / sqrt(C)
};
int er_altcorr_launch(const AltCorrArgs& a, hipStream_t st);

// coords grid (model/model_utils.py:24-27): out[b][0][y][x] = x, out[b][1][y][x] = y  (+ init if not NULL)
int er_coords_init_launch(float* coords0, float* coords1, const float* flow_init, int batch, int h, int w, hipStream_t st);
// flow = coords1 - coords0 written into dst (channel offset dst_coff of a dst_ctotal-channel tensor)
int er_flow_launch(const float* coords0, const float* coords1, float* dst, int dst_ctotal, int dst_coff, int batch, int hw,
                   hipStream_t st);
// coords1 += delta
int er_axpy_launch(float* y, const float* x, long n, hipStream_t st);
int er_sum_launch(float* out, const float* a, const float* b, long n, hipStream_t st);      // out = a + b
int er_mul_channels_launch(float* out, const float* a, int a_ctotal, int a_coff, const float* b, int batch, int c, long hw, hipStream_t st);

// convex upsampling (model/eraft.py:83-94) of flow = coords1 - coords0 with mask [B][576][H][W], written
// unpadded: out[b][c][Y - top][X - left] for the (8H x 8W) result cropped to [oh][ow]
// delta / coords1_next (optional, both or neither): the flow that is upsampled is coords1 + delta - coords0 and the updated
// coordinates coords1 + delta (model/eraft.py:149) are written to coords1_next (a buffer other than coords1: neighbouring threads
// still read the old values) - saves the separate coords1 += delta launch
int er_convex_up_launch(const float* coords0, const float* coords1, const float* mask, float* out, int batch, int h, int w,
                        int top, int left, int oh, int ow, hipStream_t st, const float* delta = nullptr, float* coords1_next = nullptr);
