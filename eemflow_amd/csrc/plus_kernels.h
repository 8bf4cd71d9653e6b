// Warp / resampling kernels of EEMFlow+ (see plus_kernels.hip).
#pragma once
#include "common.h"

// mode 0: EEMFlow_cdc.warp (align_corners=True); 1: torch_warp (align False); 2: WarpingLayer_no_div (align False + mask)
// x [b][c][h][w]; flow = channels 0,1 of a [b][flow_ctotal][h][w] tensor; out channels [out_coff, out_coff+c) of [b][out_ctotal][h][w]
int pl_warp_launch(const float* x, const float* flow, int flow_ctotal, float* out, int out_ctotal, int out_coff, int batch, int c, int h,
                   int w, int mode, hipStream_t st);
int pl_upflow_launch(const float* in, float* out, int batch, int h, int w, int oh, int ow, int rate, hipStream_t st);
// up to five upsamplings to the same output size as one launch (falls back to pl_upflow_launch per job when ow % 4 or alignment say so)
int pl_upflow_multi_launch(const float* const* in, float* const* out, const int* h, const int* w, int njobs, int batch, int oh, int ow,
                           int rate, hipStream_t st);
int pl_scale_flow_launch(float* f, int batch, int hw, float su, float sv, hipStream_t st);
int pl_blend_launch(const float* warped, const float* flow_init, const float* xout, float* out, int batch, int hw, hipStream_t st);
// torch_warp(flow_init, xout[:, 0:2]) blended with flow_init by sigmoid(xout[:, 2]) -> out [b][2][hw] and channels [cat_coff, +2) of cat
int pl_warp_blend_launch(const float* flow_init, const float* xout, float* out, float* cat, int cat_ctotal, int cat_coff, int batch, int h, int w,
                         hipStream_t st);
// dst[:, d_coff : d_coff + c] = src[:, s_coff : s_coff + c]   (src NULL: zeros)
int pl_copy_channels_launch(const float* src, int s_ctotal, int s_coff, float* dst, int d_ctotal, int d_coff, int c, int batch, int hw,
                            hipStream_t st);
// round 6: warp_blend + the warp of feature_2 by the flow_up it forms (EEMFlow+.py:189) as one launch; upsample2d_flow_as + its in-place
// doubling of the coarse flow (into fc_scaled, a second buffer) + WarpingLayer_no_div(x, flow_init) as one launch (h x w >= 2 hc x wc)
int pl_warp_blend_warp_launch(const float* flow_init, const float* xout, float* out, float* cat, int cat_ctotal, int cat_coff, const float* f2,
                              float* fw, int c2, int batch, int h, int w, hipStream_t st);
int pl_upflow_warp_launch(const float* fc, float* fc_scaled, int hc, int wc, float* fi, const float* x, float* out, int out_ctotal, int out_coff,
                          int batch, int c, int h, int w, hipStream_t st);
