// Training-step kernels (see train.hip).
#pragma once
#include "common.h"

struct WgradArgs {
    const float* x;        // forward input  [N][x_ctotal][hin][win], channels [x_coff, x_coff + cin)
    int x_ctotal, x_coff, cin;
    const float* g;        // output gradient [N][g_ctotal][hout][wout], channel co -> g_coff + co * g_cmul
    const float* gate;     // forward output (same indexing as g) for LeakyReLU', or NULL
    int g_ctotal, g_coff, g_cmul, cout;
    float* dw;             // [cout][cin][k][k], accumulated with atomics (zero it first)
    int n, hin, win, hout, wout, k, stride, pad;
    const float* zero_page;   // >= 16 bytes of zeros (LDS-DMA source for padding), or NULL: generic kernel only
    float* db;                // wgrad_enc only: bias gradient [cout], accumulated with atomics, or NULL
    // generic kernel only: rectangular filters (kh != 0 overrides k / pad: kh x kw taps, padding ph / pw) and a dW that is
    // a channel slice of a wider weight tensor: dw[co][dw_coff + ci][tap] with dw_cin input channels per cout (0: cin)
    int kh = 0, kw = 0, ph = 0, pw = 0, dw_cin = 0, dw_coff = 0;
    // wgrad_ring.hip only: the input as up to three channel-concatenated tensors (the torch.cat of model/update.py:44,51,79 is never
    // materialised): xs[s] is [N][xsc[s]][hin][win]; nxseg = 0: the single tensor x above.  cin = the channels of all segments
    int nxseg = 0;
    const float* xs[3] = {nullptr, nullptr, nullptr};
    int xsc[3] = {0, 0, 0};
};

int tr_loss_launch(const float* flow, const float* gt, const float* valid, float* dflow, int batch, int hw, float weight,
                   double* stats, hipStream_t st);
int tr_upsample_bwd_launch(const float* d, float* tmp, float* out, int nc, int oh, int ow, int h, int w, hipStream_t st);
int tr_pool_bwd_launch(const float* dpool, float* g, long nc, int h, int w, int k, int gh, int gw, int accumulate, const float* gate,
                       hipStream_t st);
int tr_corr_bwd_launch(const float* dcv, int dcv_ctotal, const float* f1, const float* f2, float* d1, float* d2, int batch, int c,
                       int h, int w, const int* taps, int ntaps, hipStream_t st);
struct CorrBwdJob { const float* dcv; const float* f1; const float* f2; float* d1; float* d2; int dcv_ctotal, c; };
// up to three correlations of one grid shape in one launch (the three stages of the tail)
int tr_corr_bwd_launch_jobs(const CorrBwdJob* jobs, int njobs, int batch, int h, int w, const int* taps, int ntaps, hipStream_t st);
int tr_bias_grad_launch(const float* g, const float* gate, int g_ctotal, int g_coff, int g_cmul, int cout, int n, int hw, float* db,
                        hipStream_t st);
struct BiasJob {
    const float* g;
    const float* gate;
    float* db;
    int g_ctotal, g_coff, g_cmul, cout, n, hw;
};
int tr_wgrad_launch(const WgradArgs& a, hipStream_t st);
// encoder-shaped jobs (3x3, pad 1, no gate, 16/32/64 couts, 16-byte aligned rows): wgrad_enc.hip
bool wgrad_enc_supported(const WgradArgs& a);
int wgrad_enc_launch(const WgradArgs& a, hipStream_t st);
// the same LDS-tiled kernel for stride-1 layers of any width with 3x3 / 1x5 / 5x1 / 1x1 filters (64-cout chunks on blockIdx.z;
// honours kh / kw / ph / pw and dw_cin / dw_coff): E-RAFT's residual stacks, update block and heads
bool wgrad_wide_supported(const WgradArgs& a);
int wgrad_wide_launch(const WgradArgs& a, hipStream_t st);
// round 6: the same jobs on LDS rings that run ahead of the MFMAs (wgrad_ring.hip: one 8-wave block per CU walks a run of output rows,
// G staged once for up to 64 input channels); EEM_NO_WGRAD_RING=1 (read per call) keeps the kernels above
bool wgrad_ring_supported(const WgradArgs& a);            // 3x3 (stride 1 / 2), 1x5, 5x1 (stride 1); cin, cout >= 16; honours kh / kw / dw_cin / dw_coff
int wgrad_ring_launch(const WgradArgs& a, hipStream_t st);
bool wgrad_ring_preferred(const WgradArgs& a);            // the shapes where it is the faster kernel (measured; EEM_WGRAD_RING=all / none)
// 3x3 stride-1 convs of at most 8 couts on the vector pipe, weight and bias gradient in one launch (train.hip); x / x_ctotal / x_coff,
// g / g_ctotal / g_coff, dw_cin / dw_coff, db as above
bool wgrad_few_supported(const WgradArgs& a);
int wgrad_few_launch(const WgradArgs& a, hipStream_t st);
// every 3x3 conv of the 1/64-grid tail (all decoders, groups and layers) in ONE launch: weight and bias gradients (wgrad_tail.hip)
bool wgrad_tail_supported(const WgradArgs* jobs, int njobs, int n, int h, int w);
int wgrad_tail_launch(const WgradArgs* jobs, int njobs, int n, int h, int w, hipStream_t st);
// several convs of the same kernel size / stride in one launch (blockIdx.z = job); at most WGRAD_MAX_JOBS
#define WGRAD_MAX_JOBS 16
int tr_wgrad_launch_batch(const WgradArgs* jobs, int njobs, hipStream_t st);
int tr_bias_grad_launch_batch(const BiasJob* jobs, int njobs, hipStream_t st);
// data gradient of a stride-2 3x3 conv (pad 1) with the pooling branch and the LeakyReLU' gate of the producing layer folded
// in: dx = (convT(dy, w) + dpool / k^2) * (gate > 0 ? 1 : 0.1)   (dgrad_s2.hip; 16<-32 and 32<-64 channels)
struct DgradS2Args {
    const float* dy;       // [n][cout][hout][wout]
    const float* w;        // [cout][cin][3][3]
    float* dx;             // [n][cin][hin][win]
    const float* gate;     // forward tensor at dx's shape, or NULL
    const float* dpool;    // [n][cin][gh][gw] gradient of the k x k average pooling of the same tensor, or NULL
    const float* zero_page;
    float* trash;          // >= 512 writable bytes: sink for the stores of out-of-image lanes
    int n, cin, cout, hin, win, hout, wout, pool_k, gh, gw;
};
bool dgrad_s2_supported(const DgradS2Args& a);
int dgrad_s2_launch(const DgradS2Args& a, hipStream_t st);
// the same for wide layers (E-RAFT's encoders: cout 96 / 128, any cin; 3x3 pad 1 or the 1x1 pad-0 shortcut): no gate, no pooling branch
bool dgrad_s2w_supported(const DgradS2Args& a, int ksize);
int dgrad_s2w_launch(const DgradS2Args& a, int ksize, hipStream_t st);
int tr_sumsq_launch(const float* g, long n, double* out, hipStream_t st);
// repack_launch that also advances *nskip when *sumsq is not finite and clears *sumsq (the launch behind tr_adamw_launch)
int tr_repack_after_step_launch(const float* flat, const int* idx, float* arena, long n, double* sumsq, int* nskip, hipStream_t st);
int tr_adamw_launch(float* p, const float* g, float* m, float* v, long n, const double* sumsq, float clip, float lr, float wd,
                    float eps, float b1, float b2, long step, int* nskip, hipStream_t st);
