/*
 * eemflow_hip.h - C ABI of libeemflow_hip.so: the MI355X (gfx950) implementation of EEMFlow's
 * dense-flow hot path.  Plain pointers and sizes only; every device pointer is a HIP device
 * address on the context's device, every `stream` is a hipStream_t passed as void* (NULL = the
 * default stream).  All functions return 0 on success; on failure they return non-zero and
 * eemflow_last_error() describes the problem (thread-local, valid until the next failing call).
 *
 * The reference (boomluo02/EEMFlow) has no FFI of its own on this path: its boundary is the
 * nn.Module interface plus two third-party torch extensions.  Each entry point below names the
 * reference interface it replaces (paths relative to the reference repo root).
 *
 * Tensors are dense NCHW fp32 unless stated.  Ownership: the caller owns every buffer it passes;
 * the context owns its weights and workspaces and never retains caller pointers beyond a call.
 * Threading: one context per host thread/stream; calls on one context must not overlap.
 */
#ifndef EEMFLOW_HIP_H
#define EEMFLOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* Opaque contexts (declared ahead of the visibility block: their C++ members are not part of the ABI). */
typedef struct eemflow_ctx eemflow_ctx;
typedef struct eraft_ctx eraft_ctx;
typedef struct eemplus_ctx eemplus_ctx;

/* The library is built with -fvisibility=hidden: the entry points declared here (and nothing else - no kernel launcher, no C++
 * helper) are its dynamic symbols. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* ABI version of this header (bumped on incompatible change). */
int eemflow_abi_version(void);

/* Last error message of the calling thread ("" if none). */
const char* eemflow_last_error(void);

/* Create / destroy a model context on HIP device `device`.
 * Replaces: EEMFlow.__init__ + model.to(device)  (model/EEMFlow/EEMFlow.py:72-112,
 * test_EEMFlow_HREM.py:53-55,113). */
int eemflow_create(int device, eemflow_ctx** out);
void eemflow_destroy(eemflow_ctx* ctx);

/* Load all parameters from one flat host fp32 vector holding the 66 tensors of the reference
 * state_dict() back to back in registration order (pconv1_1.0.weight, pconv1_1.0.bias, ...,
 * out_conv.bias; 714 352 floats for n_first_channels=5, groups=5).
 * Replaces: model.load_state_dict(checkpoint['state_dict'])  (test_EEMFlow_HREM.py:62-66). */
int eemflow_load_weights(eemflow_ctx* ctx, const float* flat_host, size_t nfloats, int n_first_channels,
                         int groups);

/* New values for the parameters of an already loaded model, from a DEVICE vector in the same order: one
 * device-to-device copy plus the on-device re-pack of every kernel-side layout (no host round trip).  This is what a
 * training loop that owns its optimizer calls after optimizer.step() changed the nn.Parameters in place.
 * Replaces: the implicit "parameters are read where they live" of nn.Module (train_mvsec.py:257 scaler.step). */
int eemflow_update_weights(eemflow_ctx* ctx, const float* flat_device, size_t nfloats, void* stream);

/* Configure the replicate padder for images of `height` x `width` (pad to a multiple of 64, 'chairs'
 * mode: width split left/right, height bottom only).  pad_out = [left, right, top, bottom].
 * Replaces: EEMFlow.change_imagesize -> InputPadder(img_size, 'chairs', 64)
 * (model/EEMFlow/EEMFlow.py:114-116, utils/image_utils.py:129-137). */
int eemflow_set_image_size(eemflow_ctx* ctx, int height, int width, int pad_out[4]);

/* Replay the forward pass through a cached HIP graph (1, default) or launch kernels eagerly (0).  Graphs are keyed on shapes,
 * not on the caller's buffers: the two launches that touch events1 / events2 / flow_out read those pointers from a device
 * table that one tiny launch rewrites when a call brings other buffers, so fresh tensors every frame (the reference's
 * evaluation loop, test_mvsec.py:580-597) replay the same graph.  Up to four shapes stay cached (least recently used out). */
int eemflow_use_graph(eemflow_ctx* ctx, int enable);

/* Throughput hint: the application keeps `n` frames in flight on this GPU (one context and one HIP stream each, e.g. the
 * evaluation loop of test_mvsec.py:580-597 pipelined over several samples).  With n >= 3 the persistent encoder kernels
 * launch fewer, longer blocks - the frames time-slice the CUs and per-block prologues are CU time another frame could use
 * (1280x720, four in flight: +3.5 % frames/s, +8 % latency of a single frame).  Default 1: lowest single-frame latency.
 * Give the process enough hardware queues for its streams (GPU_MAX_HW_QUEUES). */
int eemflow_set_frames_in_flight(eemflow_ctx* ctx, int n);

/* enable != 0: the event volumes handed to eemflow_forward / eemflow_forward_many are RAW voxel grids, each followed in memory by its
 * four-float normalisation record - what eemflow_voxelize / _pair / _many write with normalize = 2.  The first convolution applies
 * (v - mean) / sd to the non-zero voxels as it reads them, so the voxelizer's read-modify-write of every grid (2 x 18.4 MB per volume at
 * 1280x720x5) is never made.  Needs the 5-bin first layer on volumes whose rows are 16-byte multiples and that pad on the bottom only
 * (HREM 1280x720: yes; MVSEC 346x260: no - normalise in the voxelizer there); the call fails otherwise.  In a contiguous batch
 * (eemflow_forward) every sample is [C*H*W + 4] floats long.  Training entry points ignore the flag.
 * Replaces: loader/loader_utils.py:527-535 (the normalisation inside EventSequenceToVoxelGrid_Pytorch.__call__) feeding
 * model/EEMFlow/EEMFlow.py:135 (pconv1_1). */
int eemflow_set_deferred_input_norm(eemflow_ctx* ctx, int enable);

/* Graph-cache statistics: out3 = {captures, replays, io-table rewrites}. */
int eemflow_graph_stats(eemflow_ctx* ctx, long long out3[3]);

/* Inference forward: events1/events2 [batch][C][in_h][in_w] -> flow_out [batch][2][out_h][out_w].
 * The reference upsamples the 1/64 grid straight to the *input* size (out_h,out_w = in_h,in_w) or,
 * in training with out_mesh_size, to 16x16.
 * Replaces: EEMFlow.forward(events1, events2)[1][0]  (model/EEMFlow/EEMFlow.py:122-183), i.e. the
 * call made by TestRaftEvents.run_network (test_mvsec.py:1444-1455). */
int eemflow_forward(eemflow_ctx* ctx, const float* events1, const float* events2, int batch, int in_h,
                    int in_w, float* flow_out, int out_h, int out_w, void* stream);

/* nframes (1..16) INDEPENDENT samples as one batch-nframes chain: events1[i] / events2[i] are [1][C][in_h][in_w] tensors and
 * flow_out[i] a [1][2][out_h][out_w] tensor, each wherever the caller has it (16-byte aligned; the three arrays are HOST arrays of
 * device pointers, read before the call returns).  Bitwise the result of eemflow_forward on the batch of those frames: the same
 * kernels in the same launch configuration - only the first conv and the upsampling find each frame through a device table of
 * per-frame pointers, which one small launch rewrites in front of the cached graph's replay.  Why: one frame's launches leave most
 * of the 256 CUs idle (64 - 120 tiles in the 32 / 64-channel layers, 240 pixels in the tail); a caller with several samples at hand -
 * the evaluation loop below - gets the batched chain without first copying them into one tensor.
 * Replaces: n iterations of `for sample in loader: model(events1=im1, events2=im2)` at batch 1, test_mvsec.py:580-597 /
 * TestRaftEvents.run_network (test_mvsec.py:1444-1455). */
int eemflow_forward_many(eemflow_ctx* ctx, int nframes, const float* const* events1, const float* const* events2,
                         float* const* flow_out, int in_h, int in_w, int out_h, int out_w, void* stream);

/* Per-kernel timing of the forward schedule: the schedule runs `reps` + 1 times as the chain it is (eagerly, the first pass warms and
 * is not counted) with a pair of HIP events around EVERY launch, recorded on `stream` (the stream the kernels run on); `ms` is the
 * average per launch - each kernel measured behind its producer, as it runs in a forward.  `flops` / `bytes` are the ALGORITHMIC
 * work of one launch (conv MACs x 2; compulsory input + output + weight bytes).  flow_out receives the correct flow.
 * reps < 0: the other form - every kernel of the schedule -reps times back to back between one pair of events (launches of a few
 * microseconds, which an event pair per launch would mostly measure the gaps of).
 * Replaces: the reference's time_eval() wall-clock loop (model/EEMFlow/EEMFlow.py:201-225), per kernel. */
typedef struct eemflow_kernel_stat {
    char name[48];
    double flops;
    double bytes;
    float ms;
    int blocks;      /* workgroups of the launch (0: not reported): the encoder's kernels are persistent, one workgroup per CU,
                        so blocks < 256 means the launch occupies that many of the 256 CUs */
    int pipe;        /* matrix pipe of the launch's contraction: 0 = fp32 MFMA, 1 = fp32 products as six bf16-piece MFMAs (conv_bx3.hip),
                        2 = Winograd F(4x4,3x3) on the fp32 MFMA (a quarter of the direct form's multiplies), 3 = F(2x2,3x3) (1 / 2.25) */
    int reserved;
} eemflow_kernel_stat;
int eemflow_time_kernels(eemflow_ctx* ctx, const float* events1, const float* events2, int batch, int in_h,
                         int in_w, float* flow_out, int out_h, int out_w, int reps, eemflow_kernel_stat* stats,
                         int max_stats, int* nstats, void* stream);

/* Copy an intermediate of the LAST forward into dst (device).  Names: "f11","f12","f13" (stage
 * outputs, images 0..B-1 = events1, B..2B-1 = events2), "pool_1".."pool_3", "cat_1".."cat_3"
 * ([cv(53) | r(16)]), "flow_1".."flow_3" (as channels 0-1, 2-3, 4-5 of "flowcat"), "flowcat",
 * "coarse".  dims_out = [n, c, h, w].  For parity tests against the oracle's stage tensors. */
int eemflow_get_stage(eemflow_ctx* ctx, const char* name, float* dst, size_t dst_capacity_floats,
                      int dims_out[4], void* stream);

/* One decoder (k = 1..3) on a caller-supplied [batch][69][h][w] input -> [batch][2][h][w].
 * Replaces: Decoder.forward  (model/EEMFlow/EEMFlow.py:59-69). */
int eemflow_decoder(eemflow_ctx* ctx, int k, const float* x, int batch, int h, int w, float* out, void* stream);

/* 9x9 local correlation of two [batch][c][h][w] maps, 53 selected taps, scaled by 1/c
 * -> out [batch][53][h][w].
 * Replaces: Correlation.forward + torch.index_select  (model/EEMFlow/EEMFlow.py:14-23,160), i.e.
 * spatial_correlation_sampler.SpatialCorrelationSampler(1, 9, 1, 0, 1) (requirements.txt:131). */
int eemflow_local_corr53(const float* f1, const float* f2, int batch, int c, int h, int w, float* out,
                         void* stream);

/* Bilinear resize, align_corners=False: in [nc][h][w] -> out [nc][oh][ow].
 * Replaces: EEMFlow.upsample_flow = F.interpolate(..., mode='bilinear', align_corners=False)
 * (model/EEMFlow/EEMFlow.py:118-120). */
int eemflow_upsample_bilinear(const float* in, float* out, int nc, int h, int w, int oh, int ow, void* stream);

/* Evaluation statistics of one flow field against its ground truth, on the device.
 * Replaces: Test.flow_error (test_mvsec.py:291-346).  flow_gt, flow_pred: [2][h][w]; event_img: [h][w] event counts for the
 * 'sparse' evaluation or NULL ('dense'); max_row: rows >= max_row are ignored (190 for the MVSEC is_car crop, else >= h -
 * note the reference passes the WIDTH there, test_mvsec.py:296).  out5 (device, 5 doubles): sum EE, sum |gt|, n_points,
 * count(EE < 1), count(EE < 3 or EE < 0.1 |gt|) over the pixels with finite non-zero ground truth (and event_img > 0). */
int eemflow_flow_error(const float* flow_gt, const float* flow_pred, const float* event_img, int h, int w, int max_row,
                       double* out5, void* stream);

/* The same statistics for n (1..16) samples of one image size by ONE launch - the samples of an eemflow_forward_many call: flow_gt[i],
 * flow_pred[i], event_img[i] (event_img may be NULL: 'dense') are HOST arrays of device pointers; out5n (device): n x 5 doubles, row i
 * as eemflow_flow_error gives for sample i.
 * Replaces: n calls of Test.flow_error in the evaluation loop (test_mvsec.py:580-597 -> :291-346). */
int eemflow_flow_error_many(int n, const float* const* flow_gt, const float* const* flow_pred, const float* const* event_img, int h,
                            int w, int max_row, double* out5n, void* stream);

/* Event voxelization: events [n][4] f64 (t, x, y, p) on the device, time-sorted, as held by the
 * reference's EventSequence -> grid [bins][h][w] fp32.  idx_left / idx_right (optional, may be NULL)
 * receive, per event, the int64 flat index x + y*w + bin*w*h of the left / right temporal vote, or -1
 * where the reference masks the vote out.
 * normalize: 0 raw grid; 1 normalised grid (the reference's normalize=True); 2 "deferred" - the grid stays raw and the four floats
 * BEHIND it (grid[bins*h*w .. +4): the buffer must have them) receive {mean, sd, scale ? 1 : 0, any ? 1 : 0} for
 * eemflow_set_deferred_input_norm's consumer.
 * Replaces: EventSequenceToVoxelGrid_Pytorch.__call__  (loader/loader_utils.py:447-537). */
int eemflow_voxelize(const double* events, int64_t n, int bins, int h, int w, int normalize, float* grid,
                     int64_t* idx_left, int64_t* idx_right, void* stream);

/* The two event volumes of a sample (event_volume_old, event_volume_new) in ONE three-launch sequence: the same grids as two
 * eemflow_voxelize calls, bit for bit.
 * Replaces: the two EventSequenceToVoxelGrid_Pytorch calls of a dataset sample  (loader/HREM.py:226-232, loader/MVSEC.py:166-177). */
int eemflow_voxelize_pair(const double* events1, int64_t n1, const double* events2, int64_t n2, int bins, int h, int w,
                          int normalize, float* grid1, float* grid2, void* stream);

/* nsets (1..32) event sets of ONE grid shape - e.g. both volumes of every sample of an eemflow_forward_many call - voxelized by ONE
 * three-launch sequence: events[k] [n_events[k]][4] f64 (t, x, y, p) -> grids[k] [bins][h][w] f32, each bitwise what eemflow_voxelize
 * gives for that set alone.  `events`, `n_events`, `grids` are HOST arrays, read before the call returns.
 * Replaces: the per-sample calls of EventSequenceToVoxelGrid_Pytorch.__call__ (loader/loader_utils.py:459-537) in a loader that has
 * several samples at hand (loader/HREM.py:226-232 for each of them). */
int eemflow_voxelize_many(int nsets, const double* const* events, const int64_t* n_events, int bins, int h, int w, int normalize,
                          float* const* grids, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training step of EEMFlow (train_mvsec.py:229-258).  Weights live on the device in state_dict order; one
 * flat gradient buffer in the same order is what a data-parallel job all-reduces (RCCL) between the two calls.
 * ---------------------------------------------------------------------------------------------- */

/* The two halves of autograd through EEMFlow.forward, for torch.autograd.Function (eemflow_amd/eemflow.py):
 * eemflow_forward_train runs the forward eagerly and keeps every activation in the context's workspace;
 * eemflow_backward turns d loss / d flow [batch][2][out_h][out_w] (any loss the caller differentiated) into the flat
 * parameter gradient grad_out (device, 714 352 floats, state_dict order).  `serial_out` identifies the forward whose
 * activations the context holds; eemflow_backward refuses (non-zero return) when another forward has overwritten them -
 * the caller then repeats the forward.  events1/events2 passed to eemflow_backward are that forward's inputs.
 * Replaces: model(im1, im2) under autograd + loss.backward()  (train_mvsec.py:245-253,377-386; autograd of
 * model/EEMFlow/EEMFlow.py:122-183). */
int eemflow_forward_train(eemflow_ctx* ctx, const float* events1, const float* events2, int batch, int in_h, int in_w,
                          float* flow_out, int out_h, int out_w, int64_t* serial_out, void* stream);
int eemflow_backward(eemflow_ctx* ctx, int64_t serial, const float* events1, const float* events2, const float* dflow,
                     float* grad_out, void* stream);

/* One term of sequence_loss and its gradient: weight * mean over batch*2*h*w of valid*|flow - gt| with
 * valid = (valid >= 0.5) & (|gt| < 400).  dflow_out [batch][2][h][w] = d term / d flow.  stats6 (DEVICE, 6 doubles, added
 * to - the caller zeroes them): sum of valid |flow - gt| (loss term = weight * stats6[0] / (batch*2*h*w)), sum of EPE over
 * valid pixels, valid count, count(EPE < 1), count(EPE < 3), count(EPE < 5).  No host synchronisation.
 * Replaces: train.sequence_loss  (train_mvsec.py:201-227) and its autograd. */
int eemflow_sequence_loss(const float* flow, const float* flow_gt, const float* valid, int batch, int h, int w, float weight,
                          float* dflow_out, double* stats6, void* stream);

/* Forward (train-mode output size), sequence loss for the single prediction (weight = gamma^0 = 1 unless the
 * caller scales it), and the full backward pass.  flow_gt [batch][2][out_h][out_w], valid [batch][out_h][out_w];
 * flow_out [batch][2][out_h][out_w] receives the prediction; grad_out (device, 714 352 floats) receives
 * d loss / d parameter; stats_out (host, 5 doubles, may be NULL): loss, mean EPE over valid pixels, valid
 * count, fraction < 1 px, fraction < 3 px.
 * Replaces: run_network + sequence_loss + scaler.scale(loss).backward()
 * (train_mvsec.py:245-253,201-227; autograd of model/EEMFlow/EEMFlow.py:122-183). */
int eemflow_forward_backward(eemflow_ctx* ctx, const float* events1, const float* events2, const float* flow_gt,
                             const float* valid, int batch, int in_h, int in_w, int out_h, int out_w,
                             float loss_weight, float* flow_out, float* grad_out, double* stats_out, void* stream);

/* clip_grad_norm_(clip) + AdamW(lr, weight_decay, eps, betas 0.9/0.999) on the device-resident weights, then the
 * re-pack of every kernel-side weight layout.  lr is the caller's OneCycleLR value for this step.
 * A gradient with an inf / NaN in it skips the step (see eemflow_optimizer_skipped_steps).
 * Replaces: scaler.unscale_ + clip_grad_norm_ + scaler.step(optimizer)  (train_mvsec.py:254-258,178-183). */
int eemflow_optimizer_step(eemflow_ctx* ctx, const float* grad, float lr, float weight_decay, float eps, float clip,
                           void* stream);

/* The statistics of the last eemflow_forward_backward (called with stats_out = NULL) without a stream synchronisation between the
 * backward and the optimizer step: _async enqueues the copy of the raw sums to pinned host memory and an event on `stream`; the
 * caller enqueues what follows (gradient all-reduce, eemflow_optimizer_step, the next forward); _wait blocks on that event alone and
 * returns stats[5] = loss, mean EPE, valid count, fraction < 1 px, fraction < 3 px - the numbers eemflow_forward_backward returns.
 * Replaces: the .item() reads of sequence_loss's metrics (train_mvsec.py:219-226) - after the step is in flight instead of before. */
int eemflow_train_stats_async(eemflow_ctx* ctx, void* stream);
int eemflow_train_stats_wait(eemflow_ctx* ctx, double stats_out[5]);

/* Steps eemflow_optimizer_step has skipped so far: a gradient holding an inf or a NaN changes neither weights nor moments and does
 * not advance the bias corrections - what GradScaler.step does for the reference (train_mvsec.py:237,257; the schedule still
 * advances, :258).  Synchronises `stream`. */
int eemflow_optimizer_skipped_steps(eemflow_ctx* ctx, int* out, void* stream);

/* Copy the current weights (state_dict order, 714 352 floats) to a device buffer - checkpointing
 * (train_EEMFlow_HREM.py:127-130 saves model.module.state_dict()). */
int eemflow_get_weights(eemflow_ctx* ctx, float* dst, size_t nfloats, void* stream);

/* ------------------------------------------------------------------------------------------------
 * E-RAFT (model/eraft.py): feature / context encoders, all-pairs correlation pyramid, 9x9 x 4-level
 * lookup, SepConvGRU update block, convex upsampling.  Inference (eval-mode BatchNorm).
 * ---------------------------------------------------------------------------------------------- */

/* Replaces: ERAFT.__init__ + .to(device)  (model/eraft.py:40-62). */
int eraft_create(int device, eraft_ctx** out);
void eraft_destroy(eraft_ctx* ctx);

/* All float tensors of the reference state_dict() (179 entries; the integer num_batches_tracked buffers are
 * skipped) back to back in registration order: fnet.*, cnet.* (BatchNorm weight, bias, running_mean,
 * running_var; the aliased downsample.1.* entries are present, as in the checkpoint), update_block.*.
 * Replaces: load_state_dict for model/eraft.py:57-62 modules. */
int eraft_load_weights(eraft_ctx* ctx, const float* flat_host, size_t nfloats, int n_first_channels);

/* events1/2 [batch][C][in_h][in_w]; pad = [left, right, top, bottom] from InputPadder(img_size, 'chairs', 32)
 * (model/eraft.py:65-67); flow_init [batch][2][H/8][W/8] of the padded size or NULL.
 * flow_out [iters][batch][2][in_h][in_w]: every iteration's convex-upsampled, unpadded flow.
 * Replaces: ERAFT.forward(events1, events2, iters, flow_init)[1]  (model/eraft.py:97-159). */
int eraft_forward(eraft_ctx* ctx, const float* events1, const float* events2, int batch, int in_h, int in_w,
                  const int pad[4], int iters, const float* flow_init, float* flow_out, void* stream);

/* n (1..16) independent batch-1 samples, each in its own tensors (events1[i] / events2[i] [1][C][in_h][in_w]; flow_out[i]
 * [iters][1][2][in_h][in_w], or [1][1][2][in_h][in_w] after eraft_set_final_only), as ONE batch-n forward - bitwise what eraft_forward
 * returns for the samples stacked into a batch (flow_init: none).
 * Replaces: n iterations of the evaluation loop at batch 1 (test_mvsec.py:580-597 -> run_network, :1444-1455) for ERAFT. */
int eraft_forward_many(eraft_ctx* ctx, int n, const float* const* events1, const float* const* events2, int in_h, int in_w,
                       const int pad[4], int iters, float* const* flow_out, void* stream);

/* Intermediates of the LAST forward: "fmap" ([2B,256,h,w]: fmap1 then fmap2), "inp", "flow_low", "pyr0".."pyr3", and - only
 * after eraft_keep_stages(ctx, 1), which adds four device copies to every forward - "corr0" (first lookup), "net1", "mask1",
 * "delta1" (after the first update). */
int eraft_keep_stages(eraft_ctx* ctx, int enable);

/* Correlation features computed on the fly instead of from an all-pairs volume (1: on, 0 = default: the volume stays resident).
 * Replaces the `alt_cuda_corr` pattern of model/flowformer/corr.py:60-91 (RAFT's `alternate_corr`): the 81 taps of a
 * pixel at level l are bilinear samples of <fmap1[:, p], avg_pool2d^l(fmap2)[:, q]> / sqrt(C) over the 10 x 10 cells its window
 * touches - no B * (HW)^2 * 4/3 floats (829 MB per sample at 1280x720), ~250x the lookup's memory traffic per iteration.  Same
 * features up to summation order; eraft_get_stage("pyr<l>") is not available in this mode. */
int eraft_set_alternate_corr(eraft_ctx* ctx, int enable);

/* 1: eraft_forward writes ONE prediction, flow_out [1][batch][2][in_h][in_w] = the last entry of the reference's list - what the
 * evaluation loop reads (test_mvsec.py:1455 `batch['flow_list'][-1]`).  The mask head and the convex upsampling of iterations
 * 0 .. iters - 2 are not launched (their results feed nothing else: model/eraft.py:141-157); the hidden state, coords1 and the last
 * prediction are bit for bit those of the full forward.  0 = default: every iteration's prediction, as ERAFT.forward returns them. */
int eraft_set_final_only(eraft_ctx* ctx, int enable);

/* Throughput hint, as eemflow_set_frames_in_flight: the application keeps `n` E-RAFT forwards in flight on this GPU (one context
 * and HIP stream each).  With n >= 3 the 16-aligned stride-1 convs use 4-row tiles from 512 blocks on (2 048 otherwise): the
 * other frames fill the CUs a short launch leaves idle, and each weight fragment is read half as often (640x480, 12 iterations,
 * batch 4, three in flight: 168 -> 175 frames/s; one forward at a time would lose 4 %).  Default 1. */
int eraft_set_frames_in_flight(eraft_ctx* ctx, int n);
int eraft_get_stage(eraft_ctx* ctx, const char* name, float* dst, size_t dst_capacity_floats, int dims_out[4],
                    void* stream);

/* CorrBlock(fmap1, fmap2, num_levels=4, radius=4)(coords): fmaps [batch][c][h][w], coords [batch][2][h][w]
 * -> out [batch][324][h][w].  Replaces: model/corr.py:13-60 (+ model/model_utils.py:7-21). */
int eraft_corr_lookup(eraft_ctx* ctx, const float* fmap1, const float* fmap2, const float* coords, int batch, int c,
                      int h, int w, float* out, void* stream);

/* Adjoint of eraft_corr_lookup w.r.t. the correlation pyramid (the reference detaches coords every iteration,
 * model/eraft.py:141, so there is no coordinate gradient): dout [batch][324][h][w] -> dpyr_l [batch*h*w][h>>l][w>>l],
 * l = 0..3 (zeroed here, then scatter-added).  Replaces: autograd of CorrBlock.__call__  (model/corr.py:29-50). */
int eraft_corr_lookup_bwd(const float* coords, const float* dout, int batch, int h, int w, float* dpyr0, float* dpyr1,
                          float* dpyr2, float* dpyr3, void* stream);

/* Adjoint of the pyramid construction: folds dpyr3 -> dpyr2 -> dpyr1 -> dpyr0 through the avg_pool2d chain (in place:
 * dpyr0..2 are modified) and returns d fmap1, d fmap2 [batch][c][h][w] of corr = fmap1^T fmap2 / sqrt(c).  h*w % 4 == 0.
 * Replaces: autograd of CorrBlock.__init__ / CorrBlock.corr  (model/corr.py:13-27,53-60). */
int eraft_corr_pyramid_bwd(const float* fmap1, const float* fmap2, float* dpyr0, float* dpyr1, float* dpyr2, const float* dpyr3,
                           int batch, int c, int h, int w, float* dfmap1, float* dfmap2, void* stream);

/* Adjoint of eraft_convex_upsample: dout [batch][2][8h][8w] -> dflow [batch][2][h][w], dmask [batch][576][h][w].
 * Replaces: autograd of ERAFT.upsample_flow  (model/eraft.py:83-94). */
int eraft_convex_upsample_bwd(const float* flow, const float* mask, const float* dout, int batch, int h, int w, float* dflow,
                              float* dmask, void* stream);

/* Convex upsampling: flow [batch][2][h][w], mask [batch][576][h][w] -> out [batch][2][8h][8w].
 * Replaces: ERAFT.upsample_flow  (model/eraft.py:83-94). */
int eraft_convex_upsample(eraft_ctx* ctx, const float* flow, const float* mask, int batch, int h, int w, float* out,
                          void* stream);

/* ------------------------------------------------------------------------------------------------
 * EEMFlow+ (EEMFlow_cdc, model/EEMFlow/EEMFlow+.py + cdc_utils.py): the coarse-to-fine bilinear flow-warp loop.
 * ---------------------------------------------------------------------------------------------- */

/* Replaces: EEMFlow_cdc.__init__ + .to(device)  (model/EEMFlow/EEMFlow+.py:75-135). */
int eemplus_create(int device, eemplus_ctx** out);
void eemplus_destroy(eemplus_ctx* ctx);

/* The 136 tensors of the reference state_dict() back to back in registration order (the parameters the
 * reference registers but never uses - up3..up6, cdc_model.upsample_output_conv - are present and skipped).
 * Replaces: load_state_dict of EEMFlow_cdc. */
int eemplus_load_weights(eemplus_ctx* ctx, const float* flat_host, size_t nfloats, int n_first_channels, int groups);

/* events1/2 [batch][C][in_h][in_w], pad = [left, right, top, bottom] of InputPadder(img_size, 'chairs', 64);
 * flow_out [5][batch][2][in_h][in_w]: the predictions of levels 6, 5, 4, 3, 2 at full resolution.
 * Replaces: EEMFlow_cdc.forward(events1, events2)[1]  (model/EEMFlow/EEMFlow+.py:158-234). */
int eemplus_forward(eemplus_ctx* ctx, const float* events1, const float* events2, int batch, int in_h, int in_w,
                    const int pad[4], float* flow_out, void* stream);

/* n (1..16) independent batch-1 samples, each in its own tensors (events1[i] / events2[i] [1][C][in_h][in_w], flow_out[i]
 * [5][1][2][in_h][in_w]), as ONE batch-n chain of launches - bitwise what eemplus_forward returns for the samples stacked into a batch.
 * Replaces: n iterations of the evaluation loop at batch 1 (test_mvsec.py:580-597 -> run_network, :1444-1455) for EEMFlow_cdc. */
int eemplus_forward_many(eemplus_ctx* ctx, int n, const float* const* events1, const float* const* events2, int in_h, int in_w,
                         const int pad[4], float* const* flow_out, void* stream);

/* Intermediates of the LAST forward: "flow2".."flow6" (low-resolution level flows, incl. the in-place doubling
 * the reference applies to flow3..flow6), "flow_up2".."flow_up5" and "flow_init2".."flow_init5" (cdc_model's upsampled
 * input flow of each level). */
int eemplus_get_stage(eemplus_ctx* ctx, const char* name, float* dst, size_t dst_capacity_floats, int dims_out[4],
                      void* stream);

/* Teacher-forced level: level l (5..2) of the coarse-to-fine loop, re-run on the feature pyramid of the LAST eemplus_forward
 * from a caller-supplied flow_init [batch][2][h_l][w_l] - cdc_model's upsampled input flow, the value the discontinuous
 * `grid_sample(ones) >= 1` mask of WarpingLayer_no_div is computed from.  flow_up_out / flow_out [batch][2][h_l][w_l] (either
 * may be NULL) receive flow_up_l (cdc_model's output) and flow_l (decoder_l + flow_up_l).  Pins every level to the flow
 * tolerance separately; the chained forward can only be compared statistically past the first mask.
 * Replaces: one l-block of EEMFlow_cdc.forward  (model/EEMFlow/EEMFlow+.py:183-229, cdc_utils.py:156-174). */
int eemplus_level(eemplus_ctx* ctx, int level, const float* flow_init, float* flow_up_out, float* flow_out, void* stream);

/* Throughput hint, as eraft_set_frames_in_flight (4-row tiles of the LDS-tiled convs from 512 blocks on): EEMFlow+ 1280x720 with
 * four frames in flight 579 -> 595 frames/s.  Default 1. */
int eemplus_set_frames_in_flight(eemplus_ctx* ctx, int n);

/* Backward bilinear warp of x [batch][c][h][w] by flow [batch][2][h][w].  mode 0: EEMFlow_cdc.warp
 * (EEMFlow+.py:137-149, align_corners=True); 1: tensor_tools.torch_warp (utils_luo/tools.py:2262-2306,
 * align_corners=False); 2: WarpingLayer_no_div (cdc_utils.py:50-78, align_corners=False and the
 * grid_sample(ones) >= 1 mask). */
int eemplus_warp(const float* x, const float* flow, int batch, int c, int h, int w, int mode, float* out, void* stream);

/* Adjoint of eemplus_warp: dout [batch][c][h][w] -> dx [batch][c][h][w] (zeroed here, scatter-added), dflow [batch][2][h][w].
 * The `>= 1` mask of mode 2 carries no gradient, as in the reference.  Replaces: autograd of F.grid_sample in the three
 * warps (EEMFlow+.py:137-149, cdc_utils.py:50-78, utils_luo/tools.py:2262-2306). */
int eemplus_warp_bwd(const float* x, const float* flow, const float* dout, int batch, int c, int h, int w, int mode, float* dx,
                     float* dflow, void* stream);

/* upsample2d_flow_as(inputs, target, 'bilinear', if_rate)  (cdc_utils.py:80-103): out [batch][2][oh][ow];
 * with if_rate != 0 `inputs` is scaled in place afterwards, as the reference does. */
int eemplus_upsample_flow_as(float* inputs, int batch, int h, int w, int oh, int ow, int if_rate, float* out,
                             void* stream);

/* ------------------------------------------------------------------------------------------------
 * Operator-level entry points (eemop_*): the differentiable building blocks of the E-RAFT training graph - what
 * torch.autograd records when train_mvsec.py:245-258 runs model(im1, im2) -> sequence_loss -> backward on model/eraft.py with the
 * module in train() (train-mode BatchNorm in cnet, model/extractor.py:31-35; SepConvGRU / motion encoder / heads of
 * model/update.py:6-106 unrolled 12 times).  All tensors are caller-owned dense NCHW fp32 on the current device;
 * eemflow_amd/ops.py wraps each pair as a torch.autograd.Function.
 * ---------------------------------------------------------------------------------------------- */

/* Packed-weight cache of the convolution operators below.  The kernels read weights in their own packed order; without a hint every
 * call repacks `w` (one small launch).  eemop_pack_hint names the weight tensor of the NEXT conv calls of this thread: `token` is an
 * identity the caller never reuses for another tensor (0 = no caching), `version` the tensor's modification counter (torch's
 * `Tensor._version`); a packing is redone only when (token, w pointer, stream) is new or the version differs - a recurrent model that
 * applies one set of weights twelve times per step (model/eraft.py:141-157) packs them once per step and once more for the data
 * gradients.  eemop_pack_forget frees what a token holds.  No counterpart in the reference: ATen's conv reads the weights in place. */
int eemop_pack_hint(long long token, long long version);
int eemop_pack_forget(long long token);
long long eemop_pack_cache_bytes(void);   /* device bytes the cache holds (all tokens, this process) */
/* conv2d of up to three channel-concatenated inputs (the torch.cat of model/update.py:44,51,79 is never materialised):
 * x_s [n][c_s][hin][win] (x1 / x2 may be NULL), w [cout][c0+c1+c2][kh][kw], bias [cout] or NULL; act 0 none, 1 ReLU, 2 sigmoid,
 * 3 tanh; out[n][out_coff + co][hout][wout] of an out_ctotal-channel tensor = out_scale * act(conv + bias).
 * Replaces: nn.Conv2d.forward (+ the activation that follows it) in model/extractor.py, model/update.py. */
int eemop_conv2d_fwd(const float* x0, int c0, const float* x1, int c1, const float* x2, int c2, const float* w, const float* bias,
                     int n, int hin, int win, int cout, int kh, int kw, int stride, int ph, int pw, int act, float out_scale, float* out,
                     int out_ctotal, int out_coff, void* stream);
/* d loss / d x for the input-channel slice [ci0, ci0 + cic) of a conv with cin input channels: dx [n][cic][hin][win] from
 * dy [n][cout][hout][wout] (the gradient w.r.t. the conv's output BEFORE its activation).  stride 1 or 2.
 * Replaces: autograd of nn.Conv2d w.r.t. its input. */
int eemop_conv2d_bwd_data(const float* dy, const float* w, int n, int hin, int win, int cin, int ci0, int cic, int cout, int kh, int kw,
                          int stride, int ph, int pw, float* dx, void* stream);
/* dw [cout][cin][kh][kw] (slice [ci0, ci0 + cic) of the input channels) += dy (x) x, db [cout] += sum dy (db may be NULL); x is
 * that slice's input [n][cic][hin][win].  Accumulating: the caller zeroes dw / db.  Kernel shapes built: 3x3, 1x1 (stride 1, 2),
 * 1x5, 5x1, 7x7 (stride 1, 2; at most 32 input channels).  Replaces: autograd of nn.Conv2d w.r.t. weight and bias. */
int eemop_conv2d_bwd_weight(const float* x, const float* dy, int n, int hin, int win, int cin, int ci0, int cic, int cout, int kh, int kw,
                            int stride, int ph, int pw, float* dw, float* db, void* stream);
/* The same for a conv over up to three channel-concatenated inputs in ONE call (x_s [n][c_s][hin][win]; x1 / x2 may be NULL):
 * dw [cout][c0 + c1 + c2][kh][kw] += dy (x) cat(x0, x1, x2), db [cout] += sum dy when not NULL.  The segments ride one launch where that
 * is the faster form (the GRU's (1, 5) / (5, 1) convs over [h | inp | motion]), one launch per segment elsewhere.
 * Replaces: autograd of nn.Conv2d w.r.t. its weight and bias for the convs of model/update.py:33-60,63-81 whose input is a torch.cat. */
int eemop_conv2d_bwd_weight_cat(const float* x0, int c0, const float* x1, int c1, const float* x2, int c2, const float* dy, int n, int hin,
                                int win, int cout, int kh, int kw, int stride, int ph, int pw, float* dw, float* db, void* stream);
/* out = scale * dy * act'(y) for y = act(pre): kind 1 ReLU, 2 sigmoid, 3 tanh, 0 identity.  Replaces: autograd of F.relu /
 * torch.sigmoid / torch.tanh (model/update.py:14,45-47,54-56,74-79,95; model/eraft.py:130-131) and of the 0.25 mask scale (:105). */
int eemop_act_bwd(const float* dy, const float* y, long long n, int kind, float scale, float* out, void* stream);
/* out = a + b (kind 0), a - b (1), a * b (2), relu(a + b) (3), alpha * a (4; b unused).  Replaces: coords1 - coords0, coords1 +
 * delta_flow (model/eraft.py:144,149), r * h (model/update.py:47,56), relu(x + y) (model/extractor.py:57). */
int eemop_binary(int kind, const float* a, const float* b, float alpha, long long n, float* out, void* stream);
/* out = in[0] + ... + in[count - 1] (1 <= count <= 8, host array of device pointers), one launch.  Replaces: the chain of adds autograd
 * runs when a tensor has several consumers - the hidden state, the context features, the motion features and the correlation pyramid
 * across the twelve unrolled iterations (model/eraft.py:139-157, model/update.py:43-60); Python: ops.FanOut. */
int eemop_sum_n(const float* const* in, int count, long long n, float* out, void* stream);
/* h' = (1 - z) h + z q and its adjoint (dz = dout (q - h), dh = dout (1 - z), dq = dout z).  Replaces: model/update.py:48,57. */
int eemop_gru_blend(const float* z, const float* h, const float* q, long long n, float* out, void* stream);
int eemop_gru_blend_bwd(const float* dout, const float* z, const float* h, const float* q, long long n, float* dz, float* dh, float* dq,
                        void* stream);
/* dst[n][dst_coff + c] = src[n][src_coff + c], c < cc: torch.cat([out, flow], dim=1) (model/update.py:81) and its split. */
int eemop_copy_channels(const float* src, int src_ctotal, int src_coff, float* dst, int dst_ctotal, int dst_coff, int cc, int n, int hw,
                        void* stream);
/* coords0 = coords1 = pixel grid [batch][2][h][w] (channel 0 = x, 1 = y), coords1 += flow_init when given.
 * Replaces: ERAFT.initialize_flow + the flow_init add  (model/eraft.py:73-81,136-137). */
int eemop_coords_init(float* coords0, float* coords1, const float* flow_init, int batch, int h, int w, void* stream);
/* F.pad(mode='replicate') of [nc][h][w]  (InputPadder.pad, utils/image_utils.py:139-140). */
int eemop_replicate_pad(const float* in, float* out, int nc, int h, int w, int left, int right, int top, int bottom, void* stream);
/* InstanceNorm2d(affine=False, eps 1e-5) over `planes` (n, c) planes of hw pixels: out = relu?(IN(x)) or relu(IN(x) + res) when res
 * is given; backward (x = the norm's input, y = its output) for the no-residual forms.
 * Replaces: nn.InstanceNorm2d (+ ReLU) of fnet, model/extractor.py:31-35,43-57, and its autograd. */
int eemop_instnorm_fwd(const float* x, const float* res, int planes, int hw, int relu, float* out, void* stream);
int eemop_instnorm_bwd(const float* x, const float* y, const float* dy, int planes, int hw, int relu, float* dx, void* stream);
/* BatchNorm2d in TRAINING mode: batch mean / biased variance over (n, hw) per channel, y = relu?((x - mean) rstd w + b); the running
 * statistics are updated IN PLACE (momentum, unbiased variance) and save_mean / save_rstd [c] kept for the backward, which returns
 * dx, dweight [c], dbias [c].  Replaces: nn.BatchNorm2d.forward in train() of cnet (model/extractor.py:31-35,123-133;
 * train_mvsec.py:231-235 leaves BatchNorm unfrozen) and its autograd. */
int eemop_batchnorm_train_fwd(const float* x, const float* weight, const float* bias, float* running_mean, float* running_var, int n,
                              int c, int hw, float momentum, float eps, int relu, float* y, float* save_mean, float* save_rstd,
                              void* stream);
int eemop_batchnorm_train_bwd(const float* x, const float* y, const float* dy, const float* weight, const float* save_mean,
                              const float* save_rstd, int n, int c, int hw, int relu, float* dx, float* dweight, float* dbias, void* stream);
/* BatchNorm2d in EVAL mode - frozen running statistics, an affine map per channel: y = relu?((x - running_mean) rstd w + b) with
 * rstd = 1 / sqrt(running_var + eps); backward returns dx = g w rstd, dweight [c] = sum g (x - running_mean) rstd, dbias [c] = sum g
 * (g = dy gated by the ReLU).  Replaces: nn.BatchNorm2d.forward in eval() after ERAFT.freeze_bn (model/eraft.py:69-72) and its autograd. */
int eemop_batchnorm_eval_fwd(const float* x, const float* weight, const float* bias, const float* running_mean, const float* running_var,
                             int n, int c, int hw, float eps, int relu, float* y, void* stream);
int eemop_batchnorm_eval_bwd(const float* x, const float* y, const float* dy, const float* weight, const float* running_mean,
                             const float* running_var, int n, int c, int hw, float eps, int relu, float* dx, float* dweight, float* dbias,
                             void* stream);
/* CorrBlock on caller tensors: the pyramid pyr_l [batch*h*w][h >> l][w >> l] (adjoint: eraft_corr_pyramid_bwd) and the 4-level 9x9
 * lookup at `coords` [batch][2][h][w] -> out [batch][324][h][w] (adjoint: eraft_corr_lookup_bwd).  model/corr.py:13-60. */
int eemop_corr_pyramid_fwd(const float* fmap1, const float* fmap2, int batch, int c, int h, int w, float* pyr0, float* pyr1, float* pyr2,
                           float* pyr3, void* stream);
int eemop_corr_lookup_fwd(const float* pyr0, const float* pyr1, const float* pyr2, const float* pyr3, const float* coords, int batch, int h,
                          int w, float* out, void* stream);
/* ERAFT.upsample_flow on caller tensors: flow [batch][2][h][w], mask [batch][576][h][w] -> out [batch][2][8h][8w]; `zeros` is an
 * all-zero [batch][2][h][w] tensor (adjoint: eraft_convex_upsample_bwd).  model/eraft.py:83-94. */
int eemop_convex_upsample_fwd(const float* zeros, const float* flow, const float* mask, int batch, int h, int w, float* out, void* stream);

/* ---- further operators of the EEMFlow+ autograd graph (model/EEMFlow/EEMFlow+.py:158-234, cdc_utils.py:50-177) */
/* out = act(x): 1 ReLU, 2 sigmoid, 3 tanh, 4 LeakyReLU(0.1)  (torch.sigmoid of the upsampler's mask, cdc_utils.py:166). */
int eemop_act_fwd(const float* x, long long n, int kind, float* out, void* stream);
/* channel_shuffle (EEMFlow+.py:52-58) of [n][c][hw] and its inverse (= its adjoint). */
int eemop_shuffle_channels(const float* src, float* dst, int n, int c, int groups, int hw, int inverse, void* stream);
/* out[:, 0] = su * x[:, 0], out[:, 1] = sv * x[:, 1] of a flow [n][2][hw]: the rate of upsample2d_flow_as (cdc_utils.py:85-93);
 * its own adjoint. */
int eemop_scale_flow(const float* x, int n, int hw, float su, float sv, float* out, void* stream);
/* F.avg_pool2d(2, 2) over the last two dims of [planes][h][w] and its adjoint  (EEMFlow+.py:170-175). */
int eemop_pool2_fwd(const float* x, float* out, long long planes, int h, int w, void* stream);
int eemop_pool2_bwd(const float* dy, float* dx, long long planes, int h, int w, void* stream);
/* F.interpolate(mode='bilinear', align_corners=True) of [nc][h][w] -> [nc][oh][ow] and its adjoint  (cdc_utils.py:83). */
int eemop_resize_ac_fwd(const float* in, float* out, int nc, int h, int w, int oh, int ow, void* stream);
int eemop_resize_ac_bwd(const float* dout, float* dx, int nc, int h, int w, int oh, int ow, void* stream);
/* adjoint of eemflow_local_corr53: dcv [batch][53][h][w] -> df1, df2 [batch][c][h][w]. */
int eemop_local_corr53_bwd(const float* dcv, const float* f1, const float* f2, int batch, int c, int h, int w, float* df1, float* df2,
                           void* stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* EEMFLOW_HIP_H */
