// Internals shared by api.hip (inference C ABI) and train_api.hip (training step): the context, workspace
// management and the forward schedule.  Everything here has internal linkage.
#pragma once
#include <stdarg.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/eemflow_hip.h"
#include "common.h"
#include "gconv.h"
#include "eraft_kernels.h"
#include "wnc.h"

// ------------------------------------------------------------------------------- context
namespace {

const int kTaps53[53] = {0,  2,  4,  6,  8,  10, 12, 14, 16, 18, 20, 21, 22, 23, 24, 26, 28, 29,
                         30, 31, 32, 33, 34, 36, 38, 39, 40, 41, 42, 44, 46, 47, 48, 49, 50, 51,
                         52, 54, 56, 57, 58, 59, 60, 62, 64, 66, 68, 70, 72, 74, 76, 78, 80};
constexpr int kNTaps = 53;
constexpr int kDecIn = kNTaps + 16;   // 69
constexpr int kDecW = 100;

struct TailW {                         // one packed small-grid conv
    size_t wpk = 0, bias = 0;          // float offsets into the weight arena
    int cin = 0, cout = 0, ksize = 3;
};

struct DevBuf {
    float* p = nullptr;
    size_t cap = 0;                    // floats
};

struct Shape {
    int batch = 0, in_h = 0, in_w = 0, out_h = 0, out_w = 0;
    int hp = 0, wp = 0;                // padded extent
    int h1 = 0, w1 = 0, h2 = 0, w2 = 0, h3 = 0, w3 = 0;
    int gh = 0, gw = 0;                // 1/64 grid
    // fused stage pooling (fast path): partial-sum buffer dims per stage, fuse[k] = conv epilogue pools stage k
    bool fuse[3] = {false, false, false};
    int prow[3] = {0, 0, 0}, pcol[3] = {0, 0, 0}, th[3] = {0, 0, 0};
};

}  // namespace

struct eemflow_ctx {
    int device = 0;
    bool weights_loaded = false;
    int cin0 = 5, groups = 5;
    // padder
    bool have_pad = false;
    int pad[4] = {0, 0, 0, 0};
    // weights: `flat` is the device-resident master copy in state_dict order; `arena` holds every packed
    // form the kernels read (MFMA fragment orders, biases, transposed weights for the data gradients) and is
    // rebuilt from `flat` by one gather kernel driven by `pack_idx` (arena[i] = flat[pack_idx[i]-1], 0 -> 0.f)
    float* flat = nullptr;
    size_t nflat = 0;
    int* pack_idx = nullptr;
    size_t arena_floats = 0;
    float* arena = nullptr;
    size_t enc_w[ENC_NUM], enc_w2[ENC_NUM], enc_b[ENC_NUM];
    bool enc_has2[ENC_NUM];
    // Winograd-domain weights of the stride-1 C->C encoder layers, both forms (F(2x2,3x3): conv_wino.hip / conv_wino32.hip,
    // F(4x4,3x3): conv_wino4.hip): computed from `flat` by wino_transform_launch when a launch first needs them after a (re)pack.
    // Slot [f4 * 2 + dir][l]: dir 0 forward weights, dir 1 W^T flipped (data gradient)
    float* wino = nullptr;
    size_t wino_off[4][ENC_NUM];
    bool wino_ok[4][ENC_NUM] = {};
    size_t s2r_off[ENC_NUM];           // the stride-2 layers' weights in conv_s2r.hip's order (same buffer, same lazy refresh)
    bool s2r_ok[ENC_NUM] = {};
    bool enc_s2r[ENC_NUM] = {};
    size_t bx3_off[ENC_NUM];           // pconv2_1's weights as pre-split bf16 fragments (conv_bx3.hip; same buffer, same lazy refresh)
    bool bx3_ok[ENC_NUM] = {};
    bool enc_bx3[ENC_NUM] = {};
    bool enc_wino[ENC_NUM];
    bool use_wino = true;              // EEM_WINO=0 in the environment keeps the direct-convolution kernels
    // which stride-1 layers run F(4x4,3x3), by channel count (bit 0: C = 16, 1: C = 32, 2: C = 64).  F(4x4) blocks are 8 waves on 16 x
    // {128, 64, 32} pixel tiles: 240 / 120 / 60 blocks per frame at 1280x720 - the C = 32 / 64 layers then occupy half / a quarter of
    // the chip for longer (24 / 38 us against 17.6 / 15.3) but cost 0.68 / 0.62 of the CU time, so they are the throughput choice
    // (several frames in flight: eemflow_set_frames_in_flight >= 3) and F(2x2) the latency choice; C = 16 wins both ways.
    // EEM_WINO=2: never; EEM_WINO4_LAYERS=<mask>: always that mask
    int f4_mask_env = -1;
    // (a forward of four or more samples has the tiles to fill the chip with F(4x4) blocks too: 8 260 against 7 400 frames/s for
    // batches of four, two in flight)
    int f4_mask(int batch) const {
        if (f4_mask_env >= 0) return f4_mask_env;
        return (frames_in_flight >= 3 || batch >= 4) ? 7 : 1;
    }
    bool layer_f4(int cin, int batch) const { return (f4_mask(batch) >> (cin == 16 ? 0 : cin == 32 ? 1 : 2)) & 1; }
    float* zero_page = nullptr;
    // the decoders' two wide 3x3 layers (conv1 69 -> 100, conv5 100 -> 64; EEMFlow.py:38-71) on the Winograd F(2x2) kernel of conv_wnc.hip:
    // their streams and slice biases, made from `flat` on the device (ensure_dec_wnc) whenever the weights have changed there
    float* dec_wnc = nullptr;
    bool dec_wnc_ok = false;
    size_t dec_w1[3][4] = {}, dec_w5[3][2] = {}, dec_b1[3] = {}, dec_b5[3] = {};      // float offsets into dec_wnc
    TailW rconv[3], dconv1[3], dgroup[3][3][5], dconv5[3], dconv6[3], dconv7[3], outc;
    // training: per-conv descriptors (flat offsets of weight/bias, packed transposed weights for gconv dgrad)
    struct ConvRef {
        size_t w = 0, b = 0, wT = 0;                 // flat offsets of weight / bias; arena offset of gconv-packed W^T
        size_t wT_enc = 0, wT_enc2 = 0, zero_bias = 0;   // stride-1 encoder layers: W^T packed for the encoder kernels
        size_t wT_tail = 0;                              // tail convs: W^T packed for tail_conv_kernel (batched data gradients)
        bool has_tail = false;
        bool fast_dgrad = false;
        int cin = 0, cout = 0, k = 3, stride = 1;
    };
    ConvRef t_enc[ENC_NUM], t_rconv[3], t_dconv1[3], t_dgroup[3][3][5], t_dconv5[3], t_dconv6[3], t_dconv7[3], t_outc;
    // training workspace + optimizer state
    DevBuf padded, g_a1, g_f11, g_a2, g_b2, g_f12, g_a3, g_b3, g_f13, g_pool[3], g_cat[3], g_ta[3], g_tb[3], g_tc[3], g_td[3],
        g_t64[3], g_t32[3], g_flowcat, g_coarse, g_flow, ups_tmp, grad_flat, adam_m, adam_v, scalars;
    long opt_step = 0;
    // eemflow_forward_train / eemflow_backward pairing: the serial of the forward whose activations the workspace holds
    const float *train_e1 = nullptr, *train_e2 = nullptr;
    bool have_train_fwd = false;
    long train_serial = 0;
    Shape train_shape;                                   // the shape of THAT forward (`last` follows every forward, inference included)
    // every entry point that writes the shared workspace without keeping the activations calls this: a pending backward then finds a
    // newer serial and its caller recomputes the forward (eemflow.py) instead of differentiating somebody else's activations
    void workspace_overwritten() { have_train_fwd = false; train_serial += 1; }
    // backward pass: weight / bias gradients are leaves of the chain of data gradients, so they run on this context-owned side stream
    // (fork: an event after the gradient they read; join: the caller's stream waits for the last one before backward returns)
    hipStream_t wstream = nullptr;
    static constexpr int kWEvents = 32;
    hipEvent_t wev[kWEvents] = {};
    hipEvent_t wjoin = nullptr;
    hipEvent_t prep_ev = nullptr;                        // the side-stream prologue of a training forward (train_api.hip: forward_train_impl)
    bool sumsq_zeroed = false;                           // the optimizer's sum-of-squares cell: cleared once by a fill, then by each step's re-packing launch
    int wev_next = 0;
    int* taps = nullptr;
    // workspaces
    DevBuf a1, f11, a2, b2, f12, a3, b3, f13, pool[3], ppart[3], cat[3], ta[3], tb[3], tc[3], td[3], t64[3], t32[3], flowcat, coarse;
    Shape last;
    bool have_last = false;
    // graph cache: up to kMaxGraphs captured forwards keyed on SHAPES only.  The two launches that touch caller buffers (the
    // first conv, the upsample) read their pointers from `io_table` (device: {events1, events2, flow_out}), which one tiny
    // launch rewrites in front of a replay whenever the caller hands over other buffers - fresh tensors per frame replay the
    // same graph.  Every entry bakes in workspace pointers: a reallocation (ensure) drops them all.
    bool use_graph = true;
    // diagnostic (EEM_SPANS=1, eager launches): events at frame start / encoder end / frame end; every 64 frames the averages of the
    // encoder chain's and the tail chain's spans under whatever else runs on the chip go to stderr (tools/spans.sh)
    hipEvent_t span_ev[3] = {nullptr, nullptr, nullptr};
    double span_sum[2] = {0.0, 0.0};
    int span_n = 0;
    bool span_pending = false;
    bool skip_counter_zeroed = false;                    // train_api.hip: the device-side count of skipped optimizer steps
    // eemflow_train_stats_async / _wait: the loss statistics of a step on their way to pinned host memory behind an event, so that the
    // host can enqueue the optimizer step (and the next forward) before it reads them
    double* stats_host = nullptr;
    hipEvent_t stats_ev = nullptr;
    bool stats_pending = false;
    hipStream_t cstream = nullptr;                       // the statistics' copy stream: the loss sums leave right behind the loss kernel
    hipEvent_t loss_ev = nullptr;                        // (round 6), not behind the whole backward
    double stats_scale = 0.0;                            // gamma weight / (B * 2 * out_h * out_w) of the forward they belong to
    // inference leaves f13 unwritten when pconv3_3's epilogue pools it (nothing else reads it); the training forward keeps every
    // activation (keep_stage_stores), and eemflow_get_stage("f13") re-runs the layer with stores when the last forward skipped them
    bool enc0_generic = false;                           // n_first_channels != 5: pconv1_1 = replicate-pad launch + gconv.hip
    size_t enc0_gw = 0;
    bool keep_stage_stores = false;
    bool f13_skipped = false;
    // pconv1_1 computed inside pconv1_2's block (conv_enc12.hip; inference, 5-bin first layer; OPT-IN: EEM_FUSE12=1, set before the context sizes its buffers):
    // `a1` is then never written - eemflow_get_stage("a1") re-runs pconv1_1 alone on the last call's event volumes
    DevBuf fuse_scratch;
    bool a1_skipped = false;
    const float* last_e1 = nullptr;                      // the last forward's caller buffers (contiguous form) / its io-table form
    const float* last_e2 = nullptr;
    int last_io_frames = 0;
    int frames_in_flight = 1;                            // eemflow_set_frames_in_flight: >= 3 shrinks the persistent encoder grids
    // eemflow_set_deferred_input_norm: the event volumes handed to forward / forward_many are RAW voxel grids with their normalisation
    // record behind them (eemflow_voxelize*, normalize = 2); pconv1_1 normalises as it reads
    bool deferred_norm = false;
    struct Key {
        int batch, in_h, in_w, out_h, out_w, pad[4];
        int aligned16;                                   // all three caller buffers 16-byte aligned (kernel selection depends on it)
        int io_frames;                                   // 0: one batch in contiguous tensors; n: n single-frame buffer triples (eemflow_forward_many)
        int deferred_norm;
        bool operator==(const Key& o) const {
            return batch == o.batch && in_h == o.in_h && in_w == o.in_w && out_h == o.out_h && out_w == o.out_w &&
                   pad[0] == o.pad[0] && pad[1] == o.pad[1] && pad[2] == o.pad[2] && pad[3] == o.pad[3] && aligned16 == o.aligned16 &&
                   io_frames == o.io_frames && deferred_norm == o.deferred_norm;
        }
    };
    struct GraphEntry {
        Key key;
        Shape shape;
        bool f13_skipped = false;                        // the captured schedule leaves f13 unwritten (every replay does, then)
        bool a1_skipped = false;                         // ... and a1 (the fused first two layers)
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        long last_use = 0;
    };
    static constexpr int kMaxGraphs = 4;
    std::vector<GraphEntry> graphs;
    long graph_clock = 0;
    const void** io_table = nullptr;                     // device: 3 * EEM_MAX_COALESCE pointers
    const void* io_host[3 * EEM_MAX_COALESCE] = {};      // what the table holds once the launches issued so far have run
    int io_host_n = 0;                                   // entries of io_host in use (3: the contiguous form)
    int cur_io_frames = 0;                               // the schedule being issued reads per-frame triples (set around run_forward)
    void* io_stream = nullptr;                           // stream of the last table write / replay
    long graph_captures = 0, graph_replays = 0, io_updates = 0;   // statistics (eemflow_graph_stats)
};

namespace {

// bumped whenever ensure() moves a buffer: cached graphs hold workspace pointers (see alloc_workspace)
static thread_local unsigned long g_realloc_events = 0;

int ensure(DevBuf& b, size_t floats) {
    if (floats <= b.cap) return EEM_OK;
    ++g_realloc_events;
    if (b.p) EEM_HIP_CHECK(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    EEM_HIP_CHECK(hipMalloc(&b.p, floats * sizeof(float)));
    b.cap = floats;
    return EEM_OK;
}

// The device-resident flat weights changed (load / optimizer step): every Winograd-domain copy is stale; a launch recomputes the one it
// needs (ensure_wino) - a training step touches two forms of five layers, an inference loop one
int refresh_wino(eemflow_ctx* c, hipStream_t) {
    for (int f = 0; f < 4; ++f)
        for (int l = 0; l < ENC_NUM; ++l) c->wino_ok[f][l] = false;
    for (int l = 0; l < ENC_NUM; ++l) c->s2r_ok[l] = false;
    for (int l = 0; l < ENC_NUM; ++l) c->bx3_ok[l] = false;
    c->dec_wnc_ok = false;
    return EEM_OK;
}
// The packed forms a launch can actually take (the opt-in kernels' switches are read per launch - conv_s2r.hip, conv_bx3.hip - and so are
// these): after every optimizer step the packed copies are stale, and a transform nobody reads was six ~5 us launches in the chain of
// a training step's forward
inline bool s2r_wanted() { const char* on = getenv("EEM_S2R"); return on && on[0] == '1'; }
inline bool bx3_wanted(int l) {
    const EncLayerDesc& d = kEncLayers[l];
    if (d.stride == 2) return true;
    const char* m = getenv("EEM_BX3_S1");
    return ((m ? atoi(m) : 0) & (d.cin == 32 ? 1 : 2)) != 0;
}
int ensure_bx3(eemflow_ctx* c, int l, hipStream_t st, const float** w_out) {
    if (!c->bx3_ok[l]) {
        const int rc = bx3_transform_launch(c->flat + c->t_enc[l].w, kEncLayers[l].cin, kEncLayers[l].cout, c->wino + c->bx3_off[l], st);
        if (rc != EEM_OK) return rc;
        c->bx3_ok[l] = true;
    }
    *w_out = c->wino + c->bx3_off[l];
    return EEM_OK;
}
int ensure_s2r(eemflow_ctx* c, int l, hipStream_t st, const float** w_out) {
    if (!c->s2r_ok[l]) {
        const int rc = s2r_transform_launch(c->flat + c->t_enc[l].w, kEncLayers[l].cin, kEncLayers[l].cout, c->wino + c->s2r_off[l], st);
        if (rc != EEM_OK) return rc;
        c->s2r_ok[l] = true;
    }
    *w_out = c->wino + c->s2r_off[l];
    return EEM_OK;
}
// Winograd-domain weights of layer l (dir 0: forward, 1: data gradient) in the form the policy picks; *f4_out says which
int ensure_wino(eemflow_ctx* c, int l, int dir, int batch, hipStream_t st, const float** w_out, int* f4_out) {
    const int ch = kEncLayers[l].cin;
    const int f4 = c->layer_f4(ch, batch) ? 1 : 0;
    const int slot = f4 * 2 + dir;
    if (!c->wino_ok[slot][l]) {
        const int rc = wino_transform_launch(c->flat + c->t_enc[l].w, ch, dir, c->wino + c->wino_off[slot][l], st, f4);
        if (rc != EEM_OK) return rc;
        c->wino_ok[slot][l] = true;
    }
    *w_out = c->wino + c->wino_off[slot][l];
    *f4_out = f4;
    return EEM_OK;
}
// The decoders' conv1 / conv5 on the Winograd kernel: wanted for grids whose rows are 16-byte multiples (1280x720: 12 x 20 cells; MVSEC's
// 5 x 6 stays on the small-grid kernel) from four samples per launch on - the rule of the encoder's F(4x4) forms (f4_mask): one frame alone
// is three 20-us tiles per (decoder, slice) where the small-grid kernel needs 5 - 9 us (`latency_ms_b1` 0.192 -> 0.207 ms with the kernel
// at every batch).  EEM_DEC_WNC (read per call): 0 off, 1 at every batch (tests that compare a batch with its shards pin the form), unset: by batch.
// The training forward takes the same rule.
inline bool dec_wnc_wanted(const eemflow_ctx* c, int gw, int batch) {
    const char* e = getenv("EEM_DEC_WNC");
    if (c->dec_wnc == nullptr || gw % 4 != 0 || (e && e[0] == '0')) return false;
    return (e && e[0] == '1') || batch >= 4;
}
int ensure_dec_wnc(eemflow_ctx* c, hipStream_t st) {
    if (c->dec_wnc_ok || !c->dec_wnc) return EEM_OK;
    WncPackArgs a;
    memset(&a, 0, sizeof(a));
    for (int k = 0; k < 3; ++k) {
        for (int s = 0; s < 4; ++s)
            a.job[a.njobs++] = {c->flat + c->t_dconv1[k].w, c->flat + c->t_dconv1[k].b, c->dec_wnc + c->dec_w1[k][s], c->dec_wnc + c->dec_b1[k] + 32 * s,
                                kDecW, kDecIn, 32 * s, 0};
        for (int s = 0; s < 2; ++s)
            a.job[a.njobs++] = {c->flat + c->t_dconv5[k].w, c->flat + c->t_dconv5[k].b, c->dec_wnc + c->dec_w5[k][s], c->dec_wnc + c->dec_b5[k] + 32 * s,
                                64, kDecW, 32 * s, 0};
    }
    const int rc = wnc_pack_device_launch(a, st);
    if (rc != EEM_OK) return rc;
    c->dec_wnc_ok = true;
    return EEM_OK;
}
// a training step's Winograd weights - forward and data-gradient forms of every F(4x4) layer - refreshed by ONE launch in front of its
// forward (ensure_wino then finds them valid); the F(2x2) forms stay with ensure_wino
int ensure_train_wino(eemflow_ctx* c, int batch, hipStream_t st) {
    if (!c->use_wino) return EEM_OK;
    const float* w[W4_WT_JOBS]; float* out[W4_WT_JOBS]; int ch[W4_WT_JOBS], flip[W4_WT_JOBS];
    int slot_of[W4_WT_JOBS], layer_of[W4_WT_JOBS], n = 0;
    for (int l = 0; l < ENC_NUM; ++l) {
        if (!c->enc_wino[l] || !c->layer_f4(kEncLayers[l].cin, batch)) continue;
        for (int dir = 0; dir < 2; ++dir) {
            const int slot = 2 + dir;
            if (c->wino_ok[slot][l] || n == W4_WT_JOBS) continue;
            w[n] = c->flat + c->t_enc[l].w; out[n] = c->wino + c->wino_off[slot][l]; ch[n] = kEncLayers[l].cin; flip[n] = dir;
            slot_of[n] = slot; layer_of[n] = l; ++n;
        }
    }
    if (n == 0) return EEM_OK;
    const int rc = wino4_transform_multi_launch(w, ch, flip, out, n, st);
    if (rc != EEM_OK) return rc;
    for (int i = 0; i < n; ++i) c->wino_ok[slot_of[i]][layer_of[i]] = true;
    return EEM_OK;
}

// before a graph capture / replay: the forward copies exist (a transform launched inside a capture would replay with every frame)
int ensure_forward_wino(eemflow_ctx* c, int batch, hipStream_t st) {
    { const int rcd = ensure_dec_wnc(c, st); if (rcd != EEM_OK) return rcd; }
    for (int l = 0; l < ENC_NUM; ++l) {
        const float* ws;
        if (c->enc_s2r[l] && s2r_wanted()) { const int rc = ensure_s2r(c, l, st, &ws); if (rc != EEM_OK) return rc; }
        if (c->enc_bx3[l] && bx3_wanted(l)) { const int rc = ensure_bx3(c, l, st, &ws); if (rc != EEM_OK) return rc; }
    }
    if (!c->use_wino) return EEM_OK;
    for (int l = 0; l < ENC_NUM; ++l) {
        if (!c->enc_wino[l]) continue;
        const float* w; int f4;
        const int rc = ensure_wino(c, l, 0, batch, st, &w, &f4);
        if (rc != EEM_OK) return rc;
    }
    return EEM_OK;
}

void drop_graph(eemflow_ctx* c) {
    for (eemflow_ctx::GraphEntry& e : c->graphs) {
        if (e.exec) (void)hipGraphExecDestroy(e.exec);
        if (e.graph) (void)hipGraphDestroy(e.graph);
    }
    c->graphs.clear();
}

int compute_shape(eemflow_ctx* c, int batch, int in_h, int in_w, int out_h, int out_w, Shape* s) {
    s->batch = batch; s->in_h = in_h; s->in_w = in_w; s->out_h = out_h; s->out_w = out_w;
    s->hp = in_h + c->pad[2] + c->pad[3];
    s->wp = in_w + c->pad[0] + c->pad[1];
    auto half = [](int v) { return (v - 1) / 2 + 1; };          // conv k3 s2 p1
    s->h1 = half(s->hp); s->w1 = half(s->wp);
    s->h2 = half(s->h1); s->w2 = half(s->w1);
    s->h3 = half(s->h2); s->w3 = half(s->w2);
    s->gh = s->h1 / 32; s->gw = s->w1 / 32;
    EEM_REQUIRE(s->gh >= 1 && s->gw >= 1, "input %dx%d (padded %dx%d) is too small for the 1/64 grid", in_h, in_w,
                s->hp, s->wp);
    // the reference concatenates the three decoders' flows (EEMFlow.py:179): the three pooled grids
    // must agree or torch.cat raises
    EEM_REQUIRE(s->h2 / 16 == s->gh && s->h3 / 8 == s->gh && s->w2 / 16 == s->gw && s->w3 / 8 == s->gw,
                "pooled grids of the three stages differ for padded size %dx%d (the reference's torch.cat "
                "fails too)", s->hp, s->wp);
    // stage pooling can ride in the epilogue of pconv1_2 / pconv2_3 / pconv3_3 when those run the fast path
    const int last[3] = {ENC_1_2, ENC_2_3, ENC_3_3};
    const int hs[3] = {s->h1, s->h2, s->h3}, ws[3] = {s->w1, s->w2, s->w3}, ks[3] = {32, 16, 8};
    for (int k = 0; k < 3; ++k) {
        const EncLayerDesc& d = kEncLayers[last[k]];
        int th, tw, pk;
        const bool wino = c->use_wino && c->enc_wino[last[k]] && wino_supported(d.cin, d.cout, d.stride, ws[k]);
        if (wino) wino_tile(d.cin, c->layer_f4(d.cin, batch) ? 1 : 0, &th, &tw, &pk);
        else enc2_tile(d.cin, d.cout, &th, &tw, &pk);
        s->fuse[k] = (wino || (c->enc_has2[last[k]] && enc2_supported(d.cin, d.cout, d.stride, ws[k]))) && pk == ks[k];
        s->th[k] = th;
        s->prow[k] = ceil_div(hs[k], th);
        s->pcol[k] = ceil_div(ws[k], tw) * (tw / ks[k]);
    }
    return EEM_OK;
}

int alloc_workspace_raw(eemflow_ctx* c, const Shape& s);
// workspace for shape s; cached graphs survive unless a buffer had to move
int alloc_workspace(eemflow_ctx* c, const Shape& s) {
    const unsigned long before = g_realloc_events;
    const int rc = alloc_workspace_raw(c, s);
    if (g_realloc_events != before) drop_graph(c);
    return rc;
}

int alloc_workspace_raw(eemflow_ctx* c, const Shape& s) {
    const size_t n2 = 2 * (size_t)s.batch, B = s.batch, g = (size_t)s.gh * s.gw;
    int rc;
#define ENS(buf, n) if ((rc = ensure(buf, n)) != EEM_OK) return rc
    ENS(c->a1, n2 * 16 * s.h1 * s.w1);  ENS(c->f11, n2 * 16 * s.h1 * s.w1);
    ENS(c->a2, n2 * 32 * s.h2 * s.w2);  ENS(c->b2, n2 * 32 * s.h2 * s.w2);  ENS(c->f12, n2 * 32 * s.h2 * s.w2);
    ENS(c->a3, n2 * 64 * s.h3 * s.w3);  ENS(c->b3, n2 * 64 * s.h3 * s.w3);  ENS(c->f13, n2 * 64 * s.h3 * s.w3);
    const int pc[3] = {16, 32, 64};
    for (int k = 0; k < 3; ++k) {
        ENS(c->pool[k], n2 * pc[k] * g);
        if (s.fuse[k]) ENS(c->ppart[k], n2 * pc[k] * (size_t)s.prow[k] * s.pcol[k]);
        ENS(c->cat[k], B * kDecIn * g);
        ENS(c->ta[k], B * kDecW * g);   ENS(c->tb[k], B * kDecW * g);
        ENS(c->tc[k], B * kDecW * g);   ENS(c->td[k], B * kDecW * g);
        ENS(c->t64[k], B * 64 * g);     ENS(c->t32[k], B * 32 * g);
    }
    ENS(c->flowcat, B * 6 * g);  ENS(c->coarse, B * 2 * g);
    if (c->enc0_generic) { ENS(c->padded, n2 * c->cin0 * (size_t)s.hp * s.wp); }
    else {                                                   // block scratch of the OPT-IN fused first two layers only (41 MB)
        const char* eon = getenv("EEM_FUSE12");
        if (eon && eon[0] == '1') { ENS(c->fuse_scratch, enc12_scratch_floats(256)); }
    }
#undef ENS
    return EEM_OK;
}

TailConvJob make_job(const eemflow_ctx* c, const TailW& w, const float* in, int in_ctotal, int in_coff, float* out,
                     int out_ctotal, int out_coff, int out_cmul, int act) {
    TailConvJob j;
    j.in = in; j.wpk = c->arena + w.wpk; j.bias = c->arena + w.bias; j.out = out;
    j.cin = w.cin; j.cout = w.cout;
    j.in_ctotal = in_ctotal; j.in_coff = in_coff;
    j.out_ctotal = out_ctotal; j.out_coff = out_coff; j.out_cmul = out_cmul; j.act = act;
    j.gate = nullptr; j.in_cmul = 1; j.add = nullptr;
    return j;
}

// Every kernel launch of the schedule goes through a Hook: normally it just launches; in timing mode (eemflow_time_kernels) every
// launch of a pass is bracketed by its own pair of HIP events on the launch stream - the schedule runs as the CHAIN it is, each kernel
// behind its producer, `reps` passes - and the durations are averaged per launch with its algorithmic FLOPs / bytes.  (Until round 4 a
// kernel was repeated back to back instead: at ten frames per launch that read 11 % slow for the layers whose input the launch before
// had just left in the Infinity Cache - the rocprofv3 per-kernel averages of the timed loop said so.)
struct Hook {
    hipStream_t st = nullptr;
    bool timing = false;
    int reps = 1;
    bool repeat = false;                                 // timing, the other form: each kernel `reps` times back to back between ONE pair of events
    int pass = 0;                                        // timing: pass being issued
    size_t slot = 0;                                     // timing: launch index inside the pass
    std::vector<hipEvent_t> evs;                         // timing: two events per launch of a pass, reused pass after pass
    std::vector<eemflow_kernel_stat> stats;

    // diagnostic BUILDS only (-DEEM_DIAG: `EEM_BUILD_TAG=diag EEM_EXTRA_FLAGS=-DEEM_DIAG python -m eemflow_amd.build`, loaded through
    // EEM_LIB_PATH; the release library does not read these variables): EEM_SKIP_KERNELS="enc.pconv2_1;dec." skips the launches whose
    // name starts with one of the prefixes - the flow is garbage, the frame rate says what that launch costs BESIDE the others
    // (tools/marginal.sh); read once per process
#ifndef EEM_DIAG
    static bool skipped(const char*) { return false; }
#else
    static bool skipped(const char* name) {
        static const std::string list = [] {
            const char* e = getenv("EEM_SKIP_KERNELS");
            if (e && e[0]) fprintf(stderr, "eemflow_hip: EEM_SKIP_KERNELS=\"%s\" is set - the launches it names are skipped and the flow is GARBAGE (diagnostic runs only)\n", e);
            return std::string(e ? e : "");
        }();
        if (list.empty()) return false;
        size_t pos = 0;
        while (pos <= list.size()) {
            size_t end = list.find(';', pos);
            if (end == std::string::npos) end = list.size();
            if (end > pos && strncmp(name, list.c_str() + pos, end - pos) == 0) return true;
            pos = end + 1;
        }
        return false;
    }
#endif

    template <class F>
    int run(const char* name, double flops, double bytes, F&& launch) {
#ifdef EEM_DIAG
        if (skipped(name)) {                                  // EEM_SKIP_SPIN_US=<us>: one lane holds the launch's place in the stream for <us>
            static const float spin = [] { const char* e = getenv("EEM_SKIP_SPIN_US"); return e ? (float)atof(e) : 0.f; }();
            return spin > 0.f ? spin_launch(spin, st) : EEM_OK;
        }
#endif
        if (!timing) return launch(st);
        if (repeat) {
            // (a launch of a few microseconds is mostly the gap an event pair adds around it: the single-frame table repeats each kernel
            // instead - its inputs then come from whatever cache the repetition before left them in, which at one frame per launch is
            // where the launch before it in the chain leaves them too)
            if (evs.size() < 2) { evs.resize(2, nullptr); EEM_HIP_CHECK(hipEventCreate(&evs[0])); EEM_HIP_CHECK(hipEventCreate(&evs[1])); }
            eem_last_grid_blocks = eem_last_grid_threads = eem_last_pipe = 0;
            int rc = launch(st);
            if (rc != EEM_OK) return rc;
            EEM_HIP_CHECK(hipEventRecord(evs[0], st));
            for (int i = 0; i < reps; ++i)
                if ((rc = launch(st)) != EEM_OK) return rc;
            EEM_HIP_CHECK(hipEventRecord(evs[1], st));
            EEM_HIP_CHECK(hipEventSynchronize(evs[1]));
            float ms = 0.f;
            EEM_HIP_CHECK(hipEventElapsedTime(&ms, evs[0], evs[1]));
            eemflow_kernel_stat ks;
            memset(&ks, 0, sizeof(ks));
            strncpy(ks.name, name, sizeof(ks.name) - 1);
            ks.flops = flops; ks.bytes = bytes; ks.ms = ms;      // (divided by reps by the caller, like the chain form's sums)
            ks.blocks = eem_last_grid_blocks;
            ks.pipe = eem_last_pipe;
            stats.push_back(ks);
            return EEM_OK;
        }
        if (evs.size() < 2 * (slot + 1)) {
            evs.resize(2 * (slot + 1), nullptr);
            EEM_HIP_CHECK(hipEventCreate(&evs[2 * slot]));
            EEM_HIP_CHECK(hipEventCreate(&evs[2 * slot + 1]));
        }
        eem_last_grid_blocks = eem_last_grid_threads = eem_last_pipe = 0;
        EEM_HIP_CHECK(hipEventRecord(evs[2 * slot], st));
        const int rc = launch(st);
        if (rc != EEM_OK) return rc;
        EEM_HIP_CHECK(hipEventRecord(evs[2 * slot + 1], st));
        if (pass == 0) {
            eemflow_kernel_stat ks;
            memset(&ks, 0, sizeof(ks));
            strncpy(ks.name, name, sizeof(ks.name) - 1);
            ks.flops = flops; ks.bytes = bytes; ks.ms = 0.f;
            ks.blocks = eem_last_grid_blocks;                     // 0: a launcher that does not report its grid
            ks.pipe = eem_last_pipe;
            stats.push_back(ks);
        }
        ++slot;
        return EEM_OK;
    }
    // after a pass has been issued: wait for it and add its durations (pass 0 is the warm-up and is not counted when reps > 1)
    int collect(bool count) {
        for (size_t i = 0; i < slot && i < stats.size(); ++i) {
            EEM_HIP_CHECK(hipEventSynchronize(evs[2 * i + 1]));
            float ms = 0.f;
            EEM_HIP_CHECK(hipEventElapsedTime(&ms, evs[2 * i], evs[2 * i + 1]));
            if (count) stats[i].ms += ms;
        }
        slot = 0;
        ++pass;
        return EEM_OK;
    }
    void release() {
        for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
        evs.clear();
    }
};

double tail_flops(const TailConvLaunch& L) {
    double f = 0;
    for (int i = 0; i < L.njobs; ++i)
        f += 2.0 * L.batch * L.h * L.w * (double)L.job[i].cout * L.job[i].cin * L.ksize * L.ksize;
    return f;
}
double tail_bytes(const TailConvLaunch& L) {
    double b = 0;
    for (int i = 0; i < L.njobs; ++i)
        b += 4.0 * ((double)L.batch * L.h * L.w * (L.job[i].cin + L.job[i].cout) +
                    (double)L.job[i].cout * L.job[i].cin * L.ksize * L.ksize + L.job[i].cout);
    return b;
}
int run_tail(Hook& hk, const char* name, const TailConvLaunch& L) {
    return hk.run(name, tail_flops(L), tail_bytes(L), [&](hipStream_t st) { return tail_conv_launch(L, st); });
}

// decoder convs 1..7 for decoders [k0,k1); input cat buffers `cat[k]`, final 2-ch flow of decoder k goes to
// channels [2*(k-kbase), +2) of `flow_dst` (which has flow_ctotal channels)
int run_decoders(eemflow_ctx* c, int k0, int k1, const float* const cat[3], int batch, int h, int w, float* flow_dst,
                 int flow_ctotal, int kbase, Hook& hk) {
    int rc;
    TailConvLaunch L;
    L.batch = batch; L.h = h; L.w = w; L.ksize = 3;
    // conv1 (69 -> 100) and conv5 (100 -> 64) on the Winograd kernel where the grid and the batch allow (dec_wnc_wanted): the decoders' 32-cout
    // slices as the jobs of one launch.  woff: the streams' offsets in dec_wnc, wper per decoder; *done says whether the launch was made
    auto wide = [&](const char* name, int cin, int cout, const float* const* in, float* const* outp, const size_t* woff, int wper,
                    const size_t* boff, bool* done) -> int {
        *done = false;
        if (!dec_wnc_wanted(c, w, batch)) return EEM_OK;
        int r2 = ensure_dec_wnc(c, hk.st);
        if (r2 != EEM_OK) return r2;
        WncArgs wa;
        memset(&wa, 0, sizeof(wa));
        wa.nchunks = wnc_chunks(cin, wa.chunk_off);
        wa.cin = cin; wa.n = batch; wa.h = h; wa.w = w; wa.act = 1; wa.m16 = 0;
        wa.zero_page = c->zero_page; wa.trash = c->zero_page + 256;
        const int ns = (cout + 31) / 32;
        for (int k = k0; k < k1; ++k)
            for (int s = 0; s < ns; ++s) {
                if (wa.njobs == WNC_MAX_JOBS) return EEM_OK;           // (more decoders than a launch has jobs: the small-grid kernel)
                WncJob& J = wa.job[wa.njobs++];
                J.in = in[k]; J.in_ctotal = cin; J.in_coff = 0;
                J.w = c->dec_wnc + woff[k * wper + s]; J.bias = c->dec_wnc + boff[k] + 32 * s;
                J.out = outp[k]; J.out_ctotal = cout; J.out_coff = 32 * s; J.out_cmul = 1; J.cout = cout - 32 * s < 32 ? cout - 32 * s : 32;
                J.res = nullptr;
            }
        if (!wnc_supported(wa)) return EEM_OK;
        const double px = (double)batch * h * w * (k1 - k0);
        r2 = hk.run(name, 2.0 * px * cin * cout * 9, 4.0 * px * (cin + cout), [&](hipStream_t st) {
            const int r3 = wnc_launch(wa, st);
            eem_last_pipe = 3;
            return r3;
        });
        *done = r2 == EEM_OK;
        return r2;
    };
    bool on_wnc = false;
    {
        float* outs[3] = {c->ta[0].p, c->ta[1].p, c->ta[2].p};
        if ((rc = wide("dec.conv1 69->100", kDecIn, kDecW, cat, outs, &c->dec_w1[0][0], 4, c->dec_b1, &on_wnc)) != EEM_OK) return rc;
    }
    if (!on_wnc) {
        L.njobs = 0;
        for (int k = k0; k < k1; ++k) L.job[L.njobs++] = make_job(c, c->dconv1[k], cat[k], kDecIn, 0, c->ta[k].p, kDecW, 0, 1, 1);
        if ((rc = run_tail(hk, "dec.conv1 69->100", L)) != EEM_OK) return rc;
    }
    // conv2..4: grouped 100 -> 100, each followed by channel_shuffle (EEMFlow.py:51-57):
    // group g, in-group channel j lands in channel j*groups + g
    const int G = c->groups, per = kDecW / G;
    const char* gname[3] = {"dec.conv2 grouped+shuffle", "dec.conv3 grouped+shuffle", "dec.conv4 grouped+shuffle"};
    for (int layer = 0; layer < 3; ++layer) {
        L.njobs = 0;
        for (int k = k0; k < k1; ++k) {
            // every activation keeps its own buffer (the training step reads them back): ta -> tb -> tc -> td
            float* chain[4] = {c->ta[k].p, c->tb[k].p, c->tc[k].p, c->td[k].p};
            float* src = chain[layer];
            float* dst = chain[layer + 1];
            for (int g = 0; g < G; ++g) {
                if (G == 1) L.job[L.njobs++] = make_job(c, c->dgroup[k][layer][g], src, kDecW, 0, dst, kDecW, 0, 1, 1);
                else L.job[L.njobs++] = make_job(c, c->dgroup[k][layer][g], src, kDecW, g * per, dst, kDecW, g, G, 1);
            }
        }
        if ((rc = run_tail(hk, gname[layer], L)) != EEM_OK) return rc;
    }
    {
        const float* ins[3] = {c->td[0].p, c->td[1].p, c->td[2].p};
        float* outs[3] = {c->t64[0].p, c->t64[1].p, c->t64[2].p};
        if ((rc = wide("dec.conv5 100->64", kDecW, 64, ins, outs, &c->dec_w5[0][0], 2, c->dec_b5, &on_wnc)) != EEM_OK) return rc;
    }
    if (!on_wnc) {
        L.njobs = 0;
        for (int k = k0; k < k1; ++k) L.job[L.njobs++] = make_job(c, c->dconv5[k], c->td[k].p, kDecW, 0, c->t64[k].p, 64, 0, 1, 1);
        if ((rc = run_tail(hk, "dec.conv5 100->64", L)) != EEM_OK) return rc;
    }
    L.njobs = 0;
    for (int k = k0; k < k1; ++k) L.job[L.njobs++] = make_job(c, c->dconv6[k], c->t64[k].p, 64, 0, c->t32[k].p, 32, 0, 1, 1);
    if ((rc = run_tail(hk, "dec.conv6 64->32", L)) != EEM_OK) return rc;
    L.njobs = 0;
    for (int k = k0; k < k1; ++k)
        L.job[L.njobs++] = make_job(c, c->dconv7[k], c->t32[k].p, 32, 0, flow_dst, flow_ctotal, 2 * (k - kbase), 1, 0);
    return run_tail(hk, "dec.conv7 32->2", L);
}

// io: nullptr (eager: the caller's pointers go into the launches) or the context's device table (graph capture)
// prepadded (optional): both event volumes already replicate-padded into one [2B][cin][hp][wp] batch (the training forward keeps that
// copy for the first layer's weight gradient anyway): the first layer then reads it with no padding of its own, which puts inputs whose
// rows are not 16-byte multiples or that pad on the left (MVSEC: 346-pixel rows, 19 columns) on the LDS-DMA kernel of conv_enc1.hip
static bool spans_on() {
#ifdef EEM_DIAG
    static const bool on = [] { const char* e = getenv("EEM_SPANS"); return e && e[0] == '1'; }();
    return on;
#else
    return false;                                        // EEM_SPANS is a diagnostic-build switch (-DEEM_DIAG)
#endif
}
static int span_mark(eemflow_ctx* c, int i, hipStream_t st) {
    if (!spans_on()) return EEM_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cs);
    if (cs != hipStreamCaptureStatusNone) return EEM_OK;
    if (i == 0 && c->span_pending) {
        EEM_HIP_CHECK(hipEventSynchronize(c->span_ev[2]));
        float a = 0.f, b = 0.f;
        EEM_HIP_CHECK(hipEventElapsedTime(&a, c->span_ev[0], c->span_ev[1]));
        EEM_HIP_CHECK(hipEventElapsedTime(&b, c->span_ev[1], c->span_ev[2]));
        c->span_sum[0] += a; c->span_sum[1] += b;
        if (++c->span_n == 64) {
            fprintf(stderr, "EEM_SPANS ctx %p: encoder chain %.1f us, tail chain %.1f us (64 frames)\n", (void*)c,
                    c->span_sum[0] / 64 * 1e3, c->span_sum[1] / 64 * 1e3);
            c->span_sum[0] = c->span_sum[1] = 0.0; c->span_n = 0;
        }
        c->span_pending = false;
    }
    if (!c->span_ev[i]) EEM_HIP_CHECK(hipEventCreate(&c->span_ev[i]));
    EEM_HIP_CHECK(hipEventRecord(c->span_ev[i], st));
    if (i == 2) c->span_pending = true;
    return EEM_OK;
}

// One encoder layer of the schedule (layer index = position in the chain).  may_skip_store: a layer whose output is read only
// through its fused pooling partial sums - pconv3_3 in inference - may leave the feature map unwritten (7.9 MB per frame at
// 1280x720); eemflow_get_stage("f13") then re-runs that one layer with stores.
int run_enc_layer(eemflow_ctx* c, const Shape& s, int li, const float* e1, const float* e2, Hook& hk, const void* const* io,
                  const float* prepadded, bool may_skip_store) {
    int rc;
    const int n2 = 2 * s.batch;
    struct Step { int layer; const char* name; const float* in; float* out; int hin, win, hout, wout; };
    const Step steps[ENC_NUM] = {
        {ENC_1_1, "enc.pconv1_1 5->16 s2 +pad", nullptr, c->a1.p, s.hp, s.wp, s.h1, s.w1},
        {ENC_1_2, "enc.pconv1_2 16->16", c->a1.p, c->f11.p, s.h1, s.w1, s.h1, s.w1},
        {ENC_2_1, "enc.pconv2_1 16->32 s2", c->f11.p, c->a2.p, s.h1, s.w1, s.h2, s.w2},
        {ENC_2_2, "enc.pconv2_2 32->32", c->a2.p, c->b2.p, s.h2, s.w2, s.h2, s.w2},
        {ENC_2_3, "enc.pconv2_3 32->32", c->b2.p, c->f12.p, s.h2, s.w2, s.h2, s.w2},
        {ENC_3_1, "enc.pconv3_1 32->64 s2", c->f12.p, c->a3.p, s.h2, s.w2, s.h3, s.w3},
        {ENC_3_2, "enc.pconv3_2 64->64", c->a3.p, c->b3.p, s.h3, s.w3, s.h3, s.w3},
        {ENC_3_3, "enc.pconv3_3 64->64", c->b3.p, c->f13.p, s.h3, s.w3, s.h3, s.w3}};
    if (li == ENC_1_1 && c->enc0_generic) {
        // n_first_channels != 5 (EEMFlow.py:72,75): replicate pad of both volumes into one batch (image_utils.py:129-140), then the generic
        // strided convolution + LeakyReLU
        const float* padded = prepadded;
        if (padded == nullptr) {
            EEM_REQUIRE(io == nullptr, "the generic first layer runs eagerly (no graph io table)");
            if ((rc = er_pad2_launch(e1, e2, c->padded.p, s.batch * c->cin0, s.in_h, s.in_w, c->pad[0], c->pad[1], c->pad[2], c->pad[3], hk.st)) != EEM_OK) return rc;
            padded = c->padded.p;
        }
        GConvArgs g;
        memset(&g, 0, sizeof(g));
        g.nseg = 1;
        g.seg[0].ptr = padded; g.seg[0].c = c->cin0; g.seg[0].ctotal = c->cin0; g.seg[0].coff = 0;
        g.wpk = c->arena + c->enc0_gw; g.shift = c->arena + c->enc_b[ENC_1_1];
        g.zero_page = c->zero_page;
        g.out = c->a1.p; g.out_ctotal = 16; g.out_coff = 0;
        g.n = n2; g.hin = s.hp; g.win = s.wp; g.hout = s.h1; g.wout = s.w1; g.cout = 16;
        g.kh = g.kw = 3; g.stride = 2; g.pad_h = g.pad_w = 1;
        g.act = GACT_LEAKY; g.epi = GEPI_PLAIN; g.out_scale = 1.f;
        const double opix = (double)n2 * s.h1 * s.w1;
        return hk.run("enc.pconv1_1 generic +pad", 2.0 * opix * 16 * c->cin0 * 9, 4.0 * ((double)n2 * s.in_h * s.in_w * c->cin0 + opix * 16),
                      [&](hipStream_t st) { return gconv_launch(g, st); });
    }
    {
        const Step& sp = steps[li];

        EncConvArgs a;
        const EncLayerDesc& d = kEncLayers[sp.layer];
        a.in0 = sp.layer == ENC_1_1 ? e1 : sp.in;
        a.in1 = sp.layer == ENC_1_1 ? e2 : nullptr;
        a.wpk = c->arena + c->enc_w[sp.layer];
        a.wpk2 = c->enc_has2[sp.layer] ? c->arena + c->enc_w2[sp.layer] : nullptr;
        a.wwino = nullptr;
        a.wino_f4 = 0;
        if (c->use_wino && c->enc_wino[sp.layer] && (rc = ensure_wino(c, sp.layer, 0, s.batch, hk.st, &a.wwino, &a.wino_f4)) != EEM_OK) return rc;
        a.ws2r = nullptr;
        if (c->enc_s2r[sp.layer] && s2r_wanted() && (rc = ensure_s2r(c, sp.layer, hk.st, &a.ws2r)) != EEM_OK) return rc;
        a.wbx3 = nullptr;
        if (c->enc_bx3[sp.layer] && bx3_wanted(sp.layer) && (rc = ensure_bx3(c, sp.layer, hk.st, &a.wbx3)) != EEM_OK) return rc;
        a.zero_page = c->zero_page;
        a.trash = c->zero_page + 256;
        a.bias = c->arena + c->enc_b[sp.layer];
        a.out = sp.out;
        a.nimg = n2; a.nimg0 = sp.layer == ENC_1_1 ? s.batch : n2;
        a.hin = sp.hin; a.win = sp.win; a.hout = sp.hout; a.wout = sp.wout;
        a.hraw = sp.layer == ENC_1_1 ? s.in_h : sp.hin;
        a.wraw = sp.layer == ENC_1_1 ? s.in_w : sp.win;
        a.pad_top = sp.layer == ENC_1_1 ? c->pad[2] : 0;
        a.pad_left = sp.layer == ENC_1_1 ? c->pad[0] : 0;
        if (sp.layer == ENC_1_1 && prepadded != nullptr) {
            a.in0 = prepadded;
            a.in1 = prepadded + (size_t)s.batch * c->cin0 * s.hp * s.wp;
            a.hraw = s.hp; a.wraw = s.wp; a.pad_top = 0; a.pad_left = 0;
        }
        a.act = 1;
        a.gate = nullptr;
        a.pool_partial = nullptr;
        a.pool_k = 0;
        a.io = sp.layer == ENC_1_1 ? io : nullptr;
        a.io_frames = (sp.layer == ENC_1_1 && io != nullptr) ? c->cur_io_frames : 0;
        a.in_norm = (sp.layer == ENC_1_1 && c->deferred_norm && prepadded == nullptr) ? 1 : 0;
        a.no_store = 0;
        {   // batched chains (EEM_ZIGZAG=<layer mask>, experiment): this layer walks the images back to front
            static const int zz = [] { const char* e = getenv("EEM_ZIGZAG"); return e ? atoi(e) : 0; }();
            a.reverse = (s.batch >= 2 && ((zz >> sp.layer) & 1)) ? 1 : 0;
            // ... or in COLUMNS (EEM_COLWALK=<layer mask>; default: the two 64-channel layers of a batched chain).  Measured at ten frames
            // per launch (rocprofv3 FETCH_SIZE): the 32-pixel-wide tiles of the 64-channel layers fetch 22.3 MB per frame in row order
            // and 10.8 / 10.3 in column order (a tile row touches three cache lines for one of payload, and in row order the neighbour
            // that shares two of them comes a whole tile later); 32 channels 29.0 -> 32.2 (worse), 16 channels unchanged; frame rate the
            // same within noise either way - those layers are bound by their transforms, not their bytes
            static const int cw = [] { const char* e = getenv("EEM_COLWALK"); return e ? atoi(e) : (1 << ENC_3_2) | (1 << ENC_3_3); }();
            if (s.batch >= 2 && ((cw >> sp.layer) & 1)) a.reverse = 2;
            // ... or INTERLEAVED (EEM_WALK3=<layer mask>, round 6): an XCD's blocks take every G-th tile of its range, so neighbouring
            // tiles are in flight together (conv_wino4.hip)
            // Measured at ten frames per launch (profiles/r06_walk3.txt): FETCH_SIZE per frame pconv1_2 49.5 -> 32.8 MB (31.5 of input),
            // pconv2_2 / 2_3 29.0 -> 17.0, pconv3_2 / 3_3 10.8 / 10.3 (column walk) -> 9.9 / 9.3; encoder 342 -> 299 MB per frame;
            // 10 290 -> 10 500 frames/s over 400 steps.  Default for every stride-1 layer of a batched chain (supersedes the column walk).
            static const int w3 = [] { const char* e = getenv("EEM_WALK3");
                                       return e ? atoi(e) : (1 << ENC_1_2) | (1 << ENC_2_2) | (1 << ENC_2_3) | (1 << ENC_3_2) | (1 << ENC_3_3); }();
            if (s.batch >= 2 && ((w3 >> sp.layer) & 1)) a.reverse = 3;
            static const int nts = [] { const char* e = getenv("EEM_NT_STORE"); return e ? atoi(e) : 0; }();
            a.nt_store = ((nts >> sp.layer) & 1) | ((nts >> 8) & 2);          // (bit 9, diagnostic builds: the weight-slice experiment of conv_wino4.hip)
        }
        // several frames in flight: kernels of different frames time-slice the CUs, so a block's prologue (DMA plan, first tile's
        // landing) is CU time another frame could use - fewer blocks with more tiles each (measured at 1280x720 with four in flight:
        // +3.5 % frames/s, +8 % single-frame latency; the 64-channel layers have one tile per CU and keep the full grid)
        // (pconv1_1, HBM-bound: 12 / 14 / 20 blocks per XCD give the same frame rate within 1 % - 8 610-8 660 / 8 580-8 620 / 8 520-8 580 -,
        // so it takes the fewest CUs: 32 us on 96 of them)
        static const int kInFlightBlocks[ENC_NUM] = {12, 24, 0, 29, 29, 0, 0, 0};
        a.blocks_per_xcd = c->frames_in_flight >= 3 ? kInFlightBlocks[sp.layer] : 0;
        for (int k = 0; k < 3; ++k)
            if (s.fuse[k] && sp.layer == (k == 0 ? ENC_1_2 : k == 1 ? ENC_2_3 : ENC_3_3)) {
                a.pool_partial = c->ppart[k].p;
                a.pool_k = k == 0 ? 32 : k == 1 ? 16 : 8;
                static const bool keep_f13 = [] { const char* e = getenv("EEM_KEEP_F13"); return e && e[0] == '1'; }();
                if (k == 2 && may_skip_store && !keep_f13) a.no_store = 1;
            }
        const double opix = (double)n2 * sp.hout * sp.wout;
        const double flops = 2.0 * opix * d.cout * d.cin * 9;
        const double ipix = sp.layer == ENC_1_1 ? (double)n2 * s.in_h * s.in_w : (double)n2 * sp.hin * sp.win;
        const double bytes = 4.0 * (ipix * d.cin + opix * d.cout + (double)d.cout * d.cin * 9 + d.cout);
        rc = hk.run(sp.name, flops, bytes,
                    [&](hipStream_t st) { return enc_conv_launch(d.cin, d.cout, d.stride, a, st); });
        if (rc != EEM_OK) return rc;
    }
    return EEM_OK;
}

// The first two encoder layers as ONE launch (conv_enc12.hip) when the schedule allows it: *done says whether it ran
int run_enc12(eemflow_ctx* c, const Shape& s, const float* e1, const float* e2, Hook& hk, const void* const* io, const float* prepadded,
              bool* done) {
    *done = false;
    // opt-in (EEM_FUSE12=1; read per schedule build - a cached graph keeps the form it was captured with): measured SLOWER than the two
    // launches it replaces, kept for the traffic it saves and as the record of that measurement
    const char* eon = getenv("EEM_FUSE12");
    const bool off = !(eon && eon[0] == '1');
    if (off || c->deferred_norm || c->keep_stage_stores || c->enc0_generic || prepadded != nullptr || !c->use_wino || !c->enc_wino[ENC_1_2] ||
        !c->layer_f4(16, s.batch) || !s.fuse[0] || c->fuse_scratch.p == nullptr)
        return EEM_OK;
    int rc;
    Enc12Args a;
    memset(&a, 0, sizeof(a));
    a.in0 = e1; a.in1 = e2; a.io = io; a.io_frames = io != nullptr ? c->cur_io_frames : 0;
    a.wpk1 = c->arena + c->enc_w[ENC_1_1]; a.bias1 = c->arena + c->enc_b[ENC_1_1];
    int f4 = 0;
    if ((rc = ensure_wino(c, ENC_1_2, 0, s.batch, hk.st, &a.u2, &f4)) != EEM_OK) return rc;
    if (!f4) return EEM_OK;
    a.bias2 = c->arena + c->enc_b[ENC_1_2];
    a.zero_page = c->zero_page; a.trash = c->zero_page + 256;
    a.out = c->f11.p; a.pool_partial = c->ppart[0].p; a.scratch = c->fuse_scratch.p;
    a.nimg = 2 * s.batch; a.nimg0 = s.batch;
    a.hraw = s.in_h; a.wraw = s.in_w; a.pad_top = c->pad[2];
    a.hin = s.hp; a.win = s.wp; a.h1 = s.h1; a.w1 = s.w1;
    if (c->pad[0] != 0 || !enc12_supported(a)) return EEM_OK;
    const int blocks = enc12_blocks(a.nimg, a.h1, a.w1, 0);
    if (enc12_scratch_floats(blocks) > c->fuse_scratch.cap) return EEM_OK;
    const double n2 = 2.0 * s.batch, opix = n2 * s.h1 * s.w1;
    const double flops = 2.0 * opix * 16 * (5 + 16) * 9;
    const double bytes = 4.0 * (n2 * s.in_h * s.in_w * 5 + opix * 16 + 16.0 * (5 + 16) * 9 + 32);
    rc = hk.run("enc.pconv1_1+1_2 fused 5->16 s2 +pad, 16->16", flops, bytes, [&](hipStream_t st) { return enc12_launch(a, blocks, st); });
    if (rc == EEM_OK) *done = true;
    return rc;
}

int run_forward_impl(eemflow_ctx* c, const Shape& s, const float* e1, const float* e2, float* out, Hook& hk,
                     const void* const* io, const float* prepadded);
int run_forward(eemflow_ctx* c, const Shape& s, const float* e1, const float* e2, float* out, Hook& hk,
                const void* const* io = nullptr, const float* prepadded = nullptr) {
    int rc = span_mark(c, 0, hk.st);
    if (rc == EEM_OK) rc = run_forward_impl(c, s, e1, e2, out, hk, io, prepadded);
    if (rc == EEM_OK) rc = span_mark(c, 2, hk.st);
    return rc;
}

int run_forward_impl(eemflow_ctx* c, const Shape& s, const float* e1, const float* e2, float* out, Hook& hk,
                     const void* const* io, const float* prepadded) {
    int rc;
    const int n2 = 2 * s.batch;
    // ---- encoder (both event volumes as one batch; shared weights, EEMFlow.py:135-140)
    // pconv1_1 + pconv1_2 as one launch when nothing needs a1 itself (inference; the training forward keeps every activation)
    bool fused12 = false;
    if ((rc = run_enc12(c, s, e1, e2, hk, io, prepadded, &fused12)) != EEM_OK) return rc;
    c->a1_skipped = fused12;
    for (int li = fused12 ? ENC_2_1 : 0; li < ENC_NUM; ++li)
        if ((rc = run_enc_layer(c, s, li, e1, e2, hk, io, prepadded, !c->keep_stage_stores)) != EEM_OK) return rc;
    c->f13_skipped = !c->keep_stage_stores;
    // ---- stage pooling to the common 1/64 grid (EEMFlow.py:144-154), 53-tap correlation and rconv into the decoders' input
    // [cv | r] (EEMFlow.py:160-163).  Fused form (default): ONE launch whose correlation / rconv blocks read the conv epilogues'
    // pooling partial sums directly and whose extra blocks write the finished pooled maps (tail_fused.hip).  Stages whose conv
    // ran the generic kernel are pooled from the stored feature map first.  EEM_NO_TAIL_FUSE=1: the three separate launches.
    const size_t g = (size_t)s.gh * s.gw;
    const int pc[3] = {16, 32, 64};
    static const bool no_fuse = [] { const char* e = getenv("EEM_NO_TAIL_FUSE"); return e && e[0] == '1'; }();
    {
        const float* feat[3] = {c->f11.p, c->f12.p, c->f13.p};
        const int hs[3] = {s.h1, s.h2, s.h3}, ws[3] = {s.w1, s.w2, s.w3}, ks[3] = {32, 16, 8};
        PoolFinJob fj[3];
        PoolJob pj[3];
        int nf = 0, np = 0;
        double fin_elems = 0, pool_elems = 0;
        for (int k = 0; k < 3; ++k) {
            if (s.fuse[k]) {
                fj[nf++] = {c->ppart[k].p, c->pool[k].p, pc[k], s.prow[k], s.pcol[k], ks[k] / s.th[k], ks[k]};
                fin_elems += (double)n2 * pc[k] * s.gh * s.gw * (ks[k] / s.th[k] + 1);
            } else {
                pj[np++] = {feat[k], c->pool[k].p, pc[k], hs[k], ws[k], ks[k]};
                pool_elems += (double)n2 * pc[k] * hs[k] * ws[k];
            }
        }
        if (np) {
            rc = hk.run("pool 32/16/8", pool_elems, 4.0 * pool_elems,
                        [&](hipStream_t st) { return pool_launch(pj, np, n2, st); });
            if (rc != EEM_OK) return rc;
        }
        if ((rc = span_mark(c, 1, hk.st)) != EEM_OK) return rc;
        if (!no_fuse) {
            TailHeadArgs ha;
            memset(&ha, 0, sizeof(ha));
            for (int k = 0; k < 3; ++k) {
                PooledSrc& ps = ha.src[k];
                if (s.fuse[k]) {
                    const int rows = ks[k] / s.th[k];
                    ps.base = c->ppart[k].p; ps.rows = rows; ps.rstride = s.pcol[k]; ps.ystride = rows * s.pcol[k];
                    ps.cstride = s.prow[k] * s.pcol[k]; ps.nstride = pc[k] * ps.cstride; ps.scale = 1.f / (float)(ks[k] * ks[k]);
                } else {
                    ps.base = c->pool[k].p; ps.rows = 1; ps.rstride = 0; ps.ystride = s.gw; ps.cstride = (int)g;
                    ps.nstride = pc[k] * (int)g; ps.scale = 1.f;
                }
                ha.c[k] = pc[k];
                ha.cat[k] = c->cat[k].p;
                ha.pool_out[k] = s.fuse[k] ? c->pool[k].p : nullptr;
                ha.rw[k] = c->arena + c->rconv[k].wpk;
                ha.rb[k] = c->arena + c->rconv[k].bias;
            }
            ha.batch = s.batch; ha.gh = s.gh; ha.gw = s.gw; ha.ntaps = kNTaps; ha.cat_ctotal = kDecIn;
            const double fl = 2.0 * s.batch * g * (kNTaps * (16 + 32 + 64) + 16.0 * 9 * (16 + 32 + 64));
            rc = hk.run("tail head: pool+corr53+rconv", fl, 4.0 * (fin_elems + 3.0 * s.batch * g * kDecIn),
                        [&](hipStream_t st) { return tail_head_launch(ha, kTaps53, st); });
            if (rc != EEM_OK) return rc;
        } else {
            if (nf) {
                rc = hk.run("pool finalize (fused partials)", fin_elems, 4.0 * fin_elems, [&](hipStream_t st) {
                    return pool_finalize_launch(fj, nf, n2, s.gh, s.gw, st);
                });
                if (rc != EEM_OK) return rc;
            }
            CorrJob cj[3];
            for (int k = 0; k < 3; ++k)
                cj[k] = {c->pool[k].p, c->pool[k].p + (size_t)s.batch * pc[k] * g, c->cat[k].p, pc[k], kDecIn};
            rc = hk.run("local_corr 9x9 (53 taps)", 2.0 * s.batch * g * kNTaps * (16 + 32 + 64),
                        4.0 * s.batch * g * (2.0 * (16 + 32 + 64) + 3.0 * kNTaps),
                        [&](hipStream_t st) { return corr_launch(cj, 3, s.batch, s.gh, s.gw, c->taps, kNTaps, st); });
            if (rc != EEM_OK) return rc;
            TailConvLaunch L0;
            L0.batch = s.batch; L0.h = s.gh; L0.w = s.gw; L0.ksize = 3; L0.njobs = 0;
            for (int k = 0; k < 3; ++k)
                L0.job[L0.njobs++] = make_job(c, c->rconv[k], c->pool[k].p, pc[k], 0, c->cat[k].p, kDecIn, kNTaps, 1, 1);
            if ((rc = run_tail(hk, "rconv {16,32,64}->16", L0)) != EEM_OK) return rc;
        }
    }
    TailConvLaunch L;
    L.batch = s.batch; L.h = s.gh; L.w = s.gw; L.ksize = 3; L.njobs = 0;
    // ---- decoders, out_conv, upsample (EEMFlow.py:164-181)
    const float* cats[3] = {c->cat[0].p, c->cat[1].p, c->cat[2].p};
    const bool fuse_up = !no_fuse && tail_up_supported(s.gh, s.gw, s.out_h, s.out_w);
    if ((rc = run_decoders(c, 0, 3, cats, s.batch, s.gh, s.gw, c->flowcat.p, 6, 0, hk)) != EEM_OK) return rc;
    if (fuse_up) {
        // out_conv + bilinear upsample in one launch; `coarse` is its side output
        TailUpArgs ua;
        memset(&ua, 0, sizeof(ua));
        ua.wo = c->flat + c->t_outc.w; ua.bo = c->flat + c->t_outc.b;
        ua.flowcat = c->flowcat.p; ua.coarse = c->coarse.p; ua.out = out; ua.io = io;
        ua.io_frames = io != nullptr ? c->cur_io_frames : 0;
        ua.batch = s.batch; ua.gh = s.gh; ua.gw = s.gw; ua.oh = s.out_h; ua.ow = s.out_w;
        ua.out_aligned16 = ((uintptr_t)out & 15) == 0;
        const double opix = (double)s.batch * 2 * s.out_h * s.out_w;
        return hk.run("tail up: out_conv+upsample", 8.0 * opix + 2.0 * s.batch * g * 12,
                      4.0 * (opix + (double)s.batch * 8 * g), [&](hipStream_t st) { return tail_up_launch(ua, st); });
    }
    L.ksize = 1; L.njobs = 1;
    L.job[0] = make_job(c, c->outc, c->flowcat.p, 6, 0, c->coarse.p, 2, 0, 1, 0);
    if ((rc = run_tail(hk, "out_conv 1x1 6->2", L)) != EEM_OK) return rc;
    const double opix = (double)s.batch * 2 * s.out_h * s.out_w;
    return hk.run("upsample bilinear", 8.0 * opix, 4.0 * (opix + (double)s.batch * 2 * g), [&](hipStream_t st) {
        return upsample_launch(c->coarse.p, out, s.batch * 2, s.gh, s.gw, s.out_h, s.out_w, st, io, io != nullptr ? c->cur_io_frames : 0);
    });
}

}  // namespace

