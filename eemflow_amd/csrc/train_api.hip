// Training step of EEMFlow on the C ABI (train_mvsec.py:229-258 for one iteration): forward with every
// activation kept, sequence loss, the full backward pass into one flat gradient buffer (state_dict order - the
// buffer a data-parallel job all-reduces over RCCL), then gradient clipping + AdamW on the device-resident
// weights and the re-pack of every MFMA weight layout.
#include "api_internal.h"
#include "eraft_kernels.h"
#include "train.h"

#include <vector>

namespace {

typedef eemflow_ctx::ConvRef ConvRef;

// The context's side stream (weight gradients; the prologue of a training forward), or `st` itself under EEM_NO_WGRAD_STREAM=1 (read per
// call: the tests run both forms)
int side_stream(eemflow_ctx* c, hipStream_t st, hipStream_t* out) {
    const char* off = getenv("EEM_NO_WGRAD_STREAM");
    *out = st;
    if (off && off[0] == '1') return EEM_OK;
    if (!c->wstream) {
        EEM_HIP_CHECK(hipStreamCreateWithFlags(&c->wstream, hipStreamNonBlocking));
        for (hipEvent_t& e : c->wev) EEM_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        EEM_HIP_CHECK(hipEventCreateWithFlags(&c->wjoin, hipEventDisableTiming));
        EEM_HIP_CHECK(hipEventCreateWithFlags(&c->prep_ev, hipEventDisableTiming));
    }
    *out = c->wstream;
    return EEM_OK;
}

struct Bwd {
    eemflow_ctx* c;
    hipStream_t st;
    float* grad;           // flat gradient buffer
    hipStream_t wst;       // stream of the weight / bias gradient launches (the context's side stream, or st)
    bool grad_zeroed = false;   // the forward's prologue cleared `grad` on the side stream (forward_train_impl)

    // the side stream picks up after everything queued on st so far (the gradient a weight-gradient launch reads is complete)
    int fork() {
        if (wst == st) return EEM_OK;
        hipEvent_t e = c->wev[c->wev_next++ % eemflow_ctx::kWEvents];
        EEM_HIP_CHECK(hipEventRecord(e, st));
        EEM_HIP_CHECK(hipStreamWaitEvent(wst, e, 0));
        return EEM_OK;
    }
    int join() {
        if (wst == st) return EEM_OK;
        EEM_HIP_CHECK(hipEventRecord(c->wjoin, wst));
        EEM_HIP_CHECK(hipStreamWaitEvent(st, c->wjoin, 0));
        return EEM_OK;
    }

    // data gradient of a conv layer through gconv: dX = convT(dY * act'(Y), W)
    int dgrad(const ConvRef& r, const float* dy, const float* y_gate, int g_ctotal, int g_coff, int g_cmul, int n, int hout,
              int wout, int hin, int win, float* dx, int dx_ctotal, int dx_coff) {
        GConvArgs a;
        memset(&a, 0, sizeof(a));
        a.nseg = 1;
        a.seg[0].ptr = dy; a.seg[0].c = r.cout; a.seg[0].ctotal = g_ctotal; a.seg[0].coff = g_coff; a.seg[0].cmul = g_cmul;
        a.seg[0].gate = y_gate;
        a.wpk = c->arena + r.wT;
        a.out = dx; a.out_ctotal = dx_ctotal; a.out_coff = dx_coff;
        a.n = n; a.hin = hout; a.win = wout; a.hout = hin; a.wout = win; a.cout = r.cin;
        a.kh = r.k; a.kw = r.k; a.stride = 1;
        a.pad_h = a.pad_w = r.k - 1 - (r.k == 3 ? 1 : 0);      // k - 1 - pad
        a.tstride = r.stride;
        a.act = GACT_NONE; a.epi = GEPI_PLAIN; a.out_scale = 1.f;
        return gconv_launch(a, st);
    }
    // data-gradient job of a tail conv for tail_conv_kernel (W^T packed at load time): reads `cin_d` = r.cout channels
    // of dy at g_coff + i * g_cmul (gated by y_gate), writes r.cin channels of dx at dx_coff..
    TailConvJob djob(const ConvRef& r, const float* dy, const float* y_gate, int g_ctotal, int g_coff, int g_cmul, float* dx,
                     int dx_ctotal, int dx_coff) const {
        TailConvJob j;
        j.in = dy; j.gate = y_gate; j.in_ctotal = g_ctotal; j.in_coff = g_coff; j.in_cmul = g_cmul; j.add = nullptr;
        j.wpk = c->arena + r.wT_tail; j.bias = nullptr;
        j.cin = r.cout; j.cout = r.cin;
        j.out = dx; j.out_ctotal = dx_ctotal; j.out_coff = dx_coff; j.out_cmul = 1; j.act = 0;
        return j;
    }
    // queued variant: the weight gradients of one tail layer (all decoders / groups) go out as ONE launch (flush_wgrads)
    WgradArgs wq[WGRAD_MAX_JOBS];
    BiasJob bq[WGRAD_MAX_JOBS];
    int nwq = 0;
    // round 6: the 3x3 convs of the tail are collected over ALL its layers and leave as ONE launch when the tail's chain is through
    // (flush_tail; wgrad_tail.hip) - their operands are the tail's activations and gradients, each in a buffer of its own
    std::vector<WgradArgs> tq;
    bool tail_one_launch = false;
    int wgrad_q(const ConvRef& r, const float* x, int x_ctotal, int x_coff, const float* dy, const float* y_gate, int g_ctotal,
                int g_coff, int g_cmul, int n, int hin, int win, int hout, int wout) {
        if (tail_one_launch && r.k == 3) {
            WgradArgs w;
            w.x = x; w.x_ctotal = x_ctotal; w.x_coff = x_coff; w.cin = r.cin;
            w.g = dy; w.gate = y_gate; w.g_ctotal = g_ctotal; w.g_coff = g_coff; w.g_cmul = g_cmul; w.cout = r.cout;
            w.dw = grad + r.w; w.db = grad + r.b;
            w.n = n; w.hin = hin; w.win = win; w.hout = hout; w.wout = wout; w.k = 3; w.stride = r.stride; w.pad = 1;
            w.zero_page = nullptr;
            tq.push_back(w);
            return EEM_OK;
        }
        WgradArgs& w = wq[nwq++];
        w.x = x; w.x_ctotal = x_ctotal; w.x_coff = x_coff; w.cin = r.cin;
        w.g = dy; w.gate = y_gate; w.g_ctotal = g_ctotal; w.g_coff = g_coff; w.g_cmul = g_cmul; w.cout = r.cout;
        w.dw = grad + r.w;
        w.n = n; w.hin = hin; w.win = win; w.hout = hout; w.wout = wout; w.k = r.k; w.stride = r.stride; w.pad = r.k == 3 ? 1 : 0;
        w.zero_page = nullptr; w.db = nullptr;
        BiasJob& b = bq[nwq - 1];
        b.g = dy; b.gate = y_gate; b.db = grad + r.b; b.g_ctotal = g_ctotal; b.g_coff = g_coff; b.g_cmul = g_cmul;
        b.cout = r.cout; b.n = n; b.hw = hout * wout;
        return EEM_OK;
    }
    int flush_wgrads() {
        if (nwq == 0) return EEM_OK;
        int rc = fork();
        if (rc == EEM_OK) rc = tr_wgrad_launch_batch(wq, nwq, wst);
        if (rc == EEM_OK) rc = tr_bias_grad_launch_batch(bq, nwq, wst);
        nwq = 0;
        return rc;
    }
    int flush_tail(int n, int h, int w) {
        if (tq.empty()) return flush_wgrads();
        int rc = fork();
        if (rc == EEM_OK) rc = wgrad_tail_launch(tq.data(), (int)tq.size(), n, h, w, wst);
        tq.clear();
        if (rc == EEM_OK && nwq > 0) {                   // (the deferred ones, behind the same fork)
            rc = tr_wgrad_launch_batch(wq, nwq, wst);
            if (rc == EEM_OK) rc = tr_bias_grad_launch_batch(bq, nwq, wst);
            nwq = 0;
        }
        return rc;
    }
    // weight + bias gradient of a conv layer into the flat buffer
    int wgrad(const ConvRef& r, const float* x, int x_ctotal, int x_coff, const float* dy, const float* y_gate, int g_ctotal,
              int g_coff, int g_cmul, int n, int hin, int win, int hout, int wout) {
        WgradArgs w;
        w.x = x; w.x_ctotal = x_ctotal; w.x_coff = x_coff; w.cin = r.cin;
        w.g = dy; w.gate = y_gate; w.g_ctotal = g_ctotal; w.g_coff = g_coff; w.g_cmul = g_cmul; w.cout = r.cout;
        w.dw = grad + r.w;
        w.n = n; w.hin = hin; w.win = win; w.hout = hout; w.wout = wout; w.k = r.k; w.stride = r.stride; w.pad = r.k == 3 ? 1 : 0;
        w.zero_page = c->zero_page; w.db = grad + r.b;
        int rc = fork();
        if (rc != EEM_OK) return rc;
        if (wgrad_ring_supported(w) && wgrad_ring_preferred(w)) return wgrad_ring_launch(w, wst);   // weight and bias gradient in one kernel
        if (wgrad_enc_supported(w)) return wgrad_enc_launch(w, wst);
        rc = tr_wgrad_launch(w, wst);
        if (rc != EEM_OK) return rc;
        return tr_bias_grad_launch(dy, y_gate, g_ctotal, g_coff, g_cmul, r.cout, n, hout * wout, grad + r.b, wst);
    }
};

int alloc_train(eemflow_ctx* c, const Shape& s) {
    const size_t n2 = 2 * (size_t)s.batch, B = s.batch, g = (size_t)s.gh * s.gw;
    int rc;
#define ENS(buf, n) if ((rc = ensure(buf, n)) != EEM_OK) return rc
    ENS(c->padded, n2 * c->cin0 * s.hp * s.wp);
    ENS(c->g_a1, n2 * 16 * s.h1 * s.w1);  ENS(c->g_f11, n2 * 16 * s.h1 * s.w1);
    ENS(c->g_a2, n2 * 32 * s.h2 * s.w2);  ENS(c->g_b2, n2 * 32 * s.h2 * s.w2);  ENS(c->g_f12, n2 * 32 * s.h2 * s.w2);
    ENS(c->g_a3, n2 * 64 * s.h3 * s.w3);  ENS(c->g_b3, n2 * 64 * s.h3 * s.w3);  ENS(c->g_f13, n2 * 64 * s.h3 * s.w3);
    const int pc[3] = {16, 32, 64};
    for (int k = 0; k < 3; ++k) {
        ENS(c->g_pool[k], n2 * pc[k] * g);  ENS(c->g_cat[k], B * kDecIn * g);
        ENS(c->g_ta[k], B * kDecW * g);  ENS(c->g_tb[k], B * kDecW * g);  ENS(c->g_tc[k], B * kDecW * g);  ENS(c->g_td[k], B * kDecW * g);
        ENS(c->g_t64[k], B * 64 * g);  ENS(c->g_t32[k], B * 32 * g);
    }
    ENS(c->g_flowcat, B * 6 * g);  ENS(c->g_coarse, B * 2 * g);
    ENS(c->g_flow, B * 2 * (size_t)s.out_h * s.out_w);
    ENS(c->ups_tmp, B * 2 * (size_t)s.out_h * s.gw);
    ENS(c->grad_flat, c->nflat);
    ENS(c->scalars, 24);
#undef ENS
    return EEM_OK;
}

}  // namespace

// Backward pass of the LAST eager forward of this context (its activations are still in the workspace): dflow
// [B][2][out_h][out_w] -> flat gradient (state_dict order).  e1 / e2 are that forward's inputs (pconv1_1's weight
// gradient reads them through the replicate pad).
static int backward_chain(eemflow_ctx* c, const Shape& s, const float* e1, const float* e2, const float* dflow, float* grad_out,
                          hipStream_t st, Bwd& bw);

static int backward_impl(eemflow_ctx* c, const Shape& s, const float* e1, const float* e2, const float* dflow, float* grad_out,
                         hipStream_t st, bool grad_zeroed = false) {
    hipStream_t wst;
    const int rs = side_stream(c, st, &wst);
    if (rs != EEM_OK) return rs;
    Bwd bw{c, st, grad_out, wst};
    bw.grad_zeroed = grad_zeroed;
    const int rc = backward_chain(c, s, e1, e2, dflow, grad_out, st, bw);
    const int rj = bw.join();                        // also after an error: nothing of this pass stays behind on the side stream
    return rc != EEM_OK ? rc : rj;
}

static int backward_chain(eemflow_ctx* c, const Shape& s, const float* e1, const float* e2, const float* dflow, float* grad_out,
                          hipStream_t st, Bwd& bw) {
    int rc;
    const int B = s.batch, n2 = 2 * s.batch, in_h = s.in_h, in_w = s.in_w;
    const size_t g = (size_t)s.gh * s.gw;
    if (!bw.grad_zeroed) EEM_HIP_CHECK(hipMemsetAsync(grad_out, 0, c->nflat * sizeof(float), st));
    // ---- upsample backward (EEMFlow.py:118-120)
    if ((rc = tr_upsample_bwd_launch(dflow, c->ups_tmp.p, c->g_coarse.p, B * 2, s.out_h, s.out_w, s.gh, s.gw, st)) != EEM_OK) return rc;
    // ---- the 1/64-grid tail, last layer first.  Weight / bias gradients: one launch per conv.  Data gradients: the
    // same conv layer of all three decoders (and all five groups) as the jobs of ONE tail_conv_kernel launch - a 9-way
    // K split per block like the forward, instead of 60 register-gather launches of 32 blocks each.
    const int G = c->groups, per = kDecW / G;
    const int pc[3] = {16, 32, 64};
    const int gh = s.gh, gw = s.gw;
    TailConvLaunch TL;
    TL.batch = B; TL.h = gh; TL.w = gw;
    auto run_jobs = [&](int ksize) {
        TL.ksize = ksize;
        // this layer's weight gradients: one launch - or, with the tail's 3x3 layers leaving as one launch at the end (flush_tail), the
        // few others (the 1x1 out_conv) wait for that launch's fork: one event record less in the chain (~7 us each)
        int r = bw.tail_one_launch ? EEM_OK : bw.flush_wgrads();
        if (r == EEM_OK) r = tail_conv_launch(TL, st);
        TL.njobs = 0;
        return r;
    };
    TL.njobs = 0;
    {
        // one launch for the tail's 3x3 weight gradients when the map fits a K part (wgrad_tail.hip; EEM_NO_WGRAD_TAIL=1, read per call)
        WgradArgs probe;
        probe.k = 3; probe.kh = 0; probe.stride = 1; probe.pad = 1; probe.hin = probe.hout = gh; probe.win = probe.wout = gw; probe.n = B;
        probe.cin = probe.cout = 16;
        bw.tail_one_launch = wgrad_tail_supported(&probe, 1, B, gh, gw);
    }
    // out_conv (1x1, no activation)
    if ((rc = bw.wgrad_q(c->t_outc, c->flowcat.p, 6, 0, c->g_coarse.p, nullptr, 2, 0, 1, B, gh, gw, gh, gw)) != EEM_OK) return rc;
    TL.job[TL.njobs++] = bw.djob(c->t_outc, c->g_coarse.p, nullptr, 2, 0, 1, c->g_flowcat.p, 6, 0);
    if ((rc = run_jobs(1)) != EEM_OK) return rc;
    // conv7: 32 -> 2, no activation; its output gradient is channels [2k, 2k+2) of g_flowcat
    for (int k = 0; k < 3; ++k) {
        if ((rc = bw.wgrad_q(c->t_dconv7[k], c->t32[k].p, 32, 0, c->g_flowcat.p, nullptr, 6, 2 * k, 1, B, gh, gw, gh, gw)) != EEM_OK) return rc;
        TL.job[TL.njobs++] = bw.djob(c->t_dconv7[k], c->g_flowcat.p, nullptr, 6, 2 * k, 1, c->g_t32[k].p, 32, 0);
    }
    if ((rc = run_jobs(3)) != EEM_OK) return rc;
    // conv6: 64 -> 32 (gate = its output t32)
    for (int k = 0; k < 3; ++k) {
        if ((rc = bw.wgrad_q(c->t_dconv6[k], c->t64[k].p, 64, 0, c->g_t32[k].p, c->t32[k].p, 32, 0, 1, B, gh, gw, gh, gw)) != EEM_OK) return rc;
        TL.job[TL.njobs++] = bw.djob(c->t_dconv6[k], c->g_t32[k].p, c->t32[k].p, 32, 0, 1, c->g_t64[k].p, 64, 0);
    }
    if ((rc = run_jobs(3)) != EEM_OK) return rc;
    // conv5: 100 -> 64, input = td (shuffled output of conv4)
    for (int k = 0; k < 3; ++k) {
        if ((rc = bw.wgrad_q(c->t_dconv5[k], c->td[k].p, kDecW, 0, c->g_t64[k].p, c->t64[k].p, 64, 0, 1, B, gh, gw, gh, gw)) != EEM_OK) return rc;
        TL.job[TL.njobs++] = bw.djob(c->t_dconv5[k], c->g_t64[k].p, c->t64[k].p, 64, 0, 1, c->g_td[k].p, kDecW, 0);
    }
    if ((rc = run_jobs(3)) != EEM_OK) return rc;

    // conv4, conv3, conv2: grouped + channel shuffle; group g's output channel j lives at j*G + g of the shuffled
    // tensor, its inputs are channels [g*per, (g+1)*per) of the previous activation
    for (int layer = 2; layer >= 0; --layer) {
        for (int k = 0; k < 3; ++k) {
            float* act[4] = {c->ta[k].p, c->tb[k].p, c->tc[k].p, c->td[k].p};
            float* gact[4] = {c->g_ta[k].p, c->g_tb[k].p, c->g_tc[k].p, c->g_td[k].p};
            for (int gi = 0; gi < G; ++gi) {
                const ConvRef& r = c->t_dgroup[k][layer][gi];
                if ((rc = bw.wgrad_q(r, act[layer], kDecW, gi * per, gact[layer + 1], act[layer + 1], kDecW, gi, G, B, gh, gw, gh, gw)) != EEM_OK) return rc;
                TL.job[TL.njobs++] = bw.djob(r, gact[layer + 1], act[layer + 1], kDecW, gi, G, gact[layer], kDecW, gi * per);
            }
        }
        if ((rc = run_jobs(3)) != EEM_OK) return rc;
    }
    // conv1: 69 -> 100, input = cat_k
    for (int k = 0; k < 3; ++k) {
        if ((rc = bw.wgrad_q(c->t_dconv1[k], c->cat[k].p, kDecIn, 0, c->g_ta[k].p, c->ta[k].p, kDecW, 0, 1, B, gh, gw, gh, gw)) != EEM_OK) return rc;
        TL.job[TL.njobs++] = bw.djob(c->t_dconv1[k], c->g_ta[k].p, c->ta[k].p, kDecW, 0, 1, c->g_cat[k].p, kDecIn, 0);
    }
    if ((rc = run_jobs(3)) != EEM_OK) return rc;
    // rconv_k: pooled features of events1 -> channels [53, 69) of cat_k (gate = those channels)
    for (int k = 0; k < 3; ++k) {
        if ((rc = bw.wgrad_q(c->t_rconv[k], c->pool[k].p, pc[k], 0, c->g_cat[k].p, c->cat[k].p, kDecIn, kNTaps, 1, B, gh, gw, gh, gw)) != EEM_OK) return rc;
        TL.job[TL.njobs++] = bw.djob(c->t_rconv[k], c->g_cat[k].p, c->cat[k].p, kDecIn, kNTaps, 1, c->g_pool[k].p, pc[k], 0);
    }
    if ((rc = run_jobs(3)) != EEM_OK) return rc;
    // every 3x3 weight gradient of the tail: one launch on the side stream (three launches beside the chain - after conv5, after the
    // grouped layers, here - measured 178 us of kernels for 99 and a slower step: the chain's own launches wait behind their blocks)
    if ((rc = bw.flush_tail(B, gh, gw)) != EEM_OK) return rc;
    // correlation: adds to d pool1, writes d pool2
    {
        CorrBwdJob cj[3];
        for (int k = 0; k < 3; ++k)
            cj[k] = CorrBwdJob{c->g_cat[k].p, c->pool[k].p, c->pool[k].p + (size_t)B * pc[k] * g, c->g_pool[k].p,
                               c->g_pool[k].p + (size_t)B * pc[k] * g, kDecIn, pc[k]};
        if ((rc = tr_corr_bwd_launch_jobs(cj, 3, B, gh, gw, c->taps, kNTaps, st)) != EEM_OK) return rc;      // one launch for the three stages
    }
    // ---- encoder (EEMFlow.py:135-154): both event volumes as one batch of 2B images; c->padded holds them replicate-padded since the
    // forward (forward_train_impl), which read its first layer from there
    struct L { int layer; const float* x; int xc, hin, win; const float* y; float* gy; int hout, wout; float* gx; };
    const L ls[ENC_NUM] = {
        {ENC_3_3, c->b3.p, 64, s.h3, s.w3, c->f13.p, c->g_f13.p, s.h3, s.w3, c->g_b3.p},
        {ENC_3_2, c->a3.p, 64, s.h3, s.w3, c->b3.p, c->g_b3.p, s.h3, s.w3, c->g_a3.p},
        {ENC_3_1, c->f12.p, 32, s.h2, s.w2, c->a3.p, c->g_a3.p, s.h3, s.w3, c->g_f12.p},
        {ENC_2_3, c->b2.p, 32, s.h2, s.w2, c->f12.p, c->g_f12.p, s.h2, s.w2, c->g_b2.p},
        {ENC_2_2, c->a2.p, 32, s.h2, s.w2, c->b2.p, c->g_b2.p, s.h2, s.w2, c->g_a2.p},
        {ENC_2_1, c->f11.p, 16, s.h1, s.w1, c->a2.p, c->g_a2.p, s.h2, s.w2, c->g_f11.p},
        {ENC_1_2, c->a1.p, 16, s.h1, s.w1, c->f11.p, c->g_f11.p, s.h1, s.w1, c->g_a1.p},
        {ENC_1_1, c->padded.p, c->cin0, s.hp, s.wp, c->a1.p, c->g_a1.p, s.h1, s.w1, nullptr}};
    // Encoder gradients are stored "pre-gated": g_* = dL/dY (.) LeakyReLU'(Y), the gradient w.r.t. the conv's
    // pre-activation, so weight/bias gradients read them as they are and the stride-1 data gradients run on the
    // encoder's own fast conv kernels (W^T, zero bias, no activation, epilogue gate = the next layer's output).
    if ((rc = tr_pool_bwd_launch(c->g_pool[2].p, c->g_f13.p, (long)n2 * 64, s.h3, s.w3, 8, s.gh, s.gw, 0, c->f13.p, st)) != EEM_OK) return rc;
    for (const L& l : ls) {
        const ConvRef& r = c->t_enc[l.layer];
        if (!l.gx) {
            // the first layer has no data gradient: its weight gradient follows the chain on st, where its operand was produced, and runs
            // beside pconv1_2's on the side stream instead of behind it (EEM_WGRAD_LAST_SIDE=1: the round-5 order, for measurements)
            static const bool side = [] { const char* e = getenv("EEM_WGRAD_LAST_SIDE"); return e && e[0] == '1'; }();
            hipStream_t keep = bw.wst;
            if (!side) bw.wst = st;
            rc = bw.wgrad(r, l.x, l.xc, 0, l.gy, nullptr, r.cout, 0, 1, n2, l.hin, l.win, l.hout, l.wout);
            bw.wst = keep;
            if (rc != EEM_OK) return rc;
            continue;
        }
        if ((rc = bw.wgrad(r, l.x, l.xc, 0, l.gy, nullptr, r.cout, 0, 1, n2, l.hin, l.win, l.hout, l.wout)) != EEM_OK) return rc;
        if (r.fast_dgrad) {
            EncConvArgs a;
            memset(&a, 0, sizeof(a));
            a.in0 = l.gy;
            a.wpk = c->arena + r.wT_enc; a.wpk2 = c->arena + r.wT_enc2; a.bias = c->arena + r.zero_bias;
            a.wwino = nullptr;
            a.wino_f4 = 0;
            if (c->use_wino && c->enc_wino[l.layer] && (rc = ensure_wino(c, l.layer, 1, s.batch, st, &a.wwino, &a.wino_f4)) != EEM_OK) return rc;
            a.zero_page = c->zero_page; a.trash = c->zero_page + 256;
            a.out = l.gx;
            a.nimg = n2; a.nimg0 = n2;
            a.hin = l.hout; a.win = l.wout; a.hout = l.hin; a.wout = l.win; a.hraw = l.hout; a.wraw = l.wout;
            a.act = 0;
            a.gate = l.x;                                        // -> gradient w.r.t. the previous conv's pre-activation
            {   // (the interleaved tile walk of the batched forward chains, api_internal.h; EEM_WALK3_TRAIN=0 keeps the contiguous ranges)
                static const bool w3 = [] { const char* e = getenv("EEM_WALK3_TRAIN"); return !(e && e[0] == '0'); }();
                if (w3 && n2 >= 4) a.reverse = 3;
            }
            if ((rc = enc_conv_launch(r.cout, r.cin, 1, a, st)) != EEM_OK) return rc;
        } else {
            // stride-2 layers read a stage output, which also feeds the pooling
            const bool first = l.layer == ENC_2_1;
            DgradS2Args d;
            d.dy = l.gy; d.w = c->flat + r.w; d.dx = l.gx; d.gate = l.x;
            d.dpool = c->g_pool[first ? 0 : 1].p; d.pool_k = first ? 32 : 16; d.gh = s.gh; d.gw = s.gw;
            d.zero_page = c->zero_page; d.trash = c->zero_page + 256;
            d.n = n2; d.cin = r.cin; d.cout = r.cout; d.hin = l.hin; d.win = l.win; d.hout = l.hout; d.wout = l.wout;
            if ((l.layer == ENC_2_1 || l.layer == ENC_3_1) && dgrad_s2_supported(d)) {
                if ((rc = dgrad_s2_launch(d, st)) != EEM_OK) return rc;      // conv^T + pooling branch + gate in one kernel
                continue;
            }
            if ((rc = bw.dgrad(r, l.gy, nullptr, r.cout, 0, 1, n2, l.hout, l.wout, l.hin, l.win, l.gx, r.cin, 0)) != EEM_OK) return rc;
            // stride-2 layers read a stage output, which also feeds the pooling: add that branch, then gate
            if (l.layer == ENC_3_1 && (rc = tr_pool_bwd_launch(c->g_pool[1].p, c->g_f12.p, (long)n2 * 32, s.h2, s.w2, 16, s.gh, s.gw, 1, c->f12.p, st)) != EEM_OK) return rc;
            if (l.layer == ENC_2_1 && (rc = tr_pool_bwd_launch(c->g_pool[0].p, c->g_f11.p, (long)n2 * 16, s.h1, s.w1, 32, s.gh, s.gw, 1, c->f11.p, st)) != EEM_OK) return rc;
        }
    }
    return EEM_OK;
}

// Eager forward that keeps every activation for a following eemflow_backward (train-mode output size).
// what eemflow_forward_backward wants cleared before its loss / backward: done by the forward's side-stream prologue when there is one
struct StepZero { float* grad; double* stats; bool done; };

static int forward_train_impl(eemflow_ctx* c, const float* e1, const float* e2, int batch, int in_h, int in_w, int out_h, int out_w,
                              float* flow_out, hipStream_t st, Shape* sout, StepZero* zero = nullptr) {
    Shape s;
    int rc = compute_shape(c, batch, in_h, in_w, out_h, out_w, &s);
    if (rc != EEM_OK) return rc;
    if ((rc = alloc_workspace(c, s)) != EEM_OK) return rc;      // cached inference graphs survive unless a buffer moves
    if ((rc = alloc_train(c, s)) != EEM_OK) return rc;
    c->last = s;
    c->have_last = true;
    c->train_e1 = e1;
    c->train_e2 = e2;
    c->have_train_fwd = false;
    // both volumes replicate-padded into one batch: the first layer's weight gradient needs that copy, and reading the forward's first
    // layer from it lets 346-pixel rows (MVSEC: not a 16-byte multiple, 19 columns of left padding) use the LDS-DMA kernel too
    const int B = batch;
    // Round 6: everything in front of the first conv that does not read the inputs - the Winograd / bf16-piece forms of the weights the
    // optimizer step just changed (three ~5 us launches), and for eemflow_forward_backward the clearing of the gradient buffer and of the
    // loss sums (two fills, each with a ~7 us bubble behind it) - runs on the side stream BESIDE the padding kernel instead of in the
    // chain.  EEM_TRAIN_SIDE_PREP=0 keeps them on st (measurements).
    hipStream_t wst;
    if ((rc = side_stream(c, st, &wst)) != EEM_OK) return rc;
    static const bool side_prep = [] { const char* e = getenv("EEM_TRAIN_SIDE_PREP"); return !(e && e[0] == '0'); }();
    if (wst != st && side_prep) {
        hipEvent_t e = c->wev[c->wev_next++ % eemflow_ctx::kWEvents];
        EEM_HIP_CHECK(hipEventRecord(e, st));                 // (behind the previous step: its backward read these weight forms, its optimizer the gradient)
        EEM_HIP_CHECK(hipStreamWaitEvent(wst, e, 0));
        if (zero) {
            EEM_HIP_CHECK(hipMemsetAsync(zero->grad, 0, c->nflat * sizeof(float), wst));
            EEM_HIP_CHECK(hipMemsetAsync(zero->stats, 0, 8 * sizeof(double), wst));
            zero->done = true;
        }
        if ((rc = ensure_train_wino(c, batch, wst)) != EEM_OK) return rc;
        if (dec_wnc_wanted(c, s.gw, batch) && (rc = ensure_dec_wnc(c, wst)) != EEM_OK) return rc;
        for (int l = 0; l < ENC_NUM; ++l) {
            const float* ws;
            if (c->enc_bx3[l] && bx3_wanted(l) && (rc = ensure_bx3(c, l, wst, &ws)) != EEM_OK) return rc;
        }
        EEM_HIP_CHECK(hipEventRecord(c->prep_ev, wst));
        if ((rc = er_pad2_launch(e1, e2, c->padded.p, B * c->cin0, in_h, in_w, c->pad[0], c->pad[1], c->pad[2], c->pad[3], st)) != EEM_OK) return rc;
        EEM_HIP_CHECK(hipStreamWaitEvent(st, c->prep_ev, 0));
    } else {
        if ((rc = ensure_train_wino(c, batch, st)) != EEM_OK) return rc;
        if (dec_wnc_wanted(c, s.gw, batch) && (rc = ensure_dec_wnc(c, st)) != EEM_OK) return rc;
        if ((rc = er_pad2_launch(e1, e2, c->padded.p, B * c->cin0, in_h, in_w, c->pad[0], c->pad[1], c->pad[2], c->pad[3], st)) != EEM_OK) return rc;
    }
    Hook hk;
    hk.st = st;
    static const bool no_prepad = [] { const char* e = getenv("EEM_NO_PREPAD_FWD"); return e && e[0] == '1'; }();
    c->keep_stage_stores = true;                          // the backward pass reads every activation, f13 included
    rc = run_forward(c, s, e1, e2, flow_out, hk, nullptr, no_prepad ? nullptr : c->padded.p);
    c->keep_stage_stores = false;
    if (rc != EEM_OK) return rc;
    c->have_train_fwd = true;
    c->train_serial += 1;
    c->train_shape = s;
    *sout = s;
    return EEM_OK;
}

extern "C" int eemflow_forward_train(eemflow_ctx* c, const float* e1, const float* e2, int batch, int in_h, int in_w, float* flow_out,
                                     int out_h, int out_w, int64_t* serial_out, void* stream) {
    EEM_REQUIRE(c && e1 && e2 && flow_out, "eemflow_forward_train: NULL argument");
    EEM_REQUIRE(c->weights_loaded && c->have_pad, "eemflow_forward_train: load weights and set the image size first");
    EEM_REQUIRE(c->groups >= 1 && c->groups <= 5, "eemflow_forward_train: groups = %d", c->groups);
    EEM_REQUIRE(batch >= 1 && in_h >= 1 && in_w >= 1 && out_h >= 1 && out_w >= 1, "eemflow_forward_train: bad sizes");
    EEM_HIP_CHECK(hipSetDevice(c->device));
    Shape s;
    int rc = forward_train_impl(c, e1, e2, batch, in_h, in_w, out_h, out_w, flow_out, (hipStream_t)stream, &s);
    if (rc == EEM_OK && serial_out) *serial_out = c->train_serial;
    return rc;
}

extern "C" int eemflow_backward(eemflow_ctx* c, int64_t serial, const float* e1, const float* e2, const float* dflow, float* grad_out,
                                void* stream) {
    EEM_REQUIRE(c && e1 && e2 && dflow && grad_out, "eemflow_backward: NULL argument");
    EEM_REQUIRE(c->have_train_fwd, "eemflow_backward: no eemflow_forward_train has run on this context");
    EEM_REQUIRE(serial == c->train_serial, "eemflow_backward: the activations of forward %lld were overwritten by forward %lld "
                "(one forward per backward and context; run eemflow_forward_train again)", (long long)serial, (long long)c->train_serial);
    EEM_HIP_CHECK(hipSetDevice(c->device));
    return backward_impl(c, c->train_shape, e1, e2, dflow, grad_out, (hipStream_t)stream);
}

// sequence_loss term of one prediction + its gradient, on the device (no host synchronisation): stats6 (device, 6 doubles,
// ACCUMULATED - zero them first): sum of valid |flow - gt|, sum EPE, valid count, count(EPE < 1), count(EPE < 3), count(EPE < 5).
extern "C" int eemflow_sequence_loss(const float* flow, const float* flow_gt, const float* valid, int batch, int h, int w, float weight,
                                     float* dflow_out, double* stats6, void* stream) {
    EEM_REQUIRE(flow && flow_gt && valid && dflow_out && stats6, "eemflow_sequence_loss: NULL argument");
    EEM_REQUIRE(batch >= 1 && h >= 1 && w >= 1, "eemflow_sequence_loss: bad sizes");
    return tr_loss_launch(flow, flow_gt, valid, dflow_out, batch, h * w, weight, stats6, (hipStream_t)stream);
}

// Forward + loss + backward.  grad_out (device, nflat floats, state_dict order) receives d loss / d parameter;
// stats_out (host, 5 doubles): loss, mean epe, valid count, fraction < 1 px, fraction < 3 px.
extern "C" int eemflow_forward_backward(eemflow_ctx* c, const float* e1, const float* e2, const float* flow_gt, const float* valid,
                                        int batch, int in_h, int in_w, int out_h, int out_w, float gamma_weight, float* flow_out,
                                        float* grad_out, double* stats_out, void* stream) {
    EEM_REQUIRE(c && e1 && e2 && flow_gt && valid && grad_out && flow_out, "eemflow_forward_backward: NULL argument");
    EEM_REQUIRE(c->weights_loaded && c->have_pad, "eemflow_forward_backward: load weights and set the image size first");
    EEM_REQUIRE(c->groups >= 1 && c->groups <= 5, "eemflow_forward_backward: groups = %d", c->groups);
    EEM_HIP_CHECK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    Shape s;
    int rc;
    if ((rc = ensure(c->scalars, 24)) != EEM_OK) return rc;
    double* stats = (double*)c->scalars.p;
    StepZero zero{grad_out, stats, false};
    if ((rc = forward_train_impl(c, e1, e2, batch, in_h, in_w, out_h, out_w, flow_out, st, &s, &zero)) != EEM_OK) return rc;
    const int B = batch;
    if (!zero.done) EEM_HIP_CHECK(hipMemsetAsync(stats, 0, 8 * sizeof(double), st));
    // ---- loss and d loss / d flow (train_mvsec.py:201-227)
    // (measured and not kept: the loss and pass X of the upsampling adjoint as one launch, d loss / d flow kept in LDS rows - a wave per
    // output row of both channels.  Grid-stride over 512 blocks: 47 us, what the two launches take (30 + 17) - 2 048 waves do not hide
    // their own round trips; a block per four rows with the six sums finished by the last block behind __threadfence(): 190 us - a
    // device-scope fence per block writes an XCD's L2 back)
    if ((rc = tr_loss_launch(flow_out, flow_gt, valid, c->g_flow.p, B, s.out_h * s.out_w, gamma_weight, stats, st)) != EEM_OK) return rc;
    c->stats_scale = (double)gamma_weight / ((double)B * 2.0 * s.out_h * s.out_w);
    // The five sums are final HERE: without stats_out they leave for pinned host memory on a stream of their own, right behind the loss
    // kernel (an event), while the backward runs - eemflow_train_stats_wait then returns as soon as the host has enqueued the optimizer
    // step, and the next step's launches queue up behind a GPU that is still busy (round 5: the copy sat behind the whole backward, the
    // host woke up when the GPU was already idle, and ~100 us per step passed before the next forward's first launch).
    // EEM_TRAIN_LATE_STATS=1 (read per call) keeps round 5's order.
    const char* late = getenv("EEM_TRAIN_LATE_STATS");
    if (!stats_out && !(late && late[0] == '1')) {
        if (!c->stats_host) {
            EEM_HIP_CHECK(hipHostMalloc((void**)&c->stats_host, 8 * sizeof(double), hipHostMallocDefault));
            EEM_HIP_CHECK(hipEventCreateWithFlags(&c->stats_ev, hipEventDisableTiming));
        }
        if (!c->cstream) {
            EEM_HIP_CHECK(hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking));
            EEM_HIP_CHECK(hipEventCreateWithFlags(&c->loss_ev, hipEventDisableTiming));
        }
        // (measured: letting the copy ride the backward's first fork event instead of one of its own - one event record less in the
        // chain - makes the step 0.04 ms SLOWER: the host returns from eemflow_train_stats_wait that much later)
        EEM_HIP_CHECK(hipEventRecord(c->loss_ev, st));
        EEM_HIP_CHECK(hipStreamWaitEvent(c->cstream, c->loss_ev, 0));
        EEM_HIP_CHECK(hipMemcpyAsync(c->stats_host, stats, 5 * sizeof(double), hipMemcpyDeviceToHost, c->cstream));
        EEM_HIP_CHECK(hipEventRecord(c->stats_ev, c->cstream));
        c->stats_pending = true;
    }
    if ((rc = backward_impl(c, s, e1, e2, c->g_flow.p, grad_out, st, zero.done)) != EEM_OK) return rc;
    if (stats_out) {
        double hst[5];
        EEM_HIP_CHECK(hipMemcpyAsync(hst, stats, sizeof(hst), hipMemcpyDeviceToHost, st));
        EEM_HIP_CHECK(hipStreamSynchronize(st));
        const double cnt = hst[2] > 0 ? hst[2] : 1.0;
        stats_out[0] = hst[0] * c->stats_scale;
        stats_out[1] = hst[1] / cnt; stats_out[2] = hst[2]; stats_out[3] = hst[3] / cnt; stats_out[4] = hst[4] / cnt;
    }
    return EEM_OK;
}

// The statistics of the last eemflow_forward_backward (called with stats_out = NULL) without stalling the stream: _async copies the five
// raw sums to pinned host memory and records an event behind the copy; the caller enqueues whatever follows (all-reduce, optimizer
// step), then _wait blocks on that event only and returns loss, mean EPE, valid count, fractions < 1 px / < 3 px.  With the
// synchronous form the GPU idled ~80 us per step between the backward and the optimizer (tools/fwd_timeline.sh) while the host woke up.
extern "C" int eemflow_train_stats_async(eemflow_ctx* c, void* stream) {
    EEM_REQUIRE(c, "eemflow_train_stats_async: NULL context");
    EEM_REQUIRE(c->scalars.p && c->stats_scale > 0.0, "eemflow_train_stats_async: no eemflow_forward_backward has run");
    EEM_HIP_CHECK(hipSetDevice(c->device));
    if (c->stats_pending) return EEM_OK;                 // (already on their way: eemflow_forward_backward sent them behind the loss kernel)
    if (!c->stats_host) {
        EEM_HIP_CHECK(hipHostMalloc((void**)&c->stats_host, 8 * sizeof(double), hipHostMallocDefault));
        EEM_HIP_CHECK(hipEventCreateWithFlags(&c->stats_ev, hipEventDisableTiming));
    }
    hipStream_t st = (hipStream_t)stream;
    EEM_HIP_CHECK(hipMemcpyAsync(c->stats_host, c->scalars.p, 5 * sizeof(double), hipMemcpyDeviceToHost, st));
    EEM_HIP_CHECK(hipEventRecord(c->stats_ev, st));
    c->stats_pending = true;
    return EEM_OK;
}

extern "C" int eemflow_train_stats_wait(eemflow_ctx* c, double* stats_out) {
    EEM_REQUIRE(c && stats_out, "eemflow_train_stats_wait: NULL argument");
    EEM_REQUIRE(c->stats_pending, "eemflow_train_stats_wait: no eemflow_train_stats_async is outstanding");
    EEM_HIP_CHECK(hipEventSynchronize(c->stats_ev));
    c->stats_pending = false;
    const double* hst = c->stats_host;
    const double cnt = hst[2] > 0 ? hst[2] : 1.0;
    stats_out[0] = hst[0] * c->stats_scale;
    stats_out[1] = hst[1] / cnt; stats_out[2] = hst[2]; stats_out[3] = hst[3] / cnt; stats_out[4] = hst[4] / cnt;
    return EEM_OK;
}

// clip_grad_norm_(max_norm = clip) + AdamW on the device-resident weights, then re-pack (train_mvsec.py:178-183,255-257).
// `grad` is the (already all-reduced, in data-parallel jobs) flat gradient; lr comes from the host-side OneCycle schedule.
extern "C" int eemflow_optimizer_step(eemflow_ctx* c, const float* grad, float lr, float weight_decay, float eps, float clip,
                                      void* stream) {
    EEM_REQUIRE(c && grad, "eemflow_optimizer_step: NULL argument");
    EEM_REQUIRE(c->weights_loaded, "eemflow_optimizer_step: no weights loaded");
    EEM_HIP_CHECK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (c->adam_m.cap < c->nflat) {
        if ((rc = ensure(c->adam_m, c->nflat)) != EEM_OK || (rc = ensure(c->adam_v, c->nflat)) != EEM_OK) return rc;
        EEM_HIP_CHECK(hipMemsetAsync(c->adam_m.p, 0, c->nflat * sizeof(float), st));
        EEM_HIP_CHECK(hipMemsetAsync(c->adam_v.p, 0, c->nflat * sizeof(float), st));
        c->opt_step = 0;
        c->skip_counter_zeroed = false;
    }
    if ((rc = ensure(c->scalars, 24)) != EEM_OK) return rc;
    double* sumsq = (double*)c->scalars.p + 7;
    int* nskip = (int*)(c->scalars.p + 16);                  // steps skipped for a non-finite gradient (device-side count)
    if (!c->skip_counter_zeroed) {
        EEM_HIP_CHECK(hipMemsetAsync(nskip, 0, sizeof(int), st));
        c->skip_counter_zeroed = true;
    }
    if (!c->sumsq_zeroed)                                    // (afterwards the re-packing launch leaves it cleared for the next step)
        EEM_HIP_CHECK(hipMemsetAsync(sumsq, 0, sizeof(double), st));
    c->sumsq_zeroed = false;                                 // (true again once this step's re-packing launch is in the queue)
    if ((rc = tr_sumsq_launch(grad, (long)c->nflat, sumsq, st)) != EEM_OK) return rc;
    c->opt_step += 1;
    if ((rc = tr_adamw_launch(c->flat, grad, c->adam_m.p, c->adam_v.p, (long)c->nflat, sumsq, clip, lr, weight_decay, eps, 0.9f,
                              0.999f, c->opt_step, nskip, st)) != EEM_OK) return rc;
    // (cached graphs read the arena in place: same addresses, new values)
    // (its first thread also counts a skipped step and clears the sum of squares: no launches of their own)
    if ((rc = tr_repack_after_step_launch(c->flat, c->pack_idx, c->arena, (long)c->arena_floats, sumsq, nskip, st)) != EEM_OK) return rc;
    c->sumsq_zeroed = true;
    return refresh_wino(c, st);
}

// Steps eemflow_optimizer_step skipped so far because the gradient held an inf or a NaN (synchronises the stream).
extern "C" int eemflow_optimizer_skipped_steps(eemflow_ctx* c, int* out, void* stream) {
    EEM_REQUIRE(c && out, "eemflow_optimizer_skipped_steps: NULL argument");
    *out = 0;
    if (!c->skip_counter_zeroed) return EEM_OK;
    EEM_HIP_CHECK(hipSetDevice(c->device));
    EEM_HIP_CHECK(hipMemcpyAsync(out, c->scalars.p + 16, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    EEM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return EEM_OK;
}

// Copy the device-resident weights (state_dict order) to `dst` (device) - checkpointing / syncing nn.Parameters.
extern "C" int eemflow_get_weights(eemflow_ctx* c, float* dst, size_t nfloats, void* stream) {
    EEM_REQUIRE(c && dst && c->weights_loaded, "eemflow_get_weights: bad arguments");
    EEM_REQUIRE(nfloats == c->nflat, "eemflow_get_weights: expected %zu floats, got %zu", c->nflat, nfloats);
    EEM_HIP_CHECK(hipMemcpyAsync(dst, c->flat, nfloats * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return EEM_OK;
}
