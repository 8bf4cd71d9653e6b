// C ABI of EEMFlow+ (EEMFlow_cdc, model/EEMFlow/EEMFlow+.py:74-234): encoder, 6-level feature pyramid, and the
// coarse-to-fine loop {cdc_model self-guided upsampling (warp, dense estimator, mask blend) -> warp -> 9x9
// correlation -> decoder + residual}.  Convolutions run on gconv; the torch.cat's are channel offsets.
#include <string.h>

#include <string>
#include <vector>

#include "../../include/eemflow_hip.h"
#include "eraft_kernels.h"
#include "gconv.h"
#include "plus_kernels.h"
#include "wnc.h"

#include <utility>

namespace {

struct PBuf { float* p = nullptr; size_t cap = 0; };

int pensure(PBuf& b, size_t floats) {
    if (floats <= b.cap) return EEM_OK;
    if (b.p) EEM_HIP_CHECK(hipFree(b.p));
    b.p = nullptr; b.cap = 0;
    EEM_HIP_CHECK(hipMalloc(&b.p, floats * sizeof(float)));
    b.cap = floats;
    return EEM_OK;
}

// a buffer whose spare channels are multiplied by zero weights: no NaN bit patterns may lie in it
int pensure_zeroed(PBuf& b, size_t floats) {
    if (floats <= b.cap) return EEM_OK;
    int rc = pensure(b, floats);
    if (rc != EEM_OK) return rc;
    EEM_HIP_CHECK(hipMemset(b.p, 0, floats * sizeof(float)));
    return EEM_OK;
}

struct PLayer {
    size_t wpk = 0, wpk16 = 0, wpkb = 0, wtail = 0, wfew = 0, bias = 0;
    size_t wwnc[3] = {0, 0, 0};          // Winograd streams of the layer's 32-cout slices (conv_wnc.hip)
    bool hasb = false, has16 = false, has_tail = false, has_few = false, has_wnc = false, wnc16 = false;
    int cin = 0, cout = 0, k = 3, stride = 1;
};

const int kTaps[53] = {0,  2,  4,  6,  8,  10, 12, 14, 16, 18, 20, 21, 22, 23, 24, 26, 28, 29, 30, 31, 32, 33, 34, 36, 38, 39, 40,
                       41, 42, 44, 46, 47, 48, 49, 50, 51, 52, 54, 56, 57, 58, 59, 60, 62, 64, 66, 68, 70, 72, 74, 76, 78, 80};
constexpr int kDW = 96, kDIn = 87, kDense = 184;
// The decoder input [53 correlation taps | 32 rconv | 2 flow] lives in a 96-channel buffer and dec1 is packed as 96 -> 96 with zero
// weights for the nine spare channels: 16-aligned, it runs on the LDS-tiled kernel (180 x 320: 170 -> 118 us).  The spare channels hold
// whatever a coarser level left there (finite activations; the buffer is zeroed when allocated), times zero.
constexpr int kCat = 96;

}  // namespace

struct eemplus_ctx {
    int device = 0, cin0 = 15, groups = 3;
    int frames_in_flight = 1;      // eemplus_set_frames_in_flight
    bool loaded = false;
    float* arena = nullptr;
    size_t zero_off = 0;           // zero page inside the arena (LDS-DMA source for padding); a write sink follows 1024 floats in
    // the encoder is EEMFlow's (EEMFlow+.py:100-107 = EEMFlow.py:75-82): its eight layers run on the encoder kernels of conv_enc*.hip /
    // conv_wino*.hip when the first layer has 5 input channels; packed copies of their weights:
    size_t enc_w[8] = {0}, enc_w2[8] = {0}, enc_raw[8] = {0};
    bool enc_has2[8] = {false}, enc_fast = false;
    float* wino = nullptr;
    size_t wino_off[8] = {0};
    bool enc_wino[8] = {false};
    int* taps = nullptr;
    PLayer enc[8], rconv[7], dec1[7], decg[7][3][3], dec5[7], dec6[7], dec7[7], de[6], c1x1[6];
    PBuf padded, f[7], a2, dense, xout, finit[7], tw, fup[7], fw, cat, d[4], t64, t32, flow[7], flow_alt[7];
    // round 6: each level owns its dense-estimator buffer, projection of feature_2 and decoder input (index = level; `dense` above stays the
    // encoder's scratch), so that what a level computes from the feature pyramid alone - the two 1x1 projections and rconv, level_units -
    // CAN run on a side stream beside the coarser levels' chain (EEM_PLUS_SIDE=1; measured slower, see plus_forward_impl)
    PBuf dense_l[7], a2_l[7], cat_l[7];
    size_t cat6_key = 0;                 // (batch, map size) for which cat_l[6]'s two flow channels hold zeros (plus_forward_impl, level 6)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_lvl[7] = {};
    int B = 0, hl[7] = {0}, wl[7] = {0};
    bool have_last = false;
};

namespace {

struct Cur { const float* p; const float* end; const float* take(size_t n) { const float* r = p; p += n; return r; } };
struct Pk {
    std::vector<float> host;
    size_t push(size_t n) { size_t off = host.size(); host.resize(off + ((n + 3) & ~(size_t)3), 0.f); return off; }
};

void mk(Pk& pk, PLayer& L, const float* w, const float* b, int cin, int cout, int k, int stride) {
    L.cin = cin; L.cout = cout; L.k = k; L.stride = stride;
    const int cs[1] = {cin};
    L.wpk = pk.push(gconv_packed_floats(cout, cs, 1, k, k));
    gconv_pack(w, cout, cs, 1, k, k, pk.host.data() + L.wpk);
    L.has16 = gconv16_shape(cout, cs, 1, k, k, stride);
    if (L.has16) {
        L.wpk16 = pk.push(gconv16_packed_floats(cout, cs, 1, k, k));
        gconv16_pack(w, cout, cs, 1, k, k, pk.host.data() + L.wpk16);
    }
    L.hasb = gconvb_shape(cout, cs, 1, k, k, stride);
    if (L.hasb) {
        L.wpkb = pk.push(gconvb_packed_floats(cout, cs, 1, k, k));
        gconvb_pack(w, cout, cs, 1, k, k, pk.host.data() + L.wpkb);
    }
    L.has_few = cout <= 8 && k == 3 && stride == 1;                 // direct convolution on the vector pipe (gconv.h: fewout_*)
    if (L.has_few) {
        L.wfew = pk.push(fewout_packed_floats(cin, k, k));
        fewout_pack(w, cout, cin, k, k, pk.host.data() + L.wfew);
    }
    // 3x3 layers of >= 32 input channels: the F(2x2) Winograd kernel of the fine pyramid levels (32-cout slices; up to 96 couts)
    L.has_wnc = k == 3 && stride == 1 && cin >= 32 && cin <= 32 * WNC_MAX_CHUNKS && cout <= 96;
    if (L.has_wnc) {
        L.wnc16 = cout <= 16;                                        // one job on the 16x16x4 MFMA (half the matrix-pipe time)
        for (int sl = 0; sl * 32 < cout; ++sl) {
            L.wwnc[sl] = pk.push(wnc_packed_floats(cin, L.wnc16));
            wnc_pack(w, cout, cin, sl * 32, L.wnc16, pk.host.data() + L.wwnc[sl]);
        }
    }
    L.bias = pk.push(cout + 32);                                    // (the Winograd kernel reads 32 biases per slice; the spare ones are zeros)
    memcpy(pk.host.data() + L.bias, b, cout * sizeof(float));
    // decoder convs also get the small-grid packing (tail_conv_kernel, used on the coarse pyramid levels)
    L.has_tail = stride == 1 && ((k == 3 && cin <= 184) || (k == 1 && cin <= 100));
    if (L.has_tail) {
        L.wtail = pk.push(tail_packed_floats(cin, cout, k));
        tail_pack_weights(w, cin, cout, k, pk.host.data() + L.wtail);
    }
}

// The fine levels' 3x3 layers on the Winograd kernel (conv_wnc.hip).  Measured per layer (tools/plus_timeline.sh, 1280x720): at 192 x 320
// every layer wins (dense estimator 242 -> 140 us, decoder 337 -> 220); at 96 x 160 (the kernel's 4 x 32 tiles, 120 per job and sample) the
// launches of several jobs win - the decoder's first conv 25 -> 22 us, its grouped layers 16 -> 11, its 96 -> 64 conv 24 -> 16 - and the
// one-job layers are level with the LDS-tiled kernel (120 blocks on 256 CUs); with four samples per call they win too, and so does
// 48 x 80 (1 290 -> 1 374 frames/s).  What decides is how many (tile, job, sample) triples a launch has for the chip: from
// EEM_PLUS_WNC_MINPAIRS (128) on.  EEM_PLUS_WNC_MINPX / EEM_PLUS_WNC_MINPX_JOBS, when set, replace that by the first rule - maps of at
// least so many pixels, for one-job launches / launches of several jobs (defaults 30000 / 10000; 0 sends every level through the kernel:
// the tests).  All read per call.
bool wnc_wanted(int h, int w, int njobs, int n) {
    if (w % 4) return false;
    const char* m = getenv("EEM_PLUS_WNC_MINPX");
    const char* mj = getenv("EEM_PLUS_WNC_MINPX_JOBS");
    const long px = (long)h * w;
    if (m || mj) {
        const long lim1 = m ? atol(m) : 30000L, limj = mj ? atol(mj) : 10000L;
        return px >= lim1 || (njobs >= 2 && px >= limj);
    }
    const char* sm = getenv("EEM_WNC_SMALL_MAXPX");                 // (wnc_launch's choice of block tile: 4 x 32 below it, 4 x 64 from it on)
    const int tw = px < (sm ? atol(sm) : 30000L) ? 32 : 64;
    const char* mp = getenv("EEM_PLUS_WNC_MINPAIRS");
    return (long)((h + 3) / 4) * ((w + tw - 1) / tw) * njobs * n >= (mp ? atol(mp) : 128L);
}
void wnc_common(eemplus_ctx* c, WncArgs& a, int cin, int n, int h, int w, int act) {
    memset(&a, 0, sizeof(a));
    a.nchunks = wnc_chunks(cin, a.chunk_off);
    a.cin = cin;
    a.n = n; a.h = h; a.w = w; a.act = act == GACT_LEAKY;
    a.zero_page = c->arena + c->zero_off;
    a.trash = c->arena + c->zero_off + 1024;
}
// one layer: its 32-cout slices are the jobs
// (policy_jobs: the job count the kernel choice is made for - a group launched on its own decides like the launch of all groups)
bool conv_wnc_args(eemplus_ctx* c, const PLayer& L, const float* in, int in_ctotal, int in_coff, int n, int h, int w, float* out,
                   int out_ctotal, int out_coff, int out_cmul, int act, WncArgs& a, int policy_jobs = 0) {
    const int pj = policy_jobs > 0 ? policy_jobs : (L.cout + 31) / 32;
    if (!L.has_wnc || !wnc_wanted(h, w, pj, n) || (act != GACT_LEAKY && act != GACT_NONE)) return false;
    wnc_common(c, a, L.cin, n, h, w, act);
    a.m16 = L.wnc16;
    const int cm = out_cmul > 1 ? out_cmul : 1;
    for (int sl = 0; sl * 32 < L.cout; ++sl) {
        WncJob& J = a.job[a.njobs++];
        J.in = in; J.in_ctotal = in_ctotal; J.in_coff = in_coff;
        J.w = c->arena + L.wwnc[sl]; J.bias = c->arena + L.bias + sl * 32;
        J.out = out; J.out_ctotal = out_ctotal; J.out_coff = out_coff + sl * 32 * cm; J.out_cmul = cm;
        J.cout = L.cout - sl * 32 < 32 ? L.cout - sl * 32 : 32;
    }
    return wnc_supported(a);
}

int conv(eemplus_ctx* c, const PLayer& L, const float* in, int in_ctotal, int in_coff, int n, int hin, int win, float* out,
         int out_ctotal, int out_coff, int out_cmul, int act, const float* add, hipStream_t st, int policy_jobs = 0) {
    if (add == nullptr) {
        WncArgs wa;
        if (conv_wnc_args(c, L, in, in_ctotal, in_coff, n, hin, win, out, out_ctotal, out_coff, out_cmul, act, wa, policy_jobs)) return wnc_launch(wa, st);
    }
    // small maps (the coarse pyramid levels): the small-grid kernel of EEMFlow's tail
    static const bool no_tail = [] { const char* e = getenv("EEM_PLUS_NO_TAIL"); return e && e[0] == '1'; }();
    const bool add_ok = add == nullptr || (out_ctotal == L.cout && out_coff == 0 && out_cmul <= 1);   // residual indexed like the output
    // EEM_PLUS_TAIL_MAXCIN (read per call; 184): the widest layer the small-grid kernel takes - 100 keeps the dense estimator's four wide
    // convs (128 .. 184 channels) on the LDS-tiled / few-cout kernels, as before round 5
    const char* emc = getenv("EEM_PLUS_TAIL_MAXCIN");
    const int tail_maxcin = emc ? atoi(emc) : 184;
    static const long conv_tail_max = [] { const char* e = getenv("EEM_PLUS_CONV_TAIL_MAX"); return e ? atol(e) : 4096L; }();   // cells
    if (!no_tail && L.has_tail && L.cin <= tail_maxcin && add_ok && (long)hin * win <= conv_tail_max && (act == GACT_LEAKY || act == GACT_NONE)) {
        TailConvLaunch T;
        T.batch = n; T.h = hin; T.w = win; T.ksize = L.k; T.njobs = 1;
        TailConvJob& j = T.job[0];
        j.in = in; j.wpk = c->arena + L.wtail; j.bias = c->arena + L.bias; j.out = out;
        j.cin = L.cin; j.cout = L.cout; j.in_ctotal = in_ctotal; j.in_coff = in_coff;
        j.out_ctotal = out_ctotal; j.out_coff = out_coff; j.out_cmul = out_cmul > 1 ? out_cmul : 1; j.act = act == GACT_LEAKY;
        j.gate = nullptr; j.in_cmul = 1;
        j.add = add;                                               // GEPI_ADD: residual with the output's shape (out_coff 0)
        return tail_conv_launch(T, st);
    }
    GConvArgs a;
    memset(&a, 0, sizeof(a));
    a.nseg = 1;
    a.seg[0].ptr = in; a.seg[0].c = L.cin; a.seg[0].ctotal = in_ctotal; a.seg[0].coff = in_coff;
    a.wpk = c->arena + L.wpk; a.shift = c->arena + L.bias;
    a.wpk16 = L.has16 ? c->arena + L.wpk16 : nullptr;
    a.wpkb = L.hasb ? c->arena + L.wpkb : nullptr;
    a.wfew = L.has_few ? c->arena + L.wfew : nullptr;
    a.zero_page = c->arena + c->zero_off;
    a.out = out; a.out_ctotal = out_ctotal; a.out_coff = out_coff; a.out_cmul = out_cmul;
    const int pad = (L.k - 1) / 2;
    a.n = n; a.hin = hin; a.win = win; a.hout = (hin + 2 * pad - L.k) / L.stride + 1; a.wout = (win + 2 * pad - L.k) / L.stride + 1;
    a.cout = L.cout; a.kh = a.kw = L.k; a.stride = L.stride; a.pad_h = a.pad_w = pad;
    a.act = act; a.out_scale = 1.f;
    a.in_flight = c->frames_in_flight;
    if (add) { a.epi = GEPI_ADD; a.e0 = add; a.e0_ctotal = L.cout; a.e0_coff = 0; }
    return gconv_launch(a, st);
}

// Decoder (EEMFlow+.py:38-71): cat [B][87] -> flow [B][2] (+ residual)
int run_decoder(eemplus_ctx* c, int l, int B, int h, int w, const float* residual, hipStream_t st) {
    float* const cat = c->cat_l[l].p;                 // [53 correlation taps | 32 rconv | 2 flow | 9 spare] of this level
    int rc;
    const size_t g = (size_t)h * w;
    for (int i = 0; i < 4; ++i)
        if ((rc = pensure(c->d[i], B * kDW * g)) != EEM_OK) return rc;
    if ((rc = pensure(c->t64, B * 64 * g)) != EEM_OK || (rc = pensure(c->t32, B * 32 * g)) != EEM_OK ||
        (rc = pensure(c->flow[l], B * 2 * g)) != EEM_OK)
        return rc;
    const int G = c->groups, per = kDW / G;
    // coarse levels: every conv is a handful of 16x16 tiles - the small-grid kernel of EEMFlow's tail (one launch per layer,
    // the groups of a layer as jobs, K split over the waves of a block) instead of one generic-conv launch per group
    static const bool no_tail = [] { const char* e = getenv("EEM_PLUS_NO_TAIL"); return e && e[0] == '1'; }();
    static const long tail_max = [] { const char* e = getenv("EEM_PLUS_TAIL_MAX"); return e ? atol(e) : 4096L; }();
    const bool small = !no_tail && (long)h * w <= tail_max && c->dec1[l].has_tail && c->decg[l][0][0].has_tail && c->dec5[l].has_tail &&
                       c->dec6[l].has_tail && G * 1 <= TAIL_MAX_JOBS;
    if (small) {
        auto job = [&](const PLayer& L, const float* in, int in_ctotal, int in_coff, float* out, int out_ctotal, int out_coff, int out_cmul) {
            TailConvJob j;
            j.in = in; j.wpk = c->arena + L.wtail; j.bias = c->arena + L.bias; j.out = out;
            j.cin = L.cin; j.cout = L.cout; j.in_ctotal = in_ctotal; j.in_coff = in_coff;
            j.out_ctotal = out_ctotal; j.out_coff = out_coff; j.out_cmul = out_cmul; j.act = 1;
            j.gate = nullptr; j.in_cmul = 1; j.add = nullptr;
            return j;
        };
        TailConvLaunch L;
        L.batch = B; L.h = h; L.w = w; L.ksize = 3;
        L.njobs = 1; L.job[0] = job(c->dec1[l], cat, kCat, 0, c->d[0].p, kDW, 0, 1);
        if ((rc = tail_conv_launch(L, st)) != EEM_OK) return rc;
        for (int layer = 0; layer < 3; ++layer) {
            L.njobs = 0;
            for (int gi = 0; gi < G; ++gi)
                L.job[L.njobs++] = job(c->decg[l][layer][gi], c->d[layer].p, kDW, gi * per, c->d[layer + 1].p, kDW, G == 1 ? 0 : gi, G == 1 ? 1 : G);
            if ((rc = tail_conv_launch(L, st)) != EEM_OK) return rc;
        }
        L.njobs = 1; L.job[0] = job(c->dec5[l], c->d[3].p, kDW, 0, c->t64.p, 64, 0, 1);
        if ((rc = tail_conv_launch(L, st)) != EEM_OK) return rc;
        L.njobs = 1; L.job[0] = job(c->dec6[l], c->t64.p, 64, 0, c->t32.p, 32, 0, 1);
        if ((rc = tail_conv_launch(L, st)) != EEM_OK) return rc;
    } else {
        if ((rc = conv(c, c->dec1[l], cat, kCat, 0, B, h, w, c->d[0].p, kDW, 0, 1, GACT_LEAKY, nullptr, st)) != EEM_OK) return rc;
        const char* eng = getenv("EEM_PLUS_NO_GROUPED");             // read per forward: a test runs both forms in one process
        const bool no_grouped = eng && eng[0] == '1';
        for (int layer = 0; layer < 3; ++layer) {
            // the G groups of a layer as ONE launch of the LDS-tiled kernel when their packings lie at equal distances in the arena
            // (they do: same shapes, packed one after the other) and the launch qualifies
            if (G > 1 && !no_grouped && c->decg[l][layer][0].has_wnc && wnc_wanted(h, w, G, B)) {
                // the groups as the jobs of ONE Winograd launch: group gi reads channels [gi*per, (gi+1)*per), its output j goes to j*G + gi
                WncArgs wa;
                wnc_common(c, wa, per, B, h, w, GACT_LEAKY);
                for (int gi = 0; gi < G; ++gi) {
                    const PLayer& Lg = c->decg[l][layer][gi];
                    WncJob& J = wa.job[wa.njobs++];
                    J.in = c->d[layer].p; J.in_ctotal = kDW; J.in_coff = gi * per;
                    J.w = c->arena + Lg.wwnc[0]; J.bias = c->arena + Lg.bias;
                    J.out = c->d[layer + 1].p; J.out_ctotal = kDW; J.out_coff = gi; J.out_cmul = G; J.cout = per;
                }
                if (per == 32 && wnc_supported(wa)) {
                    if ((rc = wnc_launch(wa, st)) != EEM_OK) return rc;
                    continue;
                }
            }
            if (G > 1 && !no_grouped) {
                const PLayer& L0 = c->decg[l][layer][0];
                const PLayer& L1 = c->decg[l][layer][1];
                bool even = L0.has16;
                for (int gi = 1; gi < G && even; ++gi) {
                    const PLayer& Lg = c->decg[l][layer][gi];
                    even = Lg.has16 && Lg.wpk16 - L0.wpk16 == (size_t)gi * (L1.wpk16 - L0.wpk16) && Lg.bias - L0.bias == (size_t)gi * (L1.bias - L0.bias);
                }
                if (even) {
                    GConvArgs a;
                    memset(&a, 0, sizeof(a));
                    a.nseg = 1;
                    a.seg[0].ptr = c->d[layer].p; a.seg[0].c = per; a.seg[0].ctotal = kDW; a.seg[0].coff = 0;
                    a.wpk = c->arena + L0.wpk; a.wpk16 = c->arena + L0.wpk16; a.shift = c->arena + L0.bias;
                    a.zero_page = c->arena + c->zero_off;
                    a.out = c->d[layer + 1].p; a.out_ctotal = kDW; a.out_coff = 0; a.out_cmul = G;
                    a.n = B; a.hin = a.hout = h; a.win = a.wout = w;
                    a.cout = per; a.kh = a.kw = 3; a.stride = 1; a.pad_h = a.pad_w = 1;
                    a.act = GACT_LEAKY; a.out_scale = 1.f; a.in_flight = c->frames_in_flight;
                    a.groups = G; a.g_wstride16 = (long)(L1.wpk16 - L0.wpk16); a.g_pstride = (long)(L1.bias - L0.bias); a.g_ocoff = 1;
                    if (gconv16_supported(a)) {
                        if ((rc = gconv_launch(a, st)) != EEM_OK) return rc;
                        continue;
                    }
                }
            }
            for (int gi = 0; gi < G; ++gi) {
                // group gi reads channels [gi*per, (gi+1)*per); channel_shuffle puts its output j at j*G + gi
                const int oc = G == 1 ? 0 : gi, om = G == 1 ? 1 : G;
                if ((rc = conv(c, c->decg[l][layer][gi], c->d[layer].p, kDW, gi * per, B, h, w, c->d[layer + 1].p, kDW, oc, om, GACT_LEAKY,
                               nullptr, st, G)) != EEM_OK) return rc;
            }
        }
        if ((rc = conv(c, c->dec5[l], c->d[3].p, kDW, 0, B, h, w, c->t64.p, 64, 0, 1, GACT_LEAKY, nullptr, st)) != EEM_OK) return rc;
        if ((rc = conv(c, c->dec6[l], c->t64.p, 64, 0, B, h, w, c->t32.p, 32, 0, 1, GACT_LEAKY, nullptr, st)) != EEM_OK) return rc;
    }
    return conv(c, c->dec7[l], c->t32.p, 32, 0, B, h, w, c->flow[l].p, 2, 0, 1, GACT_NONE, residual, st);
}

}  // namespace

extern "C" int eemplus_create(int device, eemplus_ctx** out) {
    EEM_REQUIRE(out != nullptr, "eemplus_create: out is NULL");
    int ndev = 0;
    EEM_HIP_CHECK(hipGetDeviceCount(&ndev));
    EEM_REQUIRE(device >= 0 && device < ndev, "eemplus_create: device %d of %d", device, ndev);
    EEM_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    EEM_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    EEM_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, "built for gfx950 (MI355X) only; device %d is %s", device, prop.gcnArchName);
    eemplus_ctx* c = new eemplus_ctx();
    c->device = device;
    hipError_t e = hipMalloc(&c->taps, sizeof(kTaps));
    if (e == hipSuccess) e = hipMemcpy(c->taps, kTaps, sizeof(kTaps), hipMemcpyHostToDevice);
    if (e != hipSuccess) { eem_set_error("eemplus_create: %s", hipGetErrorString(e)); delete c; return EEM_ERR_HIP; }
    *out = c;
    return EEM_OK;
}

extern "C" void eemplus_destroy(eemplus_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    PBuf* one[] = {&c->padded, &c->a2, &c->dense, &c->xout, &c->tw, &c->fw, &c->cat, &c->t64, &c->t32,
                   &c->d[0], &c->d[1], &c->d[2], &c->d[3]};
    for (PBuf* b : one) if (b->p) (void)hipFree(b->p);
    if (c->side) {
        (void)hipStreamSynchronize(c->side);
        (void)hipStreamDestroy(c->side);
        (void)hipEventDestroy(c->ev_fork);
        for (hipEvent_t e : c->ev_lvl) if (e) (void)hipEventDestroy(e);
    }
    for (int l = 0; l < 7; ++l) {
        if (c->dense_l[l].p) (void)hipFree(c->dense_l[l].p);
        if (c->a2_l[l].p) (void)hipFree(c->a2_l[l].p);
        if (c->cat_l[l].p) (void)hipFree(c->cat_l[l].p);
        if (c->f[l].p) (void)hipFree(c->f[l].p);
        if (c->fup[l].p) (void)hipFree(c->fup[l].p);
        if (c->finit[l].p) (void)hipFree(c->finit[l].p);
        if (c->flow[l].p) (void)hipFree(c->flow[l].p);
        if (c->flow_alt[l].p) (void)hipFree(c->flow_alt[l].p);
    }
    if (c->arena) (void)hipFree(c->arena);
    if (c->wino) (void)hipFree(c->wino);
    if (c->taps) (void)hipFree(c->taps);
    delete c;
}

extern "C" int eemplus_load_weights(eemplus_ctx* c, const float* flat, size_t nfloats, int n_first_channels, int groups) {
    EEM_REQUIRE(c && flat, "eemplus_load_weights: NULL argument");
    EEM_REQUIRE(n_first_channels >= 1 && n_first_channels <= 64, "eemplus_load_weights: n_first_channels=%d", n_first_channels);
    EEM_REQUIRE(groups == 3 || groups == 1, "eemplus_load_weights: groups must be 3 (reference default) or 1, got %d", groups);
    EEM_HIP_CHECK(hipSetDevice(c->device));
    Cur cur{flat, flat + nfloats};
    Pk pk;
    c->zero_off = pk.push(4096);
    auto layer = [&](PLayer& L, int cin, int cout, int k, int stride) {
        const float* w = cur.take((size_t)cout * cin * k * k);
        const float* b = cur.take(cout);
        mk(pk, L, w, b, cin, cout, k, stride);
    };
    const int ec[8][3] = {{n_first_channels, 16, 2}, {16, 16, 1}, {16, 32, 2}, {32, 32, 1}, {32, 32, 1}, {32, 64, 2}, {64, 64, 1}, {64, 64, 1}};
    c->enc_fast = n_first_channels == 5 && !getenv("EEM_PLUS_GENERIC_ENC");
    for (int i = 0; i < 8; ++i) {
        const float* w = cur.p;
        layer(c->enc[i], ec[i][0], ec[i][1], 3, ec[i][2]);
        if (!c->enc_fast) continue;
        const int cin = ec[i][0], cout = ec[i][1];
        c->enc_w[i] = pk.push(enc_packed_floats(cin, cout));
        enc_pack_weights(w, cin, cout, pk.host.data() + c->enc_w[i]);
        c->enc_has2[i] = enc2_supported(cin, cout, ec[i][2], 4);
        if (c->enc_has2[i]) {
            c->enc_w2[i] = pk.push(enc2_packed_floats(cin, cout));
            enc2_pack_weights(w, cin, cout, pk.host.data() + c->enc_w2[i]);
        }
        c->enc_wino[i] = wino_supported(cin, cout, ec[i][2], 4);
        if (c->enc_wino[i]) {
            c->enc_raw[i] = pk.push((size_t)cout * cin * 9);
            memcpy(pk.host.data() + c->enc_raw[i], w, (size_t)cout * cin * 9 * sizeof(float));
        }
    }
    const int rc_in[7] = {0, 0, 32, 64, 64, 64, 64};
    for (int l = 2; l <= 6; ++l) layer(c->rconv[l], rc_in[l], 32, 3, 1);
    for (int i = 0; i < 4; ++i) (void)cur.take(2 * 2 * 4 * 4 + 2);                     // up3..up6: registered, never used
    const int per = kDW / groups;
    for (int l = 2; l <= 6; ++l) {
        {
            const float* w = cur.take((size_t)kDW * kDIn * 9);
            const float* b = cur.take(kDW);
            std::vector<float> wp((size_t)kDW * kCat * 9, 0.f);
            for (int co = 0; co < kDW; ++co) memcpy(wp.data() + (size_t)co * kCat * 9, w + (size_t)co * kDIn * 9, (size_t)kDIn * 9 * sizeof(float));
            mk(pk, c->dec1[l], wp.data(), b, kCat, kDW, 3, 1);
        }
        for (int j = 0; j < 3; ++j) {
            const float* w = cur.take((size_t)kDW * per * 9);
            const float* b = cur.take(kDW);
            for (int g = 0; g < groups; ++g) mk(pk, c->decg[l][j][g], w + (size_t)g * per * per * 9, b + g * per, per, per, 3, 1);
        }
        layer(c->dec5[l], kDW, 64, 3, 1);
        layer(c->dec6[l], 64, 32, 3, 1);
        layer(c->dec7[l], 32, 2, 3, 1);
    }
    const int de_in[6] = {64, 96, 128, 160, 176, 184}, de_out[6] = {32, 32, 32, 16, 8, 3};
    for (int i = 0; i < 6; ++i) layer(c->de[i], de_in[i], de_out[i], 3, 1);
    (void)cur.take((size_t)16 * 3 * 9 + 16 + 16 * 16 * 9 + 16 + 32 * 16 * 9 + 32 + 32 * 32 * 9 + 32);   // upsample_output_conv: unused
    const int c1_in[6] = {15, 16, 32, 64, 64, 64};
    for (int i = 0; i < 6; ++i) layer(c->c1x1[i], c1_in[i], 32, 1, 1);
    EEM_REQUIRE(cur.p == cur.end, "eemplus_load_weights: the 136-tensor layout needs %zu floats, got %zu", (size_t)(cur.p - flat), nfloats);
    if (c->arena) EEM_HIP_CHECK(hipFree(c->arena));
    c->arena = nullptr;
    EEM_HIP_CHECK(hipMalloc(&c->arena, pk.host.size() * sizeof(float)));
    EEM_HIP_CHECK(hipMemcpy(c->arena, pk.host.data(), pk.host.size() * sizeof(float), hipMemcpyHostToDevice));
    if (c->wino) { EEM_HIP_CHECK(hipFree(c->wino)); c->wino = nullptr; }
    if (c->enc_fast) {
        size_t tot = 0;
        for (int i = 0; i < 8; ++i)
            if (c->enc_wino[i]) { c->wino_off[i] = tot; tot += wino_packed_floats(ec[i][1]); }
        if (tot) {
            EEM_HIP_CHECK(hipMalloc(&c->wino, tot * sizeof(float)));
            for (int i = 0; i < 8; ++i)
                if (c->enc_wino[i]) {
                    const int rcw = wino_transform_launch(c->arena + c->enc_raw[i], ec[i][1], 0, c->wino + c->wino_off[i], nullptr, 0);
                    if (rcw != EEM_OK) return rcw;
                }
            EEM_HIP_CHECK(hipDeviceSynchronize());
        }
    }
    c->cin0 = n_first_channels; c->groups = groups; c->loaded = true;
    return EEM_OK;
}

// the buffers a level owns (see eemplus_ctx::dense_l)
static int level_buffers(eemplus_ctx* c, int l, int B) {
    const size_t g = (size_t)c->hl[l] * c->wl[l];
    int rc;
    if ((rc = pensure_zeroed(c->cat_l[l], B * kCat * g)) != EEM_OK) return rc;
    if (l <= 5 && ((rc = pensure(c->dense_l[l], B * kDense * g)) != EEM_OK || (rc = pensure(c->a2_l[l], B * 32 * g)) != EEM_OK)) return rc;
    return EEM_OK;
}

// What level l computes from the feature pyramid alone (EEMFlow+.py:184-185,191 and :179 for level 6): the 1x1 projections of both
// feature maps - feature_1's goes straight into the dense buffer's x slot - and rconv_l into the decoder input.  Nothing here reads a
// coarser level's flow, so a forward runs these on the side stream while the coarse levels' launch-bound chain has the chip.
static int level_units(eemplus_ctx* c, int l, int B, hipStream_t st, bool skip_rconv = false) {
    int rc;
    const int C[7] = {0, 16, 32, 64, 64, 64, 64};
    const int h = c->hl[l], w = c->wl[l];
    const float* f1 = c->f[l].p;
    const float* f2 = c->f[l].p + (size_t)B * C[l] * h * w;
    if (l <= 5) {
        // (the coarse levels: both projections - the same weights on the two feature maps - as the two jobs of ONE small-grid launch)
        static const bool no_tail = [] { const char* e = getenv("EEM_PLUS_NO_TAIL"); return e && e[0] == '1'; }();
        static const long conv_tail_max = [] { const char* e = getenv("EEM_PLUS_CONV_TAIL_MAX"); return e ? atol(e) : 4096L; }();
        const PLayer& P = c->c1x1[l];
        const char* epf0 = getenv("EEM_PLUS_NO_FUSE");
        if (!no_tail && P.has_tail && (long)h * w <= conv_tail_max && !(epf0 && epf0[0] == '1')) {
            TailConvLaunch T;
            T.batch = B; T.h = h; T.w = w; T.ksize = P.k; T.njobs = 2;
            for (int q = 0; q < 2; ++q) {
                TailConvJob& j = T.job[q];
                j.in = q == 0 ? f1 : f2; j.wpk = c->arena + P.wtail; j.bias = c->arena + P.bias;
                j.out = q == 0 ? c->dense_l[l].p : c->a2_l[l].p;
                j.cin = P.cin; j.cout = P.cout; j.in_ctotal = C[l]; j.in_coff = 0;
                j.out_ctotal = q == 0 ? kDense : 32; j.out_coff = q == 0 ? 120 : 0; j.out_cmul = 1; j.act = 1;
                j.gate = nullptr; j.in_cmul = 1; j.add = nullptr;
            }
            if ((rc = tail_conv_launch(T, st)) != EEM_OK) return rc;
        } else {
            if ((rc = conv(c, P, f1, C[l], 0, B, h, w, c->dense_l[l].p, kDense, 120, 1, GACT_LEAKY, nullptr, st)) != EEM_OK) return rc;
            if ((rc = conv(c, P, f2, C[l], 0, B, h, w, c->a2_l[l].p, 32, 0, 1, GACT_LEAKY, nullptr, st)) != EEM_OK) return rc;
        }
    }
    if (skip_rconv) return EEM_OK;                                 // (run_level: rconv rides the mask estimator's first launch)
    return conv(c, c->rconv[l], f1, C[l], 0, B, h, w, c->cat_l[l].p, kCat, 53, 1, GACT_LEAKY, nullptr, st);
}

// One level l = 5..2 of the coarse-to-fine loop (EEMFlow+.py:183-229) on the features of the current forward: cdc_model
// self-guided upsampling of flow[l+1] -> flow_up[l], warp, 9x9 correlation, decoder + residual -> flow[l].
// units_done: level_units(l) is already in flight (or done) on a stream `st` has been made to wait for.
static int run_level(eemplus_ctx* c, int l, int B, const float* forced_init, hipStream_t st, bool units_done) {
    int rc;
    const int C[7] = {0, 16, 32, 64, 64, 64, 64};
    const int* hl = c->hl; const int* wl = c->wl;
    auto f1 = [&](int k) { return c->f[k].p; };
    auto f2 = [&](int k) { return c->f[k].p + (size_t)B * C[k] * hl[k] * wl[k]; };
    const int h = hl[l], w = wl[l], hc = hl[l + 1], wc = wl[l + 1];
    const size_t g = (size_t)h * w;
    if ((rc = level_buffers(c, l, B)) != EEM_OK ||
        (rc = pensure(c->xout, B * 3 * g)) != EEM_OK || (rc = pensure(c->finit[l], B * 2 * g)) != EEM_OK ||
        (rc = pensure(c->tw, B * 2 * g)) != EEM_OK || (rc = pensure(c->fup[l], B * 2 * g)) != EEM_OK ||
        (rc = pensure(c->fw, B * C[l] * g)) != EEM_OK)
        return rc;
    float* const fi = c->finit[l].p;              // cdc_model's upsampled flow_init, kept per level (stage "flow_init<l>")
    float* const dense = c->dense_l[l].p;
    float* const a2 = c->a2_l[l].p;
    float* const cat = c->cat_l[l].p;
    // coarse levels (small-grid kernel): rconv_l - 64 -> 32 over feature_1, the shape of the mask estimator's first conv - is the second
    // job of that conv's launch instead of a launch of its own (EEM_PLUS_NO_FUSE=1, read per forward: apart)
    bool rconv_rides = false;
    {
        static const bool no_tail = [] { const char* e = getenv("EEM_PLUS_NO_TAIL"); return e && e[0] == '1'; }();
        static const long conv_tail_max = [] { const char* e = getenv("EEM_PLUS_CONV_TAIL_MAX"); return e ? atol(e) : 4096L; }();
        const char* epf0 = getenv("EEM_PLUS_NO_FUSE");
        const char* emc = getenv("EEM_PLUS_TAIL_MAXCIN");
        rconv_rides = !units_done && !no_tail && !(epf0 && epf0[0] == '1') && (long)h * w <= conv_tail_max && c->de[0].has_tail &&
                      c->rconv[l].has_tail && (emc ? atoi(emc) : 184) >= 64 && TAIL_MAX_JOBS >= 2 &&
                      !(c->rconv[l].has_wnc && wnc_wanted(h, w, 1, B));      // (a map forced onto the Winograd kernel keeps rconv there)
    }
    if (!units_done && (rc = level_units(c, l, B, st, rconv_rides)) != EEM_OK) return rc;
    // cdc_model.forward (cdc_utils.py:156-174)
    if (forced_init) {
        // teacher-forced level (eemplus_level): cdc_model's upsampled flow_init is supplied by the caller
        EEM_HIP_CHECK(hipMemcpyAsync(fi, forced_init, B * 2 * g * 4, hipMemcpyDeviceToDevice, st));
    }
    // warp + blend + the copy of flow_up into the decoder's input as one launch (EEM_PLUS_NO_FUSE=1, read per forward: the separate launches)
    const char* epf = getenv("EEM_PLUS_NO_FUSE");
    const bool fuse3 = !(epf && epf[0] == '1');
    if (!forced_init && (hc != h || wc != w) && fuse3 && (size_t)h * w >= 2 * (size_t)hc * wc) {
        // upsampling, the coarse flow's doubling and the warp by the upsampled flow as ONE launch; the doubled coarse flow lands in a
        // second buffer (the launch's other threads still read the plain one) that takes the coarse flow's place from here on
        if ((rc = pensure(c->flow_alt[l + 1], B * 2 * (size_t)hc * wc)) != EEM_OK) return rc;
        if ((rc = pl_upflow_warp_launch(c->flow[l + 1].p, c->flow_alt[l + 1].p, hc, wc, fi, a2, dense, kDense, 152, B, 32, h, w, st)) != EEM_OK) return rc;
        std::swap(c->flow[l + 1], c->flow_alt[l + 1]);
    } else {
        if (forced_init) {
        } else if (hc != h || wc != w) {
            if ((rc = pl_upflow_launch(c->flow[l + 1].p, fi, B, hc, wc, h, w, 1, st)) != EEM_OK) return rc;
            // in-place side effect of upsample2d_flow_as(if_rate=True) on the coarser flow (cdc_utils.py:85-86)
            if ((rc = pl_scale_flow_launch(c->flow[l + 1].p, B, hc * wc, (float)w / (float)wc, (float)h / (float)hc, st)) != EEM_OK) return rc;
        } else {
            EEM_HIP_CHECK(hipMemcpyAsync(fi, c->flow[l + 1].p, B * 2 * g * 4, hipMemcpyDeviceToDevice, st));
        }
        if ((rc = pl_warp_launch(a2, fi, 2, dense, kDense, 152, B, 32, h, w, 2, st)) != EEM_OK) return rc;
    }
    const int din[6] = {64, 96, 128, 160, 176, 184}, dout_off[5] = {88, 56, 24, 8, 0};
    for (int i = 0; i < 5; ++i) {
        if (i == 0 && rconv_rides) {
            TailConvLaunch T;
            T.batch = B; T.h = h; T.w = w; T.ksize = 3; T.njobs = 2;
            for (int q = 0; q < 2; ++q) {
                const PLayer& P = q == 0 ? c->de[0] : c->rconv[l];
                TailConvJob& j = T.job[q];
                j.in = q == 0 ? dense : f1(l); j.wpk = c->arena + P.wtail; j.bias = c->arena + P.bias;
                j.out = q == 0 ? dense : cat;
                j.cin = P.cin; j.cout = P.cout; j.in_ctotal = q == 0 ? kDense : C[l]; j.in_coff = q == 0 ? kDense - din[0] : 0;
                j.out_ctotal = q == 0 ? kDense : kCat; j.out_coff = q == 0 ? dout_off[0] : 53; j.out_cmul = 1; j.act = 1;
                j.gate = nullptr; j.in_cmul = 1; j.add = nullptr;
            }
            if ((rc = tail_conv_launch(T, st)) != EEM_OK) return rc;
            continue;
        }
        if ((rc = conv(c, c->de[i], dense, kDense, kDense - din[i], B, h, w, dense, kDense, dout_off[i], 1, GACT_LEAKY, nullptr, st)) != EEM_OK) return rc;
    }
    if ((rc = conv(c, c->de[5], dense, kDense, 0, B, h, w, c->xout.p, 3, 0, 1, GACT_NONE, nullptr, st)) != EEM_OK) return rc;
    if (fuse3) {
        // ... and the warp of feature_2 by that flow_up (:189) in the same launch
        if ((rc = pl_warp_blend_warp_launch(fi, c->xout.p, c->fup[l].p, cat, kCat, 85, f2(l), c->fw.p, C[l], B, h, w, st)) != EEM_OK) return rc;
    } else {
        if ((rc = pl_warp_launch(fi, c->xout.p, 3, c->tw.p, 2, 0, B, 2, h, w, 1, st)) != EEM_OK) return rc;
        if ((rc = pl_blend_launch(c->tw.p, fi, c->xout.p, c->fup[l].p, B, (int)g, st)) != EEM_OK) return rc;
        // warp, correlate, decode (:189-193)
        if ((rc = pl_warp_launch(f2(l), c->fup[l].p, 2, c->fw.p, C[l], 0, B, C[l], h, w, 0, st)) != EEM_OK) return rc;
    }
    CorrJob cj = {f1(l), c->fw.p, cat, C[l], kCat};
    if ((rc = corr_launch(&cj, 1, B, h, w, c->taps, 53, st)) != EEM_OK) return rc;
    if (!fuse3 && (rc = pl_copy_channels_launch(c->fup[l].p, 2, 0, cat, kCat, 85, 2, B, (int)g, st)) != EEM_OK) return rc;
    if ((rc = run_decoder(c, l, B, h, w, c->fup[l].p, st)) != EEM_OK) return rc;
    return EEM_OK;
}

// One forward over `batch` samples.  frames == 0: events1 / events2 / flow_out are the batched tensors of eemplus_forward.  frames = n > 0
// (eemplus_forward_many): e1v / e2v / outv hold n pointers to single samples - the pad launches read each sample from its own tensor and
// the five full-resolution predictions of sample i go to outv[i] [5][1][2][in_h][in_w]; everything between runs as the batch-n chain.
static int plus_forward_impl(eemplus_ctx* c, const float* e1, const float* e2, const float* const* e1v, const float* const* e2v, int frames,
                             int batch, int in_h, int in_w, const int pad[4], float* out, float* const* outv, hipStream_t st) {
    const int B = batch, n2 = 2 * batch, hp = in_h + pad[2] + pad[3], wp = in_w + pad[0] + pad[1];
    int rc;
    // ---- pad, encoder on both volumes (EEMFlow+.py:162-169)
    if ((rc = pensure(c->padded, (size_t)n2 * c->cin0 * hp * wp)) != EEM_OK) return rc;
    if (frames == 0) {
        if ((rc = er_pad2_launch(e1, e2, c->padded.p, B * c->cin0, in_h, in_w, pad[0], pad[1], pad[2], pad[3], st)) != EEM_OK) return rc;
    } else {
        const size_t img = (size_t)c->cin0 * hp * wp;
        for (int i = 0; i < frames; ++i) {
            if ((rc = er_pad_launch(e1v[i], c->padded.p + (size_t)i * img, c->cin0, in_h, in_w, pad[0], pad[1], pad[2], pad[3], st)) != EEM_OK) return rc;
            if ((rc = er_pad_launch(e2v[i], c->padded.p + (size_t)(B + i) * img, c->cin0, in_h, in_w, pad[0], pad[1], pad[2], pad[3], st)) != EEM_OK) return rc;
        }
    }
    auto half = [](int v) { return (v - 1) / 2 + 1; };
    int* hl = c->hl; int* wl = c->wl;
    hl[1] = half(hp); wl[1] = half(wp); hl[2] = half(hl[1]); wl[2] = half(wl[1]); hl[3] = half(hl[2]); wl[3] = half(wl[2]);
    for (int l = 4; l <= 6; ++l) { hl[l] = hl[l - 1] / 2; wl[l] = wl[l - 1] / 2; }
    EEM_REQUIRE(hl[6] >= 1 && wl[6] >= 1, "eemplus_forward: padded input %dx%d is too small for the 6-level pyramid", hp, wp);
    const int C[7] = {0, 16, 32, 64, 64, 64, 64};
    for (int l = 1; l <= 6; ++l)
        if ((rc = pensure(c->f[l], (size_t)n2 * C[l] * hl[l] * wl[l])) != EEM_OK) return rc;
    // scratch for the two-conv stages: reuse `dense` (large) and `cat`
    {
        const size_t s1 = (size_t)n2 * 16 * hl[1] * wl[1], s2 = (size_t)n2 * 32 * hl[2] * wl[2], s3 = (size_t)n2 * 64 * hl[3] * wl[3];
        size_t big = s1 > s2 ? s1 : s2; big = big > s3 ? big : s3;
        if ((rc = pensure(c->dense, big)) != EEM_OK || (rc = pensure(c->fw, big)) != EEM_OK) return rc;
        float* t0 = c->dense.p; float* t1 = c->fw.p;
        struct EStep { const float* in; float* out; int hin, win, hout, wout; };
        const EStep es[8] = {{c->padded.p, t0, hp, wp, hl[1], wl[1]},     {t0, c->f[1].p, hl[1], wl[1], hl[1], wl[1]},
                             {c->f[1].p, t0, hl[1], wl[1], hl[2], wl[2]}, {t0, t1, hl[2], wl[2], hl[2], wl[2]},
                             {t1, c->f[2].p, hl[2], wl[2], hl[2], wl[2]}, {c->f[2].p, t0, hl[2], wl[2], hl[3], wl[3]},
                             {t0, t1, hl[3], wl[3], hl[3], wl[3]},        {t1, c->f[3].p, hl[3], wl[3], hl[3], wl[3]}};
        for (int i = 0; i < 8; ++i) {
            const PLayer& L = c->enc[i];
            if (c->enc_fast) {
                EncConvArgs a;
                memset(&a, 0, sizeof(a));
                a.in0 = es[i].in; a.in1 = nullptr;
                a.wpk = c->arena + c->enc_w[i];
                a.wpk2 = c->enc_has2[i] ? c->arena + c->enc_w2[i] : nullptr;
                a.wwino = c->enc_wino[i] ? c->wino + c->wino_off[i] : nullptr;
                a.zero_page = c->arena + c->zero_off;
                a.trash = c->arena + c->zero_off + 1024;
                a.bias = c->arena + L.bias;
                a.out = es[i].out;
                a.nimg = n2; a.nimg0 = n2;
                a.hin = es[i].hin; a.win = es[i].win; a.hout = es[i].hout; a.wout = es[i].wout;
                a.hraw = es[i].hin; a.wraw = es[i].win;
                a.act = 1;
                if ((rc = enc_conv_launch(L.cin, L.cout, L.stride, a, st)) != EEM_OK) return rc;
            } else if ((rc = conv(c, L, es[i].in, L.cin, 0, n2, es[i].hin, es[i].win, es[i].out, L.cout, 0, 1, GACT_LEAKY, nullptr, st)) != EEM_OK) {
                return rc;
            }
        }
    }
    // avg_pool2d(2,2) x3 (:170-175), one launch
    if ((rc = er_pool2x3_launch(c->f[3].p, c->f[4].p, c->f[5].p, c->f[6].p, (long)n2 * 64, hl[3], wl[3], st)) != EEM_OK) return rc;

    auto f1 = [&](int l) { return c->f[l].p; };
    auto f2 = [&](int l) { return c->f[l].p + (size_t)B * C[l] * hl[l] * wl[l]; };
    // ---- what the levels compute from the pyramid alone (level_units), on a side stream beside the coarse levels' chain: OPT-IN
    // (EEM_PLUS_SIDE=1, read per forward).  Measured at 1280x720 over 40 forwards: 841 - 866 frames/s against 884 - 891 in the chain - the
    // fork's event record and the five waits cost the chain more than the ~75 us of launches they take out of it, and a forward that
    // starts on an idle GPU is held up by the host enqueueing the side stream's 12 launches first (tools/plus_timeline.sh).
    for (int l = 6; l >= 2; --l)
        if ((rc = level_buffers(c, l, B)) != EEM_OK) return rc;
    const char* ens = getenv("EEM_PLUS_SIDE");
    const bool side = ens && ens[0] == '1';
    if (side) {
        if (!c->side) {
            EEM_HIP_CHECK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
            EEM_HIP_CHECK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
            for (int l = 2; l <= 6; ++l) EEM_HIP_CHECK(hipEventCreateWithFlags(&c->ev_lvl[l], hipEventDisableTiming));
        }
        EEM_HIP_CHECK(hipEventRecord(c->ev_fork, st));        // the pyramid is complete (and the previous forward is through with the buffers)
        EEM_HIP_CHECK(hipStreamWaitEvent(c->side, c->ev_fork, 0));
        for (int l = 6; l >= 2; --l) {
            if ((rc = level_units(c, l, B, c->side)) != EEM_OK) return rc;
            EEM_HIP_CHECK(hipEventRecord(c->ev_lvl[l], c->side));
        }
    }
    // ---- level 6 (:177-181)
    {
        const int h = hl[6], w = wl[6];
        const size_t g = (size_t)h * w;
        float* const cat = c->cat_l[6].p;
        CorrJob cj = {f1(6), f2(6), cat, 64, kCat};
        if ((rc = corr_launch(&cj, 1, B, h, w, c->taps, 53, st)) != EEM_OK) return rc;
        // level 6 has no coarser flow: its two flow channels of the decoder input are zeros (:179).  Nothing writes them at this level, so
        // the fill is launched once per (batch, map size) - the buffer's layout - and not per forward (a re-allocated buffer comes zeroed)
        const size_t key = ((size_t)B << 40) | g;
        if (c->cat6_key != key) {
            if ((rc = pl_copy_channels_launch(nullptr, 0, 0, cat, kCat, 85, 2, B, (int)g, st)) != EEM_OK) return rc;
            c->cat6_key = key;
        }
        if (side) EEM_HIP_CHECK(hipStreamWaitEvent(st, c->ev_lvl[6], 0));
        else if ((rc = level_units(c, 6, B, st)) != EEM_OK) return rc;
        if ((rc = run_decoder(c, 6, B, h, w, nullptr, st)) != EEM_OK) return rc;
    }
    // ---- levels 5..2 (:183-229)
    for (int l = 5; l >= 2; --l) {
        if (side) EEM_HIP_CHECK(hipStreamWaitEvent(st, c->ev_lvl[l], 0));
        if ((rc = run_level(c, l, B, nullptr, st, side)) != EEM_OK) return rc;
    }
    // ---- five full-resolution predictions, coarse to fine (:231-232); flow6..flow3 carry the doubling above
    for (int f = 0; f < (frames ? frames : 1); ++f) {
        const float* ins[5]; float* outs[5]; int hs[5], ws[5];
        for (int i = 0, l = 6; l >= 2; --l, ++i) {
            ins[i] = c->flow[l].p + (frames ? (size_t)f * 2 * hl[l] * wl[l] : 0);
            outs[i] = (frames ? outv[f] : out) + (size_t)i * (frames ? 1 : B) * 2 * in_h * in_w;
            hs[i] = hl[l]; ws[i] = wl[l];
        }
        if ((rc = pl_upflow_multi_launch(ins, outs, hs, ws, 5, frames ? 1 : B, in_h, in_w, 1, st)) != EEM_OK) return rc;
    }
    c->B = B; c->have_last = true;
    return EEM_OK;
}

extern "C" int eemplus_forward(eemplus_ctx* c, const float* e1, const float* e2, int batch, int in_h, int in_w, const int pad[4],
                               float* out, void* stream) {
    EEM_REQUIRE(c && e1 && e2 && out && pad, "eemplus_forward: NULL argument");
    EEM_REQUIRE(c->loaded, "eemplus_forward: no weights loaded");
    EEM_REQUIRE(batch >= 1 && in_h >= 1 && in_w >= 1, "eemplus_forward: bad sizes");
    EEM_HIP_CHECK(hipSetDevice(c->device));
    return plus_forward_impl(c, e1, e2, nullptr, nullptr, 0, batch, in_h, in_w, pad, out, nullptr, (hipStream_t)stream);
}

// n independent samples of the evaluation loop (test_mvsec.py:580-597: one model(events1, events2) per sample at batch 1), each in its
// own tensors, as ONE batch-n chain: the coarse pyramid levels' launches - 60 of a forward's 111, each at the ~4.7 us a dependent launch
// costs - carry n samples instead of one (1280x720: 678 frames/s one sample at a time, 838 / 954 / 1 024 at n = 2 / 4 / 8).
extern "C" int eemplus_forward_many(eemplus_ctx* c, int n, const float* const* events1, const float* const* events2, int in_h, int in_w,
                                    const int pad[4], float* const* flow_out, void* stream) {
    EEM_REQUIRE(c && events1 && events2 && flow_out && pad, "eemplus_forward_many: NULL argument");
    EEM_REQUIRE(c->loaded, "eemplus_forward_many: no weights loaded");
    EEM_REQUIRE(n >= 1 && n <= 16 && in_h >= 1 && in_w >= 1, "eemplus_forward_many: n = %d (1..16) samples of %dx%d", n, in_h, in_w);
    for (int i = 0; i < n; ++i) EEM_REQUIRE(events1[i] && events2[i] && flow_out[i], "eemplus_forward_many: NULL pointer for sample %d", i);
    EEM_HIP_CHECK(hipSetDevice(c->device));
    return plus_forward_impl(c, nullptr, nullptr, events1, events2, n, n, in_h, in_w, pad, nullptr, flow_out, (hipStream_t)stream);
}

extern "C" int eemplus_set_frames_in_flight(eemplus_ctx* c, int n) {
    EEM_REQUIRE(c && n >= 1, "eemplus_set_frames_in_flight: need a context and n >= 1");
    c->frames_in_flight = n;
    return EEM_OK;
}

extern "C" int eemplus_get_stage(eemplus_ctx* c, const char* name, float* dst, size_t cap, int dims[4], void* stream) {
    EEM_REQUIRE(c && name && dims, "eemplus_get_stage: NULL argument");
    EEM_REQUIRE(c->have_last, "eemplus_get_stage: no forward has run");
    const std::string nm(name);
    const float* src = nullptr;
    if (nm.size() == 5 && nm.compare(0, 4, "flow") == 0 && nm[4] >= '2' && nm[4] <= '6') {
        const int l = nm[4] - '0';
        src = c->flow[l].p; dims[0] = c->B; dims[1] = 2; dims[2] = c->hl[l]; dims[3] = c->wl[l];
    } else if (nm.size() == 8 && nm.compare(0, 7, "flow_up") == 0 && nm[7] >= '2' && nm[7] <= '5') {
        const int l = nm[7] - '0';
        src = c->fup[l].p; dims[0] = c->B; dims[1] = 2; dims[2] = c->hl[l]; dims[3] = c->wl[l];
    } else if (nm.size() == 10 && nm.compare(0, 9, "flow_init") == 0 && nm[9] >= '2' && nm[9] <= '5') {
        const int l = nm[9] - '0';
        src = c->finit[l].p; dims[0] = c->B; dims[1] = 2; dims[2] = c->hl[l]; dims[3] = c->wl[l];
    } else {
        eem_set_error("eemplus_get_stage: unknown stage '%s'", name);
        return EEM_ERR_ARG;
    }
    const size_t n = (size_t)dims[0] * dims[1] * dims[2] * dims[3];
    if (dst == nullptr) return EEM_OK;
    EEM_REQUIRE(cap >= n, "eemplus_get_stage: '%s' needs %zu floats, buffer holds %zu", name, n, cap);
    EEM_HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return EEM_OK;
}

// Teacher-forced level: level l = 5..2 of the LAST forward's pyramid re-run from a caller-supplied cdc_model flow_init
// [B][2][h_l][w_l] (the value the `>= 1.0` warp mask is computed from); flow_up_out / flow_out [B][2][h_l][w_l] (either may be NULL)
// receive flow_up_l and flow_l.  The context's flow_l / flow_up_l stages are overwritten.
extern "C" int eemplus_level(eemplus_ctx* c, int level, const float* flow_init, float* flow_up_out, float* flow_out, void* stream) {
    EEM_REQUIRE(c && flow_init, "eemplus_level: NULL argument");
    EEM_REQUIRE(c->have_last, "eemplus_level: no forward has run (the level runs on its feature pyramid)");
    EEM_REQUIRE(level >= 2 && level <= 5, "eemplus_level: level %d (2..5)", level);
    EEM_HIP_CHECK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    int rc = run_level(c, level, c->B, flow_init, st, false);
    if (rc != EEM_OK) return rc;
    const size_t n = (size_t)c->B * 2 * c->hl[level] * c->wl[level] * sizeof(float);
    if (flow_up_out) EEM_HIP_CHECK(hipMemcpyAsync(flow_up_out, c->fup[level].p, n, hipMemcpyDeviceToDevice, st));
    if (flow_out) EEM_HIP_CHECK(hipMemcpyAsync(flow_out, c->flow[level].p, n, hipMemcpyDeviceToDevice, st));
    return EEM_OK;
}

// The three warps as standalone ops: mode 0 EEMFlow_cdc.warp, 1 torch_warp, 2 WarpingLayer_no_div
extern "C" int eemplus_warp(const float* x, const float* flow, int batch, int ch, int h, int w, int mode, float* out, void* stream) {
    EEM_REQUIRE(x && flow && out && mode >= 0 && mode <= 2, "eemplus_warp: bad arguments");
    return pl_warp_launch(x, flow, 2, out, ch, 0, batch, ch, h, w, mode, (hipStream_t)stream);
}

// upsample2d_flow_as(inputs, target, 'bilinear', if_rate): out [b][2][oh][ow]; with if_rate the INPUT is scaled in place afterwards
extern "C" int eemplus_upsample_flow_as(float* inputs, int batch, int h, int w, int oh, int ow, int if_rate, float* out, void* stream) {
    EEM_REQUIRE(inputs && out, "eemplus_upsample_flow_as: NULL argument");
    int rc = pl_upflow_launch(inputs, out, batch, h, w, oh, ow, if_rate, (hipStream_t)stream);
    if (rc != EEM_OK || !if_rate) return rc;
    return pl_scale_flow_launch(inputs, batch, h * w, (float)ow / (float)w, (float)oh / (float)h, (hipStream_t)stream);
}
