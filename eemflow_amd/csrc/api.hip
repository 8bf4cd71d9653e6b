// C ABI of libeemflow_hip.so (declared in include/eemflow_hip.h): context, weight packing,
// workspace management, the forward schedule and its HIP-graph cache.
#include <stdarg.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/eemflow_hip.h"
#include "common.h"

// ------------------------------------------------------------------------------- errors
static thread_local char g_err[512] = "";

void eem_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* eemflow_last_error(void) { return g_err; }
extern "C" int eemflow_abi_version(void) { return 1; }

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
    // weights
    float* arena = nullptr;
    size_t enc_w[ENC_NUM], enc_w2[ENC_NUM], enc_b[ENC_NUM];
    bool enc_has2[ENC_NUM];
    float* zero_page = nullptr;
    TailW rconv[3], dconv1[3], dgroup[3][3][5], dconv5[3], dconv6[3], dconv7[3], outc;
    int* taps = nullptr;
    // workspaces
    DevBuf a1, f11, a2, b2, f12, a3, b3, f13, pool[3], ppart[3], cat[3], ta[3], tb[3], t64[3], t32[3], flowcat, coarse;
    void* vox_scratch = nullptr;
    Shape last;
    bool have_last = false;
    // graph cache
    bool use_graph = true;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    struct Key {
        const float *e1, *e2;
        float* out;
        int batch, in_h, in_w, out_h, out_w, pad[4];
    } graph_key;
    bool have_graph = false;
};

namespace {

int ensure(DevBuf& b, size_t floats) {
    if (floats <= b.cap) return EEM_OK;
    if (b.p) EEM_HIP_CHECK(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    EEM_HIP_CHECK(hipMalloc(&b.p, floats * sizeof(float)));
    b.cap = floats;
    return EEM_OK;
}

void drop_graph(eemflow_ctx* c) {
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    if (c->graph) (void)hipGraphDestroy(c->graph);
    c->graph_exec = nullptr;
    c->graph = nullptr;
    c->have_graph = false;
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
        enc2_tile(d.cin, d.cout, &th, &tw, &pk);
        s->fuse[k] = c->enc_has2[last[k]] && enc2_supported(d.cin, d.cout, d.stride, ws[k]) && pk == ks[k];
        s->th[k] = th;
        s->prow[k] = ceil_div(hs[k], th);
        s->pcol[k] = ceil_div(ws[k], tw) * (tw / ks[k]);
    }
    return EEM_OK;
}

int alloc_workspace(eemflow_ctx* c, const Shape& s) {
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
        ENS(c->t64[k], B * 64 * g);     ENS(c->t32[k], B * 32 * g);
    }
    ENS(c->flowcat, B * 6 * g);  ENS(c->coarse, B * 2 * g);
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
    return j;
}

// Every kernel launch of the schedule goes through a Hook: normally it just launches; in timing mode
// (eemflow_time_kernels) it launches the same kernel `reps` times back to back between two HIP
// events on the launch stream and records the average duration with its algorithmic FLOPs / bytes.
struct Hook {
    hipStream_t st = nullptr;
    bool timing = false;
    int reps = 1;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<eemflow_kernel_stat> stats;

    template <class F>
    int run(const char* name, double flops, double bytes, F&& launch) {
        if (!timing) return launch(st);
        int rc = launch(st);                                  // warm (also keeps data flowing downstream)
        if (rc != EEM_OK) return rc;
        EEM_HIP_CHECK(hipEventRecord(ev0, st));
        for (int i = 0; i < reps; ++i)
            if ((rc = launch(st)) != EEM_OK) return rc;
        EEM_HIP_CHECK(hipEventRecord(ev1, st));
        EEM_HIP_CHECK(hipEventSynchronize(ev1));
        float ms = 0.f;
        EEM_HIP_CHECK(hipEventElapsedTime(&ms, ev0, ev1));
        eemflow_kernel_stat ks;
        memset(&ks, 0, sizeof(ks));
        strncpy(ks.name, name, sizeof(ks.name) - 1);
        ks.flops = flops; ks.bytes = bytes; ks.ms = ms / (float)reps;
        stats.push_back(ks);
        return EEM_OK;
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
    // conv1: 69 -> 100
    L.njobs = 0;
    for (int k = k0; k < k1; ++k) L.job[L.njobs++] = make_job(c, c->dconv1[k], cat[k], kDecIn, 0, c->ta[k].p, kDecW, 0, 1, 1);
    if ((rc = run_tail(hk, "dec.conv1 69->100", L)) != EEM_OK) return rc;
    // conv2..4: grouped 100 -> 100, each followed by channel_shuffle (EEMFlow.py:51-57):
    // group g, in-group channel j lands in channel j*groups + g
    const int G = c->groups, per = kDecW / G;
    const char* gname[3] = {"dec.conv2 grouped+shuffle", "dec.conv3 grouped+shuffle", "dec.conv4 grouped+shuffle"};
    for (int layer = 0; layer < 3; ++layer) {
        L.njobs = 0;
        for (int k = k0; k < k1; ++k) {
            float* src = (layer & 1) ? c->tb[k].p : c->ta[k].p;
            float* dst = (layer & 1) ? c->ta[k].p : c->tb[k].p;
            for (int g = 0; g < G; ++g) {
                if (G == 1) L.job[L.njobs++] = make_job(c, c->dgroup[k][layer][g], src, kDecW, 0, dst, kDecW, 0, 1, 1);
                else L.job[L.njobs++] = make_job(c, c->dgroup[k][layer][g], src, kDecW, g * per, dst, kDecW, g, G, 1);
            }
        }
        if ((rc = run_tail(hk, gname[layer], L)) != EEM_OK) return rc;
    }
    // after three layers the result sits in tb
    L.njobs = 0;
    for (int k = k0; k < k1; ++k) L.job[L.njobs++] = make_job(c, c->dconv5[k], c->tb[k].p, kDecW, 0, c->t64[k].p, 64, 0, 1, 1);
    if ((rc = run_tail(hk, "dec.conv5 100->64", L)) != EEM_OK) return rc;
    L.njobs = 0;
    for (int k = k0; k < k1; ++k) L.job[L.njobs++] = make_job(c, c->dconv6[k], c->t64[k].p, 64, 0, c->t32[k].p, 32, 0, 1, 1);
    if ((rc = run_tail(hk, "dec.conv6 64->32", L)) != EEM_OK) return rc;
    L.njobs = 0;
    for (int k = k0; k < k1; ++k)
        L.job[L.njobs++] = make_job(c, c->dconv7[k], c->t32[k].p, 32, 0, flow_dst, flow_ctotal, 2 * (k - kbase), 1, 0);
    return run_tail(hk, "dec.conv7 32->2", L);
}

int run_forward(eemflow_ctx* c, const Shape& s, const float* e1, const float* e2, float* out, Hook& hk) {
    int rc;
    const int n2 = 2 * s.batch;
    // ---- encoder (both event volumes as one batch; shared weights, EEMFlow.py:135-140)
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
    for (const Step& sp : steps) {
        EncConvArgs a;
        const EncLayerDesc& d = kEncLayers[sp.layer];
        a.in0 = sp.layer == ENC_1_1 ? e1 : sp.in;
        a.in1 = sp.layer == ENC_1_1 ? e2 : nullptr;
        a.wpk = c->arena + c->enc_w[sp.layer];
        a.wpk2 = c->enc_has2[sp.layer] ? c->arena + c->enc_w2[sp.layer] : nullptr;
        a.zero_page = c->zero_page;
        a.trash = c->zero_page + 64;
        a.bias = c->arena + c->enc_b[sp.layer];
        a.out = sp.out;
        a.nimg = n2; a.nimg0 = sp.layer == ENC_1_1 ? s.batch : n2;
        a.hin = sp.hin; a.win = sp.win; a.hout = sp.hout; a.wout = sp.wout;
        a.hraw = sp.layer == ENC_1_1 ? s.in_h : sp.hin;
        a.wraw = sp.layer == ENC_1_1 ? s.in_w : sp.win;
        a.pad_top = sp.layer == ENC_1_1 ? c->pad[2] : 0;
        a.pad_left = sp.layer == ENC_1_1 ? c->pad[0] : 0;
        a.act = 1;
        a.pool_partial = nullptr;
        a.pool_k = 0;
        for (int k = 0; k < 3; ++k)
            if (s.fuse[k] && sp.layer == (k == 0 ? ENC_1_2 : k == 1 ? ENC_2_3 : ENC_3_3)) {
                a.pool_partial = c->ppart[k].p;
                a.pool_k = k == 0 ? 32 : k == 1 ? 16 : 8;
            }
        const double opix = (double)n2 * sp.hout * sp.wout;
        const double flops = 2.0 * opix * d.cout * d.cin * 9;
        const double ipix = sp.layer == ENC_1_1 ? (double)n2 * s.in_h * s.in_w : (double)n2 * sp.hin * sp.win;
        const double bytes = 4.0 * (ipix * d.cin + opix * d.cout + (double)d.cout * d.cin * 9 + d.cout);
        rc = hk.run(sp.name, flops, bytes,
                    [&](hipStream_t st) { return enc_conv_launch(d.cin, d.cout, d.stride, a, st); });
        if (rc != EEM_OK) return rc;
    }
    // ---- stage pooling to the common 1/64 grid (EEMFlow.py:144-154): finish the partial sums the conv
    // epilogues wrote; stages whose conv ran the generic kernel are pooled from the stored feature map
    {
        const float* feat[3] = {c->f11.p, c->f12.p, c->f13.p};
        const int pcs[3] = {16, 32, 64}, hs[3] = {s.h1, s.h2, s.h3}, ws[3] = {s.w1, s.w2, s.w3}, ks[3] = {32, 16, 8};
        PoolFinJob fj[3];
        PoolJob pj[3];
        int nf = 0, np = 0;
        double fin_elems = 0, pool_elems = 0;
        for (int k = 0; k < 3; ++k) {
            if (s.fuse[k]) {
                fj[nf++] = {c->ppart[k].p, c->pool[k].p, pcs[k], s.prow[k], s.pcol[k], ks[k] / s.th[k], ks[k]};
                fin_elems += (double)n2 * pcs[k] * s.gh * s.gw * (ks[k] / s.th[k] + 1);
            } else {
                pj[np++] = {feat[k], c->pool[k].p, pcs[k], hs[k], ws[k], ks[k]};
                pool_elems += (double)n2 * pcs[k] * hs[k] * ws[k];
            }
        }
        if (nf) {
            rc = hk.run("pool finalize (fused partials)", fin_elems, 4.0 * fin_elems, [&](hipStream_t st) {
                return pool_finalize_launch(fj, nf, n2, s.gh, s.gw, st);
            });
            if (rc != EEM_OK) return rc;
        }
        if (np) {
            rc = hk.run("pool 32/16/8", pool_elems, 4.0 * pool_elems,
                        [&](hipStream_t st) { return pool_launch(pj, np, n2, st); });
            if (rc != EEM_OK) return rc;
        }
    }
    // ---- correlation (53 taps) and rconv into the decoders' input [cv | r] (EEMFlow.py:160-163)
    const size_t g = (size_t)s.gh * s.gw;
    const int pc[3] = {16, 32, 64};
    CorrJob cj[3];
    for (int k = 0; k < 3; ++k)
        cj[k] = {c->pool[k].p, c->pool[k].p + (size_t)s.batch * pc[k] * g, c->cat[k].p, pc[k], kDecIn};
    rc = hk.run("local_corr 9x9 (53 taps)", 2.0 * s.batch * g * kNTaps * (16 + 32 + 64),
                4.0 * s.batch * g * (2.0 * (16 + 32 + 64) + 3.0 * kNTaps),
                [&](hipStream_t st) { return corr_launch(cj, 3, s.batch, s.gh, s.gw, c->taps, kNTaps, st); });
    if (rc != EEM_OK) return rc;
    TailConvLaunch L;
    L.batch = s.batch; L.h = s.gh; L.w = s.gw; L.ksize = 3; L.njobs = 0;
    for (int k = 0; k < 3; ++k)
        L.job[L.njobs++] = make_job(c, c->rconv[k], c->pool[k].p, pc[k], 0, c->cat[k].p, kDecIn, kNTaps, 1, 1);
    if ((rc = run_tail(hk, "rconv {16,32,64}->16", L)) != EEM_OK) return rc;
    // ---- decoders, out_conv, upsample (EEMFlow.py:164-181)
    const float* cats[3] = {c->cat[0].p, c->cat[1].p, c->cat[2].p};
    if ((rc = run_decoders(c, 0, 3, cats, s.batch, s.gh, s.gw, c->flowcat.p, 6, 0, hk)) != EEM_OK) return rc;
    L.ksize = 1; L.njobs = 1;
    L.job[0] = make_job(c, c->outc, c->flowcat.p, 6, 0, c->coarse.p, 2, 0, 1, 0);
    if ((rc = run_tail(hk, "out_conv 1x1 6->2", L)) != EEM_OK) return rc;
    const double opix = (double)s.batch * 2 * s.out_h * s.out_w;
    return hk.run("upsample bilinear", 8.0 * opix, 4.0 * (opix + (double)s.batch * 2 * g), [&](hipStream_t st) {
        return upsample_launch(c->coarse.p, out, s.batch * 2, s.gh, s.gw, s.out_h, s.out_w, st);
    });
}

}  // namespace

// ------------------------------------------------------------------------------- C ABI
extern "C" int eemflow_create(int device, eemflow_ctx** out) {
    EEM_REQUIRE(out != nullptr, "eemflow_create: out is NULL");
    int ndev = 0;
    EEM_HIP_CHECK(hipGetDeviceCount(&ndev));
    EEM_REQUIRE(device >= 0 && device < ndev, "eemflow_create: device %d of %d", device, ndev);
    EEM_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    EEM_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    EEM_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
                "this library is built for gfx950 (MI355X) only; device %d is %s", device, prop.gcnArchName);
    eemflow_ctx* c = new eemflow_ctx();
    c->device = device;
    hipError_t e = hipMalloc(&c->taps, sizeof(kTaps53));
    if (e == hipSuccess) e = hipMemcpy(c->taps, kTaps53, sizeof(kTaps53), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&c->vox_scratch, voxel_scratch_bytes());
    if (e == hipSuccess) e = hipMalloc(&c->zero_page, 1024);
    if (e == hipSuccess) e = hipMemset(c->zero_page, 0, 1024);
    if (e != hipSuccess) {
        eem_set_error("eemflow_create: %s", hipGetErrorString(e));
        delete c;
        return EEM_ERR_HIP;
    }
    *out = c;
    return EEM_OK;
}

extern "C" void eemflow_destroy(eemflow_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    drop_graph(c);
    DevBuf* bufs[] = {&c->a1, &c->f11, &c->a2, &c->b2, &c->f12, &c->a3, &c->b3, &c->f13, &c->flowcat, &c->coarse};
    for (DevBuf* b : bufs) if (b->p) (void)hipFree(b->p);
    for (int k = 0; k < 3; ++k) {
        if (c->ppart[k].p) (void)hipFree(c->ppart[k].p);
        DevBuf* kb[] = {&c->pool[k], &c->cat[k], &c->ta[k], &c->tb[k], &c->t64[k], &c->t32[k]};
        for (DevBuf* b : kb) if (b->p) (void)hipFree(b->p);
    }
    if (c->arena) (void)hipFree(c->arena);
    if (c->taps) (void)hipFree(c->taps);
    if (c->zero_page) (void)hipFree(c->zero_page);
    if (c->vox_scratch) (void)hipFree(c->vox_scratch);
    delete c;
}

extern "C" int eemflow_load_weights(eemflow_ctx* c, const float* flat, size_t nfloats, int n_first_channels,
                                    int groups) {
    EEM_REQUIRE(c && flat, "eemflow_load_weights: NULL argument");
    EEM_REQUIRE(n_first_channels == 5, "only n_first_channels == 5 (num_voxel_bins 5, config/a_meshflow.json) is "
                                       "built; got %d", n_first_channels);
    EEM_REQUIRE(groups == 5 || groups == 1, "groups must be 5 (reference default) or 1; got %d", groups);
    EEM_HIP_CHECK(hipSetDevice(c->device));
    const int per = kDecW / groups;
    // ---- expected flat size
    size_t expect = 0;
    for (int l = 0; l < ENC_NUM; ++l) {
        const int cin = l == 0 ? n_first_channels : kEncLayers[l].cin;
        expect += (size_t)kEncLayers[l].cout * cin * 9 + kEncLayers[l].cout;
    }
    const int rc_in[3] = {16, 32, 64};
    for (int k = 0; k < 3; ++k) expect += (size_t)16 * rc_in[k] * 9 + 16;
    const size_t dec = (size_t)kDecW * kDecIn * 9 + kDecW + 3 * ((size_t)kDecW * per * 9 + kDecW) +
                       (size_t)64 * kDecW * 9 + 64 + (size_t)32 * 64 * 9 + 32 + (size_t)2 * 32 * 9 + 2;
    expect += 3 * dec + 2 * 6 + 2;
    EEM_REQUIRE(nfloats == expect, "eemflow_load_weights: expected %zu floats for the 66-tensor layout, got %zu",
                expect, nfloats);

    // ---- pack everything into one host arena, then upload once
    std::vector<float> host;
    auto push = [&host](size_t n) { size_t off = host.size(); host.resize(off + ((n + 3) & ~(size_t)3), 0.f); return off; };
    const float* p = flat;
    for (int l = 0; l < ENC_NUM; ++l) {
        const int cin = l == 0 ? n_first_channels : kEncLayers[l].cin, cout = kEncLayers[l].cout;
        c->enc_w[l] = push(enc_packed_floats(cin, cout));
        enc_pack_weights(p, cin, cout, host.data() + c->enc_w[l]);
        c->enc_has2[l] = enc2_supported(cin, cout, kEncLayers[l].stride, 4);
        if (c->enc_has2[l]) {
            c->enc_w2[l] = push(enc2_packed_floats(cin, cout));
            enc2_pack_weights(p, cin, cout, host.data() + c->enc_w2[l]);
        }
        p += (size_t)cout * cin * 9;
        c->enc_b[l] = push(cout);
        memcpy(host.data() + c->enc_b[l], p, cout * sizeof(float));
        p += cout;
    }
    auto tail = [&](TailW& t, int cin, int cout, int ksize, const float* w, const float* b) {
        t.cin = cin; t.cout = cout; t.ksize = ksize;
        t.wpk = push(tail_packed_floats(cin, cout, ksize));
        tail_pack_weights(w, cin, cout, ksize, host.data() + t.wpk);
        t.bias = push(cout);
        memcpy(host.data() + t.bias, b, cout * sizeof(float));
    };
    for (int k = 0; k < 3; ++k) {
        tail(c->rconv[k], rc_in[k], 16, 3, p, p + (size_t)16 * rc_in[k] * 9);
        p += (size_t)16 * rc_in[k] * 9 + 16;
    }
    for (int k = 0; k < 3; ++k) {
        tail(c->dconv1[k], kDecIn, kDecW, 3, p, p + (size_t)kDecW * kDecIn * 9);
        p += (size_t)kDecW * kDecIn * 9 + kDecW;
        for (int layer = 0; layer < 3; ++layer) {
            const float* w = p;
            const float* b = p + (size_t)kDecW * per * 9;
            for (int g = 0; g < groups; ++g)     // group g = output channels [g*per, (g+1)*per), its own `per` inputs
                tail(c->dgroup[k][layer][g], per, per, 3, w + (size_t)g * per * per * 9, b + g * per);
            p += (size_t)kDecW * per * 9 + kDecW;
        }
        tail(c->dconv5[k], kDecW, 64, 3, p, p + (size_t)64 * kDecW * 9);  p += (size_t)64 * kDecW * 9 + 64;
        tail(c->dconv6[k], 64, 32, 3, p, p + (size_t)32 * 64 * 9);        p += (size_t)32 * 64 * 9 + 32;
        tail(c->dconv7[k], 32, 2, 3, p, p + (size_t)2 * 32 * 9);          p += (size_t)2 * 32 * 9 + 2;
    }
    tail(c->outc, 6, 2, 1, p, p + 12);
    p += 14;
    if ((size_t)(p - flat) != nfloats) {
        eem_set_error("eemflow_load_weights: internal layout walk consumed %zu of %zu floats", (size_t)(p - flat), nfloats);
        return EEM_ERR_STATE;
    }
    drop_graph(c);
    if (c->arena) EEM_HIP_CHECK(hipFree(c->arena));
    c->arena = nullptr;
    EEM_HIP_CHECK(hipMalloc(&c->arena, host.size() * sizeof(float)));
    EEM_HIP_CHECK(hipMemcpy(c->arena, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    c->cin0 = n_first_channels;
    c->groups = groups;
    c->weights_loaded = true;
    return EEM_OK;
}

extern "C" int eemflow_set_image_size(eemflow_ctx* c, int height, int width, int pad_out[4]) {
    EEM_REQUIRE(c, "eemflow_set_image_size: NULL context");
    EEM_REQUIRE(height > 0 && width > 0, "eemflow_set_image_size: %dx%d", height, width);
    const int r = 64;   // eval_pad_rate (EEMFlow.py:116); formula utils/image_utils.py:132-137, 'chairs' mode
    const int pad_ht = (((height / r) + 1) * r - height) % r;
    const int pad_wd = (((width / r) + 1) * r - width) % r;
    c->pad[0] = pad_wd / 2; c->pad[1] = pad_wd - pad_wd / 2; c->pad[2] = 0; c->pad[3] = pad_ht;
    c->have_pad = true;
    if (pad_out) memcpy(pad_out, c->pad, sizeof(c->pad));
    return EEM_OK;
}

extern "C" int eemflow_use_graph(eemflow_ctx* c, int enable) {
    EEM_REQUIRE(c, "eemflow_use_graph: NULL context");
    c->use_graph = enable != 0;
    if (!c->use_graph) drop_graph(c);
    return EEM_OK;
}

extern "C" int eemflow_forward(eemflow_ctx* c, const float* e1, const float* e2, int batch, int in_h, int in_w,
                               float* out, int out_h, int out_w, void* stream) {
    EEM_REQUIRE(c && e1 && e2 && out, "eemflow_forward: NULL argument");
    EEM_REQUIRE(c->weights_loaded, "eemflow_forward: no weights loaded");
    EEM_REQUIRE(c->have_pad, "eemflow_forward: call eemflow_set_image_size first (the reference needs "
                             "change_imagesize before forward too)");
    EEM_REQUIRE(batch >= 1 && in_h >= 1 && in_w >= 1 && out_h >= 1 && out_w >= 1, "eemflow_forward: bad sizes");
    EEM_HIP_CHECK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    Shape s;
    int rc = compute_shape(c, batch, in_h, in_w, out_h, out_w, &s);
    if (rc != EEM_OK) return rc;

    eemflow_ctx::Key key = {e1, e2, out, batch, in_h, in_w, out_h, out_w, {c->pad[0], c->pad[1], c->pad[2], c->pad[3]}};
    if (c->use_graph && c->have_graph && memcmp(&key, &c->graph_key, sizeof(key)) == 0) {
        EEM_HIP_CHECK(hipGraphLaunch(c->graph_exec, st));
        return EEM_OK;
    }
    // (re)allocation invalidates pointers baked into a cached graph
    drop_graph(c);
    if ((rc = alloc_workspace(c, s)) != EEM_OK) return rc;
    c->last = s;
    c->have_last = true;
    Hook hk;
    hk.st = st;
    if (!c->use_graph) return run_forward(c, s, e1, e2, out, hk);

    hipStream_t cap = st;
    bool own_stream = false;
    if (cap == nullptr) {        // the legacy default stream cannot be captured
        EEM_HIP_CHECK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
        own_stream = true;
    }
    EEM_HIP_CHECK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
    hk.st = cap;
    rc = run_forward(c, s, e1, e2, out, hk);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(cap, &g);
    if (own_stream) (void)hipStreamDestroy(cap);
    if (rc != EEM_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) { eem_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return EEM_ERR_HIP; }
    c->graph = g;
    EEM_HIP_CHECK(hipGraphInstantiate(&c->graph_exec, c->graph, nullptr, nullptr, 0));
    c->graph_key = key;
    c->have_graph = true;
    EEM_HIP_CHECK(hipGraphLaunch(c->graph_exec, st));
    return EEM_OK;
}

extern "C" int eemflow_time_kernels(eemflow_ctx* c, const float* e1, const float* e2, int batch, int in_h, int in_w,
                                    float* out, int out_h, int out_w, int reps, eemflow_kernel_stat* stats, int max_stats,
                                    int* nstats, void* stream) {
    EEM_REQUIRE(c && e1 && e2 && out && stats && nstats, "eemflow_time_kernels: NULL argument");
    EEM_REQUIRE(c->weights_loaded && c->have_pad, "eemflow_time_kernels: load weights and set the image size first");
    EEM_REQUIRE(reps >= 1, "eemflow_time_kernels: reps=%d", reps);
    EEM_HIP_CHECK(hipSetDevice(c->device));
    Shape s;
    int rc = compute_shape(c, batch, in_h, in_w, out_h, out_w, &s);
    if (rc != EEM_OK) return rc;
    drop_graph(c);
    if ((rc = alloc_workspace(c, s)) != EEM_OK) return rc;
    c->last = s;
    c->have_last = true;
    Hook hk;
    hk.st = (hipStream_t)stream;
    hk.timing = true;
    hk.reps = reps;
    EEM_HIP_CHECK(hipEventCreate(&hk.ev0));
    EEM_HIP_CHECK(hipEventCreate(&hk.ev1));
    rc = run_forward(c, s, e1, e2, out, hk);
    (void)hipEventDestroy(hk.ev0);
    (void)hipEventDestroy(hk.ev1);
    if (rc != EEM_OK) return rc;
    *nstats = (int)hk.stats.size();
    EEM_REQUIRE(*nstats <= max_stats, "eemflow_time_kernels: %d kernels, room for %d", *nstats, max_stats);
    memcpy(stats, hk.stats.data(), hk.stats.size() * sizeof(eemflow_kernel_stat));
    return EEM_OK;
}

extern "C" int eemflow_get_stage(eemflow_ctx* c, const char* name, float* dst, size_t cap, int dims[4], void* stream) {
    EEM_REQUIRE(c && name && dims, "eemflow_get_stage: NULL argument");
    EEM_REQUIRE(c->have_last, "eemflow_get_stage: no forward has run");
    const Shape& s = c->last;
    const int n2 = 2 * s.batch;
    const float* src = nullptr;
    std::string nm(name);
    if (nm == "f11") { src = c->f11.p; dims[0] = n2; dims[1] = 16; dims[2] = s.h1; dims[3] = s.w1; }
    else if (nm == "f12") { src = c->f12.p; dims[0] = n2; dims[1] = 32; dims[2] = s.h2; dims[3] = s.w2; }
    else if (nm == "f13") { src = c->f13.p; dims[0] = n2; dims[1] = 64; dims[2] = s.h3; dims[3] = s.w3; }
    else if (nm == "a1") { src = c->a1.p; dims[0] = n2; dims[1] = 16; dims[2] = s.h1; dims[3] = s.w1; }
    else if (nm == "flowcat") { src = c->flowcat.p; dims[0] = s.batch; dims[1] = 6; dims[2] = s.gh; dims[3] = s.gw; }
    else if (nm == "coarse") { src = c->coarse.p; dims[0] = s.batch; dims[1] = 2; dims[2] = s.gh; dims[3] = s.gw; }
    else if (nm.size() == 6 && nm.compare(0, 5, "pool_") == 0 && nm[5] >= '1' && nm[5] <= '3') {
        const int k = nm[5] - '1';
        const int pc[3] = {16, 32, 64};
        src = c->pool[k].p; dims[0] = n2; dims[1] = pc[k]; dims[2] = s.gh; dims[3] = s.gw;
    } else if (nm.size() == 5 && nm.compare(0, 4, "cat_") == 0 && nm[4] >= '1' && nm[4] <= '3') {
        src = c->cat[nm[4] - '1'].p; dims[0] = s.batch; dims[1] = kDecIn; dims[2] = s.gh; dims[3] = s.gw;
    } else {
        eem_set_error("eemflow_get_stage: unknown stage '%s'", name);
        return EEM_ERR_ARG;
    }
    const size_t n = (size_t)dims[0] * dims[1] * dims[2] * dims[3];
    if (dst == nullptr) return EEM_OK;       // size query
    EEM_REQUIRE(cap >= n, "eemflow_get_stage: '%s' needs %zu floats, buffer holds %zu", name, n, cap);
    EEM_HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return EEM_OK;
}

extern "C" int eemflow_decoder(eemflow_ctx* c, int k, const float* x, int batch, int h, int w, float* out, void* stream) {
    EEM_REQUIRE(c && x && out, "eemflow_decoder: NULL argument");
    EEM_REQUIRE(c->weights_loaded, "eemflow_decoder: no weights loaded");
    EEM_REQUIRE(k >= 1 && k <= 3 && batch >= 1 && h >= 1 && w >= 1, "eemflow_decoder: bad arguments");
    EEM_HIP_CHECK(hipSetDevice(c->device));
    drop_graph(c);                           // may grow the shared scratch buffers
    const size_t g = (size_t)h * w, B = batch;
    int rc;
    const int i = k - 1;
    if ((rc = ensure(c->ta[i], B * kDecW * g)) || (rc = ensure(c->tb[i], B * kDecW * g)) ||
        (rc = ensure(c->t64[i], B * 64 * g)) || (rc = ensure(c->t32[i], B * 32 * g)))
        return rc;
    const float* cats[3] = {x, x, x};
    Hook hk;
    hk.st = (hipStream_t)stream;
    return run_decoders(c, i, i + 1, cats, batch, h, w, out, 2, i, hk);
}

extern "C" int eemflow_local_corr53(const float* f1, const float* f2, int batch, int cch, int h, int w, float* out,
                                    void* stream) {
    EEM_REQUIRE(f1 && f2 && out && batch >= 1 && cch >= 1 && h >= 1 && w >= 1, "eemflow_local_corr53: bad arguments");
    static thread_local int* taps = nullptr;
    static thread_local int taps_dev = -1;
    int dev = 0;
    EEM_HIP_CHECK(hipGetDevice(&dev));
    if (taps == nullptr || taps_dev != dev) {
        EEM_HIP_CHECK(hipMalloc(&taps, sizeof(kTaps53)));
        EEM_HIP_CHECK(hipMemcpy(taps, kTaps53, sizeof(kTaps53), hipMemcpyHostToDevice));
        taps_dev = dev;
    }
    CorrJob j = {f1, f2, out, cch, kNTaps};
    return corr_launch(&j, 1, batch, h, w, taps, kNTaps, (hipStream_t)stream);
}

extern "C" int eemflow_upsample_bilinear(const float* in, float* out, int nc, int h, int w, int oh, int ow, void* stream) {
    EEM_REQUIRE(in && out && nc >= 1 && h >= 1 && w >= 1 && oh >= 1 && ow >= 1, "eemflow_upsample_bilinear: bad arguments");
    return upsample_launch(in, out, nc, h, w, oh, ow, (hipStream_t)stream);
}

extern "C" int eemflow_voxelize(const double* events, int64_t n, int bins, int h, int w, int normalize, float* grid,
                                int64_t* idx_left, int64_t* idx_right, void* stream) {
    EEM_REQUIRE(events && grid, "eemflow_voxelize: NULL argument");
    static thread_local void* scratch = nullptr;
    static thread_local int scratch_dev = -1;
    int dev = 0;
    EEM_HIP_CHECK(hipGetDevice(&dev));
    if (scratch == nullptr || scratch_dev != dev) {
        EEM_HIP_CHECK(hipMalloc(&scratch, voxel_scratch_bytes()));
        scratch_dev = dev;
    }
    return voxel_launch(events, n, bins, h, w, normalize, grid, idx_left, idx_right, scratch, (hipStream_t)stream);
}
