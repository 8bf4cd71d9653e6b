// C ABI for the E-RAFT part of the path (declared in include/eemflow_hip.h): context, weight parsing with
// eval-mode BatchNorm folding, packing, workspace and the forward schedule of model/eraft.py:97-159.
#include <string.h>

#include <string>
#include <algorithm>
#include <vector>

#include "../../include/eemflow_hip.h"
#include "eraft_kernels.h"
#include "gconv.h"
#include "wnc.h"

namespace {

struct Buf {
    float* p = nullptr;
    size_t cap = 0;
};

int ensure(Buf& b, size_t floats) {
    if (floats <= b.cap) return EEM_OK;
    if (b.p) EEM_HIP_CHECK(hipFree(b.p));
    b.p = nullptr; b.cap = 0;
    EEM_HIP_CHECK(hipMalloc(&b.p, floats * sizeof(float)));
    b.cap = floats;
    return EEM_OK;
}

struct Layer {                    // one convolution, weights packed for gconv
    size_t wpk = 0, wpk16 = 0, wpkb = 0, wfew = 0, wstem = 0, scale = 0, shift = 0;
    bool hasb = false, has_stem = false;
    size_t wraw = 0, wf4 = 0;      // 64 -> 64 3x3 stride-1 layers: OIHW weights (BatchNorm scale folded in) and their F(4x4,3x3) form
    bool has_scale = false, has16 = false, has_few = false, has_f4 = false;
    size_t wwnc[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // 96 -> 96 / 128 -> 128 3x3 stride-1 layers: F(2x2) streams of their 32-cout slices (conv_wnc.hip), BatchNorm scale folded in
    bool has_wnc = false;
    int cout = 0, kh = 1, kw = 1, stride = 1, ph = 0, pw = 0;
    int cs[3] = {0, 0, 0}, nseg = 1;
};

struct Block {                    // ResidualBlock (model/extractor.py:7-57)
    Layer conv1, conv2, down;
    bool has_down = false;
};

struct Encoder {                  // BasicEncoder (model/extractor.py:119-190)
    Layer conv1, conv2a, conv2b;  // conv2b: second half of the output conv (cnet: net | inp split)
    Block blk[6];
    bool batch_norm = false;
};

}  // namespace

struct eraft_ctx {
    int device = 0;
    bool loaded = false;
    int cin0 = 5;
    float* arena = nullptr;
    size_t zero_off = 0;           // 64 zero floats inside the arena (LDS-DMA source for padding)
    Encoder fnet, cnet;
    Layer convc1, convc2, convf1, convf2, conv, gz[2], gr[2], gq[2], fh1, fh2, mk0, mk2;
    Layer gzr[2], heads1;          // z | r of a GRU pass and flow-head | mask-head conv1 as ONE launch each (small batches)
    // The GRU's convs read hx = [h | inp | motion] (model/update.py:43-60) and `inp` - the context features - is the same in all twelve
    // iterations: its third of every conv (the weights' input channels 128..255, + the bias) is computed ONCE per forward into a
    // per-pixel addend (gzr_c / gq_c -> czr / cq), and the convs inside the loop run over [h | motion] only (gzr_hm / gq_hm: K = 256
    // instead of 384) with that addend in front of their activation (GConvArgs::pre).  EEM_ERAFT_NO_PRE=1 keeps the full convs.
    Layer gzr_hm[2], gq_hm[2], gzr_c[2], gq_c[2];
    Buf czr[2], cq[2];
    // workspace
    Buf padded, s[5], fmap, net[3], inp, pyr[4], c0, c1, c1b, corr, cor1, corflo, flo1, motion, z, rh, fhid, delta, mhid, mask;
    Buf st_corr0, st_net1, st_mask1, st_delta1, zeros;
    Buf many_out;                  // eraft_forward_many: the batched predictions before they are copied to the samples' own tensors
    int B = 0, h8 = 0, w8 = 0, ph[4] = {0, 0, 0, 0}, pw[4] = {0, 0, 0, 0};
    bool have_last = false;
    bool keep_stages = false;      // copy corr0 / net1 / mask1 / delta1 aside in the first iteration (parity tests)
    const float* c1last = nullptr;   // coords1 after the last iteration of the last forward (c1 or c1b)
    int frames_in_flight = 1;      // eraft_set_frames_in_flight
    float* wino = nullptr;         // F(4x4,3x3) forms of the 64 -> 64 encoder convs (wino4_transform_launch at load time)
    float* trash = nullptr;        // 1 KB sink for the Winograd kernel's out-of-image stores
    double* nstat = nullptr;       // per-chunk sums of the large-plane instance norm (er_instnorm_launch)
    size_t nstat_cap = 0;
    // independent branches of the forward on a context-owned side stream (fork / join by events): the context network beside the feature
    // network and the correlation volume, convf1 -> convf2 beside convc1 -> convc2, the flow head's last conv beside the mask head's
    hipStream_t side = nullptr;
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    Buf s2[5];                     // the context network's own activations (the feature network runs at the same time)
    bool alt_corr = false;         // eraft_set_alternate_corr: correlation features on the fly, no all-pairs volume
    bool final_only = false;       // eraft_set_final_only: only the last iteration's prediction leaves the forward
    Buf f2l[3];                    // avg-pooled fmap2, levels 1..3 (alt_corr)
    bool stages_valid = false;
};

namespace {

constexpr int kCorrPad = 336;     // 324 correlation features padded to a multiple of 16 channels

struct Cursor {
    const float* p;
    const float* end;
    const float* take(size_t n) { const float* r = p; p += n; return r; }
};

struct Packer {
    std::vector<float> host;
    size_t push(size_t n) { size_t off = host.size(); host.resize(off + ((n + 3) & ~(size_t)3), 0.f); return off; }
};

struct BN { const float *w, *b, *rm, *rv; };

BN take_bn(Cursor& c, int ch) {
    BN bn;
    bn.w = c.take(ch); bn.b = c.take(ch); bn.rm = c.take(ch); bn.rv = c.take(ch);
    return bn;
}

// conv [cout][cin][kh][kw] + bias, optional eval-mode BatchNorm folded into (scale, shift); `co0, con` select an
// output-channel slice (cnet.conv2 is applied as two convs: tanh half and relu half)
void make_layer(Packer& pk, Layer& L, const float* w, const float* bias, int cout_all, int co0, int con, const int* cs, int nseg,
                int kh, int kw, int stride, int ph, int pw, const BN* bn) {
    int cin = 0;
    for (int s = 0; s < nseg; ++s) { cin += cs[s]; L.cs[s] = cs[s]; }
    L.nseg = nseg; L.cout = con; L.kh = kh; L.kw = kw; L.stride = stride; L.ph = ph; L.pw = pw;
    const float* wsl = w + (size_t)co0 * cin * kh * kw;
    L.wpk = pk.push(gconv_packed_floats(con, cs, nseg, kh, kw));
    gconv_pack(wsl, con, cs, nseg, kh, kw, pk.host.data() + L.wpk);
    L.has16 = gconv16_shape(con, cs, nseg, kh, kw, stride) && ph == kh / 2 && pw == kw / 2;
    if (L.has16) {
        L.wpk16 = pk.push(gconv16_packed_floats(con, cs, nseg, kh, kw));
        gconv16_pack(wsl, con, cs, nseg, kh, kw, pk.host.data() + L.wpk16);
    }
    L.hasb = gconvb_shape(con, cs, nseg, kh, kw, stride) && ph == kh / 2 && pw == kw / 2;
    if (L.hasb) {
        L.wpkb = pk.push(gconvb_packed_floats(con, cs, nseg, kh, kw));
        gconvb_pack(wsl, con, cs, nseg, kh, kw, pk.host.data() + L.wpkb);
    }
    L.has_stem = kh == 7 && kw == 7 && stride == 2 && ph == 3 && pw == 3 && nseg == 1 && cin <= 5 && con == 64 && co0 == 0;   // the encoders' stem: gconv.h stem7_*
    if (L.has_stem) {
        L.wstem = pk.push(stem7_packed_floats(cin));
        stem7_pack(wsl, cin, pk.host.data() + L.wstem);
    }
    L.has_few = con <= 8 && kh == 3 && kw == 3 && stride == 1 && nseg == 1 && ph == 1 && pw == 1;   // flow head 256 -> 2: gconv.h fewout_*
    if (L.has_few) {
        L.wfew = pk.push(fewout_packed_floats(cin, kh, kw));
        fewout_pack(wsl, con, cin, kh, kw, pk.host.data() + L.wfew);
    }
    // the encoder's 64 -> 64 residual convs (model/extractor.py:145, layer1) also run on EEMFlow's Winograd F(4x4,3x3) kernel
    // (conv_wino4.hip): raw weights kept for the device-side transform, an eval-mode BatchNorm's scale folded into them
    L.has_f4 = nseg == 1 && cin == 64 && con == 64 && co0 == 0 && kh == 3 && kw == 3 && stride == 1 && ph == 1 && pw == 1;
    if (L.has_f4) {
        L.wraw = pk.push((size_t)64 * 64 * 9);
        for (int co = 0; co < 64; ++co) {
            const float sc = bn ? bn->w[co] / sqrtf(bn->rv[co] + 1e-5f) : 1.f;
            for (int i = 0; i < 64 * 9; ++i) pk.host[L.wraw + (size_t)co * 576 + i] = wsl[(size_t)co * 576 + i] * sc;
        }
    }
    // the encoder's 96 -> 96 and 128 -> 128 residual convs (model/extractor.py:146-147, layer2 / layer3) on the Winograd F(2x2,3x3) kernel
    // of EEMFlow+'s fine levels (conv_wnc.hip: 32-cout slices as the jobs of one launch), an eval-mode BatchNorm's scale folded in
    // (and, round 6, the motion encoder's convc2 256 -> 192, convf2 128 -> 64 and conv 256 -> 126, model/update.py:66-81: any single-tensor
    // 3x3 layer of 32 .. 256 input channels and up to 256 couts qualifies; which of them take the kernel is the call sites' choice)
    L.has_wnc = nseg == 1 && co0 == 0 && kh == 3 && kw == 3 && stride == 1 && ph == 1 && pw == 1 && cin >= 32 && cin <= 32 * WNC_MAX_CHUNKS &&
                con >= 32 && con <= 256 && !(cin == 64 && con == 64);      // (wwnc[8]: up to eight 32-cout slices)
    if (L.has_wnc) {
        std::vector<float> wsc((size_t)con * cin * 9);
        for (int co = 0; co < con; ++co) {
            const float sc = bn ? bn->w[co] / sqrtf(bn->rv[co] + 1e-5f) : 1.f;
            for (int i = 0; i < cin * 9; ++i) wsc[(size_t)co * cin * 9 + i] = wsl[(size_t)co * cin * 9 + i] * sc;
        }
        for (int sl = 0; sl * 32 < con; ++sl) {
            L.wwnc[sl] = pk.push(wnc_packed_floats(cin, 0));
            wnc_pack(wsc.data(), con, cin, sl * 32, 0, pk.host.data() + L.wwnc[sl]);
        }
    }
    L.shift = pk.push(con);
    if (bn) {
        L.has_scale = true;
        L.scale = pk.push(con);
        for (int i = 0; i < con; ++i) {
            const int co = co0 + i;
            const float sc = bn->w[co] / sqrtf(bn->rv[co] + 1e-5f);
            pk.host[L.scale + i] = sc;
            pk.host[L.shift + i] = (bias[co] - bn->rm[co]) * sc + bn->b[co];
        }
    } else {
        for (int i = 0; i < con; ++i) pk.host[L.shift + i] = bias[co0 + i];
    }
    (void)cout_all;
}

void parse_encoder(Cursor& c, Packer& pk, Encoder& E, bool batch_norm, int cin0, int out_dim, bool split_out) {
    E.batch_norm = batch_norm;
    BN bn1{};
    if (batch_norm) bn1 = take_bn(c, 64);                         // norm1 is registered before conv1
    {
        const float* w = c.take((size_t)64 * cin0 * 49);
        const float* b = c.take(64);
        const int cs[1] = {cin0};
        make_layer(pk, E.conv1, w, b, 64, 0, 64, cs, 1, 7, 7, 2, 3, 3, batch_norm ? &bn1 : nullptr);
    }
    const int dims[3] = {64, 96, 128};
    int in_planes = 64, bi = 0;
    for (int l = 0; l < 3; ++l)
        for (int r = 0; r < 2; ++r, ++bi) {
            Block& bk = E.blk[bi];
            const int planes = dims[l];
            const int stride = (r == 0 && l > 0) ? 2 : 1;
            const int cin = r == 0 ? in_planes : planes;
            const float* w1 = c.take((size_t)planes * cin * 9);
            const float* b1 = c.take(planes);
            const float* w2 = c.take((size_t)planes * planes * 9);
            const float* b2 = c.take(planes);
            BN n1{}, n2{}, n3{};
            if (batch_norm) { n1 = take_bn(c, planes); n2 = take_bn(c, planes); }
            bk.has_down = stride != 1;
            const int cs1[1] = {cin}, cs2[1] = {planes};
            make_layer(pk, bk.conv1, w1, b1, planes, 0, planes, cs1, 1, 3, 3, stride, 1, 1, batch_norm ? &n1 : nullptr);
            make_layer(pk, bk.conv2, w2, b2, planes, 0, planes, cs2, 1, 3, 3, 1, 1, 1, batch_norm ? &n2 : nullptr);
            if (bk.has_down) {
                if (batch_norm) n3 = take_bn(c, planes);           // norm3, then downsample.0, then its alias downsample.1
                const float* wd = c.take((size_t)planes * cin);
                const float* bd = c.take(planes);
                if (batch_norm) (void)take_bn(c, planes);
                make_layer(pk, bk.down, wd, bd, planes, 0, planes, cs1, 1, 1, 1, stride, 0, 0, batch_norm ? &n3 : nullptr);
            }
            if (r == 0) in_planes = planes;
        }
    const float* w = c.take((size_t)out_dim * 128);
    const float* b = c.take(out_dim);
    const int cs[1] = {128};
    if (split_out) {
        make_layer(pk, E.conv2a, w, b, out_dim, 0, out_dim / 2, cs, 1, 1, 1, 1, 0, 0, nullptr);
        make_layer(pk, E.conv2b, w, b, out_dim, out_dim / 2, out_dim / 2, cs, 1, 1, 1, 1, 0, 0, nullptr);
    } else {
        make_layer(pk, E.conv2a, w, b, out_dim, 0, out_dim, cs, 1, 1, 1, 1, 0, 0, nullptr);
    }
}

GConvArgs conv_args(const eraft_ctx* c, const Layer& L, int n, int hin, int win, float* out, int out_ctotal, int out_coff, int act) {
    GConvArgs a;
    memset(&a, 0, sizeof(a));
    a.nseg = L.nseg;
    a.wpk = c->arena + L.wpk;
    a.wpk16 = L.has16 ? c->arena + L.wpk16 : nullptr;
    a.wpkb = L.hasb ? c->arena + L.wpkb : nullptr;
    a.wfew = L.has_few ? c->arena + L.wfew : nullptr;
    a.wstem = L.has_stem ? c->arena + L.wstem : nullptr;
    a.zero_page = c->arena + c->zero_off;
    a.scale = L.has_scale ? c->arena + L.scale : nullptr;
    a.shift = c->arena + L.shift;
    a.out = out; a.out_ctotal = out_ctotal; a.out_coff = out_coff;
    a.n = n; a.hin = hin; a.win = win;
    a.hout = (hin + 2 * L.ph - L.kh) / L.stride + 1;
    a.wout = (win + 2 * L.pw - L.kw) / L.stride + 1;
    a.cout = L.cout; a.kh = L.kh; a.kw = L.kw; a.stride = L.stride; a.pad_h = L.ph; a.pad_w = L.pw;
    a.act = act; a.epi = GEPI_PLAIN; a.out_scale = 1.f;
    a.in_flight = c->frames_in_flight;
    return a;
}

void set_seg(GConvArgs& a, int i, const float* ptr, int cch, int ctotal, int coff) {
    a.seg[i].ptr = ptr; a.seg[i].c = cch; a.seg[i].ctotal = ctotal; a.seg[i].coff = coff;
}

// A 64 -> 64 3x3 conv of the encoder on the Winograd F(4x4,3x3) kernel: out = act(conv(x) + shift) [then relu(res + .)];
// act: 0 none (an InstanceNorm follows), 2 ReLU (eval BatchNorm folded into weights and shift).  From 128 tiles of 32x16 pixels on
// (fewer leave most CUs without a block and the LDS-tiled kernel wins); EEM_ERAFT_NO_F4=1 keeps every conv on gconv.
bool f4_eligible(const Layer& L, int n, int h, int w) {
    const char* e = getenv("EEM_ERAFT_NO_F4");
    if (e && e[0] == '1') return false;
    return L.has_f4 && w % 4 == 0 && (long)n * ((h + 15) / 16) * ((w + 31) / 32) >= 128;
}
int run_f4(eraft_ctx* c, const Layer& L, const float* x, int n, int h, int w, float* out, int act, const float* res, hipStream_t st) {
    EncConvArgs a;
    memset((void*)&a, 0, sizeof(a));
    a.in0 = x; a.wwino = c->wino + L.wf4; a.wino_f4 = 1;
    a.zero_page = c->arena + c->zero_off; a.trash = c->trash;
    a.bias = c->arena + L.shift; a.out = out;
    a.nimg = n; a.nimg0 = n; a.hin = h; a.win = w; a.hout = h; a.wout = w; a.hraw = h; a.wraw = w;
    a.act = act; a.res = res;
    {   // the interleaved tile walk (conv_wino4.hip, round 6): neighbouring tiles in flight together; EEM_WALK3_ERAFT=0 keeps contiguous ranges
        static const bool w3 = [] { const char* e = getenv("EEM_WALK3_ERAFT"); return !(e && e[0] == '0'); }();
        if (w3) a.reverse = 3;
    }
    return wino4_launch(64, a, st);
}

// A 96 -> 96 / 128 -> 128 3x3 conv of the encoder on the Winograd F(2x2,3x3) kernel (same contract as run_f4).  From 128 (tile, slice)
// pairs on; EEM_ERAFT_NO_WNC=1 (read per call) keeps every conv on gconv.
bool wnc_eligible(const Layer& L, int n, int h, int w) {
    const char* e = getenv("EEM_ERAFT_NO_WNC");
    if (e && e[0] == '1') return false;
    return L.has_wnc && w % 4 == 0 && (long)n * ((h + 3) / 4) * ((w + 31) / 32) * (L.cout / 32) >= 128;
}
int run_wnc(eraft_ctx* c, const Layer& L, const float* x, int n, int h, int w, float* out, int act, const float* res, hipStream_t st,
            int out_ctotal = 0, int out_coff = 0) {
    WncArgs a;
    memset(&a, 0, sizeof(a));
    a.nchunks = wnc_chunks(L.cs[0], a.chunk_off);
    a.cin = L.cs[0];
    a.n = n; a.h = h; a.w = w; a.act = act;
    a.zero_page = c->arena + c->zero_off; a.trash = c->trash;
    for (int sl = 0; sl * 32 < L.cout; ++sl) {
        WncJob& J = a.job[a.njobs++];
        J.in = x; J.in_ctotal = L.cs[0]; J.in_coff = 0;
        J.w = c->arena + L.wwnc[sl]; J.bias = c->arena + L.shift + sl * 32;
        J.out = out; J.out_ctotal = out_ctotal ? out_ctotal : L.cout; J.out_coff = out_coff + sl * 32; J.out_cmul = 1;
        J.cout = L.cout - sl * 32 < 32 ? L.cout - sl * 32 : 32;
        J.res = res;
    }
    EEM_REQUIRE(wnc_supported(a), "run_wnc: the launch does not qualify (alignment)");
    return wnc_launch(a, st);
}

// the motion encoder's three 3x3 convs (model/update.py:66-81) on the same kernel: EEM_ERAFT_WNC_UPD=<mask> (read per call; bit 0 convc2
// 256 -> 192, bit 1 convf2 128 -> 64, bit 2 conv 256 -> 126).  Measured at 640x480 x 12, batch 4, one box: none 289.5 frames/s, convf2 298.7
// (it ran the LDS-tiled kernel's two-K-group form - 240 blocks - at 25 TFLOP/s beside convc2: 115 us), convc2 287, conv 296, all three 297;
// batch 1 within 0.5 %: the default is convf2 alone
bool wnc_upd(const Layer& L, int bit, int n, int h, int w) {
    const char* e = getenv("EEM_ERAFT_WNC_UPD");
    const int mask = e ? atoi(e) : 2;
    return ((mask >> bit) & 1) && wnc_eligible(L, n, h, w);
}

// BasicEncoder forward on `n` images [n][cin0][hp][wp]; result of the residual stack in *feat ([n][128][hp/8][wp/8]).
int run_encoder(eraft_ctx* c, const Encoder& E, const float* x, int n, int cin0, int hp, int wp, float** feat, hipStream_t st, Buf* sc = nullptr) {
    int rc;
    if (!sc) sc = c->s;
    float* X = sc[0].p; float* R = sc[1].p; float* Y = sc[2].p; float* D = sc[3].p; float* O = sc[4].p;
    int h = hp, w = wp;
    {   // conv1 7x7 s2 + norm1 + relu
        GConvArgs a = conv_args(c, E.conv1, n, h, w, E.batch_norm ? X : R, 64, 0, E.batch_norm ? GACT_RELU : GACT_NONE);
        set_seg(a, 0, x, cin0, cin0, 0);
        if ((rc = gconv_launch(a, st)) != EEM_OK) return rc;
        h = a.hout; w = a.wout;
        if (!E.batch_norm && (rc = er_instnorm_launch(R, X, nullptr, n * 64, h * w, 1, st, c->nstat, c->nstat_cap)) != EEM_OK) return rc;
    }
    const int dims[3] = {64, 96, 128};
    int in_planes = 64, bi = 0;
    for (int l = 0; l < 3; ++l)
        for (int r = 0; r < 2; ++r, ++bi) {
            const Block& bk = E.blk[bi];
            const int planes = dims[l], cin = r == 0 ? in_planes : planes;
            // y = relu(norm1(conv1(x)))
            GConvArgs a1 = conv_args(c, bk.conv1, n, h, w, E.batch_norm ? Y : R, planes, 0, E.batch_norm ? GACT_RELU : GACT_NONE);
            set_seg(a1, 0, X, cin, cin, 0);
            if (f4_eligible(bk.conv1, n, h, w)) rc = run_f4(c, bk.conv1, X, n, h, w, E.batch_norm ? Y : R, E.batch_norm ? 2 : 0, nullptr, st);
            else if (wnc_eligible(bk.conv1, n, h, w)) rc = run_wnc(c, bk.conv1, X, n, h, w, E.batch_norm ? Y : R, E.batch_norm ? 2 : 0, nullptr, st);
            else rc = gconv_launch(a1, st);
            if (rc != EEM_OK) return rc;
            const int ho = a1.hout, wo = a1.wout;
            if (!E.batch_norm && (rc = er_instnorm_launch(R, Y, nullptr, n * planes, ho * wo, 1, st, c->nstat, c->nstat_cap)) != EEM_OK) return rc;
            // shortcut
            const float* res = X;
            if (bk.has_down) {
                GConvArgs ad = conv_args(c, bk.down, n, h, w, E.batch_norm ? D : R, planes, 0, GACT_NONE);
                set_seg(ad, 0, X, cin, cin, 0);
                if ((rc = gconv_launch(ad, st)) != EEM_OK) return rc;
                if (!E.batch_norm && (rc = er_instnorm_launch(R, D, nullptr, n * planes, ho * wo, 0, st, c->nstat, c->nstat_cap)) != EEM_OK) return rc;
                res = D;
            }
            // out = relu(res + relu(norm2(conv2(y))))
            GConvArgs a2 = conv_args(c, bk.conv2, n, ho, wo, E.batch_norm ? O : R, planes, 0, E.batch_norm ? GACT_RELU : GACT_NONE);
            set_seg(a2, 0, Y, planes, planes, 0);
            if (E.batch_norm) { a2.epi = GEPI_ADD_RELU; a2.e0 = res; a2.e0_ctotal = planes; a2.e0_coff = 0; }
            if (f4_eligible(bk.conv2, n, ho, wo)) rc = run_f4(c, bk.conv2, Y, n, ho, wo, E.batch_norm ? O : R, E.batch_norm ? 2 : 0, E.batch_norm ? res : nullptr, st);
            else if (wnc_eligible(bk.conv2, n, ho, wo)) rc = run_wnc(c, bk.conv2, Y, n, ho, wo, E.batch_norm ? O : R, E.batch_norm ? 2 : 0, E.batch_norm ? res : nullptr, st);
            else rc = gconv_launch(a2, st);
            if (rc != EEM_OK) return rc;
            if (!E.batch_norm && (rc = er_instnorm_launch(R, O, res, n * planes, ho * wo, 1, st, c->nstat, c->nstat_cap)) != EEM_OK) return rc;
            float* t = X; X = O; O = t;
            h = ho; w = wo;
            if (r == 0) in_planes = planes;
        }
    *feat = X;
    return EEM_OK;
}

int build_pyramid(eraft_ctx* c, const float* f1, const float* f2, int batch, int ch, int h, int w, hipStream_t st) {
    int rc;
    const size_t hw = (size_t)h * w;
    c->ph[0] = h; c->pw[0] = w;
    for (int l = 1; l < 4; ++l) { c->ph[l] = c->ph[l - 1] / 2; c->pw[l] = c->pw[l - 1] / 2; }
    for (int l = 0; l < 4; ++l) {
        EEM_REQUIRE(c->ph[l] >= 1 && c->pw[l] >= 1, "correlation pyramid level %d is empty for a %dx%d feature map", l, h, w);
        if ((rc = ensure(c->pyr[l], (size_t)batch * hw * c->ph[l] * c->pw[l])) != EEM_OK) return rc;
    }
    if ((rc = er_allpairs_launch(f1, f2, c->pyr[0].p, batch, ch, (int)hw, st)) != EEM_OK) return rc;
    return er_pool2x3_launch(c->pyr[0].p, c->pyr[1].p, c->pyr[2].p, c->pyr[3].p, (long)batch * hw, h, w, st);
}

// alt_corr: the level sizes and the avg_pool2d(2) chain of fmap2 itself (pooling the volume's last two dimensions = pooling fmap2)
int build_feature_pyramid(eraft_ctx* c, const float* f2, int batch, int ch, int h, int w, hipStream_t st) {
    int rc;
    c->ph[0] = h; c->pw[0] = w;
    for (int l = 1; l < 4; ++l) { c->ph[l] = c->ph[l - 1] / 2; c->pw[l] = c->pw[l - 1] / 2; }
    const float* src = f2;
    for (int l = 1; l < 4; ++l) {
        EEM_REQUIRE(c->ph[l] >= 1 && c->pw[l] >= 1, "correlation pyramid level %d is empty for a %dx%d feature map", l, h, w);
        if ((rc = ensure(c->f2l[l - 1], (size_t)batch * ch * c->ph[l] * c->pw[l])) != EEM_OK) return rc;
        if ((rc = er_pool2_launch(src, c->f2l[l - 1].p, (long)batch * ch, c->ph[l - 1], c->pw[l - 1], st)) != EEM_OK) return rc;
        src = c->f2l[l - 1].p;
    }
    return EEM_OK;
}

// flow_dst (optional): the lookup launch also writes flow = coords - flow_c0 there (resident-volume form; the on-the-fly form
// keeps the separate launch)
int run_lookup(eraft_ctx* c, const float* coords, float* out, int out_ctotal, int batch, int h, int w, hipStream_t st,
               const float* flow_c0 = nullptr, float* flow_dst = nullptr, int flow_ctotal = 0, int flow_coff = 0, hipEvent_t done_ev = nullptr) {
    if (c->alt_corr) {
        if (flow_dst) {
            const int rcf = er_flow_launch(flow_c0, coords, flow_dst, flow_ctotal, flow_coff, batch, h * w, st);
            if (rcf != EEM_OK) return rcf;
        }
        AltCorrArgs aa;
        const size_t g = (size_t)h * w;
        aa.f1 = c->fmap.p;
        aa.f2[0] = c->fmap.p + (size_t)batch * 256 * g;
        for (int l = 1; l < 4; ++l) aa.f2[l] = c->f2l[l - 1].p;
        for (int l = 0; l < 4; ++l) { aa.ph[l] = c->ph[l]; aa.pw[l] = c->pw[l]; }
        aa.coords = coords; aa.out = out; aa.batch = batch; aa.c = 256; aa.h = h; aa.w = w; aa.out_ctotal = out_ctotal;
        aa.scale = 1.0f / 16.0f;                                 // 1 / sqrt(256)
        const int rca = er_altcorr_launch(aa, st);
        if (rca == EEM_OK && done_ev) EEM_HIP_CHECK(hipEventRecord(done_ev, st));
        return rca;
    }
    LookupArgs la;
    for (int l = 0; l < 4; ++l) { la.pyr[l] = c->pyr[l].p; la.ph[l] = c->ph[l]; la.pw[l] = c->pw[l]; }
    la.coords = coords; la.out = out; la.batch = batch; la.h = h; la.w = w; la.out_ctotal = out_ctotal;
    la.coords0 = flow_c0; la.flow_dst = flow_dst; la.flow_ctotal = flow_ctotal; la.flow_coff = flow_coff;
    return er_lookup_launch(la, st, done_ev);
}

}  // namespace

extern "C" int eraft_create(int device, eraft_ctx** out) {
    EEM_REQUIRE(out != nullptr, "eraft_create: out is NULL");
    int ndev = 0;
    EEM_HIP_CHECK(hipGetDeviceCount(&ndev));
    EEM_REQUIRE(device >= 0 && device < ndev, "eraft_create: device %d of %d", device, ndev);
    EEM_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    EEM_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    EEM_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, "built for gfx950 (MI355X) only; device %d is %s", device,
                prop.gcnArchName);
    eraft_ctx* c = new eraft_ctx();
    c->device = device;
    *out = c;
    return EEM_OK;
}

extern "C" void eraft_destroy(eraft_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    Buf* all[] = {&c->padded, &c->s[0], &c->s[1], &c->s[2], &c->s[3], &c->s[4], &c->fmap, &c->net[0], &c->net[1], &c->net[2], &c->inp,
                  &c->pyr[0], &c->pyr[1], &c->pyr[2], &c->pyr[3], &c->c0, &c->c1, &c->c1b, &c->corr, &c->cor1, &c->corflo, &c->flo1,
                  &c->motion, &c->z, &c->rh, &c->fhid, &c->delta, &c->mhid, &c->mask, &c->st_corr0, &c->st_net1, &c->st_mask1,
                  &c->st_delta1, &c->zeros, &c->many_out, &c->f2l[0], &c->f2l[1], &c->f2l[2], &c->czr[0], &c->czr[1], &c->cq[0], &c->cq[1],
                  &c->s2[0], &c->s2[1], &c->s2[2], &c->s2[3], &c->s2[4]};
    for (Buf* b : all) if (b->p) (void)hipFree(b->p);
    if (c->nstat) (void)hipFree(c->nstat);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
    if (c->join_ev) (void)hipEventDestroy(c->join_ev);
    if (c->arena) (void)hipFree(c->arena);
    if (c->wino) (void)hipFree(c->wino);
    if (c->trash) (void)hipFree(c->trash);
    delete c;
}

extern "C" int eraft_load_weights(eraft_ctx* c, const float* flat, size_t nfloats, int n_first_channels) {
    EEM_REQUIRE(c && flat, "eraft_load_weights: NULL argument");
    EEM_REQUIRE(n_first_channels >= 1 && n_first_channels <= 64, "eraft_load_weights: n_first_channels=%d", n_first_channels);
    EEM_HIP_CHECK(hipSetDevice(c->device));
    Cursor cur{flat, flat + nfloats};
    Packer pk;
    c->zero_off = pk.push(64);
    parse_encoder(cur, pk, c->fnet, false, n_first_channels, 256, false);
    parse_encoder(cur, pk, c->cnet, true, n_first_channels, 256, true);
    const float* last_w = nullptr;
    const float* last_b = nullptr;
    auto plain = [&](Layer& L, int cout, const int* cs, int nseg, int kh, int kw, int ph, int pw) {
        int cin = 0;
        for (int s = 0; s < nseg; ++s) cin += cs[s];
        const float* w = cur.take((size_t)cout * cin * kh * kw);
        const float* b = cur.take(cout);
        last_w = w; last_b = b;
        make_layer(pk, L, w, b, cout, 0, cout, cs, nseg, kh, kw, 1, ph, pw, nullptr);
    };
    // two layers that read the same input, stacked along the output channels
    auto stacked = [&](Layer& L, const float* w0, const float* b0, const float* w1, const float* b1, int cout_each, const int* cs, int nseg,
                       int kh, int kw, int ph, int pw) {
        int cin = 0;
        for (int s = 0; s < nseg; ++s) cin += cs[s];
        const size_t nw = (size_t)cout_each * cin * kh * kw;
        std::vector<float> w(2 * nw), b(2 * (size_t)cout_each);
        memcpy(w.data(), w0, nw * sizeof(float)); memcpy(w.data() + nw, w1, nw * sizeof(float));
        memcpy(b.data(), b0, cout_each * sizeof(float)); memcpy(b.data() + cout_each, b1, cout_each * sizeof(float));
        make_layer(pk, L, w.data(), b.data(), 2 * cout_each, 0, 2 * cout_each, cs, nseg, kh, kw, 1, ph, pw, nullptr);
    };
    const int c324[1] = {324}, c256[1] = {256}, c2[1] = {2}, c128[1] = {128}, c3x128[3] = {128, 128, 128};
    plain(c->convc1, 256, c324, 1, 1, 1, 0, 0);          // model/update.py:63-71
    {   // second packing of convc1 for the 336-channel correlation buffer: zero columns for the 12 pad channels
        const float* w = cur.p - ((size_t)256 * 324 + 256);
        std::vector<float> wp((size_t)256 * kCorrPad, 0.f);
        for (int co = 0; co < 256; ++co) memcpy(&wp[(size_t)co * kCorrPad], w + (size_t)co * 324, 324 * sizeof(float));
        const int cpad[1] = {kCorrPad};
        c->convc1.has16 = gconv16_shape(256, cpad, 1, 1, 1, 1);
        c->convc1.wpk16 = pk.push(gconv16_packed_floats(256, cpad, 1, 1, 1));
        gconv16_pack(wp.data(), 256, cpad, 1, 1, 1, pk.host.data() + c->convc1.wpk16);
    }
    plain(c->convc2, 192, c256, 1, 3, 3, 1, 1);
    plain(c->convf1, 128, c2, 1, 7, 7, 3, 3);
    plain(c->convf2, 64, c128, 1, 3, 3, 1, 1);
    plain(c->conv, 126, c256, 1, 3, 3, 1, 1);
    // a conv over [h | inp | motion] split by input channels: `hm` over [h | motion] without bias, `ctx` over inp with the bias
    auto split_ctx = [&](Layer& hm, Layer& ctx, const float* w, const float* b, int cout, int kh, int kw, int ph, int pw) {
        const int taps = kh * kw;
        std::vector<float> whm((size_t)cout * 256 * taps), wc((size_t)cout * 128 * taps), zero(cout, 0.f);
        for (int co = 0; co < cout; ++co) {
            const float* src = w + (size_t)co * 384 * taps;
            memcpy(&whm[(size_t)co * 256 * taps], src, (size_t)128 * taps * sizeof(float));
            memcpy(&whm[((size_t)co * 256 + 128) * taps], src + (size_t)256 * taps, (size_t)128 * taps * sizeof(float));
            memcpy(&wc[(size_t)co * 128 * taps], src + (size_t)128 * taps, (size_t)128 * taps * sizeof(float));
        }
        const int c2x128[2] = {128, 128};
        make_layer(pk, hm, whm.data(), zero.data(), cout, 0, cout, c2x128, 2, kh, kw, 1, ph, pw, nullptr);
        make_layer(pk, ctx, wc.data(), b, cout, 0, cout, c128, 1, kh, kw, 1, ph, pw, nullptr);
    };
    for (int pass = 0; pass < 2; ++pass) {               // model/update.py:33-44: hx = [h | inp | motion]; 1x5 pass, then 5x1
        const int kh = pass ? 5 : 1, kw = pass ? 1 : 5, ph = pass ? 2 : 0, pw = pass ? 0 : 2;
        plain(c->gz[pass], 128, c3x128, 3, kh, kw, ph, pw);
        const float* wz = last_w; const float* bz = last_b;
        plain(c->gr[pass], 128, c3x128, 3, kh, kw, ph, pw);
        stacked(c->gzr[pass], wz, bz, last_w, last_b, 128, c3x128, 3, kh, kw, ph, pw);
        {
            const size_t nw = (size_t)128 * 384 * kh * kw;
            std::vector<float> w2(2 * nw), b2(256);
            memcpy(w2.data(), wz, nw * sizeof(float)); memcpy(w2.data() + nw, last_w, nw * sizeof(float));
            memcpy(b2.data(), bz, 128 * sizeof(float)); memcpy(b2.data() + 128, last_b, 128 * sizeof(float));
            split_ctx(c->gzr_hm[pass], c->gzr_c[pass], w2.data(), b2.data(), 256, kh, kw, ph, pw);
        }
        plain(c->gq[pass], 128, c3x128, 3, kh, kw, ph, pw);
        split_ctx(c->gq_hm[pass], c->gq_c[pass], last_w, last_b, 128, kh, kw, ph, pw);
    }
    plain(c->fh1, 256, c128, 1, 3, 3, 1, 1);             // model/update.py:6-14
    const float* wf = last_w; const float* bf = last_b;
    plain(c->fh2, 2, c256, 1, 3, 3, 1, 1);
    plain(c->mk0, 256, c128, 1, 3, 3, 1, 1);             // model/update.py:92-95
    stacked(c->heads1, wf, bf, last_w, last_b, 256, c128, 1, 3, 3, 1, 1);
    plain(c->mk2, 576, c256, 1, 1, 1, 0, 0);
    EEM_REQUIRE(cur.p == cur.end, "eraft_load_weights: the 179-tensor layout needs %zu floats (num_batches_tracked "
                                  "excluded), got %zu", (size_t)(cur.p - flat), nfloats);
    if (c->arena) EEM_HIP_CHECK(hipFree(c->arena));
    c->arena = nullptr;
    EEM_HIP_CHECK(hipMalloc(&c->arena, pk.host.size() * sizeof(float)));
    EEM_HIP_CHECK(hipMemcpy(c->arena, pk.host.data(), pk.host.size() * sizeof(float), hipMemcpyHostToDevice));
    {   // Winograd-domain weights of the 64 -> 64 encoder convs
        std::vector<Layer*> f4;
        for (Encoder* E : {&c->fnet, &c->cnet})
            for (Block& bk : E->blk)
                for (Layer* L : {&bk.conv1, &bk.conv2})
                    if (L->has_f4) f4.push_back(L);
        if (c->wino) { EEM_HIP_CHECK(hipFree(c->wino)); c->wino = nullptr; }
        if (!f4.empty()) {
            const size_t each = wino4_packed_floats(64);
            EEM_HIP_CHECK(hipMalloc(&c->wino, f4.size() * each * sizeof(float)));
            for (size_t i = 0; i < f4.size(); ++i) {
                f4[i]->wf4 = i * each;
                const int rcw = wino4_transform_launch(c->arena + f4[i]->wraw, 64, 0, c->wino + f4[i]->wf4, nullptr);
                if (rcw != EEM_OK) return rcw;
            }
            EEM_HIP_CHECK(hipDeviceSynchronize());
        }
        if (!c->trash) EEM_HIP_CHECK(hipMalloc(&c->trash, 4096));
    }
    c->cin0 = n_first_channels;
    c->loaded = true;
    return EEM_OK;
}

// frames == 0: e1 / e2 / out are eraft_forward's batched tensors.  frames = n > 0 (eraft_forward_many): e1v / e2v hold n pointers to single
// samples, read by the pad launches; `out` is the context's staging buffer for the batched predictions.
static int eraft_forward_impl(eraft_ctx* c, const float* e1, const float* e2, const float* const* e1v, const float* const* e2v, int frames,
                              int batch, int in_h, int in_w, const int pad[4], int iters, const float* flow_init, float* out, hipStream_t st) {
    const int B = batch, hp = in_h + pad[2] + pad[3], wp = in_w + pad[0] + pad[1];
    EEM_REQUIRE(hp % 8 == 0 && wp % 8 == 0, "eraft_forward: padded size %dx%d must be a multiple of 8", hp, wp);
    const int h8 = hp / 8, w8 = wp / 8;
    const size_t g = (size_t)h8 * w8;
    int rc;
    // ---- workspace
    const size_t big = (size_t)2 * B * 64 * ((hp + 1) / 2) * ((wp + 1) / 2);
#define ENS(b, n) if ((rc = ensure(b, n)) != EEM_OK) return rc
    ENS(c->padded, (size_t)2 * B * c->cin0 * hp * wp);
    for (int i = 0; i < 5; ++i) ENS(c->s[i], big);
    {   // scratch of the large-plane instance norm: the first stage has the most (planes x chunks)
        const int hw1 = ((hp + 1) / 2) * ((wp + 1) / 2);
        size_t need = er_instnorm_scratch_doubles(2 * B * 64, hw1);
        need = std::max(need, er_instnorm_scratch_doubles(2 * B * 96, hw1 / 4));
        need = std::max(need, er_instnorm_scratch_doubles(2 * B * 128, hw1 / 16));
        if (need > c->nstat_cap) {
            if (c->nstat) EEM_HIP_CHECK(hipFree(c->nstat));
            c->nstat = nullptr; c->nstat_cap = 0;
            EEM_HIP_CHECK(hipMalloc(&c->nstat, need * sizeof(double)));
            c->nstat_cap = need;
        }
    }
    ENS(c->fmap, (size_t)2 * B * 256 * g);
    ENS(c->net[0], B * 128 * g); ENS(c->net[1], B * 128 * g); ENS(c->net[2], B * 128 * g); ENS(c->inp, B * 128 * g);
    ENS(c->c0, B * 2 * g); ENS(c->c1, B * 2 * g); ENS(c->c1b, B * 2 * g); ENS(c->corr, B * kCorrPad * g); ENS(c->cor1, B * 256 * g);
    ENS(c->corflo, B * 256 * g); ENS(c->flo1, B * 128 * g); ENS(c->motion, B * 128 * g); ENS(c->z, B * 256 * g);
    ENS(c->rh, B * 128 * g); ENS(c->fhid, B * 512 * g); ENS(c->delta, B * 2 * g); ENS(c->mhid, B * 256 * g);
    ENS(c->mask, B * 576 * g);
    if (c->keep_stages) { ENS(c->st_corr0, B * 324 * g); ENS(c->st_net1, B * 128 * g); ENS(c->st_mask1, B * 576 * g); ENS(c->st_delta1, B * 2 * g); }
#undef ENS
    // Independent branches on the context's side stream (EEM_ERAFT_NO_OVERLAP=1, read per forward: everything on the caller's stream).
    // One forward at a time most launches of this model are 40 - 300 blocks for 256 CUs: the branches fill CUs the main chain leaves idle.
    const char* eno = getenv("EEM_ERAFT_NO_OVERLAP");
    const bool overlap = !(eno && eno[0] == '1');
    if (overlap) {
        if (!c->side) {
            EEM_HIP_CHECK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
            EEM_HIP_CHECK(hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming));
            EEM_HIP_CHECK(hipEventCreateWithFlags(&c->join_ev, hipEventDisableTiming));
        }
        for (int i = 0; i < 5; ++i)
            if ((rc = ensure(c->s2[i], big / 2)) != EEM_OK) return rc;
    }
    // The prediction of an iteration - mask head (3x3 128 -> 256, 1x1 256 -> 576) and convex upsampling - feeds nothing inside the loop
    // (model/eraft.py:141-157: the recurrence is net -> flow head -> coords1 -> lookup): it runs on the side stream beside the NEXT
    // iteration, which starts as soon as the flow head has updated coords1.  One forward at a time the update block's launches leave
    // most CUs idle (60x80 cells at batch 1 are 75 - 300 blocks): E-RAFT 640x480 x 12 batch 1 / 4: tools/bench_eraft.py, DESIGN.md 4b.
    // EEM_ERAFT_NO_LAG=1 (read per forward): the mask head inside the iteration, as before.
    const char* enl = getenv("EEM_ERAFT_NO_LAG");
    const char* enf0 = getenv("EEM_ERAFT_NO_FUSE");
    const bool lagged = overlap && !(enl && enl[0] == '1') && !(enf0 && enf0[0] == '1');
    hipStream_t sd = overlap ? c->side : st;
    // fork: the side stream continues from here on the caller's stream; join: the caller's stream waits for the side stream's work so far
    // An error return between a fork and its join must not leave work queued on the side stream that the caller's stream never waits
    // for (the caller may free or reuse its buffers, or re-enter the context): the guard joins on every way out of this function
    struct ForkGuard {
        hipStream_t st, sd; hipEvent_t join_ev; bool forked = false;
        ~ForkGuard() {
            if (!forked) return;
            if (hipEventRecord(join_ev, sd) != hipSuccess || hipStreamWaitEvent(st, join_ev, 0) != hipSuccess) (void)hipStreamSynchronize(sd);
        }
    } guard{st, sd, c->join_ev};
    auto fork = [&]() -> int {
        if (!overlap) return EEM_OK;
        EEM_HIP_CHECK(hipEventRecord(c->fork_ev, st));
        EEM_HIP_CHECK(hipStreamWaitEvent(sd, c->fork_ev, 0));
        guard.forked = true;
        return EEM_OK;
    };
    auto join = [&]() -> int {
        if (!overlap) return EEM_OK;
        EEM_HIP_CHECK(hipEventRecord(c->join_ev, sd));
        EEM_HIP_CHECK(hipStreamWaitEvent(st, c->join_ev, 0));
        guard.forked = false;
        return EEM_OK;
    };
    // (workspace of the forked region, before the fork: ensure() may free / allocate and synchronise the device)
    for (int pass = 0; pass < 2; ++pass)
        if ((rc = ensure(c->czr[pass], (size_t)B * 256 * g)) != EEM_OK || (rc = ensure(c->cq[pass], (size_t)B * 128 * g)) != EEM_OK) return rc;
    // pad channels of the correlation buffer (never written by the lookup)
    EEM_HIP_CHECK(hipMemset2DAsync(c->corr.p + (size_t)324 * g, (size_t)kCorrPad * g * 4, 0, (size_t)(kCorrPad - 324) * g * 4, B, st));
    const int cin0 = c->cin0;
    // ---- pad both event volumes into one batch (model/eraft.py:106-109)
    float* pad1 = c->padded.p;
    float* pad2 = c->padded.p + (size_t)B * cin0 * hp * wp;
    if (frames == 0) {
        if ((rc = er_pad2_launch(e1, e2, pad1, B * cin0, in_h, in_w, pad[0], pad[1], pad[2], pad[3], st)) != EEM_OK) return rc;
    } else {
        const size_t img = (size_t)cin0 * hp * wp;
        for (int i = 0; i < frames; ++i) {
            if ((rc = er_pad_launch(e1v[i], pad1 + (size_t)i * img, cin0, in_h, in_w, pad[0], pad[1], pad[2], pad[3], st)) != EEM_OK) return rc;
            if ((rc = er_pad_launch(e2v[i], pad2 + (size_t)i * img, cin0, in_h, in_w, pad[0], pad[1], pad[2], pad[3], st)) != EEM_OK) return rc;
        }
    }
    float* cfeat = nullptr;
    if ((rc = fork()) != EEM_OK) return rc;                            // (the padded volumes are on their way)
    // (the feature network's launches are enqueued FIRST: it is the longer chain - two images, instance norms - and the host needs ~0.3 ms
    // to enqueue either network's ~45 launches; a forward that starts on an idle GPU - the first of a timed region, every forward of a loop
    // that reads each result - had the chip run the context network alone for that long before the feature network's first kernel arrived)
    const char* enp = getenv("EEM_ERAFT_NO_PRE");
    const char* ens0 = getenv("EEM_ERAFT_NO_STACK");
    const bool use_pre = !(enp && enp[0] == '1') && !(ens0 && ens0[0] == '1');
    auto run_fnet = [&]() -> int {
        // ---- feature network on [image1; image2] (:116), then its 1x1 output conv
        float* feat = nullptr;
        if ((rc = run_encoder(c, c->fnet, c->padded.p, 2 * B, cin0, hp, wp, &feat, st)) != EEM_OK) return rc;
        {
            GConvArgs a = conv_args(c, c->fnet.conv2a, 2 * B, h8, w8, c->fmap.p, 256, 0, GACT_NONE);
            set_seg(a, 0, feat, 128, 128, 0);
            if ((rc = gconv_launch(a, st)) != EEM_OK) return rc;
        }
        // ---- all-pairs correlation pyramid (:121)
        if (c->alt_corr) rc = build_feature_pyramid(c, c->fmap.p + (size_t)B * 256 * g, B, 256, h8, w8, st);
        else rc = build_pyramid(c, c->fmap.p, c->fmap.p + (size_t)B * 256 * g, B, 256, h8, w8, st);
        if (rc != EEM_OK) return rc;
        return EEM_OK;
    };
    auto run_cnet = [&]() -> int {
        // ---- context network on image1 (:126-131): net = tanh(first half), inp = relu(second half)
        if ((rc = run_encoder(c, c->cnet, pad1, B, cin0, hp, wp, &cfeat, sd, overlap ? c->s2 : nullptr)) != EEM_OK) return rc;
        {
            GConvArgs a = conv_args(c, c->cnet.conv2a, B, h8, w8, c->net[0].p, 128, 0, GACT_TANH);
            set_seg(a, 0, cfeat, 128, 128, 0);
            if ((rc = gconv_launch(a, sd)) != EEM_OK) return rc;
            GConvArgs b2 = conv_args(c, c->cnet.conv2b, B, h8, w8, c->inp.p, 128, 0, GACT_RELU);
            set_seg(b2, 0, cfeat, 128, 128, 0);
            if ((rc = gconv_launch(b2, sd)) != EEM_OK) return rc;
        }
        if ((rc = er_coords_init_launch(c->c0.p, c->c1.p, flow_init, B, h8, w8, sd)) != EEM_OK) return rc;
        // the context features' part of the GRU convs, once per forward (see eraft_ctx)
        if (use_pre) {
            for (int pass = 0; pass < 2; ++pass) {
                GConvArgs a = conv_args(c, c->gzr_c[pass], B, h8, w8, c->czr[pass].p, 256, 0, GACT_NONE);
                set_seg(a, 0, c->inp.p, 128, 128, 0);
                if ((rc = gconv_launch(a, sd)) != EEM_OK) return rc;
                a = conv_args(c, c->gq_c[pass], B, h8, w8, c->cq[pass].p, 128, 0, GACT_NONE);
                set_seg(a, 0, c->inp.p, 128, 128, 0);
                if ((rc = gconv_launch(a, sd)) != EEM_OK) return rc;
            }
        }
        return EEM_OK;
    };
    // EEM_ERAFT_CNET_FIRST=1 (read per forward): the order through round 6's first half
    const char* ecf = getenv("EEM_ERAFT_CNET_FIRST");
    if (ecf && ecf[0] == '1') { if ((rc = run_cnet()) != EEM_OK || (rc = run_fnet()) != EEM_OK) return rc; }
    else if ((rc = run_fnet()) != EEM_OK || (rc = run_cnet()) != EEM_OK) return rc;
    if ((rc = join()) != EEM_OK) return rc;                            // net, inp, the context parts of the GRU convs, coords
    // final_only: the predictions of iterations 0 .. iters - 2 are never formed - their mask head (the 3x3 128 -> 256 and 1x1 256 -> 576
    // convs) and convex upsampling are not launched; the hidden state and coords1 go through the same launches with the same operands,
    // so the one prediction that leaves is bit for bit the last of the full list (tests/test_eraft_hip.py)
    float* c1p = c->c1.p;
    float* c1q = c->c1b.p;
    // lagged schedule: the prediction of iteration i - mask head and convex upsampling - goes onto the SIDE stream behind the flow branch of
    // iteration i + 1, after that iteration's join event is recorded there: no stream and no event of its own (an event recorded on the
    // chain's stream costs it ~7 us, tools/eraft_timeline.sh).  Order on the side stream: [convf(i+1), join event(i+1), prediction(i),
    // convf(i+2), join event(i+2), ...] - the chain's wait for join event(i+2) is also its wait for prediction(i), whose hidden state
    // (three rotating buffers) and coords1 buffer iteration i + 2 overwrites.  Measured against a third stream that started the mask
    // head right behind the GRU (two event records per iteration on the chain's stream): 196.0 against 192.4 frames/s at batch 1, 283.3
    // against 277.2 at batch 4.
    struct { bool on = false; const float* c1 = nullptr; const float* hidden = nullptr; int oi = 0; } pend;
    auto side_mask = [&]() -> int {
        if (!pend.on) return EEM_OK;
        pend.on = false;
        // (work queued on the side stream AFTER a join(): the guard joins it again if a later launch of this iteration fails and the
        // function returns early - `out`, the hidden state and coords1 are still being used over there)
        if (sd != st) guard.forked = true;
        GConvArgs m = conv_args(c, c->mk0, B, h8, w8, c->mhid.p, 256, 0, GACT_RELU);
        set_seg(m, 0, pend.hidden, 128, 128, 0);
        int r2 = gconv_launch(m, sd);
        if (r2 != EEM_OK) return r2;
        m = conv_args(c, c->mk2, B, h8, w8, c->mask.p, 576, 0, GACT_NONE);
        set_seg(m, 0, c->mhid.p, 256, 256, 0);
        m.out_scale = 0.25f;
        if ((r2 = gconv_launch(m, sd)) != EEM_OK) return r2;
        // :155-157 the convex upsampling of coords1 - coords0
        return er_convex_up_launch(c->c0.p, pend.c1, c->mask.p, out + (size_t)pend.oi * B * 2 * in_h * in_w, B, h8, w8, pad[2], pad[0], in_h,
                                   in_w, sd);
    };
    for (int it = 0; it < iters; ++it) {
        const bool emit = !c->final_only || it == iters - 1 || (it == 0 && c->keep_stages);   // (mask1 is a kept stage)
        // the hidden state alternates between two buffers, with a third between the GRU's passes: the state an iteration leaves is read
        // by its lagging mask head during the next iteration and overwritten only in the one after
        float* nin = c->net[it & 1].p;
        float* net = c->net[(it + 1) & 1].p;                               // (the state this iteration leaves)
        // coords1 lives in two buffers: iteration `it` reads c1[it & 1], the convex-upsampling launch at its end writes the updated
        // coordinates into the other one
        // EEM_ERAFT_NO_FUSE=1 (read per forward): the separate flow / coords1 += delta launches, for A/B runs and the equality test
        const char* enf = getenv("EEM_ERAFT_NO_FUSE");
        const bool fuse_small = !(enf && enf[0] == '1');
        float* c1cur = fuse_small ? c1p : c->c1.p;
        float* c1nxt = fuse_small ? c1q : c->c1.p;
        // :142 lookup; :144 flow = coords1 - coords0 into the motion features' last two channels (update.py:81), by the same launch
        // (the fork of the flow branch below: the lookup's own completion signal is the event the side stream waits for - EEM_ERAFT_FORK_RECORD=1,
        // read per forward: a hipEventRecord behind it, as through round 6's first half)
        const char* efr = getenv("EEM_ERAFT_FORK_RECORD");
        const bool fork_by_launch = fuse_small && overlap && !(efr && efr[0] == '1');
        if (fuse_small) {
            if ((rc = run_lookup(c, c1cur, c->corr.p, kCorrPad, B, h8, w8, st, c->c0.p, c->motion.p, 128, 126, fork_by_launch ? c->fork_ev : nullptr)) != EEM_OK) return rc;
        } else {
            if ((rc = run_lookup(c, c1cur, c->corr.p, kCorrPad, B, h8, w8, st)) != EEM_OK) return rc;
            if ((rc = er_flow_launch(c->c0.p, c1cur, c->motion.p, 128, 126, B, (int)g, st)) != EEM_OK) return rc;
        }
        // motion encoder (model/update.py:73-81): the flow branch convf1 -> convf2 on the side stream beside the correlation branch
        // the 324 correlation features live in a 336-channel buffer (12 zero channels, zero weight columns) so that the
        // 1x1 conv qualifies for the 16-aligned LDS-tiled kernel; the generic kernel reads the first 324
        // (measured and dropped: the correlation branch convc1 -> convc2 on the SIDE stream and the shorter flow branch on the chain's, so
        // that both waits sit at the head of a queue that has been idle for a while and the kernels with completion signals - lookup,
        // convc2 through hipExtLaunchKernelGGL - have nothing critical behind them on their own stream: 200 against 213 frames/s at batch 1,
        // 299 against 311 at batch 4.  A kernel on another stream starts ~10 us after the signal it waits for; the next kernel on the
        // signalling kernel's own stream ~5 us late)
        if (fork_by_launch) {                                          // (the lookup wrote the flow channels of `motion`)
            EEM_HIP_CHECK(hipStreamWaitEvent(sd, c->fork_ev, 0));
            guard.forked = true;
        } else if ((rc = fork()) != EEM_OK) return rc;
        GConvArgs a = conv_args(c, c->convf1, B, h8, w8, c->flo1.p, 128, 0, GACT_RELU);
        set_seg(a, 0, c->motion.p, 2, 128, 126);
        if ((rc = gconv_launch(a, sd)) != EEM_OK) return rc;
        if (wnc_upd(c->convf2, 1, B, h8, w8)) {
            if ((rc = run_wnc(c, c->convf2, c->flo1.p, B, h8, w8, c->corflo.p, 2, nullptr, sd, 256, 192)) != EEM_OK) return rc;
        } else {
            a = conv_args(c, c->convf2, B, h8, w8, c->corflo.p, 256, 192, GACT_RELU);
            set_seg(a, 0, c->flo1.p, 128, 128, 0);
            if ((rc = gconv_launch(a, sd)) != EEM_OK) return rc;
        }
        a = conv_args(c, c->convc1, B, h8, w8, c->cor1.p, 256, 0, GACT_RELU);
        set_seg(a, 0, c->corr.p, kCorrPad, kCorrPad, 0);
        if (!gconv16_supported(a)) { a.wpk16 = nullptr; set_seg(a, 0, c->corr.p, 324, kCorrPad, 0); }
        if ((rc = gconv_launch(a, st)) != EEM_OK) return rc;
        if (wnc_upd(c->convc2, 0, B, h8, w8)) {
            if ((rc = run_wnc(c, c->convc2, c->cor1.p, B, h8, w8, c->corflo.p, 2, nullptr, st, 256, 0)) != EEM_OK) return rc;
        } else {
            a = conv_args(c, c->convc2, B, h8, w8, c->corflo.p, 256, 0, GACT_RELU);
            set_seg(a, 0, c->cor1.p, 256, 256, 0);
            if ((rc = gconv_launch(a, st)) != EEM_OK) return rc;
        }
        if ((rc = join()) != EEM_OK) return rc;
        if (lagged && (rc = side_mask()) != EEM_OK) return rc;             // (the previous iteration's prediction, behind the join event)
        if (wnc_upd(c->conv, 2, B, h8, w8)) {
            if ((rc = run_wnc(c, c->conv, c->corflo.p, B, h8, w8, c->motion.p, 2, nullptr, st, 128, 0)) != EEM_OK) return rc;
        } else {
            a = conv_args(c, c->conv, B, h8, w8, c->motion.p, 128, 0, GACT_RELU);
            set_seg(a, 0, c->corflo.p, 256, 256, 0);
            if ((rc = gconv_launch(a, st)) != EEM_OK) return rc;
        }
        // SepConvGRU (model/update.py:43-60): horizontal then vertical pass
        float* hcur = nin;
        float* hnext = c->net[2].p;
        // Layers that read the same input run as ONE launch stacked along the output channels: z | r of a GRU pass (r * h then is a
        // small elementwise launch), and the first convs of the flow head and the mask head.  One forward at a time at batch 1 these were
        // 200-block launches for 256 CUs each - one round of 5 / 6-row tiles now instead of two launches that each leave the chip
        // partly empty (an update iteration at 60x80: 490 -> 430 us, 121 -> 129 frames/s; + 1.5 % at batch 4 / 8, the same with
        // frames in flight).  EEM_ERAFT_NO_STACK=1 (read per forward) keeps them apart.
        const char* ens = getenv("EEM_ERAFT_NO_STACK");
        const bool stack = !(ens && ens[0] == '1');
        for (int pass = 0; pass < 2; ++pass) {
            if (stack) {
                if (use_pre) {
                    a = conv_args(c, c->gzr_hm[pass], B, h8, w8, c->z.p, 256, 0, GACT_SIGMOID);
                    set_seg(a, 0, hcur, 128, 128, 0); set_seg(a, 1, c->motion.p, 128, 128, 0);
                    a.pre = c->czr[pass].p; a.pre_ctotal = 256; a.pre_coff = 0;
                } else {
                    a = conv_args(c, c->gzr[pass], B, h8, w8, c->z.p, 256, 0, GACT_SIGMOID);
                    set_seg(a, 0, hcur, 128, 128, 0); set_seg(a, 1, c->inp.p, 128, 128, 0); set_seg(a, 2, c->motion.p, 128, 128, 0);
                }
                // r leaves the launch as r * h (GEPI_ZR); EEM_ERAFT_NO_ZR=1: the separate elementwise launch
                const char* enz = getenv("EEM_ERAFT_NO_ZR");
                const bool zr_epi = !(enz && enz[0] == '1');
                if (zr_epi) { a.epi = GEPI_ZR; a.split = 128; a.out2 = c->rh.p; a.out2_ctotal = 128; a.e0 = hcur; a.e0_ctotal = 128; a.e0_coff = 0; }
                if ((rc = gconv_launch(a, st)) != EEM_OK) return rc;
                if (!zr_epi && (rc = er_mul_channels_launch(c->rh.p, c->z.p, 256, 128, hcur, B, 128, (long)g, st)) != EEM_OK) return rc;
            } else {
                a = conv_args(c, c->gz[pass], B, h8, w8, c->z.p, 128, 0, GACT_SIGMOID);
                set_seg(a, 0, hcur, 128, 128, 0); set_seg(a, 1, c->inp.p, 128, 128, 0); set_seg(a, 2, c->motion.p, 128, 128, 0);
                if ((rc = gconv_launch(a, st)) != EEM_OK) return rc;
                a = conv_args(c, c->gr[pass], B, h8, w8, c->rh.p, 128, 0, GACT_SIGMOID);
                set_seg(a, 0, hcur, 128, 128, 0); set_seg(a, 1, c->inp.p, 128, 128, 0); set_seg(a, 2, c->motion.p, 128, 128, 0);
                a.epi = GEPI_MUL; a.e0 = hcur; a.e0_ctotal = 128; a.e0_coff = 0;
                if ((rc = gconv_launch(a, st)) != EEM_OK) return rc;
            }
            if (use_pre) {
                a = conv_args(c, c->gq_hm[pass], B, h8, w8, hnext, 128, 0, GACT_TANH);
                set_seg(a, 0, c->rh.p, 128, 128, 0); set_seg(a, 1, c->motion.p, 128, 128, 0);
                a.pre = c->cq[pass].p; a.pre_ctotal = 128; a.pre_coff = 0;
            } else {
                a = conv_args(c, c->gq[pass], B, h8, w8, hnext, 128, 0, GACT_TANH);
                set_seg(a, 0, c->rh.p, 128, 128, 0); set_seg(a, 1, c->inp.p, 128, 128, 0); set_seg(a, 2, c->motion.p, 128, 128, 0);
            }
            a.epi = GEPI_GRU; a.e0 = hcur; a.e0_ctotal = 128; a.e0_coff = 0; a.e1 = c->z.p; a.e1_ctotal = stack ? 256 : 128; a.e1_coff = 0;
            if ((rc = gconv_launch(a, st)) != EEM_OK) return rc;
            hcur = hnext; hnext = net;
        }
        // after two passes the new hidden state is in `net`
        // flow head and mask head (model/update.py:102-105)
        if (lagged) {
            a = conv_args(c, c->fh1, B, h8, w8, c->fhid.p, 256, 0, GACT_RELU);
            set_seg(a, 0, net, 128, 128, 0);
            if ((rc = gconv_launch(a, st)) != EEM_OK) return rc;
            // the flow head's last conv leaves delta_flow and :149 coords1 + delta_flow (the other coords1 buffer) in one launch
            a = conv_args(c, c->fh2, B, h8, w8, c->delta.p, 2, 0, GACT_NONE);
            set_seg(a, 0, c->fhid.p, 256, 256, 0);
            a.epi = GEPI_SUM2; a.e0 = c1cur; a.e0_ctotal = 2; a.e0_coff = 0; a.out2 = c1nxt; a.out2_ctotal = 2;
            if (fewout_supported(a)) {
                if ((rc = gconv_launch(a, st)) != EEM_OK) return rc;
            } else {                                                         // (maps of less than 256 cells: the generic kernel, then the sum)
                a.epi = GEPI_PLAIN; a.e0 = nullptr; a.out2 = nullptr;
                if ((rc = gconv_launch(a, st)) != EEM_OK) return rc;
                if ((rc = er_sum_launch(c1nxt, c1cur, c->delta.p, (long)B * 2 * g, st)) != EEM_OK) return rc;
            }
            if (emit) {
                pend.on = true; pend.c1 = c1nxt; pend.hidden = net; pend.oi = c->final_only ? 0 : it;
                if (it == iters - 1 || (it == 0 && c->keep_stages)) {        // nothing follows / the kept stages are copied on `st`
                    if ((rc = fork()) != EEM_OK || (rc = side_mask()) != EEM_OK || (rc = join()) != EEM_OK) return rc;
                }
            }
            { float* t = c1p; c1p = c1q; c1q = t; }
        } else {
            const float* mh = c->mhid.p;
            int head_ct = 256, mh_off = 0;
            if (stack && emit) {
                a = conv_args(c, c->heads1, B, h8, w8, c->fhid.p, 512, 0, GACT_RELU);
                set_seg(a, 0, net, 128, 128, 0);
                if ((rc = gconv_launch(a, st)) != EEM_OK) return rc;
                mh = c->fhid.p; head_ct = 512; mh_off = 256;
            } else {
                a = conv_args(c, c->fh1, B, h8, w8, c->fhid.p, 256, 0, GACT_RELU);
                set_seg(a, 0, net, 128, 128, 0);
                if ((rc = gconv_launch(a, st)) != EEM_OK) return rc;
                if (emit) {
                    a = conv_args(c, c->mk0, B, h8, w8, c->mhid.p, 256, 0, GACT_RELU);
                    set_seg(a, 0, net, 128, 128, 0);
                    if ((rc = gconv_launch(a, st)) != EEM_OK) return rc;
                }
            }
            if (!emit) {
                // the flow head's last conv, then :149 coords1 = coords1 + delta_flow in place
                a = conv_args(c, c->fh2, B, h8, w8, c->delta.p, 2, 0, GACT_NONE);
                set_seg(a, 0, c->fhid.p, 256, head_ct, 0);
                if ((rc = gconv_launch(a, st)) != EEM_OK) return rc;
                if ((rc = er_axpy_launch(c1cur, c->delta.p, (long)B * 2 * g, st)) != EEM_OK) return rc;
                continue;
            }
            // the flow head's last conv (256 -> 2) beside the mask head's (256 -> 576)
            if ((rc = fork()) != EEM_OK) return rc;
            a = conv_args(c, c->fh2, B, h8, w8, c->delta.p, 2, 0, GACT_NONE);
            set_seg(a, 0, c->fhid.p, 256, head_ct, 0);
            if ((rc = gconv_launch(a, sd)) != EEM_OK) return rc;
            a = conv_args(c, c->mk2, B, h8, w8, c->mask.p, 576, 0, GACT_NONE);
            set_seg(a, 0, mh, 256, head_ct, mh_off);
            a.out_scale = 0.25f;
            if ((rc = gconv_launch(a, st)) != EEM_OK) return rc;
            if ((rc = join()) != EEM_OK) return rc;
            // :149 coords1 = coords1 + delta_flow and :155-157 the convex upsampling of coords1 - coords0, one launch
            const int oi = c->final_only ? 0 : it;
            if (fuse_small) {
                if ((rc = er_convex_up_launch(c->c0.p, c1cur, c->mask.p, out + (size_t)oi * B * 2 * in_h * in_w, B, h8, w8, pad[2],
                                              pad[0], in_h, in_w, st, c->delta.p, c1nxt)) != EEM_OK) return rc;
                float* t = c1p; c1p = c1q; c1q = t;
            } else {
                if ((rc = er_axpy_launch(c->c1.p, c->delta.p, (long)B * 2 * g, st)) != EEM_OK) return rc;
                if ((rc = er_convex_up_launch(c->c0.p, c->c1.p, c->mask.p, out + (size_t)oi * B * 2 * in_h * in_w, B, h8, w8, pad[2],
                                              pad[0], in_h, in_w, st)) != EEM_OK) return rc;
            }
        }
        if (it == 0 && c->keep_stages) {
            EEM_HIP_CHECK(hipMemcpy2DAsync(c->st_corr0.p, 324 * g * 4, c->corr.p, kCorrPad * g * 4, 324 * g * 4, B,
                                           hipMemcpyDeviceToDevice, st));
            EEM_HIP_CHECK(hipMemcpyAsync(c->st_net1.p, net, B * 128 * g * 4, hipMemcpyDeviceToDevice, st));
            EEM_HIP_CHECK(hipMemcpyAsync(c->st_mask1.p, c->mask.p, B * 576 * g * 4, hipMemcpyDeviceToDevice, st));
            EEM_HIP_CHECK(hipMemcpyAsync(c->st_delta1.p, c->delta.p, B * 2 * g * 4, hipMemcpyDeviceToDevice, st));
        }
    }
    {
        const char* enf = getenv("EEM_ERAFT_NO_FUSE");
        c->c1last = !(enf && enf[0] == '1') ? c1p : c->c1.p;
    }
    c->B = B; c->h8 = h8; c->w8 = w8; c->have_last = true;
    c->stages_valid = c->keep_stages;
    return EEM_OK;
}

extern "C" int eraft_forward(eraft_ctx* c, const float* e1, const float* e2, int batch, int in_h, int in_w, const int pad[4],
                             int iters, const float* flow_init, float* out, void* stream) {
    EEM_REQUIRE(c && e1 && e2 && out && pad, "eraft_forward: NULL argument");
    EEM_REQUIRE(c->loaded, "eraft_forward: no weights loaded");
    EEM_REQUIRE(batch >= 1 && in_h >= 1 && in_w >= 1 && iters >= 1, "eraft_forward: bad sizes");
    EEM_HIP_CHECK(hipSetDevice(c->device));
    return eraft_forward_impl(c, e1, e2, nullptr, nullptr, 0, batch, in_h, in_w, pad, iters, flow_init, out, (hipStream_t)stream);
}

// n independent samples of the evaluation loop (test_mvsec.py:580-597: one model(events1, events2) per sample at batch 1), each in its
// own tensors, as ONE batch-n forward: at 60x80 cells a sample alone leaves most CUs idle in every launch of the update block (640x480 x
// 12: 194 frames/s one sample at a time, 276 - 285 at four per call).  The predictions leave the batched chain through a staging
// buffer: one strided device copy per sample (0.3 % of the forward at four samples).
extern "C" int eraft_forward_many(eraft_ctx* c, int n, const float* const* events1, const float* const* events2, int in_h, int in_w,
                                  const int pad[4], int iters, float* const* flow_out, void* stream) {
    EEM_REQUIRE(c && events1 && events2 && flow_out && pad, "eraft_forward_many: NULL argument");
    EEM_REQUIRE(c->loaded, "eraft_forward_many: no weights loaded");
    EEM_REQUIRE(n >= 1 && n <= 16 && in_h >= 1 && in_w >= 1 && iters >= 1, "eraft_forward_many: n = %d (1..16) samples of %dx%d, %d iterations", n,
                in_h, in_w, iters);
    for (int i = 0; i < n; ++i) EEM_REQUIRE(events1[i] && events2[i] && flow_out[i], "eraft_forward_many: NULL pointer for sample %d", i);
    EEM_HIP_CHECK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    const int nout = c->final_only ? 1 : iters;
    const size_t per = (size_t)2 * in_h * in_w;                          // floats of one prediction of one sample
    int rc = ensure(c->many_out, (size_t)nout * n * per);
    if (rc != EEM_OK) return rc;
    if ((rc = eraft_forward_impl(c, nullptr, nullptr, events1, events2, n, n, in_h, in_w, pad, iters, nullptr, c->many_out.p, st)) != EEM_OK) return rc;
    for (int i = 0; i < n; ++i)                                          // [nout][n][2][H][W] -> sample i's [nout][1][2][H][W]
        EEM_HIP_CHECK(hipMemcpy2DAsync(flow_out[i], per * 4, c->many_out.p + (size_t)i * per, (size_t)n * per * 4, per * 4, nout,
                                       hipMemcpyDeviceToDevice, st));
    return EEM_OK;
}

// Keep the first iteration's corr0 / net1 / mask1 / delta1 for eraft_get_stage (four device copies per forward; off by default).
extern "C" int eraft_keep_stages(eraft_ctx* c, int enable) {
    EEM_REQUIRE(c, "eraft_keep_stages: NULL context");
    c->keep_stages = enable != 0;
    return EEM_OK;
}

// The evaluation loop reads flow_list[-1] only (test_mvsec.py:1455): with this switch the forward writes that one prediction.
extern "C" int eraft_set_final_only(eraft_ctx* c, int enable) {
    EEM_REQUIRE(c, "eraft_set_final_only: NULL context");
    c->final_only = enable != 0;
    return EEM_OK;
}

extern "C" int eraft_set_alternate_corr(eraft_ctx* c, int enable) {
    EEM_REQUIRE(c, "eraft_set_alternate_corr: NULL context");
    c->alt_corr = enable != 0;
    return EEM_OK;
}

extern "C" int eraft_set_frames_in_flight(eraft_ctx* c, int n) {
    EEM_REQUIRE(c && n >= 1, "eraft_set_frames_in_flight: need a context and n >= 1");
    c->frames_in_flight = n;
    return EEM_OK;
}

extern "C" int eraft_get_stage(eraft_ctx* c, const char* name, float* dst, size_t cap, int dims[4], void* stream) {
    EEM_REQUIRE(c && name && dims, "eraft_get_stage: NULL argument");
    EEM_REQUIRE(c->have_last, "eraft_get_stage: no forward has run");
    const std::string nm(name);
    const float* src = nullptr;
    dims[2] = c->h8; dims[3] = c->w8;
    if ((nm == "corr0" || nm == "net1" || nm == "mask1" || nm == "delta1") && !c->stages_valid) {
        eem_set_error("eraft_get_stage: '%s' is only kept after eraft_keep_stages(ctx, 1)", name);
        return EEM_ERR_STATE;
    }
    if (nm == "fmap") { src = c->fmap.p; dims[0] = 2 * c->B; dims[1] = 256; }
    else if (nm == "inp") { src = c->inp.p; dims[0] = c->B; dims[1] = 128; }
    else if (nm == "corr0") { src = c->st_corr0.p; dims[0] = c->B; dims[1] = 324; }
    else if (nm == "net1") { src = c->st_net1.p; dims[0] = c->B; dims[1] = 128; }
    else if (nm == "mask1") { src = c->st_mask1.p; dims[0] = c->B; dims[1] = 576; }
    else if (nm == "delta1") { src = c->st_delta1.p; dims[0] = c->B; dims[1] = 2; }
    else if (nm == "flow_low") { src = nullptr; dims[0] = c->B; dims[1] = 2; }
    else if (nm.size() == 4 && nm.compare(0, 3, "pyr") == 0 && nm[3] >= '0' && nm[3] <= '3') {
        const int l = nm[3] - '0';
        EEM_REQUIRE(!c->alt_corr && c->pyr[l].p, "eraft_get_stage: '%s' does not exist - the last forward computed the correlation on the fly", name);
        src = c->pyr[l].p; dims[0] = c->B * c->h8 * c->w8; dims[1] = 1; dims[2] = c->ph[l]; dims[3] = c->pw[l];
    } else {
        eem_set_error("eraft_get_stage: unknown stage '%s'", name);
        return EEM_ERR_ARG;
    }
    const size_t n = (size_t)dims[0] * dims[1] * dims[2] * dims[3];
    if (dst == nullptr) return EEM_OK;
    EEM_REQUIRE(cap >= n, "eraft_get_stage: '%s' needs %zu floats, buffer holds %zu", name, n, cap);
    if (nm == "flow_low") {
        int rc = er_flow_launch(c->c0.p, c->c1last, dst, 2, 0, c->B, c->h8 * c->w8, (hipStream_t)stream);
        return rc;
    }
    EEM_HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return EEM_OK;
}

// CorrBlock as a standalone op (model/corr.py:13-50): pyramid of (fmap1, fmap2), then one lookup at `coords`
extern "C" int eraft_corr_lookup(eraft_ctx* c, const float* fmap1, const float* fmap2, const float* coords, int batch, int ch,
                                 int h, int w, float* out, void* stream) {
    EEM_REQUIRE(c && fmap1 && fmap2 && coords && out, "eraft_corr_lookup: NULL argument");
    EEM_HIP_CHECK(hipSetDevice(c->device));
    int rc = build_pyramid(c, fmap1, fmap2, batch, ch, h, w, (hipStream_t)stream);
    if (rc != EEM_OK) return rc;
    c->B = batch; c->h8 = h; c->w8 = w; c->have_last = true;
    return run_lookup(c, coords, out, 324, batch, h, w, (hipStream_t)stream);
}

// ERAFT.upsample_flow (model/eraft.py:83-94): flow [B][2][h][w], mask [B][576][h][w] -> out [B][2][8h][8w]
extern "C" int eraft_convex_upsample(eraft_ctx* c, const float* flow, const float* mask, int batch, int h, int w, float* out,
                                     void* stream) {
    EEM_REQUIRE(c && flow && mask && out, "eraft_convex_upsample: NULL argument");
    EEM_HIP_CHECK(hipSetDevice(c->device));
    int rc = ensure(c->zeros, (size_t)batch * 2 * h * w);
    if (rc != EEM_OK) return rc;
    EEM_HIP_CHECK(hipMemsetAsync(c->zeros.p, 0, (size_t)batch * 2 * h * w * 4, (hipStream_t)stream));
    return er_convex_up_launch(c->zeros.p, flow, mask, out, batch, h, w, 0, 0, 8 * h, 8 * w, (hipStream_t)stream);
}
