// C ABI of libeemflow_hip.so (declared in include/eemflow_hip.h): context, weight packing,
// workspace management, the forward schedule and its HIP-graph cache.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "api_internal.h"
#include <mutex>

// ------------------------------------------------------------------------------- errors
static thread_local char g_err[512] = "";
thread_local int eem_last_grid_blocks = 0, eem_last_grid_threads = 0, eem_last_pipe = 0;

void eem_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* eemflow_last_error(void) { return g_err; }
extern "C" int eemflow_abi_version(void) { return 1; }

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
    if (e == hipSuccess) e = hipMalloc(&c->zero_page, 4096);
    if (e == hipSuccess) e = hipMemset(c->zero_page, 0, 4096);
    if (e == hipSuccess) e = hipMalloc(&c->io_table, 3 * EEM_MAX_COALESCE * sizeof(void*));
    if (e == hipSuccess) e = hipMemset(c->io_table, 0, 3 * EEM_MAX_COALESCE * sizeof(void*));
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
    DevBuf* bufs[] = {&c->a1, &c->f11, &c->a2, &c->b2, &c->f12, &c->a3, &c->b3, &c->f13, &c->flowcat, &c->coarse,
                      &c->padded, &c->fuse_scratch, &c->g_a1, &c->g_f11, &c->g_a2, &c->g_b2, &c->g_f12, &c->g_a3, &c->g_b3, &c->g_f13,
                      &c->g_flowcat, &c->g_coarse, &c->g_flow, &c->ups_tmp, &c->grad_flat, &c->adam_m, &c->adam_v, &c->scalars};
    for (DevBuf* b : bufs) if (b->p) (void)hipFree(b->p);
    for (int k = 0; k < 3; ++k) {
        if (c->ppart[k].p) (void)hipFree(c->ppart[k].p);
        DevBuf* kb[] = {&c->pool[k], &c->cat[k], &c->ta[k], &c->tb[k], &c->tc[k], &c->td[k], &c->t64[k], &c->t32[k],
                        &c->g_pool[k], &c->g_cat[k], &c->g_ta[k], &c->g_tb[k], &c->g_tc[k], &c->g_td[k], &c->g_t64[k], &c->g_t32[k]};
        for (DevBuf* b : kb) if (b->p) (void)hipFree(b->p);
    }
    if (c->cstream) { (void)hipStreamSynchronize(c->cstream); (void)hipStreamDestroy(c->cstream); }
    if (c->loss_ev) (void)hipEventDestroy(c->loss_ev);
    if (c->stats_ev) (void)hipEventDestroy(c->stats_ev);
    if (c->stats_host) (void)hipHostFree(c->stats_host);
    for (hipEvent_t e : c->span_ev) if (e) (void)hipEventDestroy(e);
    if (c->arena) (void)hipFree(c->arena);
    if (c->wino) (void)hipFree(c->wino);
    if (c->dec_wnc) (void)hipFree(c->dec_wnc);
    if (c->flat) (void)hipFree(c->flat);
    if (c->pack_idx) (void)hipFree(c->pack_idx);
    if (c->taps) (void)hipFree(c->taps);
    if (c->zero_page) (void)hipFree(c->zero_page);
    if (c->io_table) (void)hipFree(c->io_table);
    if (c->wstream) {
        (void)hipStreamSynchronize(c->wstream);
        (void)hipStreamDestroy(c->wstream);
        for (hipEvent_t e : c->wev) if (e) (void)hipEventDestroy(e);
        if (c->wjoin) (void)hipEventDestroy(c->wjoin);
        if (c->prep_ev) (void)hipEventDestroy(c->prep_ev);
    }
    delete c;
}

extern "C" int eemflow_load_weights(eemflow_ctx* c, const float* flat, size_t nfloats, int n_first_channels,
                                    int groups) {
    EEM_REQUIRE(c && flat, "eemflow_load_weights: NULL argument");
    // EEMFlow(config, groups, n_first_channels) (EEMFlow.py:72-75): 5 (num_voxel_bins of config/a_meshflow.json) runs the first layer on
    // its dedicated kernels; any other count on the generic convolution behind a replicate-pad launch (run_enc_layer)
    EEM_REQUIRE(n_first_channels >= 1 && n_first_channels <= 64, "n_first_channels must be in [1, 64]; got %d", n_first_channels);
    // Decoder(in_channels, groups) (EEMFlow.py:37-47): conv2..conv4 are 100 -> 100 convolutions in `groups` groups.  Built: every divisor of
    // 100 up to 5 - a decoder layer's groups of the three decoders are the jobs of ONE small-grid launch (TAIL_MAX_JOBS = 16)
    EEM_REQUIRE(groups == 1 || groups == 2 || groups == 4 || groups == 5, "groups must be 1, 2, 4 or 5 (the reference's default); got %d", groups);
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

    // ---- build the pack table: run the host packers on an "index-valued" copy of the flat vector (element i
    // holds i+1, exactly representable in fp32; 0 marks zero padding).  The arena is then a pure gather of the
    // device-resident flat weights, so an optimizer step re-packs on the device without touching the host.
    std::vector<float> idxflat(nfloats);
    for (size_t i = 0; i < nfloats; ++i) idxflat[i] = (float)(i + 1);
    const float* const base = idxflat.data();
    std::vector<float> host;
    auto push = [&host](size_t n) { size_t off = host.size(); host.resize(off + ((n + 3) & ~(size_t)3), 0.f); return off; };
    // transposed + flipped weights T[ci][co][k-1-ky][k-1-kx] packed for gconv: the data gradient of a conv is a
    // conv of the output gradient with T (transposed-stride for stride 2)
    size_t enc_floats = 0;
    for (int l = 0; l < ENC_NUM; ++l) {
        const int cin = l == 0 ? n_first_channels : kEncLayers[l].cin;
        enc_floats += (size_t)kEncLayers[l].cout * cin * 9 + kEncLayers[l].cout;
    }
    auto pack_T = [&](eemflow_ctx::ConvRef& r, const float* w, const float* b, int cin, int cout, int k, int stride) {
        r.w = (size_t)(w - base); r.b = (size_t)(b - base); r.cin = cin; r.cout = cout; r.k = k; r.stride = stride;
        std::vector<float> T((size_t)cin * cout * k * k);
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int ky = 0; ky < k; ++ky)
                    for (int kx = 0; kx < k; ++kx)
                        T[(((size_t)ci * cout + co) * k + (k - 1 - ky)) * k + (k - 1 - kx)] =
                            w[(((size_t)co * cin + ci) * k + ky) * k + kx];
        const int cs[1] = {cout};
        r.wT = push(gconv_packed_floats(cin, cs, 1, k, k));
        gconv_pack(T.data(), cin, cs, 1, k, k, host.data() + r.wT);
        // stride-1 encoder layers: the data gradient is itself one of the encoder's conv shapes (cin <-> cout),
        // so it runs on the same fast kernels with W^T
        r.fast_dgrad = (k == 3 && stride == 1 && cin == cout && (cin == 16 || cin == 32 || cin == 64) && w < base + enc_floats);
        if (r.fast_dgrad) {
            r.wT_enc = push(enc_packed_floats(cout, cin));
            enc_pack_weights(T.data(), cout, cin, host.data() + r.wT_enc);
            r.wT_enc2 = push(enc2_packed_floats(cout, cin));
            enc2_pack_weights(T.data(), cout, cin, host.data() + r.wT_enc2);
            r.zero_bias = push(cin);
        }
        // tail convs (1/64 grid): the data gradient runs on tail_conv_kernel, all decoders / groups of a layer in one launch
        r.has_tail = !(w < base + enc_floats);
        if (r.has_tail) {
            r.wT_tail = push(tail_packed_floats(cout, cin, k));
            tail_pack_weights(T.data(), cout, cin, k, host.data() + r.wT_tail);
        }
    };

    const float* p = base;
    for (int l = 0; l < ENC_NUM; ++l) {
        const int cin = l == 0 ? n_first_channels : kEncLayers[l].cin, cout = kEncLayers[l].cout;
        c->enc_w[l] = push(enc_packed_floats(cin, cout));
        enc_pack_weights(p, cin, cout, host.data() + c->enc_w[l]);
        c->enc_has2[l] = enc2_supported(cin, cout, kEncLayers[l].stride, 4);
        if (c->enc_has2[l]) {
            c->enc_w2[l] = push(enc2_packed_floats(cin, cout));
            enc2_pack_weights(p, cin, cout, host.data() + c->enc_w2[l]);
        }
        pack_T(c->t_enc[l], p, p + (size_t)cout * cin * 9, cin, cout, 3, kEncLayers[l].stride);
        p += (size_t)cout * cin * 9;
        c->enc_b[l] = push(cout);
        memcpy(host.data() + c->enc_b[l], p, cout * sizeof(float));
        p += cout;
    }
    c->enc0_generic = n_first_channels != 5;
    if (c->enc0_generic) {                                           // pconv1_1 for gconv.hip: [16][cin0][3][3], the flat vector's first tensor
        const int cs0[1] = {n_first_channels};
        c->enc0_gw = push(gconv_packed_floats(16, cs0, 1, 3, 3));
        gconv_pack(base, 16, cs0, 1, 3, 3, host.data() + c->enc0_gw);
    }
    auto tail = [&](TailW& t, eemflow_ctx::ConvRef& r, int cin, int cout, int ksize, const float* w, const float* b) {
        t.cin = cin; t.cout = cout; t.ksize = ksize;
        t.wpk = push(tail_packed_floats(cin, cout, ksize));
        tail_pack_weights(w, cin, cout, ksize, host.data() + t.wpk);
        t.bias = push(cout);
        memcpy(host.data() + t.bias, b, cout * sizeof(float));
        pack_T(r, w, b, cin, cout, ksize, 1);
    };
    for (int k = 0; k < 3; ++k) {
        tail(c->rconv[k], c->t_rconv[k], rc_in[k], 16, 3, p, p + (size_t)16 * rc_in[k] * 9);
        p += (size_t)16 * rc_in[k] * 9 + 16;
    }
    for (int k = 0; k < 3; ++k) {
        tail(c->dconv1[k], c->t_dconv1[k], kDecIn, kDecW, 3, p, p + (size_t)kDecW * kDecIn * 9);
        p += (size_t)kDecW * kDecIn * 9 + kDecW;
        for (int layer = 0; layer < 3; ++layer) {
            const float* w = p;
            const float* b = p + (size_t)kDecW * per * 9;
            for (int g = 0; g < groups; ++g)     // group g = output channels [g*per, (g+1)*per), its own `per` inputs
                tail(c->dgroup[k][layer][g], c->t_dgroup[k][layer][g], per, per, 3, w + (size_t)g * per * per * 9, b + g * per);
            p += (size_t)kDecW * per * 9 + kDecW;
        }
        tail(c->dconv5[k], c->t_dconv5[k], kDecW, 64, 3, p, p + (size_t)64 * kDecW * 9);  p += (size_t)64 * kDecW * 9 + 64;
        tail(c->dconv6[k], c->t_dconv6[k], 64, 32, 3, p, p + (size_t)32 * 64 * 9);        p += (size_t)32 * 64 * 9 + 32;
        tail(c->dconv7[k], c->t_dconv7[k], 32, 2, 3, p, p + (size_t)2 * 32 * 9);          p += (size_t)2 * 32 * 9 + 2;
    }
    tail(c->outc, c->t_outc, 6, 2, 1, p, p + 12);
    p += 14;
    if ((size_t)(p - base) != nfloats) {
        eem_set_error("eemflow_load_weights: internal layout walk consumed %zu of %zu floats", (size_t)(p - base), nfloats);
        return EEM_ERR_STATE;
    }
    std::vector<int> idx(host.size());
    for (size_t i = 0; i < host.size(); ++i) idx[i] = (int)host[i];
    drop_graph(c);
    if (c->arena) EEM_HIP_CHECK(hipFree(c->arena));
    if (c->pack_idx) EEM_HIP_CHECK(hipFree(c->pack_idx));
    if (c->flat) EEM_HIP_CHECK(hipFree(c->flat));
    c->arena = nullptr; c->pack_idx = nullptr; c->flat = nullptr;
    c->arena_floats = host.size();
    c->nflat = nfloats;
    EEM_HIP_CHECK(hipMalloc(&c->arena, host.size() * sizeof(float)));
    EEM_HIP_CHECK(hipMalloc(&c->pack_idx, host.size() * sizeof(int)));
    EEM_HIP_CHECK(hipMalloc(&c->flat, nfloats * sizeof(float)));
    EEM_HIP_CHECK(hipMemcpy(c->pack_idx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
    EEM_HIP_CHECK(hipMemcpy(c->flat, flat, nfloats * sizeof(float), hipMemcpyHostToDevice));
    int rc = repack_launch(c->flat, c->pack_idx, c->arena, (long)c->arena_floats, nullptr);
    if (rc != EEM_OK) return rc;
    {   // Winograd-domain weights for the stride-1 C->C layers
        const char* e = getenv("EEM_WINO");
        c->use_wino = !(e && atoi(e) == 0);
        const char* e4 = getenv("EEM_WINO4_LAYERS");
        c->f4_mask_env = (e && atoi(e) == 2) ? 0 : (e4 ? atoi(e4) & 7 : -1);
        size_t off = 0;
        for (int l = 0; l < ENC_NUM; ++l) {
            const EncLayerDesc& d = kEncLayers[l];
            c->enc_wino[l] = l > 0 && wino_supported(d.cin, d.cout, d.stride, 4);
            if (!c->enc_wino[l]) continue;
            for (int f = 0; f < 4; ++f) { c->wino_off[f][l] = off; off += wino_packed_floats(d.cin); }
        }
        for (int l = 0; l < ENC_NUM; ++l) {
            const EncLayerDesc& d = kEncLayers[l];
            c->enc_s2r[l] = s2r_shape(d.cin, d.cout, d.stride);
            if (c->enc_s2r[l]) { c->s2r_off[l] = off; off += s2r_packed_floats(d.cin, d.cout); }
            c->enc_bx3[l] = bx3_shape(d.cin, d.cout, d.stride);
            if (c->enc_bx3[l]) { c->bx3_off[l] = off; off += bx3_packed_floats(d.cin, d.cout); }
        }
        if (c->wino) EEM_HIP_CHECK(hipFree(c->wino));
        c->wino = nullptr;
        EEM_HIP_CHECK(hipMalloc(&c->wino, off * sizeof(float)));
        if ((rc = refresh_wino(c, nullptr)) != EEM_OK) return rc;
    }
    {   // the decoders' conv1 / conv5 streams for the Winograd kernel of conv_wnc.hip (ensure_dec_wnc fills them)
        size_t off = 0;
        for (int k = 0; k < 3; ++k) {
            for (int s = 0; s < 4; ++s) { c->dec_w1[k][s] = off; off += wnc_packed_floats(kDecIn, 0); }
            for (int s = 0; s < 2; ++s) { c->dec_w5[k][s] = off; off += wnc_packed_floats(kDecW, 0); }
            c->dec_b1[k] = off; off += 128;
            c->dec_b5[k] = off; off += 64;
        }
        if (c->dec_wnc) EEM_HIP_CHECK(hipFree(c->dec_wnc));
        c->dec_wnc = nullptr;
        EEM_HIP_CHECK(hipMalloc(&c->dec_wnc, off * sizeof(float)));
        c->dec_wnc_ok = false;
    }
    EEM_HIP_CHECK(hipDeviceSynchronize());
    c->cin0 = n_first_channels;
    c->groups = groups;
    c->weights_loaded = true;
    c->opt_step = 0;
    c->skip_counter_zeroed = false;      // the device-side count of skipped steps restarts with the step count (bias corrections use step - skipped)
    c->have_train_fwd = false;
    // a reloaded model starts a new optimisation: stale AdamW moments must not meet a fresh bias correction
    if (c->adam_m.p) EEM_HIP_CHECK(hipMemset(c->adam_m.p, 0, c->adam_m.cap * sizeof(float)));
    if (c->adam_v.p) EEM_HIP_CHECK(hipMemset(c->adam_v.p, 0, c->adam_v.cap * sizeof(float)));
    return EEM_OK;
}

// New values for the already laid-out parameters, from a DEVICE vector (state_dict order): one device-to-device copy
// and the re-pack - what a training loop with its own optimizer needs after every step.  Optimizer state of
// eemflow_optimizer_step is left alone (the caller owns the optimisation in that case).
extern "C" int eemflow_update_weights(eemflow_ctx* c, const float* flat_device, size_t nfloats, void* stream) {
    EEM_REQUIRE(c && flat_device, "eemflow_update_weights: NULL argument");
    EEM_REQUIRE(c->weights_loaded, "eemflow_update_weights: call eemflow_load_weights once first (it builds the pack tables)");
    EEM_REQUIRE(nfloats == c->nflat, "eemflow_update_weights: expected %zu floats, got %zu", c->nflat, nfloats);
    EEM_HIP_CHECK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    EEM_HIP_CHECK(hipMemcpyAsync(c->flat, flat_device, nfloats * sizeof(float), hipMemcpyDeviceToDevice, st));
    int rc = repack_launch(c->flat, c->pack_idx, c->arena, (long)c->arena_floats, st);
    if (rc != EEM_OK) return rc;
    c->have_train_fwd = false;
    return refresh_wino(c, st);
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

extern "C" int eemflow_set_deferred_input_norm(eemflow_ctx* c, int enable) {
    EEM_REQUIRE(c, "eemflow_set_deferred_input_norm: NULL context");
    c->deferred_norm = enable != 0;                                   // (part of the graph key: both forms may be cached side by side)
    return EEM_OK;
}

extern "C" int eemflow_set_frames_in_flight(eemflow_ctx* c, int n) {
    EEM_REQUIRE(c && n >= 1, "eemflow_set_frames_in_flight: need a context and n >= 1");
    if ((c->frames_in_flight >= 3) != (n >= 3)) drop_graph(c);       // the cached graphs hold the other grid sizes
    c->frames_in_flight = n;
    return EEM_OK;
}

// One forward of `batch` samples.  nframes == 0: events1 / events2 / flow are contiguous [batch, ...] tensors (e1[0], e2[0], out[0]).
// nframes == batch >= 1 (eemflow_forward_many): every sample is its own single-frame buffer triple {e1[i], e2[i], out[i]}; the two
// launches that touch caller memory find them through the io table's per-frame triples, everything between is the batch-n chain.
static int forward_common(eemflow_ctx* c, int nframes, const float* const* e1, const float* const* e2, float* const* out, int batch,
                          int in_h, int in_w, int out_h, int out_w, void* stream) {
    EEM_HIP_CHECK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    Shape s;
    int rc = compute_shape(c, batch, in_h, in_w, out_h, out_w, &s);
    if (rc != EEM_OK) return rc;
    const int nptr = nframes > 0 ? 3 * nframes : 3;
    const void* want[3 * EEM_MAX_COALESCE];
    uintptr_t bits = 0;
    for (int i = 0; i < (nframes > 0 ? nframes : 1); ++i) {
        want[3 * i] = e1[i]; want[3 * i + 1] = e2[i]; want[3 * i + 2] = out[i];
        bits |= (uintptr_t)e1[i] | (uintptr_t)e2[i] | (uintptr_t)out[i];
    }
    const int aligned = (bits & 15) == 0;
    // the table must name this call's buffers before the schedule reads it (stream-ordered; one context = one stream at a time)
    auto write_table = [&]() -> int {
        bool same = c->io_host_n == nptr && c->io_stream == stream;
        for (int i = 0; same && i < nptr; ++i) same = c->io_host[i] == want[i];
        if (same) return EEM_OK;
        const int r = nframes > 0 ? io_table_many_launch(c->io_table, nframes, e1, e2, out, st)
                                  : io_table_launch(c->io_table, e1[0], e2[0], out[0], st);
        if (r != EEM_OK) return r;
        for (int i = 0; i < nptr; ++i) c->io_host[i] = want[i];
        c->io_host_n = nptr;
        c->io_stream = stream;
        c->io_updates += 1;
        return EEM_OK;
    };

    c->workspace_overwritten();
    c->last_e1 = e1[0]; c->last_e2 = e2[0]; c->last_io_frames = nframes;
    if ((rc = ensure_forward_wino(c, batch, st)) != EEM_OK) return rc;          // outside any capture
    if (!c->use_graph || c->enc0_generic) {                           // (the generic first layer pads through the caller's pointers: no io table)
        if ((rc = alloc_workspace(c, s)) != EEM_OK) return rc;
        c->last = s;
        c->have_last = true;
        Hook hk;
        hk.st = st;
        if (nframes == 0) return run_forward(c, s, e1[0], e2[0], out[0], hk);
        if ((rc = write_table()) != EEM_OK) return rc;                // eager launches of per-frame buffers read the table too
        c->cur_io_frames = nframes;
        rc = run_forward(c, s, e1[0], e2[0], out[0], hk, c->io_table);
        c->cur_io_frames = 0;
        return rc;
    }
    const eemflow_ctx::Key key = {batch, in_h, in_w, out_h, out_w, {c->pad[0], c->pad[1], c->pad[2], c->pad[3]}, aligned, nframes,
                                  c->deferred_norm ? 1 : 0};
    eemflow_ctx::GraphEntry* ent = nullptr;
    for (eemflow_ctx::GraphEntry& g : c->graphs)
        if (g.key == key) ent = &g;
    if (ent == nullptr) {
        if ((rc = alloc_workspace(c, s)) != EEM_OK) return rc;          // may drop every entry (buffers moved)
        if ((int)c->graphs.size() >= eemflow_ctx::kMaxGraphs) {           // evict the least recently used entry
            size_t lru = 0;
            for (size_t i = 1; i < c->graphs.size(); ++i)
                if (c->graphs[i].last_use < c->graphs[lru].last_use) lru = i;
            if (c->graphs[lru].exec) (void)hipGraphExecDestroy(c->graphs[lru].exec);
            if (c->graphs[lru].graph) (void)hipGraphDestroy(c->graphs[lru].graph);
            c->graphs.erase(c->graphs.begin() + lru);
        }
        hipStream_t cap = st;
        bool own_stream = false;
        if (cap == nullptr) {        // the legacy default stream cannot be captured
            EEM_HIP_CHECK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
            own_stream = true;
        }
        EEM_HIP_CHECK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
        Hook hk;
        hk.st = cap;
        c->cur_io_frames = nframes;
        rc = run_forward(c, s, e1[0], e2[0], out[0], hk, c->io_table);
        c->cur_io_frames = 0;
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(cap, &g);
        if (own_stream) (void)hipStreamDestroy(cap);
        if (rc != EEM_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e != hipSuccess) { eem_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return EEM_ERR_HIP; }
        eemflow_ctx::GraphEntry ne;
        ne.key = key;
        ne.shape = s;
        ne.graph = g;
        ne.f13_skipped = c->f13_skipped;             // what run_forward_impl decided for this schedule
        ne.a1_skipped = c->a1_skipped;
        hipError_t ie = hipGraphInstantiate(&ne.exec, g, nullptr, nullptr, 0);
        if (ie != hipSuccess) {
            (void)hipGraphDestroy(g);
            eem_set_error("hipGraphInstantiate: %s", hipGetErrorString(ie));
            return EEM_ERR_HIP;
        }
        c->graphs.push_back(ne);
        ent = &c->graphs.back();
        c->graph_captures += 1;
    }
    if ((rc = write_table()) != EEM_OK) return rc;
    ent->last_use = ++c->graph_clock;
    c->last = ent->shape;
    c->have_last = true;
    c->graph_replays += 1;
    EEM_HIP_CHECK(hipGraphLaunch(ent->exec, st));
    // a replay runs the CAPTURED schedule: f13 is unwritten again whenever that schedule skipped its stores, whatever
    // eemflow_get_stage("f13") re-ran and cleared after an earlier frame
    c->f13_skipped = ent->f13_skipped;
    c->a1_skipped = ent->a1_skipped;
    return EEM_OK;
}

extern "C" int eemflow_forward(eemflow_ctx* c, const float* e1, const float* e2, int batch, int in_h, int in_w,
                               float* out, int out_h, int out_w, void* stream) {
    EEM_REQUIRE(c && e1 && e2 && out, "eemflow_forward: NULL argument");
    EEM_REQUIRE(c->weights_loaded, "eemflow_forward: no weights loaded");
    EEM_REQUIRE(c->have_pad, "eemflow_forward: call eemflow_set_image_size first (the reference needs "
                             "change_imagesize before forward too)");
    EEM_REQUIRE(batch >= 1 && in_h >= 1 && in_w >= 1 && out_h >= 1 && out_w >= 1, "eemflow_forward: bad sizes");
    return forward_common(c, 0, &e1, &e2, &out, batch, in_h, in_w, out_h, out_w, stream);
}

// n independent samples of the evaluation loop (test_mvsec.py:580-597: one forward per sample, batch 1) as ONE batch-n chain: the
// frames stay where the caller has them - n unrelated [1, C, H, W] event-volume pairs, n unrelated [1, 2, oh, ow] flow tensors.
// Same arithmetic as eemflow_forward on the batch of those frames (bitwise: the same kernels in the same launch configuration).
extern "C" int eemflow_forward_many(eemflow_ctx* c, int nframes, const float* const* e1, const float* const* e2, float* const* out,
                                    int in_h, int in_w, int out_h, int out_w, void* stream) {
    EEM_REQUIRE(c && e1 && e2 && out, "eemflow_forward_many: NULL argument");
    EEM_REQUIRE(nframes >= 1 && nframes <= EEM_MAX_COALESCE, "eemflow_forward_many: 1..%d frames per call; got %d", EEM_MAX_COALESCE, nframes);
    EEM_REQUIRE(c->weights_loaded, "eemflow_forward_many: no weights loaded");
    EEM_REQUIRE(c->have_pad, "eemflow_forward_many: call eemflow_set_image_size first");
    EEM_REQUIRE(in_h >= 1 && in_w >= 1 && out_h >= 1 && out_w >= 1, "eemflow_forward_many: bad sizes");
    EEM_REQUIRE(!c->enc0_generic, "eemflow_forward_many: built for the 5-bin first layer (n_first_channels == 5)");
    for (int i = 0; i < nframes; ++i) {
        EEM_REQUIRE(e1[i] && e2[i] && out[i], "eemflow_forward_many: frame %d has a NULL buffer", i);
        EEM_REQUIRE((((uintptr_t)e1[i] | (uintptr_t)e2[i] | (uintptr_t)out[i]) & 15) == 0,
                    "eemflow_forward_many: frame %d: buffers must be 16-byte aligned (torch allocations are)", i);
    }
    return forward_common(c, nframes, e1, e2, out, nframes, in_h, in_w, out_h, out_w, stream);
}

// Graph-cache statistics of a context: captures (stream captures + instantiations), replays, io-table rewrites.
extern "C" int eemflow_graph_stats(eemflow_ctx* c, long long out3[3]) {
    EEM_REQUIRE(c && out3, "eemflow_graph_stats: NULL argument");
    out3[0] = c->graph_captures; out3[1] = c->graph_replays; out3[2] = c->io_updates;
    return EEM_OK;
}

extern "C" int eemflow_time_kernels(eemflow_ctx* c, const float* e1, const float* e2, int batch, int in_h, int in_w,
                                    float* out, int out_h, int out_w, int reps, eemflow_kernel_stat* stats, int max_stats,
                                    int* nstats, void* stream) {
    EEM_REQUIRE(c && e1 && e2 && out && stats && nstats, "eemflow_time_kernels: NULL argument");
    EEM_REQUIRE(c->weights_loaded && c->have_pad, "eemflow_time_kernels: load weights and set the image size first");
    EEM_REQUIRE(reps != 0, "eemflow_time_kernels: reps=%d", reps);
    EEM_HIP_CHECK(hipSetDevice(c->device));
    Shape s;
    int rc = compute_shape(c, batch, in_h, in_w, out_h, out_w, &s);
    if (rc != EEM_OK) return rc;
    if ((rc = alloc_workspace(c, s)) != EEM_OK) return rc;
    c->workspace_overwritten();
    c->last = s;
    c->have_last = true;
    c->last_e1 = e1; c->last_e2 = e2; c->last_io_frames = 0;
    Hook hk;
    hk.st = (hipStream_t)stream;
    hk.timing = true;
    hk.repeat = reps < 0;                                // reps < 0: every kernel -reps times back to back (the single-frame table)
    if (reps < 0) reps = -reps;
    hk.reps = reps;
    if (hk.repeat) {
        rc = run_forward(c, s, e1, e2, out, hk);
    } else {
        // pass 0 warms (weights' transforms, clocks, caches) and names the launches; passes 1 .. reps are averaged
        for (int p = 0; p <= reps && rc == EEM_OK; ++p) {
            rc = run_forward(c, s, e1, e2, out, hk);
            if (rc == EEM_OK) rc = hk.collect(p > 0);
        }
    }
    hk.release();
    if (rc != EEM_OK) return rc;
    for (eemflow_kernel_stat& k : hk.stats) k.ms /= (float)reps;
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
    else if (nm == "a2") { src = c->a2.p; dims[0] = n2; dims[1] = 32; dims[2] = s.h2; dims[3] = s.w2; }
    else if (nm == "b2") { src = c->b2.p; dims[0] = n2; dims[1] = 32; dims[2] = s.h2; dims[3] = s.w2; }
    else if (nm == "a3") { src = c->a3.p; dims[0] = n2; dims[1] = 64; dims[2] = s.h3; dims[3] = s.w3; }
    else if (nm == "b3") { src = c->b3.p; dims[0] = n2; dims[1] = 64; dims[2] = s.h3; dims[3] = s.w3; }
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
    if (nm == "a1" && c->a1_skipped) {
        // the last forward computed pconv1_1 inside pconv1_2's block and never wrote a1: run that layer alone, on the same event volumes
        // (the caller's tensors of that forward must still be alive - they are whenever a stage is asked for right after a forward)
        EEM_REQUIRE(c->last_e1 && c->last_e2, "eemflow_get_stage: 'a1' needs the last forward's event volumes");
        EEM_HIP_CHECK(hipSetDevice(c->device));
        Hook hk;
        hk.st = (hipStream_t)stream;
        c->cur_io_frames = c->last_io_frames;
        const int rc = run_enc_layer(c, s, ENC_1_1, c->last_e1, c->last_e2, hk, c->last_io_frames > 0 ? c->io_table : nullptr, nullptr, false);
        c->cur_io_frames = 0;
        if (rc != EEM_OK) return rc;
        c->a1_skipped = false;
    }
    if (nm == "f13" && c->f13_skipped) {
        // the last forward pooled f13 in pconv3_3's epilogue and left the map itself unwritten: run that one layer again, with stores
        // (its input b3 is still in the workspace)
        EEM_HIP_CHECK(hipSetDevice(c->device));
        Hook hk;
        hk.st = (hipStream_t)stream;
        const int rc = run_enc_layer(c, s, ENC_NUM - 1, nullptr, nullptr, hk, nullptr, nullptr, false);
        if (rc != EEM_OK) return rc;
        c->f13_skipped = false;
    }
    EEM_REQUIRE(cap >= n, "eemflow_get_stage: '%s' needs %zu floats, buffer holds %zu", name, n, cap);
    EEM_HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return EEM_OK;
}

extern "C" int eemflow_decoder(eemflow_ctx* c, int k, const float* x, int batch, int h, int w, float* out, void* stream) {
    EEM_REQUIRE(c && x && out, "eemflow_decoder: NULL argument");
    EEM_REQUIRE(c->weights_loaded, "eemflow_decoder: no weights loaded");
    EEM_REQUIRE(k >= 1 && k <= 3 && batch >= 1 && h >= 1 && w >= 1, "eemflow_decoder: bad arguments");
    EEM_HIP_CHECK(hipSetDevice(c->device));
    const size_t g = (size_t)h * w, B = batch;
    int rc;
    const int i = k - 1;
    const unsigned long moved = g_realloc_events;
    if ((rc = ensure(c->ta[i], B * kDecW * g)) || (rc = ensure(c->tb[i], B * kDecW * g)) ||
        (rc = ensure(c->tc[i], B * kDecW * g)) || (rc = ensure(c->td[i], B * kDecW * g)) ||
        (rc = ensure(c->t64[i], B * 64 * g)) || (rc = ensure(c->t32[i], B * 32 * g)))
        return rc;
    if (g_realloc_events != moved) drop_graph(c);    // a scratch buffer the cached graphs point into has moved
    const float* cats[3] = {x, x, x};
    c->workspace_overwritten();
    Hook hk;
    hk.st = (hipStream_t)stream;
    return run_decoders(c, i, i + 1, cats, batch, h, w, out, 2, i, hk);
}

extern "C" int eemflow_local_corr53(const float* f1, const float* f2, int batch, int cch, int h, int w, float* out,
                                    void* stream) {
    EEM_REQUIRE(f1 && f2 && out && batch >= 1 && cch >= 1 && h >= 1 && w >= 1, "eemflow_local_corr53: bad arguments");
    static int* taps_of[64] = {};                  // the tap list, once per device (library-owned: callers' threads come and go)
    static std::mutex taps_lock;
    int dev = 0;
    EEM_HIP_CHECK(hipGetDevice(&dev));
    EEM_REQUIRE(dev >= 0 && dev < 64, "eemflow_local_corr53: device index");
    int* taps = nullptr;
    {
        std::lock_guard<std::mutex> guard(taps_lock);
        if (taps_of[dev] == nullptr) {
            EEM_HIP_CHECK(hipMalloc(&taps_of[dev], sizeof(kTaps53)));
            EEM_HIP_CHECK(hipMemcpy(taps_of[dev], kTaps53, sizeof(kTaps53), hipMemcpyHostToDevice));
        }
        taps = taps_of[dev];
    }
    CorrJob j = {f1, f2, out, cch, kNTaps};
    return corr_launch(&j, 1, batch, h, w, taps, kNTaps, (hipStream_t)stream);
}

extern "C" int eemflow_upsample_bilinear(const float* in, float* out, int nc, int h, int w, int oh, int ow, void* stream) {
    EEM_REQUIRE(in && out && nc >= 1 && h >= 1 && w >= 1 && oh >= 1 && ow >= 1, "eemflow_upsample_bilinear: bad arguments");
    return upsample_launch(in, out, nc, h, w, oh, ow, (hipStream_t)stream);
}

// njobs = 1 or 2 voxelizations of one grid shape in one launch sequence (voxel.hip)
static int voxelize_jobs(int njobs, const double* const* events, const int64_t* n, int bins, int h, int w, int normalize, float* const* grid,
                         int64_t* const* idx_left, int64_t* const* idx_right, void* stream) {
    // scratch arenas (band counters, moments, 16 B per event of vote records), grown on demand and owned by the LIBRARY, not by the
    // calling thread (a loader's worker threads come and go; per-thread arenas were never freed): one per (device, stream) for up to
    // eight streams, so that the voxelizations of frames in flight on different streams do not wait for each other; a ninth stream
    // takes over the least recently used arena after waiting for the kernels that last used it.  The lock is held over the three
    // launches and the event record, so a take-over always sees the event of the arena's last user.
    struct Arena { void* p = nullptr; size_t cap = 0; int dev = -1; hipEvent_t done = nullptr; void* stream = nullptr; unsigned long used = 0; };
    static Arena arenas[8];
    static unsigned long tick = 0;
    static std::mutex arena_lock;
    std::lock_guard<std::mutex> guard(arena_lock);
    int dev = 0;
    EEM_HIP_CHECK(hipGetDevice(&dev));
    Arena* ar = nullptr;
    for (Arena& a : arenas)
        if (a.p && a.dev == dev && a.stream == stream) { ar = &a; break; }
    if (!ar) {
        for (Arena& a : arenas)
            if (!ar || (!a.p && ar->p) || (!!a.p == !!ar->p && a.used < ar->used)) ar = &a;      // an empty slot, else the oldest
        if (ar->p && ar->dev == dev) EEM_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, ar->done, 0));
        ar->stream = stream;
    }
    ar->used = ++tick;
    EEM_REQUIRE(njobs >= 1 && njobs <= VOX_MAX_JOBS, "voxelize: %d event sets per call (1..%d)", njobs, VOX_MAX_JOBS);
    size_t part[VOX_MAX_JOBS] = {}, need = 0;
    for (int k = 0; k < njobs; ++k) { part[k] = (voxel_scratch_bytes(n[k]) + 255) & ~(size_t)255; need += part[k]; }
    if (ar->p == nullptr || ar->dev != dev || ar->cap < need) {
        if (ar->p) {                                                             // synchronises with work using it
            int cur = dev;
            if (ar->dev != dev) EEM_HIP_CHECK(hipSetDevice(ar->dev));
            EEM_HIP_CHECK(hipFree(ar->p));
            if (ar->dev != cur) { EEM_HIP_CHECK(hipEventDestroy(ar->done)); ar->done = nullptr; EEM_HIP_CHECK(hipSetDevice(cur)); }
        }
        ar->p = nullptr;
        ar->cap = need + need / 4;
        EEM_HIP_CHECK(hipMalloc(&ar->p, ar->cap));
        if (!ar->done) EEM_HIP_CHECK(hipEventCreateWithFlags(&ar->done, hipEventDisableTiming));
        ar->dev = dev;
    }
    void* scratch[VOX_MAX_JOBS];
    { char* q = (char*)ar->p; for (int k = 0; k < njobs; ++k) { scratch[k] = q; q += part[k]; } }
    hipEvent_t done = ar->done;
    const int rc = voxel_launch_jobs(njobs, events, n, bins, h, w, normalize, grid, idx_left, idx_right, scratch, (hipStream_t)stream);
    if (rc == EEM_OK) EEM_HIP_CHECK(hipEventRecord(done, (hipStream_t)stream));
    return rc;
}

extern "C" int eemflow_voxelize(const double* events, int64_t n, int bins, int h, int w, int normalize, float* grid,
                                int64_t* idx_left, int64_t* idx_right, void* stream) {
    EEM_REQUIRE(events && grid, "eemflow_voxelize: NULL argument");
    return voxelize_jobs(1, &events, &n, bins, h, w, normalize, &grid, &idx_left, &idx_right, stream);
}

// The event sets of SEVERAL samples (both volumes of each: 2 x the coalescing width of eemflow_forward_many) in ONE three-launch sequence:
// at the evaluation loop's 2 x 10^5 events per volume a voxelization is three launches at their fixed costs (~47 us per pair of grids,
// whatever the pair count up to the chip's width), so a call per sample spends more stream time voxelizing than the forward takes.
extern "C" int eemflow_voxelize_many(int nsets, const double* const* events, const int64_t* n_events, int bins, int h, int w, int normalize,
                                     float* const* grids, void* stream) {
    EEM_REQUIRE(events && n_events && grids, "eemflow_voxelize_many: NULL argument");
    EEM_REQUIRE(nsets >= 1 && nsets <= VOX_MAX_JOBS, "eemflow_voxelize_many: 1..%d event sets per call; got %d", VOX_MAX_JOBS, nsets);
    int64_t* none[VOX_MAX_JOBS] = {};
    for (int k = 0; k < nsets; ++k) EEM_REQUIRE(events[k] && grids[k], "eemflow_voxelize_many: set %d has a NULL buffer", k);
    return voxelize_jobs(nsets, events, n_events, bins, h, w, normalize, grids, none, none, stream);
}

// Both event sets of a sample (loader/HREM.py:226-232: event_volume_old, event_volume_new) in ONE three-launch sequence instead of
// two: the kernels are latency chains at these sizes (36 us for 2 x 10^5 events), two of them side by side cost little more than one
extern "C" int eemflow_voxelize_pair(const double* events1, int64_t n1, const double* events2, int64_t n2, int bins, int h, int w,
                                     int normalize, float* grid1, float* grid2, void* stream) {
    EEM_REQUIRE(events1 && events2 && grid1 && grid2, "eemflow_voxelize_pair: NULL argument");
    const double* ev[2] = {events1, events2};
    const int64_t n[2] = {n1, n2};
    float* g[2] = {grid1, grid2};
    int64_t* none[2] = {nullptr, nullptr};
    return voxelize_jobs(2, ev, n, bins, h, w, normalize, g, none, none, stream);
}
