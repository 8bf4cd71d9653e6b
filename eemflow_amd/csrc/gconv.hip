// Generic convolution as implicit GEMM on v_mfma_f32_32x32x2_f32:  D[cout][pixel] = sum_k A[cout][k] B[k][pixel],
// k = (tap, segment, channel).  Pixels are the flattened output index (any width works); a wave owns NPW
// pixel tiles of 32 and MTW cout tiles of 32.  Operands come straight from global memory (the maps of this
// part of the path are small and L2-resident): A fragments from the host-packed weight stream (one coalesced
// 256-B load per k-step), B gathered with the tap displacement and the zero-padding test per lane.
// Inputs may be the concatenation of up to three tensors (segments) - the reference's torch.cat's
// (model/update.py:44,51,79,99) are never materialised.  The epilogue applies a per-channel scale/shift
// (bias, folded eval-mode BatchNorm), the activation and the GRU / residual combinations.
#include <algorithm>

#include "gconv.h"

namespace {

__device__ __forceinline__ float g_act(float v, int act) {
    switch (act) {
        case GACT_RELU: return v > 0.f ? v : 0.f;
        case GACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case GACT_TANH: return tanhf(v);
        case GACT_LEAKY: return v > 0.f ? v : 0.1f * v;
        default: return v;
    }
}

// SPLITK = 4 / 8 / 16 waves (with NPW = MTW = 1): the block owns ONE 32-pixel x 32-cout tile and its waves take the k-batches
// round robin, meet in LDS and share the epilogue - for layers whose tile count leaves most SIMDs empty (E-RAFT's 60x80
// update block at batch 1: 152 blocks of 4 single-tile waves for 256 CUs; EEMFlow+'s estimator at 23 x 40 and 45 x 80: 29 / 113
// tiles, where a wave's time is the latency of its batches' first-touch weight loads and more waves mean fewer batches each).
template <int NPW, int MTW, int SPLITK = 0, int UU = 0>
__global__ __launch_bounds__(SPLITK ? SPLITK * 64 : 256) void gconv_kernel(GConvArgs a) {
    static_assert(!SPLITK || (NPW == 1 && MTW == 1), "split-K is built for single-tile waves");
    static_assert(SPLITK == 0 || SPLITK == 4 || SPLITK == 8 || SPLITK == 16, "split-K wave counts");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int n = blockIdx.z;
    const int cot0 = blockIdx.y * MTW;
    const int hwo = a.hout * a.wout, hwi = a.hin * a.win;
    const int p0 = SPLITK ? blockIdx.x * 32 : (blockIdx.x * 4 + wave) * (NPW * 32);
    if (p0 >= hwo) return;

    int oy[NPW], ox[NPW];
    bool pv[NPW];
#pragma unroll
    for (int t = 0; t < NPW; ++t) {
        const int p = p0 + t * 32 + j;
        pv[t] = p < hwo;
        const int pc = pv[t] ? p : 0;
        oy[t] = pc / a.wout;
        ox[t] = pc - oy[t] * a.wout;
    }
    int ksteps = 0;
    for (int s = 0; s < a.nseg; ++s) ksteps += (a.seg[s].c + 1) >> 1;
    ksteps *= a.kh * a.kw;

    f32x16 acc[NPW][MTW];
#pragma unroll
    for (int t = 0; t < NPW; ++t)
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][m][r] = 0.f;

    const float* wp = a.wpk + (size_t)cot0 * ksteps * 64 + lane;
    // a wave's second cout tile may lie past the last one (96 or 126 couts: three / four tiles, two per wave): its fragments are read
    // from the first tile again - the products are dropped in the epilogue, the reads must stay inside the packed stream
    int mt_off[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) mt_off[m] = (cot0 + m) * 32 < a.cout ? m : 0;
    // The k-loop is a flat sequence of batches of U k-steps: (tap, segment, first channel pair).
    //  * Addressing: a FULL batch (every channel of every pair exists) uses one base pointer per operand plus constant
    //    steps and no per-k-step predicates; the general form costs ~27 VALU / SALU instructions of 64-bit address and
    //    predicate arithmetic per MFMA (PMC: the kernel was issue-bound on them).
    //  * Latency: the operands of batch i+1 are requested before the MFMAs of batch i issue - across taps and segments
    //    too - with two register sets and scheduling barriers that keep the requests where they are written (a wave of
    //    a small layer otherwise has one batch in flight, and each is a cold miss on the weights).
    // k-steps per batch (one tap, one segment, U channel pairs): a segment of fewer pairs leaves the other slots as zero
    // work, so few-channel inputs (E-RAFT's 2-channel flow through a 7x7 conv, the 5-bin event volumes) use short batches
    constexpr int U = UU ? UU : ((NPW * MTW == 1) ? 16 : 8);
    struct Pos { int tap, s, cp0, ks; };                     // ks = k-step index of the batch's first pair
    const int ntaps = a.kh * a.kw;
    auto advance1 = [&](Pos q) {
        if (q.tap >= ntaps) return q;
        const int np = (a.seg[q.s].c + 1) >> 1;
        q.ks += min(U, np - q.cp0);
        q.cp0 += U;
        if (q.cp0 >= np) { q.cp0 = 0; if (++q.s == a.nseg) { q.s = 0; ++q.tap; } }
        return q;
    };
    auto advance = [&](Pos q) {                              // to this wave's next batch
        q = advance1(q);
#pragma unroll
        for (int w = 1; w < SPLITK; ++w) q = advance1(q);
        return q;
    };
    auto load = [&](const Pos& q, float (&av)[U][MTW], float (&bv)[U][NPW]) {
        const int ty = q.tap / a.kw, tx = q.tap - ty * a.kw;
        int off[NPW];
        bool tv[NPW];
#pragma unroll
        for (int t = 0; t < NPW; ++t) {
            int iy = oy[t] * a.stride - a.pad_h + ty, ix = ox[t] * a.stride - a.pad_w + tx;
            bool par = true;
            if (a.tstride > 1) {
                par = iy >= 0 && ix >= 0 && (iy % a.tstride) == 0 && (ix % a.tstride) == 0;
                iy /= a.tstride; ix /= a.tstride;
            }
            tv[t] = pv[t] && par && iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win;
            off[t] = tv[t] ? iy * a.win + ix : 0;
        }
        const GConvSeg sg = a.seg[q.s];
        const int cmul = sg.cmul > 1 ? sg.cmul : 1;
        const size_t seg_off = ((size_t)n * sg.ctotal + sg.coff) * hwi;
        const float* base = sg.ptr + seg_off;
        const float* gbase = sg.gate ? sg.gate + seg_off : nullptr;
        if ((q.cp0 + U) * 2 <= sg.c) {
            const float* ap = wp + (size_t)q.ks * 64;
            const size_t bstep = (size_t)2 * cmul * hwi;
            const float* bp[NPW];
            const float* gp[NPW];
#pragma unroll
            for (int t = 0; t < NPW; ++t) {
                const size_t o = (size_t)(q.cp0 * 2 + h) * cmul * hwi + off[t];
                bp[t] = base + o;
                gp[t] = gbase ? gbase + o : nullptr;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int m = 0; m < MTW; ++m) av[u][m] = ap[(size_t)mt_off[m] * ksteps * 64 + u * 64];
#pragma unroll
                for (int t = 0; t < NPW; ++t) {
                    float x = *bp[t];
                    bp[t] += bstep;
                    if (gbase) { x *= (*gp[t] > 0.f) ? 1.f : 0.1f; gp[t] += bstep; }
                    bv[u][t] = tv[t] ? x : 0.f;
                }
            }
        } else {
            const int cp_n = (sg.c + 1) >> 1;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cp = q.cp0 + u;
                const bool ok = cp < cp_n;
                const int c = cp * 2 + h;
                const bool cv = ok && c < sg.c;
                const size_t coffs = (size_t)(cv ? c : 0) * cmul * hwi;
                const float* bp = base + coffs;
                const int kk = ok ? q.ks + u : q.ks;         // surplus slots re-read a valid fragment; their B is 0
#pragma unroll
                for (int m = 0; m < MTW; ++m) av[u][m] = wp[((size_t)mt_off[m] * ksteps + kk) * 64];
#pragma unroll
                for (int t = 0; t < NPW; ++t) {
                    float x = bp[off[t]];
                    if (gbase) x *= (gbase[coffs + off[t]] > 0.f) ? 1.f : 0.1f;
                    bv[u][t] = (tv[t] && cv) ? x : 0.f;
                }
            }
        }
    };
    auto mfma = [&](const float (&av)[U][MTW], const float (&bv)[U][NPW]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < NPW; ++t)
#pragma unroll
                for (int m = 0; m < MTW; ++m)
                    acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][m], bv[u][t], acc[t][m], 0, 0, 0);
    };
    {
        float avA[U][MTW], bvA[U][NPW], avB[U][MTW], bvB[U][NPW];
        Pos cur = {0, 0, 0, 0};
        if (SPLITK)
            for (int w = 0; w < wave; ++w) cur = advance1(cur);
        if (cur.tap < ntaps) load(cur, avA, bvA);
        while (cur.tap < ntaps) {
            const Pos nx = advance(cur);
            const bool more = nx.tap < ntaps;
            if (more) load(nx, avB, bvB);
            __builtin_amdgcn_sched_barrier(0);
            mfma(avA, bvA);
            __builtin_amdgcn_sched_barrier(0);
            if (!more) break;
            cur = advance(nx);
            const bool more2 = cur.tap < ntaps;
            if (more2) load(cur, avA, bvA);
            __builtin_amdgcn_sched_barrier(0);
            mfma(avB, bvB);
            __builtin_amdgcn_sched_barrier(0);
            if (!more2) break;
        }
    }

    // ---- epilogue
    auto finish = [&](float v, int co, int p) {
        if (a.scale) v *= a.scale[co];
        if (a.shift) v += a.shift[co];
        if (a.pre) v += a.pre[((size_t)n * a.pre_ctotal + a.pre_coff + co) * hwo + p];
        v = g_act(v, a.act);
        if (a.epi == GEPI_MUL) {
            v *= a.e0[((size_t)n * a.e0_ctotal + a.e0_coff + co) * hwo + p];
        } else if (a.epi == GEPI_GRU) {
            const float hh = a.e0[((size_t)n * a.e0_ctotal + a.e0_coff + co) * hwo + p];
            const float z = a.e1[((size_t)n * a.e1_ctotal + a.e1_coff + co) * hwo + p];
            v = (1.f - z) * hh + z * v;
        } else if (a.epi == GEPI_ADD_RELU) {
            v += a.e0[((size_t)n * a.e0_ctotal + a.e0_coff + co) * hwo + p];
            v = v > 0.f ? v : 0.f;
        } else if (a.epi == GEPI_ADD) {
            v += a.e0[((size_t)n * a.e0_ctotal + a.e0_coff + co) * hwo + p];
        }
        const int oc = a.out_coff + co * (a.out_cmul > 1 ? a.out_cmul : 1);
        float* dst = a.out + ((size_t)n * a.out_ctotal + oc) * hwo + p;
        // GEPI_ZR: the r half leaves as r * h, to the second output.  One store through a selected pointer: a second store in this
        // else-if chain (with an early exit) made hipcc emit code that stores through `out2` on the other epilogues' paths too
        if (a.epi == GEPI_ZR && co >= a.split) {
            const int c2 = co - a.split;
            v *= a.e0[((size_t)n * a.e0_ctotal + a.e0_coff + c2) * hwo + p];
            dst = a.out2 + ((size_t)n * a.out2_ctotal + c2) * hwo + p;
        }
        *dst = v * a.out_scale;
    };
    if (SPLITK) {
        // [wave][register][lane]: every wave leaves its 16 partial sums, then wave w finishes registers R*w .. R*w + R-1
        constexpr int SK = SPLITK ? SPLITK : 4, R = 16 / SK;
        __shared__ float red[SK][16][64];
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[0][0][r];
        __syncthreads();
        const int p = p0 + j;
        if (!pv[0]) return;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int r = wave * R + rr;
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < SK; q += 4) v += (red[q][r][lane] + red[q + 1][r][lane]) + (red[q + 2][r][lane] + red[q + 3][r][lane]);
            const int co = cot0 * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (co < a.cout) finish(v, co, p);
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < NPW; ++t) {
        if (!pv[t]) continue;
        const int p = p0 + t * 32 + j;
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (cot0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (co < a.cout) finish(acc[t][m][r], co, p);
            }
    }
}

}  // namespace

size_t gconv_packed_floats(int cout, const int* cs, int nseg, int kh, int kw) {
    int cps = 0;
    for (int s = 0; s < nseg; ++s) cps += (cs[s] + 1) / 2;
    return (size_t)ceil_div(cout, 32) * kh * kw * cps * 64;
}

void gconv_pack(const float* w, int cout, const int* cs, int nseg, int kh, int kw, float* packed) {
    int cin = 0, cps = 0;
    for (int s = 0; s < nseg; ++s) { cin += cs[s]; cps += (cs[s] + 1) / 2; }
    const int taps = kh * kw, ksteps = taps * cps;
    for (int cot = 0; cot < ceil_div(cout, 32); ++cot) {
        int ks = 0;
        for (int tap = 0; tap < taps; ++tap) {
            int cbase = 0;
            for (int s = 0; s < nseg; ++s) {
                for (int cp = 0; cp < (cs[s] + 1) / 2; ++cp, ++ks)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int co = cot * 32 + (lane & 31), c = cp * 2 + (lane >> 5);
                        float v = 0.f;
                        if (co < cout && c < cs[s]) v = w[((size_t)co * cin + cbase + c) * taps + tap];
                        packed[((size_t)cot * ksteps + ks) * 64 + lane] = v;
                    }
                cbase += cs[s];
            }
        }
    }
}

namespace {

// Few-channel inputs (E-RAFT's 2-channel flow through the 7x7 convf1, model/update.py:70; the 5-bin event volumes through the
// encoders' 7x7 stride-2 conv1, model/extractor.py:136): the k-steps of a pass are the taps of one channel pair.  The generic kernel
// walks them as 49 batches of 1-4 k-steps, each a round trip to L2 (36 us for 120 MFLOP at 60x80); here a wave requests the A
// fragments and the B values of ALL the taps of a pair at once - 2 x TAPS registers, one latency - and then issues the TAPS MFMAs.
// Wave = 32 pixels x 32 couts, block = 4 pixel tiles; weights in gconv_pack's order (k-step = tap).
template <int KH, int KW>
__global__ __launch_bounds__(256) void gconv_taps_kernel(GConvArgs a) {
    constexpr int TAPS = KH * KW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int n = blockIdx.z, cot = blockIdx.y;
    const int hwo = a.hout * a.wout, hwi = a.hin * a.win;
    const int p0 = (blockIdx.x * 4 + wave) * 32;
    if (p0 >= hwo) return;
    const int p = p0 + j;
    const bool pv = p < hwo;
    const int pc = pv ? p : 0;
    const int oy = pc / a.wout, ox = pc - oy * a.wout;
    const GConvSeg& sg = a.seg[0];
    const int cps = (sg.c + 1) >> 1;                              // channel pairs: one pass (round trip) each
    const float* wp = a.wpk + (size_t)cot * TAPS * cps * 64 + lane;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 1
    for (int cp = 0; cp < cps; ++cp) {
        const bool cv = cp * 2 + h < sg.c;                        // an odd channel count leaves the last pair's second half empty
        const float* in = sg.ptr + ((size_t)n * sg.ctotal + sg.coff + (cv ? cp * 2 + h : 0)) * hwi;
        float av[TAPS], bv[TAPS];
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int iy = oy * a.stride - a.pad_h + t / KW, ix = ox * a.stride - a.pad_w + t % KW;
            const bool ok = pv && cv && iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win;
            av[t] = wp[(t * cps + cp) * 64];
            const float x = in[ok ? iy * a.win + ix : 0];
            bv[t] = ok ? x : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);          // without it the scheduler pairs every tap's two loads with its MFMA: TAPS round trips
#pragma unroll
        for (int t = 0; t < TAPS; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);
    }
    if (!pv) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (co >= a.cout) continue;
        float v = acc[r];
        if (a.scale) v *= a.scale[co];
        if (a.shift) v += a.shift[co];
        v = g_act(v, a.act);
        const int oc = a.out_coff + co * (a.out_cmul > 1 ? a.out_cmul : 1);
        a.out[((size_t)n * a.out_ctotal + oc) * hwo + p] = v * a.out_scale;
    }
}

// direct 3x3 convolution for <= 8 output channels.  Block = 64 consecutive pixels x FEW_WAVES channel groups: wave g takes the
// channels [g * cpw, (g + 1) * cpw) of all 64 pixels (every tap is a 256-byte run per wave; the 8 weights of a (channel, tap) are
// one uniform 32-byte load, scalar registers feed the FMAs), the partial sums meet in LDS and thread (co, pixel) finishes one output.
// Splitting the channels rather than the pixels over the waves is what fills the chip: 180 x 320 pixels are 900 wave-rows, less
// than one per SIMD, and a lone wave per SIMD cannot hide its own load latency.
// FEW_WAVES = 8, CA = 1 (a channel ahead, 63 VGPRs, 8 waves per SIMD) for launches that fill the chip; 16 waves x CA channels per
// request group, a group ahead, for the layers of many channels, where a wave's time is the number of its request round trips
// (E-RAFT's flow head 256 -> 2 at 60x80: 75 blocks; 26 -> 12 us with CA = 2; four channels per group are no faster, and 16 waves
// at batch 4 - 300 blocks - are slower than 8: 257 against 278 frames/s).  NCO = 2: the layers of at most two couts (that flow head)
// keep two accumulators instead of eight.
template <int FEW_WAVES, int CA, int NCO>
__global__ __launch_bounds__(FEW_WAVES * 64, (FEW_WAVES == 8 ? 8 : 4)) void fewout_kernel(GConvArgs a, int cpw) {
    __shared__ float part[FEW_WAVES][NCO][64];
    const int hw = a.hin * a.win;
    const int lane = threadIdx.x & 63;
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long total = (long)a.n * hw;
    const long idx = (long)blockIdx.x * 64 + lane;
    const bool live = idx < total;
    const long idc = live ? idx : total - 1;
    const int p = idc % hw, n = idc / hw;
    const int y = p / a.win, x = p - y * a.win;
    const GConvSeg& sg = a.seg[0];
    const int c0 = g * cpw, c1 = min(c0 + cpw, sg.c);
    const float* in = sg.ptr + ((size_t)n * sg.ctotal + sg.coff + c0) * hw;
    // (pairs of couts as packed registers: v_pk_fma_f32 does two of the layer's FMAs per issue slot - this kernel is the vector pipe's)
    constexpr int NP = (NCO + 1) / 2;
    f32x2 acc2[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) acc2[k] = f32x2{0.f, 0.f};
    int off[9];
    bool keep[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        const bool ok = yy >= 0 && yy < a.hin && xx >= 0 && xx < a.win;
        off[t] = ok ? yy * a.win + xx : p;
        keep[t] = ok;
    }
    constexpr int WS = NCO <= 2 ? 2 : 8;                            // floats per (channel, tap) of the packing (fewout_pack)
    const float* w = a.wfew + (size_t)c0 * 9 * WS;
    // two register sets, the next group's taps (CA channels x 9) in flight under this group's FMAs (requests past the last channel
    // repeat it; their products are skipped)
    float va[CA][9], vb[CA][9];
    auto request = [&](float (&v)[CA][9], int c) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < CA; ++q) {
            const float* plane = in + (size_t)(min(c + q, c1 - 1) - c0) * hw;
#pragma unroll
            for (int t = 0; t < 9; ++t) v[q][t] = plane[off[t]];
        }
    };
    auto fma = [&](const float (&v)[CA][9], int c) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < CA; ++q) {
            if (c + q >= c1) break;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float vv = keep[t] ? v[q][t] : 0.f;
                const f32x2* w8 = reinterpret_cast<const f32x2*>(w + ((size_t)(c + q - c0) * 9 + t) * WS);
                const f32x2 v2 = {vv, vv};
#pragma unroll
                for (int k = 0; k < NP; ++k) acc2[k] = __builtin_elementwise_fma(v2, w8[k], acc2[k]);
            }
        }
    };
    if (c0 < c1) {
        request(va, c0);
#pragma unroll 1
        for (int c = c0; c < c1; c += 2 * CA) {
            request(vb, c + CA);
            fma(va, c);
            if (c + CA >= c1) break;
            request(va, c + 2 * CA);
            fma(vb, c + CA);
        }
    }
#pragma unroll
    for (int co = 0; co < NCO; ++co) part[g][co][lane] = acc2[co >> 1][co & 1];
    __syncthreads();
    // thread (co = g, pixel = lane) sums the channel groups in order and finishes the output
    const int co = g;
    if (!live || co >= NCO || co >= a.cout) return;
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < FEW_WAVES; ++k) r += part[k][co][lane];
    if (a.scale) r *= a.scale[co];
    if (a.shift) r += a.shift[co];
    if (a.act == GACT_RELU) r = r > 0.f ? r : 0.f;
    else if (a.act == GACT_LEAKY) r = r > 0.f ? r : 0.1f * r;
    else if (a.act == GACT_SIGMOID) r = 1.f / (1.f + expf(-r));
    else if (a.act == GACT_TANH) r = tanhf(r);
    if (a.epi == GEPI_ADD) r += a.e0[((size_t)n * a.e0_ctotal + a.e0_coff + co) * hw + p];
    const int oc = a.out_coff + co * (a.out_cmul > 1 ? a.out_cmul : 1);
    r *= a.out_scale;
    a.out[((size_t)n * a.out_ctotal + oc) * hw + p] = r;
    if (a.epi == GEPI_SUM2) a.out2[((size_t)n * a.out2_ctotal + co) * hw + p] = a.e0[((size_t)n * a.e0_ctotal + a.e0_coff + co) * hw + p] + r;
}


// The same conv for launches of less than a wave per SIMD (E-RAFT's flow head 256 -> 2 at 60x80: 75 blocks of the form above, a wave's time =
// its eight request round trips to a map the previous launch left in the Infinity Cache).  Block = 32 consecutive pixels x 32 channel
// groups: the two halves of a wave take different groups of the same 32 pixels, a lane holds ALL taps of its group's <= 8 channels in
// registers - one round trip - and 150 blocks share the vector-memory path of twice as many CUs.  The layer's weights ([cin][tap][2],
// 18 KB at 256 channels; [cin][tap][8] for the layers of 3 .. 8 couts) are staged in LDS under that round trip and a lane reads its half's pair per (channel, tap) as one 8-byte LDS
// read: the first version selected between two uniform (scalar-register) pairs per lane and spent 1 190 vector instructions per wave on
// 145 FMAs - moves out of scalar registers and selects - which is what its 13.7 us were (the 64-pixel form: 13.6).  Summation order
// differs from the form above (32 partial sums of 8 channels instead of 16 of 16): equal within rounding, not bitwise.  The template
// also builds for 4 and 8 couts (weights [cin][tap][8]); EEMFlow+'s mask estimator tail at 90 x 160 (176 -> 8, 184 -> 3: 450 such blocks,
// two rounds) measured 22.4 and 18.8 us on it against 18.4 and 18.0 on the 64-pixel form, so only the two-cout layers are sent here.
template <int NCO>
__global__ __launch_bounds__(1024) void fewout_wide_kernel(GConvArgs a, int cpl) {
    constexpr int CPL = 8;
    constexpr int WS = NCO <= 2 ? 2 : 8;                          // floats per (channel, tap) of the packing (fewout_pack)
    constexpr int NP = (NCO + 1) / 2;
    constexpr int WPAIRS = 32 * CPL * 9 * WS / 2, WIT = (WPAIRS + 1023) / 1024;
    __shared__ f32x2 wl[WPAIRS];                                  // [channel][tap][WS / 2] pairs; channels past the layer's last: zeros
    __shared__ float part[32][NCO][32];
    const int hw = a.hin * a.win;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = lane >> 5, px = lane & 31;
    const long total = (long)a.n * hw;
    const long idx = (long)blockIdx.x * 32 + px;
    const long idc = idx < total ? idx : total - 1;
    const int p = idc % hw, n = idc / hw;
    const int y = p / a.win, x = p - y * a.win;
    const GConvSeg& sg = a.seg[0];
    // the weights' requests first: their data is back before the taps' (requests return in order)
    const int npair = sg.c * 9 * WS / 2;
    f32x2 wreg[WIT];
#pragma unroll
    for (int k = 0; k < WIT; ++k) {
        const int e = tid + 1024 * k;
        wreg[k] = e < npair ? reinterpret_cast<const f32x2*>(a.wfew)[e] : f32x2{0.f, 0.f};
    }
    const int c0 = (2 * g + sub) * cpl;
    const float* in = sg.ptr + ((size_t)n * sg.ctotal + sg.coff) * hw;
    int off[9];
    bool keep[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        const bool ok = yy >= 0 && yy < a.hin && xx >= 0 && xx < a.win;
        off[t] = ok ? yy * a.win + xx : p;
        keep[t] = ok;
    }
    float v[CPL][9];
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const float* plane = in + (size_t)min(c0 + q, sg.c - 1) * hw;     // (requests past the layer's last channel meet zero weights)
#pragma unroll
        for (int t = 0; t < 9; ++t) v[q][t] = plane[off[t]];
    }
#pragma unroll
    for (int k = 0; k < WIT; ++k)
        if (tid + 1024 * k < WPAIRS) wl[tid + 1024 * k] = wreg[k];
    __syncthreads();
    f32x2 acc[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) acc[k] = f32x2{0.f, 0.f};
    const f32x2* wc = wl + c0 * 9 * (WS / 2);
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        if (q >= cpl) break;                                       // (uniform: channels c0 + cpl .. are the next group's)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float vv = keep[t] ? v[q][t] : 0.f;
#pragma unroll
            for (int k = 0; k < NP; ++k) acc[k] = __builtin_elementwise_fma(f32x2{vv, vv}, wc[(q * 9 + t) * (WS / 2) + k], acc[k]);
        }
    }
#pragma unroll
    for (int co = 0; co < NCO; ++co) part[2 * g + sub][co][px] = acc[co >> 1][co & 1];
    __syncthreads();
    const int co = tid >> 5;
    const long oidx = (long)blockIdx.x * 32 + (tid & 31);
    if (co >= NCO || co >= a.cout || oidx >= total) return;
    const int op = oidx % hw, on = oidx / hw;
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) r += part[k][co][tid & 31];
    if (a.scale) r *= a.scale[co];
    if (a.shift) r += a.shift[co];
    if (a.act == GACT_RELU) r = r > 0.f ? r : 0.f;
    else if (a.act == GACT_LEAKY) r = r > 0.f ? r : 0.1f * r;
    else if (a.act == GACT_SIGMOID) r = 1.f / (1.f + expf(-r));
    else if (a.act == GACT_TANH) r = tanhf(r);
    if (a.epi == GEPI_ADD) r += a.e0[((size_t)on * a.e0_ctotal + a.e0_coff + co) * hw + op];
    const int oc = a.out_coff + co * (a.out_cmul > 1 ? a.out_cmul : 1);
    r *= a.out_scale;
    a.out[((size_t)on * a.out_ctotal + oc) * hw + op] = r;
    if (a.epi == GEPI_SUM2) a.out2[((size_t)on * a.out2_ctotal + co) * hw + op] = a.e0[((size_t)on * a.e0_ctotal + a.e0_coff + co) * hw + op] + r;
}

}  // namespace

size_t fewout_packed_floats(int cin, int kh, int kw) { return (size_t)cin * kh * kw * 8; }

// [cin][tap][8]; layers of at most two couts: [cin][tap][2] in the first quarter of the same allocation - a channel's 18 floats are
// consecutive, so the wide form's uniform loads are two wide scalar loads per channel instead of nine 8-byte ones
void fewout_pack(const float* w, int cout, int cin, int kh, int kw, float* packed) {
    const int taps = kh * kw, ws = cout <= 2 ? 2 : 8;
    for (int c = 0; c < cin; ++c)
        for (int t = 0; t < taps; ++t)
            for (int co = 0; co < ws; ++co)
                packed[((size_t)c * taps + t) * ws + co] = co < cout ? w[((size_t)co * cin + c) * taps + t] : 0.f;
}

bool fewout_supported(const GConvArgs& a) {
    static const long few_min_px = [] { const char* m = getenv("EEM_FEWOUT_MINPX"); return m ? atol(m) : 256L; }();
    const char* e = getenv("EEM_NO_FEWOUT");                         // read per call: a test flips it inside one process
    if (e && e[0] == '1') return false;
    return a.wfew && a.nseg == 1 && a.cout <= 8 && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.tstride <= 1 && a.pad_h == 1 && a.pad_w == 1 &&
           a.seg[0].cmul <= 1 && a.seg[0].gate == nullptr && a.pre == nullptr && (a.epi == GEPI_PLAIN || a.epi == GEPI_ADD || a.epi == GEPI_SUM2) && a.hout == a.hin && a.wout == a.win &&
           (long)a.hin * a.win >= few_min_px;                        // smaller maps: the split-K launch of the generic kernel
}

int fewout_launch(const GConvArgs& a, hipStream_t stream) {
    const long n = (long)a.n * a.hin * a.win;
    const unsigned blocks = (unsigned)((n + 63) / 64);
    // less than a wave per SIMD - or, for the layers of 3 .. 8 couts (EEMFlow+'s mask estimator tail at 96 x 160: 240 blocks), less than
    // EEM_FEWOUT_SMALL_BLOCKS (read once; 512): a wave's time there is the number of its request round trips (22 channels, one ahead)
    static const long small_blocks = [] { const char* e = getenv("EEM_FEWOUT_SMALL_BLOCKS"); return e ? atol(e) : 512L; }();
    const bool small = (blocks * 8 < 1024 || (a.cout > 2 && (long)blocks < small_blocks)) && a.seg[0].c >= 64;
    if (a.cout <= 2) {                                                // E-RAFT's flow head
        // EEM_FEWOUT_WIDE=0 (read per call: the equality test flips it): the 64-pixel form for the small launches too
        const char* ew = getenv("EEM_FEWOUT_WIDE");
        if (small && a.seg[0].c <= 256 && !(ew && ew[0] == '0')) {
            hipLaunchKernelGGL((fewout_wide_kernel<2>), dim3((unsigned)((n + 31) / 32)), dim3(1024), 0, stream, a, (a.seg[0].c + 31) / 32);
        } else if (small) hipLaunchKernelGGL((fewout_kernel<16, 2, 2>), dim3(blocks), dim3(1024), 0, stream, a, (a.seg[0].c + 15) / 16);
        else hipLaunchKernelGGL((fewout_kernel<8, 1, 2>), dim3(blocks), dim3(512), 0, stream, a, (a.seg[0].c + 7) / 8);
    } else if (a.cout <= 4 && !small) {                               // EEMFlow+'s mask estimator tail 184 -> 3 (cdc_utils.py:151)
        hipLaunchKernelGGL((fewout_kernel<8, 1, 4>), dim3(blocks), dim3(512), 0, stream, a, (a.seg[0].c + 7) / 8);
    } else if (small) {
        hipLaunchKernelGGL((fewout_kernel<16, 2, 8>), dim3(blocks), dim3(1024), 0, stream, a, (a.seg[0].c + 15) / 16);
    } else {
        hipLaunchKernelGGL((fewout_kernel<8, 1, 8>), dim3(blocks), dim3(512), 0, stream, a, (a.seg[0].c + 7) / 8);
    }
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int gconv_launch(const GConvArgs& a, hipStream_t stream) {
    EEM_REQUIRE(a.nseg >= 1 && a.nseg <= 3 && a.n >= 1 && a.cout >= 1, "gconv_launch: bad arguments");
    if (a.groups > 1) {
        EEM_REQUIRE(gconv16_supported(a), "gconv_launch: a grouped launch needs the LDS-tiled kernel (the caller checks gconv16_supported)");
        return gconv16_launch(a, stream);
    }
    if (fewout_supported(a)) return fewout_launch(a, stream);
    if (stem7_supported(a)) return stem7_launch(a, stream);
    if (gconvb_supported(a)) return gconvb_launch(a, stream);
    if (gconv16_supported(a)) return gconv16_launch(a, stream);
    const int hwo = a.hout * a.wout;
    const int cot = ceil_div(a.cout, 32);
    // big problems: 2x2 tiles per wave (half the operand traffic per MFMA); small ones: 1x1 for parallelism
    const long waves22 = (long)ceil_div(hwo, 64) * ceil_div(cot, 2) * a.n;
    int maxpairs = 0;
    for (int sgi = 0; sgi < a.nseg; ++sgi) maxpairs = std::max(maxpairs, (a.seg[sgi].c + 1) / 2);
    {
        const char* e = getenv("EEM_NO_TAPS_KERNEL");                // read per call: a test flips it inside one process
        const bool off = e && e[0] == '1';
        const bool shape7 = a.kh == 7 && a.kw == 7, shape3 = a.kh == 3 && a.kw == 3;
        // (one pair only: with the 5-bin volumes' three pairs through a stride-2 7x7 the 2x2-tile batches of the generic kernel are
        // faster - measured, E-RAFT 122 -> 117 frames/s)
        static const int taps_maxc = [] { const char* e = getenv("EEM_TAPS_MAXC"); return e ? atoi(e) : 6; }();   // 7x7: up to three channel pairs (the encoders' stems on 5-bin volumes; 2 = the flow conv only)
        if (!off && a.nseg == 1 && a.seg[0].c <= (shape7 ? taps_maxc : 2) && (shape7 || shape3) && a.tstride <= 1 && a.epi == GEPI_PLAIN && a.pre == nullptr && a.seg[0].gate == nullptr &&
            a.seg[0].cmul <= 1) {
            dim3 grid(ceil_div(hwo, 128), cot, a.n);
            if (shape7) hipLaunchKernelGGL((gconv_taps_kernel<7, 7>), grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((gconv_taps_kernel<3, 3>), grid, dim3(256), 0, stream, a);
            EEM_HIP_CHECK(hipGetLastError());
            return EEM_OK;
        }
    }
    if (maxpairs <= 4) {                                             // short batches for few-channel inputs
        if (waves22 >= 2048 && cot >= 2) {
            dim3 grid(ceil_div(hwo, 256), ceil_div(cot, 2), a.n);
            if (maxpairs <= 1) hipLaunchKernelGGL((gconv_kernel<2, 2, false, 1>), grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((gconv_kernel<2, 2, false, 4>), grid, dim3(256), 0, stream, a);
        } else {
            dim3 grid(ceil_div(hwo, 128), cot, a.n);
            if (maxpairs <= 1) hipLaunchKernelGGL((gconv_kernel<1, 1, false, 1>), grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((gconv_kernel<1, 1, false, 4>), grid, dim3(256), 0, stream, a);
        }
        EEM_HIP_CHECK(hipGetLastError());
        return EEM_OK;
    }
    if (waves22 >= 2048 && cot >= 2) {
        dim3 grid(ceil_div(hwo, 256), ceil_div(cot, 2), a.n);
        hipLaunchKernelGGL((gconv_kernel<2, 2>), grid, dim3(256), 0, stream, a);
    } else if ((long)ceil_div(hwo, 64) * cot * a.n >= 2048) {
        dim3 grid(ceil_div(hwo, 256), cot, a.n);
        hipLaunchKernelGGL((gconv_kernel<2, 1>), grid, dim3(256), 0, stream, a);
    } else {
        int ksteps = 0;
        for (int sgi = 0; sgi < a.nseg; ++sgi) ksteps += (a.seg[sgi].c + 1) / 2;
        ksteps *= a.kh * a.kw;
        const char* esk = getenv("EEM_NO_SPLITK");                   // read per call: a test flips it inside one process
        const bool no_splitk = esk && esk[0] == '1';
        static const long splitk_max = [] { const char* e = getenv("EEM_SPLITK_MAX"); return e ? atol(e) : 512L; }();
        // deep single-cout-tile layers (E-RAFT's 256 -> 2 flow head) always split: one wave per 32 pixels would walk all of K
        const long plain_blocks = (long)ceil_div(hwo, 128) * cot * a.n;
        if (!no_splitk && ksteps >= 128 && (plain_blocks < splitk_max || (cot == 1 && ksteps >= 512 && plain_blocks < 4096))) {
            dim3 grid(ceil_div(hwo, 32), cot, a.n);
            // tiles that leave most of the chip empty: more waves per tile, fewer (latency-bound) batches per wave
            const long tiles = (long)grid.x * cot * a.n;
            const int batches = ksteps / 16;
            static const int force = [] { const char* e = getenv("EEM_SPLITK_WAVES"); return e ? atoi(e) : 0; }();
            int skw = tiles <= 64 && batches >= 32 ? 16 : tiles <= 160 && batches >= 16 ? 8 : 4;
            if (force) skw = force;
            if (skw == 16) hipLaunchKernelGGL((gconv_kernel<1, 1, 16>), grid, dim3(1024), 0, stream, a);
            else if (skw == 8) hipLaunchKernelGGL((gconv_kernel<1, 1, 8>), grid, dim3(512), 0, stream, a);
            else hipLaunchKernelGGL((gconv_kernel<1, 1, 4>), grid, dim3(256), 0, stream, a);
        } else {
            dim3 grid(ceil_div(hwo, 128), cot, a.n);
            hipLaunchKernelGGL((gconv_kernel<1, 1>), grid, dim3(256), 0, stream, a);
        }
    }
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}
