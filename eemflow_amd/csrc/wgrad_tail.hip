// Weight + bias gradients of ALL the 3x3 convs of the 1/64-grid tail (EEMFlow.py:37-69,96-102: three decoders x {69 -> 100, three grouped
// 100 -> 100 layers of five 20 -> 20 groups, 100 -> 64, 64 -> 32, 32 -> 2} and the three rconv_k) as ONE launch at the end of the tail's
// backward chain (round 6).  Before: one wgrad_small_kernel + one bias launch per layer on the weight-gradient stream - 8 + 9 launches,
// 460 us of that stream per step at 346x260 batch 32 for 1.15 GFLOP (8.5 us of the fp32 matrix pipe), every block holding a whole CU's
// LDS; they delayed the encoder's weight gradients and cost the encoder's data gradients a third of the chip while they ran.
//
// The maps are tiny (5 x 6 .. 12 x 20 pixels per image) and the contraction short (K = batch x pixels = 960 .. 1 920), so the split is over
// the OUTPUT: a block = (conv, 16 input channels, K part of <= 256 pixels = whole images); its four waves take the conv's 16-cout tiles in
// turn (wave w: tiles w, w + 4), nine 16-column (ci, tap) tiles each on v_mfma_f32_16x16x4_f32.  The block's operands are staged once:
// the haloed X patch [16 ci][images][h + 2][w + 2] (zeros in the halo: no bounds test in the k-loop), G of all couts with the
// LeakyReLU' gate of the conv's output folded in ([co][pixels], odd pitch).  K parts meet in dW through fp32 atomics (<= 8 adders per
// weight); the bias gradient rides in the blocks of input chunk 0.  The job table (61 convs at five groups) lives in device memory and
// is uploaded when it changes (pointers and shapes are those of the context's workspace: once per shape).
#include <string.h>

#include <vector>

#include "common.h"
#include "train.h"

namespace {

struct TwJob {
    const float* x;
    const float* g;
    const float* gate;
    float* dw;
    float* db;
    int x_ctotal, x_coff, cin;
    int g_ctotal, g_coff, g_cmul, cout;
    int dw_cin, dw_coff;
};
struct TwBlock { short job, chunk, kpart, pad; };

constexpr int TW_MAXPX = 256;                        // most pixels of a K part (one image at least)
constexpr int TW_AIMPX = 128;                        // ... and what a part aims at: two blocks per CU cover each other's staging

__global__ __launch_bounds__(256) void wgrad_tail_kernel(const TwJob* __restrict__ jobs, const TwBlock* __restrict__ blocks, int n, int h, int w,
                                                         int ipt, int gp, int dbg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const TwBlock bk = blocks[blockIdx.x];
    const TwJob jb = jobs[bk.job];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int hw = h * w, XC = w + 2, PL = (h + 2) * XC;
    const int n0 = bk.kpart * ipt;
    const int nimg = min(ipt, n - n0);
    const int npx = nimg * hw;                           // pixels of this K part
    const int ksteps = (npx + 3) >> 2;
    const int ci0 = bk.chunk * 16, cin_here = min(16, jb.cin - ci0);
    const int MT = (jb.cout + 15) >> 4;
    float* Gs = lds;                                      // [4 waves][16][gp]: a wave's cout tile of the current turn
    float* Xs = lds + 64 * gp;                            // [16][ipt][PL]
    int* xtab = reinterpret_cast<int*>(Xs + 16 * ipt * PL);      // pixel -> offset of its (ky = 0, kx = 0) tap inside a channel's patch

    // two small tables first (a division per ENTRY, none per staged element): pixel -> offsets, patch position -> source offset
    int* gtab = xtab + 4 * ksteps;                        // pixel -> offset of the pixel inside G's (image, channel) planes, or -1
    int* ptab = gtab + 4 * ksteps;                        // patch position -> offset inside an input plane, or -1 (the zero halo)
    for (int p = threadIdx.x; p < 4 * ksteps; p += 256) {
        const int img = p / hw, q = p - img * hw;
        const int y = q / w, x = q - y * w;
        xtab[p] = p < npx ? img * PL + y * XC + x : 0;     // (pixels past the part carry G = 0)
        gtab[p] = p < npx ? img * jb.g_ctotal * hw + q : -1;
    }
    for (int r2 = threadIdx.x; r2 < PL; r2 += 256) {
        const int ry = r2 / XC, rx = r2 - ry * XC;
        const int iy = ry - 1, ix = rx - 1;
        ptab[r2] = (iy >= 0 && iy < h && ix >= 0 && ix < w) ? iy * w + ix : -1;
    }
    __syncthreads();
    const float* gbase = jb.g + ((size_t)n0 * jb.g_ctotal + jb.g_coff) * hw;
    const float* tbase = jb.gate ? jb.gate + ((size_t)n0 * jb.g_ctotal + jb.g_coff) * hw : nullptr;
    {
        // X patches, a (channel, image) pair per wave and turn - sixteen pairs' loads in flight together (one load per turn is one
        // memory round trip per pair)
        const float* xbase = jb.x + ((size_t)n0 * jb.x_ctotal + jb.x_coff + ci0) * hw;
        const int npairs = 16 * ipt;
        for (int r2 = lane; r2 < PL; r2 += 64) {
            const int so = ptab[r2];
            const int sc = so >= 0 ? so : 0;
            for (int p0 = wave; p0 < npairs; p0 += 64) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int pr = min(p0 + 4 * u, npairs - 1);
                    const int ci = pr / ipt, img = pr - ci * ipt;
                    v[u] = xbase[((size_t)min(img, nimg - 1) * jb.x_ctotal + min(ci, cin_here - 1)) * hw + sc];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int pr = p0 + 4 * u;
                    const int ci = pr / ipt, img = pr - ci * ipt;
                    if (pr < npairs) Xs[pr * PL + r2] = (so >= 0 && ci < cin_here && img < nimg) ? v[u] : 0.f;
                }
            }
        }
    }
    __syncthreads();

    int boff[9];
    const int nvalid = cin_here * 9;
    const int ntiles = (nvalid + 15) >> 4;
#pragma unroll
    for (int nt = 0; nt < 9; ++nt) {
        int nn = nt * 16 + j;
        nn = nn < nvalid ? nn : 0;
        const int ci = nn / 9, tap = nn - ci * 9;
        boff[nt] = ci * ipt * PL + (tap / 3) * XC + (tap % 3);
    }
    const int dwcin = jb.dw_cin ? jb.dw_cin : jb.cin;
    for (int mt = wave; mt < MT; mt += 4) {
        f32x4 acc[9];
#pragma unroll
        for (int nt = 0; nt < 9; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        float bsum = 0.f;
        // this wave's 16 rows of G (gate folded in) into its OWN slab of LDS: all 16 x 2 loads of a lane in flight together, no block
        // barrier (the block-wide staging of all couts, eight rows per turn, was eight dependent memory round trips: 16 us of a block's 38)
        float* Gw = Gs + wave * 16 * gp;
        for (int p = lane; p < ((dbg & 4) ? 0 : 4 * ksteps); p += 64) {
            const int go = gtab[p];
            const size_t po = go >= 0 ? go : 0;
            float v[16], t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int co = min(mt * 16 + u, jb.cout - 1);
                const size_t o = (size_t)co * jb.g_cmul * hw + po;
                v[u] = gbase[o];
                t[u] = tbase ? tbase[o] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                Gw[u * gp + p] = (go >= 0 && mt * 16 + u < jb.cout) ? v[u] * (t[u] > 0.f ? 1.f : 0.1f) : 0.f;
        }
        const float* ga = Gw + j * gp + g;
#pragma unroll 2
        for (int s = 0; s < ((dbg & 2) ? 0 : ksteps); ++s) {
            const float av = ga[4 * s];
            const int xo = xtab[4 * s + g];
            bsum += av;
#pragma unroll
            for (int nt = 0; nt < 9; ++nt)                    // (a 4-channel chunk of a 20-channel group fills 3 of the 9 column tiles)
                if (nt < ntiles) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Xs[boff[nt] + xo], acc[nt], 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < 9; ++nt) {
            const int nn = nt * 16 + j;
            if (nn >= nvalid || (dbg & 1)) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {                     // D[co = 4 g + r][n = j]
                const int co = mt * 16 + 4 * g + r;
                if (co < jb.cout) atomicAdd(&jb.dw[((size_t)co * dwcin + jb.dw_coff + ci0) * 9 + nn], acc[nt][r]);
            }
        }
        if (jb.db && bk.chunk == 0) {
            bsum += __shfl_xor(bsum, 16);
            bsum += __shfl_xor(bsum, 32);
            const int co = mt * 16 + j;
            if (g == 0 && co < jb.cout) atomicAdd(&jb.db[co], bsum);
        }
    }
}

struct TailCache {
    std::vector<TwJob> jobs;
    std::vector<TwBlock> blocks;
    TwJob* djobs = nullptr;
    TwBlock* dblocks = nullptr;
    size_t cap_jobs = 0, cap_blocks = 0;
    int dev = -1;
};
thread_local TailCache g_tail_cache[4];              // a few contexts per thread (keyed by the first job's dw pointer)

}  // namespace

bool wgrad_tail_supported(const WgradArgs* jobs, int njobs, int n, int h, int w) {
    const char* e = getenv("EEM_NO_WGRAD_TAIL");                      // read per call: a test flips it inside one process
    if (e && e[0] == '1') return false;
    if (njobs < 1 || njobs > 30000 || h * w > TW_MAXPX || n < 1) return false;
    for (int i = 0; i < njobs; ++i) {
        const WgradArgs& a = jobs[i];
        if (a.k != 3 || a.kh != 0 || a.stride != 1 || a.pad != 1 || a.hin != h || a.win != w || a.hout != h || a.wout != w || a.n != n ||
            a.cout < 1 || a.cout > 128 || a.cin < 1 || a.nxseg != 0)
            return false;
    }
    return true;
}

// all jobs: 3x3, stride 1, pad 1, maps of h x w <= 256 pixels, n images; dw / db (WgradArgs::db, may be NULL) zeroed by the caller
int wgrad_tail_launch(const WgradArgs* jobs, int njobs, int n, int h, int w, hipStream_t st) {
    const int hw = h * w;
    int ipt = TW_AIMPX / hw;
    if (ipt < 1) ipt = 1;
    if (ipt > n) ipt = n;
    const int nk = ceil_div(n, ipt);
    int dev = 0;
    EEM_HIP_CHECK(hipGetDevice(&dev));
    TailCache* tc = nullptr;
    for (TailCache& t : g_tail_cache)
        if (t.dev == dev && !t.jobs.empty() && t.jobs[0].dw == jobs[0].dw) tc = &t;
    if (!tc) {
        static thread_local int next = 0;
        tc = &g_tail_cache[next++ & 3];
    }
    std::vector<TwJob> hj(njobs);
    std::vector<TwBlock> hb;
    int mtmax = 1;
    for (int i = 0; i < njobs; ++i) {
        const WgradArgs& a = jobs[i];
        hj[i] = TwJob{a.x, a.g, a.gate, a.dw, a.db, a.x_ctotal, a.x_coff, a.cin, a.g_ctotal, a.g_coff,
                      a.g_cmul ? a.g_cmul : 1, a.cout, a.dw_cin, a.dw_coff};
        mtmax = std::max(mtmax, ceil_div(a.cout, 16));
        for (int ch = 0; ch < ceil_div(a.cin, 16); ++ch)
            for (int kp = 0; kp < nk; ++kp) hb.push_back(TwBlock{(short)i, (short)ch, (short)kp, 0});
    }
    const bool same = tc->dev == dev && tc->jobs.size() == hj.size() && tc->blocks.size() == hb.size() &&
                      memcmp(tc->jobs.data(), hj.data(), hj.size() * sizeof(TwJob)) == 0 &&
                      memcmp(tc->blocks.data(), hb.data(), hb.size() * sizeof(TwBlock)) == 0;
    if (!same) {
        if (tc->dev != dev || tc->cap_jobs < hj.size() || tc->cap_blocks < hb.size()) {
            if (tc->djobs && tc->dev == dev) { (void)hipFree(tc->djobs); (void)hipFree(tc->dblocks); }
            tc->cap_jobs = hj.size() + 16; tc->cap_blocks = hb.size() + 256;
            EEM_HIP_CHECK(hipMalloc((void**)&tc->djobs, tc->cap_jobs * sizeof(TwJob)));
            EEM_HIP_CHECK(hipMalloc((void**)&tc->dblocks, tc->cap_blocks * sizeof(TwBlock)));
            tc->dev = dev;
        }
        EEM_HIP_CHECK(hipStreamSynchronize(st));                          // (a launch of the previous table may still read it)
        EEM_HIP_CHECK(hipMemcpy(tc->djobs, hj.data(), hj.size() * sizeof(TwJob), hipMemcpyHostToDevice));
        EEM_HIP_CHECK(hipMemcpy(tc->dblocks, hb.data(), hb.size() * sizeof(TwBlock), hipMemcpyHostToDevice));
        tc->jobs = hj;
        tc->blocks = hb;
    }
    const int kmax = 4 * ceil_div(ipt * hw, 4);
    const int gp = kmax | 1;                                               // odd pitch: the A-operand column reads spread over the banks
    const int PL = (h + 2) * (w + 2);
    const size_t lds_bytes = ((size_t)64 * gp + 16 * ipt * PL + 2 * kmax + PL) * 4;
    EEM_REQUIRE(lds_bytes <= 160 * 1024, "wgrad_tail: %zu bytes of LDS", lds_bytes);
    static bool raised = false;
    if (!raised) {
        EEM_HIP_CHECK(hipFuncSetAttribute((const void*)wgrad_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised = true;
    }
    static const int dbg = [] { const char* e = getenv("EEM_TW_DBG"); return e ? atoi(e) : 0; }();     // measurement: 1 no atomics, 2 no k-loop, 4 no G staging
    hipLaunchKernelGGL(wgrad_tail_kernel, dim3((unsigned)hb.size()), dim3(256), lds_bytes, st, tc->djobs, tc->dblocks, n, h, w, ipt, gp, dbg);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}
