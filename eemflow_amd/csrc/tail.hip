// Everything that runs on EEMFlow's 1/64-resolution grid (reference: model/EEMFlow/EEMFlow.py):
//   stage pooling (:144-154), 9x9 local correlation + tap select (:14-23,160), rconv_k and the
//   three decoders with grouped convs + channel shuffle (:37-69,96-102,161-176), out_conv (:104,180)
//   and the bilinear upsample back to the input size (:118-120,181).
// The grid is tiny (12x20 at 1280x720) so these kernels are latency-, not throughput-bound: each
// conv layer is one launch that covers all decoders/groups/samples ("jobs"), spreads 16x16 MFMA
// tiles over the chip and splits K over the four waves of a block.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------- pooling
struct PoolArgs {
    PoolJob job[3];
    int first_wave[4];      // prefix of output counts per job
    int njobs, nimg;
};

__global__ __launch_bounds__(256) void pool_kernel(PoolArgs a) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);      // one wave per pooled output value
    if (gw >= a.first_wave[a.njobs]) return;
    int ji = 0;
    while (ji + 1 < a.njobs && gw >= a.first_wave[ji + 1]) ++ji;
    const PoolJob jb = a.job[ji];
    const int oh = jb.h / jb.k, ow = jb.w / jb.k;
    int o = gw - a.first_wave[ji];
    const int ox = o % ow; o /= ow;
    const int oy = o % oh; o /= oh;                          // o = n*c + channel
    const float* src = jb.in + ((size_t)o * jb.h + (size_t)oy * jb.k) * jb.w + (size_t)ox * jb.k;
    const int kk = jb.k * jb.k;
    float s = 0.f;
    for (int i = lane; i < kk; i += 64) {
        const int ry = i / jb.k, rx = i - ry * jb.k;
        s += src[(size_t)ry * jb.w + rx];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) jb.out[((size_t)o * oh + oy) * ow + ox] = s / (float)kk;
}

struct PoolFinArgs {
    PoolFinJob job[3];
    int njobs, nimg, gh, gw;
};

__global__ __launch_bounds__(256) void pool_finalize_kernel(PoolFinArgs a) {
    int idx = blockIdx.x * 256 + threadIdx.x;
    const int g = a.gh * a.gw;
    for (int ji = 0; ji < a.njobs; ++ji) {
        const PoolFinJob jb = a.job[ji];
        const int total = a.nimg * jb.c * g;
        if (idx < total) {
            const int gx = idx % a.gw, gy = (idx / a.gw) % a.gh, nc = idx / g;
            const float* src = jb.partial + ((size_t)nc * jb.prows + (size_t)gy * jb.rows) * jb.pcols + gx;
            float s = 0.f;
            for (int i = 0; i < jb.rows; ++i) s += src[(size_t)i * jb.pcols];
            jb.out[idx] = s / (float)(jb.k * jb.k);
            return;
        }
        idx -= total;
    }
}

// ------------------------------------------------------------------------------- local correlation
struct CorrArgs {
    CorrJob job[3];
    int njobs, batch, h, w, ntaps;
    const int* taps;
};

// Four adjacent lanes share one output (tap, pixel) and split its channels (c = sub, sub+4, ...): the grid is tiny and
// L2-resident, so the kernel is one chain of dependent-latency loads per thread - a quarter of the channels per thread
// with all of them in flight cuts that chain (18 -> ~6 us inside the forward graph); the partial sums meet by DPP.
__global__ __launch_bounds__(256) void corr_kernel(CorrArgs a) {
    const int hw = a.h * a.w;
    const int per_job = a.batch * a.ntaps * hw;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    int idx = gid >> 2;
    const int sub = gid & 3;
    const bool live = idx < per_job * a.njobs;
    if (!live) idx = 0;                                   // keep the quad together for the cross-lane sum
    const int ji = idx / per_job;
    idx -= ji * per_job;
    const CorrJob jb = a.job[ji];
    const int p = idx % hw; idx /= hw;
    const int ti = idx % a.ntaps;
    const int b = idx / a.ntaps;
    const int y = p / a.w, x = p - y * a.w;
    const int tap = a.taps[ti];                 // dy-major index into the 9x9 window
    const int yy = y + tap / 9 - 4, xx = x + tap % 9 - 4;
    float s = 0.f;
    if (live && yy >= 0 && yy < a.h && xx >= 0 && xx < a.w) {
        const float* p1 = jb.f1 + (size_t)b * jb.c * hw + p;
        const float* p2 = jb.f2 + (size_t)b * jb.c * hw + yy * a.w + xx;
#pragma unroll 16
        for (int c = sub; c < jb.c; c += 4) s = fmaf(p1[(size_t)c * hw], p2[(size_t)c * hw], s);
    }
    s = dpp_add<0xB1>(s);                       // quad_perm [1,0,3,2]
    s = dpp_add<0x4E>(s);                       // quad_perm [2,3,0,1]
    if (live && sub == 0) jb.out[((size_t)b * jb.out_ctotal + ti) * hw + p] = s / (float)jb.c;
}

// The same correlation for large maps (EEMFlow+'s fine pyramid levels: 180x320 pixels): a block owns a 16x16-pixel tile,
// the f2 patch with its 4-pixel halo goes through LDS eight channels at a time (zeros outside the image), every thread keeps
// its pixel's 53 tap sums in registers and reads f1 once per channel - instead of re-reading f1 for every tap and spending
// four lanes per output.
template <int NT>
__global__ __launch_bounds__(256) void corr_tiled_kernel(CorrArgs a, int tiles_x) {
    constexpr int CK = 8, TS = 16, HS = TS + 8, PITCH = HS + 1;
    __shared__ float tile[CK][HS][PITCH];
    const int ji = blockIdx.z / a.batch, b = blockIdx.z % a.batch;
    const CorrJob jb = a.job[ji];
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int y0 = ty * TS, x0 = tx * TS;
    const int y = y0 + ly, x = x0 + lx;
    const bool inside = y < a.h && x < a.w;
    const int hw = a.h * a.w;
    int off[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { const int tap = a.taps[t]; off[t] = (tap / 9) * PITCH + (tap % 9); }   // window origin = (-4, -4)
    float s[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) s[t] = 0.f;
    const float* f1 = jb.f1 + (size_t)b * jb.c * hw;
    const float* f2 = jb.f2 + (size_t)b * jb.c * hw;
    const int p = inside ? y * a.w + x : 0;
    for (int c0 = 0; c0 < jb.c; c0 += CK) {
        __syncthreads();
        for (int e = threadIdx.x; e < CK * HS * HS; e += 256) {
            const int c = e / (HS * HS), r = e - c * (HS * HS);
            const int ry = r / HS, rx = r - ry * HS;
            const int gy = y0 - 4 + ry, gx = x0 - 4 + rx;
            const bool ok = c0 + c < jb.c && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
            tile[c][ry][rx] = ok ? f2[(size_t)(c0 + c) * hw + gy * a.w + gx] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CK; ++c) {
            const float v = (inside && c0 + c < jb.c) ? f1[(size_t)(c0 + c) * hw + p] : 0.f;
            const float* base = &tile[c][ly][lx];
#pragma unroll
            for (int t = 0; t < NT; ++t) s[t] = fmaf(v, base[off[t]], s[t]);
        }
    }
    if (!inside) return;
    const float inv = 1.f / (float)jb.c;
#pragma unroll
    for (int t = 0; t < NT; ++t) jb.out[((size_t)b * jb.out_ctotal + t) * hw + p] = s[t] * inv;
}

// ------------------------------------------------------------------------------- small-grid conv
// D[cout 16][pixel 16] tiles on v_mfma_f32_16x16x4_f32; pixels are the flattened (y*w+x) index.
// One wave per filter tap (9 waves for 3x3): a wave's K range is its tap x all channel groups, so the
// tap displacement and the bounds test are per-wave constants and ALL of its operand loads (<= MAXCG
// weight fragments + MAXCG input gathers) are issued before the first MFMA - the grid is tiny and
// L2-resident, so the kernel is one memory round trip + <= 25 MFMAs + a 9-way LDS reduction.
template <int KS, int MAXCG>
__global__ __launch_bounds__(64 * KS * KS) void tail_conv_kernel(TailConvLaunch L) {
    constexpr int KK = KS * KS;
    __shared__ f32x4 part[KK][64];
    const int lane = threadIdx.x & 63, t = threadIdx.x >> 6;
    const int ji = blockIdx.z / L.batch, b = blockIdx.z % L.batch;
    const TailConvJob jb = L.job[ji];
    const int cot = blockIdx.y;
    if (cot * 16 >= jb.cout) return;
    const int hw = L.h * L.w;
    const int j = lane & 15, g = lane >> 4;
    const int p = blockIdx.x * 16 + j;
    const bool pvalid = p < hw;
    const int y = p / L.w, x = p - y * L.w;
    const int yy = y + (KS == 3 ? t / 3 - 1 : 0), xx = x + (KS == 3 ? t % 3 - 1 : 0);
    const bool valid = pvalid && yy >= 0 && yy < L.h && xx >= 0 && xx < L.w;
    const int cg = (jb.cin + 3) >> 2;
    const size_t ioff = ((size_t)b * jb.in_ctotal + jb.in_coff) * hw + (valid ? yy * L.w + xx : 0);
    const float* in = jb.in + ioff;
    const float* gt = jb.gate ? jb.gate + ioff : nullptr;
    const int cmul = jb.in_cmul > 1 ? jb.in_cmul : 1;
    const float* wp = jb.wpk + ((size_t)cot * KK + t) * cg * 64 + lane;

    float av[MAXCG], bv[MAXCG];
#pragma unroll
    for (int q = 0; q < MAXCG; ++q) {
        const int qc = q < cg ? q : cg - 1;
        const int c = qc * 4 + g;
        const float a0 = wp[(size_t)qc * 64];
        const size_t coff = (size_t)(c < jb.cin ? c : jb.cin - 1) * cmul * hw;
        float b0 = in[coff];
        if (gt) b0 *= gt[coff] > 0.f ? 1.f : 0.1f;
        av[q] = q < cg ? a0 : 0.f;                       // weights of padded channels are packed as zeros
        bv[q] = (valid && c < jb.cin) ? b0 : 0.f;
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < MAXCG; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], acc, 0, 0, 0);

    if (KK > 1) {
        part[t][lane] = acc;
        __syncthreads();
        if (t != 0) return;
        acc = part[0][lane];
#pragma unroll
        for (int k = 1; k < KK; ++k) acc += part[k][lane];          // fixed order: bitwise repeatable
    }
    if (pvalid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = cot * 16 + g * 4 + r;
            if (co < jb.cout) {
                float v = acc[r] + (jb.bias ? jb.bias[co] : 0.f);
                if (jb.act) v = v > 0.f ? v : 0.1f * v;
                const size_t o = ((size_t)b * jb.out_ctotal + co * jb.out_cmul + jb.out_coff) * hw + p;
                if (jb.add) v += jb.add[o];                        // residual at the output's own index (EEMFlow+ decoders)
                jb.out[o] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------- bilinear resize
// F.interpolate(mode='bilinear', align_corners=False): src = max(scale*(dst+0.5)-0.5, 0)
__device__ __forceinline__ void src_index(float scale, int dst, int in_size, int& i0, int& i1, float& l1) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

__global__ void io_table_kernel(const void** table, const void* e1, const void* e2, void* out) {
    table[0] = e1; table[1] = e2; table[2] = out;
}

__global__ __launch_bounds__(256) void upsample_kernel(const float* __restrict__ in, float* __restrict__ out_arg,
                                                       int nc, int h, int w, int oh, int ow, const void* const* io) {
    float* __restrict__ out = io ? (float*)io[2] : out_arg;
    const int xq = ceil_div(ow, 4);
    const long total = (long)nc * oh * xq;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int q = idx % xq;
    const int oy = (idx / xq) % oh;
    const int c = idx / ((long)xq * oh);
    const float sy = (float)h / (float)oh, sx = (float)w / (float)ow;
    int y0, y1; float ly;
    src_index(sy, oy, h, y0, y1, ly);
    const float* r0 = in + ((size_t)c * h + y0) * w;
    const float* r1 = in + ((size_t)c * h + y1) * w;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ox = q * 4 + i;
        int x0, x1; float lx;
        src_index(sx, ox < ow ? ox : ow - 1, w, x0, x1, lx);
        const float top = (1.f - lx) * r0[x0] + lx * r0[x1];
        const float bot = (1.f - lx) * r1[x0] + lx * r1[x1];
        v[i] = (1.f - ly) * top + ly * bot;
    }
    float* dst = out + ((size_t)c * oh + oy) * ow + q * 4;
    if ((ow & 3) == 0) {
        *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (q * 4 + i < ow) dst[i] = v[i];
    }
}

}  // namespace

// ------------------------------------------------------------------------------- host side
int pool_launch(const PoolJob* jobs, int njobs, int nimg, hipStream_t stream) {
    EEM_REQUIRE(njobs >= 1 && njobs <= 3, "pool_launch: njobs=%d", njobs);
    PoolArgs a;
    a.njobs = njobs;
    a.nimg = nimg;
    int total = 0;
    for (int i = 0; i < njobs; ++i) {
        a.job[i] = jobs[i];
        a.first_wave[i] = total;
        total += nimg * jobs[i].c * (jobs[i].h / jobs[i].k) * (jobs[i].w / jobs[i].k);
    }
    for (int i = njobs; i < 4; ++i) a.first_wave[i] = total;
    if (total == 0) return EEM_OK;
    hipLaunchKernelGGL(pool_kernel, dim3(ceil_div(total, 4)), dim3(256), 0, stream, a);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int pool_finalize_launch(const PoolFinJob* jobs, int njobs, int nimg, int gh, int gw, hipStream_t stream) {
    EEM_REQUIRE(njobs >= 1 && njobs <= 3, "pool_finalize_launch: njobs=%d", njobs);
    PoolFinArgs a;
    a.njobs = njobs; a.nimg = nimg; a.gh = gh; a.gw = gw;
    int total = 0;
    for (int i = 0; i < njobs; ++i) { a.job[i] = jobs[i]; total += nimg * jobs[i].c * gh * gw; }
    if (total == 0) return EEM_OK;
    hipLaunchKernelGGL(pool_finalize_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, a);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int corr_launch(const CorrJob* jobs, int njobs, int batch, int h, int w, const int* taps_dev, int ntaps,
                hipStream_t stream) {
    EEM_REQUIRE(njobs >= 1 && njobs <= 3, "corr_launch: njobs=%d", njobs);
    CorrArgs a;
    for (int i = 0; i < njobs; ++i) a.job[i] = jobs[i];
    a.njobs = njobs; a.batch = batch; a.h = h; a.w = w; a.ntaps = ntaps; a.taps = taps_dev;
    const long total = (long)njobs * batch * ntaps * h * w;
    if (total == 0) return EEM_OK;
    static const bool plain = [] { const char* e = getenv("EEM_CORR_PLAIN"); return e && e[0] == '1'; }();
    if (!plain && ntaps == 53 && (long)h * w >= 32768) {                 // below: too few 16x16 tiles to fill the chip
        const int tiles_x = ceil_div(w, 16);
        hipLaunchKernelGGL((corr_tiled_kernel<53>), dim3(tiles_x * ceil_div(h, 16), 1, njobs * batch), dim3(256), 0, stream, a, tiles_x);
        EEM_HIP_CHECK(hipGetLastError());
        return EEM_OK;
    }
    hipLaunchKernelGGL(corr_kernel, dim3((unsigned)((total * 4 + 255) / 256)), dim3(256), 0, stream, a);   // 4 lanes per output
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

size_t tail_packed_floats(int cin, int cout, int ksize) {
    return (size_t)ceil_div(cout, 16) * ksize * ksize * ceil_div(cin, 4) * 64;
}

void tail_pack_weights(const float* w, int cin, int cout, int ksize, float* packed) {
    const int cg = ceil_div(cin, 4), kk = ksize * ksize, ksteps = kk * cg;
    for (int cot = 0; cot < ceil_div(cout, 16); ++cot)
        for (int s = 0; s < ksteps; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int co = cot * 16 + (lane & 15);
                const int t = s / cg, c = (s % cg) * 4 + (lane >> 4);
                float v = 0.f;
                if (co < cout && c < cin) v = w[((size_t)co * cin + c) * kk + t];
                packed[((size_t)cot * ksteps + s) * 64 + lane] = v;
            }
}

template <int KS, int MAXCG>
static void tail_launch_t(const TailConvLaunch& l, dim3 grid, hipStream_t stream) {
    hipLaunchKernelGGL((tail_conv_kernel<KS, MAXCG>), grid, dim3(64 * KS * KS), 0, stream, l);
}

int tail_conv_launch(const TailConvLaunch& l, hipStream_t stream) {
    EEM_REQUIRE(l.njobs >= 1 && l.njobs <= TAIL_MAX_JOBS, "tail_conv_launch: njobs=%d", l.njobs);
    EEM_REQUIRE(l.ksize == 3 || l.ksize == 1, "tail_conv_launch: ksize=%d", l.ksize);
    int max_cout = 0, max_cg = 0;
    for (int i = 0; i < l.njobs; ++i) {
        max_cout = l.job[i].cout > max_cout ? l.job[i].cout : max_cout;
        const int cg = ceil_div(l.job[i].cin, 4);
        max_cg = cg > max_cg ? cg : max_cg;
    }
    EEM_REQUIRE(max_cg <= 25, "tail_conv_launch: cin > 100 is not built");
    dim3 grid(ceil_div(l.h * l.w, 16), ceil_div(max_cout, 16), l.njobs * l.batch);
    if (l.ksize == 1) {
        if (max_cg <= 2) tail_launch_t<1, 2>(l, grid, stream);
        else tail_launch_t<1, 25>(l, grid, stream);
    } else if (max_cg <= 5) tail_launch_t<3, 5>(l, grid, stream);
    else if (max_cg <= 8) tail_launch_t<3, 8>(l, grid, stream);
    else if (max_cg <= 16) tail_launch_t<3, 16>(l, grid, stream);
    else if (max_cg <= 18) tail_launch_t<3, 18>(l, grid, stream);
    else tail_launch_t<3, 25>(l, grid, stream);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int upsample_launch(const float* in, float* out, int nc, int h, int w, int oh, int ow, hipStream_t stream, const void* const* io) {
    const long total = (long)nc * oh * ceil_div(ow, 4);
    if (total == 0) return EEM_OK;
    hipLaunchKernelGGL(upsample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, in, out, nc,
                       h, w, oh, ow, io);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}

int io_table_launch(const void** table, const void* e1, const void* e2, void* out, hipStream_t stream) {
    hipLaunchKernelGGL(io_table_kernel, dim3(1), dim3(1), 0, stream, table, e1, e2, out);
    EEM_HIP_CHECK(hipGetLastError());
    return EEM_OK;
}
