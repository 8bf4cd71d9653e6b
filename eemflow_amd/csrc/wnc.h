// Winograd F(2x2,3x3) for EEMFlow+'s mid-size 3x3 layers (conv_wnc.hip): any input depth >= 32 -> slices of 32 output channels.
#pragma once
#include "common.h"

constexpr int WNC_MAX_JOBS = 12;     // (EEMFlow's decoder: 3 decoders x 4 slices of 100 couts in one launch)
constexpr int WNC_MAX_CHUNKS = 8;

// One job = 32 output channels (a group of a grouped conv, or a 32-channel slice of a wider layer) over the whole batch.
struct WncJob {
    const float* in;       // [n][in_ctotal][h][w]; the job reads channels in_coff + chunk_off[k] .. + 32 for every chunk k
    int in_ctotal, in_coff;
    const float* w;        // wnc_pack's stream for this job
    const float* bias;     // 32 floats readable (16 in the m16 form); values beyond cout are never stored
    float* out;            // [n][out_ctotal][h][w]; cout co goes to channel out_coff + co * out_cmul
    int out_ctotal, out_coff, out_cmul, cout;
    const float* res;      // optional residual, indexed like `out`: out = relu(res + act(conv + bias)) (model/extractor.py:48-57)
};

struct WncArgs {
    WncJob job[WNC_MAX_JOBS];
    int njobs, nchunks;
    int cin;               // input depth of every job: chunk k starts at channel min(32 k, cin - 32) of the job's range (wnc_chunks)
    int chunk_off[WNC_MAX_CHUNKS];
    int n, h, w;
    int act;               // 0: none, 1: LeakyReLU(0.1), 2: ReLU
    int m16;               // the streams are the 16-cout form's (wnc_pack(.., m16 = 1)): jobs of at most 16 couts on the 16x16x4 MFMA
    const float* zero_page;   // >= 16 bytes of zeros (out-of-image pieces of the staged tile)
    float* trash;             // >= 512 floats nobody reads (stores of lanes outside the image / beyond cout)
};

// chunks of 32 input channels that cover [0, cin): 0, 32, ... and, when cin % 32 != 0, a last one at cin - 32 whose first
// 32 - cin % 32 channels repeat the chunk before it (their weights are packed as zeros).  Returns the count (cin >= 32, <= 256).
int wnc_chunks(int cin, int* chunk_off);
// floats of one job's stream
size_t wnc_packed_floats(int cin, int m16);
// U = G g G^T of the 32 (m16: 16) output channels from co0 on of w [cout][cin][3][3] in the kernel's fragment order (zeros beyond cout)
void wnc_pack(const float* w, int cout, int cin, int co0, int m16, float* packed);
bool wnc_supported(const WncArgs& a);
int wnc_launch(const WncArgs& a, hipStream_t st);

// The same packing on the device, for weights that change there (a training step, eemflow_update_weights): job j writes wnc_pack(w, cout,
// cin, co0, m16)'s stream to `packed` - the very expression, shared with the host function - and the slice's 32 (m16: 16) biases, zero
// beyond cout, to `bias_out`.  One launch for all jobs.
constexpr int WNC_PACK_MAX_JOBS = 24;
struct WncPackJob {
    const float* w;        // [cout][cin][3][3]
    const float* bias;     // [cout]
    float* packed;
    float* bias_out;
    int cout, cin, co0, m16;
};
struct WncPackArgs {
    WncPackJob job[WNC_PACK_MAX_JOBS];
    int njobs;
};
int wnc_pack_device_launch(const WncPackArgs& a, hipStream_t st);

