// GRU cell of the UIS-RNN core network: CoreRNN (tal/diarization/uisrnn/uisrnn.py:20-39),
// called one observation at a time by the beam search (:412-414,438-440).
//   gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh  (two dense layers on the fp32 matrix cores)
//   r = s(gi_r + gh_r); z = s(gi_z + gh_z); n = tanh(gi_n + r * gh_n); h' = (1 - z) * n + z * h
// The gate math is one fused elementwise kernel (torch.nn.GRU gate order r, z, n).
//
// Round 6: the whole cell as ONE launch for the shapes the beam search runs it on (a few dozen rows: one per surviving beam).
// The call is made once per observation of an episode (uisrnn.py:412-414) and was three dependent launches of ~6 us for 4.7 MB of
// weights.  gru_step_kernel: a workgroup owns FOUR hidden units j and 16 rows; its MFMA tile (v_mfma_f32_16x16x4_f32, weights as the
// A operand) has the 16 "columns" (4 units) x (r, z, n_x, n_h) over the concatenated K = In + H axis --
//     r, z:  [W_i | W_h] . [x | h]        n_x: [W_in | 0] . [x | h]        n_h: [0 | W_hn] . [x | h]
// (n needs its two halves apart: n = tanh(gi_n + r * gh_n)), so a lane ends with exactly the four sums of ONE (unit, row) pair and the
// gate math runs on registers.  Four waves split K in 16-wide chunks; every weight and activation fragment of a wave is a 16-byte
// load issued before its first MFMA (one memory round trip per launch, as the decoder's skinny dense layers); the four partial tiles
// are added in wave order through LDS.  W_ih, W_hh are streamed once per 16 rows; lanes of the zero half of the n columns load nothing.
#include "common.h"

namespace tal {

typedef float gru_f32x4 __attribute__((ext_vector_type(4)));

template <int NCH>      // 16-wide K chunks a wave holds in flight at once (NCH x 2 x 4 registers)
__global__ __launch_bounds__(256) void gru_step_kernel(const float* __restrict__ x, const float* __restrict__ h, const float* __restrict__ w_ih,
                                                      const float* __restrict__ w_hh, const float* __restrict__ b_ih,
                                                      const float* __restrict__ b_hh, float* __restrict__ h_out, int B, int In, int H) {
    __shared__ gru_f32x4 part[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // as an operand supplier: tile index (lane & 15) = weight column 4 * unit + gate, or row of the batch; k quarter (lane >> 4)
    const int col = lane & 15, g = col & 3, kq = lane >> 4;
    const int j = blockIdx.x * 4 + (col >> 2);
    const int64_t b = (int64_t)blockIdx.y * 16 + col;
    const int64_t bc = b < B ? b : B - 1;
    const int nch_x = In >> 4, nch = (In + H) >> 4;
    const float* wx = g < 3 ? w_ih + ((int64_t)g * H + j) * In : nullptr;                        // r, z, n_x read W_ih
    const float* wh = g != 2 ? w_hh + ((int64_t)(g == 3 ? 2 : g) * H + j) * H : nullptr;         // r, z, n_h read W_hh
    const float* ax = x + bc * In;
    const float* ah = h + bc * H;
    gru_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const gru_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = w; c0 < nch; c0 += 4 * NCH) {
        gru_f32x4 wv[NCH], av[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = c0 + 4 * i;
            wv[i] = zero;
            av[i] = zero;
            if (c < nch) {
                const bool xp = c < nch_x;
                const int k = (xp ? c : c - nch_x) * 16 + 4 * kq;
                const float* wr = xp ? wx : wh;
                if (wr) wv[i] = *reinterpret_cast<const gru_f32x4*>(wr + k);
                av[i] = *reinterpret_cast<const gru_f32x4*>((xp ? ax : ah) + k);
            }
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i][s], av[i][s], acc, 0, 0, 0);
    }
    part[w][lane] = acc;
    __syncthreads();
    if (w != 0) return;
    // D[i = 4 (lane >> 4) + e][n = lane & 15]: the four gate sums e of unit (lane >> 4) for row (lane & 15), waves added in order
    gru_f32x4 t = part[0][lane];
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        const gru_f32x4 p = part[q][lane];
        t[0] += p[0]; t[1] += p[1]; t[2] += p[2]; t[3] += p[3];
    }
    const int ju = blockIdx.x * 4 + (lane >> 4);
    if (b >= B) return;
    const float r = 1.f / (1.f + expf(-((t[0] + b_ih[ju]) + b_hh[ju])));
    const float z = 1.f / (1.f + expf(-((t[1] + b_ih[H + ju]) + b_hh[H + ju])));
    const float n = tanhf((t[2] + b_ih[2 * H + ju]) + r * (t[3] + b_hh[2 * H + ju]));
    h_out[b * H + ju] = (1.f - z) * n + z * h[b * H + ju];
}

__global__ __launch_bounds__(256) void gru_gate_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                      const float* __restrict__ h, float* __restrict__ h_out, int B,
                                                      int H) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, j = i - b * H;
    const float* gib = gi + (int64_t)b * 3 * H;
    const float* ghb = gh + (int64_t)b * 3 * H;
    const float r = 1.f / (1.f + expf(-(gib[j] + ghb[j])));
    const float z = 1.f / (1.f + expf(-(gib[H + j] + ghb[H + j])));
    const float n = tanhf(gib[2 * H + j] + r * ghb[2 * H + j]);
    h_out[i] = (1.f - z) * n + z * h[i];
}

}  // namespace tal

using namespace tal;

extern "C" size_t tal_gru_cell_workspace_bytes(int B, int H) { return (size_t)(B > 0 ? B : 1) * 6 * H * sizeof(float); }

extern "C" int tal_gru_cell_fwd(const float* x, const float* h, int B, int In, int H, const float* w_ih,
                                const float* w_hh, const float* b_ih, const float* b_hh, float* h_out, void* workspace,
                                size_t workspace_bytes, void* stream) {
    TAL_CHECK_ARG(x && h && w_ih && w_hh && b_ih && b_hh && h_out && workspace, "tal_gru_cell_fwd: null pointer");
    TAL_CHECK_ARG(B > 0 && In > 0 && H > 0 && In % 4 == 0 && H % 4 == 0, "tal_gru_cell_fwd: bad shape B=%d In=%d H=%d", B, In, H);
    TAL_CHECK_ARG(h != h_out, "tal_gru_cell_fwd: in-place update not supported");
    if (workspace_bytes < tal_gru_cell_workspace_bytes(B, H)) {
        set_error("tal_gru_cell_fwd: workspace %zu < %zu bytes", workspace_bytes, tal_gru_cell_workspace_bytes(B, H));
        return TAL_ENOMEM;
    }
    hipStream_t s = (hipStream_t)stream;
    // the one-launch step: the beam search's shapes (rows = surviving beams); more rows re-read the weights once per 16
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!opt(OPT_GRU_UNFUSED) && B <= 256 && In % 16 == 0 && H % 16 == 0 && al16(x) && al16(h) && al16(w_ih) && al16(w_hh)) {
        hipLaunchKernelGGL((gru_step_kernel<16>), dim3((unsigned)(H / 4), (unsigned)cdiv(B, 16)), dim3(256), 0, s, x, h, w_ih, w_hh, b_ih,
                           b_hh, h_out, B, In, H);
        TAL_CHECK_LAUNCH("tal_gru_cell_fwd (one launch)");
        return TAL_OK;
    }
    float* gi = reinterpret_cast<float*>(workspace);
    float* gh = gi + (size_t)B * 3 * H;
    int rc = launch_linear(x, w_ih, b_ih, nullptr, 0.f, 0, B, 3 * H, In, gi, s);
    if (rc) return rc;
    rc = launch_linear(h, w_hh, b_hh, nullptr, 0.f, 0, B, 3 * H, H, gh, s);
    if (rc) return rc;
    hipLaunchKernelGGL(gru_gate_kernel, dim3((unsigned)cdiv((int64_t)B * H, 256)), dim3(256), 0, s, gi, gh, h, h_out, B,
                       H);
    TAL_CHECK_LAUNCH("tal_gru_cell_fwd");
    return TAL_OK;
}
