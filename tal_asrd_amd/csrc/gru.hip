// GRU cell of the UIS-RNN core network: CoreRNN (tal/diarization/uisrnn/uisrnn.py:20-39),
// called one observation at a time by the beam search (:412-414,438-440).
//   gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh  (two dense layers on the fp32 matrix cores)
//   r = s(gi_r + gh_r); z = s(gi_z + gh_z); n = tanh(gi_n + r * gh_n); h' = (1 - z) * n + z * h
// The gate math is one fused elementwise kernel (torch.nn.GRU gate order r, z, n).
#include "common.h"

namespace tal {

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
