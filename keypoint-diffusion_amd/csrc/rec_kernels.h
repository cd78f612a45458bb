// Small kernels shared by the two keypoint receptor encoders (defined in rec_encoder.hip).
#pragma once
#include "engine.h"

namespace kpd {

// out[i] = i * scale, i < n
kpd_status launch_iota_scaled(int *out, int n, int scale, hipStream_t st);
// per-graph mean of node rows [n][S] -> [B][S]   (dgl.readout_nodes mean)
kpd_status launch_graph_mean(const float *s, const int *ptr, int B, int S, float *out, hipStream_t st);
// out[node][:] = Wt^T in[node][:], Wt = square weight stored transposed [in][out], S <= 256
kpd_status launch_linear_rows(const float *in, int n, int S, const float *Wt, float *out, hipStream_t st);
// attention-pooled keypoint positions: softmax over all receptor atoms of the keypoint's graph of <ft_src, ft_dst> / sqrt(S),
// exponentiated without max-subtraction as upstream; kp_x = sum_r softmax * rec_x[r]
kpd_status launch_kp_attention(const float *ft_src, const float *ft_dst, const float *rec_x, const int *rec_ptr, int n_kp, int K,
                               int S, float *kp_x, hipStream_t st);

}  // namespace kpd
