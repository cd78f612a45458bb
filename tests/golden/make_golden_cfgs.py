"""Configurations of the composed golden fixtures (shared by the generator and the tests)."""

GVP_CFGS = {
    'gvp_kp': dict(vector_size=16, n_convs=3, n_hidden_scalars=256, message_norm=10.0, update_kp=True, ll_k=0, kl_k=7,
                   n_message_gvps=3, n_update_gvps=2, n_noise_gvps=4, dropout=0.1),
    'gvp_mean': dict(vector_size=16, n_convs=2, n_hidden_scalars=128, message_norm='mean', update_kp=True, ll_k=0,
                     kl_k=5, n_message_gvps=3, n_update_gvps=2, n_noise_gvps=4, dropout=0.1),
    'gvp_norm0': dict(vector_size=16, n_convs=2, n_hidden_scalars=128, message_norm=0, update_kp=True, ll_k=0,
                      kl_k=5, n_message_gvps=2, n_update_gvps=1, n_noise_gvps=3, dropout=0.0),
}

RECENC_CFGS = {'recenc_mean': dict(in_scalar_size=10, out_scalar_size=128, n_message_gvps=3, n_update_gvps=2, vector_size=16,
                        n_rr_convs=2, n_rk_convs=2, message_norm='mean', k_closest=5, kp_rad=0, dropout=0.1,
                        n_keypoints=6),
    'recenc_norm10': dict(in_scalar_size=10, out_scalar_size=128, n_message_gvps=1, n_update_gvps=1, vector_size=16,
                          n_rr_convs=3, n_rk_convs=2, message_norm=10.0, k_closest=4, kp_rad=0, dropout=0.0,
                          n_keypoints=5)}

# EGNN keypoint receptor encoder (models/receptor_encoder.py): the shipped egnn_20kp shape and two variants
RECEGNN_CFGS = {
    'recegnn_20kp': dict(n_convs=4, n_keypoints=20, in_n_node_feat=10, use_sameres_feat=True, hidden_n_node_feat=128,
                         out_n_node_feat=128, use_tanh=True, coords_range=10, message_norm=0.0, kp_rad=0.0, k_closest=5,
                         norm=True, fix_pos=False, n_kk_convs=0),
    'recegnn_small': dict(n_convs=2, n_keypoints=7, in_n_node_feat=10, use_sameres_feat=False, hidden_n_node_feat=64,
                          out_n_node_feat=32, use_tanh=False, coords_range=10, message_norm=5.0, kp_rad=0.0, k_closest=3,
                          norm=False, fix_pos=False, n_kk_convs=0),
    'recegnn_fixpos': dict(n_convs=1, n_keypoints=4, in_n_node_feat=10, use_sameres_feat=True, hidden_n_node_feat=48,
                           out_n_node_feat=48, use_tanh=True, coords_range=10, message_norm=0.0, kp_rad=0.0, k_closest=4,
                           norm=True, fix_pos=True, n_kk_convs=0),
}


def same_res_feature(src, dst):
    """Synthetic rr `same_res` edge feature of the fixtures ([E,1] float, as the dataset stores a bool column)."""
    return ((src + 2 * dst) % 3 == 0).float().view(-1, 1)
