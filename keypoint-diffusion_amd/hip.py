"""ctypes binding of libkpd_hip.so (C ABI in include/kpd.h).

The library is the product: there is no PyTorch / CPU fallback.  If the shared object has
not been built (`python __graft_entry__.py` or `make -C keypoint-diffusion_amd/csrc`) every
entry point raises.
"""
import ctypes as C
import os
from typing import Dict, Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# KPD_LIB: another build of the same library (profiles/tools load the TOOLS build, csrc/tools_build/libkpd_hip.so, this way);
# bench.py refuses to run with it set and checks kpd_build_flags() == 0
LIB_PATH = os.environ.get('KPD_LIB') or os.path.join(_HERE, 'csrc', 'libkpd_hip.so')
_lib = None

c_int_p = C.POINTER(C.c_int32)
c_float_p = C.POINTER(C.c_float)


class KpdBatch(C.Structure):
    _fields_ = [('B', C.c_int32), ('n_lig', C.c_int32), ('n_kp', C.c_int32), ('max_lig', C.c_int32),
                ('max_kp', C.c_int32), ('lig_ptr', C.c_void_p), ('kp_ptr', C.c_void_p), ('lig_x', C.c_void_p),
                ('lig_h', C.c_void_p), ('kp_x', C.c_void_p), ('kp_h', C.c_void_p), ('kp_v', C.c_void_p),
                ('n_kk', C.c_int32), ('kk_src', C.c_void_p), ('kk_dst', C.c_void_p), ('kk_rowptr', C.c_void_p)]


class KpdLigGraph(C.Structure):
    _fields_ = [('cap_ll', C.c_int32), ('cap_kl', C.c_int32),
                ('ll_src', C.c_void_p), ('ll_dst', C.c_void_p), ('ll_rowptr', C.c_void_p),
                ('kl_src', C.c_void_p), ('kl_dst', C.c_void_p), ('kl_rowptr', C.c_void_p),
                ('lk_src', C.c_void_p), ('lk_dst', C.c_void_p), ('lk_rowptr', C.c_void_p),
                ('ll_per_graph', C.c_void_p), ('counts', C.c_void_p)]


class KpdEgnnConfig(C.Structure):
    _fields_ = [('atom_nf', C.c_int32), ('rec_nf', C.c_int32), ('n_layers', C.c_int32), ('hidden_nf', C.c_int32),
                ('use_tanh', C.c_int32), ('norm', C.c_int32), ('update_kp_feat', C.c_int32),
                ('message_norm', C.c_float), ('ll_k', C.c_int32), ('kl_k', C.c_int32),
                ('ll_cutoff', C.c_float), ('kl_cutoff', C.c_float), ('coords_range', C.c_float)]


class KpdGvpConfig(C.Structure):
    _fields_ = [('n_lig_scalars', C.c_int32), ('n_kp_scalars', C.c_int32), ('vector_size', C.c_int32),
                ('n_convs', C.c_int32), ('n_hidden_scalars', C.c_int32), ('update_kp', C.c_int32),
                ('message_norm_mode', C.c_int32), ('message_norm', C.c_float), ('ll_k', C.c_int32), ('kl_k', C.c_int32),
                ('ll_cutoff', C.c_float), ('kl_cutoff', C.c_float), ('n_message_gvps', C.c_int32),
                ('n_update_gvps', C.c_int32), ('n_noise_gvps', C.c_int32)]


class KpdRecencConfig(C.Structure):
    _fields_ = [('in_scalar_size', C.c_int32), ('out_scalar_size', C.c_int32), ('vector_size', C.c_int32),
                ('n_rr_convs', C.c_int32), ('n_rk_convs', C.c_int32), ('n_message_gvps', C.c_int32),
                ('n_update_gvps', C.c_int32), ('message_norm_mode', C.c_int32), ('message_norm', C.c_float),
                ('k_closest', C.c_int32), ('n_keypoints', C.c_int32), ('rr_cutoff', C.c_float), ('rk_cutoff', C.c_float),
                ('kk_cutoff', C.c_float), ('kp_rad', C.c_float)]


class KpdRecegnnConfig(C.Structure):
    _fields_ = [('n_convs', C.c_int32), ('n_keypoints', C.c_int32), ('in_n_node_feat', C.c_int32),
                ('hidden_n_node_feat', C.c_int32), ('out_n_node_feat', C.c_int32), ('use_sameres_feat', C.c_int32),
                ('use_tanh', C.c_int32), ('norm', C.c_int32), ('fix_pos', C.c_int32), ('coords_range', C.c_float),
                ('message_norm', C.c_float), ('k_closest', C.c_int32), ('kk_cutoff', C.c_float), ('kp_rad', C.c_float)]


class KpdWgradItem(C.Structure):
    _fields_ = [('A', C.c_void_p), ('B', C.c_void_p), ('lda', C.c_int32), ('ldb', C.c_int32), ('K', C.c_int32), ('C', C.c_void_p), ('ldc', C.c_int32),
                ('B2', C.c_void_p), ('ldb2', C.c_int32), ('nb2', C.c_int32), ('Cx1', C.c_void_p), ('ldx1', C.c_int32), ('colsum', C.c_void_p),
                ('A2', C.c_void_p), ('lda2', C.c_int32), ('na2', C.c_int32), ('Cx2', C.c_void_p), ('ldx2', C.c_int32), ('colsum2', C.c_void_p),
                ('B3', C.c_void_p), ('ldb3', C.c_int32), ('nb3', C.c_int32), ('Cx3', C.c_void_p), ('ldx3', C.c_int32)]


class KpdRecBatch(C.Structure):
    _fields_ = [('B', C.c_int32), ('n_rec', C.c_int32), ('max_rec', C.c_int32), ('rec_ptr', C.c_void_p),
                ('rec_x', C.c_void_p), ('rec_h', C.c_void_p), ('n_rr', C.c_int32), ('rr_src', C.c_void_p),
                ('rr_dst', C.c_void_p), ('rr_rowptr', C.c_void_p)]


class KpdRecOut(C.Structure):
    _fields_ = [('kp_x', C.c_void_p), ('kp_h', C.c_void_p), ('kp_v', C.c_void_p), ('rk_src', C.c_void_p),
                ('rk_dst', C.c_void_p), ('cap_kk', C.c_int32), ('kk_src', C.c_void_p), ('kk_dst', C.c_void_p),
                ('kk_per_graph', C.c_void_p), ('counts', C.c_void_p)]


class KpdError(RuntimeError):
    pass


def lib():
    """Load (once) and return the HIP library; raise loudly when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KpdError(f'{LIB_PATH} not found: the HIP extension must be built first '
                       f'(python -c "import __graft_entry__ as g; g.build()"). There is no CPU fallback.')
    L = C.CDLL(LIB_PATH)
    L.kpd_last_error.restype = C.c_char_p
    L.kpd_version.restype = C.c_int
    L.kpd_build_flags.restype = C.c_int
    for name in EXPORTS:
        getattr(L, name)          # AttributeError if a declared symbol is not exported
    L.kpd_egnn_create.argtypes = [C.POINTER(KpdEgnnConfig), C.POINTER(C.c_void_p)]
    L.kpd_egnn_destroy.argtypes = [C.c_void_p]
    L.kpd_egnn_destroy.restype = None
    L.kpd_egnn_load_weight.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_void_p]
    L.kpd_egnn_commit.argtypes = [C.c_void_p]
    L.kpd_egnn_reserve.argtypes = [C.c_void_p] + [C.c_int32] * 6
    L.kpd_egnn_forward.argtypes = [C.c_void_p, C.POINTER(KpdBatch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.kpd_egnn_debug_state.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.kpd_egnn_last_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]
    L.kpd_egnn_profile.argtypes = [C.c_void_p, C.c_int32]
    L.kpd_egnn_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.kpd_gvp_create.argtypes = [C.POINTER(KpdGvpConfig), C.POINTER(C.c_void_p)]
    L.kpd_gvp_destroy.argtypes = [C.c_void_p]
    L.kpd_gvp_destroy.restype = None
    L.kpd_gvp_load_weight.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_void_p]
    L.kpd_gvp_commit.argtypes = [C.c_void_p]
    L.kpd_gvp_reserve.argtypes = [C.c_void_p] + [C.c_int32] * 6
    L.kpd_gvp_forward.argtypes = [C.c_void_p, C.POINTER(KpdBatch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.kpd_gvp_debug_state.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.kpd_gvp_last_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]
    L.kpd_gvp_profile.argtypes = [C.c_void_p, C.c_int32]
    L.kpd_gvp_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.kpd_recenc_create.argtypes = [C.POINTER(KpdRecencConfig), C.POINTER(C.c_void_p)]
    L.kpd_recenc_destroy.argtypes = [C.c_void_p]
    L.kpd_recenc_destroy.restype = None
    L.kpd_recenc_load_weight.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_void_p]
    L.kpd_recenc_commit.argtypes = [C.c_void_p]
    L.kpd_recenc_reserve.argtypes = [C.c_void_p] + [C.c_int32] * 4
    L.kpd_recenc_forward.argtypes = [C.c_void_p, C.POINTER(KpdRecBatch), C.POINTER(KpdRecOut), C.c_void_p]
    L.kpd_recegnn_create.argtypes = [C.POINTER(KpdRecegnnConfig), C.POINTER(C.c_void_p)]
    L.kpd_recegnn_destroy.argtypes = [C.c_void_p]
    L.kpd_recegnn_destroy.restype = None
    L.kpd_recegnn_load_weight.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_void_p]
    L.kpd_recegnn_commit.argtypes = [C.c_void_p]
    L.kpd_recegnn_reserve.argtypes = [C.c_void_p] + [C.c_int32] * 4
    L.kpd_recegnn_forward.argtypes = [C.c_void_p, C.POINTER(KpdRecBatch), C.c_void_p, C.POINTER(KpdRecOut), C.c_void_p, C.c_void_p,
                                      C.c_void_p]
    L.kpd_build_lig_graph.argtypes = [C.POINTER(KpdBatch), C.c_float, C.c_int32, C.c_float, C.c_int32, C.POINTER(KpdLigGraph),
                                      C.c_void_p]
    L.kpd_sample_update.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 8 + [C.c_int32, C.c_void_p]
    L.kpd_complex_noise.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_void_p,
                                    C.c_void_p]
    L.kpd_step_coefficients.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.kpd_egnn_trainer_create.argtypes = [C.POINTER(KpdEgnnConfig), C.POINTER(C.c_void_p)]
    L.kpd_egnn_trainer_destroy.argtypes = [C.c_void_p]
    L.kpd_egnn_trainer_destroy.restype = None
    L.kpd_egnn_trainer_bind.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32]
    L.kpd_egnn_trainer_reserve.argtypes = [C.c_void_p] + [C.c_int32] * 6
    L.kpd_egnn_trainer_forward.argtypes = [C.c_void_p, C.POINTER(KpdBatch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.kpd_egnn_trainer_backward.argtypes = [C.c_void_p] + [C.c_void_p] * 7
    L.kpd_egnn_trainer_profile.argtypes = [C.c_void_p, C.c_int32]
    L.kpd_egnn_trainer_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.kpd_gvp_trainer_create.argtypes = [C.POINTER(KpdGvpConfig), C.POINTER(C.c_void_p)]
    L.kpd_gvp_trainer_destroy.argtypes = [C.c_void_p]
    L.kpd_gvp_trainer_destroy.restype = None
    L.kpd_gvp_trainer_bind.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32]
    L.kpd_gvp_trainer_reserve.argtypes = [C.c_void_p] + [C.c_int32] * 6
    L.kpd_gvp_trainer_forward.argtypes = [C.c_void_p, C.POINTER(KpdBatch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.kpd_gvp_trainer_backward.argtypes = [C.c_void_p] + [C.c_void_p] * 8
    L.kpd_gvp_trainer_set_dropout.argtypes = [C.c_void_p, C.c_float, C.c_uint64]
    L.kpd_gvp_trainer_message_path.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    L.kpd_wgrad_batch.argtypes = [C.c_int32, C.c_int32, C.POINTER(KpdWgradItem), C.c_void_p, C.c_int64, C.c_void_p]
    L.kpd_adam_step.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64,
                                C.c_double, C.c_void_p]
    L.kpd_dropout_mask.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]
    L.kpd_recenc_trainer_create.argtypes = [C.POINTER(KpdRecencConfig), C.POINTER(C.c_void_p)]
    L.kpd_recenc_trainer_destroy.argtypes = [C.c_void_p]
    L.kpd_recenc_trainer_destroy.restype = None
    L.kpd_recenc_trainer_bind.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32]
    L.kpd_recenc_trainer_set_dropout.argtypes = [C.c_void_p, C.c_float, C.c_uint64]
    L.kpd_recenc_trainer_reserve.argtypes = [C.c_void_p] + [C.c_int32] * 4
    L.kpd_recenc_trainer_forward.argtypes = [C.c_void_p, C.POINTER(KpdRecBatch), C.POINTER(KpdRecOut), C.c_void_p]
    L.kpd_recenc_trainer_backward.argtypes = [C.c_void_p] + [C.c_void_p] * 4
    L.kpd_recegnn_trainer_create.argtypes = [C.POINTER(KpdRecegnnConfig), C.POINTER(C.c_void_p)]
    L.kpd_recegnn_trainer_destroy.argtypes = [C.c_void_p]
    L.kpd_recegnn_trainer_destroy.restype = None
    L.kpd_recegnn_trainer_bind.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32]
    L.kpd_recegnn_trainer_reserve.argtypes = [C.c_void_p] + [C.c_int32] * 4
    L.kpd_recegnn_trainer_forward.argtypes = [C.c_void_p, C.POINTER(KpdRecBatch), C.c_void_p, C.POINTER(KpdRecOut), C.c_void_p, C.c_void_p,
                                              C.c_void_p]
    L.kpd_recegnn_trainer_backward.argtypes = [C.c_void_p] + [C.c_void_p] * 3
    L.kpd_ot_emd_uniform.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    L.kpd_sgemm.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                            C.c_float, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.kpd_rec_graph_scratch_bytes.argtypes = [C.c_int32, C.c_int32]
    L.kpd_rec_graph_scratch_bytes.restype = C.c_int64
    L.kpd_build_rec_graph.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p,
                                      C.c_int32] + [C.c_void_p] * 8
    L.kpd_xyz_scratch_bytes.argtypes = [C.c_int32, C.c_int32]
    L.kpd_xyz_scratch_bytes.restype = C.c_int64
    L.kpd_xyz_emit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = L
    return L


# every symbol include/kpd.h declares (checked by tests/test_abi.py against the header text)
EXPORTS = [
    'kpd_last_error', 'kpd_version', 'kpd_build_flags', 'kpd_build_lig_graph',
    'kpd_egnn_create', 'kpd_egnn_destroy', 'kpd_egnn_load_weight', 'kpd_egnn_commit', 'kpd_egnn_reserve',
    'kpd_egnn_forward', 'kpd_egnn_debug_state', 'kpd_egnn_last_counts', 'kpd_egnn_profile',
    'kpd_egnn_profile_read', 'kpd_sample_update', 'kpd_step_coefficients', 'kpd_complex_noise',
    'kpd_gvp_create', 'kpd_gvp_destroy', 'kpd_gvp_load_weight', 'kpd_gvp_commit', 'kpd_gvp_reserve',
    'kpd_gvp_forward', 'kpd_gvp_debug_state', 'kpd_gvp_profile', 'kpd_gvp_profile_read', 'kpd_gvp_last_counts',
    'kpd_recenc_create', 'kpd_recenc_destroy', 'kpd_recenc_load_weight', 'kpd_recenc_commit', 'kpd_recenc_reserve',
    'kpd_recenc_forward',
    'kpd_recegnn_create', 'kpd_recegnn_destroy', 'kpd_recegnn_load_weight', 'kpd_recegnn_commit', 'kpd_recegnn_reserve',
    'kpd_recegnn_forward',
    'kpd_xyz_scratch_bytes', 'kpd_xyz_emit', 'kpd_rec_graph_scratch_bytes', 'kpd_build_rec_graph',
    'kpd_egnn_trainer_create', 'kpd_egnn_trainer_destroy', 'kpd_egnn_trainer_bind', 'kpd_egnn_trainer_reserve',
    'kpd_egnn_trainer_forward', 'kpd_egnn_trainer_backward', 'kpd_egnn_trainer_profile', 'kpd_egnn_trainer_profile_read', 'kpd_gvp_trainer_last_counts',
    'kpd_gvp_trainer_message_path', 'kpd_wgrad_batch', 'kpd_adam_step',
    'kpd_gvp_trainer_create', 'kpd_gvp_trainer_destroy', 'kpd_gvp_trainer_bind', 'kpd_gvp_trainer_reserve',
    'kpd_gvp_trainer_forward', 'kpd_gvp_trainer_backward', 'kpd_gvp_trainer_set_dropout', 'kpd_dropout_mask',
    'kpd_recenc_trainer_create', 'kpd_recenc_trainer_destroy', 'kpd_recenc_trainer_bind', 'kpd_recenc_trainer_set_dropout',
    'kpd_recenc_trainer_reserve', 'kpd_recenc_trainer_forward', 'kpd_recenc_trainer_backward',
    'kpd_recegnn_trainer_create', 'kpd_recegnn_trainer_destroy', 'kpd_recegnn_trainer_bind', 'kpd_recegnn_trainer_reserve',
    'kpd_recegnn_trainer_forward', 'kpd_recegnn_trainer_backward', 'kpd_ot_emd_uniform', 'kpd_sgemm',
]


def check(status: int):
    if status != 0:
        raise KpdError(f'kpd status {status}: {lib().kpd_last_error().decode()}')


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _dev_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise KpdError(f'{name} must live on the GPU (got {t.device}); the hot path has no CPU implementation')
    return t.contiguous().float()


class PreparedBatch:
    """Device-side, int32, dst-sorted view of the static part of a batch (built once per batch)."""

    def __init__(self, lig_counts: torch.Tensor, kp_counts: torch.Tensor, kk_src: torch.Tensor, kk_dst: torch.Tensor,
                 device):
        lig_counts = lig_counts.cpu().long()
        kp_counts = kp_counts.cpu().long()
        self.B = int(lig_counts.numel())
        self.n_lig = int(lig_counts.sum())
        self.n_kp = int(kp_counts.sum())
        self.max_lig = int(lig_counts.max())
        self.max_kp = int(kp_counts.max())
        if int(lig_counts.min()) < 1 or int(kp_counts.min()) < 1:
            raise KpdError('every complex needs at least one ligand atom and one keypoint')
        zero = torch.zeros(1, dtype=torch.long)
        self.lig_ptr = torch.cat([zero, lig_counts.cumsum(0)]).int().to(device)
        self.kp_ptr = torch.cat([zero, kp_counts.cumsum(0)]).int().to(device)
        kk_src = kk_src.to(device).long()
        kk_dst = kk_dst.to(device).long()
        if kk_src.numel():
            order = torch.argsort(kk_dst * (self.n_kp + 1) + kk_src)      # (dst, src) lexicographic
            kk_src, kk_dst = kk_src[order], kk_dst[order]
        self.n_kk = int(kk_src.numel())
        self.kk_src = kk_src.int().contiguous()
        self.kk_dst = kk_dst.int().contiguous()
        deg = torch.bincount(kk_dst, minlength=self.n_kp) if self.n_kk else torch.zeros(self.n_kp, dtype=torch.long, device=device)
        self.kk_rowptr = torch.cat([torch.zeros(1, dtype=torch.long, device=device), deg.cumsum(0)]).int().contiguous()

    def struct(self, lig_x, lig_h, kp_x, kp_h, kp_v=None) -> KpdBatch:
        return KpdBatch(self.B, self.n_lig, self.n_kp, self.max_lig, self.max_kp, _ptr(self.lig_ptr), _ptr(self.kp_ptr),
                        _ptr(lig_x), _ptr(lig_h), _ptr(kp_x), _ptr(kp_h), _ptr(kp_v), self.n_kk, _ptr(self.kk_src),
                        _ptr(self.kk_dst), _ptr(self.kk_rowptr))


def build_lig_graph(pb: PreparedBatch, lig_x: torch.Tensor, kp_x: torch.Tensor, ll_cutoff: float, kl_k: int, ll_k: int = 0,
                    kl_cutoff: float = 0.0):
    """Standalone graph build (kpd_build_lig_graph); returns a dict of int32 device tensors.  ll_k > 0: kNN lig-lig graph
    instead of the radius graph; kl_k == 0: radius keypoint->ligand graph of radius kl_cutoff instead of kNN."""
    dev = lig_x.device
    lig_x, kp_x = _dev_f32(lig_x, 'lig_x'), _dev_f32(kp_x, 'kp_x')
    cap_ll = max(pb.n_lig * min(pb.max_lig - 1, ll_k if ll_k > 0 else 200), 1)
    cap_kl = max(pb.n_kp * (kl_k if kl_k > 0 else min(pb.max_lig, 100)), 1)
    i32 = lambda n: torch.zeros(n, dtype=torch.int32, device=dev)
    out = dict(ll_src=i32(cap_ll), ll_dst=i32(cap_ll), ll_rowptr=i32(pb.n_lig + 1),
               kl_src=i32(cap_kl), kl_dst=i32(cap_kl), kl_rowptr=i32(pb.n_lig + 1),
               lk_src=i32(cap_kl), lk_dst=i32(cap_kl), lk_rowptr=i32(pb.n_kp + 1),
               ll_per_graph=i32(pb.B), counts=i32(8))
    lg = KpdLigGraph(cap_ll, cap_kl, *[_ptr(out[k]) for k in
                                      ('ll_src', 'll_dst', 'll_rowptr', 'kl_src', 'kl_dst', 'kl_rowptr',
                                       'lk_src', 'lk_dst', 'lk_rowptr', 'll_per_graph', 'counts')])
    bt = pb.struct(lig_x, None, kp_x, None)
    check(lib().kpd_build_lig_graph(C.byref(bt), float(ll_cutoff), int(ll_k), float(kl_cutoff), int(kl_k), C.byref(lg), _stream()))
    return out


class EgnnEngine:
    """Owns one kpd_egnn handle: packed weights + workspace for LigRecDynamics.forward."""

    def __init__(self, atom_nf, rec_nf, n_layers, hidden_nf, use_tanh, norm, update_kp_feat, message_norm, ll_k, kl_k,
                 ll_cutoff, kl_cutoff, coords_range=10.0):
        self.cfg = KpdEgnnConfig(int(atom_nf), int(rec_nf), int(n_layers), int(hidden_nf), int(bool(use_tanh)),
                                 int(bool(norm)), int(bool(update_kp_feat)), float(message_norm), int(ll_k), int(kl_k),
                                 float(ll_cutoff), float(kl_cutoff), float(coords_range))
        self.atom_nf = int(atom_nf)
        self._h = C.c_void_p()
        check(lib().kpd_egnn_create(C.byref(self.cfg), C.byref(self._h)))
        self._reserved = None

    def __del__(self):
        if getattr(self, '_h', None) and _lib is not None:
            _lib.kpd_egnn_destroy(self._h)
            self._h = None

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        L = lib()
        st = _stream()
        keep = []
        for name, t in sd.items():
            t = _dev_f32(t.detach(), name)
            keep.append(t)
            shape = (C.c_int64 * t.dim())(*t.shape)
            check(L.kpd_egnn_load_weight(self._h, name.encode(), t.data_ptr(), shape, t.dim(), st))
        torch.cuda.current_stream().synchronize()     # packing kernels read the source tensors
        check(L.kpd_egnn_commit(self._h))

    def reserve(self, pb: PreparedBatch):
        key = (pb.B, pb.n_lig, pb.n_kp, pb.n_kk, pb.max_lig, pb.max_kp)
        if self._reserved is not None and all(a <= b for a, b in zip(key, self._reserved)):
            return
        torch.cuda.synchronize()
        check(lib().kpd_egnn_reserve(self._h, *key))
        self._reserved = key if self._reserved is None else tuple(max(a, b) for a, b in zip(key, self._reserved))

    def forward(self, pb: PreparedBatch, lig_x, lig_h, kp_x, kp_h, t):
        self.reserve(pb)
        lig_x, lig_h = _dev_f32(lig_x, 'lig x_0'), _dev_f32(lig_h, 'lig h_0')
        kp_x, kp_h = _dev_f32(kp_x, 'kp x_0'), _dev_f32(kp_h, 'kp h_0')
        t = _dev_f32(t, 'timestep')
        eps_h = torch.empty(pb.n_lig, self.atom_nf, device=lig_x.device)
        eps_x = torch.empty(pb.n_lig, 3, device=lig_x.device)
        bt = pb.struct(lig_x, lig_h, kp_x, kp_h)
        check(lib().kpd_egnn_forward(self._h, C.byref(bt), t.data_ptr(), eps_h.data_ptr(), eps_x.data_ptr(), _stream()))
        return eps_h, eps_x

    def debug(self, what: str, n_floats: int = 0, device=None) -> Optional[torch.Tensor]:
        out = torch.empty(max(n_floats, 1), device=device or 'cuda')
        check(lib().kpd_egnn_debug_state(self._h, what.encode(), out.data_ptr(), n_floats, _stream()))
        return out if n_floats else None

    def profile(self, enable: bool):
        check(lib().kpd_egnn_profile(self._h, int(enable)))

    def profile_read(self):
        """(total ms, launches) of the fused edge kernel since profile(True)."""
        ms, n = C.c_double(), C.c_int32()
        check(lib().kpd_egnn_profile_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def last_counts(self):
        arr = (C.c_int32 * 8)()
        check(lib().kpd_egnn_last_counts(self._h, arr, _stream()))
        return dict(E_ll=arr[0], E_kl=arr[1], E_lk=arr[2], E_kk=arr[3], tiles=arr[4], tiles_last=arr[5], E_last=arr[6])

    def gemm_mode(self) -> str:
        """The GEMM mode the engine runs in, as the library reports it ('f32' = exact fp32 MFMA, 'f16x2' = split f16 products)."""
        arr = (C.c_int32 * 8)()
        check(lib().kpd_egnn_last_counts(self._h, arr, _stream()))
        return 'f16x2' if arr[7] else 'f32'

    def set_gemm_mode(self, mode: str):
        self.debug(f'gemm={mode}')


_PARAM_GEN = [0]


def _on_register_parameter(module, name, param):
    _PARAM_GEN[0] += 1


torch.nn.modules.module.register_module_parameter_registration_hook(_on_register_parameter)


def param_generation() -> int:
    """Counts `register_parameter` calls of every module in the process (a global torch hook): the cached parameter lists of the
    denoiser modules are rebuilt when it moves, so a swapped Parameter object is never read through a stale list."""
    return _PARAM_GEN[0]


class EgnnTrainer:
    """Owns one kpd_egnn_trainer handle: forward with saved layer states + backward of LigRecDynamics.forward.
    Parameters are bound by reference name to their live storage (read in place every step); gradients are written
    into fresh zero tensors per backward call and handed to autograd."""

    def __init__(self, cfg: 'KpdEgnnConfig', atom_nf: int, rec_nf: int):
        self.cfg, self.atom_nf, self.rec_nf = cfg, int(atom_nf), int(rec_nf)
        self._h = C.c_void_p()
        check(lib().kpd_egnn_trainer_create(C.byref(self.cfg), C.byref(self._h)))
        self._reserved = None

    def __del__(self):
        if getattr(self, '_h', None) and _lib is not None:
            _lib.kpd_egnn_trainer_destroy(self._h)
            self._h = None

    def bind(self, names, weights, grads):
        L = lib()
        for name, w, g in zip(names, weights, grads):
            if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()):
                raise KpdError(f'parameter {name} must be a contiguous fp32 GPU tensor')
            shape = (C.c_int64 * w.dim())(*w.shape)
            check(L.kpd_egnn_trainer_bind(self._h, name.encode(), w.data_ptr(), None if g is None else g.data_ptr(), shape, w.dim()))

    def reserve(self, pb: PreparedBatch):
        key = (pb.B, pb.n_lig, pb.n_kp, pb.n_kk, pb.max_lig, pb.max_kp)
        if self._reserved is not None and all(a <= b for a, b in zip(key, self._reserved)):
            return
        torch.cuda.synchronize()
        check(lib().kpd_egnn_trainer_reserve(self._h, *key))
        self._reserved = key if self._reserved is None else tuple(max(a, b) for a, b in zip(key, self._reserved))

    def forward(self, pb: PreparedBatch, lig_x, lig_h, kp_x, kp_h, t):
        self.reserve(pb)
        eps_h = torch.empty(pb.n_lig, self.atom_nf, device=lig_x.device)
        eps_x = torch.empty(pb.n_lig, 3, device=lig_x.device)
        bt = pb.struct(lig_x, lig_h, kp_x, kp_h)
        check(lib().kpd_egnn_trainer_forward(self._h, C.byref(bt), t.data_ptr(), eps_h.data_ptr(), eps_x.data_ptr(), _stream()))
        return eps_h, eps_x

    def backward(self, d_eps_h, d_eps_x, d_lig_h, d_lig_x, d_kp_h, d_kp_x):
        check(lib().kpd_egnn_trainer_backward(self._h, d_eps_h.data_ptr(), d_eps_x.data_ptr(), _ptr(d_lig_h), _ptr(d_lig_x),
                                              _ptr(d_kp_h), _ptr(d_kp_x), _stream()))

    def profile(self, enable: bool):
        check(lib().kpd_egnn_trainer_profile(self._h, int(enable)))

    def profile_read(self):
        """{'fwd' | 'bwd': (total ms, launches, edges)} of k_egnn_edge_train / k_egnn_edge_bwd since profile(True)."""
        ms, n, e = (C.c_double * 2)(), (C.c_int32 * 2)(), (C.c_double * 2)()
        check(lib().kpd_egnn_trainer_profile_read(self._h, ms, n, e))
        return {'fwd': (ms[0], n[0], e[0]), 'bwd': (ms[1], n[1], e[1])}


class GvpEngine:
    """Owns one kpd_gvp handle: packed weights + workspace for LigRecDynamicsGVP.forward."""

    def __init__(self, n_lig_scalars, n_kp_scalars, vector_size, n_convs, n_hidden_scalars, update_kp, message_norm,
                 ll_k, kl_k, ll_cutoff, kl_cutoff, n_message_gvps, n_update_gvps, n_noise_gvps):
        if message_norm == 'mean':
            mode, val = 1, 1.0
        elif message_norm == 0:
            mode, val = 2, 0.0
        else:
            mode, val = 0, float(message_norm)
        self.cfg = KpdGvpConfig(int(n_lig_scalars), int(n_kp_scalars), int(vector_size), int(n_convs),
                                int(n_hidden_scalars), int(bool(update_kp)), mode, val, int(ll_k), int(kl_k),
                                float(ll_cutoff), float(kl_cutoff), int(n_message_gvps), int(n_update_gvps),
                                int(n_noise_gvps))
        self.n_lig_scalars = int(n_lig_scalars)
        self._h = C.c_void_p()
        check(lib().kpd_gvp_create(C.byref(self.cfg), C.byref(self._h)))
        self._reserved = None

    def __del__(self):
        if getattr(self, '_h', None) and _lib is not None:
            _lib.kpd_gvp_destroy(self._h)
            self._h = None

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        L = lib()
        st = _stream()
        keep = []
        for name, t in sd.items():
            if t.numel() == 0:                     # dropout.vector_dropout.dummy_param
                continue
            t = _dev_f32(t.detach(), name)
            keep.append(t)
            shape = (C.c_int64 * t.dim())(*t.shape)
            check(L.kpd_gvp_load_weight(self._h, name.encode(), t.data_ptr(), shape, t.dim(), st))
        torch.cuda.current_stream().synchronize()
        check(L.kpd_gvp_commit(self._h))

    def reserve(self, pb: PreparedBatch):
        key = (pb.B, pb.n_lig, pb.n_kp, pb.n_kk, pb.max_lig, pb.max_kp)
        if self._reserved is not None and all(a <= b for a, b in zip(key, self._reserved)):
            return
        torch.cuda.synchronize()
        check(lib().kpd_gvp_reserve(self._h, *key))
        self._reserved = key if self._reserved is None else tuple(max(a, b) for a, b in zip(key, self._reserved))

    def forward(self, pb: PreparedBatch, lig_x, lig_h, kp_x, kp_h, kp_v, t):
        self.reserve(pb)
        lig_x, lig_h = _dev_f32(lig_x, 'lig x_0'), _dev_f32(lig_h, 'lig h_0')
        kp_x, kp_h, kp_v = _dev_f32(kp_x, 'kp x_0'), _dev_f32(kp_h, 'kp h_0'), _dev_f32(kp_v, 'kp v_0')
        t = _dev_f32(t, 'timestep')
        if tuple(kp_v.shape) != (pb.n_kp, int(self.cfg.vector_size), 3):
            raise KpdError(f'kp v_0 has shape {tuple(kp_v.shape)}, expected ({pb.n_kp}, {int(self.cfg.vector_size)}, 3)')
        eps_h = torch.empty(pb.n_lig, self.n_lig_scalars, device=lig_x.device)
        eps_x = torch.empty(pb.n_lig, 3, device=lig_x.device)
        bt = pb.struct(lig_x, lig_h, kp_x, kp_h, kp_v)
        check(lib().kpd_gvp_forward(self._h, C.byref(bt), t.data_ptr(), eps_h.data_ptr(), eps_x.data_ptr(), _stream()))
        return eps_h, eps_x

    def debug(self, what: str, n_floats: int = 0, device=None) -> Optional[torch.Tensor]:
        out = torch.empty(max(n_floats, 1), device=device or 'cuda')
        check(lib().kpd_gvp_debug_state(self._h, what.encode(), out.data_ptr(), n_floats, _stream()))
        return out if n_floats else None

    def profile(self, enable: bool):
        check(lib().kpd_gvp_profile(self._h, int(enable)))

    def profile_read(self):
        """(total ms, launches) of the message-chain kernel since profile(True)."""
        ms, n = C.c_double(), C.c_int32()
        check(lib().kpd_gvp_profile_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def last_counts(self):
        arr = (C.c_int32 * 8)()
        check(lib().kpd_gvp_last_counts(self._h, arr, _stream()))
        return dict(E_ll=arr[0], E_kl=arr[1], E_lk=arr[2], E_kk=arr[3], tiles=arr[4], tiles_last=arr[5], E_last=arr[6])

    def gemm_mode(self) -> str:
        """The GEMM mode the engine runs in, as the library reports it ('f16x2' needs 256 hidden scalars; otherwise 'f32')."""
        arr = (C.c_int32 * 8)()
        check(lib().kpd_gvp_last_counts(self._h, arr, _stream()))
        return 'f16x2' if arr[7] else 'f32'

    def set_gemm_mode(self, mode: str):
        self.debug(f'gemm={mode}')


class GvpTrainer:
    """Owns one kpd_gvp_trainer handle: forward with saved conv states + backward of LigRecDynamicsGVP.forward."""

    def __init__(self, cfg: 'KpdGvpConfig'):
        self.cfg = cfg
        self._h = C.c_void_p()
        check(lib().kpd_gvp_trainer_create(C.byref(self.cfg), C.byref(self._h)))
        self._reserved = None

    def __del__(self):
        if getattr(self, '_h', None) and _lib is not None:
            _lib.kpd_gvp_trainer_destroy(self._h)
            self._h = None

    def bind(self, names, weights, grads):
        L = lib()
        for name, w, g in zip(names, weights, grads):
            if w.numel() == 0:                     # dropout.vector_dropout.dummy_param
                continue
            if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()):
                raise KpdError(f'parameter {name} must be a contiguous fp32 GPU tensor')
            shape = (C.c_int64 * w.dim())(*w.shape)
            check(L.kpd_gvp_trainer_bind(self._h, name.encode(), w.data_ptr(), None if g is None else g.data_ptr(), shape, w.dim()))

    def reserve(self, pb: PreparedBatch):
        key = (pb.B, pb.n_lig, pb.n_kp, pb.n_kk, pb.max_lig, pb.max_kp)
        if self._reserved is not None and all(a <= b for a, b in zip(key, self._reserved)):
            return
        torch.cuda.synchronize()
        check(lib().kpd_gvp_trainer_reserve(self._h, *key))
        self._reserved = key if self._reserved is None else tuple(max(a, b) for a, b in zip(key, self._reserved))

    def set_dropout(self, rate: float, seed: int):
        check(lib().kpd_gvp_trainer_set_dropout(self._h, float(rate), int(seed) & (2 ** 64 - 1)))

    def forward(self, pb: PreparedBatch, lig_x, lig_h, kp_x, kp_h, kp_v, t):
        self.reserve(pb)
        eps_h = torch.empty(pb.n_lig, self.cfg.n_lig_scalars, device=lig_x.device)
        eps_x = torch.empty(pb.n_lig, 3, device=lig_x.device)
        bt = pb.struct(lig_x, lig_h, kp_x, kp_h, kp_v)
        check(lib().kpd_gvp_trainer_forward(self._h, C.byref(bt), t.data_ptr(), eps_h.data_ptr(), eps_x.data_ptr(), _stream()))
        return eps_h, eps_x

    def backward(self, d_eps_h, d_eps_x, d_lig_h, d_kp_h, d_kp_v, d_lig_x=None, d_kp_x=None):
        check(lib().kpd_gvp_trainer_backward(self._h, d_eps_h.data_ptr(), d_eps_x.data_ptr(), _ptr(d_lig_h), _ptr(d_kp_h), _ptr(d_kp_v),
                                             _ptr(d_lig_x), _ptr(d_kp_x), _stream()))

    def last_counts(self):
        arr = (C.c_int32 * 4)()
        check(lib().kpd_gvp_trainer_last_counts(self._h, arr))
        return dict(E_ll=arr[0], E_kl=arr[1], E_lk=arr[2], E_kk=arr[3])

    def message_path(self) -> int:
        """1: the convs' edge messages run through the register-chained kernels (hidden width 256), 0: one GVP at a time."""
        v = C.c_int32(0)
        check(lib().kpd_gvp_trainer_message_path(self._h, C.byref(v)))
        return int(v.value)


def dropout_mask(seed: int, conv: int, node_type: int, position: int, kind: int, n: int, rate: float, device='cuda') -> torch.Tensor:
    """One dropout stream of the GVP training path (kpd_dropout_mask): n entries in {0, 1 / (1 - rate)}."""
    out = torch.empty(int(n), device=device, dtype=torch.float32)
    check(lib().kpd_dropout_mask(int(seed) & (2 ** 64 - 1), int(conv), int(node_type), int(position), int(kind), int(n), float(rate),
                                 out.data_ptr(), _stream()))
    return out


def _norm_mode(message_norm):
    if message_norm == 'mean':
        return 1, 1.0
    if message_norm == 0:
        return 2, 0.0
    return 0, float(message_norm)


def sorted_csr(src: torch.Tensor, dst: torch.Tensor, n_dst: int, device, return_order: bool = False):
    """(src, dst) sorted by (dst, src) as int32 + CSR row pointer over dst (+ the permutation for edge data)."""
    src, dst = src.to(device).long(), dst.to(device).long()
    order = torch.arange(src.numel(), device=device)
    if src.numel():
        order = torch.argsort(dst * (int(src.max()) + 1) + src)
        src, dst = src[order], dst[order]
    deg = torch.bincount(dst, minlength=n_dst) if src.numel() else torch.zeros(n_dst, dtype=torch.long, device=device)
    rowptr = torch.cat([torch.zeros(1, dtype=torch.long, device=device), deg.cumsum(0)])
    out = (src.int().contiguous(), dst.int().contiguous(), rowptr.int().contiguous())
    return out + (order,) if return_order else out


class RecEncEngine:
    """Owns one kpd_recenc handle: packed weights + workspace for ReceptorEncoderGVP.forward."""

    def __init__(self, in_scalar_size, out_scalar_size, vector_size, n_rr_convs, n_rk_convs, n_message_gvps, n_update_gvps,
                 message_norm, k_closest, n_keypoints, rr_cutoff, rk_cutoff, kk_cutoff, kp_rad=0.0):
        mode, val = _norm_mode(message_norm)
        self.cfg = KpdRecencConfig(int(in_scalar_size), int(out_scalar_size), int(vector_size), int(n_rr_convs),
                                   int(n_rk_convs), int(n_message_gvps), int(n_update_gvps), mode, val, int(k_closest),
                                   int(n_keypoints), float(rr_cutoff), float(rk_cutoff), float(kk_cutoff), float(kp_rad))
        # rk edges per keypoint: k of the kNN graph, or at most 10 of the radius graph (receptor_encoder_gvp.py:306)
        self.S, self.K, self.k = int(out_scalar_size), int(n_keypoints), int(k_closest) if k_closest else 10
        self._h = C.c_void_p()
        check(lib().kpd_recenc_create(C.byref(self.cfg), C.byref(self._h)))

    def __del__(self):
        if getattr(self, '_h', None) and _lib is not None:
            _lib.kpd_recenc_destroy(self._h)
            self._h = None

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        L = lib()
        st = _stream()
        keep = []
        for name, t in sd.items():
            if t.numel() == 0:
                continue
            t = _dev_f32(t.detach(), name)
            keep.append(t)
            shape = (C.c_int64 * t.dim())(*t.shape)
            check(L.kpd_recenc_load_weight(self._h, name.encode(), t.data_ptr(), shape, t.dim(), st))
        torch.cuda.current_stream().synchronize()
        check(L.kpd_recenc_commit(self._h))

    def forward(self, rec_counts: torch.Tensor, rec_x, rec_h, rr_src, rr_dst):
        return _recenc_call(self, lib().kpd_recenc_reserve, lib().kpd_recenc_forward, rec_counts, rec_x, rec_h, rr_src, rr_dst)


def _recenc_call(eng, reserve_fn, forward_fn, rec_counts, rec_x, rec_h, rr_src, rr_dst):
    """Shared by RecEncEngine.forward and RecEncTrainer.forward: batch structs, output buffers, one library call."""
    dev = rec_x.device
    rec_counts = rec_counts.cpu().long()
    B, n_rec, max_rec = int(rec_counts.numel()), int(rec_counts.sum()), int(rec_counts.max())
    if int(rec_counts.min()) < 1:
        raise KpdError('every pocket needs at least one receptor atom')
    rec_ptr = torch.cat([torch.zeros(1, dtype=torch.long), rec_counts.cumsum(0)]).int().to(dev)
    rec_x, rec_h = _dev_f32(rec_x, 'rec x_0'), _dev_f32(rec_h, 'rec h_0')
    s, d, rowptr = sorted_csr(rr_src, rr_dst, n_rec, dev)
    torch.cuda.synchronize()
    check(reserve_fn(eng._h, B, n_rec, int(s.numel()), max_rec))
    n_kp = B * eng.K
    cap_kk = max(n_kp * min(eng.K - 1, 100), 1)
    f32 = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
    i32 = lambda n: torch.zeros(n, device=dev, dtype=torch.int32)
    out = dict(kp_x=f32(n_kp, 3), kp_h=f32(n_kp, eng.S), kp_v=f32(n_kp, int(eng.cfg.vector_size), 3), rk_src=i32(n_kp * eng.k),
               rk_dst=i32(n_kp * eng.k), kk_src=i32(cap_kk), kk_dst=i32(cap_kk), kk_per_graph=i32(B), counts=i32(2))
    bt = KpdRecBatch(B, n_rec, max_rec, _ptr(rec_ptr), _ptr(rec_x), _ptr(rec_h), int(s.numel()), _ptr(s), _ptr(d),
                     _ptr(rowptr))
    ro = KpdRecOut(_ptr(out['kp_x']), _ptr(out['kp_h']), _ptr(out['kp_v']), _ptr(out['rk_src']), _ptr(out['rk_dst']),
                   cap_kk, _ptr(out['kk_src']), _ptr(out['kk_dst']), _ptr(out['kk_per_graph']), _ptr(out['counts']))
    check(forward_fn(eng._h, C.byref(bt), C.byref(ro), _stream()))
    e_kk, e_rk = out['counts'].tolist()            # once per pocket: a host sync here is fine
    out['kk_src'], out['kk_dst'] = out['kk_src'][:e_kk], out['kk_dst'][:e_kk]
    out['rk_src'], out['rk_dst'] = out['rk_src'][:e_rk], out['rk_dst'][:e_rk]
    out['_keep'] = (rec_ptr, rec_x, rec_h, s, d, rowptr)      # the training engine reads these again in its backward pass
    return out


class RecEncTrainer:
    """Owns one kpd_recenc_trainer handle: ReceptorEncoderGVP.forward with saved node states + its backward pass.  Parameters
    are bound by reference name to live storage; gradients are accumulated into the buffers bound for the backward call."""

    def __init__(self, cfg: 'KpdRecencConfig'):
        self.cfg = cfg
        self.S, self.K = int(cfg.out_scalar_size), int(cfg.n_keypoints)
        self.k = int(cfg.k_closest) if cfg.k_closest else 10
        self._h = C.c_void_p()
        check(lib().kpd_recenc_trainer_create(C.byref(self.cfg), C.byref(self._h)))

    def __del__(self):
        if getattr(self, '_h', None) and _lib is not None:
            _lib.kpd_recenc_trainer_destroy(self._h)
            self._h = None

    def bind(self, names, weights, grads):
        L = lib()
        for name, w, g in zip(names, weights, grads):
            if w.numel() == 0:                     # dropout.vector_dropout.dummy_param
                continue
            if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()):
                raise KpdError(f'parameter {name} must be a contiguous fp32 GPU tensor')
            shape = (C.c_int64 * w.dim())(*w.shape)
            check(L.kpd_recenc_trainer_bind(self._h, name.encode(), w.data_ptr(), None if g is None else g.data_ptr(), shape, w.dim()))

    def set_dropout(self, rate: float, seed: int):
        check(lib().kpd_recenc_trainer_set_dropout(self._h, float(rate), int(seed) & (2 ** 64 - 1)))

    def forward(self, rec_counts, rec_x, rec_h, rr_src, rr_dst):
        return _recenc_call(self, lib().kpd_recenc_trainer_reserve, lib().kpd_recenc_trainer_forward, rec_counts, rec_x, rec_h,
                            rr_src, rr_dst)

    def backward(self, d_kp_x, d_kp_h, d_kp_v):
        check(lib().kpd_recenc_trainer_backward(self._h, _ptr(d_kp_x), _ptr(d_kp_h), _ptr(d_kp_v), _stream()))


class RecEgnnEngine:
    """Owns one kpd_recegnn handle: weights + workspace for ReceptorEncoder.forward (models/receptor_encoder.py)."""

    def __init__(self, n_convs, n_keypoints, in_n_node_feat, hidden_n_node_feat, out_n_node_feat, use_sameres_feat, use_tanh,
                 coords_range, message_norm, k_closest, norm, fix_pos, kk_cutoff, kp_rad=0.0):
        self.cfg = KpdRecegnnConfig(int(n_convs), int(n_keypoints), int(in_n_node_feat), int(hidden_n_node_feat),
                                    int(out_n_node_feat), int(bool(use_sameres_feat)), int(bool(use_tanh)), int(bool(norm)),
                                    int(bool(fix_pos)), float(coords_range), float(message_norm), int(k_closest), float(kk_cutoff),
                                    float(kp_rad))
        self.D, self.K, self.k, self.ef = int(out_n_node_feat), int(n_keypoints), int(k_closest), bool(use_sameres_feat)
        self.rk_cap = int(k_closest) if k_closest else 100       # rk edges per keypoint: k, or at most 100 within kp_rad (:246)
        self._h = C.c_void_p()
        check(lib().kpd_recegnn_create(C.byref(self.cfg), C.byref(self._h)))

    def __del__(self):
        if getattr(self, '_h', None) and _lib is not None:
            _lib.kpd_recegnn_destroy(self._h)
            self._h = None

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        L = lib()
        st = _stream()
        keep = []
        for name, t in sd.items():
            if t.numel() == 0:
                continue
            t = _dev_f32(t.detach(), name)
            keep.append(t)
            shape = (C.c_int64 * t.dim())(*t.shape)
            check(L.kpd_recegnn_load_weight(self._h, name.encode(), t.data_ptr(), shape, t.dim(), st))
        torch.cuda.current_stream().synchronize()
        check(L.kpd_recegnn_commit(self._h))

    def forward(self, rec_counts: torch.Tensor, rec_x, rec_h, rr_src, rr_dst, same_res=None):
        return _recegnn_call(self, lib().kpd_recegnn_reserve, lib().kpd_recegnn_forward, rec_counts, rec_x, rec_h, rr_src, rr_dst, same_res)


def _recegnn_call(eng, reserve_fn, forward_fn, rec_counts, rec_x, rec_h, rr_src, rr_dst, same_res):
    """Shared by RecEgnnEngine.forward and RecEgnnTrainer.forward."""
    dev = rec_x.device
    rec_counts = rec_counts.cpu().long()
    B, n_rec, max_rec = int(rec_counts.numel()), int(rec_counts.sum()), int(rec_counts.max())
    if int(rec_counts.min()) < eng.k:
        raise KpdError(f'every pocket needs at least k_closest={eng.k} receptor atoms (the reference stacks exactly k '
                       f'neighbour distances per keypoint)')
    rec_ptr = torch.cat([torch.zeros(1, dtype=torch.long), rec_counts.cumsum(0)]).int().to(dev)
    rec_x, rec_h = _dev_f32(rec_x, 'rec x_0'), _dev_f32(rec_h, 'rec h_0')
    s, d, rowptr, order = sorted_csr(rr_src, rr_dst, n_rec, dev, return_order=True)
    a = None
    if eng.ef:
        if same_res is None:
            raise KpdError("use_sameres_feat needs g.edges['rr'].data['same_res']")
        a = same_res.to(dev).reshape(-1)[order].float().contiguous()
    torch.cuda.synchronize()
    check(reserve_fn(eng._h, B, n_rec, int(s.numel()), max_rec))
    n_kp = B * eng.K
    cap_kk = max(n_kp * min(eng.K - 1, 100), 1)
    f32 = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
    i32 = lambda n: torch.zeros(n, device=dev, dtype=torch.int32)
    cap_rk = n_kp * min(eng.rk_cap, max_rec)
    out = dict(kp_x=f32(n_kp, 3), kp_h=f32(n_kp, eng.D), rk_src=i32(cap_rk), rk_dst=i32(cap_rk),
               kk_src=i32(cap_kk), kk_dst=i32(cap_kk), kk_per_graph=i32(B), counts=i32(2), rec_h=f32(n_rec, eng.D),
               rec_x=f32(n_rec, 3))
    bt = KpdRecBatch(B, n_rec, max_rec, _ptr(rec_ptr), _ptr(rec_x), _ptr(rec_h), int(s.numel()), _ptr(s), _ptr(d), _ptr(rowptr))
    ro = KpdRecOut(_ptr(out['kp_x']), _ptr(out['kp_h']), None, _ptr(out['rk_src']), _ptr(out['rk_dst']), cap_kk,
                   _ptr(out['kk_src']), _ptr(out['kk_dst']), _ptr(out['kk_per_graph']), _ptr(out['counts']))
    check(forward_fn(eng._h, C.byref(bt), _ptr(a) if a is not None else None, C.byref(ro), _ptr(out['rec_h']),
                     _ptr(out['rec_x']), _stream()))
    e_kk, e_rk = out['counts'].tolist()            # once per pocket: a host sync here is fine
    out['kk_src'], out['kk_dst'] = out['kk_src'][:e_kk], out['kk_dst'][:e_kk]
    out['rk_src'], out['rk_dst'] = out['rk_src'][:e_rk], out['rk_dst'][:e_rk]
    out['_keep'] = (rec_ptr, rec_x, rec_h, s, d, rowptr, a)      # the training engine reads these again in its backward pass
    return out


class RecEgnnTrainer:
    """Owns one kpd_recegnn_trainer handle: ReceptorEncoder.forward with saved layer states + its backward pass."""

    def __init__(self, cfg: 'KpdRecegnnConfig'):
        self.cfg = cfg
        self.D, self.K, self.k, self.ef = int(cfg.out_n_node_feat), int(cfg.n_keypoints), int(cfg.k_closest), bool(cfg.use_sameres_feat)
        self.rk_cap = self.k if self.k else 100                   # rk edges per keypoint: k, or at most 100 within kp_rad
        self._h = C.c_void_p()
        check(lib().kpd_recegnn_trainer_create(C.byref(self.cfg), C.byref(self._h)))

    def __del__(self):
        if getattr(self, '_h', None) and _lib is not None:
            _lib.kpd_recegnn_trainer_destroy(self._h)
            self._h = None

    def bind(self, names, weights, grads):
        L = lib()
        for name, w, g in zip(names, weights, grads):
            if w.numel() == 0:
                continue
            if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()):
                raise KpdError(f'parameter {name} must be a contiguous fp32 GPU tensor')
            shape = (C.c_int64 * w.dim())(*w.shape)
            check(L.kpd_recegnn_trainer_bind(self._h, name.encode(), w.data_ptr(), None if g is None else g.data_ptr(), shape, w.dim()))

    def forward(self, rec_counts, rec_x, rec_h, rr_src, rr_dst, same_res=None):
        return _recegnn_call(self, lib().kpd_recegnn_trainer_reserve, lib().kpd_recegnn_trainer_forward, rec_counts, rec_x, rec_h,
                             rr_src, rr_dst, same_res)

    def backward(self, d_kp_x, d_kp_h):
        check(lib().kpd_recegnn_trainer_backward(self._h, _ptr(d_kp_x), _ptr(d_kp_h), _stream()))


def ot_emd_uniform(costs, n_threads: int = 0):
    """Exact transport plans (uniform masses) for a list of [n_i, m_i] cost matrices (numpy float64, host): kpd_ot_emd_uniform.
    Host-side like the reference's POT call; the library spreads the problems over host threads (ctypes releases the GIL)."""
    import numpy as np
    if not costs:
        return []
    ns = np.asarray([c.shape[0] for c in costs], dtype=np.int32)
    ms = np.asarray([c.shape[1] for c in costs], dtype=np.int32)
    sizes = ns.astype(np.int64) * ms.astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    flat = np.concatenate([np.ascontiguousarray(c, dtype=np.float64).reshape(-1) for c in costs])
    plan = np.empty_like(flat)
    if n_threads <= 0:
        # one thread per problem where the host has the hardware threads (the solves of a batch are a burst of ~20 ms each: a CPU quota
        # averaged over a scheduling period still lets them all start at once; measured 67 -> ~20 ms per B = 64 batch on a 16-CPU quota)
        n_threads = max(1, min(len(os.sched_getaffinity(0)), 64))
    check(lib().kpd_ot_emd_uniform(len(costs), ns.ctypes.data, ms.ctypes.data, offs.ctypes.data, flat.ctypes.data, plan.ctypes.data,
                                   int(n_threads)))
    return [plan[o:o + s].reshape(int(a), int(b)) for o, s, a, b in zip(offs, sizes, ns, ms)]


def zero_grads_like(params, wanted):
    """Zero-filled gradient buffers for the parameters whose flag in `wanted` is set (None for the others), carved out of ONE flat
    allocation with one fill -- a model has a few hundred parameter tensors, and a fill kernel each is most of a millisecond per step.
    Offsets are multiples of 64 floats (256 B), so every view is as aligned as a tensor of its own."""
    sizes = [((p.numel() + 63) // 64) * 64 if (w and p.numel()) else 0 for p, w in zip(params, wanted)]
    total = sum(sizes)
    if total == 0:
        return [None] * len(params)
    ref = next(p for p, n in zip(params, sizes) if n)
    flat = torch.zeros(total, device=ref.device, dtype=ref.dtype)
    out, off = [], 0
    for p, n in zip(params, sizes):
        out.append(flat[off:off + p.numel()].view(p.shape) if n else None)
        off += n
    return out


def sgemm(a: torch.Tensor, b: torch.Tensor, trans_a=False, trans_b=False, alpha=1.0, beta=0.0, out: torch.Tensor = None,
          workspace: torch.Tensor = None, colsum: torch.Tensor = None) -> torch.Tensor:
    """out = alpha op(a) op(b) + beta out through kpd_sgemm (the GEMM of the training engines).  a, b, out: 2-D fp32 device tensors whose
    last dimension is contiguous; row strides and storage offsets are passed as they are (views of wider arrays are the tested case).
    `workspace`: contiguous fp32 device scratch that lets a K-dominated product be split along K.  `colsum` (trans_a and not trans_b):
    contiguous [M] tensor that receives += the column sums of a."""
    for t, name in ((a, 'a'), (b, 'b')):
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and (t.shape[1] <= 1 or t.stride(1) == 1)):
            raise KpdError(f'sgemm: {name} must be a 2-D fp32 device tensor with a contiguous last dimension')
    M, K = (a.shape[1], a.shape[0]) if trans_a else tuple(a.shape)
    K2, N = (b.shape[1], b.shape[0]) if trans_b else tuple(b.shape)
    if K != K2:
        raise KpdError(f'sgemm: inner sizes {K} and {K2} differ')
    if out is None:
        out = torch.zeros(M, N, device=a.device)
    if tuple(out.shape) != (M, N) or not out.is_cuda or out.dtype != torch.float32 or (N > 1 and out.stride(1) != 1):
        raise KpdError('sgemm: out must be an fp32 device tensor [M, N] with a contiguous last dimension')
    ld = lambda t: int(t.stride(0)) if t.shape[0] > 1 else max(int(t.shape[1]), 1)
    check(lib().kpd_sgemm(int(trans_a), int(trans_b), M, N, K, float(alpha), a.data_ptr(), ld(a), b.data_ptr(), ld(b), float(beta),
                          out.data_ptr(), ld(out), colsum.data_ptr() if colsum is not None else None,
                          workspace.data_ptr() if workspace is not None else None,
                          int(workspace.numel()) if workspace is not None else 0, _stream()))
    return out


def adam_step(table_dev: torch.Tensor, max_numel: int, mode: int, lr=0.0, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, step=1, clip_value=0.0):
    """kpd_adam_step over a device table [n, 5] of int64 (parameter, gradient, exp_avg, exp_avg_sq pointers, element count): optim.py builds it."""
    if not (table_dev.is_cuda and table_dev.dtype == torch.int64 and table_dev.dim() == 2 and table_dev.shape[1] == 5 and table_dev.is_contiguous()):
        raise KpdError('adam_step: the table must be a contiguous int64 device tensor [n, 5]')
    check(lib().kpd_adam_step(table_dev.data_ptr(), int(table_dev.shape[0]), int(max_numel), int(mode), float(lr), float(beta1), float(beta2), float(eps),
                              float(weight_decay), int(step), float(clip_value), _stream()))


def wgrad_batch(kind: int, items, workspace: torch.Tensor):
    """kpd_wgrad_batch: `items` = list of dicts with the fields of kpd_wgrad_item (tensors or None, ints); leading dimensions are taken from
    the tensors' row strides.  Outputs are accumulated in place."""
    arr = (KpdWgradItem * len(items))()
    ld = lambda t: int(t.stride(0)) if t.shape[0] > 1 else max(int(t.shape[-1]), 1)
    p = lambda t: t.data_ptr() if t is not None else None
    for k, it in enumerate(items):
        e = arr[k]
        e.A, e.B, e.lda, e.ldb, e.K = p(it['A']), p(it['B']), ld(it['A']), ld(it['B']), int(it['A'].shape[0])
        c = it.get('C')
        e.C, e.ldc = p(c), ld(c) if c is not None else 0
        for src, ldf, nf, dst, ldd in (('B2', 'ldb2', 'nb2', 'Cx1', 'ldx1'), ('A2', 'lda2', 'na2', 'Cx2', 'ldx2'), ('B3', 'ldb3', 'nb3', 'Cx3', 'ldx3')):
            t, o = it.get(src), it.get(dst)
            setattr(e, src, p(t)); setattr(e, ldf, ld(t) if t is not None else 0); setattr(e, nf, int(it.get(nf, 0)))
            setattr(e, dst, p(o)); setattr(e, ldd, ld(o) if o is not None else 0)
        e.colsum, e.colsum2 = p(it.get('colsum')), p(it.get('colsum2'))
    check(lib().kpd_wgrad_batch(int(kind), len(items), arr, workspace.data_ptr(), int(workspace.numel()), _stream()))


def step_coefficients(gamma: torch.Tensor, s: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """[B,3] reverse-step coefficients from the gamma table (kpd_step_coefficients): one launch per step."""
    gamma, s, t = _dev_f32(gamma, 'gamma'), _dev_f32(s, 's'), _dev_f32(t, 't')
    if s.shape != t.shape or s.dim() != 1:
        raise KpdError(f'step_coefficients: s {tuple(s.shape)} and t {tuple(t.shape)} must be equal 1-D tensors')
    coef = torch.empty(s.shape[0], 3, device=s.device)
    check(lib().kpd_step_coefficients(gamma.data_ptr(), int(gamma.shape[0]), s.data_ptr(), t.data_ptr(), int(s.shape[0]),
                                      coef.data_ptr(), _stream()))
    return coef


def complex_noise(pb: PreparedBatch, width: int, complex_ids: torch.Tensor, seed: int, step: int, tag: int) -> torch.Tensor:
    """[n_lig, width] N(0,1) noise that depends only on (seed, complex id, step, tag, position in the complex)
    (kpd_complex_noise): a sharded run draws what the single-process run draws."""
    if not (complex_ids.is_cuda and complex_ids.dtype == torch.int64 and complex_ids.numel() == pb.B):
        raise KpdError('complex_ids must be an int64 GPU tensor with one id per complex')
    out = torch.empty(pb.n_lig, int(width), device=complex_ids.device, dtype=torch.float32)
    check(lib().kpd_complex_noise(pb.B, _ptr(pb.lig_ptr), int(width), _ptr(complex_ids.contiguous()), int(seed) & (2 ** 64 - 1),
                                  int(step), int(tag), _ptr(out), _stream()))
    return out


def build_rec_graph(rec_x: torch.Tensor, rec_ptr: torch.Tensor, max_rec: int, r: float, res_idx: Optional[torch.Tensor] = None,
                    max_nn: int = 100):
    """rr radius graph (+ same-residue flags) of a batch of pockets on the GPU (kpd_build_rec_graph).
    rec_x [n_rec,3] fp32, rec_ptr [B+1] int32, res_idx [n_rec] int32 or None, all on the GPU.
    Returns (src, dst, per_graph [B], same_res bool [E] or None), dst-major with global row numbers.  One host sync
    (the edge count) — this is input-pipeline work, once per batch, not step-path work."""
    rec_x = _dev_f32(rec_x, 'rec_x')
    if not (rec_ptr.is_cuda and rec_ptr.dtype == torch.int32):
        raise KpdError('rec_ptr must be an int32 GPU tensor')
    n_rec, B = rec_x.shape[0], rec_ptr.numel() - 1
    if res_idx is not None and not (res_idx.is_cuda and res_idx.dtype == torch.int32 and res_idx.numel() == n_rec):
        raise KpdError('res_idx must be an int32 GPU tensor with one entry per receptor atom')
    dev = rec_x.device
    cap = n_rec * max(1, min(int(max_nn), int(max_rec) - 1))
    src = torch.empty(cap, dtype=torch.int32, device=dev)
    dst = torch.empty(cap, dtype=torch.int32, device=dev)
    rowptr = torch.empty(n_rec + 1, dtype=torch.int32, device=dev)
    per_graph = torch.empty(B, dtype=torch.int32, device=dev)
    counts = torch.empty(2, dtype=torch.int32, device=dev)
    same = torch.empty(cap, dtype=torch.uint8, device=dev) if res_idx is not None else None
    scratch = torch.empty(int(lib().kpd_rec_graph_scratch_bytes(n_rec, B)), dtype=torch.uint8, device=dev)
    check(lib().kpd_build_rec_graph(_ptr(rec_x), _ptr(rec_ptr), B, n_rec, int(max_rec), float(r), int(max_nn),
                                    _ptr(res_idx.contiguous()) if res_idx is not None else None, cap, _ptr(src), _ptr(dst),
                                    _ptr(rowptr), _ptr(per_graph), _ptr(same), _ptr(counts), _ptr(scratch), _stream()))
    E = int(counts[0].item())
    if E > cap:
        raise KpdError(f'rr graph: {E} edges exceed the capacity {cap} (internal sizing error)')
    return src[:E], dst[:E], per_graph, (same[:E].bool() if same is not None else None)


def xyz_emit(pos: torch.Tensor, feat: torch.Tensor, lig_ptr: torch.Tensor, elements):
    """Element decode + XYZ text of a batch of ligands on the GPU (kpd_xyz_emit).
    pos [N,3], feat [N,F] fp32 GPU tensors, lig_ptr [B+1] int32 GPU tensor, elements: F symbols.
    Returns (element index per atom [N] int32 GPU tensor, text bytes, text_ptr list of B+1 offsets); the only host
    synchronisation is the copy of the finished text."""
    pos, feat = _dev_f32(pos, 'pos'), _dev_f32(feat, 'feat')
    if not (lig_ptr.is_cuda and lig_ptr.dtype == torch.int32 and lig_ptr.dim() == 1 and lig_ptr.numel() >= 1):
        raise KpdError('lig_ptr must be an int32 GPU tensor of B + 1 offsets')
    N, B, F = pos.shape[0], lig_ptr.numel() - 1, feat.shape[1] if feat.dim() == 2 else -1
    if pos.shape != (N, 3) or feat.shape[0] != N or F != len(elements):
        raise KpdError(f'xyz_emit: pos {tuple(pos.shape)}, feat {tuple(feat.shape)}, {len(elements)} element symbols')
    packed = []
    for el in elements:
        b = el.encode('ascii')
        if not 1 <= len(b) <= 4:
            raise KpdError(f'element symbol {el!r} must be 1-4 ASCII characters')
        packed.append(int.from_bytes(b.ljust(4, b'\0'), 'little'))
    dev = pos.device
    symbols = torch.tensor(packed, dtype=torch.int32, device=dev)        # ASCII: the top bit is never set
    elem = torch.empty(N, dtype=torch.int32, device=dev)
    capacity = 72 * N + 16 * B + 16         # a line is at most 71 bytes, a header at most 12
    text = torch.empty(capacity, dtype=torch.uint8, device=dev)
    text_ptr = torch.empty(B + 1, dtype=torch.int64, device=dev)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    scratch = torch.empty(int(lib().kpd_xyz_scratch_bytes(N, B)), dtype=torch.uint8, device=dev)
    check(lib().kpd_xyz_emit(_ptr(pos), _ptr(feat), _ptr(lig_ptr), N, B, F, _ptr(symbols), _ptr(elem), _ptr(text), capacity,
                             _ptr(text_ptr), _ptr(status), _ptr(scratch), _stream()))
    ptr = text_ptr.cpu().tolist()
    st = int(status.item())
    if st & 2:
        raise KpdError('xyz_emit: text buffer too small (internal sizing error)')
    if st & 1:
        raise KpdError('xyz_emit: a coordinate with |x| >= 2^53 cannot be printed (diverged sample?)')
    return elem, bytes(text[:ptr[-1]].cpu().numpy()), ptr


def sample_update(pb: PreparedBatch, atom_nf, lig_x, lig_h, kp_x, eps_x, eps_h, noise_x, noise_h, coef):
    """In-place reverse-diffusion update + ligand-COM removal (kpd_sample_update)."""
    for t in (lig_x, lig_h, kp_x):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
            raise KpdError('sample_update state tensors must be contiguous fp32 GPU tensors (updated in place)')
    args = [_dev_f32(a, 'arg') for a in (eps_x, eps_h, noise_x, noise_h, coef)]
    check(lib().kpd_sample_update(pb.B, _ptr(pb.lig_ptr), _ptr(pb.kp_ptr), int(atom_nf), _ptr(lig_x), _ptr(lig_h),
                                  _ptr(kp_x), *[a.data_ptr() for a in args], pb.max_lig, _stream()))
