"""Range guard of the opt-in f16x2 mode (`-m gpu`): weights whose packed, scaled form leaves the finite f16 range must make the
mode refuse loudly -- at engine creation when KPD_GEMM=f16x2 asked for it, at the switch when `gemm_mode` asks for it -- while the
exact fp32 mode keeps working; and a model's explicit `gemm_mode` survives every engine rebuild."""
import pytest
import torch

from keypoint_diffusion_amd import graph as G
from keypoint_diffusion_amd import hip, synth
from keypoint_diffusion_amd.dynamics import LigRecDynamics
from keypoint_diffusion_amd.dynamics_gvp import LigRecDynamicsGVP

from . import util
from .test_gvp_gpu import GVP_ALL_ATOM

pytestmark = pytest.mark.gpu
CUT = util.CUTOFFS_ALL_ATOM


def _batch(cuda, v=None):
    gs = synth.synth_complexes([70, 50], [9, 12], 20, CUT, seed=3)
    return util.fixed_encode(G.batch(gs), n_vec=v).to(cuda)


@pytest.mark.parametrize('arch', ['egnn', 'gvp'])
def test_out_of_range_weights_refuse_f16x2_and_keep_f32(cuda, arch, monkeypatch):
    monkeypatch.delenv('KPD_GEMM', raising=False)
    if arch == 'egnn':
        model = synth.fill_state_dict_(LigRecDynamics(10, 10, graph_cutoffs=CUT, **util.EGNN_C2), 0).eval().to(cuda)
        big = model.egnn.conv_layers[2].edge_mlp['kl'][2].weight
        g = _batch(cuda)
    else:
        model = synth.fill_state_dict_(LigRecDynamicsGVP(10, 10, graph_cutoffs=CUT, **GVP_ALL_ATOM), 0).eval().to(cuda)
        big = model.noise_predictor.conv_layers[1].edge_message_fns['kp_kl_lig'][1].to_feats_out[0].weight
        g = _batch(cuda, v=16)
    t = torch.tensor([0.3, 0.8], device=cuda)
    with torch.no_grad():
        model.gemm_mode = 'f16x2'
        h0, x0 = model(g, t, None)                                  # in range: the mode works
        assert model.engine().gemm_mode() == 'f16x2'
        big[3, 5] = 80.0                                            # 80 * 2^10 > 65504 whatever the block's scaling (>= 1 here)
        with pytest.raises(hip.KpdError, match='do not fit the f16x2 mode'):
            model(g, t, None)
        model.gemm_mode = 'f32'
        h1, x1 = model(g, t, None)                                  # the exact mode is unaffected
        assert torch.isfinite(h1).all() and torch.isfinite(x1).all() and model.engine().gemm_mode() == 'f32'
        monkeypatch.setenv('KPD_GEMM', 'f16x2')                     # asked for at creation: the commit refuses
        model.gemm_mode = None
        big[3, 5] = 81.0
        with pytest.raises(hip.KpdError, match='do not fit the f16x2 mode'):
            model(g, t, None)


def test_explicit_gemm_mode_survives_engine_rebuilds(cuda, monkeypatch):
    monkeypatch.delenv('KPD_GEMM', raising=False)
    model = synth.fill_state_dict_(LigRecDynamics(10, 10, graph_cutoffs=CUT, **util.EGNN_C2), 0).eval().to(cuda)
    g = _batch(cuda)
    t = torch.tensor([0.3, 0.8], device=cuda)
    with torch.no_grad():
        assert model.engine().gemm_mode() == 'f32'                 # library default
        model.gemm_mode = 'f16x2'
        model(g, t, None)
        first = model.engine()
        model.lig_decoder[2].bias.add_(0.25)                        # in-place update: the engine is rebuilt ...
        model(g, t, None)
        assert model.engine() is not first and model.engine().gemm_mode() == 'f16x2'      # ... in the chosen mode
        model.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
        model(g, t, None)
        assert model.engine().gemm_mode() == 'f16x2'
        model.gemm_mode = 'f32'
        model(g, t, None)
        assert model.engine().gemm_mode() == 'f32'


def test_explicit_f16x2_on_a_narrow_gvp_model_raises_instead_of_falling_back(cuda, monkeypatch):
    """The f16x2 mode of the GVP denoiser exists for 256 hidden scalars only; the library stores f32 for a narrower model.  An
    explicit `gemm_mode = 'f16x2'` must then raise, not run another mode silently."""
    monkeypatch.delenv('KPD_GEMM', raising=False)
    cfg = dict(GVP_ALL_ATOM, n_hidden_scalars=64)
    model = synth.fill_state_dict_(LigRecDynamicsGVP(10, 10, graph_cutoffs=CUT, **cfg), 0).eval().to(cuda)
    g = _batch(cuda, v=16)
    t = torch.tensor([0.3, 0.8], device=cuda)
    with torch.no_grad():
        model(g, t, None)
        model.gemm_mode = 'f16x2'
        with pytest.raises(hip.KpdError, match='was requested but the engine runs'):
            model(g, t, None)
        model.gemm_mode = 'f32'
        model(g, t, None)


def test_a_swapped_parameter_object_rebuilds_the_engine_at_once(cuda, monkeypatch):
    """`module.weight = nn.Parameter(...)` deep inside the denoiser replaces the Parameter OBJECT: the cached parameter list of the
    engine key must not hide it (it used to, for up to 63 forwards)."""
    monkeypatch.delenv('KPD_GEMM', raising=False)
    model = synth.fill_state_dict_(LigRecDynamics(10, 10, graph_cutoffs=CUT, **dict(util.EGNN_C2, n_layers=2)), 0).eval().to(cuda)
    g = _batch(cuda)
    t = torch.tensor([0.3, 0.8], device=cuda)
    with torch.no_grad():
        h0, _ = model(g, t, None)
        first = model.engine()
        lin = model.lig_decoder[2]
        lin.bias = torch.nn.Parameter(lin.bias.detach() + 0.5)
        h1, _ = model(g, t, None)
    assert model.engine() is not first
    assert torch.allclose(h1, h0 + 0.5, atol=1e-5)
