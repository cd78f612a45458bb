"""One rank of the sharded-sampling rehearsal: `python -m tests.sharded_worker OUT.pt` with RANK / WORLD_SIZE / MASTER_* in the
environment.  Every rank uses cuda:0 and the process group runs over gloo (RCCL refuses two ranks on one device); the code
path -- KeypointDiffusion._sample under an initialised process group -- is the one an 8-GPU run takes over RCCL."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_REC = [90, 140, 60, 75, 120]
N_LIG = [[11, 17], [9], [14, 6, 12], [8, 8, 15], [10, 13]]          # 11 complexes: more than 8 ranks, fewer than 2 per rank
SEED = 99
T = 8


def build_model(device):
    from keypoint_diffusion_amd import synth
    from keypoint_diffusion_amd.ligand_diffuser import KeypointDiffusion
    from tests import util
    m = KeypointDiffusion(10, 10, None, n_timesteps=T, architecture='egnn', rec_encoder_type='fixed',
                          graph_config=dict(n_keypoints=20, graph_cutoffs=util.CUTOFFS_ALL_ATOM), dynamics_config=util.EGNN_C2,
                          rec_encoder_config={}, precision=1e-5)
    synth.fill_state_dict_(m, 13)
    return m.eval().to(device)


def pockets(device):
    from keypoint_diffusion_amd import synth
    from tests import util
    out = []
    for g in synth.synth_complexes(N_REC, [1] * len(N_REC), 20, util.CUTOFFS_ALL_ATOM, seed=3):
        g = g.to(device)
        g.remove_nodes(g.nodes('lig'), ntype='lig')
        out.append(g)
    return out


def run_rank(dev):
    """What one rank of the job does once its process group is up (the body shared by the process ranks below and by the
    thread ranks of `tests/util.py::run_threaded_world`)."""
    model = build_model(dev).use_complex_noise(SEED)
    return model._sample(pockets(dev), N_LIG, rec_enc_batch_size=2, diff_batch_size=2)


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo', rank=rank, world_size=world)
    samples = run_rank(torch.device('cuda:0'))
    torch.save({'rank': rank, 'samples': samples}, f'{sys.argv[1]}.{rank}')
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
