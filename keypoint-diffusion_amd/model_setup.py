"""YAML config -> model, with the mapping of the reference's model_setup.py:4-63 (sections are
splatted as constructor kwargs; missing `architecture` => 'egnn', missing `rec_encoder_type` =>
'learned'; unknown keys such as `no_cg` / `n_keypoints` are accepted by the constructors)."""
from pathlib import Path

from .ligand_diffuser import KeypointDiffusion


def model_from_config(config: dict, require_dataset_dir: bool = True) -> KeypointDiffusion:
    diffusion = dict(config['diffusion'])
    architecture = diffusion.get('architecture', 'egnn')
    rec_encoder_type = diffusion.get('rec_encoder_type', 'learned')
    use_fake_atoms = config['dataset'].get('max_fake_atom_frac', 0) > 0
    n_rec_feat = len(config['dataset']['rec_elements'])
    # C-alpha datasets carry a 20-way residue one-hot although rec_elements lists 10 entries; saved
    # configs record the true width under `reconstruction` (trained_models/egnn_ca/config.yml)
    n_rec_feat = config.get('reconstruction', {}).get('n_rec_atom_feat', n_rec_feat)
    n_lig_feat = len(config['dataset']['lig_elements']) + (1 if use_fake_atoms else 0)
    if rec_encoder_type == 'learned':
        n_kp_feat = config['rec_encoder']['out_n_node_feat'] if architecture == 'egnn' \
            else config['rec_encoder_gvp']['out_scalar_size']
    else:
        n_kp_feat = n_rec_feat
    if architecture == 'gvp':
        rec_cfg = dict(config['rec_encoder_gvp'], in_scalar_size=n_rec_feat)
        dyn_cfg = config['dynamics_gvp']
    else:
        rec_cfg = dict(config.get('rec_encoder', {}), in_n_node_feat=n_rec_feat)
        dyn_cfg = config['dynamics']
    dataset_dir = Path(config['dataset']['location'])
    if not require_dataset_dir and not (dataset_dir / 'train_n_node_joint_dist.pkl').exists():
        dataset_dir = None
    return KeypointDiffusion(n_lig_feat, n_kp_feat, processed_dataset_dir=dataset_dir,
                             graph_config=config['graph'], dynamics_config=dyn_cfg, rec_encoder_config=rec_cfg,
                             rec_encoder_loss_config=config.get('rec_encoder_loss', {}), use_fake_atoms=use_fake_atoms,
                             **diffusion)
