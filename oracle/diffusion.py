"""CPU restatement of the diffusion-side arithmetic around the denoiser (test infrastructure
only).  Follows models/ligand_diffuser.py: PredefinedNoiseSchedule :654-690,
polynomial_schedule :636-650, clip_noise_schedule :620-633, sigma/alpha :232-238,
sigma_and_alpha_t_given_s :240-252, remove_com :185-203, sample_p_zs_given_zt :497-538.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import graph_ops as G
from .batch import OBatch


def gamma_table(timesteps: int, precision: float, power: float = 2.0) -> torch.Tensor:
    steps = timesteps + 1
    xs = np.linspace(0, steps, steps)
    alphas2 = (1 - np.power(xs / steps, power)) ** 2
    alphas2 = np.concatenate([np.ones(1), alphas2], axis=0)
    step = np.clip(alphas2[1:] / alphas2[:-1], a_min=0.001, a_max=1.)
    alphas2 = np.cumprod(step, axis=0)
    alphas2 = (1 - 2 * precision) * alphas2 + precision
    sigmas2 = 1 - alphas2
    return torch.from_numpy(-(np.log(alphas2) - np.log(sigmas2))).float()


def gamma_at(table: torch.Tensor, t: torch.Tensor, timesteps: int) -> torch.Tensor:
    return table[torch.round(t * timesteps).long()]                            # :688-690


def sigma(gamma):
    return torch.sqrt(torch.sigmoid(gamma))


def alpha(gamma):
    return torch.sqrt(torch.sigmoid(-gamma))


def sigma_and_alpha_t_given_s(gamma_t, gamma_s):
    sigma2 = -torch.expm1(F.softplus(gamma_s) - F.softplus(gamma_t))
    log_a2 = F.logsigmoid(-gamma_t) - F.logsigmoid(-gamma_s)
    return sigma2, torch.sqrt(sigma2), torch.exp(0.5 * log_a2)


def remove_com(batch: OBatch, com: str) -> OBatch:
    nt = {'ligand': 'lig', 'receptor': 'kp'}[com]
    c = G.segment_mean_nodes(batch.x[nt], batch.n[nt])
    batch.x['lig'] = batch.x['lig'] - c[G.counts_to_batch_idx(batch.n['lig'])]
    batch.x['kp'] = batch.x['kp'] - c[G.counts_to_batch_idx(batch.n['kp'])]
    return batch


def sample_step(batch: OBatch, eps_h, eps_x, s, t, table, timesteps, pos_noise, feat_noise) -> OBatch:
    """One reverse step given the denoiser output and the fresh noise (:505-536)."""
    g_s, g_t = gamma_at(table, s, timesteps), gamma_at(table, t, timesteps)
    s2_ts, s_ts, a_ts = sigma_and_alpha_t_given_s(g_t, g_s)
    sig_s, sig_t = sigma(g_s), sigma(g_t)
    lb = G.counts_to_batch_idx(batch.n['lig'])
    var = (s2_ts / a_ts / sig_t)[lb].view(-1, 1)
    a = a_ts[lb].view(-1, 1)
    mu_x = batch.x['lig'] / a - var * eps_x
    mu_h = batch.h['lig'] / a - var * eps_h
    sg = (s_ts * sig_s / sig_t)[lb].view(-1, 1)
    batch.x['lig'] = mu_x + sg * pos_noise
    batch.h['lig'] = mu_h + sg * feat_noise
    return remove_com(batch, 'ligand')
