"""Input pipeline on the device.

Mirrors `build_initial_complex_graph` (data_processing/pdbbind_processing.py:221-274) and `ProteinLigandDataset`
(data_processing/crossdocked/dataset.py:16-146, 187-199) for the flat processed-dataset layout (`lig_pos, lig_feat,
rec_pos, rec_feat, interface_points, rec_res_idx, *_segments, rec_files, lig_files`).  The flat arrays are uploaded
once and stay resident in HBM; a batch of complexes is cut out of them with index arithmetic and its receptor
radius graph + same-residue flags are built for the whole batch in one launch sequence (`kpd_build_rec_graph`), so
evaluation loops in the style of test.py run without DGL, torch_cluster or per-complex host work.  No CPU
implementation: the HIP library must be present.
"""
import pickle
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Union

import torch

from . import graph as G
from . import hip


def _complex_graphs(rec_pos, rec_feat, rec_res_idx, rec_counts: List[int], n_keypoints: int, cutoffs: dict,
                    lig_pos, lig_feat, lig_counts: List[int]) -> G.HeteroBatch:
    """Batched graph of B complexes given flat (already gathered) node arrays on the GPU."""
    dev = rec_pos.device
    B = len(rec_counts)
    n_rec, n_lig, n_kp = sum(rec_counts), sum(lig_counts), B * n_keypoints
    rec_n = torch.tensor(rec_counts, dtype=torch.long)
    rec_ptr = torch.zeros(B + 1, dtype=torch.long)
    rec_ptr[1:] = rec_n.cumsum(0)
    g = G.HeteroBatch({'rec': n_rec, 'kp': n_kp, 'lig': n_lig}, device=dev)
    g.set_batch_num_nodes({'rec': rec_n, 'kp': torch.full((B,), n_keypoints, dtype=torch.long),
                           'lig': torch.tensor(lig_counts, dtype=torch.long)})
    if n_rec:
        res = rec_res_idx.to(torch.int32) if rec_res_idx is not None else None
        src, dst, per_graph, same = hip.build_rec_graph(rec_pos, rec_ptr.to(torch.int32).to(dev), max(rec_counts), cutoffs['rr'],
                                                        res_idx=res, max_nn=100)                    # :245, :248
        g._edges['rr'] = (src.long(), dst.long())
        rr_counts = per_graph.long()
        if same is not None:
            g._edata['rr']['same_res'] = same.view(-1, 1)                                           # :272
    else:
        rr_counts = torch.zeros(B, dtype=torch.long, device=dev)
    # complete rec -> kp edges, keypoint-major inside every complex (:251-253)
    rep = (rec_n * n_keypoints).to(dev)
    cid = torch.repeat_interleave(torch.arange(B, device=dev), rep)
    start = torch.zeros(B + 1, dtype=torch.long, device=dev)
    start[1:] = rep.cumsum(0)
    local = torch.arange(int(start[-1]), device=dev) - start[cid]
    nr = rec_n.to(dev)[cid]
    g._edges['rk'] = (rec_ptr.to(dev)[cid] + local % nr.clamp(min=1), cid * n_keypoints + local // nr.clamp(min=1))
    g.set_batch_num_edges({'rr': rr_counts, 'rk': rep, **{et: torch.zeros(B, dtype=torch.long) for et in ('kk', 'kl', 'll', 'lk')}})
    g.nodes['rec'].data['x_0'] = rec_pos
    g.nodes['rec'].data['h_0'] = rec_feat
    if lig_pos is not None:
        g.nodes['lig'].data['x_0'] = lig_pos
        g.nodes['lig'].data['h_0'] = lig_feat
    return g


def build_initial_complex_graph(rec_atom_positions: torch.Tensor, rec_atom_features: torch.Tensor, pocket_res_idx: torch.Tensor,
                                n_keypoints: int, cutoffs: dict, lig_atom_positions: torch.Tensor = None,
                                lig_atom_features: torch.Tensor = None) -> G.HeteroBatch:
    """pdbbind_processing.py:221-274 for one complex, tensors on the GPU."""
    if (lig_atom_positions is not None) ^ (lig_atom_features is not None):
        raise ValueError('ligand position and features must be either be both supplied or both left as None')
    n_lig = 0 if lig_atom_positions is None else lig_atom_positions.shape[0]
    return _complex_graphs(rec_atom_positions, rec_atom_features, pocket_res_idx, [rec_atom_positions.shape[0]], n_keypoints,
                           cutoffs, lig_atom_positions, lig_atom_features, [n_lig])


class ProteinLigandDataset:
    """crossdocked/dataset.py:16-146 on resident device arrays.  `processed_data_file` is the flat pickle the reference's
    processing scripts write (or the same dict, already loaded)."""

    def __init__(self, name: str, processed_data_file: Union[str, Path, dict], rec_elements: List[str], lig_elements: List[str],
                 n_keypoints: int, graph_cutoffs: dict, lig_box_padding: Union[int, float] = 6,
                 pocket_cutoff: Union[int, float] = 4, receptor_k: int = 3, load_data: bool = True,
                 use_boltzmann_ot: bool = False, max_fake_atom_frac: float = 0.0, device='cuda', **kwargs):
        if max_fake_atom_frac > 0:
            raise NotImplementedError('fake atoms are unused by every shipped config (max_fake_atom_frac: 0.0)')
        self.name = name
        self.max_fake_atom_frac = max_fake_atom_frac
        self.n_keypoints = n_keypoints
        self.graph_cutoffs = graph_cutoffs
        self.load_data = load_data
        self._data = processed_data_file if isinstance(processed_data_file, dict) else None
        self.data_file = Path('in_memory.pkl' if self._data is not None else processed_data_file)
        self.device = torch.device(device)
        self.rec_elements = rec_elements
        self.rec_element_map: Dict[str, int] = {element: idx for idx, element in enumerate(rec_elements)}
        self.rec_element_map['other'] = len(rec_elements)
        self.lig_elements = lig_elements
        self.lig_element_map: Dict[str, int] = {element: idx for idx, element in enumerate(lig_elements)}
        self.lig_element_map['other'] = len(lig_elements)
        self.lig_reverse_map = {v: k for k, v in self.lig_element_map.items()}
        self.lig_box_padding, self.pocket_cutoff, self.use_boltzmann_ot = lig_box_padding, pocket_cutoff, use_boltzmann_ot
        self.process()

    def process(self):
        if not self.load_data:
            self.lig_segments = torch.tensor([0])
            return
        data = self._data
        if data is None:
            with open(self.data_file, 'rb') as f:
                data = pickle.load(f)
        dev = self.device
        if dev.type != 'cuda':
            raise hip.KpdError('the dataset arrays must live on the GPU; the input pipeline has no CPU implementation')
        up = lambda t: torch.as_tensor(t).to(dev)
        self.lig_pos, self.lig_feat = up(data['lig_pos']).float(), up(data['lig_feat']).float()      # .float(): dataset.py:128-129
        self.rec_pos, self.rec_feat = up(data['rec_pos']).float(), up(data['rec_feat']).float()
        self.interface_points = up(data['interface_points'])
        self.rec_res_idx = up(data['rec_res_idx']).to(torch.int32)
        # segment tables stay on the host: they drive slicing and shapes, never device work
        self.rec_segments = torch.as_tensor(data['rec_segments']).long().cpu()
        self.lig_segments = torch.as_tensor(data['lig_segments']).long().cpu()
        self.ip_segments = torch.as_tensor(data['ip_segments']).long().cpu()
        self.rec_files, self.lig_files = data.get('rec_files'), data.get('lig_files')

    def __len__(self):
        return self.lig_segments.shape[0] - 1

    def _rows(self, seg: torch.Tensor, idxs: Sequence[int]):
        lo, hi = seg[idxs], seg[[i + 1 for i in idxs]]
        counts = (hi - lo).tolist()
        rows = torch.cat([torch.arange(int(a), int(b)) for a, b in zip(lo.tolist(), hi.tolist())]) if idxs else torch.zeros(0, dtype=torch.long)
        return rows.to(self.device), counts

    def get_batch(self, idxs: Sequence[int]):
        """The batched graph of complexes `idxs` (what collate_fn(dataset[i] for i in idxs) returns, dataset.py:187-194),
        built in one launch sequence, and their interface points."""
        idxs = [int(i) % len(self) if int(i) < 0 else int(i) for i in idxs]
        rr, rc = self._rows(self.rec_segments, idxs)
        lr, lc = self._rows(self.lig_segments, idxs)
        g = _complex_graphs(self.rec_pos[rr], self.rec_feat[rr], self.rec_res_idx[rr], rc, self.n_keypoints, self.graph_cutoffs,
                            self.lig_pos[lr], self.lig_feat[lr], lc)
        ips = tuple(self.interface_points[int(self.ip_segments[i]):int(self.ip_segments[i + 1])] for i in idxs)
        return g, ips

    def __getitem__(self, i):
        g, ips = self.get_batch([i])
        return g, ips[0]

    def lig_atom_idx_to_element(self, element_idxs: List[int]):
        return [self.lig_reverse_map[element_idx] for element_idx in element_idxs]

    @property
    def type_counts_file(self) -> Path:
        dataset_split = self.data_file.name.split('_')[0]
        return self.data_file.parent / f'{dataset_split}_type_counts.pkl'

    @property
    def dataset_dir(self) -> Path:
        return self.data_file.parent

    def get_files(self, idx: int):
        return self.rec_files[idx], self.lig_files[idx]


def collate_fn(examples: list):
    """dataset.py:187-194."""
    complex_graphs, interface_points = zip(*examples)
    return G.batch(list(complex_graphs)), interface_points


class _BatchLoader:
    def __init__(self, dataset: ProteinLigandDataset, batch_size: int, shuffle: bool = False, generator: Optional[torch.Generator] = None):
        self.dataset, self.batch_size, self.shuffle, self.generator = dataset, batch_size, shuffle, generator

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.dataset)
        order = torch.randperm(n, generator=self.generator).tolist() if self.shuffle else list(range(n))
        for a in range(0, n, self.batch_size):
            yield self.dataset.get_batch(order[a:a + self.batch_size])


def get_dataloader(dataset: ProteinLigandDataset, batch_size: int, num_workers: int = 1, shuffle: bool = False, **kwargs) -> _BatchLoader:
    """dataset.py:196-199: batches of complexes (drop_last=False).  The arrays are resident on the GPU, so there are no
    worker processes; `num_workers` is accepted and ignored."""
    return _BatchLoader(dataset, batch_size, shuffle=shuffle, generator=kwargs.get('generator'))
