"""(n_rec, n_lig) pairs drawn from the reference's SHIPPED size statistics -> tests/golden/size_pairs.json.

Run in the build container only (it reads /root/reference/data/*/train_n_node_joint_dist.pkl -- data files: the smoothed joint
histogram of receptor-pocket and ligand atom counts of the training split, what models/n_nodes_dist.py:14-22 loads; no reference code
is imported).  The JSON holds sizes only: for each dataset the histogram's bounds and 512 pairs drawn from it with a seeded numpy
generator (row = receptor size, column = ligand size, probability = the histogram entry).  tests/test_size_sweep_gpu.py builds its
randomized parity batches from them."""
import json
import os
import pickle

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/data'
DATASETS = {'all_atom': 'bindingmoad_processed', 'ca': 'bindingmoad_ca'}


def main():
    out = {}
    for tag, d in DATASETS.items():
        with open(os.path.join(REF, d, 'train_n_node_joint_dist.pkl'), 'rb') as f:
            hist, rec_bounds, lig_bounds = pickle.load(f)
        hist = np.asarray(hist, dtype=np.float64)
        assert hist.shape == (rec_bounds[1] - rec_bounds[0] + 1, lig_bounds[1] - lig_bounds[0] + 1)
        p = hist.ravel() / hist.sum()
        rng = np.random.default_rng(20260101)
        flat = rng.choice(p.size, size=512, p=p)
        n_rec = rec_bounds[0] + flat // hist.shape[1]
        n_lig = lig_bounds[0] + flat % hist.shape[1]
        marg_rec = hist.sum(1) / hist.sum()
        marg_lig = hist.sum(0) / hist.sum()
        out[tag] = {'source': f'data/{d}/train_n_node_joint_dist.pkl', 'rec_bounds': [int(b) for b in rec_bounds],
                    'lig_bounds': [int(b) for b in lig_bounds],
                    'mean_n_rec': float((marg_rec * np.arange(rec_bounds[0], rec_bounds[1] + 1)).sum()),
                    'mean_n_lig': float((marg_lig * np.arange(lig_bounds[0], lig_bounds[1] + 1)).sum()),
                    'pairs': [[int(a), int(b)] for a, b in zip(n_rec, n_lig)]}
    with open(os.path.join(HERE, 'size_pairs.json'), 'w') as f:
        json.dump(out, f, separators=(',', ':'))
    for tag, v in out.items():
        print(tag, v['rec_bounds'], v['lig_bounds'], round(v['mean_n_rec'], 1), round(v['mean_n_lig'], 1), v['pairs'][:4])


if __name__ == '__main__':
    main()
