"""Operand-level view of the first-forward deviation (diagnostic; -DKPD_EDGE_DBG build): for rows 0..2 of every wave of every tile the
coordinate branch's A-build taps the gathered P rows (ps, pd), the distance, w_r and the LDS row offsets exactly as it consumes them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device('cuda:0')
model = bench.build_model(dev)
g = bench.build_batch(model, 64, 300, 25, 1234, dev)
t = torch.linspace(0.05, 1.0, 64, device=dev)
eng = model.dynamics.engine()
eng.debug('layers=1')
eng.reserve(g.prepared())
eng.debug('edge_dbg=1')
T = 6193
n_row, n_op = T * 64 * 4, T * 3 * 4 * 64 * 12
outs = []
with torch.no_grad():
    for i in range(3):
        model.dynamics(g, t, None)
        buf = eng.debug('edge_dbg', n_row + n_op)
        outs.append((buf[:n_row].view(T, 64, 4).clone(), buf[n_row:].view(T, 3, 4, 64, 12).clone()))
names = ['ps0', 'ps1', 'ps2', 'ps3', 'pd0', 'pd1', 'pd2', 'pd3', 'd', 'w0', 'src_off', 'dst_off']
for i in (0, 2):
    dr = (outs[i][0] != outs[1][0]).any(2)
    do = outs[i][1] != outs[1][1]
    print(f'run {i} vs run 1: row taps differ at (tile, row) {dr.nonzero().tolist()[:8]}; operand taps differ in {int(do.sum())} values')
    idx = do.any(4).any(3).nonzero().tolist()
    for tl, rr, wv in idx[:6]:
        lanes = do[tl, rr, wv].any(1).nonzero().flatten().tolist()
        comps = do[tl, rr, wv].any(0).nonzero().flatten().tolist()
        print(f'   tile {tl} wave {wv} row-in-wave {rr}: {len(lanes)} lanes {lanes[:6]}.., components {[names[c] for c in comps]}')
        l = lanes[0]
        a, b = outs[i][1][tl, rr, wv, l], outs[1][1][tl, rr, wv, l]
        print('      lane', l, 'run', i, [round(float(v), 5) for v in a[:10]], [int(v) for v in a[10:].view(torch.int32)])
        print('      lane', l, 'run 1', [round(float(v), 5) for v in b[:10]], [int(v) for v in b[10:].view(torch.int32)])
        for c0, nm in ((0, 'ps'), (4, 'pd')):
            if any(c in comps for c in range(c0, c0 + 4)):
                for rr2 in range(3):
                    if torch.equal(outs[i][1][tl, rr, wv, :, c0:c0 + 4], outs[1][1][tl, rr2, wv, :, c0:c0 + 4]):
                        print(f'      {nm} of the deviating run equals {nm} of row-in-wave {rr2} of the good run')
                z = int((outs[i][1][tl, rr, wv, :, c0:c0 + 4] == 0).all())
                print(f'      {nm} all zero: {bool(z)}')
