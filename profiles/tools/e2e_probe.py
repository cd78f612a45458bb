"""Per-pass timing of the end-to-end sampling run (encoder / upload / reverse loop): python profiles/tools/e2e_probe.py [--gc]"""
import sys, time, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0')
for wl in ('gvp_40kp', 'egnn_all_atom'):
    w = bench.WORKLOADS[wl]
    model = bench.build_model(dev, wl)
    with torch.no_grad():
        for it in range(4):
            g = bench.raw_batch(64, 300, 25, 4321, dev, wl)
            if w['enc'] != 'learned':
                g = g.to(dev)
            if '--gc' in sys.argv:
                import gc; gc.collect()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            enc = model.encode_receptors(g)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            enc = enc.to(dev)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            pos, feat = model.sample_from_encoded_receptors(enc)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            print(wl, it, 'encode %.1f ms, .to %.1f ms, loop %.3f s' % (1e3 * (t1 - t0), 1e3 * (t2 - t1), t3 - t2), flush=True)
    del model
