"""Diagnostic: run the default tiny config with and without fused BN statistics; report the first tensors that differ."""
import sys, os, subprocess, pickle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
if len(sys.argv) > 1:
    import numpy as np, torch
    import cmr_landmark_detection_amd as rvip
    from oracle import rvip_oracle as O
    import test_gpu_model as T
    cfg = T._cfg(); B = 4
    model = rvip.get_model(cfg, metrics=[])
    x, y = O.synthetic_batch(B, cfg['DIM'], 2, seed=3)
    eng = model._engine(B)
    eng.load_input(x, y); eng.forward(True)
    torch.cuda.synchronize()
    snap = {'A:' + k: t.float().cpu().numpy().copy() for k, t in eng.act.items()}
    snap['bn_scratch_fwd'] = eng.bn_scratch.cpu().numpy().copy()
    eng.backward(); torch.cuda.synchronize()
    snap.update({'A2:' + k: t.float().cpu().numpy().copy() for k, t in eng.act.items()})
    snap.update({'G:' + k: t.float().cpu().numpy().copy() for k, t in eng.grd.items()})
    snap.update({'DZ:' + k: t.float().cpu().numpy().copy() for k, t in eng.dz.items()})
    snap.update({'GS:' + k: t.float().cpu().numpy().copy() for k, t in eng.gskip.items()})
    snap['bn_scratch_bwd'] = eng.bn_scratch.cpu().numpy().copy()
    snap['grad'] = model._params.grad.cpu().numpy().copy()
    order = [st.conv for st in model.plan.stages]
    pickle.dump((snap, order), open(sys.argv[1], 'wb'))
else:
    for mode in ('1', '0'):
        env = dict(os.environ, RVIP_FUSE_STATS=mode)
        subprocess.check_call([sys.executable, __file__, '/tmp/snap_%s.pkl' % mode], env=env)
    import numpy as np
    (a, order), (b, _) = pickle.load(open('/tmp/snap_1.pkl', 'rb')), pickle.load(open('/tmp/snap_0.pkl', 'rb'))
    for k in a:
        d = np.abs(a[k] - b[k]).max()
        s = np.abs(b[k]).max()
        if d > 1e-5 * max(s, 1e-9):
            print('%-40s max diff %.3e (scale %.3e) n_diff %d / %d' % (k, d, s, int((np.abs(a[k] - b[k]) > 1e-5 * max(s, 1e-9)).sum()), a[k].size))
    print('stage order:', order)
