"""Gradients of one step at identical weights / batch under RVIP_BWD_OVERLAP = 0 and W (and W with the second stream disabled):
max relative deviation per tensor kind.  python tools/ab_overlap.py [precision] [W]"""
import os, sys, numpy as np, importlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmr_landmark_detection_amd as rvip
from oracle import rvip_oracle as O
M = rvip.Loss_and_metrics
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
W = sys.argv[2] if len(sys.argv) > 2 else '128'
cfg = dict(DIM=[64, 64], FILTERS=32, DEPTH=3, BATCH_NORMALISATION=True, ACTIVATION='relu', MASK_CLASSES=2, LEARNING_RATE=1e-3, RVIP_PRECISION=prec, LOSS_FUNCTION=M.mse, SEED=11)
x, y = O.synthetic_batch(4, cfg['DIM'], 2, seed=12)
res = {}
for tag, env in (('base', {'RVIP_BWD_OVERLAP': '0'}), ('ovl', {'RVIP_BWD_OVERLAP': W}), ('ovl_serial', {'RVIP_BWD_OVERLAP': W, 'RVIP_BWD_OVERLAP_SERIAL': '1'}), ('ovl2', {'RVIP_BWD_OVERLAP': W})):
    for k in ('RVIP_BWD_OVERLAP', 'RVIP_BWD_OVERLAP_SERIAL'):
        os.environ.pop(k, None)
    os.environ.update(env)
    model = rvip.get_model(cfg, metrics=[])
    eng = model._engine(4)
    eng.load_input(x, y)
    eng.forward(training=True)
    eng.backward()
    torch.cuda.synchronize()
    res[tag] = (float(eng.loss.item()), model._params.grads_host())
    model.close()
for tag in ('ovl', 'ovl_serial', 'ovl2'):
    worst = {}
    for k, g in res['base'][1].items():
        d = float(np.abs(res[tag][1][k] - g).max() / (np.abs(g).max() + 1e-30))
        worst[k[1]] = max(worst.get(k[1], 0.0), d)
    print(prec, tag, 'loss', res[tag][0], res['base'][0], {k: '%.2e' % v for k, v in worst.items()})
