"""Gradients of one step at identical weights / batch: RVIP_BWD_PAIR=0 (two launches, fork / join) vs 1 (one launch) at the same cu_limit:
must be bit-identical.  python tools/ab_pair.py [precision] [dim] [batch]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmr_landmark_detection_amd as rvip
from oracle import rvip_oracle as O
M = rvip.Loss_and_metrics
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
cfg = dict(DIM=[dim, dim], FILTERS=32, DEPTH=4, BATCH_NORMALISATION=True, ACTIVATION='relu', MASK_CLASSES=2, LEARNING_RATE=1e-3, RVIP_PRECISION=prec, LOSS_FUNCTION=M.mse, SEED=11)
x, y = O.synthetic_batch(B, cfg['DIM'], 2, seed=12)
res = {}
for tag, env in (('two', {'RVIP_BWD_PAIR': '0', 'RVIP_BWD_UNPAIRED': 'forkjoin'}), ('one', {'RVIP_BWD_PAIR': '1', 'RVIP_BWD_UNPAIRED': 'forkjoin'}), ('serial', {'RVIP_BWD_OVERLAP': '0'})):
    for k in ('RVIP_BWD_PAIR', 'RVIP_BWD_OVERLAP', 'RVIP_BWD_UNPAIRED'):
        os.environ.pop(k, None)
    os.environ.update(env)
    model = rvip.get_model(cfg, metrics=[])
    eng = model._engine(B)
    eng.load_input(x, y)
    for _ in range(2):
        eng.forward(training=True)
        eng.backward()
    torch.cuda.synchronize()
    res[tag] = (float(eng.loss.item()), model._params.grads_host(), list(eng.paired))
    model.close()
print(prec, dim, B, 'paired layers:', res['one'][2])
for tag in ('one', 'serial'):
    worst = {}
    for k, g in res['two'][1].items():
        d = float(np.abs(res[tag][1][k] - g).max() / (np.abs(g).max() + 1e-30))
        worst[k[1]] = max(worst.get(k[1], 0.0), d)
    print(tag, 'vs two launches: loss', res[tag][0], res['two'][0], {k: '%.2e' % v for k, v in worst.items()})
