"""Diagnostic: after N Adam steps, where do the bf16 device model and the float64 oracle differ at inference?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import test_gpu_model as T
rvip, O = T.rvip, T.O

for prec in ('fp32', 'bf16'):
    cfg = T._cfg(RVIP_PRECISION=prec, FILTERS=16, DIM=[64, 64], LEARNING_RATE=1e-3)
    model = rvip.get_model(cfg, metrics=[])
    ref, layers = T._oracle_from(model, cfg)
    B = 4
    x, y = O.synthetic_batch(B, cfg['DIM'], 2, seed=12)
    for step in range(15):
        dl = model.train_on_batch(x, y)[0]
        lv, _ = ref.train_step(x.astype(np.float64), y.astype(np.float64), 'mse', T._masks(layers, B, model.seed, step))
    print(prec, 'final losses', dl, lv)
    # weights
    names = []
    for l in layers:
        if l['type'].startswith('Conv'): names += [(l['name'], 'k'), (l['name'], 'b')]
        elif l['type'] == 'BatchNormalization': names += [(l['name'], s) for s in ('gamma', 'beta', 'mmean', 'mvar')]
    refw = []
    for l in layers:
        if l['name'] in ref.params: refw += list(ref.params[l['name']])
    for (n, s), a, b in zip(names, model.get_weights(), refw):
        d = np.abs(a - b).max(); sc = np.abs(b).max()
        if s in ('mmean', 'mvar') or d > 1e-2 * max(sc, 1e-3):
            print(f'  {n:28s} {s:6s} maxdiff {d:.3e}  scale {sc:.3e}')
    xt, _ = O.synthetic_batch(2, cfg['DIM'], 2, seed=13)
    pr = ref.predict(xt.astype(np.float64)); pd = model.predict(xt)
    diff = np.abs(pd - pr)
    print(prec, 'fresh predict: mean', diff.mean(), 'median', np.median(diff), 'max', diff.max(), 'pred range', pr.min(), pr.max())
    # oracle weights loaded into a fresh device model -> isolates inference from training drift
    m2 = rvip.get_model(cfg, metrics=[]); m2.set_weights([np.asarray(w, np.float32) for w in refw])
    d2 = np.abs(m2.predict(xt) - pr)
    print(prec, 'oracle weights on device: mean', d2.mean(), 'max', d2.max())
    pdx = np.abs(model.predict(x) - ref.predict(x.astype(np.float64)))
    print(prec, 'train-batch predict: mean', pdx.mean(), 'max', pdx.max())
