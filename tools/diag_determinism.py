"""Diagnostic: repeat fwd+bwd on identical state; report which gradient/activation tensors differ between repeats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import cmr_landmark_detection_amd as rvip
from oracle import rvip_oracle as O
import test_gpu_model as T
M = rvip.Loss_and_metrics
dim = [int(sys.argv[1]), int(sys.argv[2])] if len(sys.argv) > 2 else [48, 48]
cfg = T._cfg(DEPTH=3, DIM=dim, LOSS_FUNCTION=M.bce_dice_loss)
B = 4
model = rvip.get_model(cfg, metrics=[])
x, y = O.synthetic_batch(B, cfg['DIM'], 2, seed=3)
eng = model._engine(B)
snaps = []
for rep in range(6):
    eng.load_input(x, y); eng.forward(True); eng.backward(); torch.cuda.synchronize()
    snap = {'grad': model._params.grad.clone()}
    for k, t in eng.act.items(): snap['act:' + k] = t.clone()
    for k, t in eng.grd.items(): snap['grd:' + k] = t.clone()
    for k, t in eng.dz.items(): snap['dz:' + k] = t.clone()
    for k, t in eng.gskip.items(): snap['gskip:' + k] = t.clone()
    snaps.append(snap)
order = ['act:' + st.z for st in model.plan.stages] + ['grd:' + model.plan.head['src']]
for st in reversed(model.plan.stages):
    order += ['grd:' + st.y, 'dz:' + st.z]
seen = set()
for rep in range(1, 6):
    bad = [k for k in snaps[0] if not torch.equal(snaps[0][k], snaps[rep][k])]
    first = [k for k in order if k in bad]
    print('rep', rep, 'tensors differing from rep 0:', len(bad), 'first in execution order:', first[:4])
    for k in first[:3]:
        d = (snaps[0][k].float() - snaps[rep][k].float()).abs()
        print('     ', k, 'max abs diff %.3e' % d.max().item(), 'count', int((d > 0).sum().item()), 'of', d.numel(), 'scale %.3e' % snaps[0][k].float().abs().max().item())
