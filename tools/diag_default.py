"""Diagnostic: default tiny config, per-layer grad error + BN scratch (mean / invstd) vs oracle at step 0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import cmr_landmark_detection_amd as rvip
from oracle import rvip_oracle as O
import test_gpu_model as T
cfg = T._cfg()
B = 4
model = rvip.get_model(cfg, metrics=[])
ref, layers = T._oracle_from(model, cfg)
x, y = O.synthetic_batch(B, cfg['DIM'], 2, seed=3)
eng = model._engine(B)
masks = T._masks(layers, B, model.seed, 0)
eng.load_input(x, y); eng.forward(True); eng.backward(); torch.cuda.synchronize()
lv, rg, rpred, cache = ref.loss_and_grads(x.astype(np.float64), y.astype(np.float64), 'mse', masks)
print('loss', float(eng.loss.item()), lv, 'pred err', np.abs(eng.pred.cpu().numpy() - rpred).max())
sc = eng.bn_scratch.cpu().numpy()
for st in model.plan.stages:
    if st.bn:
        o, ca = eng.bn_off[st.conv]
        mean, invstd = sc[o:o + st.cout], sc[o + ca:o + ca + st.cout]
        _, ris, rmean, rvar = cache[st.bn]
        print('  %-10s %-24s mean err %.2e  invstd rel err %.2e   z err %.2e' % (st.conv, st.bn, np.abs(mean - rmean).max(), np.abs(invstd / ris - 1).max(),
              np.abs(eng.act[st.z].float().cpu().numpy() - cache['tensors'][st.conv]).max()))
got = model._params.grads_host()
for lname, gs in rg.items():
    g = gs[0]
    wn = 'kernel' if (lname.startswith('conv') or lname == 'unet') else 'gamma'
    print('  grad %-24s rel err %.2e' % (lname, np.abs(got[(lname, wn)] - g).max() / max(np.abs(g).max(), 1e-12)))
