"""Diagnostic (not a test): per-layer gradient error of one config vs the oracle, several steps."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmr_landmark_detection_amd as rvip
from oracle import rvip_oracle as O
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_gpu_model as T
M = rvip.Loss_and_metrics
variant = dict(DEPTH=3, DIM=[48, 40], LOSS_FUNCTION=M.bce_dice_loss)
if len(sys.argv) > 1 and sys.argv[1] == 'mse':
    variant['LOSS_FUNCTION'] = M.mse
if len(sys.argv) > 2:
    variant['DIM'] = [int(sys.argv[2]), int(sys.argv[3])]
cfg = T._cfg(**variant)
kind = M.resolve_loss(cfg['LOSS_FUNCTION'])
B = 4
model = rvip.get_model(cfg, metrics=[])
ref, layers = T._oracle_from(model, cfg)
x, y = O.synthetic_batch(B, cfg['DIM'], 2, seed=3)
x64, y64 = x.astype(np.float64), y.astype(np.float64)
eng = model._engine(B)
for step in range(3):
    ref.set_weights(model.get_weights())
    masks = T._masks(layers, B, model.seed, step)
    eng.load_input(x, y); eng.forward(True); eng.backward(); torch.cuda.synchronize()
    rpred, cache = ref.forward(x64, True, masks)
    if kind[0] == 'mse':
        lv, dp = O.mse_loss(y64, rpred); rg = ref.backward(cache, dp)
    else:
        lv, dl = O.bce_dice_loss(y64, rpred, w_bce=kind[1], w_dice=kind[2], logits=cache['logits']); rg = ref.backward(cache, dl, d_is_logit_grad=True)
    got = model._params.grads_host()
    print('step', step, 'loss', float(eng.loss.item()), lv, 'pred err', np.abs(eng.pred.cpu().numpy() - rpred).max())
    dlg = eng.dlogit.cpu().numpy()
    if kind[0] != 'mse':
        print('   dlogit rel err', np.abs(dlg - dl).max() / np.abs(dl).max())
    for lname, gs in rg.items():
        for i, g in enumerate(gs):
            wn = ('kernel', 'bias')[i] if (lname.startswith('conv') or lname == 'unet') else ('gamma', 'beta')[i]
            e = np.abs(got[(lname, wn)] - g).max() / max(np.abs(g).max(), 1e-12)
            if e > 5e-5:
                print('   %-24s %-6s rel err %.2e  (max|g| %.2e)' % (lname, wn, e, np.abs(g).max()))
    eng.optimizer_step(); torch.cuda.synchronize()
