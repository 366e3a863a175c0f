"""Diagnostic: dynamic range of the f16 activation-gradient tensors under the static loss scale (per tensor: max |g|, share
of non-zero elements below the f16 normal range 6.1e-5)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmr_landmark_detection_amd as rvip
M = rvip.Loss_and_metrics
for dim, f, d, b in (([256, 256], 32, 4, 32), ([512, 512], 64, 5, 8)):
    cfg = dict(DIM=dim, FILTERS=f, DEPTH=d, BATCH_NORMALISATION=True, ACTIVATION='relu', MASK_CLASSES=2,
               LEARNING_RATE=1e-4, RVIP_PRECISION='fp16', LOSS_FUNCTION=M.mse, SEED=1)
    model = rvip.get_model(cfg, metrics=[])
    G = rvip.Generators.SyntheticSAXGenerator(b, dict(DIM=dim, BATCHSIZE=b, GAUS=True, SIGMA=2, SHUFFLE=False))
    x, y = G[0]
    eng = model._engine(b)
    for it in range(3):
        eng.load_input(x, y); eng.forward(training=True); eng.backward(); torch.cuda.synchronize()
        if it == 0 or it == 2:
            print(dim, 'step', it, 'loss scale 2^%d' % int(np.log2(eng.loss_scale)), 'loss', float(eng.loss.item()))
            for name, t in list(eng.grd.items()) + [('gskip:' + k, v) for k, v in eng.gskip.items()]:
                if t is None or t.dtype != torch.float16: continue
                a = t.float().abs()
                nz = a > 0
                sub = ((a < 6.1e-5) & nz).sum().item() / max(nz.sum().item(), 1)
                print('   %-28s max %.3e  finite %s  nonzero %.3f  subnormal share %.4f' % (name, a.max().item(), bool(torch.isfinite(t).all()), nz.float().mean().item(), sub))
            am = max(v.float().abs().max().item() for v in eng.act.values() if v is not None and v.dtype == torch.float16)
            print('   activations max', am)
        eng.optimizer_step()
    del model, eng
    torch.cuda.empty_cache()
