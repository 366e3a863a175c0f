import sys, os, numpy as np, importlib, torch
sys.path.insert(0, os.getcwd())
import cmr_landmark_detection_amd as rvip
from oracle import rvip_oracle as O
M = rvip.Loss_and_metrics
tag = sys.argv[1]
cfg = dict(DIM=[224, 224], FILTERS=32, DEPTH=4, BATCH_NORMALISATION=True, BN_FIRST=False, ACTIVATION='relu', MASK_CLASSES=2, M_POOL=[2, 2], F_SIZE=[3, 3], LEARNING_RATE=1e-3, RVIP_PRECISION='fp16', LOSS_FUNCTION=M.mse, SEED=42)
model = rvip.get_model(cfg, metrics=[])
x, y = O.synthetic_batch(2, cfg['DIM'], 2, seed=42)
p = model.predict(x)
eng = model._engine(2)
torch.cuda.synchronize()
out = {'pred': p}
for k, t in eng.act.items():
    out[k] = t.float().cpu().numpy()
np.savez('/tmp/acts_%s.npz' % tag, **out)
print(tag, 'done', float(np.abs(p).mean()))
