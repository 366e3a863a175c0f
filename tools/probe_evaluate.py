"""Model.evaluate / fit(validation_data=) wall time at config 2 (the reference's fold: 1 426 training slices, 20 % validation).

    python tools/probe_evaluate.py"""
import importlib
import numpy as np
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    rvip = importlib.import_module('cmr-landmark-detection_amd')
    M = rvip.Loss_and_metrics
    cfg = dict(DIM=[256, 256], FILTERS=32, DEPTH=4, BATCH_NORMALISATION=True, ACTIVATION='relu', MASK_CLASSES=2,
               LEARNING_RATE=1e-4, RVIP_PRECISION='bf16', LOSS_FUNCTION=M.mse, SEED=42)
    model = rvip.get_model(cfg, metrics=[M.dice_coef_labels])
    g = dict(DIM=cfg['DIM'], BATCHSIZE=32, GAUS=True, SIGMA=2, SHUFFLE=True, SEED=42)
    train = rvip.Generators.SyntheticSAXGenerator(1408, g, in_memory=True)
    val = rvip.Generators.SyntheticSAXGenerator(288, dict(g, SHUFFLE=False), in_memory=True)
    model.fit(x=train, epochs=1, verbose=0)                      # warm-up: engines, capture
    model.evaluate(val)
    torch.cuda.synchronize()
    for name, kw in (('fit, no validation', {}), ('fit + validation each epoch', dict(validation_data=val))):
        t0 = time.perf_counter()
        model.fit(x=train, epochs=3, verbose=0, **kw)
        torch.cuda.synchronize()
        print('%-30s %.1f ms per epoch' % (name, 1e3 * (time.perf_counter() - t0) / 3), flush=True)
    t0 = time.perf_counter()
    for _ in range(3):
        v = model.evaluate(val)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print('evaluate(288 slices)           %.1f ms = %.0f slices/s   %s' % (1e3 * dt, 288 / dt, v), flush=True)
    xs = np.concatenate([val[i][0] for i in range(len(val))], 0)
    model.predict(xs, batch_size=32)
    t0 = time.perf_counter()
    for _ in range(3):
        p = model.predict(xs, batch_size=32)
    dt = (time.perf_counter() - t0) / 3
    print('predict(288 slices, ndarray)   %.1f ms = %.0f slices/s   out %s' % (1e3 * dt, 288 / dt, p.shape), flush=True)
    t0 = time.perf_counter()
    for _ in range(3):
        fl, pts, sz = [], [], []
        for i in range(0, 288, 32):
            a_, b_, c_ = model.predict_rvip(xs[i:i + 32])
    dt = (time.perf_counter() - t0) / 3
    print('predict_rvip(288 slices)       %.1f ms = %.0f slices/s' % (1e3 * dt, 288 / dt), flush=True)
    t0 = time.perf_counter()
    for _ in range(3):
        for i in range(len(val)):
            val[i]
    print('generator alone                %.1f ms per pass' % (1e3 * (time.perf_counter() - t0) / 3))


if __name__ == '__main__':
    main()
