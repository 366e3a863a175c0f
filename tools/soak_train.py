"""Functional soak: a few hundred optimizer steps of the headline configuration on fresh synthetic batches (bf16 and fp16):
the loss must fall steadily and stay finite; reports slices/s including the host-side batch hand-over (Model.fit path)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cmr_landmark_detection_amd as rvip
M = rvip.Loss_and_metrics
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for prec in ('bf16', 'fp16'):
    cfg = dict(DIM=[256, 256], FILTERS=32, DEPTH=4, BATCH_NORMALISATION=True, ACTIVATION='relu', MASK_CLASSES=2,
               LEARNING_RATE=1e-3, RVIP_PRECISION=prec, LOSS_FUNCTION=M.mse, SEED=3)
    model = rvip.get_model(cfg, metrics=[])
    gen = rvip.Generators.SyntheticSAXGenerator(32 * 16, dict(DIM=[256, 256], BATCHSIZE=32, GAUS=True, SIGMA=4, SHUFFLE=True, SEED=5))
    losses = []
    t0 = time.perf_counter()
    for s in range(steps):
        x, y = gen[s % len(gen)]
        losses.append(model.train_on_batch(x, y)[0])
    dt = time.perf_counter() - t0
    L = np.array(losses)
    print(prec, 'steps', steps, 'loss first/25th/last %.5f %.5f %.5f' % (L[0], L[min(24, steps - 1)], L[-1]), 'min %.5f' % L.min(),
          'finite', bool(np.isfinite(L).all()), '%.0f slices/s incl. host hand-over' % (32 * steps / dt))
    assert np.isfinite(L).all() and L[-10:].mean() < 0.5 * L[:10].mean()
    xv, yv = gen[0]
    p = model.predict(xv)
    am = p.reshape(32, -1, 2).argmax(1); at = yv.reshape(32, -1, 2).argmax(1)
    d = np.hypot(am // 256 - at // 256, am % 256 - at % 256)
    print('   landmark error after training (px): median %.1f, 90th pct %.1f' % (np.median(d), np.percentile(d, 90)))
