"""Soak of the product's own loop: Model.fit at the headline configuration over many epochs with generator pool threads, validation and
the reference's callback list -- the stager / pinned-ring protocol (engine.InputRing) and the capture guard under a long run.
    python tools/soak_fit.py [epochs=12] [steps_per_epoch=24] [workers=4]"""
import os
import sys
import tempfile
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cmr_landmark_detection_amd as rvip

M = rvip.Loss_and_metrics
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
spe = int(sys.argv[2]) if len(sys.argv) > 2 else 24
workers = int(sys.argv[3]) if len(sys.argv) > 3 else 4
for prec, loss in (('bf16', M.mse), ('fp16', M.bce_dice_loss)):
    with tempfile.TemporaryDirectory() as tmp:
        cfg = dict(DIM=[256, 256], FILTERS=32, DEPTH=4, BATCH_NORMALISATION=True, ACTIVATION='relu', MASK_CLASSES=2,
                   LEARNING_RATE=1e-3, RVIP_PRECISION=prec, LOSS_FUNCTION=loss, SEED=3, MODEL_PATH=tmp)
        gcfg = dict(DIM=[256, 256], BATCHSIZE=32, GAUS=True, SIGMA=4, SHUFFLE=True, SEED=5)
        train = rvip.Generators.SyntheticSAXGenerator(32 * spe, gcfg, in_memory=True)
        val = rvip.Generators.SyntheticSAXGenerator(64, dict(gcfg, SHUFFLE=False, SEED=6), in_memory=True)
        model = rvip.get_model(cfg, metrics=[M.dice_coef_labels])
        before = threading.active_count()
        t0 = time.perf_counter()
        hist = model.fit(x=train, validation_data=val, epochs=epochs, callbacks=rvip.KerasCallbacks.get_callbacks(cfg, train, val),
                         verbose=0, max_queue_size=6, workers=workers)
        dt = time.perf_counter() - t0
        h = hist.history
        assert threading.active_count() == before, 'fit left threads behind'
        assert np.isfinite(h['loss']).all() and np.isfinite(h['val_loss']).all() and h['loss'][-1] < h['loss'][0]
        print('%s %s: %d epochs x %d steps, workers %d: %.0f slices/s incl. validation; loss %.5f -> %.5f, val %.5f -> %.5f, launch %s' % (
            prec, getattr(loss, '__name__', 'loss'), len(h['loss']), spe, workers, 32 * spe * len(h['loss']) / dt, h['loss'][0], h['loss'][-1],
            h['val_loss'][0], h['val_loss'][-1], model._engine(32).launch_mode))
        model.close()
