"""Where an epoch of the reference's training loop spends its wall time besides the steps: get_weights, the HDF5 checkpoint,
evaluate, the callback list (tools/soak_fit.py's configuration).
    python tools/probe_epoch_costs.py"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cmr_landmark_detection_amd as rvip

M = rvip.Loss_and_metrics


def t(fn, n=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return 1e3 * (time.perf_counter() - t0) / n


with tempfile.TemporaryDirectory() as tmp:
    cfg = dict(DIM=[256, 256], FILTERS=32, DEPTH=4, BATCH_NORMALISATION=True, ACTIVATION='relu', MASK_CLASSES=2,
               LEARNING_RATE=1e-3, RVIP_PRECISION='bf16', LOSS_FUNCTION=M.mse, SEED=3, MODEL_PATH=tmp)
    gcfg = dict(DIM=[256, 256], BATCHSIZE=32, GAUS=True, SIGMA=4, SHUFFLE=True, SEED=5)
    train = rvip.Generators.SyntheticSAXGenerator(32 * 44, gcfg, in_memory=True)
    val = rvip.Generators.SyntheticSAXGenerator(288, dict(gcfg, SHUFFLE=False, SEED=6), in_memory=True)
    model = rvip.get_model(cfg, metrics=[M.dice_coef_labels])
    model.fit(x=train, epochs=1, verbose=0)
    print('get_weights            %7.1f ms' % t(model.get_weights))
    print('save_weights (HDF5)    %7.1f ms' % t(lambda: model.save_weights(os.path.join(tmp, 'w.h5'))))
    if hasattr(model, 'save'):
        print('save (model.h5)        %7.1f ms' % t(lambda: model.save(os.path.join(tmp, 'm.h5'))))
    print('evaluate (288 slices)  %7.1f ms' % t(lambda: model.evaluate(val)))
    for name, cbs, env in (('no callbacks', [], '1'), ('reference callback list, checkpoint in the call', None, '0'),
                           ('reference callback list, background checkpoint', None, '1'), ('no callbacks', [], '1'),
                           ('reference callback list, checkpoint in the call', None, '0'), ('reference callback list, background checkpoint', None, '1')):
        os.environ['RVIP_ASYNC_CHECKPOINT'] = env
        c = rvip.KerasCallbacks.get_callbacks(cfg, train, val) if cbs is None else cbs
        for cb in c:
            if type(cb).__name__ == 'ModelCheckpoint':
                cb.save_best_only = False                              # a checkpoint EVERY epoch: the worst case
        t0 = time.perf_counter()
        model.fit(x=train, validation_data=val, epochs=8, callbacks=c, verbose=0)
        print('fit + validation, %-52s %7.1f ms per epoch' % (name, 1e3 * (time.perf_counter() - t0) / 8), flush=True)
