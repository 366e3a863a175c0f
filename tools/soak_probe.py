"""fp16 / BCE-Dice leg of tools/soak_fit.py with the history printed (a regression probe): python tools/soak_probe.py [epochs] [steps] [prec] [loss] [dim] [filters] [depth] [batch]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cmr_landmark_detection_amd as rvip

M = rvip.Loss_and_metrics
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
spe = int(sys.argv[2]) if len(sys.argv) > 2 else 24
prec = sys.argv[3] if len(sys.argv) > 3 else 'fp16'
loss = M.bce_dice_loss if (len(sys.argv) <= 4 or sys.argv[4] == 'bce_dice') else M.mse
dim, filters, depth, batch = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((5, 256), (6, 32), (7, 4), (8, 32)))
with tempfile.TemporaryDirectory() as tmp:
    cfg = dict(DIM=[dim, dim], FILTERS=filters, DEPTH=depth, BATCH_NORMALISATION=True, ACTIVATION='relu', MASK_CLASSES=2,
               LEARNING_RATE=1e-3, RVIP_PRECISION=prec, LOSS_FUNCTION=loss, SEED=3, MODEL_PATH=tmp)
    gcfg = dict(DIM=[dim, dim], BATCHSIZE=batch, GAUS=True, SIGMA=4, SHUFFLE=True, SEED=5)
    train = rvip.Generators.SyntheticSAXGenerator(batch * spe, gcfg, in_memory=True)
    val = rvip.Generators.SyntheticSAXGenerator(2 * batch, dict(gcfg, SHUFFLE=False, SEED=6), in_memory=True)
    model = rvip.get_model(cfg, metrics=[M.dice_coef_labels])
    hist = model.fit(x=train, validation_data=val, epochs=epochs, callbacks=rvip.KerasCallbacks.get_callbacks(cfg, train, val),
                     verbose=0, max_queue_size=6, workers=4)
    h = hist.history
    print(prec, getattr(loss, '__name__', 'loss'), 'loss', ' '.join('%.5f' % v for v in h['loss']))
    print('val_loss', ' '.join('%.5f' % v for v in h['val_loss']))
    model.close()
