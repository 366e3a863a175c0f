"""Diagnostic: does Model.fit learn the marked-slice landmark task?  Prints loss / landmark error per setting."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cmr_landmark_detection_amd as rvip
M = rvip.Loss_and_metrics


class Marked(rvip.Generators.SyntheticSAXGenerator):
    def _slice(self, rng, h, w):
        img, mask = super()._slice(rng, h, w)
        yy, xx = np.mgrid[0:h, 0:w]
        img = 0.5 * img[..., 0]
        for c, sign in ((0, 1.0), (1, -1.0)):
            cy, cx = np.unravel_index(int(mask[..., c].argmax()), (h, w))
            img = img + sign * 0.5 * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 3.0 ** 2))
        img = (img - img.min()) / (img.max() - img.min())
        return img[..., None].astype(np.float32), mask


def run(tag, epochs=10, lr=3e-3, sigma=2, prec='bf16', loss=M.mse, **kw):
    tmp = tempfile.mkdtemp()
    cfg = dict(DIM=[64, 64], FILTERS=8, DEPTH=3, BATCH_NORMALISATION=True, ACTIVATION='relu', MASK_CLASSES=2, LEARNING_RATE=lr,
               RVIP_PRECISION=prec, LOSS_FUNCTION=loss, SEED=5, MODEL_PATH=tmp, **kw)
    g = dict(DIM=[64, 64], BATCHSIZE=16, GAUS=True, SIGMA=sigma, SHUFFLE=True, SEED=7)
    train = Marked(480, g, in_memory=True)
    val = Marked(64, dict(g, SHUFFLE=False, SEED=8), in_memory=True)
    model = rvip.get_model(cfg, metrics=[])
    h = model.fit(x=train, validation_data=val, epochs=epochs, verbose=0, workers=2).history
    err = []
    for i in range(len(val)):
        xb, yb = val[i]
        idx, _ = model.predict_landmarks(xb)
        ti = yb.reshape(yb.shape[0], -1, 2).argmax(1)
        err.append(np.hypot(idx // 64 - ti // 64, idx % 64 - ti % 64))
    err = np.concatenate(err).ravel()
    p = model.predict(val[0][0])
    print(tag, 'loss', ['%.4f' % v for v in h['loss'][::max(1, epochs // 6)]], 'val', '%.4f' % h['val_loss'][-1], 'median err %.1f' % np.median(err),
          'within2 %.2f' % (err <= 2).mean(), 'pred max %.3f' % p.max(), flush=True)


run('bcedice30_drop', epochs=30, loss=M.bce_dice_loss)
run('bcedice30_drop_fp16', epochs=30, loss=M.bce_dice_loss, prec='fp16')
run('mse30_drop01', epochs=30, DROPOUT_MIN=0.1, DROPOUT_MAX=0.1)
run('mse30_fp16', epochs=30, prec='fp16', DROPOUT_MIN=0.0, DROPOUT_MAX=0.0)
run('BceDiceLoss30', epochs=30, loss=M.BceDiceLoss())
