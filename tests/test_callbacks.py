"""Epoch-by-epoch schedule of the callbacks the reference's get_callbacks assembles (src/utils/KerasCallbacks.py:54-111,
167-174: ModelCheckpoint best-only on `loss`, ReduceLROnPlateau(factor DECAY_FACTOR, patience, cooldown 2, min_delta 1e-4,
min_lr), LRTensorBoard's `lr` log entry, EarlyStopping) against sequences derived BY HAND from the tf.keras 2.3 rules
restated in the module docstring of cmr-landmark-detection_amd/KerasCallbacks.py.  Host logic only: no GPU."""
import importlib
import math
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rvip = importlib.import_module('cmr-landmark-detection_amd')
K = rvip.KerasCallbacks
nan = float('nan')


class _FakeModel:
    def __init__(self, lr=1e-3):
        self.optimizer = rvip.Adam(lr=lr)
        self.stop_training = False
        self.saved = []

    def save_weights(self, p, overwrite=True):
        self.saved.append(p)


def _drive(cbs, losses, lr=1e-3, key='loss'):
    fm = _FakeModel(lr)
    cl = K.CallbackList(cbs, fm)
    cl.on_train_begin()
    lr_logged, lr_after, stopped = [], [], None
    for e, l in enumerate(losses):
        logs = {key: l}
        cl.on_epoch_begin(e)
        cl.on_epoch_end(e, logs)
        lr_logged.append(logs.get('lr'))
        lr_after.append(float(fm.optimizer.lr))
        if fm.stop_training:
            stopped = e
            break
    cl.on_train_end()
    return fm, lr_logged, lr_after, stopped


def test_reduce_lr_plateau_inside_cooldown():
    # patience 2, cooldown 2.  e0 improves (best 1.0); e1 wait 1; e2 wait 2 -> lr/2, counter 2; e3 counter 1 (still in cooldown:
    # no wait); e4 counter 0 -> NOT in cooldown any more in the same epoch -> wait 1; e5 wait 2 -> lr/2, counter 2; e6 improves
    # during cooldown (counter 1, best .5); e7 counter 0, wait 1; e8 wait 2 -> lr/2.
    cb = K.ReduceLROnPlateau(monitor='loss', factor=0.5, patience=2, cooldown=2, min_lr=1e-12)
    fm, logged, after, _ = _drive([cb], [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.5, 0.5, 0.5])
    assert logged == [1e-3, 1e-3, 1e-3, 5e-4, 5e-4, 5e-4, 2.5e-4, 2.5e-4, 2.5e-4]      # lr of the epoch that just ran
    assert after == [1e-3, 1e-3, 5e-4, 5e-4, 5e-4, 2.5e-4, 2.5e-4, 2.5e-4, 1.25e-4]
    assert cb.best == 0.5 and cb.cooldown_counter == 2 and cb.wait == 0


def test_reduce_lr_min_delta_tie_and_min_lr_clamp():
    # min_delta 2**-10 is exact in binary: a loss EXACTLY best - min_delta is not an improvement (strict np.less), one ulp-scale
    # step further is.  e0 best 1; e1 tie -> wait 1; e2 improves -> best, wait 0; e3 wait 1; e4 wait 2 -> reduce.
    d = 2.0 ** -10
    cb = K.ReduceLROnPlateau(monitor='loss', factor=0.1, patience=2, cooldown=0, min_delta=d, min_lr=1e-4)
    fm, logged, after, _ = _drive([cb], [1.0, 1.0 - d, 1.0 - d - 2.0 ** -20, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0], lr=3e-4)
    assert cb.best == 1.0 - d - 2.0 ** -20
    # first reduction at e4: max(3e-5, 1e-4) = min_lr; cooldown 0 -> wait restarts: e5 wait 1, e6 wait 2 but lr == min_lr: NO change and
    # (Keras) neither wait nor the cooldown counter is reset, so every later epoch re-enters the same branch.
    assert after == [3e-4, 3e-4, 3e-4, 3e-4, 1e-4, 1e-4, 1e-4, 1e-4, 1e-4]
    assert cb.wait == 4 and cb.cooldown_counter == 0
    # default min_delta 1e-4 (KerasCallbacks.py:63-70 passes none): 0.99995 is within it, 0.9998 is not
    cb = K.ReduceLROnPlateau(monitor='loss', factor=0.5, patience=5, cooldown=2)
    _drive([cb], [1.0, 0.99995])
    assert cb.best == 1.0 and cb.wait == 1
    _drive([cb], [1.0, 0.9998])                     # on_train_begin resets the state
    assert cb.best == 0.9998 and cb.wait == 0


def test_reduce_lr_nan_loss_and_auto_mode():
    # NaN never compares better: it counts as a non-improving epoch.  e0 best 1; e1 nan wait 1; e2 nan wait 2 -> reduce, counter 2;
    # e3 0.5 improves (counter 1); e4 nan counter 0, then not in cooldown: wait 1; e5 nan: wait 2 -> reduce.
    cb = K.ReduceLROnPlateau(monitor='loss', factor=0.5, patience=2, cooldown=2, mode='auto', min_lr=1e-12)
    fm, logged, after, _ = _drive([cb], [1.0, nan, nan, 0.5, nan, nan])
    assert after == [1e-3, 1e-3, 5e-4, 5e-4, 5e-4, 2.5e-4] and cb.best == 0.5
    # mode 'auto': max only when 'acc' is in the monitor name -- a dice metric is still minimised (Keras quirk)
    assert K.ReduceLROnPlateau(monitor='dice_coef_labels', mode='auto').best == np.inf
    assert K.ReduceLROnPlateau(monitor='val_acc', mode='auto').best == -np.inf
    # a missing monitor only logs lr
    cb = K.ReduceLROnPlateau(monitor='val_loss', patience=1)
    fm, logged, after, _ = _drive([cb], [1.0, 1.0, 1.0])
    assert logged == [1e-3] * 3 and after == [1e-3] * 3 and cb.wait == 0


def test_early_stopping_sequences():
    # patience 3, min_delta 0, min mode: tie is not an improvement.  e0 best 1; e1 tie wait 1; e2 0.9 best; e3 wait 1; e4 nan wait 2;
    # e5 0.9 (tie) wait 3 -> stop at epoch 5.
    cb = K.EarlyStopping(monitor='loss', patience=3, mode='min')
    fm, _, _, stopped = _drive([cb], [1.0, 1.0, 0.9, 0.95, nan, 0.9, 0.1])
    assert stopped == 5 and cb.stopped_epoch == 5 and cb.best == 0.9


def test_early_stopping_max_mode_min_delta():
    cb = K.EarlyStopping(monitor='dice', patience=2, mode='max', min_delta=0.5)
    fm, _, _, stopped = _drive([cb], [1.0, 1.5, 1.75, 2.0, 2.25, 2.25], key='dice')
    # e0 best 1.0; e1 1.5-.5 = 1.0 > 1.0 false: wait 1; e2 1.25 > 1.0: best 1.75, wait 0; e3 1.5 > 1.75 false: wait 1; e4 1.75 > 1.75 false:
    # wait 2 -> stop at 4
    assert stopped == 4 and cb.best == 1.75
    # state is reset by on_train_begin (a second fit() starts from scratch)
    fm, _, _, stopped = _drive([cb], [0.0, 0.0, 0.0], key='dice')
    assert stopped == 2 and cb.best == 0.0


def test_checkpoint_best_only_sequences(tmp_path):
    p = str(tmp_path / 'sub' / 'model.h5')
    cb = K.ModelCheckpoint(p, monitor='loss', save_best_only=True, save_weights_only=True, mode='min')
    fm, _, _, _ = _drive([cb], [1.0, 1.0, nan, 0.5, 0.5, 0.25, nan])
    assert cb.saved_epochs == [0, 3, 5] and fm.saved == [p] * 3 and cb.best == 0.25       # strict less: ties and NaN do not save
    assert os.path.isdir(str(tmp_path / 'sub'))
    cb = K.ModelCheckpoint(p, monitor='val_loss', save_best_only=True)
    fm, _, _, _ = _drive([cb], [1.0, 0.5])                                                # monitor missing from logs: never saved
    assert fm.saved == []
    cb = K.ModelCheckpoint(p, monitor='loss', save_best_only=False)
    fm, _, _, _ = _drive([cb], [1.0, 2.0, nan])
    assert cb.saved_epochs == [0, 1, 2]
    assert K.ModelCheckpoint(p, monitor='val_acc', mode='auto').best == -np.inf
    assert K.ModelCheckpoint(p, monitor='fmeasure', mode='auto').best == -np.inf
    assert K.ModelCheckpoint(p, monitor='dice_coef', mode='auto').best == np.inf


def test_get_callbacks_assembly_and_combined_schedule(tmp_path):
    cfg = dict(MODEL_PATH=str(tmp_path / 'm'), DECAY_FACTOR=0.5, REDUCE_LR_ON_PLAEAU_PATIENCE=2, EARLY_STOPPING_PATIENCE=6,
               MONITOR_FUNCTION='loss')
    cbs = K.get_callbacks(cfg)
    assert [type(c).__name__ for c in cbs] == ['ModelCheckpoint', 'ReduceLROnPlateau', 'LRLogger', 'EarlyStopping']
    ck, rl, _, es = cbs
    assert ck.filepath == os.path.join(cfg['MODEL_PATH'], 'model.h5') and ck.monitor == 'loss' and ck.save_best_only
    assert (rl.factor, rl.patience, rl.cooldown, rl.min_delta, rl.min_lr) == (0.5, 2, 2, 1e-4, 1e-12)
    assert es.patience == 6 and es.monitor == 'loss' and es.monitor_op is np.less
    with pytest.raises(KeyError):
        K.get_callbacks({})                                                                # ensure_dir(config['MODEL_PATH']) (:31)
    d = K.get_callbacks(dict(MODEL_PATH=str(tmp_path / 'd')))
    assert (d[1].factor, d[1].patience, d[3].patience) == (0.5, 5, 25)
    # losses: best at e1; then a plateau.  ReduceLR (patience 2, cooldown 2): e2 wait 1, e3 wait 2 -> 5e-4 (counter 2), e4 counter 1,
    # e5 counter 0 + wait 1, e6 wait 2 -> 2.5e-4, e7 counter 1.  EarlyStopping (patience 6): waits 1..6 on e2..e7 -> stop at e7.
    fm, logged, after, stopped = _drive(cbs, [1.0, 0.9, 0.95, 0.96, 0.97, 0.98, 0.99, 1.0, 1.1])
    assert ck.saved_epochs == [0, 1]
    assert after == [1e-3, 1e-3, 1e-3, 5e-4, 5e-4, 5e-4, 2.5e-4, 2.5e-4]
    # LRTensorBoard comes AFTER ReduceLROnPlateau in the list (:63-79) and overwrites logs['lr'] with the optimizer's value at that
    # moment, so in the assembled list the logged lr of a reducing epoch is already the reduced one
    assert logged == after
    assert stopped == 7 and es.stopped_epoch == 7


def test_polynomial_decay_scheduler(tmp_path):
    cbs = K.get_callbacks(dict(MODEL_PATH=str(tmp_path), POLY_LR_DECAY=True, EPOCHS=10, LEARNING_RATE=1e-2))
    assert [type(c).__name__ for c in cbs] == ['ModelCheckpoint', 'ReduceLROnPlateau', 'LRLogger', 'LearningRateScheduler', 'EarlyStopping']
    fm, logged, after, _ = _drive(cbs, [1.0, 0.9, 0.8], lr=1e-2)
    want = [1e-2 * (1 - e / 10.0) ** 2 for e in range(3)]                              # KerasCallbacks.py:230-243, power 2 (:85)
    assert all(math.isclose(a, b, rel_tol=1e-12) for a, b in zip(after, want))
    assert all(math.isclose(a, b, rel_tol=1e-12) for a, b in zip(logged, want))
    assert K.get_callbacks(dict(MODEL_PATH=str(tmp_path), POLY_LR_DECAY=True))[3].schedule.initAlpha == 1e-4      # default (:84)
