"""Callback protocol + the callbacks ``get_callbacks`` assembles in the reference
(src/utils/KerasCallbacks.py:20-114): best-only weights checkpoint ``model.h5`` (:54-61), ReduceLROnPlateau (:63-70),
the ``lr`` log entry of LRTensorBoard (:72-79, 167-174), optional polynomial decay through a LearningRateScheduler
(:80-87, 230-243), EarlyStopping (:105-111).  Host-side logic only; TensorBoard / matplotlib image writers and the
``OptimizerChanger`` SGD fine-tune (:89-104, only with ``metrics=``) are out of scope.

The three schedule callbacks restate tf.keras 2.3 (``tensorflow/python/keras/callbacks.py``; TensorFlow is a third-party
dependency of the reference, ``environment.yml:126``, absent here), epoch by epoch:

  ModelCheckpoint(save_best_only)   saves iff ``monitor_op(current, best)`` with a STRICT np.less / np.greater, best starts
                                    at +/-inf; a missing monitor skips the save; NaN never compares better -> never saved.
                                    mode 'auto': max iff 'acc' in monitor or monitor starts with 'fmeasure'.
  ReduceLROnPlateau                 ``logs['lr']`` is written FIRST (the lr of the epoch that just ran); then
                                    in_cooldown -> counter -= 1, wait = 0; improvement = less(current, best - min_delta)
                                    (min mode) -> best = current, wait = 0; elif NOT in_cooldown (re-evaluated after the
                                    decrement): wait += 1 and at wait >= patience, if lr > min_lr: lr = max(lr * factor,
                                    min_lr), cooldown_counter = cooldown, wait = 0.  mode 'auto': max iff 'acc' in monitor.
                                    State is reset in on_train_begin.
  EarlyStopping                     improvement = monitor_op(current - min_delta, best) with min_delta signed by the
                                    direction; else wait += 1 and stop at wait >= patience (stopped_epoch = epoch).
                                    best / wait are reset in on_train_begin.

Order of calls is Keras': set_model, on_train_begin, on_epoch_begin, on_train_batch_end, on_epoch_end(epoch, logs) with
keys loss, <metric names>, val_*, lr; on_train_end.  Callbacks run in list order, so the checkpoint sees the epoch's
logs before ReduceLROnPlateau adds ``lr`` and EarlyStopping runs last (:54-111).
"""
from __future__ import annotations

import os

import numpy as np


class Callback:
    def __init__(self):
        self.model = None

    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None): pass
    def on_train_end(self, logs=None): pass
    def on_epoch_begin(self, epoch, logs=None): pass
    def on_epoch_end(self, epoch, logs=None): pass
    def on_train_batch_end(self, batch, logs=None): pass


class CallbackList:
    def __init__(self, callbacks, model):
        self.callbacks = callbacks
        for c in callbacks:
            c.set_model(model)

    def __getattr__(self, name):
        def fan(*a, **k):
            for c in self.callbacks:
                getattr(c, name)(*a, **k)
        return fan


def _direction(mode, monitor, auto_max):
    """(op, initial best): strict comparisons, as np.less / np.greater in Keras."""
    if mode not in ('auto', 'min', 'max'):
        mode = 'auto'                                  # Keras warns and falls back
    if mode == 'max' or (mode == 'auto' and auto_max(monitor)):
        return np.greater, -np.inf
    return np.less, np.inf


class ModelCheckpoint(Callback):
    def __init__(self, filepath, monitor='val_loss', verbose=0, save_best_only=False, save_weights_only=True,
                 mode='auto', save_freq='epoch'):
        super().__init__()
        if save_freq != 'epoch':
            raise NotImplementedError("save_freq=%r: the reference saves once per epoch (KerasCallbacks.py:61)" % (save_freq,))
        if not save_weights_only:
            raise NotImplementedError('save_weights_only=False: the reference checkpoints weights only (KerasCallbacks.py:58)')
        self.filepath, self.monitor, self.verbose = filepath, monitor, verbose
        self.save_best_only = save_best_only
        self.monitor_op, self.best = _direction(mode, monitor, lambda m: 'acc' in m or m.startswith('fmeasure'))
        self.saved_epochs = []

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        filepath = self.filepath.format(epoch=epoch + 1, **logs)
        if self.save_best_only:
            cur = logs.get(self.monitor)
            if cur is None or not self.monitor_op(cur, self.best):
                return
            self.best = cur
        if self._is_chief():
            os.makedirs(os.path.dirname(os.path.abspath(filepath)), exist_ok=True)
        # collective-safe: save_weights itself writes on rank 0 only.  The file is serialised and written by a background thread
        # (RVIP_ASYNC_CHECKPOINT=0: in this call, as Keras does); it is complete under its name when fit() returns.
        if os.environ.get('RVIP_ASYNC_CHECKPOINT', '1') != '0' and hasattr(self.model, 'wait_for_checkpoint'):
            self.model.save_weights(filepath, overwrite=True, background=True)
        else:
            self.model.save_weights(filepath, overwrite=True)
        self.saved_epochs.append(epoch)

    @staticmethod
    def _is_chief():
        try:
            import torch.distributed as dist
            return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0
        except Exception:
            return True


class ReduceLROnPlateau(Callback):
    def __init__(self, monitor='val_loss', factor=0.1, patience=10, verbose=0, mode='auto', min_delta=1e-4,
                 cooldown=0, min_lr=0.0):
        super().__init__()
        if factor >= 1.0:
            raise ValueError('ReduceLROnPlateau does not support a factor >= 1.0.')
        self.monitor, self.factor, self.patience, self.verbose = monitor, factor, patience, verbose
        self.min_delta, self.cooldown, self.min_lr, self.mode = min_delta, cooldown, min_lr, mode
        self._reset()

    def _reset(self):
        op, self.best = _direction(self.mode, self.monitor, lambda m: 'acc' in m)
        if op is np.less:
            self.monitor_op = lambda a, b: np.less(a, b - self.min_delta)
        else:
            self.monitor_op = lambda a, b: np.greater(a, b + self.min_delta)
        self.cooldown_counter = 0
        self.wait = 0

    def on_train_begin(self, logs=None):
        self._reset()

    def in_cooldown(self):
        return self.cooldown_counter > 0

    def on_epoch_end(self, epoch, logs=None):
        logs = logs if logs is not None else {}
        logs['lr'] = float(self.model.optimizer.lr)
        cur = logs.get(self.monitor)
        if cur is None:
            return
        if self.in_cooldown():
            self.cooldown_counter -= 1
            self.wait = 0
        if self.monitor_op(cur, self.best):
            self.best, self.wait = cur, 0
        elif not self.in_cooldown():
            self.wait += 1
            if self.wait >= self.patience:
                old = float(self.model.optimizer.lr)
                if old > self.min_lr:
                    self.model.optimizer.lr = max(old * self.factor, self.min_lr)
                    self.cooldown_counter = self.cooldown
                    self.wait = 0


class EarlyStopping(Callback):
    def __init__(self, monitor='val_loss', min_delta=0, patience=0, verbose=0, mode='auto', baseline=None):
        super().__init__()
        self.monitor, self.patience, self.baseline = monitor, patience, baseline
        self.monitor_op, _ = _direction(mode, monitor, lambda m: 'acc' in m)
        self.min_delta = abs(min_delta) * (1 if self.monitor_op is np.greater else -1)
        self.wait = 0
        self.stopped_epoch = 0
        self.best = np.inf if self.monitor_op is np.less else -np.inf

    def on_train_begin(self, logs=None):
        self.wait = 0
        self.stopped_epoch = 0
        if self.baseline is not None:
            self.best = self.baseline
        else:
            self.best = np.inf if self.monitor_op is np.less else -np.inf

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return
        if self.monitor_op(cur - self.min_delta, self.best):
            self.best, self.wait = cur, 0
        else:
            self.wait += 1
            if self.wait >= self.patience:
                self.stopped_epoch = epoch
                self.model.stop_training = True


class LRLogger(Callback):
    """The part of LRTensorBoard that matters to the loop: ``logs['lr']`` (KerasCallbacks.py:167-174)."""

    def on_epoch_end(self, epoch, logs=None):
        if logs is not None:
            logs.update({'lr': float(self.model.optimizer.lr)})


class PolynomialDecay:
    """KerasCallbacks.py:230-243: the schedule object, ``lr(epoch) = initAlpha * (1 - epoch / maxEpochs) ** power``."""

    def __init__(self, maxEpochs=100, initAlpha=0.01, power=0.25):
        self.maxEpochs, self.initAlpha, self.power = maxEpochs, initAlpha, power

    def __call__(self, epoch):
        return float(self.initAlpha * (1 - (epoch / float(self.maxEpochs))) ** self.power)


class LearningRateScheduler(Callback):
    """tf.keras LearningRateScheduler: ``lr = schedule(epoch)`` at every epoch begin, ``logs['lr']`` at its end."""

    def __init__(self, schedule, verbose=0):
        super().__init__()
        self.schedule, self.verbose = schedule, verbose

    def on_epoch_begin(self, epoch, logs=None):
        try:
            lr = self.schedule(epoch, float(self.model.optimizer.lr))
        except TypeError:                              # Keras: old one-argument schedule API
            lr = self.schedule(epoch)
        self.model.optimizer.lr = float(lr)

    def on_epoch_end(self, epoch, logs=None):
        if logs is not None:
            logs['lr'] = float(self.model.optimizer.lr)


def get_callbacks(config=None, batch_generator=None, validation_generator=None, metrics=None):
    """The loop-relevant subset of KerasCallbacks.get_callbacks (:20-114): same config keys, defaults and list order."""
    config = config or {}
    cbs = []
    if metrics:
        raise NotImplementedError('get_callbacks(metrics=...): the OptimizerChanger SGD fine-tune (KerasCallbacks.py:89-104) is not built')
    os.makedirs(config['MODEL_PATH'], exist_ok=True)          # ensure_dir(config['MODEL_PATH']) (:31): the key is required
    cbs.append(ModelCheckpoint(os.path.join(config['MODEL_PATH'], 'model.h5'), verbose=1, save_best_only=True,
                               save_weights_only=True, monitor=config.get('SAVE_MODEL_FUNCTION', 'loss'),
                               mode=config.get('SAVE_MODEL_MODE', 'min'), save_freq='epoch'))
    cbs.append(ReduceLROnPlateau(monitor=config.get('MONITOR_FUNCTION', 'loss'), factor=config.get('DECAY_FACTOR', 0.5),
                                 patience=config.get('REDUCE_LR_ON_PLAEAU_PATIENCE', 5), verbose=1, cooldown=2,
                                 mode=config.get('MONITOR_MODE', 'auto'), min_lr=config.get('MIN_LR', 1e-12)))
    cbs.append(LRLogger())
    if config.get('POLY_LR_DECAY', False):
        cbs.append(LearningRateScheduler(PolynomialDecay(maxEpochs=config.get('EPOCHS', 100),
                                                         initAlpha=config.get('LEARNING_RATE', 1e-4), power=2), verbose=1))
    cbs.append(EarlyStopping(patience=config.get('EARLY_STOPPING_PATIENCE', 25), verbose=1,
                             monitor=config.get('MONITOR_FUNCTION', 'loss'), mode=config.get('MONITOR_MODE', 'min')))
    return cbs
