"""Callback protocol + the callbacks ``get_callbacks`` assembles in the reference
(src/utils/KerasCallbacks.py:20-114): best-only weights checkpoint (:54-61), ReduceLROnPlateau (:63-70),
the ``lr`` log entry of LRTensorBoard (:167-174), optional polynomial decay (:80-87,230-243), EarlyStopping
(:105-111).  Host-side logic only; TensorBoard / matplotlib image writers are out of scope.

Order of calls is Keras': set_model, on_train_begin, on_epoch_begin, on_train_batch_end, on_epoch_end(epoch,
logs) with keys loss, <metric names>, val_*, lr; on_train_end.
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


def _better(mode, monitor):
    if mode == 'max' or (mode == 'auto' and ('acc' in monitor or 'dice' in monitor)):
        return lambda a, b, delta=0.0: a > b + delta, -np.inf
    return lambda a, b, delta=0.0: a < b - delta, np.inf


class ModelCheckpoint(Callback):
    def __init__(self, filepath, monitor='val_loss', verbose=0, save_best_only=False, save_weights_only=True,
                 mode='auto', save_freq='epoch'):
        super().__init__()
        self.filepath, self.monitor, self.verbose = filepath, monitor, verbose
        self.save_best_only = save_best_only
        self.op, self.best = _better(mode, monitor)

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if self.save_best_only:
            if cur is None or not self.op(cur, self.best):
                return
            self.best = cur
        os.makedirs(os.path.dirname(os.path.abspath(self.filepath)), exist_ok=True)
        self.model.save_weights(self.filepath)


class ReduceLROnPlateau(Callback):
    def __init__(self, monitor='val_loss', factor=0.1, patience=10, verbose=0, mode='auto', min_delta=1e-4,
                 cooldown=0, min_lr=0.0):
        super().__init__()
        self.monitor, self.factor, self.patience, self.verbose = monitor, factor, patience, verbose
        self.min_delta, self.cooldown, self.min_lr = min_delta, cooldown, min_lr
        self.op, self.best = _better(mode, monitor)
        self.wait = 0
        self.cooldown_counter = 0

    def on_epoch_end(self, epoch, logs=None):
        logs = logs if logs is not None else {}
        logs['lr'] = float(self.model.optimizer.lr)
        cur = logs.get(self.monitor)
        if cur is None:
            return
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.wait = 0
        if self.op(cur, self.best, self.min_delta):
            self.best, self.wait = cur, 0
        elif self.cooldown_counter <= 0:
            self.wait += 1
            if self.wait >= self.patience:
                old = float(self.model.optimizer.lr)
                if old > self.min_lr:
                    self.model.optimizer.lr = max(old * self.factor, self.min_lr)
                    self.cooldown_counter = self.cooldown
                    self.wait = 0


class EarlyStopping(Callback):
    def __init__(self, monitor='val_loss', min_delta=0, patience=0, verbose=0, mode='auto'):
        super().__init__()
        self.monitor, self.min_delta, self.patience = monitor, abs(min_delta), patience
        self.op, self.best = _better(mode, monitor)
        self.wait = 0
        self.stopped_epoch = 0

    def on_train_begin(self, logs=None):
        self.wait = 0

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return
        if self.op(cur, self.best, self.min_delta):
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


class PolynomialDecay(Callback):
    """KerasCallbacks.py:230-243 semantics: lr = init * (1 - epoch/max_epochs) ** power at each epoch begin."""

    def __init__(self, max_epochs=100, init_alpha=0.01, power=1.0):
        super().__init__()
        self.max_epochs, self.init_alpha, self.power = max_epochs, init_alpha, power

    def on_epoch_begin(self, epoch, logs=None):
        self.model.optimizer.lr = self.init_alpha * (1 - (epoch / float(self.max_epochs))) ** self.power


def get_callbacks(config=None, batch_generator=None, validation_generator=None, metrics=None):
    """The loop-relevant subset of KerasCallbacks.get_callbacks (:20-114), same config keys and defaults."""
    config = config or {}
    cbs = []
    if 'MODEL_PATH' in config:
        cbs.append(ModelCheckpoint(os.path.join(config['MODEL_PATH'], 'model.npz'), verbose=1, save_best_only=True,
                                   save_weights_only=True, monitor=config.get('SAVE_MODEL_FUNCTION', 'loss'),
                                   mode=config.get('SAVE_MODEL_MODE', 'min'), save_freq='epoch'))
    cbs.append(ReduceLROnPlateau(monitor=config.get('MONITOR_FUNCTION', 'loss'), factor=config.get('DECAY_FACTOR', 0.5),
                                 patience=config.get('REDUCE_LR_ON_PLAEAU_PATIENCE', 5), verbose=1, cooldown=2,
                                 mode=config.get('MONITOR_MODE', 'auto'), min_lr=config.get('MIN_LR', 1e-12)))
    cbs.append(LRLogger())
    if config.get('POLY_LR_DECAY', False):
        cbs.append(PolynomialDecay(max_epochs=config.get('EPOCHS', 100), init_alpha=config.get('LEARNING_RATE', 0.001), power=2))
    cbs.append(EarlyStopping(monitor=config.get('MONITOR_FUNCTION', 'loss'), patience=config.get('EARLY_STOPPING_PATIENCE', 25),
                             mode=config.get('MONITOR_MODE', 'auto')))
    return cbs
