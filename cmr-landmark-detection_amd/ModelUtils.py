"""Optimizer factory by the reference's name (src/models/ModelUtils.py:75-118, ``get_optimizer``).

Only Adam is on the hot path (example_config.json "OPTIMIZER": "adam"); it is Keras' OptimizerV2 Adam:
beta_1 0.9, beta_2 0.999, epsilon 1e-7 placed OUTSIDE the bias correction (EPSILON / DECAY are read by the
reference, :86-87, and never passed on).  The object is a host-side handle: the state (m, v, iterations,
lr) lives on the device next to the flat parameter block and is advanced by ``rvip_adam_step``.
"""
from __future__ import annotations

import weakref


class _LR:
    """``model.optimizer.lr`` -- readable / assignable like the Keras variable (KerasCallbacks.py:173)."""

    def __init__(self, value):
        self._value = float(value)
        self._listeners = []                      # weak references to bound methods: the variable must not keep a model alive

    def numpy(self):
        return self._value

    def add_listener(self, method):
        """`method(value)` is called on every assignment for as long as its object lives.  Held weakly: model -> optimizer -> lr ->
        listener -> model would be a reference cycle, and a model in a cycle dies in the cyclic collector -- at an arbitrary moment,
        on an arbitrary thread -- together with the hipGraphs its engines own (see Engine.capture)."""
        ref = weakref.WeakMethod(method)
        self._listeners = [r for r in self._listeners if r() is not None and r() != method] + [ref]

    def assign(self, v):
        self._value = float(v)
        for ref in list(self._listeners):
            cb = ref()
            if cb is not None:
                cb(self._value)

    def __float__(self):
        return self._value

    def __repr__(self):
        return 'lr(%g)' % self._value


class Adam:
    def __init__(self, lr=0.001, learning_rate=None, beta_1=0.9, beta_2=0.999, epsilon=1e-7, name='adam'):
        self._lr = _LR(lr if learning_rate is None else learning_rate)
        self.beta_1, self.beta_2, self.epsilon, self.name = beta_1, beta_2, epsilon, name
        self.iterations = 0

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, v):
        self._lr.assign(float(v))

    learning_rate = lr

    def get_config(self):
        return dict(name=self.name, learning_rate=float(self._lr), beta_1=self.beta_1, beta_2=self.beta_2,
                    epsilon=self.epsilon)


def get_optimizer(config, name_suff=''):
    opt = str(config.get('OPTIMIZER', 'Adam')).lower()
    lr = config.get('LEARNING_RATE', 0.001)
    config.get('EPSILON', 1e-08)       # read and ignored, as in the reference (ModelUtils.py:86)
    config.get('DECAY', 0.0)
    if opt == 'adam':
        return Adam(lr=lr, name=opt + name_suff)
    if opt in ('adagrad', 'rmsprop', 'adadelta', 'radam', 'nadam', 'sgd'):
        raise NotImplementedError("OPTIMIZER=%r: only 'adam' is built (the reference's shipped config)" % opt)
    return Adam()                      # ModelUtils.py:113-115: unknown string -> Adam with defaults
