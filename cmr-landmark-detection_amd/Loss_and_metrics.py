"""Losses and metrics of the hot path, by the reference's names (src/models/Loss_and_metrics.py).

On the training path these objects are TAGS: ``Model.compile`` maps them to the fused HIP head kernels
(``rvip_head_fwd`` / ``rvip_head_grad`` fold sum (p-t)^2, sum BCE, sum t*p, sum t, sum p in the same
pass that writes the heat-map), so the loss and the three dice metrics cost no extra pass.  Called
directly with NumPy arrays they evaluate the same formulas on the host (validation / notebooks).

  mse / MSE                 tf.keras.losses.MSE as used by train_model.py:184
  bce_dice_loss             Loss_and_metrics.py:229-245 (w_bce 0.5, w_dice 1)
  BceDiceLoss               Loss_and_metrics.py:208-226 (w_bce 1, w_dice 1)
  dice_coef, dice_coef_labels / _lower / _upper / _myo / _lv / _rv     :124-171
"""
from __future__ import annotations

import numpy as np


def _tag(kind, **kw):
    def deco(fn):
        fn.rvip_kind = kind
        fn.rvip_args = kw
        return fn
    return deco


def dice_coef(y_true, y_pred):
    smooth = 1.
    yt = np.asarray(y_true, np.float64).ravel()
    yp = np.asarray(y_pred, np.float64).ravel()
    return (2. * (yt * yp).sum() + smooth) / (yt.sum() + yp.sum() + smooth)


@_tag('metric', sums='labels')
def dice_coef_labels(y_true, y_pred):
    return dice_coef(np.asarray(y_true)[..., -3:], np.asarray(y_pred)[..., -3:])


@_tag('metric', sums='lower')
def dice_coef_lower(y_true, y_pred):
    return dice_coef(np.asarray(y_true)[..., -2], np.asarray(y_pred)[..., -2])


@_tag('metric', sums='upper')
def dice_coef_upper(y_true, y_pred):
    return dice_coef(np.asarray(y_true)[..., -1], np.asarray(y_pred)[..., -1])


@_tag('metric', sums='lower')
def dice_coef_myo(y_true, y_pred):
    return dice_coef(np.asarray(y_true)[..., -2], np.asarray(y_pred)[..., -2])


@_tag('metric', sums='upper')
def dice_coef_lv(y_true, y_pred):
    return dice_coef(np.asarray(y_true)[..., -1], np.asarray(y_pred)[..., -1])


@_tag('metric', sums=None)
def dice_coef_rv(y_true, y_pred):
    return dice_coef(np.asarray(y_true)[..., -3], np.asarray(y_pred)[..., -3])


@_tag('metric', sums=None)
def binary_accuracy(y_true, y_pred):
    """keras.metrics.binary_accuracy (create_unet's default metric, Unets.py:81)."""
    return float(np.mean((np.asarray(y_pred) > 0.5) == (np.asarray(y_true) > 0.5)))


def _bce(y_true, y_pred):
    p = np.clip(np.asarray(y_pred, np.float64), 1e-7, 1 - 1e-7)
    t = np.asarray(y_true, np.float64)
    return -(t * np.log(p) + (1 - t) * np.log(1 - p))


@_tag('loss', loss='mse')
def mse(y_true, y_pred):
    return float(np.mean((np.asarray(y_pred, np.float64) - np.asarray(y_true, np.float64)) ** 2))


MSE = mean_squared_error = mse


@_tag('loss', loss='bce_dice', w_bce=0.5, w_dice=1.0)
def bce_dice_loss(y_true, y_pred, w_bce=0.5, w_dice=1.):
    if np.asarray(y_pred).shape[-1] == 4:
        y_pred, y_true = np.asarray(y_pred)[..., -3:], np.asarray(y_true)[..., -3:]
    return float(w_bce * _bce(y_true, y_pred).mean() - w_dice * dice_coef(y_true, y_pred))


class BceDiceLoss:
    """Loss_and_metrics.py:208-226 (the object train_model.py:182 builds for 'BcdDiceLoss').

    The reference class overrides ``tf.keras.losses.Loss.__call__`` (:220-226), which bypasses Keras' mean reduction: the
    training objective becomes the SUM over the replica's B*H*W elements (gradient = B_local*H*W times the mean form's) while
    the logged value stays the element mean -- the argument, with the tf.keras 2.3 source files it rests on, is in
    ``oracle/rvip_oracle.py::bce_dice_loss``.  ``rvip_args['reduction'] = 'sum'`` makes the engine scale the loss gradient
    accordingly; the plain function ``bce_dice_loss`` keeps 'mean'."""
    rvip_kind = 'loss'

    def __init__(self, w_bce=1., w_dice=1., binary=True, name='BcdDiceLoss'):
        if not binary:
            raise NotImplementedError('categorical variant is not on the hot path')
        self.w_bce, self.w_dice = w_bce, w_dice
        self.name = '{}_w_{}_{}'.format(name, w_bce, w_dice)
        self.__name__ = self.name
        self.rvip_args = dict(loss='bce_dice', w_bce=w_bce, w_dice=w_dice, reduction='sum')

    def __call__(self, y_true, y_pred, **kwargs):
        return bce_dice_loss(y_true, y_pred, self.w_bce, self.w_dice)


@_tag('loss', loss='unsupported')
def categorical_crossentropy(y_true, y_pred):
    """create_unet's fallback when LOSS_FUNCTION is missing (Unets.py:83); not a heat-map loss -- the HIP head
    kernels implement MSE and BCE-Dice only, so training with it raises."""
    p = np.clip(np.asarray(y_pred, np.float64), 1e-7, 1.0)
    p = p / p.sum(-1, keepdims=True)
    return float(-(np.asarray(y_true) * np.log(p)).sum(-1).mean())


def resolve_loss(loss):
    """Anything the reference passes as LOSS_FUNCTION -> (kind, w_bce, w_dice, display name)."""
    if isinstance(loss, dict):
        loss = loss.get('unet', next(iter(loss.values())))
    if isinstance(loss, str):
        key = loss.lower()
        if key in ('mse', 'mean_squared_error'):
            return 'mse', 0.0, 0.0, 'mse'
        if key in ('bcddiceloss', 'bcedice', 'bce_dice_loss', 'bce_dice'):
            return ('bce_dice', 1.0, 1.0, 'BcdDiceLoss') if key == 'bcddiceloss' else ('bce_dice', 0.5, 1.0, 'bce_dice_loss')
        raise ValueError('unknown LOSS_FUNCTION %r' % loss)
    args = getattr(loss, 'rvip_args', None)
    if args is None or getattr(loss, 'rvip_kind', None) != 'loss':
        raise ValueError('LOSS_FUNCTION %r is not one of this package\'s loss tags (mse, bce_dice_loss, BceDiceLoss)' % (loss,))
    return args['loss'], args.get('w_bce', 0.0), args.get('w_dice', 0.0), getattr(loss, '__name__', 'loss')


def loss_reduction(loss):
    """'sum' for the BceDiceLoss class form (and the string train_model.py:178-184 maps to it), else 'mean' (see BceDiceLoss)."""
    if isinstance(loss, dict):
        loss = loss.get('unet', next(iter(loss.values())))
    if isinstance(loss, str):
        return 'sum' if loss.lower() == 'bcddiceloss' else 'mean'
    return (getattr(loss, 'rvip_args', None) or {}).get('reduction', 'mean')
