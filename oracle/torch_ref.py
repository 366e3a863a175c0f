"""PyTorch-CPU implementation of the same graph: INDEPENDENT cross-check of rvip_oracle and the
``cpu_baseline`` timer of bench.py ("PyTorch-CPU stand-in for TF2-CPU").

TEST INFRASTRUCTURE (see oracle/__init__.py).  It shares only ``build_graph`` (the layer table)
with the NumPy oracle; the arithmetic is torch.nn.functional + autograd, i.e. different code.
Keras semantics kept: BN eps 1e-3 / momentum 0.99 with unbiased moving variance, first-max pooling,
nearest upsampling, [up, skip] concat order, Keras-Adam epsilon placement.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import rvip_oracle as O


def _act(x, kind):
    if kind in (None, 'linear'):
        return x
    if kind == 'relu':
        return F.relu(x)
    if kind == 'elu':
        return F.elu(x)
    if kind == 'sigmoid':
        return torch.sigmoid(x)
    raise ValueError(kind)


class TorchUNet:
    def __init__(self, config, params=None, seed=42, dtype=torch.float32):
        self.layers = O.build_graph(config)
        self.dtype = dtype
        np_params = params if params is not None else O.init_params(self.layers, seed, np.float32)
        self.params = OrderedDict()
        for k, arrs in np_params.items():
            ts = [torch.tensor(np.asarray(a), dtype=dtype) for a in arrs]
            lay = next(l for l in self.layers if l['name'] == k)
            n_train = 2
            for t in ts[:n_train]:
                t.requires_grad_(True)
            self.params[k] = ts
        self.lr = float(config.get('LEARNING_RATE', 0.001))
        self.iterations = 0
        self.m = {k: [torch.zeros_like(t) for t in v[:2]] for k, v in self.params.items()}
        self.v = {k: [torch.zeros_like(t) for t in v[:2]] for k, v in self.params.items()}

    def forward(self, x, training=False, dropout_masks=None, return_logits=False):
        """x: torch [N,H,W,C] (Keras layout); internally NCHW."""
        t = {}
        logits = None
        for l in self.layers:
            name, ty = l['name'], l['type']
            ins = [t[i] for i in l['inputs']]
            if ty == 'InputLayer':
                out = x.permute(0, 3, 1, 2).to(self.dtype)
            elif ty == 'Conv2D':
                w, b = self.params[name]
                k = l['kernel']
                pre = F.conv2d(ins[0], w.permute(3, 2, 0, 1), b, padding=(k[0] // 2, k[1] // 2))
                if name == 'unet':
                    logits = pre
                out = _act(pre, l['activation'])
            elif ty == 'Conv2DTranspose':
                w, b = self.params[name]                          # HWOI
                s = l['strides'][0]
                full = F.conv_transpose2d(ins[0], w.permute(3, 2, 0, 1), b, stride=s, padding=0)
                oh, ow = ins[0].shape[2] * s, ins[0].shape[3] * s
                out = _act(full[:, :, :oh, :ow], l['activation'])
            elif ty == 'Activation':
                out = _act(ins[0], l['activation'])
            elif ty == 'BatchNormalization':
                g, b, mm, mv = self.params[name]
                out = F.batch_norm(ins[0], mm, mv, g, b, training=training, momentum=1 - O.BN_MOMENTUM, eps=O.BN_EPS)
            elif ty == 'Dropout':
                if training and dropout_masks is not None and name in dropout_masks:
                    m = torch.as_tensor(np.asarray(dropout_masks[name]), dtype=self.dtype).permute(0, 3, 1, 2)
                    out = ins[0] * m / (1.0 - l['rate'])
                elif training and dropout_masks == 'random':
                    out = F.dropout(ins[0], l['rate'], True)
                else:
                    out = ins[0]
            elif ty == 'MaxPooling2D':
                out = F.max_pool2d(ins[0], l['pool'])
            elif ty == 'UpSampling2D':
                out = F.interpolate(ins[0], scale_factor=tuple(float(s) for s in l['size']), mode='nearest')
            elif ty == 'Concatenate':
                out = torch.cat(ins, 1)
            else:
                raise NotImplementedError(ty)
            t[name] = out
        self.tensors = t
        y = t['unet'].permute(0, 2, 3, 1)
        if return_logits:
            return y, logits.permute(0, 2, 3, 1)
        return y

    def loss(self, y_true, y_pred, logits, kind, global_batch=None):
        n = y_pred.shape[0]
        gb = n if global_batch is None else global_batch
        per = y_pred[0].numel()
        if kind == 'mse':
            return ((y_pred - y_true) ** 2).sum() / (gb * per)
        if kind == 'bce_dice':
            bce = F.binary_cross_entropy_with_logits(logits, y_true, reduction='sum') / (gb * per)
            inter = (y_true * y_pred).sum()
            dice = (2 * inter + 1.0) / (y_true.sum() + y_pred.sum() + 1.0)
            return 0.5 * bce - dice * (n / gb)
        raise ValueError(kind)

    def loss_and_grads(self, x, y, kind='mse', dropout_masks=None, global_batch=None):
        for v in self.params.values():
            for t in v[:2]:
                t.grad = None
        x = torch.as_tensor(x)
        y = torch.as_tensor(y, dtype=self.dtype)
        pred, logits = self.forward(x, True, dropout_masks, return_logits=True)
        L = self.loss(y, pred, logits, kind, global_batch)
        L.backward()
        grads = OrderedDict((k, [t.grad for t in v[:2]]) for k, v in self.params.items())
        return L.detach(), grads, pred.detach()

    def adam(self, grads):
        self.iterations += 1
        t = self.iterations
        lr_t = self.lr * math.sqrt(1 - O.ADAM_B2 ** t) / (1 - O.ADAM_B1 ** t)
        with torch.no_grad():
            for k, gs in grads.items():
                for i, g in enumerate(gs):
                    m, v, p = self.m[k][i], self.v[k][i], self.params[k][i]
                    m.mul_(O.ADAM_B1).add_(g, alpha=1 - O.ADAM_B1)
                    v.mul_(O.ADAM_B2).addcmul_(g, g, value=1 - O.ADAM_B2)
                    p.sub_(lr_t * m / (v.sqrt() + O.ADAM_EPS))

    def train_step(self, x, y, kind='mse', dropout_masks=None, global_batch=None):
        L, grads, pred = self.loss_and_grads(x, y, kind, dropout_masks, global_batch)
        self.adam(grads)
        return float(L), pred

    def numpy_params(self):
        return OrderedDict((k, [t.detach().numpy().copy() for t in v]) for k, v in self.params.items())


def time_train_steps(config, batch, steps=2, warmup=1, threads=None, seed=42):
    """CPU baseline: seconds per full training step (fwd + bwd + Keras-Adam, dropout on), fp32."""
    import time
    if threads:
        torch.set_num_threads(threads)
    net = TorchUNet(config, seed=seed)
    x, y = O.synthetic_batch(batch, config['DIM'], config.get('MASK_CLASSES', 2), seed=seed)
    for _ in range(warmup):
        net.train_step(x, y, 'mse', 'random')
    t0 = time.perf_counter()
    for _ in range(steps):
        net.train_step(x, y, 'mse', 'random')
    return (time.perf_counter() - t0) / steps
