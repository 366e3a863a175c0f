"""Keras-``Model``-shaped object returned by ``create_unet`` (SURVEY 8(b) boundary).

Surface used by the reference's callers, kept name for name:
  fit(x=Sequence, validation_data=, epochs=, callbacks=, initial_epoch=, max_queue_size=, verbose=)
      train_model.py:105-112, Train_tests.ipynb:924-931
  predict(Sequence | ndarray) -> float32 [N,*DIM,C]            predict_model.py:143, KerasCallbacks.py:481
  summary(print_fn=), save_weights / load_weights, get_weights / set_weights, compile, optimizer.lr,
  stop_training, trainable, metrics_names, count_params       train_model.py:84-89, predict_model.py:76

Host logic only lives here (epoch loop, batch sharding across ranks, callback protocol, logs); every FLOP
is dispatched to the HIP engine.  Constructing / summarising / (de)serialising a model needs no GPU;
fit / predict / train_on_batch raise without one (no CPU fallback).
"""
from __future__ import annotations

import math
import queue
import sys
import threading
import time
import weakref
from collections import OrderedDict

import numpy as np

from . import Loss_and_metrics as metr
from .KerasCallbacks import Callback as _Callback


def _he_normal(rng, shape):
    fan_in = int(np.prod(shape[:-1]))
    std = math.sqrt(2.0 / fan_in) / 0.87962566103423978          # Keras VarianceScaling truncated-normal correction
    a = rng.standard_normal(shape)
    bad = np.abs(a) > 2.0
    while bad.any():
        a[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(a) > 2.0
    return (a * std).astype(np.float32)


def _glorot_uniform(rng, shape):
    rf = int(np.prod(shape[:-2]))
    lim = math.sqrt(6.0 / (rf * shape[-2] + rf * shape[-1]))
    return rng.uniform(-lim, lim, shape).astype(np.float32)


def _variance_scaling(scale, mode, distribution):
    """Keras VarianceScaling (the named initialisers a KERNEL_INIT string can select, Unets.py:88): fan_in = receptive field x Cin,
    fan_out = receptive field x Cout; truncated normal (+-2 sigma, sigma corrected by 0.8796...) or uniform(+-sqrt(3 scale / n))."""
    def init(rng, shape):
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = rf * shape[-2], rf * shape[-1]
        n = {'fan_in': fan_in, 'fan_out': fan_out, 'fan_avg': 0.5 * (fan_in + fan_out)}[mode]
        if distribution == 'uniform':
            lim = math.sqrt(3.0 * scale / n)
            return rng.uniform(-lim, lim, shape).astype(np.float32)
        std = math.sqrt(scale / n) / 0.87962566103423978
        a = rng.standard_normal(shape)
        bad = np.abs(a) > 2.0
        while bad.any():
            a[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(a) > 2.0
        return (a * std).astype(np.float32)
    return init


_INIT = {'he_normal': _he_normal, 'glorot_uniform': _glorot_uniform,
         'he_uniform': _variance_scaling(2.0, 'fan_in', 'uniform'), 'glorot_normal': _variance_scaling(1.0, 'fan_avg', 'normal'),
         'lecun_normal': _variance_scaling(1.0, 'fan_in', 'normal'), 'lecun_uniform': _variance_scaling(1.0, 'fan_in', 'uniform'),
         'random_normal': lambda rng, s: (0.05 * rng.standard_normal(s)).astype(np.float32),
         'random_uniform': lambda rng, s: rng.uniform(-0.05, 0.05, s).astype(np.float32),
         'zeros': lambda rng, s: np.zeros(s, np.float32), 'ones': lambda rng, s: np.ones(s, np.float32)}


class History:
    def __init__(self):
        self.history = {}
        self.epoch = []


class Model:
    def __init__(self, plan, name='unet'):
        self.plan = plan
        self.name = name
        self.trainable = True
        self.stop_training = False
        self.optimizer = None
        self.loss = None
        self.metrics = []
        self.history = None
        cfg = plan.config
        self.precision = {'bf16': 'bf16', 'bfloat16': 'bf16', 'fp32': 'f32', 'f32': 'f32', 'float32': 'f32',
                          'fp16': 'f16', 'f16': 'f16', 'float16': 'f16', 'half': 'f16'}[
            str(cfg.get('RVIP_PRECISION', 'bf16')).lower()]
        self.seed = int(cfg.get('SEED', 42))
        rng = np.random.default_rng(self.seed)
        self._weights = []
        for (_, _, shape, _, init) in plan.weight_specs():
            if init not in _INIT:
                raise NotImplementedError('KERNEL_INIT=%r' % init)
            self._weights.append(_INIT[init](rng, tuple(shape)))
        self._params = None          # ParamStore once on the device
        self._engines = {}
        self._rings = {}
        self._eval_rings = {}          # batch size -> EvalRing (evaluate()'s own pinned ring)
        self._dropout_masks = None   # name -> uint8 ndarray, parity runs only

    # ---------------------------------------------------------------------------------------------
    # Keras bookkeeping
    # ---------------------------------------------------------------------------------------------
    @property
    def layers(self):
        return self.plan.layers

    def count_params(self):
        return self.plan.count_params()[0]

    def compile(self, optimizer=None, loss=None, metrics=None, **_):
        from .ModelUtils import Adam
        self.optimizer = optimizer if optimizer is not None else Adam()
        if isinstance(self.optimizer, str):
            self.optimizer = Adam()
        self.loss = loss
        self.metrics = list(metrics or [])
        self.optimizer.lr.add_listener(self._on_lr)
        self._engines = {}
        self._rings = {}
        self._eval_rings = {}          # batch size -> EvalRing (evaluate()'s own pinned ring)

    @property
    def metrics_names(self):
        return ['loss'] + [getattr(m, '__name__', str(m)) for m in self.metrics]

    def _on_lr(self, value):
        if self._params is not None:
            self._params.set_lr(value)

    def summary(self, line_length=98, print_fn=None):
        pr = print_fn or print
        pos = [int(line_length * p) for p in (.33, .55, .67, 1.)]

        def row(fields):
            line = ''
            for f, p in zip(fields, pos):
                line = (line + str(f))[:p]
                line += ' ' * (p - len(line))
            pr(line)

        pr('Model: "%s"' % self.name)
        pr('_' * line_length)
        row(['Layer (type)', 'Output Shape', 'Param #', 'Connected to'])
        pr('=' * line_length)
        for i, l in enumerate(self.plan.layers):
            shp = '(None, %s)' % ', '.join(str(s) for s in l.shape)
            if l.type == 'InputLayer':
                shp = '[%s]' % shp
            conns = ['%s[0][0]' % c for c in l.inputs] or ['']
            row(['%s (%s)' % (l.name, l.type), shp, l.params, conns[0]])
            for c in conns[1:]:
                row(['', '', '', c])
            pr(('=' if i == len(self.plan.layers) - 1 else '_') * line_length)
        tot, tr, ntr = self.plan.count_params()
        pr('Total params: {:,}'.format(tot))
        pr('Trainable params: {:,}'.format(tr))
        pr('Non-trainable params: {:,}'.format(ntr))
        pr('_' * line_length)

    # ---------------------------------------------------------------------------------------------
    # weights (Keras get_weights() order: conv kernel, bias; BN gamma, beta, moving_mean, moving_variance)
    # ---------------------------------------------------------------------------------------------
    def get_weights(self, sync=True):
        """Keras order.  Data-parallel with ``sync`` (the default): BN moving statistics are read as the MEAN over the replicas
        (Keras reads a MirroredStrategy BN variable that way) -- a COLLECTIVE, so every rank must then call get_weights /
        save_weights / evaluate together.  A caller on one rank only (`if rank == 0: model.get_weights(sync=False)`, a chief-only
        callback) passes ``sync=False`` and reads this replica's moving statistics; everything trainable is identical on all
        ranks anyway."""
        if self._params is not None:
            if sync:
                self.sync_moving_statistics()
            self._weights = self._params.download()
        return [w.copy() for w in self._weights]

    def set_weights(self, weights):
        specs = self.plan.weight_specs()
        if len(weights) != len(specs):
            raise ValueError('expected %d arrays, got %d' % (len(specs), len(weights)))
        new = []
        for (ln, wn, shape, _, _), w in zip(specs, weights):
            w = np.asarray(w, np.float32)
            if tuple(w.shape) != tuple(shape):
                raise ValueError('%s/%s: shape %s != %s' % (ln, wn, w.shape, tuple(shape)))
            new.append(w.copy())
        self._weights = new
        if self._params is not None:
            self._params.upload(self._weights)

    def weight_names(self):
        return ['%s/%s:0' % (ln, wn) for (ln, wn, _, _, _) in self.plan.weight_specs()]

    def _layers_with_weights(self):
        """[(layer name, [(Keras weight name, index into get_weights())])] over EVERY layer of the table, in model.layers order."""
        specs = self.plan.weight_specs()
        by_layer = OrderedDict((l.name, []) for l in self.plan.layers)
        for i, (ln, wn, _, _, _) in enumerate(specs):
            by_layer[ln].append(('%s/%s:0' % (ln, wn), i))
        return list(by_layer.items())

    def save_weights(self, filepath, overwrite=True, save_format=None, sync=True, background=False, **_):
        """Weights-only checkpoint (ModelCheckpoint(save_weights_only=True), KerasCallbacks.py:54-61).  ``*.h5`` / ``*.hdf5`` /
        ``*.keras`` (or save_format='h5') write the Keras-HDF5 layout ``model.load_weights`` of the reference reads
        (predict_model.py:75-76) through the in-tree HDF5 writer (keras_h5.py); ``*.npz`` keeps the NumPy container keyed
        '<layer>/<weight>:0'.  Data-parallel: BN moving statistics are mean-reduced over the replicas first (Keras
        MirroredVariable aggregation MEAN) -- a collective every rank must enter -- and only rank 0 writes; ``sync=False`` skips
        the collective (a rank-0-only caller) and writes this replica's moving statistics.

        ``background=True`` (the ModelCheckpoint callback): the weights are downloaded now (8 ms at config 2), serialised and
        written by a host-only thread (36 ms) while training goes on; the file appears under its name only when complete
        (written beside it, then os.replace), one writer at a time in call order; `wait_for_checkpoint()` -- called by the next
        save, load_weights, close() and at the end of fit() -- joins it and re-raises its error."""
        import os
        weights = self.get_weights(sync=sync)                          # collective when data-parallel (replica mean of the BN statistics)
        if self._dist()[0] != 0:
            return
        self.wait_for_checkpoint()
        if not overwrite and os.path.exists(filepath):
            raise FileExistsError(filepath)
        ext = os.path.splitext(str(filepath))[1].lower()
        if save_format in ('h5', 'hdf5', 'keras') or (save_format is None and ext in ('.h5', '.hdf5', '.keras')):
            from . import keras_h5
            layers = [(ln, [(wn, weights[i]) for wn, i in ws]) for ln, ws in self._layers_with_weights()]

            def write(path):
                keras_h5.save_keras_weights(path, layers)
        elif save_format in (None, 'npz'):
            arrs = OrderedDict(zip(self.weight_names(), weights))

            def write(path):
                with open(path, 'wb') as f:
                    np.savez(f, **arrs)
        else:
            raise ValueError("save_format=%r: 'h5' (Keras-HDF5) and 'npz' are written; the TensorFlow checkpoint format is not" % (save_format,))
        if not background:
            write(filepath)
            return
        import threading
        box = {}

        def run():
            tmp = '%s.part%d' % (filepath, os.getpid())
            try:
                write(tmp)
                os.replace(tmp, filepath)
            except BaseException as e:                                 # surfaces in wait_for_checkpoint
                box['error'] = e
                try:
                    os.remove(tmp)
                except OSError:
                    pass
        th = threading.Thread(target=run, name='rvip-checkpoint')
        th.start()
        self._checkpoint = (th, box)

    def wait_for_checkpoint(self):
        """Joins the background checkpoint writer, if any (host-only thread); re-raises what it raised."""
        ck = getattr(self, '_checkpoint', None)
        if ck is None:
            return
        self._checkpoint = None
        ck[0].join()
        if 'error' in ck[1]:
            raise ck[1]['error']

    def load_weights(self, filepath, by_name=False, **_):
        """Keras-HDF5 (weights-only ``model.h5`` or the /model_weights group of a full ``model.save`` file) or ``.npz``.
        HDF5 files load the way Keras does: by topology -- the file's layers that hold weights, in ``layer_names`` order,
        against this model's -- or ``by_name``.  Shapes must match (Keras raises ValueError as well)."""
        self.wait_for_checkpoint()
        with open(filepath, 'rb') as f:
            magic = f.read(8)
        if magic[:2] == b'PK':                                                     # NumPy .npz (zip)
            with np.load(filepath) as z:
                self.set_weights([z[k] for k in self.weight_names()])
            return
        from . import keras_h5
        file_layers, _ = keras_h5.load_keras_weights(filepath)
        mine = [(ln, ws) for ln, ws in self._layers_with_weights() if ws]
        new = self.get_weights()
        if by_name:
            for ln, ws in mine:
                if ln in file_layers and file_layers[ln]:
                    vals = file_layers[ln]
                    if len(vals) != len(ws):
                        raise ValueError('Layer %s expects %d weight(s), but the saved weights have %d element(s)' % (ln, len(ws), len(vals)))
                    for (_, i), (_, arr) in zip(ws, vals):
                        new[i] = arr
        else:
            theirs = [(ln, ws) for ln, ws in file_layers.items() if ws]
            if len(theirs) != len(mine):
                raise ValueError('You are trying to load a weight file containing %d layers into a model with %d layers.'
                                 % (len(theirs), len(mine)))
            for (ln, ws), (fn, vals) in zip(mine, theirs):
                if len(vals) != len(ws):
                    raise ValueError('Layer %s (file: %s) expects %d weight(s), but the saved weights have %d element(s)'
                                     % (ln, fn, len(ws), len(vals)))
                for (_, i), (_, arr) in zip(ws, vals):
                    new[i] = arr
        self.set_weights(new)

    def sync_moving_statistics(self):
        """BN moving mean / variance are per-replica state updated from per-replica batch statistics (plain
        BatchNormalization under MirroredStrategy, Unets.py:70-75); Keras reads such a variable as the MEAN over the replicas
        (SURVEY 2.3).  Called before evaluation / checkpointing: all-reduce(sum) / world of the flat moving block.  The mean
        REPLACES the local values: the moving average is linear in the batch statistics, so the replica mean evolves identically
        whether or not the replicas were set to it on the way, and the mean is the only view this package ever reads."""
        rank, world = self._dist()
        if world > 1 and self._params is not None:
            import torch.distributed as dist
            dist.all_reduce(self._params.moving)
            self._params.moving.div_(world)

    # ---------------------------------------------------------------------------------------------
    # device
    # ---------------------------------------------------------------------------------------------
    def _dist(self):
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                return dist.get_rank(), dist.get_world_size()
        except Exception:
            pass
        return 0, 1

    def _device(self):
        import os
        import torch
        from .engine import require_gpu
        require_gpu()
        dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)) % max(torch.cuda.device_count(), 1))
        torch.cuda.set_device(dev)
        return dev

    def _ensure_params(self):
        if self._params is None:
            from .engine import ParamStore
            lr = float(self.optimizer.lr) if self.optimizer is not None else 1e-3
            self._params = ParamStore(self.plan, self._weights, self.precision, self._device(), seed=self.seed, lr=lr)
            if self.optimizer is not None:
                self._params.set_step(self.optimizer.iterations)
        return self._params

    def _engine(self, batch):
        from .engine import Engine
        P = self._ensure_params()
        rank, world = self._dist()
        key = (int(batch), world)
        if key not in self._engines:
            import torch
            kind, w_bce, w_dice, _ = self._loss_spec(required=False)
            masks = None
            if self._dropout_masks:
                masks = {k: torch.from_numpy(np.ascontiguousarray(v[:batch], np.uint8)).to(P.device)
                         for k, v in self._dropout_masks.items()}
            self._engines[key] = Engine(P, batch, kind or 'mse', w_bce, w_dice, world=world, masks=masks,
                                        sum_reduction=(kind == 'bce_dice' and metr.loss_reduction(self.loss) == 'sum'))
        return self._engines[key]

    def _loss_spec(self, required=True):
        try:
            kind, w_bce, w_dice, name = metr.resolve_loss(self.loss)
        except ValueError:
            if required:
                raise
            return None, 0.5, 1.0, 'loss'
        if kind == 'unsupported':
            if required:
                raise NotImplementedError('LOSS_FUNCTION %s: the heat-map head kernels implement MSE and BCE-Dice' % name)
            return None, 0.5, 1.0, name
        return kind, w_bce, w_dice, name

    def set_dropout_masks(self, masks):
        """Parity hook: inject keep-masks (name -> uint8 [N,H,W,C]) instead of the counter-based stream."""
        self._dropout_masks = masks
        self._engines = {}
        self._rings = {}
        self._eval_rings = {}          # batch size -> EvalRing (evaluate()'s own pinned ring)

    # ---------------------------------------------------------------------------------------------
    # steps
    # ---------------------------------------------------------------------------------------------
    def _shard(self, x, y=None):
        """MirroredStrategy semantics (Unets.py:70-75): the generator's GLOBAL batch is split evenly by rank."""
        rank, world = self._dist()
        if world == 1:
            return x, y
        b = x.shape[0] // world
        if b * world != x.shape[0]:
            raise ValueError('global batch %d is not divisible by %d ranks' % (x.shape[0], world))
        sl = slice(rank * b, (rank + 1) * b)
        return x[sl], (None if y is None else y[sl])

    @staticmethod
    def _loss_count(eng, kind, world):
        """Elements the loss is the mean of over the global batch: every output element for MSE; BCE-Dice slices a 4-class head to
        its last three channels first (Loss_and_metrics.py:222-224, :240-242)."""
        k = eng.pred.shape[-1]
        return float(eng.pred.numel() // k * (min(k, 3) if kind == 'bce_dice' else k) * world)

    def _batch_logs(self, eng, kind, w_bce, w_dice):
        import torch
        s = eng.sums
        rank, world = self._dist()
        if world > 1:
            import torch.distributed as dist
            s = s.clone()
            dist.all_reduce(s)
        n = self._loss_count(eng, kind, world)
        dice_all = (2 * s[2] + 1) / (s[3] + s[4] + 1)
        loss = s[0] / n if kind == 'mse' else w_bce * s[1] / n - w_dice * dice_all
        vals = [loss]
        for m in self.metrics:
            which = getattr(m, 'rvip_args', {}).get('sums')
            if which == 'labels':
                vals.append(dice_all)
            elif which == 'lower':
                vals.append((2 * s[5] + 1) / (s[6] + s[7] + 1))
            elif which == 'upper':
                vals.append((2 * s[8] + 1) / (s[9] + s[10] + 1))
            else:
                vals.append(torch.full((), float('nan'), device=s.device))
        return torch.stack([v.reshape(()) for v in vals])

    def train_on_batch(self, x, y, return_dict=False):
        kind, w_bce, w_dice, _ = self._loss_spec()
        x, y = self._shard(np.asarray(x), np.asarray(y))
        eng = self._engine(x.shape[0])
        eng.load_input(x, y)
        eng.train_step()
        self.optimizer.iterations += 1
        vals = self._batch_logs(eng, kind, w_bce, w_dice).cpu().numpy().tolist()
        return dict(zip(self.metrics_names, vals)) if return_dict else vals

    def test_on_batch(self, x, y, return_dict=False, sharded=False):
        """`sharded`: (x, y) is already this rank's slice of the global batch (evaluate on a generator of this package)"""
        kind, w_bce, w_dice, _ = self._loss_spec()
        x, y = (np.asarray(x), np.asarray(y)) if sharded else self._shard(np.asarray(x), np.asarray(y))
        eng = self._engine(x.shape[0])
        eng.load_input(x, y)
        eng.forward_eval()
        vals = self._batch_logs(eng, kind, w_bce, w_dice).cpu().numpy().tolist()
        return dict(zip(self.metrics_names, vals)) if return_dict else vals

    MAX_EVAL_RINGS = 2                       # batch sizes whose pinned ring stays allocated (a generator's full and ragged last batch)

    def _eval_ring(self, eng, batch, used, need_y=True):
        """evaluate() / predict(): the pinned ring of this batch size, reset at its first use in a call (a call that ended in an
        exception may have left staged slots behind).  predict() rings hold inputs only; at most MAX_EVAL_RINGS batch sizes keep
        theirs (least recently used goes first)."""
        from .engine import EvalRing
        ring = self._eval_rings.pop(batch, None)
        if ring is not None and ring.x_stage is not eng.x_stage:
            ring.drop()
            ring = None
        if ring is None:
            idle = [b for b in self._eval_rings if b not in used]          # (a ring this call has used may still hold a download)
            while len(self._eval_rings) >= self.MAX_EVAL_RINGS and idle:
                self._eval_rings.pop(idle.pop(0)).drop()
            ring = EvalRing(eng, with_y=need_y)
        elif need_y:
            ring.ensure_targets()
        self._eval_rings[batch] = ring           # (re-inserted last: most recently used)
        if batch not in used:
            ring.reset_input_ring()
            used.add(batch)
        return ring

    def predict_on_batch(self, x):
        x = np.asarray(x, np.float32)
        eng = self._engine(x.shape[0])
        eng.load_input(x)
        eng.forward(training=False)
        return eng.pred.detach().cpu().numpy().reshape(eng.out_shape)

    def predict(self, x, batch_size=None, verbose=0, **_):
        """ndarray [N,*DIM,1] or a Sequence yielding (x, y) / x batches; returns float32 [N,*DIM,C] in order.

        Pipelined (round 4): inputs through a pinned ring (H2D under the previous batch's forward pass), heat-maps back through two
        pinned buffers on a stream of their own; the host copies batch k - 1 into the result while batch k runs."""
        if isinstance(x, np.ndarray):
            bs = batch_size or min(32, x.shape[0])
            batches = (x[i:i + bs] for i in range(0, x.shape[0], bs))
            total = x.shape[0]
        else:
            def seq():
                for i in range(len(x)):
                    b = x[i]
                    yield b[0] if isinstance(b, (tuple, list)) else b
            batches, total = seq(), None
        result, outs, off, pending, used = None, [], 0, None, set()

        def land(p):
            ring_, handle, n_, shape_, off_ = p
            if result is not None:
                ring_.fetch(handle, result[off_:off_ + n_])
            else:
                o = np.empty(shape_, np.float32)
                ring_.fetch(handle, o)
                outs.append(o)
        for xb in batches:
            xb = np.asarray(xb, np.float32)
            eng = self._engine(xb.shape[0])
            ring = self._eval_ring(eng, xb.shape[0], used, need_y=False)
            if total is not None and result is None:
                result = np.empty((total,) + tuple(eng.out_shape[1:]), np.float32)
            slot = ring.next_slot()
            ring.stage_host_batch(slot, xb, None)
            ring.feed(slot)
            eng.stage_input()
            eng.forward(training=False)
            handle = ring.download(eng.pred)
            if pending is not None:
                land(pending)
            pending = (ring, handle, xb.shape[0], eng.out_shape, off)
            off += xb.shape[0]
        if pending is not None:
            land(pending)
        if result is not None:
            return result
        return np.concatenate(outs, 0) if outs else np.zeros((0,), np.float32)

    def predict_landmarks(self, x, thr=0.5):
        """Heat-maps -> (argmax index [N,C] int64 row-major first-max, >thr label mask uint8) on the device."""
        x = np.asarray(x, np.float32)
        eng = self._engine(x.shape[0])
        eng.load_input(x)
        eng.forward(training=False)
        idx, mask = eng.landmarks(thr, want_mask=True)
        return idx.cpu().numpy().reshape(eng.out_shape[:-3] + eng.out_shape[-1:]), mask.cpu().numpy().reshape(eng.out_shape)

    def predict_rvip(self, x, thr=0.5, cc_filter=False):
        """The reference's post-threshold of a predicted batch on the device (predict_model.py:149-156 flat labels,
        Postprocess.py:108-120 largest 4-connected component per slice and label when `cc_filter`, evaluate_cv.py:418-442
        mean (y, x) per label).  Returns (flat uint8 [N,*DIM], points float32 [N,(T,)C,2] with NaN for absent labels,
        sizes int32 [N,(T,)C])."""
        import ctypes as C
        import torch
        from . import _native as N
        x = np.asarray(x, np.float32)
        eng = self._engine(x.shape[0])
        eng.load_input(x)
        eng.forward(training=False)
        n, H, W, K = eng.pred.shape
        L = N.lib()
        dev = eng.pred.device
        flat = torch.empty((n, H, W), dtype=torch.uint8, device=dev)
        pts = torch.empty((n, K, 2), dtype=torch.float32, device=dev)
        sizes = torch.empty((n, K), dtype=torch.int32, device=dev)
        wsb = L.rvip_postprocess_workspace(n, H, W, K)
        ws = torch.empty(wsb // 4 + 16, dtype=torch.int32, device=dev)
        N.check(L.rvip_postprocess(eng.pred.data_ptr(), flat.data_ptr(), pts.data_ptr(), sizes.data_ptr(), n, H, W, K,
                                   C.c_float(thr), 1 if cc_filter else 0, ws.data_ptr(), C.c_size_t(wsb),
                                   C.c_void_p(eng.stream())), 'rvip_postprocess')
        lead = eng.out_shape[:-3]
        return (flat.cpu().numpy().reshape(eng.out_shape[:-1]), pts.cpu().numpy().reshape(lead + (K, 2)),
                sizes.cpu().numpy().reshape(lead + (K,)))

    def evaluate(self, x, y=None, verbose=0, return_dict=False, **_):
        """Mean of the per-batch loss / metric values (BN on the moving statistics, no dropout).  Data-parallel: the moving
        statistics are mean-reduced over the replicas first (a collective: every rank calls evaluate)."""
        self.sync_moving_statistics()
        import torch
        kind, w_bce, w_dice, _ = self._loss_spec()
        rank, world = self._dist()
        local = (world > 1 and not isinstance(x, np.ndarray) and hasattr(x, 'batch_slice')
                 and getattr(x, 'BATCHSIZE', 0) > 0 and x.BATCHSIZE % world == 0)       # rank-local batches, as in fit()
        if isinstance(x, np.ndarray):
            batches = [(x, y)]
        elif local:
            b = x.BATCHSIZE // world
            batches = (x.batch_slice(i, rank * b, (rank + 1) * b) for i in range(len(x)))
        else:
            batches = (x[i] for i in range(len(x)))
        # Pipelined like fit(): pinned slot -> H2D on the copy stream under the previous batch's forward pass -> forward; the folded
        # sums of every batch stay on the device (one row each) and are read once at the end.  (Until round 4 every batch was a
        # pageable copy, a forward pass and a blocking read-out: 10 900 slices/s at config 2 -- tools/probe_evaluate.py.)
        rows, counts, used = [], [], set()
        for xb, yb in batches:
            xb, yb = np.asarray(xb), np.asarray(yb)
            if not local:
                xb, yb = self._shard(xb, yb)
            eng = self._engine(xb.shape[0])
            ring = self._eval_ring(eng, xb.shape[0], used)
            slot = ring.next_slot()
            ring.stage_host_batch(slot, xb, yb)
            ring.feed(slot)
            eng.stage_input()
            eng.forward_eval()
            rows.append(eng.sums.clone())
            counts.append(self._loss_count(eng, kind, world))
        if not rows:
            vals = [float('nan')] * len(self.metrics_names)
            return dict(zip(self.metrics_names, vals)) if return_dict else vals
        hist = torch.stack(rows)
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(hist)
        per_batch = self._values_from_sums(hist.cpu().numpy().astype(np.float64), np.asarray(counts, np.float64), kind, w_bce, w_dice)
        vals = per_batch.mean(0).tolist()
        return dict(zip(self.metrics_names, vals)) if return_dict else vals

    # ---------------------------------------------------------------------------------------------
    # fit (train_model.py:105-112)
    # ---------------------------------------------------------------------------------------------
    def _values_from_sums(self, s, n, kind, w_bce, w_dice):
        """Per-batch loss + metric values from rows of folded sums ([steps,16] float64; n = loss elements per step, scalar or
        [steps]): what _batch_logs computes on the device."""
        dice_all = (2 * s[:, 2] + 1) / (s[:, 3] + s[:, 4] + 1)
        loss = s[:, 0] / n if kind == 'mse' else w_bce * s[:, 1] / n - w_dice * dice_all
        cols = [loss]
        for m in self.metrics:
            which = getattr(m, 'rvip_args', {}).get('sums')
            if which == 'labels':
                cols.append(dice_all)
            elif which == 'lower':
                cols.append((2 * s[:, 5] + 1) / (s[:, 6] + s[:, 7] + 1))
            elif which == 'upper':
                cols.append((2 * s[:, 8] + 1) / (s[:, 9] + s[:, 10] + 1))
            else:
                cols.append(np.full(s.shape[0], np.nan))
        return np.stack(cols, 1)

    def fit(self, x=None, y=None, validation_data=None, epochs=1, callbacks=None, initial_epoch=0, max_queue_size=12,
            verbose=1, steps_per_epoch=None, shuffle=True, batch_size=None, workers=1, **_):
        """The reference's training loop (train_model.py:105-112; Train_tests.ipynb:924-931) on the HIP engine.

        Per step the training thread only (a) queues the host->device copy of the next batch from a pinned ring on a copy
        stream and (b) replays the captured step (Engine.train_step); generator batches are produced, sharded by rank and
        staged into pinned memory by background threads (Keras' OrderedEnqueuer: max_queue_size / workers).  Loss and metric
        sums stay on the device (one 64-byte row per step) and are read once per epoch.

        Data-parallel (one process per GPU): every rank iterates the SAME global batches -- the batch order is drawn from
        ``(SEED, epoch)`` and generators reshuffle from their own seeded stream, not from the process-global NumPy RNG -- and
        trains on its slice of each (MirroredStrategy, Unets.py:70-75).  Validation and checkpoints see the replica MEAN of
        the BN moving statistics; only rank 0 writes files and prints."""
        import torch
        from .KerasCallbacks import CallbackList
        kind, w_bce, w_dice, _ = self._loss_spec()
        if isinstance(x, np.ndarray):
            from .Generators import ArrayGenerator
            x = ArrayGenerator(x, y, batch_size or 32, shuffle=shuffle)
        gen = x
        rank, world = self._dist()
        self.history = History()
        cbs = CallbackList(list(callbacks or []) + [self.history_callback()], self)
        self.stop_training = False
        cbs.on_train_begin()
        names = self.metrics_names

        def epoch_orders():
            """per epoch: the batch order (Keras shuffles the batch order of a Sequence; seeded: identical on every rank), evaluated when the
            stager gets there (len(gen) may change in on_epoch_end)"""
            for ep in range(initial_epoch, epochs):
                steps = len(gen) if steps_per_epoch is None else min(steps_per_epoch, len(gen))
                order = np.arange(len(gen))
                if shuffle:
                    order = np.random.default_rng([self.seed, ep]).permutation(len(gen))
                yield order[:steps]
        staged = self._staged_batches(gen, epoch_orders(), max_queue_size, workers)
        ok = False                                             # set by the last statement of the loop: did THIS fit() finish cleanly
        try:
            for epoch in range(initial_epoch, epochs):
                if self.stop_training:
                    break
                cbs.on_epoch_begin(epoch)
                t0 = time.time()
                hist = eng = None
                steps = 0
                counts = []                                        # loss elements of every step's OWN batch (a Sequence may end on a smaller one)
                for step, (eng, slot, steps) in enumerate(staged.epoch()):
                    if hist is None:                               # `steps`: the length of the order the stager took for THIS epoch
                        hist = torch.zeros((steps, eng.sums.numel()), dtype=torch.float32, device=eng.sums.device)
                    if slot is not None:
                        eng.feed(slot)
                    eng.train_step()
                    hist[step].copy_(eng.sums, non_blocking=True)
                    counts.append(self._loss_count(eng, kind, world))
                    self.optimizer.iterations += 1
                    cbs.on_train_batch_end(step)
                logs = OrderedDict()
                if hist is not None:
                    if world > 1:
                        import torch.distributed as dist
                        dist.all_reduce(hist)
                    vals = self._values_from_sums(hist.cpu().numpy().astype(np.float64)[:len(counts)], np.asarray(counts, np.float64), kind, w_bce, w_dice)
                    for k, val in zip(names, vals.mean(0).tolist()):
                        logs[k] = val
                if validation_data is not None:
                    if isinstance(validation_data, (tuple, list)):
                        v = self.evaluate(validation_data[0], validation_data[1])
                    else:
                        v = self.evaluate(validation_data)
                    for k, val in zip(names, v):
                        logs['val_' + k] = val
                cbs.on_epoch_end(epoch, logs)
                if verbose and rank == 0:
                    dt = time.time() - t0
                    print('Epoch %d/%d - %.1fs - %.0fms/step - %s' % (
                        epoch + 1, epochs, dt, 1e3 * dt / max(steps, 1), ' - '.join('%s: %.4f' % kv for kv in logs.items())))
            ok = True
        finally:                                               # also when a callback stopped the training or raised: drop what was prefetched
            staged.close()
            if ok:                                                 # (not sys.exc_info(): fit() may itself run inside a caller's except block)
                self.wait_for_checkpoint()                         # the last background checkpoint is on disk when fit() returns -- or its error is raised
            else:
                try:
                    self.wait_for_checkpoint()
                except BaseException:
                    pass                                           # the training error is the one to report
        cbs.on_train_end()
        torch.cuda.synchronize()
        return self.history

    def _staged_batches(self, gen, orders, depth, workers):
        """Starts the stager thread of one fit() call (see _Stager) and returns it; the caller must close() it."""
        prev = getattr(self, '_stager', None)
        if prev is not None and prev() is not None and prev().alive():
            raise RuntimeError('fit(): the stager thread of an earlier fit() of this model is still alive')
        st = _Stager(self, gen, orders, depth, workers)
        self._stager = weakref.ref(st)
        return st

    def close(self):
        """Releases the device state (engines with their captured hipGraphs, pinned rings, parameter blocks) NOW, on the calling
        thread, after a device synchronisation -- instead of whenever the last reference happens to go.  The weights stay
        readable (they are downloaded first); the next fit / predict rebuilds the device state from them."""
        import torch
        self.wait_for_checkpoint()
        if self._params is not None:
            self.get_weights(sync=False)
            torch.cuda.synchronize()
        for eng in self._engines.values():
            eng.release()
        self._engines, self._rings, self._eval_rings, self._params = {}, {}, {}, None

    def history_callback(self):
        return _HistoryCallback(self)


def _prefetch(fetch, order, depth, workers=1, cancel=None):
    """In-order iterator over ``fetch(i) for i in order`` (called from the stager thread; fetch = the generator's __getitem__ or
    a rank's slice of it).  `workers` > 1 prepares that many batches at a time on a thread pool (NumPy / SciPy release the GIL in
    their kernels) and still delivers them in order -- fit(max_queue_size=, workers=) of train_model.py:111.  `cancel` (an
    Event) stops the submission of further batches; the pool is shut down when the iterator is closed."""
    depth = max(int(depth), 1)
    workers = max(int(workers or 1), 1)
    if workers > 1:
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        ex = ThreadPoolExecutor(max_workers=workers)
        try:
            pending, it = deque(), iter(order)
            for i in it:
                pending.append(ex.submit(fetch, int(i)))
                if len(pending) >= max(depth, workers):
                    break
            while pending:
                item = pending.popleft().result()              # re-raises a generator error
                nxt = None if (cancel is not None and cancel.is_set()) else next(it, None)
                if nxt is not None:
                    pending.append(ex.submit(fetch, int(nxt)))
                yield item
        finally:
            for f in pending:
                f.cancel()
            ex.shutdown(wait=True)
        return
    for i in order:
        if cancel is not None and cancel.is_set():
            return
        yield fetch(int(i))


class _HistoryCallback(_Callback):
    """Keras' History callback: epoch numbers and logs into ``model.history``.  (A module-level class holding the model in an
    attribute: a class defined inside fit() and closing over the model is a reference cycle of its own that keeps the model --
    and the hipGraphs of its engines -- alive until the cyclic collector runs, on whichever thread that happens.)"""

    def __init__(self, model):
        self.model = model

    def on_epoch_end(self, epoch, logs=None):
        self.model.history.epoch.append(epoch)
        for k, v in (logs or {}).items():
            self.model.history.history.setdefault(k, []).append(v)


class _Cancelled(Exception):
    pass


class _Stager:
    """The input pipeline of ONE fit() call: a thread that takes the generator's batches in the order of each epoch (produced by
    `workers` pool threads when > 1 -- Keras' OrderedEnqueuer, train_model.py:111), shards them by rank and copies them into the
    engine's pinned ring; ``epoch()`` (training thread) iterates (engine, pinned slot | None, steps of this epoch).

    THREADING CONTRACT: the stager thread and its pool touch HOST memory only -- NumPy, the generator, the pinned slots through their
    NumPy views, threading primitives.  Every HIP call of fit() (copies, events, engine creation, stream capture, synchronisation)
    is made by the training thread.  A pinned slot is handed back by the training thread through a host-side flag
    (Engine.feed -> Engine._release_slots) once the upload that read it has completed; the stager waits on that flag, not on a HIP
    event.  (Until round 3 the stager called hipEventSynchronize itself, concurrently with the training thread's stream capture
    and first-touch module loads.)

    Between two epochs the stager calls the generator's on_epoch_end() -- as soon as the epoch's last batch has been PRODUCED, as
    the OrderedEnqueuer does from its own thread -- and goes on with the next epoch's batches while the training thread still
    consumes the queue and reads out the epoch; the number of steps of an epoch travels with the epoch's first queue item, taken
    from the order the stager really used.  The very first batch of a batch size is handed over unpinned (the training thread
    creates the engine and its ring for it)."""

    _STOP = object()

    def __init__(self, model, gen, orders, depth, workers):
        self.model, self.gen, self.orders = model, gen, orders
        self.depth = max(int(depth), 1)
        self.workers = workers
        self.slots = self.depth + 4                        # Engine._release_slots frees slot k while feeding k + 2 at the latest
        self.q = queue.Queue(maxsize=self.depth)
        self.ring = model._rings                           # batch size -> engine whose pinned ring exists (kept across fits)
        self.ready = threading.Event()
        self.cancel = threading.Event()                    # set when the training thread leaves early (exception, callback stop)
        rank, world = model._dist()
        # Rank-local batches: a generator of this package hands out any slice of a batch (BaseGenerator.batch_slice), and the seeded
        # shuffles make every rank agree on what the global batch is, so a rank produces only ITS B / world samples -- as the
        # reference's single MirroredStrategy process produces every sample once (Generators.py:175-228).  A foreign Sequence is
        # asked for the whole batch, which is then sharded.
        bs = getattr(gen, 'BATCHSIZE', 0)
        self.local = world > 1 and hasattr(gen, 'batch_slice') and bs > 0 and bs % world == 0
        if self.local:
            b = bs // world
            self.fetch = lambda i: gen.batch_slice(i, rank * b, (rank + 1) * b)
        else:
            self.fetch = gen.__getitem__
        for eng in self.ring.values():
            eng.reset_input_ring()                         # (training thread, nothing in flight: slots all free, counters at zero)
        self.th = threading.Thread(target=self._work, name='rvip-stager', daemon=True)
        self.th.start()

    def alive(self):
        return self.th.is_alive()

    # -- stager thread: host memory only ----------------------------------------------------------
    def _put(self, item):
        while True:
            if self.cancel.is_set():
                raise _Cancelled()
            try:
                self.q.put(item, timeout=0.1)
                return
            except queue.Full:
                continue

    def _work(self):
        try:
            for order in self.orders:
                steps = len(order)
                batches = _prefetch(self.fetch, order, self.depth, self.workers, self.cancel)
                try:
                    for xb, yb in batches:
                        if not self.local:
                            xb, yb = self.model._shard(xb, yb)
                        eng = self.ring.get(xb.shape[0])
                        if eng is None or eng.ring_slots() < self.slots:
                            self.ready.clear()
                            self._put(('raw', xb, yb, steps))
                            while not self.ready.wait(0.1):    # the training thread builds the engine for this batch size
                                if self.cancel.is_set():
                                    raise _Cancelled()
                            continue
                        slot = eng.next_slot()
                        if not eng.stage_host_batch(slot, xb, yb, self.cancel):
                            raise _Cancelled()
                        self._put(('pin', eng, slot, steps))
                finally:
                    batches.close()                            # shuts the worker pool down (joins its threads)
                self._put(self._STOP)                          # end of this epoch's batches
                if hasattr(self.gen, 'on_epoch_end'):
                    self.gen.on_epoch_end()
        except _Cancelled:
            pass
        except BaseException as e:                             # surface generator errors in the training thread
            try:
                self._put(e)
                self._put(self._STOP)
            except _Cancelled:
                pass

    # -- training thread --------------------------------------------------------------------------
    def epoch(self):
        """the batches of the next epoch (up to the stager's end-of-epoch mark)"""
        while True:
            item = self.q.get()
            if item is self._STOP:
                return
            if isinstance(item, BaseException):
                raise item
            if item[0] == 'raw':
                _, xb, yb, steps = item
                eng = self.model._engine(xb.shape[0])
                eng.alloc_input_ring(self.slots)
                eng.load_input(xb, yb)
                if eng.ring_slots() >= self.slots:
                    self.ring[xb.shape[0]] = eng
                self.ready.set()
                yield eng, None, steps
            else:
                yield item[1], item[2], item[3]

    def close(self):
        """Also on an exception / an early stop in the training thread: the stager and its pool are JOINED (no thread of this fit()
        survives it), then the rings are reset (slots staged but never fed are free again)."""
        self.cancel.set()
        self.ready.set()
        while self.th.is_alive():
            try:
                self.q.get_nowait()
            except queue.Empty:
                pass
            self.th.join(0.05)
        for eng in self.ring.values():
            eng.reset_input_ring()
