"""Model factory: ``get_model(config, metrics)`` / ``create_unet(config, ...)``.

Host-side mirror of the reference's factory (src/models/Unets.py:61-133 ``create_unet``, :755-869
``unet``, :984-998 ``get_model``) and of the block functions it calls (src/models/KerasLayers.py:660-693
``conv_layer_fn``, :696-723 ``downsampling_block_fn``, :726-777 ``upsampling_block_fn``): same flat
UPPER-CASE config dict, same defaults (including the ``USE_UPSAMPLE='False'`` string that is truthy),
same Keras layer names / order / parameter counts, same compiled-model surface.  Nothing here is
TensorFlow: the builder emits (a) the Keras-style layer table used by ``summary()`` and the weight
interchange order and (b) a FUSED execution plan for the HIP engine, in which UpSampling2D and
Concatenate are addressing modes of the consuming conv and BatchNorm/Activation/Dropout/MaxPool ride
on one pass over the conv output.
"""
from __future__ import annotations

import numpy as np

_BASE = {'Conv': 'conv%dd', 'BatchNormalization': 'batch_normalization', 'Dropout': 'dropout',
         'MaxPooling': 'max_pooling%dd', 'UpSampling': 'up_sampling%dd', 'Concatenate': 'concatenate',
         'ConvTranspose': 'conv%dd_transpose', 'Activation': 'activation'}


class LayerSpec(dict):
    """One Keras layer of the table: name, type, inputs, shape (without batch), params, attrs."""
    __getattr__ = dict.__getitem__


class ConvStage:
    """One fused stage of the execution plan:
    [virtual upsample/concat] -> conv3x3 (+bias, +act when it precedes BN) -> [BN] -> [act] -> [dropout] -> [pool]"""

    def __init__(self, **kw):
        self.conv = None          # conv layer name (parameters: kernel, bias)
        self.src0 = None          # tensor name
        self.up0 = 0
        self.src1 = None
        self.c0 = self.c1 = 0
        self.cout = 0
        self.h = self.w = 0       # output spatial size (in-plane)
        self.d = 1                # output depth (frames of a 3-D volume; 1 for 2-D graphs)
        self.act_conv = None      # activation fused in the conv epilogue
        self.bn = None            # BN layer name
        self.act_post = None      # activation after BN (BN_FIRST)
        self.drop = None          # (dropout layer name, rate, stream id)
        self.pool = False
        self.z = self.y = self.pooled = None   # tensor names produced
        self.transpose = False
        self.__dict__.update(kw)

    @property
    def cin(self):
        return self.c0 + self.c1


class UnetPlan:
    """Layer table + fused plan + parameter inventory for one config."""

    def __init__(self, config):
        cfg = config
        self.config = dict(cfg)
        self.dim = list(cfg.get('DIM', [224, 224]))
        self.img_channels = cfg.get('IMG_CHANNELS', 1)
        self.activation = cfg.get('ACTIVATION', 'elu')
        self.batch_norm = cfg.get('BATCH_NORMALISATION', False)
        self.use_upsample = cfg.get('USE_UPSAMPLE', 'False')          # Unets.py:86: string default, truthy
        self.pad = cfg.get('PAD', 'same')
        self.kernel_init = cfg.get('KERNEL_INIT', 'he_normal')
        self.mask_classes = cfg.get('MASK_CLASSES', 3)
        self.ndims = len(cfg.get('DIM', [10, 224, 224]))
        self.m_pool = tuple(cfg.get('M_POOL', (1, 2, 2)))[-self.ndims:]
        self.f_size = tuple(cfg.get('F_SIZE', (3, 3, 3)))[-self.ndims:]
        self.filters = cfg.get('FILTERS', 16)
        self.drop_min = cfg.get('DROPOUT_MIN', 0.3)
        self.drop_max = cfg.get('DROPOUT_MAX', 0.5)
        self.bn_first = cfg.get('BN_FIRST', False)
        self.depth = cfg.get('DEPTH', 4)
        self.layers = []
        self.stages = []
        self.head = None
        self._counts = {}
        self._build()

    # -- Keras-style bookkeeping ------------------------------------------------------------------
    def _name(self, kind):
        base = _BASE[kind]
        base = base % self.ndims if '%' in base else base
        k = self._counts.get(base, 0)
        self._counts[base] = k + 1
        return base if k == 0 else '%s_%d' % (base, k)

    def _add(self, kind, type_name, inputs, shape, params=0, name=None, **attrs):
        spec = LayerSpec(name=name or self._name(kind), type=type_name, inputs=list(inputs), shape=tuple(shape),
                         params=int(params), **attrs)
        self.layers.append(spec)
        return spec

    def _conv(self, src, filters, kernel, act, name=None):
        cin = src.shape[-1]
        return self._add('Conv', 'Conv%dD' % self.ndims, [src.name], src.shape[:-1] + (filters,),
                         int(np.prod(kernel)) * cin * filters + filters, name=name, kernel=tuple(kernel),
                         activation=act, cin=cin, cout=filters)

    # -- KerasLayers.py:660-693 -------------------------------------------------------------------
    def _conv_layer(self, src, filters, stage):
        if self.bn_first:
            c = self._conv(src, filters, self.f_size, None)
            stage.conv, stage.act_conv = c.name, None
            out = c
            if self.batch_norm:
                out = self._add('BatchNormalization', 'BatchNormalization', [c.name], c.shape, 4 * filters, channels=filters)
                stage.bn = out.name
            out = self._add('Activation', 'Activation', [out.name], out.shape, activation=self.activation)
            stage.act_post = self.activation
        else:
            c = self._conv(src, filters, self.f_size, self.activation)
            stage.conv, stage.act_conv = c.name, self.activation
            out = c
            if self.batch_norm:
                out = self._add('BatchNormalization', 'BatchNormalization', [c.name], c.shape, 4 * filters, channels=filters)
                stage.bn = out.name
        stage.cout = filters
        stage.z = c.name
        stage.y = out.name
        return out

    def _dropout(self, src, rate, stage):
        d = self._add('Dropout', 'Dropout', [src.name], src.shape, rate=float(rate))
        stage.drop = (d.name, float(rate), len([s for s in self.stages if s.drop]) + 1)
        stage.y = d.name
        return d

    @staticmethod
    def _dhw(shape):
        """(depth, height, width) of a layer shape (spatial..., C); depth 1 for 2-D graphs"""
        sp = tuple(shape[:-1])
        return ((1,) + sp) if len(sp) == 2 else sp

    def _stage(self, src, **kw):
        d, h, w = self._dhw(src.shape)
        st = ConvStage(src0=src.name, c0=src.shape[-1], d=d, h=h, w=w, **kw)
        self.stages.append(st)
        return st

    def _build(self):
        if self.pad != 'same':
            raise NotImplementedError("PAD='%s': only 'same' (the reference default) is built" % self.pad)
        nd = self.ndims
        dropouts = [round(float(v), 1) for v in np.linspace(self.drop_min, self.drop_max, self.depth)]   # Unets.py:105-106
        self.dropouts = dropouts
        x = self._add(None, 'InputLayer', [], tuple(self.dim) + (self.img_channels,), name='input_1')
        f = self.filters
        skips = []
        for level in range(self.depth):                                   # Unets.py:786-807
            s1 = self._stage(x)
            c = self._conv_layer(x, f, s1)
            c = self._dropout(c, dropouts[level], s1)
            s2 = self._stage(c)
            c = self._conv_layer(c, f, s2)
            p = self._add('MaxPooling', 'MaxPooling%dD' % nd, [c.name],
                          tuple(a // b for a, b in zip(c.shape[:-1], self.m_pool)) + (f,), pool=self.m_pool)
            s2.pool, s2.pooled = True, p.name
            skips.append(c)
            x = p
            f *= 2
        s1 = self._stage(x)                                               # Unets.py:809-816
        c = self._conv_layer(x, f, s1)
        c = self._dropout(c, self.drop_max, s1)
        s2 = self._stage(c)
        lower = self._conv_layer(c, f, s2)
        drops = list(dropouts)
        for level in range(self.depth):                                   # Unets.py:819-836
            skip = skips.pop()
            f //= 2
            up_shape = tuple(a * b for a, b in zip(lower.shape[:-1], self.m_pool))
            if self.use_upsample:                                         # KerasLayers.py:753-759
                u = self._add('UpSampling', 'UpSampling%dD' % nd, [lower.name], up_shape + (lower.shape[-1],), size=self.m_pool)
                ud, uh, uw = self._dhw(up_shape + (0,))
                su = ConvStage(src0=lower.name, c0=lower.shape[-1], up0=1, d=ud, h=uh, w=uw)
                self.stages.append(su)
                uc = self._conv(u, f, self.f_size, self.activation)
                su.conv, su.act_conv, su.cout, su.z, su.y = uc.name, self.activation, f, uc.name, uc.name
            else:                                                         # KerasLayers.py:761-765
                cin = lower.shape[-1]
                uc = self._add('ConvTranspose', 'Conv%dDTranspose' % nd, [lower.name], up_shape + (f,),
                               int(np.prod(self.f_size)) * cin * f + f, kernel=tuple(self.f_size), strides=self.m_pool,
                               activation=self.activation, cin=cin, cout=f)
                ud, uh, uw = self._dhw(up_shape + (0,))
                su = ConvStage(src0=lower.name, c0=cin, up0=2, d=ud, h=uh, w=uw, transpose=True,
                               conv=uc.name, act_conv=self.activation, cout=f, z=uc.name, y=uc.name)
                self.stages.append(su)
            cat = self._add('Concatenate', 'Concatenate', [uc.name, skip.name], up_shape + (f + skip.shape[-1],))
            ud, uh, uw = self._dhw(up_shape + (0,))
            s1 = ConvStage(src0=uc.name, c0=f, src1=skip.name, c1=skip.shape[-1], d=ud, h=uh, w=uw)
            self.stages.append(s1)
            c = self._conv_layer(cat, f, s1)
            c = self._dropout(c, drops.pop(), s1)
            s2 = self._stage(c)
            lower = self._conv_layer(c, f, s2)
        head = self._conv(lower, self.mask_classes, (1,) * nd, 'sigmoid', name='unet')      # Unets.py:128
        hd_, hh_, hw_ = self._dhw(lower.shape)
        self.head = dict(conv=head.name, src=lower.name, cin=lower.shape[-1], k=self.mask_classes, d=hd_, h=hh_, w=hw_)

    # -- inventories ------------------------------------------------------------------------------
    def weight_specs(self):
        """[(layer name, weight name, shape, trainable, initializer)] in Keras get_weights() order."""
        out = []
        for l in self.layers:
            if l.type.endswith('Transpose'):
                out.append((l.name, 'kernel', l.kernel + (l.cout, l.cin), True, self.kernel_init))
                out.append((l.name, 'bias', (l.cout,), True, 'zeros'))
            elif l.type.startswith('Conv'):
                init = 'glorot_uniform' if l.name == 'unet' else self.kernel_init
                out.append((l.name, 'kernel', l.kernel + (l.cin, l.cout), True, init))
                out.append((l.name, 'bias', (l.cout,), True, 'zeros'))
            elif l.type == 'BatchNormalization':
                c = l.channels
                out += [(l.name, 'gamma', (c,), True, 'ones'), (l.name, 'beta', (c,), True, 'zeros'),
                        (l.name, 'moving_mean', (c,), False, 'zeros'), (l.name, 'moving_variance', (c,), False, 'ones')]
        return out

    def count_params(self):
        total = sum(l.params for l in self.layers)
        non_tr = sum(2 * l.channels for l in self.layers if l.type == 'BatchNormalization')
        return total, total - non_tr, non_tr

    def summary_rows(self):
        return [(l.name, l.type, (None,) + tuple(l.shape), l.params, tuple(l.inputs)) for l in self.layers]

    def flops_per_slice(self):
        """Conv MACs x2: forward, and forward+backward (dgrad + wgrad for every conv, no dgrad for the first
        layer) -- the algorithmic work BASELINE.md section 3 prices the roofline with."""
        fwd = bwd = 0.0
        first = True
        for l in self.layers:
            if l.type.startswith('Conv'):
                spatial = float(np.prod(l.shape[:-1]))
                taps = float(np.prod(l.kernel))
                if l.type.endswith('Transpose'):
                    spatial = spatial / float(np.prod(l.strides))
                f = 2.0 * spatial * taps * l.cin * l.cout
                fwd += f
                bwd += f if first else 2.0 * f
                first = False
        return fwd, fwd + bwd

    def ideal_bytes_per_slice(self, elem_bytes):
        """3 * sum (Cin + Cout) * H*W * bytes: every conv reads its input and writes its output once in
        forward, dgrad and wgrad."""
        tot = 0.0
        for l in self.layers:
            if l.type.startswith('Conv'):
                tot += (l.cin + l.cout) * float(np.prod(l.shape[:-1]))
        return 3.0 * tot * elem_bytes


def create_unet(config, metrics=None, networkname='unet', single_model=True, supervision=False):
    """Factory for the 2-D heatmap-regression U-Net (reference: Unets.py:61-133).

    ``config`` is the reference's flat UPPER-CASE dict; it is never mutated.  ``LOSS_FUNCTION`` may be a
    callable / loss object from ``Loss_and_metrics`` (or the strings 'mse' / 'BcdDiceLoss').  Returns a compiled
    Keras-like ``Model`` whose compute runs on the MI355X HIP engine.
    """
    from .keras_model import Model
    from .ModelUtils import get_optimizer
    from . import Loss_and_metrics as metr
    if supervision:
        raise NotImplementedError('supervision=True (Unets.py:840-863) is off in Train (train_model.py:83) and not built')
    if not single_model:
        raise NotImplementedError('single_model=False (stacked 2D->3D wrappers, Unets.py:136-645) is out of scope')
    plan = UnetPlan(config)
    model = Model(plan, name=networkname)
    metrics = [metr.binary_accuracy] if metrics is None else metrics
    loss_f = config.get('LOSS_FUNCTION', metr.categorical_crossentropy)
    model.compile(optimizer=get_optimizer(config, networkname), loss={'unet': loss_f}, metrics=metrics)
    return model


def get_model(config=dict(), metrics=None):
    """Reference: Unets.py:984-998 -- the LOAD branch is a no-op there too."""
    if config.get('LOAD', False):
        pass
    return create_unet(config, metrics)
