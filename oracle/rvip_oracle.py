"""NumPy restatement of the reference's heatmap-regression U-Net training step.

TEST INFRASTRUCTURE (see oracle/__init__.py): a checker, never the product.
PARITY UNPINNED: the reference holds no numerical vectors for this path; the
arithmetic lives in tensorflow==2.3.0 (environment.yml:126), absent here.  What
is pinned: ``build_graph`` reproduces the reference's stored model.summary()
(notebooks/Train/Train_tests.ipynb:440-577) line by line, and every op below is
cross-checked against PyTorch-CPU in tests/test_oracle.py.

Each function cites the reference text it restates (paths relative to
/root/reference).  Layout is Keras' own: activations NHWC, conv kernels HWIO,
transpose-conv kernels HWOI, parameters in ``layer.get_weights()`` order.
All maths runs in the dtype of the inputs (float64 for gradient checks, float32
for parity runs).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

BN_MOMENTUM = 0.99      # tf.keras.layers.BatchNormalization default (KerasLayers.py:684 passes none)
BN_EPS = 1e-3           # idem
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-7   # tf.keras.optimizers.Adam defaults (ModelUtils.py:106-107)


# ----------------------------------------------------------------------------------------------
# graph builder: src/models/Unets.py:61-133 (create_unet), :755-869 (unet),
#                src/models/KerasLayers.py:660-777 (conv_layer_fn / down / up blocks)
# ----------------------------------------------------------------------------------------------
_KERAS_BASE = {
    'Conv2D': 'conv2d', 'Conv3D': 'conv3d', 'BatchNormalization': 'batch_normalization',
    'Dropout': 'dropout', 'MaxPooling2D': 'max_pooling2d', 'MaxPooling3D': 'max_pooling3d',
    'UpSampling2D': 'up_sampling2d', 'UpSampling3D': 'up_sampling3d', 'Concatenate': 'concatenate',
    'Conv2DTranspose': 'conv2d_transpose', 'Conv3DTranspose': 'conv3d_transpose',
    'Activation': 'activation',
}


class _Namer:
    """Keras auto-naming in a fresh session: first instance is the bare snake-case class name,
    later ones get _1, _2, ... (matches Train_tests.ipynb:446-572)."""

    def __init__(self):
        self.count = {}

    def __call__(self, cls):
        base = _KERAS_BASE[cls]
        n = self.count.get(base, 0)
        self.count[base] = n + 1
        return base if n == 0 else '%s_%d' % (base, n)


def build_graph(config):
    """Layer table of the model ``create_unet(config, supervision=False)`` builds.

    Returns a list of dicts in Keras creation order:
    ``{name, type, inputs:[names], shape:(spatial..., C), params:int, ...attrs}``.
    Config keys and defaults: Unets.py:77-100.
    """
    cfg = config
    dim = list(cfg.get('DIM', [224, 224]))
    img_ch = cfg.get('IMG_CHANNELS', 1)
    activation = cfg.get('ACTIVATION', 'elu')
    batch_norm = cfg.get('BATCH_NORMALISATION', False)
    use_upsample = cfg.get('USE_UPSAMPLE', 'False')       # Unets.py:86 -- the *string*, truthy
    pad = cfg.get('PAD', 'same')
    mask_classes = cfg.get('MASK_CLASSES', 3)
    ndims = len(cfg.get('DIM', [10, 224, 224]))
    m_pool = tuple(cfg.get('M_POOL', (1, 2, 2)))[-ndims:]
    f_size = tuple(cfg.get('F_SIZE', (3, 3, 3)))[-ndims:]
    filters = cfg.get('FILTERS', 16)
    drop_1 = cfg.get('DROPOUT_MIN', 0.3)
    drop_3 = cfg.get('DROPOUT_MAX', 0.5)
    bn_first = cfg.get('BN_FIRST', False)
    depth = cfg.get('DEPTH', 4)
    if pad != 'same':
        raise NotImplementedError("oracle restates PAD='same' only (the reference default)")
    dropouts = [round(float(i), 1) for i in np.linspace(drop_1, drop_3, depth)]   # Unets.py:105-106

    nd = '%dD' % ndims
    namer = _Namer()
    layers = []

    def add(cls, inputs, shape, params=0, name=None, **attrs):
        lay = dict(name=name or namer(cls), type=cls, inputs=list(inputs), shape=tuple(shape),
                   params=int(params), **attrs)
        layers.append(lay)
        return lay

    def kvol(k):
        return int(np.prod(k))

    def conv(src, f, k, act, name=None):
        cin = src['shape'][-1]
        return add('Conv' + nd, [src['name']], src['shape'][:-1] + (f,), kvol(k) * cin * f + f,
                   name=name, kernel=tuple(k), activation=act, cin=cin, cout=f)

    def bn(src):
        c = src['shape'][-1]
        return add('BatchNormalization', [src['name']], src['shape'], 4 * c, channels=c)

    def conv_layer(src, f):                                  # KerasLayers.py:660-693
        if bn_first:
            c = conv(src, f, f_size, None)
            c = bn(c) if batch_norm else c
            return add('Activation', [c['name']], c['shape'], activation=activation)
        c = conv(src, f, f_size, activation)
        return bn(c) if batch_norm else c

    def dropout(src, rate):
        return add('Dropout', [src['name']], src['shape'], rate=float(rate))

    inp = add('InputLayer', [], tuple(dim) + (img_ch,), name='input_1')

    encoder = []
    x = inp
    f = filters
    for l in range(depth):                                   # Unets.py:786-807, KerasLayers.py:696-723
        c = conv_layer(x, f)
        c = dropout(c, dropouts[l])
        c = conv_layer(c, f)
        spatial = tuple(s // p for s, p in zip(c['shape'][:-1], m_pool))
        p = add('MaxPooling' + nd, [c['name']], spatial + (f,), pool=m_pool)
        encoder.append((c, p))
        x = p
        f *= 2
    c = conv_layer(x, f)                                     # Unets.py:809-816
    c = dropout(c, drop_3)
    c = conv_layer(c, f)
    lower = c
    drops = list(dropouts)
    for l in range(depth):                                   # Unets.py:819-836, KerasLayers.py:726-777
        skip = encoder.pop()[0]
        f //= 2
        if use_upsample:
            spatial = tuple(s * p for s, p in zip(lower['shape'][:-1], m_pool))
            u = add('UpSampling' + nd, [lower['name']], spatial + (lower['shape'][-1],), size=m_pool)
            u = conv(u, f, f_size, activation)               # bias + activation, no BN (:758-759)
        else:
            cin = lower['shape'][-1]
            spatial = tuple(s * p for s, p in zip(lower['shape'][:-1], m_pool))
            u = add('Conv%sTranspose' % nd, [lower['name']], spatial + (f,),
                    kvol(f_size) * cin * f + f, kernel=tuple(f_size), strides=m_pool,
                    activation=activation, cin=cin, cout=f)
        cat = add('Concatenate', [u['name'], skip['name']], u['shape'][:-1] + (u['shape'][-1] + skip['shape'][-1],))
        c = conv_layer(cat, f)
        c = dropout(c, drops.pop())
        c = conv_layer(c, f)
        lower = c
    conv(lower, mask_classes, (1,) * ndims, 'sigmoid', name='unet')   # Unets.py:128
    return layers


def count_params(layers):
    total = sum(l['params'] for l in layers)
    non_trainable = sum(2 * l['channels'] for l in layers if l['type'] == 'BatchNormalization')
    return total, total - non_trainable, non_trainable


def summary_rows(layers):
    """(name, type, output shape with None batch, params, connected-to names) per layer."""
    return [(l['name'], l['type'], (None,) + tuple(l['shape']), l['params'], tuple(l['inputs'])) for l in layers]


# ----------------------------------------------------------------------------------------------
# initialisers (SURVEY 8(a) note 7): he_normal = VarianceScaling(2, fan_in, truncated_normal),
# head = Keras default glorot_uniform.  RNG streams cannot match TF's; shapes/statistics do.
# ----------------------------------------------------------------------------------------------
def he_normal(rng, shape, dtype=np.float32):
    fan_in = int(np.prod(shape[:-1]))
    std = math.sqrt(2.0 / fan_in) / 0.87962566103423978
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():                                          # truncated normal: resample beyond 2 sigma
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * std).astype(dtype)


def glorot_uniform(rng, shape, dtype=np.float32):
    rf = int(np.prod(shape[:-2]))
    fan_in, fan_out = rf * shape[-2], rf * shape[-1]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, shape).astype(dtype)


def init_params(layers, seed=42, dtype=np.float32):
    """name -> list of arrays in Keras get_weights() order (Conv: kernel,bias; BN: gamma,beta,mean,var)."""
    rng = np.random.default_rng(seed)
    params = OrderedDict()
    for l in layers:
        t = l['type']
        if t.startswith('Conv') and t.endswith('Transpose'):
            k = l['kernel'] + (l['cout'], l['cin'])          # Keras transpose-conv kernel is HWOI
            # fan_in of VarianceScaling on HWOI uses shape[-2] as "in": receptive * cout
            params[l['name']] = [he_normal(rng, k, dtype), np.zeros(l['cout'], dtype)]
        elif t.startswith('Conv'):
            k = l['kernel'] + (l['cin'], l['cout'])
            init = glorot_uniform if l['name'] == 'unet' else he_normal
            params[l['name']] = [init(rng, k, dtype), np.zeros(l['cout'], dtype)]
        elif t == 'BatchNormalization':
            c = l['channels']
            params[l['name']] = [np.ones(c, dtype), np.zeros(c, dtype), np.zeros(c, dtype), np.ones(c, dtype)]
    return params


# ----------------------------------------------------------------------------------------------
# primitive ops (TF/Keras 2.3 semantics, SURVEY 8(a) notes 1-9), forward + backward
# ----------------------------------------------------------------------------------------------
def conv2d_same_fwd(x, w, b=None):
    """Conv2D(padding='same', strides=1): cross-correlation, HWIO kernel, zero pad (k-1)//2 before."""
    kh, kw, ci, co = w.shape
    n, h, wd, _ = x.shape
    pt, pl = (kh - 1) // 2, (kw - 1) // 2
    xp = np.pad(x, ((0, 0), (pt, kh - 1 - pt), (pl, kw - 1 - pl), (0, 0)))
    y = np.zeros((n, h, wd, co), dtype=np.result_type(x, w))
    for i in range(kh):
        for j in range(kw):
            y += xp[:, i:i + h, j:j + wd, :] @ w[i, j]
    if b is not None:
        y += b
    return y


def conv2d_same_bwd(x, w, dy):
    """Returns (dx, dw, db) of conv2d_same_fwd."""
    kh, kw, ci, co = w.shape
    n, h, wd, _ = x.shape
    pt, pl = (kh - 1) // 2, (kw - 1) // 2
    xp = np.pad(x, ((0, 0), (pt, kh - 1 - pt), (pl, kw - 1 - pl), (0, 0)))
    dxp = np.zeros_like(xp, dtype=np.result_type(x, w, dy))
    dw = np.zeros_like(w, dtype=dxp.dtype)
    dy2 = dy.reshape(-1, co)
    for i in range(kh):
        for j in range(kw):
            dxp[:, i:i + h, j:j + wd, :] += dy @ w[i, j].T
            dw[i, j] = xp[:, i:i + h, j:j + wd, :].reshape(-1, ci).T @ dy2
    dx = dxp[:, pt:pt + h, pl:pl + wd, :]
    return dx, dw, dy2.sum(0)


def conv3d_same_fwd(x, w, b=None):
    """Conv3D(padding='same', strides=1) on NDHWC, DHWIO kernel (KerasLayers.py:679 slices f_size[:ndims]; cfg 5):
    cross-correlation, zero pad (k-1)//2 before / the rest after, like the 2-D op."""
    kd, kh, kw, ci, co = w.shape
    n, d, h, wd, _ = x.shape
    pd, pt, pl = (kd - 1) // 2, (kh - 1) // 2, (kw - 1) // 2
    xp = np.pad(x, ((0, 0), (pd, kd - 1 - pd), (pt, kh - 1 - pt), (pl, kw - 1 - pl), (0, 0)))
    y = np.zeros((n, d, h, wd, co), dtype=np.result_type(x, w))
    for a in range(kd):
        for i in range(kh):
            for j in range(kw):
                y += xp[:, a:a + d, i:i + h, j:j + wd, :] @ w[a, i, j]
    if b is not None:
        y += b
    return y


def conv3d_same_bwd(x, w, dy):
    """Returns (dx, dw, db) of conv3d_same_fwd."""
    kd, kh, kw, ci, co = w.shape
    n, d, h, wd, _ = x.shape
    pd, pt, pl = (kd - 1) // 2, (kh - 1) // 2, (kw - 1) // 2
    xp = np.pad(x, ((0, 0), (pd, kd - 1 - pd), (pt, kh - 1 - pt), (pl, kw - 1 - pl), (0, 0)))
    dxp = np.zeros_like(xp, dtype=np.result_type(x, w, dy))
    dw = np.zeros_like(w, dtype=dxp.dtype)
    dy2 = dy.reshape(-1, co)
    for a in range(kd):
        for i in range(kh):
            for j in range(kw):
                dxp[:, a:a + d, i:i + h, j:j + wd, :] += dy @ w[a, i, j].T
                dw[a, i, j] = xp[:, a:a + d, i:i + h, j:j + wd, :].reshape(-1, ci).T @ dy2
    dx = dxp[:, pd:pd + d, pt:pt + h, pl:pl + wd, :]
    return dx, dw, dy2.sum(0)


def conv_same_fwd(x, w, b=None):
    return conv3d_same_fwd(x, w, b) if w.ndim == 5 else conv2d_same_fwd(x, w, b)


def conv_same_bwd(x, w, dy):
    return conv3d_same_bwd(x, w, dy) if w.ndim == 5 else conv2d_same_bwd(x, w, dy)


def conv2d_transpose_same_fwd(x, w, b=None, stride=2):
    """Conv2DTranspose(k, strides=s, padding='same'): the input-gradient of a SAME stride-s conv
    (pad_before = 0 for k=3,s=2): out[s*i + k] += in[i] . W[k] cropped to s*N.  Kernel HWOI."""
    kh, kw, co, ci = w.shape
    n, h, wd, _ = x.shape
    s = stride
    oh, ow = h * s, wd * s
    # SAME forward conv on the (oh, ow) image: pad_total = max((h-1)*s + k - oh, 0), before = total//2
    pt = max((h - 1) * s + kh - oh, 0) // 2
    pl = max((wd - 1) * s + kw - ow, 0) // 2
    full = np.zeros((n, (h - 1) * s + kh, (wd - 1) * s + kw, co), dtype=np.result_type(x, w))
    for i in range(kh):
        for j in range(kw):
            full[:, i:i + (h - 1) * s + 1:s, j:j + (wd - 1) * s + 1:s, :] += x @ w[i, j].T
    y = np.zeros((n, oh, ow, co), dtype=full.dtype)
    src = full[:, pt:pt + oh, pl:pl + ow, :]
    y[:, :src.shape[1], :src.shape[2], :] = src
    if b is not None:
        y += b
    return y


def conv2d_transpose_same_bwd(x, w, dy, stride=2):
    kh, kw, co, ci = w.shape
    n, h, wd, _ = x.shape
    s = stride
    oh, ow = h * s, wd * s
    pt = max((h - 1) * s + kh - oh, 0) // 2
    pl = max((wd - 1) * s + kw - ow, 0) // 2
    fh, fw = (h - 1) * s + kh, (wd - 1) * s + kw
    dfull = np.zeros((n, fh, fw, co), dtype=np.result_type(x, w, dy))
    eh, ew = min(oh, fh - pt), min(ow, fw - pl)
    dfull[:, pt:pt + eh, pl:pl + ew, :] = dy[:, :eh, :ew, :]
    dx = np.zeros_like(x, dtype=dfull.dtype)
    dw = np.zeros_like(w, dtype=dfull.dtype)
    x2 = x.reshape(-1, ci)
    for i in range(kh):
        for j in range(kw):
            g = dfull[:, i:i + (h - 1) * s + 1:s, j:j + (wd - 1) * s + 1:s, :]
            dx += g @ w[i, j]
            dw[i, j] = g.reshape(-1, co).T @ x2
    return dx, dw, dy.reshape(-1, co).sum(0)


def conv3d_transpose_same_fwd(x, w, b=None, strides=(1, 2, 2)):
    """Conv3DTranspose(k, strides=M_POOL, padding='same') (KerasLayers.py:762-765 with ndims = 3; the 3-D template pools (1, 2, 2)):
    per axis the input-gradient of a SAME stride-s conv, out[s*i + k] += in[i] . W[k] cropped to s*N from pad_before =
    max((N-1)*s + k - s*N, 0) // 2 (1 for k=3, s=1; 0 for k=3, s=2).  Kernel DHWOI, x NDHWC."""
    kd, kh, kw, co, ci = w.shape
    n, d, h, wd, _ = x.shape
    ks, dims = (kd, kh, kw), (d, h, wd)
    outs = [dims[a] * strides[a] for a in range(3)]
    fulls = [(dims[a] - 1) * strides[a] + ks[a] for a in range(3)]
    pads = [max(fulls[a] - outs[a], 0) // 2 for a in range(3)]
    full = np.zeros((n,) + tuple(fulls) + (co,), dtype=np.result_type(x, w))
    sd, sh, sw = strides
    for a in range(kd):
        for i in range(kh):
            for j in range(kw):
                full[:, a:a + (d - 1) * sd + 1:sd, i:i + (h - 1) * sh + 1:sh, j:j + (wd - 1) * sw + 1:sw, :] += x @ w[a, i, j].T
    y = np.zeros((n,) + tuple(outs) + (co,), dtype=full.dtype)
    src = full[:, pads[0]:pads[0] + outs[0], pads[1]:pads[1] + outs[1], pads[2]:pads[2] + outs[2], :]
    y[:, :src.shape[1], :src.shape[2], :src.shape[3], :] = src
    if b is not None:
        y += b
    return y


def conv3d_transpose_same_bwd(x, w, dy, strides=(1, 2, 2)):
    kd, kh, kw, co, ci = w.shape
    n, d, h, wd, _ = x.shape
    ks, dims = (kd, kh, kw), (d, h, wd)
    outs = [dims[a] * strides[a] for a in range(3)]
    fulls = [(dims[a] - 1) * strides[a] + ks[a] for a in range(3)]
    pads = [max(fulls[a] - outs[a], 0) // 2 for a in range(3)]
    ext = [min(outs[a], fulls[a] - pads[a]) for a in range(3)]
    dfull = np.zeros((n,) + tuple(fulls) + (co,), dtype=np.result_type(x, w, dy))
    dfull[:, pads[0]:pads[0] + ext[0], pads[1]:pads[1] + ext[1], pads[2]:pads[2] + ext[2], :] = dy[:, :ext[0], :ext[1], :ext[2], :]
    dx = np.zeros_like(x, dtype=dfull.dtype)
    dw = np.zeros_like(w, dtype=dfull.dtype)
    x2 = x.reshape(-1, ci)
    sd, sh, sw = strides
    for a in range(kd):
        for i in range(kh):
            for j in range(kw):
                g = dfull[:, a:a + (d - 1) * sd + 1:sd, i:i + (h - 1) * sh + 1:sh, j:j + (wd - 1) * sw + 1:sw, :]
                dx += g @ w[a, i, j]
                dw[a, i, j] = g.reshape(-1, co).T @ x2
    return dx, dw, dy.reshape(-1, co).sum(0)


def act_fwd(x, kind):
    if kind in (None, 'linear'):
        return x
    if kind == 'relu':
        return np.maximum(x, 0)
    if kind == 'elu':                                          # alpha = 1
        return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))
    if kind == 'sigmoid':
        return 1.0 / (1.0 + np.exp(-x))
    raise ValueError(kind)


def act_bwd(y, dy, kind):
    """Gradient through the activation expressed in terms of its OUTPUT y."""
    if kind in (None, 'linear'):
        return dy
    if kind == 'relu':
        return dy * (y > 0)
    if kind == 'elu':
        return dy * np.where(y > 0, 1.0, y + 1.0)
    if kind == 'sigmoid':
        return dy * y * (1.0 - y)
    raise ValueError(kind)


def bn_train_fwd(x, gamma, beta, eps=BN_EPS):
    """BatchNormalization(axis=-1), training=True: batch mean / BIASED variance normalise."""
    axes = tuple(range(x.ndim - 1))
    mean = x.mean(axes)
    var = x.var(axes)
    invstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * invstd
    return gamma * xhat + beta, (xhat, invstd, mean, var)


def bn_moving_update(mov_mean, mov_var, mean, var, count, momentum=BN_MOMENTUM, fused=True):
    """Fused 4-D kernel feeds the UNBIASED variance to the moving average (SURVEY 8(a) note 3)."""
    v = var * (count / max(count - 1.0, 1.0)) if fused else var
    return mov_mean * momentum + mean * (1 - momentum), mov_var * momentum + v * (1 - momentum)


def bn_infer_fwd(x, gamma, beta, mov_mean, mov_var, eps=BN_EPS):
    return gamma * (x - mov_mean) / np.sqrt(mov_var + eps) + beta


def bn_train_bwd(dy, gamma, cache):
    xhat, invstd, _, _ = cache
    axes = tuple(range(dy.ndim - 1))
    m = float(np.prod([dy.shape[a] for a in axes]))
    dbeta = dy.sum(axes)
    dgamma = (dy * xhat).sum(axes)
    dx = (gamma * invstd) * (dy - dbeta / m - xhat * (dgamma / m))
    return dx, dgamma, dbeta


def maxpool2x2_fwd(x, pool=(2, 2)):
    """MaxPooling2D(pool), valid; argmax = FIRST maximum in row-major window order."""
    ph, pw = pool
    n, h, w, c = x.shape
    oh, ow = h // ph, w // pw
    win = x[:, :oh * ph, :ow * pw, :].reshape(n, oh, ph, ow, pw, c).transpose(0, 1, 3, 5, 2, 4).reshape(n, oh, ow, c, ph * pw)
    idx = win.argmax(-1)                                       # numpy argmax returns the first max
    y = np.take_along_axis(win, idx[..., None], -1)[..., 0]
    return y, idx


def maxpool2x2_bwd(dy, idx, in_shape, pool=(2, 2)):
    ph, pw = pool
    n, h, w, c = in_shape
    oh, ow = h // ph, w // pw
    dwin = np.zeros((n, oh, ow, c, ph * pw), dtype=dy.dtype)
    np.put_along_axis(dwin, idx[..., None], dy[..., None], -1)
    dx = np.zeros(in_shape, dtype=dy.dtype)
    dx[:, :oh * ph, :ow * pw, :] = dwin.reshape(n, oh, ow, c, ph, pw).transpose(0, 1, 4, 2, 5, 3).reshape(n, oh * ph, ow * pw, c)
    return dx


def maxpool3d_fwd(x, pool=(1, 2, 2)):
    """MaxPooling3D(pool), valid, NDHWC; first maximum in row-major (d, h, w) window order."""
    pd, ph, pw = pool
    n, d, h, w, c = x.shape
    od, oh, ow = d // pd, h // ph, w // pw
    win = x[:, :od * pd, :oh * ph, :ow * pw, :].reshape(n, od, pd, oh, ph, ow, pw, c)
    win = win.transpose(0, 1, 3, 5, 7, 2, 4, 6).reshape(n, od, oh, ow, c, pd * ph * pw)
    idx = win.argmax(-1)
    y = np.take_along_axis(win, idx[..., None], -1)[..., 0]
    return y, idx


def maxpool3d_bwd(dy, idx, in_shape, pool=(1, 2, 2)):
    pd, ph, pw = pool
    n, d, h, w, c = in_shape
    od, oh, ow = d // pd, h // ph, w // pw
    dwin = np.zeros((n, od, oh, ow, c, pd * ph * pw), dtype=dy.dtype)
    np.put_along_axis(dwin, idx[..., None], dy[..., None], -1)
    dx = np.zeros(in_shape, dtype=dy.dtype)
    dx[:, :od * pd, :oh * ph, :ow * pw, :] = dwin.reshape(n, od, oh, ow, c, pd, ph, pw).transpose(0, 1, 5, 2, 6, 3, 7, 4).reshape(
        n, od * pd, oh * ph, ow * pw, c)
    return dx


def upsample_nearest_fwd(x, size=(2, 2)):
    for ax, sz in enumerate(size):
        if sz != 1:
            x = x.repeat(sz, axis=1 + ax)
    return x


def upsample_nearest_bwd(dy, size=(2, 2)):
    if len(size) == 3:
        n, d, h, w, c = dy.shape
        sd, sh, sw = size
        return dy.reshape(n, d // sd, sd, h // sh, sh, w // sw, sw, c).sum((2, 4, 6))
    n, h, w, c = dy.shape
    sh, sw = size
    return dy.reshape(n, h // sh, sh, w // sw, sw, c).sum((2, 4))


# ----------------------------------------------------------------------------------------------
# losses / metrics: train_model.py:184 (Keras MSE), Loss_and_metrics.py:165-171, 229-245, 134-163
# ----------------------------------------------------------------------------------------------
def mse_loss(y_true, y_pred, global_batch=None):
    """tf.keras.losses.MSE then Keras mean reduction == mean over every element.
    Returns (loss, dL/dy_pred).  With data parallelism Keras divides by the GLOBAL batch."""
    n_local = y_pred.shape[0]
    per_sample = y_pred[0].size
    gb = n_local if global_batch is None else global_batch
    diff = y_pred - y_true
    loss = (diff ** 2).sum() / (gb * per_sample)
    return loss, 2.0 * diff / (gb * per_sample)


def dice_coef(y_true, y_pred, smooth=1.0):
    inter = (y_true * y_pred).sum()
    return (2.0 * inter + smooth) / (y_true.sum() + y_pred.sum() + smooth)


def bce_dice_loss(y_true, y_pred, w_bce=0.5, w_dice=1.0, logits=None, global_batch=None, reduction='mean'):
    """Loss_and_metrics.py:229-245.  BCE is Keras binary_crossentropy (mean over the channel axis,
    then Keras' mean over B,H,W); on a Sigmoid-op output Keras uses the logits form (pass
    ``logits``), otherwise clips y_pred to [1e-7, 1-1e-7].  Dice is over the flattened replica batch
    and is not divided by the global batch (it is a scalar added to every pixel's loss).
    Returns (loss, dL/dy_pred [or dL/dlogits if logits is given]).

    ``reduction`` -- how Keras turns the un-reduced ``[B,H,W]`` tensor ``w_bce*BCE - w_dice*dice`` into the training objective.
    Two call forms exist in the reference and tf.keras 2.3 treats them DIFFERENTLY (TensorFlow is third-party, environment.yml:126,
    not runnable here: this is a reading of its source, cited by file, and stays "unpinned"):

    * 'mean' -- the FUNCTION ``bce_dice_loss`` (Train_tests.ipynb:219).  ``compile`` wraps a plain callable in
      ``LossFunctionWrapper`` (keras/engine/compile_utils.py ``LossesContainer._get_loss_object``), whose ``Loss.__call__``
      (keras/losses.py) reduces with SUM_OVER_BATCH_SIZE = mean over the B*H*W elements (keras/utils/losses_utils.py
      ``compute_weighted_loss``), then ``scale_loss_for_distribution`` divides by the replica count.
    * 'sum' -- the CLASS ``BceDiceLoss`` (Loss_and_metrics.py:207-226; train_model.py:178-184 builds it for "BcdDiceLoss", the
      template config's choice).  It subclasses ``tf.keras.losses.Loss`` but OVERRIDES ``__call__`` (:220-226), so the
      reduction in ``Loss.__call__`` never runs: ``LossesContainer.__call__`` receives the raw ``[B,H,W]`` tensor, applies only
      ``scale_loss_for_distribution`` (its ``reduction`` attribute is the inherited AUTO), and ``Model.train_step``
      (keras/engine/training.py) hands that tensor to ``_minimize`` -> ``tape.gradient(loss, variables)``; the gradient of a
      non-scalar target is the gradient of the SUM of its elements (python/eager/backprop.py ``GradientTape.gradient``).  The
      LOGGED loss is still the element mean (the ``Mean`` metric, keras/metrics.py, averages the tensor).
    Hence: same loss value, and  grad['sum'] == grad['mean'] * (B_local*H*W)  exactly (tests/test_oracle.py).  Under Adam the factor
    nearly cancels except against epsilon = 1e-7, which is why it is not harmless to pick the wrong one."""
    if reduction not in ('mean', 'sum'):
        raise ValueError(reduction)
    if y_pred.shape[-1] == 4:                                       # Loss_and_metrics.py:222-224 / :240-242: drop the background channel
        y_true, y_pred = y_true[..., -3:], y_pred[..., -3:]
        full = logits is not None and logits.shape[-1] == 4
        loss, g = bce_dice_loss(y_true, y_pred, w_bce, w_dice, logits[..., -3:] if full else logits, global_batch, reduction)
        gfull = np.zeros(y_pred.shape[:-1] + (4,), dtype=g.dtype)
        gfull[..., -3:] = g
        return loss, gfull
    n_local = y_pred.shape[0]
    gb = n_local if global_batch is None else global_batch
    per_sample = y_pred[0].size
    t, p = y_true, y_pred
    if logits is not None:
        z = logits
        bce = np.maximum(z, 0) - z * t + np.log1p(np.exp(-np.abs(z)))
        dbce_dz = (p - t)
    else:
        pc = np.clip(p, 1e-7, 1 - 1e-7)
        bce = -(t * np.log(pc) + (1 - t) * np.log(1 - pc))
        inside = (p > 1e-7) & (p < 1 - 1e-7)
        dbce_dp = np.where(inside, (pc - t) / (pc * (1 - pc)), 0.0)
    inter, st, sp = (t * p).sum(), t.sum(), p.sum()
    dice = (2 * inter + 1.0) / (st + sp + 1.0)
    # Keras: mean over B,H,W of [w_bce*mean_c(bce) - w_dice*dice]; the scalar dice is broadcast, and
    # SUM_OVER_BATCH_SIZE under a strategy divides by the global batch.
    scale = n_local / gb
    loss = w_bce * bce.sum() / (gb * per_sample) - w_dice * dice * scale
    ddice_dp = (2 * t * (st + sp + 1.0) - (2 * inter + 1.0)) / (st + sp + 1.0) ** 2
    if logits is not None:
        grad = w_bce * dbce_dz / (gb * per_sample) - w_dice * scale * ddice_dp * p * (1 - p)
    else:
        grad = w_bce * dbce_dp / (gb * per_sample) - w_dice * scale * ddice_dp
    if reduction == 'sum':
        grad = grad * float(n_local * int(np.prod(y_pred.shape[1:-1])))
    return loss, grad


def dice_metrics(y_true, y_pred):
    """dice_coef_labels / _lower / _upper (Loss_and_metrics.py:134-163)."""
    return {
        'dice_coef_labels': dice_coef(y_true[..., -3:], y_pred[..., -3:]),
        'dice_coef_lower': dice_coef(y_true[..., -2], y_pred[..., -2]),
        'dice_coef_upper': dice_coef(y_true[..., -1], y_pred[..., -1]),
    }


def bf16_round(a):
    """Round-to-nearest-even to bfloat16 and back (storage emulation for the bf16 device path)."""
    import torch
    a = np.asarray(a)
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy().astype(a.dtype)


def f16_round(a, scale=1.0):
    """Round-to-nearest-even to IEEE binary16 and back, at `scale` (storage emulation for the f16 device path; gradient
    tensors are stored multiplied by the loss scale)."""
    a = np.asarray(a)
    with np.errstate(over='ignore'):
        return ((a * scale).astype(np.float16).astype(a.dtype)) / a.dtype.type(scale)


def knife_edges(layers, cache, rel=3e-6):
    """Elements at which a float32 evaluation may legitimately take the other branch of a non-smooth op: ReLU
    pre-activations within `rel` of zero (relative to the tensor scale) and 2x2 max-pool windows whose two largest
    entries differ by less than `rel` (exact ties excluded: first-max is deterministic).  Returns a list of
    (layer name, count).  Used by the parity tests to tell an ill-posed comparison from a wrong kernel."""
    out = []
    t = cache['tensors']
    for l in layers:
        if l['type'] in ('Conv2D', 'Conv3D') and l.get('activation') == 'relu' and l['name'] in cache.get('pre', {}):
            a = np.abs(cache['pre'][l['name']])
            n = int((a < rel * max(a.max(), 1e-30)).sum())
            if n:
                out.append((l['name'], n))
        elif l['type'] == 'Activation' and l.get('activation') == 'relu':
            a = np.abs(t[l['inputs'][0]])
            n = int((a < rel * max(a.max(), 1e-30)).sum())
            if n:
                out.append((l['name'], n))
        elif l['type'] in ('MaxPooling2D', 'MaxPooling3D'):
            x = t[l['inputs'][0]]
            if x.ndim == 5:                                  # (1,2,2) pooling: frames are independent images
                if tuple(l['pool']) != (1, 2, 2):
                    continue
                x = x.reshape((-1,) + x.shape[2:])
            n_, h, w, c = x.shape
            win = x[:, :h // 2 * 2, :w // 2 * 2, :].reshape(n_, h // 2, 2, w // 2, 2, c).transpose(0, 1, 3, 5, 2, 4).reshape(-1, 4)
            srt = np.sort(win, -1)
            gap = srt[:, 3] - srt[:, 2]
            n = int(((gap > 0) & (gap < rel * max(np.abs(x).max(), 1e-30))).sum())
            if n:
                out.append((l['name'], n))
    return out


def landmark_argmax(heat):
    """Flat argmax per (slice, channel), row-major over (H, W), first max wins.  5-D volumes
    [B,T,H,W,C] are taken frame by frame -> [B,T,C]."""
    if heat.ndim == 5:
        b, t = heat.shape[:2]
        return landmark_argmax(heat.reshape((b * t,) + heat.shape[2:])).reshape(b, t, -1)
    n, h, w, c = heat.shape
    return heat.transpose(0, 3, 1, 2).reshape(n, c, h * w).argmax(-1).astype(np.int64)


def threshold_mask(heat, thr=0.5):
    """predict_model.py:149-156: per-channel heat > 0.5."""
    return heat > thr


def centroid_landmarks(heat, thr=0.5):
    """evaluate_cv.py:418-442: mean (y, x) of the thresholded label; NaN when empty."""
    n, h, w, c = heat.shape
    out = np.full((n, c, 2), np.nan)
    for i in range(n):
        for k in range(c):
            ys, xs = np.where(heat[i, :, :, k] > thr)
            if ys.size:
                out[i, k] = (ys.mean(), xs.mean())
    return out


def flat_labels(pred, thr=0.5):
    """predict_model.py:149-156 / evaluate_cv.py (preds_flat): 0, then c+1 where pred[..., c] > thr; later channels
    override earlier ones.  pred [N,H,W,C] -> uint8 [N,H,W]."""
    out = np.zeros(pred.shape[:-1], np.uint8)
    for c in range(pred.shape[-1]):
        out[pred[..., c] > thr] = c + 1
    return out


def clean_2d_cc(flat):
    """Postprocess.py:108-120 clean_3d_prediction_2d_cc: per slice and label value keep the largest 4-connected
    component (cv2.connectedComponentsWithStats(mask, 4); np.argmax -> the first of equally large ones, and cv2 / scipy
    both number components in raster order of their first pixel).  `np.unique(s)[1:]` is the reference's way of skipping
    the background: a slice without background loses its smallest label instead (kept as is)."""
    import scipy.ndimage
    four = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    out = np.zeros_like(flat)
    for i, s in enumerate(flat):
        for val in np.unique(s)[1:]:
            lab, n = scipy.ndimage.label(s == val, structure=four)
            sizes = np.bincount(lab.ravel())[1:]
            out[i][lab == 1 + int(np.argmax(sizes))] = val
    return out


def mean_rvip_points(flat, n_labels=2):
    """evaluate_cv.py:418-442 get_mean_rvip_2d per slice: mean (y, x) of every label's pixels, NaN (the reference's
    None) when absent.  Like the reference it takes `np.unique(slice)[1:]` as the labels, so on a slice without any
    background pixel the smallest label present gets no point.  -> float64 [N, n_labels, 2]"""
    out = np.full((flat.shape[0], n_labels, 2), np.nan)
    for i, s in enumerate(flat):
        for v in np.unique(s)[1:]:
            ys, xs = np.where(s == v)
            out[i, int(v) - 1] = (ys.mean(), xs.mean())
    return out


# ----------------------------------------------------------------------------------------------
# optimiser: Keras OptimizerV2 Adam (ModelUtils.py:106-107)
# ----------------------------------------------------------------------------------------------
def adam_step(theta, g, m, v, t, lr, b1=ADAM_B1, b2=ADAM_B2, eps=ADAM_EPS):
    """t = iterations + 1.  epsilon sits OUTSIDE the bias correction (differs from torch.optim.Adam)."""
    lr_t = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    theta = theta - lr_t * m / (np.sqrt(v) + eps)
    return theta, m, v


# ----------------------------------------------------------------------------------------------
# data contract: Generators.py:376-398, Preprocess.py:425-437, 471-491
# ----------------------------------------------------------------------------------------------
def normalise_minmax(a):
    import sys
    return (a - a.min()) / (a.max() - a.min() + sys.float_info.epsilon)


def gaussian_targets(mask_onehot, sigma):
    """GAUS branch: per-channel scipy gaussian_filter then GLOBAL min-max (Generators.py:385-391)."""
    import scipy.ndimage
    g = np.stack([scipy.ndimage.gaussian_filter(mask_onehot[..., c].astype(np.float32), sigma)
                  for c in range(mask_onehot.shape[-1])], axis=-1)
    return normalise_minmax(g)


# ----------------------------------------------------------------------------------------------
# the model
# ----------------------------------------------------------------------------------------------
class OracleUNet:
    """Executes build_graph(config) with the primitives above."""

    def __init__(self, config, params=None, seed=42, dtype=np.float32, quant=None, quant_grad=None):
        """quant: optional storage-rounding emulation (e.g. ``bf16_round``).  It is applied exactly where the
        device path materialises a low-precision tensor: network input, packed 3x3 kernels (not the first
        Cin=1 layer, not the head: those read the fp32 masters), every conv output, the END of each fused
        BN->[act]->[dropout] chain, and in backward the conv-output gradient and every data-gradient tensor."""
        self.config = dict(config)
        self.layers = build_graph(config)
        self.dtype = dtype
        self.quant = quant
        self.quant_grad = quant_grad if quant_grad is not None else quant     # gradient tensors (f16: rounded at the loss scale)
        consumers = {}
        for l in self.layers:
            for i in l['inputs']:
                consumers.setdefault(i, []).append(l['type'])
        self._mat = set()
        for l in self.layers:
            t = l['type']
            if t in ('InputLayer', 'MaxPooling2D', 'MaxPooling3D') or (t.startswith('Conv') and l['name'] != 'unet'):
                self._mat.add(l['name'])
            elif t in ('BatchNormalization', 'Activation', 'Dropout'):
                if not any(c in ('Activation', 'Dropout') for c in consumers.get(l['name'], [])):
                    self._mat.add(l['name'])
        convs = [l['name'] for l in self.layers if l['type'].startswith('Conv')]
        self._qweights = set(convs[1:-1])
        self.params = params if params is not None else init_params(self.layers, seed, dtype)
        self.params = OrderedDict((k, [np.asarray(a, dtype) for a in v]) for k, v in self.params.items())
        self.lr = float(config.get('LEARNING_RATE', 0.001))
        self.iterations = 0
        self.opt_m = None
        self.opt_v = None
        self.ndims = len(self.layers[0]['shape']) - 1

    # -- forward ---------------------------------------------------------------------------
    def forward(self, x, training=False, dropout_masks=None):
        """dropout_masks: name -> {0,1} keep-mask array (training only; None = dropout disabled)."""
        x = np.asarray(x, self.dtype)
        t = {}
        cache = {}
        for l in self.layers:
            name, ty = l['name'], l['type']
            ins = [t[i] for i in l['inputs']]
            if ty == 'InputLayer':
                out = x
            elif ty in ('Conv2D', 'Conv3D'):
                w, b = self.params[name]
                if self.quant is not None and name in self._qweights:
                    w = self.quant(w)
                pre = conv_same_fwd(ins[0], w, b)
                out = act_fwd(pre, l['activation'])
                cache.setdefault('pre', {})[name] = pre
                if name == 'unet':
                    cache['logits'] = pre
            elif ty == 'Conv2DTranspose':
                w, b = self.params[name]
                out = act_fwd(conv2d_transpose_same_fwd(ins[0], w, b, l['strides'][0]), l['activation'])
            elif ty == 'Conv3DTranspose':
                w, b = self.params[name]
                out = act_fwd(conv3d_transpose_same_fwd(ins[0], w, b, tuple(l['strides'])), l['activation'])
            elif ty == 'Activation':
                out = act_fwd(ins[0], l['activation'])
            elif ty == 'BatchNormalization':
                g, b, mm, mv = self.params[name]
                if training:
                    out, c = bn_train_fwd(ins[0], g, b)
                    cache[name] = c
                else:
                    out = bn_infer_fwd(ins[0], g, b, mm, mv)
            elif ty == 'Dropout':
                if training and dropout_masks is not None and name in dropout_masks:
                    keep = 1.0 - l['rate']
                    mask = np.asarray(dropout_masks[name], self.dtype)
                    out = ins[0] * mask / self.dtype(keep)
                    cache[name] = mask
                else:
                    out = ins[0]
            elif ty == 'MaxPooling2D':
                out, idx = maxpool2x2_fwd(ins[0], l['pool'])
                cache[name] = idx
            elif ty == 'MaxPooling3D':
                out, idx = maxpool3d_fwd(ins[0], l['pool'])
                cache[name] = idx
            elif ty in ('UpSampling2D', 'UpSampling3D'):
                out = upsample_nearest_fwd(ins[0], l['size'])
            elif ty == 'Concatenate':
                out = np.concatenate(ins, axis=-1)
            else:
                raise NotImplementedError(ty)
            if self.quant is not None and name in self._mat:
                out = self.quant(out)
            t[name] = out
        cache['tensors'] = t
        return t['unet'], cache

    def predict(self, x):
        return self.forward(x, training=False)[0]

    # -- backward --------------------------------------------------------------------------
    def backward(self, cache, d_out, d_is_logit_grad=False):
        """d_out = dL/d(y_pred) (or dL/d(logits) if d_is_logit_grad).  Returns name -> [grads] for the
        trainable arrays only (Conv: dkernel, dbias; BN: dgamma, dbeta)."""
        t = cache['tensors']
        grads = OrderedDict()
        dt = {}

        def acc(name, g):
            dt[name] = g if name not in dt else dt[name] + g

        acc('unet', d_out)
        for l in reversed(self.layers):
            name, ty = l['name'], l['type']
            if name not in dt:
                continue
            dy = dt.pop(name)
            ins = l['inputs']
            q = self.quant_grad
            if q is not None and (name in self._mat or ty in ('UpSampling2D', 'UpSampling3D', 'Concatenate')):
                dy = q(dy)                                  # a materialised gradient tensor (sum rounded once)
            if ty == 'InputLayer':
                continue
            if ty in ('Conv2D', 'Conv3D'):
                w, _ = self.params[name]
                if q is not None and name in self._qweights:
                    w = self.quant(w)
                dpre = dy if (name == 'unet' and d_is_logit_grad) else act_bwd(t[name], dy, l['activation'])
                if q is not None and name != 'unet':
                    dpre = q(dpre)                          # dz: what wgrad / dgrad read
                dx, dw, db = conv_same_bwd(t[ins[0]], w, dpre)
                if q is not None:
                    dx = q(dx)
                grads[name] = [dw, db]
                acc(ins[0], dx)
            elif ty == 'Conv2DTranspose':
                w, _ = self.params[name]
                dpre = act_bwd(t[name], dy, l['activation'])
                dx, dw, db = conv2d_transpose_same_bwd(t[ins[0]], w, dpre, l['strides'][0])
                grads[name] = [dw, db]
                acc(ins[0], dx)
            elif ty == 'Conv3DTranspose':
                w, _ = self.params[name]
                dpre = act_bwd(t[name], dy, l['activation'])
                dx, dw, db = conv3d_transpose_same_bwd(t[ins[0]], w, dpre, tuple(l['strides']))
                grads[name] = [dw, db]
                acc(ins[0], dx)
            elif ty == 'Activation':
                acc(ins[0], act_bwd(t[name], dy, l['activation']))
            elif ty == 'BatchNormalization':
                g = self.params[name][0]
                dx, dg, db = bn_train_bwd(dy, g, cache[name])
                grads[name] = [dg, db]
                acc(ins[0], dx)
            elif ty == 'Dropout':
                if name in cache:
                    acc(ins[0], dy * cache[name] / self.dtype(1.0 - l['rate']))
                else:
                    acc(ins[0], dy)
            elif ty == 'MaxPooling2D':
                acc(ins[0], maxpool2x2_bwd(dy, cache[name], t[ins[0]].shape, l['pool']))
            elif ty == 'MaxPooling3D':
                acc(ins[0], maxpool3d_bwd(dy, cache[name], t[ins[0]].shape, l['pool']))
            elif ty in ('UpSampling2D', 'UpSampling3D'):
                acc(ins[0], upsample_nearest_bwd(dy, l['size']))
            elif ty == 'Concatenate':
                off = 0
                for i in ins:
                    c = t[i].shape[-1]
                    acc(i, dy[..., off:off + c])
                    off += c
            else:
                raise NotImplementedError(ty)
        return OrderedDict((k, grads[k]) for k in self.params if k in grads)

    # -- one training step (Keras train_step: fwd, loss, bwd, BN moving update, Adam) -------
    def loss_and_grads(self, x, y, loss='mse', dropout_masks=None, global_batch=None, **loss_kw):
        y = np.asarray(y, self.dtype)
        pred, cache = self.forward(x, training=True, dropout_masks=dropout_masks)
        if loss == 'mse':
            val, dpred = mse_loss(y, pred, global_batch)
            grads = self.backward(cache, dpred)
        elif loss == 'bce_dice':
            val, dlogit = bce_dice_loss(y, pred, logits=cache['logits'], global_batch=global_batch, **loss_kw)     # w_bce, w_dice, reduction
            grads = self.backward(cache, dlogit, d_is_logit_grad=True)
        else:
            raise ValueError(loss)
        return val, grads, pred, cache

    def apply_bn_moving(self, cache):
        """moving stats of every BN layer, each with its own count N*H*W (fused 4-D kernel: unbiased var; 5-D inputs
        take TF 2.3's non-fused path: biased var, SURVEY 8(a) note 3)."""
        for l in self.layers:
            if l['type'] == 'BatchNormalization' and l['name'] in cache:
                cnt = float(np.prod(cache['tensors'][l['inputs'][0]].shape[:-1]))
                _, _, mean, var = cache[l['name']]
                p = self.params[l['name']]
                mm, mv = bn_moving_update(p[2], p[3], mean, var, cnt, fused=self.ndims == 2)
                p[2], p[3] = mm.astype(self.dtype), mv.astype(self.dtype)

    def apply_adam(self, grads):
        if self.opt_m is None:
            self.opt_m = {k: [np.zeros_like(a) for a in g] for k, g in grads.items()}
            self.opt_v = {k: [np.zeros_like(a) for a in g] for k, g in grads.items()}
        self.iterations += 1
        for k, gs in grads.items():
            for i, g in enumerate(gs):
                th, m, v = adam_step(self.params[k][i], g.astype(self.dtype), self.opt_m[k][i], self.opt_v[k][i],
                                     self.iterations, self.lr)
                self.params[k][i], self.opt_m[k][i], self.opt_v[k][i] = th.astype(self.dtype), m, v

    def train_step(self, x, y, loss='mse', dropout_masks=None, global_batch=None):
        val, grads, pred, cache = self.loss_and_grads(x, y, loss, dropout_masks, global_batch)
        self.apply_bn_moving(cache)
        self.apply_adam(grads)
        return val, pred

    # -- weights in Keras order ------------------------------------------------------------
    def get_weights(self):
        return [a for v in self.params.values() for a in v]

    def set_weights(self, arrays):
        it = iter(arrays)
        for k, v in self.params.items():
            self.params[k] = [np.asarray(next(it), self.dtype).reshape(a.shape) for a in v]


# ----------------------------------------------------------------------------------------------
# seeded synthetic SAX-like data; shared by tests and the CPU baseline
# ----------------------------------------------------------------------------------------------
def synthetic_batch(batch, dim, n_classes=2, sigma=2.0, seed=42, blur=8.0):
    """x [B,H,W,1] float32 in [0,1] (low-pass noise, per-slice min-max, cf. Generators.py:379);
    y [B,H,W,C] Gaussian blobs at seeded centres, globally min-max'ed per slice (cf. :385-391).
    A 3-entry dim (T,H,W) gives volumes [B,T,H,W,*]: every frame is drawn like a slice (cine stack)."""
    import scipy.ndimage
    rng = np.random.default_rng(seed)
    if len(dim) == 3:
        t, h, w = dim
        xs, ys = synthetic_batch(batch * t, (h, w), n_classes, sigma, seed, blur)
        return xs.reshape(batch, t, h, w, 1), ys.reshape(batch, t, h, w, n_classes)
    h, w = dim
    x = np.empty((batch, h, w, 1), np.float32)
    y = np.empty((batch, h, w, n_classes), np.float32)
    margin = min(16, h // 4, w // 4)
    for b in range(batch):
        img = scipy.ndimage.gaussian_filter(rng.random((h, w)), min(blur, h / 8.0))
        x[b, ..., 0] = normalise_minmax(img)
        onehot = np.zeros((h, w, n_classes), np.float32)
        for c in range(n_classes):
            cy = int(rng.integers(margin, h - margin))
            cx = int(rng.integers(margin, w - margin))
            onehot[cy, cx, c] = 1.0
        y[b] = gaussian_targets(onehot, sigma)
    return x, y


def dropout_keep_masks(layers, batch, seed=0):
    """Seeded Bernoulli keep-masks for every Dropout layer (mask injection for parity runs)."""
    rng = np.random.default_rng(seed)
    return {l['name']: (rng.random((batch,) + l['shape']) >= l['rate']).astype(np.uint8)
            for l in layers if l['type'] == 'Dropout'}
