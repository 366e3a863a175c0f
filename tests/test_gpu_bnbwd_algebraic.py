"""BatchNormalization backward without its reduction pass (ABI 5): the pieces, each through the C ABI.

  sum g    = column sums of the consumer's data-gradient output   (rvip_conv3x3_fwd_sums: plain, channel split, 2x2 block sums,
             Dropout backward in the epilogue)
  sum g*y  = sum_{t,o} W[t][c][o] * dW[t][c][o]                   (rvip_conv3x3_wgrad with dot_rows)
  stage 1  = rvip_bn_bwd_coef, held against the classic rvip_bn_bwd_reduce on the same chain, and its exact in-kernel route for
             ill-conditioned channel blocks.
The identity is the adjoint relation of the conv (autodiff of Conv2D, KerasLayers.py:683-691): <dX, X> = <dY, W * X> per input
channel.  End-to-end parity of the engine that uses these lives in tests/test_gpu_model.py."""
import ctypes as C
import zlib

import numpy as np
import pytest
import torch

import cmr_landmark_detection_amd as rvip
from oracle import rvip_oracle as O
from test_gpu_ops import P, close, conv_desc, dev, down, ds, f32, ndt, pack, rnd, stream, tdt, up

pytestmark = pytest.mark.gpu
N = rvip._native


def _bit(c):
    """RVIP_BIT_OF_CHANNEL of include/rvip_hip.h"""
    return 8 * ((c & 15) >> 2) + 4 * ((c & 31) >> 4) + (c & 3)


def bit_planes(flags):
    """bool [n, h, w, C] -> the bit-plane layout of include/rvip_hip.h: uint32 [ceil(C/32)][n*h*w]"""
    c = flags.shape[-1]
    f = flags.reshape(-1, c).astype(np.uint64)
    out = np.zeros((-(-c // 32), f.shape[0]), np.uint64)
    for ch in range(c):
        out[ch // 32] |= f[:, ch] << np.uint64(_bit(ch))
    return out.astype(np.uint32)


def planes_to_flags(words, shape):
    n, h, w, c = shape
    words = np.asarray(words, np.uint32).reshape(-(-c // 32), n * h * w)
    return np.stack([(words[ch // 32] >> np.uint32(_bit(ch))) & 1 for ch in range(c)], -1).reshape(shape).astype(bool)


def _sums_launch(d, cols):
    L = N.lib()
    rows = L.rvip_conv3x3_fwd_sums_rows(C.byref(d))
    assert rows > 0
    buf = torch.full((rows, cols), 7.0, dtype=torch.float32, device=dev())
    N.call('rvip_conv3x3_fwd_sums', C.byref(d), P(buf), C.c_size_t(buf.numel() * 4), stream())
    return down(buf).astype(np.float64)


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', [(2, 24, 40, 32, 32), (1, 16, 16, 16, 40), (2, 40, 72, 64, 64), (2, 8, 72, 8, 8)])
@pytest.mark.parametrize('mode', ['plain', 'split', 'down2', 'dropout'])
def test_dgrad_epilogue_column_sums(shape, dtype, mode):
    """The launch with column sums stores what the plain launch stores (Dropout backward applied in 'dropout') and its partial rows add
    up to the column sums of the result (taken in fp32 in front of the storage rounding: equal to the sums of the stored tensor up
    to that rounding's noise, 2^-9 / sqrt(rows) relative for the 16-bit types)."""
    n, h, w, ci, co = shape                                         # the data gradient maps dy [.., co] to dx [.., ci]
    if mode == 'split' and ci < 64:
        pytest.skip('the LDS-DMA kernels split at multiples of 32 channels')
    rng = np.random.default_rng(zlib.crc32(repr((shape, mode)).encode()) % 1000)
    wt = rnd(rng.standard_normal((3, 3, ci, co)) * 0.2, dtype)
    dy = rnd(rng.standard_normal((n, h, w, co)), dtype)
    dyd = up(dy, dtype)
    _, wd = pack(wt, dtype)
    T = tdt(dtype)

    def out():
        if mode == 'down2':
            return torch.zeros((n, h // 2, w // 2, ci), dtype=T, device=dev()), None
        if mode == 'split':
            return torch.zeros((n, h, w, 32), dtype=T, device=dev()), torch.zeros((n, h, w, ci - 32), dtype=T, device=dev())
        return torch.zeros((n, h, w, ci), dtype=T, device=dev()), None

    def desc(y, y1):
        d = conv_desc(dyd, co, 0, None, 0, wd, None, y, y1, 32 if y1 is not None else 0, n, h, w, ci, 0, dtype)
        d.down2 = 1 if mode == 'down2' else 0
        return d
    ya, y1a = out()
    N.call('rvip_conv3x3_fwd', C.byref(desc(ya, y1a)), stream())
    yb, y1b = out()
    d = desc(yb, y1b)
    rate, lid = 0.3, 4
    if mode == 'dropout':                       # gate every channel by the keep bits of a Dropout layer (what rvip_bn_apply leaves behind)
        keep = ds.keep_mask((n, h, w, ci), rate, 99, 5, lid).astype(bool)
        kbits = torch.from_numpy(bit_planes(keep).view(np.int32)).to(dev())
        d.mask_bits, d.mask_channels, d.mask_scale = kbits.data_ptr(), ci, 1.0 / (1.0 - rate)
    if mode == 'split':                         # ... or only the first half of a split result (the ReLU backward of a BN-less stage)
        gate = rng.random((n, h, w, 32)) < 0.6
        gbits = torch.from_numpy(bit_planes(gate).view(np.int32)).to(dev())
    rows = _sums_launch(d, ci)
    stored = down(yb) if y1b is None else np.concatenate([down(yb), down(y1b)], -1)
    if mode == 'dropout':
        rdx, _, _ = O.conv2d_same_bwd(np.zeros((n, h, w, ci)), wt.astype(np.float64), dy.astype(np.float64))
        close(stored, rdx * keep / np.float32(1 - rate), dtype, 'masked data gradient')
        assert not stored[~keep].any()                              # dropped positions are exact zeros
    else:
        plain = down(ya) if y1a is None else np.concatenate([down(ya), down(y1a)], -1)
        np.testing.assert_array_equal(stored, plain)                # same bits as the launch without statistics
    want = stored.astype(np.float64).reshape(-1, ci).sum(0)
    got = rows.sum(0)
    nrows = stored.reshape(-1, ci).shape[0]
    rel = 1e-5 if dtype == 'f32' else 2.0 ** -7 / np.sqrt(nrows)
    tol = rel * np.abs(stored.astype(np.float64)).reshape(-1, ci).sum(0).max() + 1e-6
    assert np.abs(got - want).max() <= tol, (np.abs(got - want).max(), tol)
    if mode == 'split':                                             # sums_from: the first half's columns may be skipped, the second half's not
        yc, y1c = out()
        d2 = desc(yc, y1c)
        d2.sums_from = 32
        rows2 = _sums_launch(d2, ci)
        np.testing.assert_array_equal(rows2[:, 32:], rows[:, 32:])
        # the first half gated by bit planes: stored = plain where the bit is set, exact zero elsewhere; the second half untouched
        yd, y1d = out()
        d3 = desc(yd, y1d)
        d3.mask_bits, d3.mask_channels, d3.mask_scale = gbits.data_ptr(), 32, 1.0
        rows3 = _sums_launch(d3, ci)
        np.testing.assert_array_equal(down(yd), np.where(gate, down(ya), 0.0))
        np.testing.assert_array_equal(down(y1d), down(y1a))
        want3 = np.where(gate, down(ya).astype(np.float64), 0.0).reshape(-1, 32).sum(0)
        assert np.abs(rows3[:, :32].sum(0) - want3).max() <= rel * np.abs(down(ya).astype(np.float64)).reshape(-1, 32).sum(0).max() + 1e-6
        np.testing.assert_array_equal(rows3[:, 32:], rows[:, 32:])


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', [(2, 24, 40, 32, 32, 0), (1, 16, 16, 16, 40, 0), (2, 40, 72, 64, 64, 0), (1, 32, 32, 64, 256, 0),
                                   (1, 16, 16, 128, 512, 0), (2, 24, 40, 24, 24, 0), (2, 16, 32, 32, 32, 1)])
def test_wgrad_dot_rows(shape, dtype):
    """dot_rows: same dw bits as the plain fold, and the rows add up to sum_{t,o} round(W) * dw per input channel -- which is
    sum_pixels X * dX (checked against the data gradient in float64)."""
    n, h, w, ci, co, upflag = shape
    rng = np.random.default_rng(zlib.crc32(repr(shape).encode()) % 1000)
    hs, ws_ = (h // 2, w // 2) if upflag else (h, w)
    x = rnd(rng.standard_normal((n, hs, ws_, ci)), dtype)
    wt = (rng.standard_normal((3, 3, ci, co)) * 0.2).astype(np.float32)          # fp32 master, NOT pre-rounded
    dy = rnd(rng.standard_normal((n, h, w, co)), dtype)
    xd, dyd, wm = up(x, dtype), up(dy, dtype), f32(wt)
    L = N.lib()
    wsb = L.rvip_conv3x3_wgrad_workspace(n, h, w, ci, co)
    wsd = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev())

    def run(dot):
        dw = torch.full((3, 3, ci, co), 7.0, dtype=torch.float32, device=dev())
        g = N.Wgrad3x3Desc()
        g.x0, g.c0, g.up0, g.x1, g.c1 = xd.data_ptr(), ci, upflag, None, 0
        g.dy, g.dw = dyd.data_ptr(), dw.data_ptr()
        g.n, g.h, g.w, g.cout, g.dtype = n, h, w, co, ndt(dtype)
        g.workspace, g.workspace_bytes = wsd.data_ptr(), wsb
        rows = None
        if dot:
            nr = L.rvip_conv3x3_wgrad_dot_rows(C.byref(g))
            assert nr == 9 * -(-co // 128)
            rows = torch.full((nr, ci), 7.0, dtype=torch.float64, device=dev())
            g.w_master, g.dot_rows, g.dot_rows_bytes = wm.data_ptr(), rows.data_ptr(), rows.numel() * 8
        N.call('rvip_conv3x3_wgrad', C.byref(g), stream())
        torch.cuda.synchronize()
        return down(dw), (rows.cpu().numpy() if dot else None)
    dw0, _ = run(False)
    dw1, rows = run(True)
    np.testing.assert_array_equal(dw0, dw1)
    wr = rnd(wt, dtype).astype(np.float64)
    want = (wr * dw1.astype(np.float64)).sum((0, 1, 3))
    got = rows.astype(np.float64).sum(0)
    scale = (np.abs(wr) * np.abs(dw1.astype(np.float64))).sum((0, 1, 3)).max()
    assert np.abs(got - want).max() <= 2e-6 * scale, (np.abs(got - want).max(), scale)
    # ... and that IS sum_pixels X * dX of the conv's input (float64 data gradient with the rounded kernel)
    xin = x.astype(np.float64)
    xfull = xin.repeat(2, 1).repeat(2, 2) if upflag else xin
    rdx, _, _ = O.conv2d_same_bwd(xfull, wr, dy.astype(np.float64))
    ident = (xfull * rdx).sum((0, 1, 2))
    assert np.abs(got - ident).max() <= (3e-3 if dtype != 'f32' else 3e-5) * scale


def _bn_chain(dtype, rate, n=2, h=24, w=40, c=32, co=64, seed=3, gamma_scale=1.0):
    """z -> BN -> [Dropout] -> y_d -> conv3x3 (W) -> ..., with a random gradient dz2 at the conv's output.  Returns everything the
    two routes of the BN backward need, on the device."""
    rng = np.random.default_rng(seed)
    rows = n * h * w
    z = rnd(np.maximum(rng.standard_normal((n, h, w, c)) * 1.5 + 0.3, 0), dtype)
    gamma = (gamma_scale * (1 + 0.3 * rng.standard_normal(c))).astype(np.float32)
    beta = (0.2 * rng.standard_normal(c)).astype(np.float32)
    wt = (rng.standard_normal((3, 3, c, co)) * 0.1).astype(np.float32)
    dz2 = rnd(rng.standard_normal((n, h, w, co)), dtype)
    L = N.lib()
    k = dict(n=n, h=h, w=w, c=c, co=co, rows=rows, rate=rate, lid=2, dtype=dtype)
    k['wsb'] = max(L.rvip_reduce_workspace(rows, 16 * c), L.rvip_conv3x3_wgrad_workspace(n, h, w, c, co))
    k['ws'] = torch.empty(k['wsb'] // 4 + 16, dtype=torch.float32, device=dev())
    k['zd'], k['gd'], k['bd'], k['wm'], k['dz2'] = up(z, dtype), f32(gamma), f32(beta), f32(wt), up(dz2, dtype)
    k['mm'], k['mv'] = f32(np.zeros(c)), f32(np.ones(c))
    k['mean'], k['invstd'], k['scale'], k['shift'] = (torch.empty(c, dtype=torch.float32, device=dev()) for _ in range(4))
    N.call('rvip_bn_train_stats', P(k['zd']), C.c_longlong(rows), c, ndt(dtype), P(k['gd']), P(k['bd']), P(k['mm']), P(k['mv']), 0.99, 1e-3, 1,
           P(k['mean']), P(k['invstd']), P(k['scale']), P(k['shift']), P(k['ws']), C.c_size_t(k['wsb']), stream())
    state = torch.zeros(8, dtype=torch.int32, device=dev())
    state[N.STATE_SEED], state[N.STATE_STEP] = 77, 3
    k['state'] = state
    y = torch.empty((n, h, w, c), dtype=tdt(dtype), device=dev())
    a = N.ApplyDesc()
    a.z, a.y, a.pooled = k['zd'].data_ptr(), y.data_ptr(), None
    a.scale, a.shift, a.act = k['scale'].data_ptr(), k['shift'].data_ptr(), 0
    a.drop_rate, a.mask, a.state, a.layer_id = rate, None, state.data_ptr(), k['lid']
    a.n, a.h, a.w, a.c, a.dtype = n, h, w, c, ndt(dtype)
    if rate:
        k['kbits'] = torch.full((-(-c // 32) * rows,), -1, dtype=torch.int32, device=dev())
        a.keep_bits = k['kbits'].data_ptr()
    N.call('rvip_bn_apply', C.byref(a), stream())
    if rate:                                    # the keep bits the forward pass leaves = the counter stream's mask
        keep = ds.keep_mask((n, h, w, c), rate, 77, 3, k['lid']).astype(bool)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(planes_to_flags(k['kbits'].cpu().numpy().view(np.uint32), (n, h, w, c)), keep)
    k['y'] = y
    k['wd'] = pack(wt, dtype)[1]
    return k


def _classic_desc(k, gy, rate, out):
    """the stage's rvip_bn_bwd_reduce descriptor: gradient gy, Dropout backward at `rate` inside the pass"""
    c = k['c']
    b = N.BnBwdDesc()
    b.dy, b.z, b.dz = gy.data_ptr(), k['zd'].data_ptr(), None
    b.gamma, b.mean, b.invstd = k['gd'].data_ptr(), k['mean'].data_ptr(), k['invstd'].data_ptr()
    b.dgamma, b.dbeta, b.coef = out['dgamma'].data_ptr(), out['dbeta'].data_ptr(), out['coef'].data_ptr()
    b.act, b.act_after_bn = N.ACT['relu'], 0
    b.drop_rate, b.mask, b.state, b.layer_id = rate, None, k['state'].data_ptr(), k['lid']
    b.rows, b.c, b.dtype = k['rows'], c, ndt(k['dtype'])
    b.workspace, b.workspace_bytes = k['ws'].data_ptr(), k['wsb']
    return b


def _outputs(c):
    return dict(dgamma=torch.full((c,), 7.0, dtype=torch.float32, device=dev()), dbeta=torch.full((c,), 7.0, dtype=torch.float32, device=dev()),
                coef=torch.full((3 * c,), 7.0, dtype=torch.float32, device=dev()))


def _classic(k, gy, rate):
    out = _outputs(k['c'])
    b = _classic_desc(k, gy, rate, out)
    N.call('rvip_bn_bwd_reduce', C.byref(b), stream())
    return {kk: down(v).astype(np.float64) for kk, v in out.items()}


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('rate', [0.0, 0.3])
def test_bn_bwd_coef_equals_the_reduction_pass(dtype, rate):
    k = _bn_chain(dtype, rate)
    n, h, w, c, co = k['n'], k['h'], k['w'], k['c'], k['co']
    L = N.lib()
    T = tdt(dtype)
    # classic: plain data gradient, then the reduction over (g, z) with the Dropout backward inside
    gy = torch.empty((n, h, w, c), dtype=T, device=dev())
    dplain = conv_desc(k['dz2'], co, 0, None, 0, k['wd'], None, gy, None, 0, n, h, w, c, 0, dtype)
    N.call('rvip_conv3x3_fwd', C.byref(dplain), stream())
    ref = _classic(k, gy, rate)
    # algebraic: weight gradient with dot rows, data gradient with column sums (+ Dropout backward), rvip_bn_bwd_coef
    dw = torch.empty((3, 3, c, co), dtype=torch.float32, device=dev())
    g = N.Wgrad3x3Desc()
    g.x0, g.c0, g.up0, g.x1, g.c1 = k['y'].data_ptr(), c, 0, None, 0
    g.dy, g.dw = k['dz2'].data_ptr(), dw.data_ptr()
    g.n, g.h, g.w, g.cout, g.dtype = n, h, w, co, ndt(dtype)
    g.workspace, g.workspace_bytes = k['ws'].data_ptr(), k['wsb']
    nd = L.rvip_conv3x3_wgrad_dot_rows(C.byref(g))
    drows = torch.empty((nd, c), dtype=torch.float64, device=dev())
    g.w_master, g.dot_rows, g.dot_rows_bytes = k['wm'].data_ptr(), drows.data_ptr(), drows.numel() * 8
    N.call('rvip_conv3x3_wgrad', C.byref(g), stream())
    g2 = torch.empty((n, h, w, c), dtype=T, device=dev())
    dst = conv_desc(k['dz2'], co, 0, None, 0, k['wd'], None, g2, None, 0, n, h, w, c, 0, dtype)
    if rate:
        dst.mask_bits, dst.mask_channels, dst.mask_scale = k['kbits'].data_ptr(), c, 1.0 / (1.0 - rate)
    nr = L.rvip_conv3x3_fwd_sums_rows(C.byref(dst))
    srows = torch.empty((nr, c), dtype=torch.float32, device=dev())
    N.call('rvip_conv3x3_fwd_sums', C.byref(dst), P(srows), C.c_size_t(srows.numel() * 4), stream())

    def coef(min_gamma):
        out = _outputs(c)
        out['flags'] = torch.full((-(-c // 32),), 7, dtype=torch.int32, device=dev())
        fb = _classic_desc(k, g2, 0.0, out)                   # what the engine hands over: the already-masked gradient, no dropout in the pass
        cd = N.BnCoefDesc()
        cd.t1[0].rows, cd.t1[0].nrows, cd.t1[0].stride, cd.t1[0].offset = srows.data_ptr(), nr, c, 0
        cd.t2[0].rows, cd.t2[0].nrows, cd.t2[0].stride, cd.t2[0].offset = drows.data_ptr(), nd, c, 0
        cd.gamma, cd.beta, cd.mean, cd.invstd = k['gd'].data_ptr(), k['bd'].data_ptr(), k['mean'].data_ptr(), k['invstd'].data_ptr()
        cd.dgamma, cd.dbeta, cd.coef, cd.flags = out['dgamma'].data_ptr(), out['dbeta'].data_ptr(), out['coef'].data_ptr(), out['flags'].data_ptr()
        cd.count, cd.c, cd.min_gamma, cd.max_beta_ratio = k['rows'], c, min_gamma, 64.0
        cd.fallback = C.pointer(fb)
        N.call('rvip_bn_bwd_coef', C.byref(cd), stream())
        return {kk: down(v).astype(np.float64) for kk, v in out.items()}
    # the two routes see the same g up to its storage rounding (the algebraic sum g*y uses the unrounded data gradient)
    tol = {'f32': 2e-5, 'bf16': 4e-3, 'f16': 6e-4}[dtype]

    def check(out):
        for name in ('dbeta', 'dgamma', 'coef'):
            got, want = out[name].reshape(-1, c), ref[name].reshape(-1, c)
            for i in range(got.shape[0]):
                assert np.abs(got[i] - want[i]).max() <= tol * np.abs(want[i]).max() + 1e-12, (name, i, np.abs(got[i] - want[i]).max(), np.abs(want[i]).max())
    out = coef(1.0 / 64)
    assert not out['flags'].any()
    check(out)
    # every channel declared ill-conditioned: the launch takes its exact route (sum g, sum g*xhat over all rows, inside the kernel)
    bad = coef(1e9)
    assert bad['flags'].all()
    check(bad)


def test_bn_bwd_coef_flags_small_gamma_and_refuses_bad_arguments():
    k = _bn_chain('f32', 0.0, c=64, gamma_scale=1.0)
    c = k['c']
    gam = np.abs(down(k['gd'])) + 0.5                               # every channel well-conditioned ...
    gam[40] = 1e-4                                                   # one channel of the second 32-channel block
    k['gd'].copy_(torch.from_numpy(gam))
    rows = torch.ones((4, c), dtype=torch.float32, device=dev())
    rows2 = torch.ones((4, c), dtype=torch.float64, device=dev())
    o = _outputs(c)
    out = [o['dgamma'], o['dbeta'], o['coef']]
    gy = torch.zeros((k['n'], k['h'], k['w'], c), dtype=torch.float32, device=dev())
    fb = _classic_desc(k, gy, 0.0, o)
    flags = torch.full((2,), 7, dtype=torch.int32, device=dev())
    cd = N.BnCoefDesc()
    cd.fallback = C.pointer(fb)
    cd.t1[0].rows, cd.t1[0].nrows, cd.t1[0].stride, cd.t1[0].offset = rows.data_ptr(), 4, c, 0
    cd.t2[0].rows, cd.t2[0].nrows, cd.t2[0].stride, cd.t2[0].offset = rows2.data_ptr(), 4, c, 0
    cd.gamma, cd.beta, cd.mean, cd.invstd = k['gd'].data_ptr(), k['bd'].data_ptr(), k['mean'].data_ptr(), k['invstd'].data_ptr()
    cd.dgamma, cd.dbeta, cd.coef, cd.flags = out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), flags.data_ptr()
    cd.count, cd.c, cd.min_gamma, cd.max_beta_ratio = k['rows'], c, 1.0 / 64, 64.0
    N.call('rvip_bn_bwd_coef', C.byref(cd), stream())
    assert down(flags).tolist() == [0, 1]                           # ... except one
    np.testing.assert_allclose(down(out[1])[:32], 4.0)               # dbeta = sum of the T1 rows in the algebraic block,
    np.testing.assert_allclose(down(out[1])[32:], 0.0)               # = sum of the (zero) gradient tensor in the block that went the exact way
    L = N.lib()
    cd.min_gamma = 0.0
    assert L.rvip_bn_bwd_coef(C.byref(cd), stream()) == -1
    cd.min_gamma, cd.t1[0].stride = 1.0 / 64, c - 1
    assert L.rvip_bn_bwd_coef(C.byref(cd), stream()) == -1
    cd.t1[0].stride, cd.fallback = c, None
    assert L.rvip_bn_bwd_coef(C.byref(cd), stream()) == -1


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', [(2, 24, 40, 32, 32, 0), (1, 16, 16, 16, 40, 0), (2, 40, 72, 64, 64, 0), (2, 24, 40, 16, 8, 2)])
def test_forward_sign_bits(shape, dtype):
    """rvip_conv3x3_fwd with sign_bits: the stored tensor is unchanged and bit (c & 31) of word [c / 32][pixel] says whether the stored
    value is > 0 -- what the ReLU backward of the stage needs (plain 9-tap launches here, incl. the zero-stuffed Conv2DTranspose read;
    the sub-pixel form is covered at the real up-conv shapes in test_gpu_real_shapes.py)."""
    n, h, w, ci, co, up0 = shape
    rng = np.random.default_rng(zlib.crc32(repr(shape).encode()) % 1000)
    hs, ws_ = (h // 2, w // 2) if up0 else (h, w)
    x = rnd(rng.standard_normal((n, hs, ws_, ci)), dtype)
    wt = rnd(rng.standard_normal((3, 3, ci, co)) * 0.2, dtype)
    b = rng.standard_normal(co).astype(np.float32)
    xd, bd = up(x, dtype), f32(b)
    wf, _ = pack(wt, dtype)
    ya = torch.empty((n, h, w, co), dtype=tdt(dtype), device=dev())
    da = conv_desc(xd, ci, up0, None, 0, wf, bd, ya, None, 0, n, h, w, co, N.ACT['relu'], dtype)
    N.call('rvip_conv3x3_fwd', C.byref(da), stream())
    yb = torch.empty_like(ya)
    db = conv_desc(xd, ci, up0, None, 0, wf, bd, yb, None, 0, n, h, w, co, N.ACT['relu'], dtype)
    sb = torch.full((-(-co // 32) * n * h * w,), -1, dtype=torch.int32, device=dev())
    db.sign_bits = sb.data_ptr()
    assert N.lib().rvip_conv3x3_sign_bits_ok(C.byref(db)) == 1
    N.call('rvip_conv3x3_fwd', C.byref(db), stream())
    assert torch.equal(ya, yb)
    flags = planes_to_flags(sb.cpu().numpy().view(np.uint32), (n, h, w, co))
    np.testing.assert_array_equal(flags, down(yb) > 0)
    assert 0.2 < flags.mean() < 0.8
