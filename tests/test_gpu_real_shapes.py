"""Op-level parity at the REAL layer shapes of the benchmarked networks: every distinct
(H, Cin[, skip], Cout, addressing mode) of BASELINE config 2 (256^2, F=32, depth 4) and the largest layers of config 4
(512^2, F=64, depth 5: 1024 -> 2048 and 2048 -> 2048 at 16^2, the up-sampled 2048 -> 1024 at 32^2, the (512+512) -> 512 concat at
64^2) -- K loops of 1..64 chunks, 1..32 output-channel tile columns, resident and rotating weights, split-K weight gradients with
the real slab sizes.  Forward (+ bias + ReLU, UpSampling2D / Concatenate as addressing modes, fused BN statistics), data gradient
(plain, split into the two concat halves, 2x2-summed for an up-sampled input) and weight gradient, in f32 / bf16 / f16, through the
C ABI against the float64 NumPy oracle.

The batch is cut to 1-4 images so the oracle stays affordable; the per-image work and every channel-dependent code path are the
full-size ones.  Inputs are drawn once per shape on a grid that bf16, f16 and f32 all represent exactly (8 significant bits, no
f16 subnormals), so ONE oracle evaluation serves the three storage types and only accumulation order + the output rounding differ."""
import ctypes as C
import functools

import numpy as np
import pytest
import torch

import cmr_landmark_detection_amd as rvip
from oracle import rvip_oracle as O
from test_gpu_ops import N, P, close, conv_desc, dev, down, f32, ndt, pack, stream, tdt, up

pytestmark = pytest.mark.gpu

# (id, n, h, c0, up0, c1, cout, bn)   h = OUTPUT size (square maps)
CFG2 = [
    ('enc0.conv2 32->32@256', 1, 256, 32, 0, 0, 32, True),
    ('enc1.conv1 32->64@128', 1, 128, 32, 0, 0, 64, True),
    ('enc1.conv2 64->64@128', 1, 128, 64, 0, 0, 64, True),
    ('enc2.conv1 64->128@64', 2, 64, 64, 0, 0, 128, True),
    ('enc2.conv2 128->128@64', 2, 64, 128, 0, 0, 128, True),
    ('enc3.conv1 128->256@32', 2, 32, 128, 0, 0, 256, True),
    ('enc3.conv2 256->256@32', 2, 32, 256, 0, 0, 256, True),
    ('mid.conv1 256->512@16', 4, 16, 256, 0, 0, 512, True),
    ('mid.conv2 512->512@16', 4, 16, 512, 0, 0, 512, True),
    ('dec0.up 512->256@32', 2, 32, 512, 1, 0, 256, False),
    ('dec0.cat 256+256->256@32', 2, 32, 256, 0, 256, 256, True),
    ('dec1.up 256->128@64', 2, 64, 256, 1, 0, 128, False),
    ('dec1.cat 128+128->128@64', 2, 64, 128, 0, 128, 128, True),
    ('dec2.up 128->64@128', 1, 128, 128, 1, 0, 64, False),
    ('dec2.cat 64+64->64@128', 1, 128, 64, 0, 64, 64, True),
    ('dec3.up 64->32@256', 1, 256, 64, 1, 0, 32, False),
    ('dec3.cat 32+32->32@256', 1, 256, 32, 0, 32, 32, True),
]
CFG4 = [
    ('cfg4 mid.conv1 1024->2048@16', 2, 16, 1024, 0, 0, 2048, True),
    ('cfg4 mid.conv2 2048->2048@16', 2, 16, 2048, 0, 0, 2048, True),
    ('cfg4 dec0.up 2048->1024@32', 1, 32, 2048, 1, 0, 1024, False),
    ('cfg4 dec2.cat 512+512->512@64', 1, 64, 512, 0, 512, 512, True),
    ('cfg4 enc0.conv2 64->64@512', 1, 512, 64, 0, 0, 64, True),
]
ALL = CFG2 + CFG4


def _grid(a, lim=6.0):
    """values every storage type holds exactly: bf16 rounding (8 significant bits), |x| in {0} U [2^-10, lim]"""
    t = torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()
    t = np.clip(t, -lim, lim)
    t[np.abs(t) < 2.0 ** -10] = 0.0
    assert np.array_equal(torch.from_numpy(t).to(torch.float16).to(torch.float32).numpy(), t)
    return t


@functools.lru_cache(maxsize=2)
def _case(idx):
    """inputs + float64 oracle results of one shape (cached: the three dtypes run back to back)"""
    name, n, h, c0, up0, c1, co, bn = ALL[idx]
    w_ = h
    ci = c0 + c1
    rng = np.random.default_rng(1000 + idx)
    hs = h // 2 if up0 else h
    x0 = _grid(rng.standard_normal((n, hs, hs, c0)))
    x1 = _grid(rng.standard_normal((n, h, w_, c1))) if c1 else None
    wt = _grid(rng.standard_normal((3, 3, ci, co)) * (0.7 / np.sqrt(9 * ci)) * 8) / 8       # he-like scale, still on the grid
    wt = _grid(wt)
    b = rng.standard_normal(co).astype(np.float32) * 0.1
    dy = _grid(rng.standard_normal((n, h, w_, co)))
    xin = O.upsample_nearest_fwd(x0) if up0 else x0
    if c1:
        xin = np.concatenate([xin, x1], -1)
    x64, w64, dy64 = xin.astype(np.float64), wt.astype(np.float64), dy.astype(np.float64)
    fwd = O.act_fwd(O.conv2d_same_fwd(x64, w64, b.astype(np.float64)), 'relu')
    dx, dw, _ = O.conv2d_same_bwd(x64, w64, dy64)
    return dict(x0=x0, x1=x1, wt=wt, b=b, dy=dy, fwd=fwd, dx=dx, dw=dw)


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('idx', range(len(ALL)), ids=[s[0].replace(' ', '_') for s in ALL])
def test_real_layer_shape_fwd_stats_dgrad_wgrad(idx, dtype):
    name, n, h, c0, up0, c1, co, bn = ALL[idx]
    w_ = h
    ci = c0 + c1
    k = _case(idx)
    L = N.lib()
    x0d = up(k['x0'], dtype)
    x1d = up(k['x1'], dtype) if c1 else None
    dyd, bd = up(k['dy'], dtype), f32(k['b'])
    wf, wd = pack(k['wt'], dtype)
    # ---- forward: bias + ReLU epilogue, UpSampling2D / Concatenate read through the addressing modes
    y = torch.empty((n, h, w_, co), dtype=tdt(dtype), device=dev())
    d = conv_desc(x0d, c0, up0, x1d, c1, wf, bd, y, None, 0, n, h, w_, co, N.ACT['relu'], dtype)
    N.call('rvip_conv3x3_fwd', C.byref(d), stream())
    close(down(y), k['fwd'], dtype, name + ' fwd')
    # ---- the same launch with the BatchNormalization statistics of the stored tensor folded into the epilogue
    if bn:
        rows = L.rvip_conv3x3_fwd_stats_rows(C.byref(d))
        assert rows > 0, 'the real shapes must take the fused-statistics path'
        ws = torch.full((rows * 2 * co + 16,), 7.0, dtype=torch.float32, device=dev())
        y2 = torch.empty_like(y)
        d.y = y2.data_ptr()
        N.call('rvip_conv3x3_fwd_stats', C.byref(d), P(ws), C.c_size_t(rows * 2 * co * 4), stream())
        assert torch.equal(y, y2)                                         # same tensor, bit for bit
        gamma, beta = np.linspace(0.5, 1.5, co).astype(np.float32), np.linspace(-0.2, 0.2, co).astype(np.float32)
        gd, btd, mm, mv = f32(gamma), f32(beta), f32(np.zeros(co)), f32(np.ones(co))
        mean, invstd, scale, shift = (torch.empty(co, dtype=torch.float32, device=dev()) for _ in range(4))
        N.call('rvip_bn_stats_finalize', P(ws), rows, C.c_longlong(n * h * w_), co, P(gd), P(btd), P(mm), P(mv), 0.99, 1e-3, 1,
               P(mean), P(invstd), P(scale), P(shift), stream())
        yq = down(y).astype(np.float64)
        _, cache = O.bn_train_fwd(yq, gamma.astype(np.float64), beta.astype(np.float64))
        np.testing.assert_allclose(down(mean), cache[2], atol=3e-6 * max(1.0, float(np.abs(cache[2]).max())))
        np.testing.assert_allclose(down(invstd), cache[1], rtol=2e-5)
        np.testing.assert_allclose(down(mv), O.bn_moving_update(np.zeros(co), np.ones(co), cache[2], cache[3], n * h * w_)[1], rtol=2e-5)
    # ---- data gradient: the same kernel on dy with the rotated operand; concat -> two outputs; up-sampled input -> 2x2 sums
    if c1:
        g0 = torch.empty((n, h, w_, c0), dtype=tdt(dtype), device=dev())
        g1 = torch.empty((n, h, w_, c1), dtype=tdt(dtype), device=dev())
        d2 = conv_desc(dyd, co, 0, None, 0, wd, None, g0, g1, c0, n, h, w_, ci, 0, dtype)
        N.call('rvip_conv3x3_fwd', C.byref(d2), stream())
        close(down(g0), k['dx'][..., :c0], dtype, name + ' dgrad (up half)')
        close(down(g1), k['dx'][..., c0:], dtype, name + ' dgrad (skip half)')
    elif up0:
        glo = torch.empty((n, h // 2, w_ // 2, c0), dtype=tdt(dtype), device=dev())
        d2 = conv_desc(dyd, co, 0, None, 0, wd, None, glo, None, 0, n, h, w_, ci, 0, dtype)
        d2.down2 = 1
        N.call('rvip_conv3x3_fwd', C.byref(d2), stream())
        close(down(glo), O.upsample_nearest_bwd(k['dx']), dtype, name + ' dgrad (2x2-summed)')
    else:
        dx = torch.empty((n, h, w_, ci), dtype=tdt(dtype), device=dev())
        d2 = conv_desc(dyd, co, 0, None, 0, wd, None, dx, None, 0, n, h, w_, ci, 0, dtype)
        N.call('rvip_conv3x3_fwd', C.byref(d2), stream())
        close(down(dx), k['dx'], dtype, name + ' dgrad')
    # ---- weight gradient: fp32 HWIO, deterministic split-K, immediate and deferred (slabs + batched fold) forms agree
    wsb = L.rvip_conv3x3_wgrad_workspace(n, h, w_, ci, co)
    ws = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev())
    dw = torch.full((3, 3, ci, co), 7.0, dtype=torch.float32, device=dev())
    g = N.Wgrad3x3Desc()
    g.x0, g.c0, g.up0 = x0d.data_ptr(), c0, up0
    g.x1, g.c1 = (x1d.data_ptr(), c1) if c1 else (None, 0)
    g.dy, g.dw = dyd.data_ptr(), dw.data_ptr()
    g.n, g.h, g.w, g.cout, g.dtype = n, h, w_, co, ndt(dtype)
    g.workspace, g.workspace_bytes = ws.data_ptr(), wsb
    N.call('rvip_conv3x3_wgrad', C.byref(g), stream())
    got = down(dw)
    scale = float(np.abs(k['dw']).max())
    # products are exact in fp32 (inputs carry 8 significant bits); only the fp32 accumulation order over N*H*W pixels differs
    assert np.abs(got - k['dw']).max() <= 2e-5 * scale * max(1.0, np.sqrt(n * h * w_ / 4096.0)), (name, np.abs(got - k['dw']).max() / scale)
    ns = L.rvip_conv3x3_wgrad_splits(C.byref(g))
    assert ns >= 1
    slabs = torch.empty(ns * 9 * ci * co, dtype=torch.float32, device=dev())
    dw2 = torch.full((3, 3, ci, co), 3.0, dtype=torch.float32, device=dev())
    g.workspace, g.workspace_bytes, g.defer_fold = slabs.data_ptr(), slabs.numel() * 4, 1
    N.call('rvip_conv3x3_wgrad', C.byref(g), stream())
    tab = (N.FoldEntry * 1)()
    tab[0].src, tab[0].dst, tab[0].nrows, tab[0].width = slabs.data_ptr(), dw2.data_ptr(), ns, 9 * ci * co
    tabd = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev())
    N.call('rvip_fold_rows_batch', P(tabd), 1, C.c_longlong(9 * ci * co), 1, stream())
    torch.cuda.synchronize()
    # (the batched fold sums the slabs in double, the immediate one in float: equal to fp32 rounding, not bitwise)
    assert float((dw - dw2).abs().max()) <= 2e-6 * scale, name + ': deferred fold differs from the immediate one'


# ------------------------------------------------------------------------------------------------------------------------------
# What the PRODUCT launches by default at these shapes (round 3): data gradients with the column sums of their result in the
# epilogue (split / 2x2-summed / Dropout backward), weight gradients whose fold writes the sum W*dW rows, the sub-pixel form of
# the up-convs, the Conv2DTranspose decoder, the pooled stages with their window argmax, and config 5's Conv3D layers.
# ------------------------------------------------------------------------------------------------------------------------------
ds = __import__('importlib').import_module('cmr-landmark-detection_amd.dropout_stream')
from test_gpu_bnbwd_algebraic import bit_planes, planes_to_flags      # noqa: E402


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('idx', range(len(CFG2)), ids=[s[0].replace(' ', '_') for s in CFG2])
def test_real_layer_shape_column_sums_and_dot_rows(idx, dtype):
    """rvip_conv3x3_fwd_sums in the mode the engine uses for this layer's data gradient and rvip_conv3x3_wgrad with dot_rows: the
    stored tensors against the float64 oracle, the column sums against the oracle's, sum W*dW against the oracle's sum X*dX."""
    name, n, h, c0, up0, c1, co, bn = ALL[idx]
    w_ = h
    ci = c0 + c1
    k = _case(idx)
    L = N.lib()
    T = tdt(dtype)
    dyd = up(k['dy'], dtype)
    _, wd = pack(k['wt'], dtype)
    state = torch.zeros(8, dtype=torch.int32, device=dev())
    state[N.STATE_SEED], state[N.STATE_STEP] = 4242, 11
    rate, lid = 0.3, 2
    modes = ['split'] if c1 else (['down2'] if up0 else ['plain', 'dropout'])
    for mode in modes:
        if mode == 'split':                    # as the engine launches it: the up-conv half gated by that stage's sign bits (its ReLU backward)
            g0, g1 = torch.empty((n, h, w_, c0), dtype=T, device=dev()), torch.empty((n, h, w_, c1), dtype=T, device=dev())
            d2 = conv_desc(dyd, co, 0, None, 0, wd, None, g0, g1, c0, n, h, w_, ci, 0, dtype)
            gate = np.random.default_rng(77 + idx).random((n, h, w_, c0)) < 0.6
            gbits = torch.from_numpy(bit_planes(gate).view(np.int32)).to(dev())
            d2.mask_bits, d2.mask_channels, d2.mask_scale = gbits.data_ptr(), c0, 1.0
            want = k['dx'].copy()
            want[..., :c0] *= gate
        elif mode == 'down2':
            g0, g1 = torch.empty((n, h // 2, w_ // 2, c0), dtype=T, device=dev()), None
            d2 = conv_desc(dyd, co, 0, None, 0, wd, None, g0, None, 0, n, h, w_, ci, 0, dtype)
            d2.down2 = 1
            want = O.upsample_nearest_bwd(k['dx'])
        else:
            g0, g1 = torch.empty((n, h, w_, ci), dtype=T, device=dev()), None
            d2 = conv_desc(dyd, co, 0, None, 0, wd, None, g0, None, 0, n, h, w_, ci, 0, dtype)
            want = k['dx']
            if mode == 'dropout':
                keepb = ds.keep_mask((n, h, w_, ci), rate, 4242, 11, lid).astype(bool)
                kbits = torch.from_numpy(bit_planes(keepb).view(np.int32)).to(dev())
                d2.mask_bits, d2.mask_channels, d2.mask_scale = kbits.data_ptr(), ci, 1.0 / (1.0 - rate)
                want = want * keepb / np.float32(1 - rate)
        rows = L.rvip_conv3x3_fwd_sums_rows(C.byref(d2))
        assert rows > 0, 'the real shapes must take the fused column-sum path'
        buf = torch.full((rows, ci), 7.0, dtype=torch.float32, device=dev())
        N.call('rvip_conv3x3_fwd_sums', C.byref(d2), P(buf), C.c_size_t(buf.numel() * 4), stream())
        stored = down(g0) if g1 is None else np.concatenate([down(g0), down(g1)], -1)
        close(stored, want, dtype, '%s dgrad (%s)' % (name, mode))
        got = down(buf).astype(np.float64).sum(0)
        ref = want.reshape(-1, ci).sum(0)
        lo = 0
        tol = 2e-5 * np.abs(want).reshape(-1, ci).sum(0).max()
        assert np.abs(got[lo:] - ref[lo:]).max() <= tol, (name, mode, np.abs(got[lo:] - ref[lo:]).max(), tol)
    # weight gradient with the dot rows
    x0d = up(k['x0'], dtype)
    x1d = up(k['x1'], dtype) if c1 else None
    wm = f32(k['wt'])
    wsb = L.rvip_conv3x3_wgrad_workspace(n, h, w_, ci, co)
    ws = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev())

    def run(dot):
        dw = torch.full((3, 3, ci, co), 7.0, dtype=torch.float32, device=dev())
        g = N.Wgrad3x3Desc()
        g.x0, g.c0, g.up0 = x0d.data_ptr(), c0, up0
        g.x1, g.c1 = (x1d.data_ptr(), c1) if c1 else (None, 0)
        g.dy, g.dw = dyd.data_ptr(), dw.data_ptr()
        g.n, g.h, g.w, g.cout, g.dtype = n, h, w_, co, ndt(dtype)
        g.workspace, g.workspace_bytes = ws.data_ptr(), wsb
        rows_t = None
        if dot:
            nd = L.rvip_conv3x3_wgrad_dot_rows(C.byref(g))
            rows_t = torch.full((nd, ci), 7.0, dtype=torch.float64, device=dev())
            g.w_master, g.dot_rows, g.dot_rows_bytes = wm.data_ptr(), rows_t.data_ptr(), rows_t.numel() * 8
        N.call('rvip_conv3x3_wgrad', C.byref(g), stream())
        torch.cuda.synchronize()
        return dw, rows_t
    dw0, _ = run(False)
    dw1, rows_t = run(True)
    assert torch.equal(dw0, dw1)
    xin = O.upsample_nearest_fwd(k['x0']) if up0 else k['x0']
    if c1:
        xin = np.concatenate([xin, k['x1']], -1)
    ident = (xin.astype(np.float64) * k['dx']).sum((0, 1, 2))               # sum_pixels X * dX per input channel (float64)
    torch.cuda.synchronize()
    t2 = rows_t.cpu().numpy().sum(0)
    scale = (np.abs(k['wt'].astype(np.float64)) * np.abs(k['dw'])).sum((0, 1, 3)).max()
    assert np.abs(t2 - ident).max() <= 2e-5 * scale * max(1.0, np.sqrt(n * h * w_ / 4096.0)), (name, np.abs(t2 - ident).max() / scale)


UPS = [i for i, s in enumerate(ALL) if s[4] == 1 and i < len(CFG2)]


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('idx', UPS, ids=[ALL[i][0].replace(' ', '_') for i in UPS])
def test_real_shape_upconv_subpixel_form(idx, dtype):
    """The up-convs as the engine launches them in the forward pass: four 2x2-tap phase convolutions on the low-resolution input
    (rvip_pack_subpixel_weights + subpix = 1), 512 -> 256 at 32^2 ... 64 -> 32 at 256^2, against the float64 oracle of
    UpSampling2D -> Conv2D (KerasLayers.py:756-758)."""
    name, n, h, c0, up0, c1, co, bn = ALL[idx]
    k = _case(idx)
    wm, bd, lod = f32(k['wt']), f32(k['b']), up(k['x0'], dtype)
    wph = torch.empty(16 * c0 * co, dtype=tdt(dtype), device=dev())
    N.call('rvip_pack_subpixel_weights', P(wm), c0, co, ndt(dtype), P(wph), stream())
    y = torch.full((n, h, h, co), 9.0, dtype=tdt(dtype), device=dev())
    d = conv_desc(lod, c0, 1, None, 0, wph, bd, y, None, 0, n, h, h, co, N.ACT['relu'], dtype)
    d.subpix = 1
    sbits = torch.full((-(-co // 32) * n * h * h,), -1, dtype=torch.int32, device=dev())      # the training launch also leaves the sign bits
    d.sign_bits = sbits.data_ptr()
    assert N.lib().rvip_conv3x3_sign_bits_ok(C.byref(d)) == 1
    N.call('rvip_conv3x3_fwd', C.byref(d), stream())
    got = down(y)
    np.testing.assert_array_equal(planes_to_flags(sbits.cpu().numpy().view(np.uint32), (n, h, h, co)), got > 0)
    scale = float(np.abs(k['fwd']).max())
    tol = (2.0 ** -6 if dtype != 'f32' else 2e-5) * scale                   # 16-bit: + one rounding of the summed taps
    assert np.abs(got - k['fwd']).max() <= tol, (name, np.abs(got - k['fwd']).max() / scale)


UPS_ALL = [i for i, s in enumerate(ALL) if s[4] == 1]          # + config 4's 2048 -> 1024 at 32^2: 128 K chunks, 32 channel columns


@pytest.mark.parametrize('dtype', ['bf16', 'f16'])
@pytest.mark.parametrize('idx', UPS_ALL, ids=[ALL[i][0].replace(' ', '_') for i in UPS_ALL])
def test_real_shape_upconv_subpixel_backward(idx, dtype):
    """The up-convs' backward as the engine launches it for the 16-bit types: the data gradient in its sub-pixel form (subpix = 2,
    rvip_pack_subpixel_dgrad_weights) WITH the column sums of its result, and the weight gradient whose 64 x 64-block layers take the
    sub-pixel form inside rvip_conv3x3_wgrad (the split count says which form ran) with the sum W*dW rows -- against the float64 oracle
    of UpSampling2D -> Conv2D (KerasLayers.py:756-758 autodiff)."""
    name, n, h, c0, up0, c1, co, bn = ALL[idx]
    k = _case(idx)
    L = N.lib()
    wm, dyd, lod = f32(k['wt']), up(k['dy'], dtype), up(k['x0'], dtype)
    # ---- data gradient
    wph = torch.empty(16 * c0 * co, dtype=tdt(dtype), device=dev())
    N.call('rvip_pack_subpixel_dgrad_weights', P(wm), c0, co, ndt(dtype), P(wph), stream())
    glo = torch.full((n, h // 2, h // 2, c0), 9.0, dtype=tdt(dtype), device=dev())
    d = conv_desc(dyd, co, 0, None, 0, wph, None, glo, None, 0, n, h, h, c0, 0, dtype)
    d.subpix = 2
    nr = L.rvip_conv3x3_fwd_sums_rows(C.byref(d))
    assert nr > 0, 'the real up-conv shapes must take the sub-pixel data gradient'
    rows = torch.full((nr, c0), 7.0, dtype=torch.float32, device=dev())
    N.call('rvip_conv3x3_fwd_sums', C.byref(d), P(rows), C.c_size_t(rows.numel() * 4), stream())
    ref = O.upsample_nearest_bwd(k['dx'])
    got = down(glo).astype(np.float64)
    scale = float(np.abs(ref).max())
    assert np.abs(got - ref).max() <= 2.0 ** -6 * scale, (name, np.abs(got - ref).max() / scale)      # + one rounding of the summed taps
    sums, want = down(rows).astype(np.float64).sum(0), got.sum((0, 1, 2))
    tol = {'bf16': 2.0 ** -8, 'f16': 2.0 ** -11}[dtype] * np.abs(got).sum((0, 1, 2)).max() + 1e-6
    assert np.abs(sums - want).max() <= tol, (name, np.abs(sums - want).max(), tol)
    # ---- weight gradient (+ dot rows)
    wsb = L.rvip_conv3x3_wgrad_workspace(n, h, h, c0, co)
    ws = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev())
    dw = torch.full((3, 3, c0, co), 7.0, dtype=torch.float32, device=dev())
    g = N.Wgrad3x3Desc()
    g.x0, g.c0, g.up0, g.x1, g.c1 = lod.data_ptr(), c0, 1, None, 0
    g.dy, g.dw = dyd.data_ptr(), dw.data_ptr()
    g.n, g.h, g.w, g.cout, g.dtype = n, h, h, co, ndt(dtype)
    g.workspace, g.workspace_bytes = ws.data_ptr(), wsb
    nd = L.rvip_conv3x3_wgrad_dot_rows(C.byref(g))
    drows = torch.full((nd, c0), 7.0, dtype=torch.float64, device=dev())
    g.w_master, g.dot_rows, g.dot_rows_bytes = wm.data_ptr(), drows.data_ptr(), drows.numel() * 8
    ns = L.rvip_conv3x3_wgrad_splits(C.byref(g))
    form = L.rvip_conv3x3_wgrad_form(C.byref(g))
    if c0 >= 64 and co >= 64:
        assert form == 1 and ns % 4 == 0 and ns >= 4, (name, form, ns)      # four phases x pixel splits
    elif c0 == 64 and co == 32:
        assert form == 2 and ns % 2 == 0 and ns >= 2, (name, form, ns)      # dec3.up of config 2: both column phases per workgroup (PB = 1)
    else:
        assert form == 0 and ns >= 1, (name, form, ns)
    N.call('rvip_conv3x3_wgrad', C.byref(g), stream())
    gotw = down(dw).astype(np.float64)
    sw = float(np.abs(k['dw']).max())
    assert np.abs(gotw - k['dw']).max() <= 2e-5 * sw * max(1.0, np.sqrt(n * h * h / 4096.0)), (name, np.abs(gotw - k['dw']).max() / sw)
    wr = k['wt'].astype(np.float64)                                               # on the grid: exact in every storage type
    t2 = (wr * gotw).sum((0, 1, 3))
    assert np.abs(drows.cpu().numpy().sum(0) - t2).max() <= 2e-6 * (np.abs(wr) * np.abs(gotw)).sum((0, 1, 3)).max()
    # ---- the same rows against the PHASE kernels (round 4, four-phase form only): T2' = sum_pixels X * dX' for the dX' the sub-pixel data
    # gradient above really wrote -- where <Wr, dW> describes the nine-tap gradient, whose phase kernels are summed taps rounded once
    g.w_phase = wph.data_ptr()
    if form != 1:
        assert L.rvip_conv3x3_wgrad(C.byref(g), stream()) == -2                   # only the four-phase slabs hold phase-resolved gradients
        return
    drows2 = torch.full((nd, c0), 7.0, dtype=torch.float64, device=dev())
    dw_b = torch.full((3, 3, c0, co), 7.0, dtype=torch.float32, device=dev())
    g.dw, g.dot_rows = dw_b.data_ptr(), drows2.data_ptr()
    N.call('rvip_conv3x3_wgrad', C.byref(g), stream())
    torch.cuda.synchronize()
    assert torch.equal(dw_b, dw)                                                  # dw itself is untouched by the choice of rows
    t2p = drows2.cpu().numpy().sum(0)
    # float64 restatement with the device's own phase kernels and slabs' content: the phase-resolved weight gradients from the oracle's
    # per-phase correlation, dotted with the phase kernels as the device rounded them
    wph_h = down(wph).astype(np.float64).reshape(4, 4, c0, co)                    # [phase][2u+v (data-gradient convention)][ci][co]
    x64, dy64 = k['x0'].astype(np.float64), k['dy'].astype(np.float64)
    xp = np.pad(x64, ((0, 0), (1, 1), (1, 1), (0, 0)))
    want = np.zeros(c0)
    for a_ in range(2):
        for b_ in range(2):
            dyp = dy64[:, a_::2, b_::2, :]                                        # phase image of the gradient, on the low-resolution grid
            for uf in range(2):
                for vf in range(2):
                    # forward convention: phase 0 reads low-res offsets (-1, 0), phase 1 reads (0, +1), per axis
                    oy, ox = (uf - 1 if a_ == 0 else uf), (vf - 1 if b_ == 0 else vf)
                    xs = xp[:, 1 + oy:1 + oy + h // 2, 1 + ox:1 + ox + h // 2, :]
                    dwp = np.einsum('nhwi,nhwo->io', xs, dyp)
                    want += (wph_h[2 * a_ + b_, 2 * (1 - uf) + (1 - vf)] * dwp).sum(1)
    scale2 = np.abs(want).max()
    assert np.abs(t2p - want).max() <= 3e-5 * max(scale2, (np.abs(wr) * np.abs(gotw)).sum((0, 1, 3)).max()), (name, np.abs(t2p - want).max(), scale2)
    # ... which IS sum_pixels X * dX' of the gradient the sub-pixel launch wrote (up to ITS storage rounding, zero-mean per element)
    ident = (x64 * got).sum((0, 1, 2))
    bound = {'bf16': 2.0 ** -8, 'f16': 2.0 ** -11}[dtype] * np.sqrt(((x64 * got) ** 2).sum((0, 1, 2))).max() * 4 + 1e-9
    assert np.abs(t2p - ident).max() <= bound + 3e-5 * scale2, (name, np.abs(t2p - ident).max(), bound)
    print('%s %s: |T2 (taps) - T2\' (phase kernels)| / |T2| = %.2e' % (name, dtype, np.abs(t2 - t2p).max() / np.abs(t2).max()))


@functools.lru_cache(maxsize=1)
def _tcase(idx):
    """Conv2DTranspose(3, strides=2, 'same') at an up-conv's shape (USE_UPSAMPLE=False, KerasLayers.py:761-765)"""
    name, n, h, c0, up0, c1, co, bn = ALL[idx]
    rng = np.random.default_rng(2000 + idx)
    x = _grid(rng.standard_normal((n, h // 2, h // 2, c0)))
    wt = _grid(_grid(rng.standard_normal((3, 3, co, c0)) * (0.7 / np.sqrt(9 * c0 / 4.0)) * 8) / 8)        # Keras HWOI
    b = rng.standard_normal(co).astype(np.float32) * 0.1
    dy = _grid(rng.standard_normal((n, h, h, co)))
    fwd = O.conv2d_transpose_same_fwd(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    dx, dw, _ = O.conv2d_transpose_same_bwd(x.astype(np.float64), wt.astype(np.float64), dy.astype(np.float64))
    return dict(x=x, wt=wt, b=b, dy=dy, fwd=fwd, dx=dx, dw=dw)


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('idx', UPS, ids=[ALL[i][0].replace(' ', '_').replace('.up', '.transpose') for i in UPS])
def test_real_shape_conv2d_transpose(idx, dtype):
    """The zero-stuffed read (up0 = 2) at the four decoder shapes: forward, input gradient (full-resolution data gradient +
    rvip_subsample_odd) and weight gradient against the oracle's transpose-conv."""
    name, n, h, c0, up0, c1, co, bn = ALL[idx]
    k = _tcase(idx)
    hl = h // 2
    weq = np.ascontiguousarray(k['wt'][::-1, ::-1].transpose(0, 1, 3, 2))    # [3][3][ci][co]: the equivalent forward kernel
    wf, wd = pack(weq, dtype)
    xd, bd, dyd = up(k['x'], dtype), f32(k['b']), up(k['dy'], dtype)
    y = torch.empty((n, h, h, co), dtype=tdt(dtype), device=dev())
    d = conv_desc(xd, c0, 2, None, 0, wf, bd, y, None, 0, n, h, h, co, 0, dtype)
    N.call('rvip_conv3x3_fwd', C.byref(d), stream())
    close(down(y), k['fwd'], dtype, name + ' transpose fwd')
    gfull = torch.empty((n, h, h, c0), dtype=tdt(dtype), device=dev())
    d2 = conv_desc(dyd, co, 0, None, 0, wd, None, gfull, None, 0, n, h, h, c0, 0, dtype)
    N.call('rvip_conv3x3_fwd', C.byref(d2), stream())
    dx = torch.empty((n, hl, hl, c0), dtype=tdt(dtype), device=dev())
    N.call('rvip_subsample_odd', P(gfull), P(dx), n, hl, hl, c0, ndt(dtype), stream())
    close(down(dx), k['dx'], dtype, name + ' transpose dgrad')
    L = N.lib()
    wsb = L.rvip_conv3x3_wgrad_workspace(n, h, h, c0, co)
    ws = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev())
    dw = torch.empty((3, 3, c0, co), dtype=torch.float32, device=dev())
    g = N.Wgrad3x3Desc()
    g.x0, g.c0, g.up0, g.x1, g.c1 = xd.data_ptr(), c0, 2, None, 0
    g.dy, g.dw = dyd.data_ptr(), dw.data_ptr()
    g.n, g.h, g.w, g.cout, g.dtype = n, h, h, co, ndt(dtype)
    g.workspace, g.workspace_bytes = ws.data_ptr(), wsb
    N.call('rvip_conv3x3_wgrad', C.byref(g), stream())
    dw_hwoi = down(dw).transpose(0, 1, 3, 2)[::-1, ::-1]
    scale = float(np.abs(k['dw']).max())
    assert np.abs(dw_hwoi - k['dw']).max() <= 2e-5 * scale * max(1.0, np.sqrt(n * h * h / 4096.0)), (name, np.abs(dw_hwoi - k['dw']).max() / scale)


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('hc', [(256, 32), (128, 64), (64, 128), (32, 256)], ids=['256x32', '128x64', '64x128', '32x256'])
def test_real_shape_pooled_stage_against_the_oracle(hc, dtype):
    """enc*.conv2's BN + MaxPooling2D stage at its real shapes: rvip_bn_apply(pooled, argmax) and the two BN-backward passes that
    take (pooled gradient, window argmax, skip gradient), each against the float64 oracle (BatchNormalization + MaxPooling2D and
    their autodiff, KerasLayers.py:684-691, 714-721) -- not only against the separate-kernel route."""
    h, c = hc
    n = 1
    L = N.lib()
    if not L.rvip_bn_apply_argmax_ok(c, ndt(dtype)):
        pytest.skip('the column-split pooled kernel covers <= 32 channel vectors (engine: separate rvip_maxpool2x2_bwd)')
    rows = n * h * h
    rng = np.random.default_rng(500 + h)
    z = _grid(np.maximum(rng.standard_normal((n, h, h, c)) * 1.5 - 0.4, 0))
    gamma = (1 + 0.3 * rng.standard_normal(c)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(c)).astype(np.float32)
    wsb = L.rvip_reduce_workspace(rows, 16 * c)
    ws = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev())
    zd, gd, bd = up(z, dtype), f32(gamma), f32(beta)
    mm, mv = f32(np.zeros(c)), f32(np.ones(c))
    mean, invstd, scale, shift = (torch.empty(c, dtype=torch.float32, device=dev()) for _ in range(4))
    N.call('rvip_bn_train_stats', P(zd), C.c_longlong(rows), c, ndt(dtype), P(gd), P(bd), P(mm), P(mv), 0.99, 1e-3, 1,
           P(mean), P(invstd), P(scale), P(shift), P(ws), C.c_size_t(wsb), stream())
    oh = h // 2
    ve = 4 if dtype == 'f32' else 8
    y = torch.empty((n, h, h, c), dtype=tdt(dtype), device=dev())
    pooled = torch.empty((n, oh, oh, c), dtype=tdt(dtype), device=dev())
    arg = torch.full((n * oh * oh * (c // ve),), -1, dtype=torch.int16, device=dev())
    state = torch.zeros(8, dtype=torch.int32, device=dev())
    a = N.ApplyDesc()
    a.z, a.y, a.pooled, a.argmax = zd.data_ptr(), y.data_ptr(), pooled.data_ptr(), arg.data_ptr()
    a.scale, a.shift, a.act = scale.data_ptr(), shift.data_ptr(), 0
    a.drop_rate, a.mask, a.state, a.layer_id = 0.0, None, state.data_ptr(), 0
    a.n, a.h, a.w, a.c, a.dtype = n, h, h, c, ndt(dtype)
    N.call('rvip_bn_apply', C.byref(a), stream())
    z64 = z.astype(np.float64)
    ybn, cache = O.bn_train_fwd(z64, gamma.astype(np.float64), beta.astype(np.float64))
    close(down(y), ybn, dtype, 'BN output')
    yq = down(y).astype(np.float64)
    pref, idx = O.maxpool2x2_fwd(yq)
    np.testing.assert_array_equal(down(pooled), pref)                       # the pool of what was stored: exact
    words = arg.cpu().numpy().astype(np.uint16).reshape(n, oh, oh, c // ve)
    np.testing.assert_array_equal(np.stack([(words >> (2 * e)) & 3 for e in range(ve)], -1).reshape(n, oh, oh, c), idx)
    # backward
    dp = _grid(rng.standard_normal((n, oh, oh, c)))
    addg = _grid(rng.standard_normal((n, h, h, c)))
    dpd, addd = up(dp, dtype), up(addg, dtype)
    dz = torch.full((n, h, h, c), 7.0, dtype=tdt(dtype), device=dev())
    dgamma, dbeta, dbias = (torch.empty(c, dtype=torch.float32, device=dev()) for _ in range(3))
    coef = torch.empty(3 * c, dtype=torch.float32, device=dev())
    b = N.BnBwdDesc()
    b.z, b.dz, b.dy = zd.data_ptr(), dz.data_ptr(), addd.data_ptr()
    b.dpooled, b.argmax, b.h, b.w = dpd.data_ptr(), arg.data_ptr(), h, h
    b.gamma, b.mean, b.invstd = gd.data_ptr(), mean.data_ptr(), invstd.data_ptr()
    b.scale, b.shift = scale.data_ptr(), shift.data_ptr()
    b.dgamma, b.dbeta, b.dbias, b.coef = dgamma.data_ptr(), dbeta.data_ptr(), dbias.data_ptr(), coef.data_ptr()
    b.act, b.act_after_bn = N.ACT['relu'], 0
    b.drop_rate, b.mask, b.state, b.layer_id = 0.0, None, state.data_ptr(), 0
    b.rows, b.c, b.dtype = rows, c, ndt(dtype)
    b.workspace, b.workspace_bytes = ws.data_ptr(), wsb
    N.call('rvip_bn_bwd_reduce', C.byref(b), stream())
    N.call('rvip_bn_bwd_apply', C.byref(b), stream())
    g = O.maxpool2x2_bwd(dp.astype(np.float64), idx, yq.shape) + addg.astype(np.float64)
    g = torch.from_numpy(g.astype(np.float32)).to(tdt(dtype)).to(torch.float64).numpy()       # the stage rounds the routed sum to its storage type
    dxbn, rdg, rdb = O.bn_train_bwd(g, gamma.astype(np.float64), cache)
    rdz = dxbn * (z64 > 0)
    close(down(dz), rdz, dtype, 'dz')
    np.testing.assert_allclose(down(dbeta), rdb, atol=2e-5 * np.abs(g).reshape(-1, c).sum(0).max())
    np.testing.assert_allclose(down(dgamma), rdg, atol=2e-5 * np.abs(g).reshape(-1, c).sum(0).max() * 4)
    qdz = down(dz).astype(np.float64)
    np.testing.assert_allclose(down(dbias), qdz.reshape(-1, c).sum(0), atol=2e-5 * np.abs(qdz).reshape(-1, c).sum(0).max())


# (id, volumes, T, h, c0, up0, c1, cout): config 5's distinct Conv3D layers with the deepest K loops (27 taps x up to 16 chunks)
CFG5 = [
    ('cfg5 mid.conv2 512->512@16', 1, 4, 16, 512, 0, 0, 512),
    ('cfg5 dec0.up 512->256@32', 1, 4, 32, 512, 1, 0, 256),
    ('cfg5 dec0.cat 256+256->256@32', 1, 4, 32, 256, 0, 256, 256),
]


@functools.lru_cache(maxsize=1)
def _case3d(idx):
    name, nb, dep, h, c0, up0, c1, co = CFG5[idx]
    ci = c0 + c1
    rng = np.random.default_rng(3000 + idx)
    hs = h // 2 if up0 else h
    x0 = _grid(rng.standard_normal((nb, dep, hs, hs, c0)))
    x1 = _grid(rng.standard_normal((nb, dep, h, h, c1))) if c1 else None
    wt = _grid(_grid(rng.standard_normal((3, 3, 3, ci, co)) * (0.7 / np.sqrt(27 * ci)) * 8) / 8)
    b = rng.standard_normal(co).astype(np.float32) * 0.1
    dy = _grid(rng.standard_normal((nb, dep, h, h, co)))
    xin = x0.repeat(2, 2).repeat(2, 3) if up0 else x0                       # UpSampling3D (1, 2, 2)
    if c1:
        xin = np.concatenate([xin, x1], -1)
    x64, w64 = xin.astype(np.float64), wt.astype(np.float64)
    fwd = O.act_fwd(O.conv3d_same_fwd(x64, w64, b.astype(np.float64)), 'relu')
    dx, dw, _ = O.conv3d_same_bwd(x64, w64, dy.astype(np.float64))
    return dict(x0=x0, x1=x1, wt=wt, b=b, dy=dy, fwd=fwd, dx=dx, dw=dw)


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('idx', range(len(CFG5)), ids=[s[0].replace(' ', '_') for s in CFG5])
def test_real_shape_conv3d(idx, dtype):
    """Conv3D(3x3x3) at config 5's channel widths on a T = 4 volume: the depth-tap K loop (27 x 16 chunks at 512 -> 512) in the
    forward pass (with fused BN statistics where the layer has BN), the data gradient (split / (1,2,2)-summed) and the weight
    gradient, against the float64 oracle."""
    from test_gpu_ops import pack_all
    name, nb, dep, h, c0, up0, c1, co = CFG5[idx]
    ci = c0 + c1
    n = nb * dep
    k = _case3d(idx)
    L = N.lib()
    T = tdt(dtype)
    x0d = up(k['x0'].reshape((n,) + k['x0'].shape[2:]), dtype)
    x1d = up(k['x1'].reshape((n,) + k['x1'].shape[2:]), dtype) if c1 else None
    dyd, bd = up(k['dy'].reshape(n, h, h, co), dtype), f32(k['b'])
    wf, wd = pack_all(k['wt'], dtype)
    y = torch.empty((n, h, h, co), dtype=T, device=dev())
    d = conv_desc(x0d, c0, up0, x1d, c1, wf, bd, y, None, 0, n, h, h, co, N.ACT['relu'], dtype)
    d.depth, d.kd = dep, 3
    N.call('rvip_conv3x3_fwd', C.byref(d), stream())
    close(down(y).reshape(k['fwd'].shape), k['fwd'], dtype, name + ' fwd')
    if not up0:
        rows = L.rvip_conv3x3_fwd_stats_rows(C.byref(d))
        assert rows > 0
        ws = torch.full((rows * 2 * co,), 7.0, dtype=torch.float32, device=dev())
        y2 = torch.empty_like(y)
        d.y = y2.data_ptr()
        N.call('rvip_conv3x3_fwd_stats', C.byref(d), P(ws), C.c_size_t(ws.numel() * 4), stream())
        assert torch.equal(y, y2)
        st = down(ws).astype(np.float64).reshape(rows, 2, co).sum(0)
        yq = down(y).astype(np.float64).reshape(-1, co)
        np.testing.assert_allclose(st[0], yq.sum(0), atol=2e-5 * np.abs(yq).sum(0).max())
        np.testing.assert_allclose(st[1], (yq * yq).sum(0), rtol=2e-5)
    dxr = k['dx'].reshape(n, h, h, ci)
    if c1:
        g0, g1 = torch.empty((n, h, h, c0), dtype=T, device=dev()), torch.empty((n, h, h, c1), dtype=T, device=dev())
        d2 = conv_desc(dyd, co, 0, None, 0, wd, None, g0, g1, c0, n, h, h, ci, 0, dtype)
        want = dxr
    elif up0:
        g0, g1 = torch.empty((n, h // 2, h // 2, c0), dtype=T, device=dev()), None
        d2 = conv_desc(dyd, co, 0, None, 0, wd, None, g0, None, 0, n, h, h, ci, 0, dtype)
        d2.down2 = 1
        want = O.upsample_nearest_bwd(dxr)
    else:
        g0, g1 = torch.empty((n, h, h, ci), dtype=T, device=dev()), None
        d2 = conv_desc(dyd, co, 0, None, 0, wd, None, g0, None, 0, n, h, h, ci, 0, dtype)
        want = dxr
    d2.depth, d2.kd = dep, 3
    rows = L.rvip_conv3x3_fwd_sums_rows(C.byref(d2))
    assert rows > 0
    buf = torch.full((rows, ci), 7.0, dtype=torch.float32, device=dev())
    N.call('rvip_conv3x3_fwd_sums', C.byref(d2), P(buf), C.c_size_t(buf.numel() * 4), stream())
    stored = down(g0) if g1 is None else np.concatenate([down(g0), down(g1)], -1)
    close(stored, want, dtype, name + ' dgrad')
    got = down(buf).astype(np.float64).sum(0)
    assert np.abs(got - want.reshape(-1, ci).sum(0)).max() <= 2e-5 * np.abs(want).reshape(-1, ci).sum(0).max()
    wsb = L.rvip_conv3x3_wgrad_workspace(n, h, h, ci, co)
    ws = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev())
    dw = torch.full((3, 3, 3, ci, co), 7.0, dtype=torch.float32, device=dev())
    g = N.Wgrad3x3Desc()
    g.x0, g.c0, g.up0 = x0d.data_ptr(), c0, up0
    g.x1, g.c1 = (x1d.data_ptr(), c1) if c1 else (None, 0)
    g.dy, g.dw = dyd.data_ptr(), dw.data_ptr()
    g.n, g.h, g.w, g.cout, g.dtype = n, h, h, co, ndt(dtype)
    g.depth, g.kd = dep, 3
    g.workspace, g.workspace_bytes = ws.data_ptr(), wsb
    wm = f32(k['wt'])
    nd = L.rvip_conv3x3_wgrad_dot_rows(C.byref(g))
    rows_t = torch.full((nd, ci), 7.0, dtype=torch.float64, device=dev())
    g.w_master, g.dot_rows, g.dot_rows_bytes = wm.data_ptr(), rows_t.data_ptr(), rows_t.numel() * 8
    N.call('rvip_conv3x3_wgrad', C.byref(g), stream())
    scale = float(np.abs(k['dw']).max())
    assert np.abs(down(dw) - k['dw']).max() <= 2e-5 * scale * max(1.0, np.sqrt(n * h * h / 4096.0)), (name, np.abs(down(dw) - k['dw']).max() / scale)
    xin = k['x0'].repeat(2, 2).repeat(2, 3) if up0 else k['x0']
    if c1:
        xin = np.concatenate([xin, k['x1']], -1)
    ident = (xin.astype(np.float64) * k['dx']).sum((0, 1, 2, 3))
    torch.cuda.synchronize()
    t2 = rows_t.cpu().numpy().sum(0)
    s2 = (np.abs(k['wt'].astype(np.float64)) * np.abs(k['dw'])).sum((0, 1, 2, 4)).max()
    assert np.abs(t2 - ident).max() <= 2e-5 * s2 * max(1.0, np.sqrt(n * h * h / 4096.0)), (name, np.abs(t2 - ident).max() / s2)


@pytest.mark.parametrize('dtype', ['bf16', 'f16'])
@pytest.mark.parametrize('case', ['one chunk', 'concat halves', 'partial chunk', 'gated sums', 'streamed input'])
def test_three_input_stages_equal_two(case, dtype, monkeypatch):
    """the loaders of the 16-bit igemm run TWO items ahead where every weight chunk stays resident in LDS and wait with a
    COUNTED s_waitcnt vmcnt(n) -- n = the wave's DMA instructions of the younger item.  A miscount would be a silent race (compute
    waves reading a stage before it has landed), so the three-stage launch must equal the two-stage one (RVIP_IGEMM_STAGES=2) bit
    for bit on the shapes whose instruction counts differ: one K chunk, a K loop over both concat halves, a partial last chunk, the
    gated data gradient (its mask words ride as one more DMA per tile) and the non-temporal input path."""
    rng = np.random.default_rng(7)
    n, h, w_ = 2, 64, 96                                   # ragged against the 16 x 32 tiles
    c0, c1, co = {'one chunk': (32, 0, 32), 'concat halves': (32, 32, 32), 'partial chunk': (40, 0, 32), 'gated sums': (32, 0, 32),
                  'streamed input': (32, 0, 32)}[case]
    ci = c0 + c1
    x0 = up(_grid(rng.standard_normal((n, h, w_, c0))), dtype)
    x1 = up(_grid(rng.standard_normal((n, h, w_, c1))), dtype) if c1 else None
    wf, _ = pack(_grid(rng.standard_normal((3, 3, ci, co)) * 0.1), dtype)
    bits = torch.from_numpy(rng.integers(0, 2 ** 32, size=(1, n * h * w_), dtype=np.uint32).view(np.int32)).to(dev())
    outs = []
    for stages in ('3', '2'):
        if stages == '2':
            monkeypatch.setenv('RVIP_IGEMM_STAGES', '2')
        else:
            monkeypatch.delenv('RVIP_IGEMM_STAGES', raising=False)
        y = torch.zeros((n, h, w_, co), dtype=tdt(dtype), device=dev())
        if case == 'gated sums':
            d = conv_desc(x0, c0, 0, None, 0, wf, None, y, None, 0, n, h, w_, co, 0, dtype)
            d.mask_bits, d.mask_channels, d.mask_scale = bits.data_ptr(), co, 2.0
            rows = N.lib().rvip_conv3x3_fwd_sums_rows(C.byref(d))
            assert rows > 0
            sums = torch.zeros(rows * co, dtype=torch.float32, device=dev())
            N.call('rvip_conv3x3_fwd_sums', C.byref(d), P(sums), C.c_size_t(sums.numel() * 4), stream())
            outs.append((y.clone(), sums.clone()))
        else:
            d = conv_desc(x0, c0, 0, x1, c1, wf, None, y, None, 0, n, h, w_, co, N.ACT['relu'], dtype)
            if case == 'streamed input':
                d.stream_in = 1
            N.call('rvip_conv3x3_fwd', C.byref(d), stream())
            outs.append((y.clone(),))
    torch.cuda.synchronize()
    assert float(outs[0][0].float().abs().max()) > 0
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b), case


@pytest.mark.parametrize('dtype', ['bf16', 'f16'])
@pytest.mark.parametrize('idx', range(len(CFG2)), ids=[s_[0].replace(' ', '_') for s_ in CFG2])
def test_weight_and_data_gradient_as_one_launch_equal_the_two_calls(idx, dtype):
    """rvip_conv3x3_wgrad_dgrad (round 5, include/rvip_hip.h): a layer's weight gradient and data gradient as the two parts of one
    grid, cu_limit compute units each, against rvip_conv3x3_wgrad + rvip_conv3x3_fwd_sums with the same limits -- the data-gradient
    tensor(s), the column-sum rows, dw and the sum W*dW rows bit for bit -- at every layer shape of config 2, in the mode the engine
    launches the layer's data gradient in (plain / gated by Dropout keep bits / split into the concat halves / sub-pixel form of an
    up-conv).  The two calls themselves are held to the float64 oracle by the tests above.  Autodiff of Conv2D, KerasLayers.py:683,689."""
    name, n, h, c0, up0, c1, co, bn = ALL[idx]
    w_ = h
    ci = c0 + c1
    k = _case(idx)
    L = N.lib()
    T = tdt(dtype)
    x0d = up(k['x0'], dtype)
    x1d = up(k['x1'], dtype) if c1 else None
    dyd = up(k['dy'], dtype)
    _, wd_packed = pack(k['wt'], dtype)
    wmaster = f32(np.ascontiguousarray(k['wt']))

    def run(paired):
        wsb = L.rvip_conv3x3_wgrad_workspace(n, h, w_, ci, co)
        ws = torch.zeros(wsb // 4 + 16, dtype=torch.float32, device=dev())
        dw = torch.full((3, 3, ci, co), 7.0, dtype=torch.float32, device=dev())
        g = N.Wgrad3x3Desc()
        g.x0, g.c0, g.up0 = x0d.data_ptr(), c0, up0
        g.x1, g.c1 = (x1d.data_ptr(), c1) if c1 else (None, 0)
        g.dy, g.dw = dyd.data_ptr(), dw.data_ptr()
        g.n, g.h, g.w, g.cout, g.dtype = n, h, w_, co, ndt(dtype)
        g.workspace, g.workspace_bytes = ws.data_ptr(), wsb
        g.cu_limit = 128
        nd = L.rvip_conv3x3_wgrad_dot_rows(C.byref(g))
        dots = torch.zeros(nd * ci, dtype=torch.float64, device=dev())
        g.w_master, g.dot_rows, g.dot_rows_bytes = wmaster.data_ptr(), dots.data_ptr(), dots.numel() * 8
        keep = []
        if c1:                                     # the concat layer's data gradient: two outputs, the up-conv half gated by its sign bits
            g0, g1 = torch.zeros((n, h, w_, c0), dtype=T, device=dev()), torch.zeros((n, h, w_, c1), dtype=T, device=dev())
            d2 = conv_desc(dyd, co, 0, None, 0, wd_packed, None, g0, g1, c0, n, h, w_, ci, 0, dtype)
            gate = np.random.default_rng(77 + idx).random((n, h, w_, c0)) < 0.6
            gbits = torch.from_numpy(bit_planes(gate).view(np.int32)).to(dev())
            d2.mask_bits, d2.mask_channels, d2.mask_scale = gbits.data_ptr(), c0, 1.0
            keep.append(gbits)
        elif up0 and L.rvip_conv3x3_wgrad_form(C.byref(g)) == 2:
            # the 64 -> 32 up-conv at 256^2: phase-PAIR weight gradient beside the nine-tap data gradient with its 2x2 sums (what the
            # engine launches there: RVIP_BNBWD_SUBPIX_CONSUMER 'ninetap')
            g0, g1 = torch.zeros((n, h // 2, w_ // 2, c0), dtype=T, device=dev()), None
            d2 = conv_desc(dyd, co, 0, None, 0, wd_packed, None, g0, None, 0, n, h, w_, ci, 0, dtype)
            d2.down2 = 1
        elif up0:                                  # an up-conv layer: both gradients in their sub-pixel forms
            g0, g1 = torch.zeros((n, h // 2, w_ // 2, c0), dtype=T, device=dev()), None
            wsub = torch.zeros(16 * ci * co, dtype=T, device=dev())
            N.call('rvip_pack_subpixel_dgrad_weights', P(wmaster), ci, co, ndt(dtype), P(wsub), stream())
            d2 = conv_desc(dyd, co, 0, None, 0, wsub, None, g0, None, 0, n, h, w_, ci, 0, dtype)
            d2.subpix = 2
            keep.append(wsub)
        else:                                      # a plain layer behind a Dropout: the result gated by the keep bits
            g0, g1 = torch.zeros((n, h, w_, ci), dtype=T, device=dev()), None
            d2 = conv_desc(dyd, co, 0, None, 0, wd_packed, None, g0, None, 0, n, h, w_, ci, 0, dtype)
            kb = np.random.default_rng(31 + idx).random((n, h, w_, ci)) < 0.7       # (the same mask in both runs)
            kbits = torch.from_numpy(bit_planes(kb).view(np.int32)).to(dev())
            d2.mask_bits, d2.mask_channels, d2.mask_scale = kbits.data_ptr(), ci, 1.0 / 0.7
            keep.append(kbits)
        d2.cu_limit = 128
        rows = L.rvip_conv3x3_fwd_sums_rows(C.byref(d2))
        if rows <= 0:
            return None
        sums = torch.zeros(rows * ci, dtype=torch.float32, device=dev())
        if paired:
            if not L.rvip_conv3x3_wgrad_dgrad_ok(C.byref(g), C.byref(d2)):
                return None
            N.call('rvip_conv3x3_wgrad_dgrad', C.byref(g), C.byref(d2), P(sums), C.c_size_t(sums.numel() * 4), stream())
        else:
            N.call('rvip_conv3x3_wgrad', C.byref(g), stream())
            N.call('rvip_conv3x3_fwd_sums', C.byref(d2), P(sums), C.c_size_t(sums.numel() * 4), stream())
        torch.cuda.synchronize()
        return [t_ for t_ in (dw, dots, g0, g1, sums) if t_ is not None]
    one = run(True)
    assert one is not None, name + ': every layer form of config 2 has a pair kernel'
    two = run(False)
    assert float(one[0].abs().max()) > 0 and float(one[2].float().abs().max()) > 0
    for a_, b_ in zip(one, two):
        assert torch.equal(a_, b_), name
