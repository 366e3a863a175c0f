"""Per-kernel parity on a real MI355X: every entry point of include/rvip_hip.h is called through the C ABI
(ctypes) and compared with the NumPy oracle on the same seeded inputs.  Tolerances: fp32 path 1e-4 relative
to the result scale (fp32 MFMA = exact fp32 products, different summation order); bf16 path: inputs are
rounded to bf16 first, so only accumulation order and the output rounding differ (2^-8 relative)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import cmr_landmark_detection_amd as rvip
from oracle import rvip_oracle as O

pytestmark = pytest.mark.gpu
N = rvip._native
ds = __import__('importlib').import_module('cmr-landmark-detection_amd.dropout_stream')


def dev():
    return torch.device('cuda', 0)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def tdt(dtype):
    return {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[dtype]


def ndt(dtype):
    return {'bf16': N.BF16, 'f16': N.F16, 'f32': N.F32}[dtype]


def rnd(a, dtype):
    """what the device will actually see: rounded to the 16-bit storage type (as float32) or unchanged"""
    t = torch.from_numpy(np.ascontiguousarray(a, np.float32))
    return t.to(tdt(dtype)).to(torch.float32).numpy()


def up(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev()).to(tdt(dtype)).contiguous()


def f32(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev())


def down(t):
    torch.cuda.synchronize()
    return t.to(torch.float32).cpu().numpy()


def close(got, ref, dtype, what=''):
    scale = max(float(np.abs(ref).max()), 1e-6)
    tol = {'bf16': 2.0 ** -7, 'f16': 2.0 ** -10, 'f32': 2e-5}[dtype] * scale      # one output rounding + accumulation order
    err = float(np.abs(got - ref).max())
    assert err <= tol, '%s: max err %.3e > tol %.3e (scale %.3e)' % (what, err, tol, scale)


def pack(w, dtype):
    """HWIO fp32 master -> packed forward / dgrad operands via the library"""
    kh, kw, ci, co = w.shape
    wm = f32(w)
    wf = torch.empty(9 * ci * co, dtype=tdt(dtype), device=dev())
    wd = torch.empty(9 * ci * co, dtype=tdt(dtype), device=dev())
    N.call('rvip_pack_conv3x3_weights', P(wm), ci, co, ndt(dtype), P(wf), P(wd), stream())
    return wf, wd


def conv_desc(x0, c0, up0, x1, c1, wp, bias, y, y1, csplit, n, h, w, cout, act, dtype):
    d = N.Conv3x3Desc()
    d.x0, d.c0, d.up0 = x0.data_ptr(), c0, up0
    d.x1, d.c1 = (x1.data_ptr() if x1 is not None else None), c1
    d.w_packed, d.bias = wp.data_ptr(), (bias.data_ptr() if bias is not None else None)
    d.y, d.y1, d.csplit = y.data_ptr(), (y1.data_ptr() if y1 is not None else None), csplit
    d.n, d.h, d.w, d.cout, d.act, d.dtype = n, h, w, cout, act, ndt(dtype)
    return d


SHAPES = [  # n, h, w, cin, cout
    (2, 24, 40, 8, 8),        # TW=32 tiles with ragged right/bottom edges
    (1, 16, 16, 16, 40),      # TW=16 tile = whole image, two output-channel tiles with a tail
    (3, 14, 14, 24, 8),       # TW=16 with masking (the 224-input net reaches 14x14)
    (1, 40, 72, 72, 64),      # three input-channel chunks incl. a partial one (bf16: 72 = 2*32+8)
]


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', SHAPES)
def test_conv3x3_fwd_dgrad_wgrad(shape, dtype):
    n, h, w, ci, co = shape
    rng = np.random.default_rng(hash(shape) % 1000)
    x = rnd(rng.standard_normal((n, h, w, ci)), dtype)
    wt = rnd(rng.standard_normal((3, 3, ci, co)) * 0.2, dtype)
    b = rng.standard_normal(co).astype(np.float32)
    dy = rnd(rng.standard_normal((n, h, w, co)), dtype)
    xd, dyd, bd = up(x, dtype), up(dy, dtype), f32(b)
    wf, wd = pack(wt, dtype)
    # forward, relu epilogue
    y = torch.empty((n, h, w, co), dtype=tdt(dtype), device=dev())
    d = conv_desc(xd, ci, 0, None, 0, wf, bd, y, None, 0, n, h, w, co, N.ACT['relu'], dtype)
    N.call('rvip_conv3x3_fwd', C.byref(d), stream())
    ref = O.act_fwd(O.conv2d_same_fwd(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64)), 'relu')
    close(down(y), ref, dtype, 'fwd')
    # data gradient = same kernel on dy with the rotated/transposed operand
    dx = torch.empty((n, h, w, ci), dtype=tdt(dtype), device=dev())
    d2 = conv_desc(dyd, co, 0, None, 0, wd, None, dx, None, 0, n, h, w, ci, 0, dtype)
    N.call('rvip_conv3x3_fwd', C.byref(d2), stream())
    rdx, rdw, _ = O.conv2d_same_bwd(x.astype(np.float64), wt.astype(np.float64), dy.astype(np.float64))
    close(down(dx), rdx, dtype, 'dgrad')
    # weight gradient (fp32 HWIO out, deterministic split-K)
    L = N.lib()
    wsb = L.rvip_conv3x3_wgrad_workspace(n, h, w, ci, co)
    ws = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev())
    dw = torch.full((3, 3, ci, co), 7.0, dtype=torch.float32, device=dev())
    g = N.Wgrad3x3Desc()
    g.x0, g.c0, g.up0, g.x1, g.c1 = xd.data_ptr(), ci, 0, None, 0
    g.dy, g.dw = dyd.data_ptr(), dw.data_ptr()
    g.n, g.h, g.w, g.cout, g.dtype = n, h, w, co, ndt(dtype)
    g.workspace, g.workspace_bytes = ws.data_ptr(), wsb
    N.call('rvip_conv3x3_wgrad', C.byref(g), stream())
    got = down(dw)
    scale = float(np.abs(rdw).max())
    assert np.abs(got - rdw).max() <= (2e-3 if dtype != 'f32' else 2e-5) * scale, np.abs(got - rdw).max() / scale
    dw2 = torch.empty_like(dw)
    g.dw = dw2.data_ptr()
    N.call('rvip_conv3x3_wgrad', C.byref(g), stream())
    assert torch.equal(dw, dw2)                               # bitwise reproducible reduction


def pack_all(w, dtype):
    """DHWIO / HWIO fp32 master -> packed operands via the table-driven one-launch re-layout (taps = 27 or 9)"""
    taps = int(np.prod(w.shape[:-2]))
    ci, co = w.shape[-2:]
    wm = f32(w)
    wf = torch.empty(taps * ci * co, dtype=tdt(dtype), device=dev())
    wd = torch.empty(taps * ci * co, dtype=tdt(dtype), device=dev())
    tab = (N.PackEntry * 1)()
    tab[0].w_off, tab[0].f_off, tab[0].d_off, tab[0].cin, tab[0].cout, tab[0].taps = 0, 0, 0, ci, co, taps
    tabd = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev())
    N.call('rvip_pack_all_conv3x3_weights', P(wm), P(tabd), 1, taps * ci * co, ndt(dtype), P(wf), P(wd), stream())
    return wf, wd


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', [(2, 4, 24, 40, 8, 8), (1, 3, 16, 16, 16, 40), (2, 5, 40, 36, 72, 64), (1, 1, 16, 16, 8, 8)])
def test_conv3d_fwd_dgrad_wgrad(shape, dtype):
    """Conv3D(3x3x3, 'same') (KerasLayers.py:679 with 3 DIM entries; cfg 5) = the 3x3 implicit GEMM with a K loop over
    three depth taps reading the neighbouring slices of the same volume; dgrad = the same kernel on the 27 reversed
    taps; wgrad = one pass per depth tap.  Volumes of different batch entries must not leak into each other."""
    nb, dep, h, w, ci, co = shape
    n = nb * dep
    rng = np.random.default_rng(sum(shape))
    x = rnd(rng.standard_normal((nb, dep, h, w, ci)), dtype)
    wt = rnd(rng.standard_normal((3, 3, 3, ci, co)) * 0.15, dtype)
    b = rng.standard_normal(co).astype(np.float32)
    dy = rnd(rng.standard_normal((nb, dep, h, w, co)), dtype)
    xd, dyd, bd = up(x, dtype), up(dy, dtype), f32(b)
    wf, wd = pack_all(wt, dtype)
    y = torch.empty((n, h, w, co), dtype=tdt(dtype), device=dev())
    d = conv_desc(xd, ci, 0, None, 0, wf, bd, y, None, 0, n, h, w, co, N.ACT['relu'], dtype)
    d.depth, d.kd = dep, 3
    N.call('rvip_conv3x3_fwd', C.byref(d), stream())
    ref = O.act_fwd(O.conv3d_same_fwd(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64)), 'relu')
    close(down(y).reshape(ref.shape), ref, dtype, 'fwd')
    dx = torch.empty((n, h, w, ci), dtype=tdt(dtype), device=dev())
    d2 = conv_desc(dyd, co, 0, None, 0, wd, None, dx, None, 0, n, h, w, ci, 0, dtype)
    d2.depth, d2.kd = dep, 3
    N.call('rvip_conv3x3_fwd', C.byref(d2), stream())
    rdx, rdw, _ = O.conv3d_same_bwd(x.astype(np.float64), wt.astype(np.float64), dy.astype(np.float64))
    close(down(dx).reshape(rdx.shape), rdx, dtype, 'dgrad')
    L = N.lib()
    wsb = L.rvip_conv3x3_wgrad_workspace(n, h, w, ci, co)
    ws = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev())
    dw = torch.full((3, 3, 3, ci, co), 7.0, dtype=torch.float32, device=dev())
    g = N.Wgrad3x3Desc()
    g.x0, g.c0, g.up0, g.x1, g.c1 = xd.data_ptr(), ci, 0, None, 0
    g.dy, g.dw = dyd.data_ptr(), dw.data_ptr()
    g.n, g.h, g.w, g.cout, g.dtype = n, h, w, co, ndt(dtype)
    g.depth, g.kd = dep, 3
    g.workspace, g.workspace_bytes = ws.data_ptr(), wsb
    N.call('rvip_conv3x3_wgrad', C.byref(g), stream())
    got = down(dw)
    scale = float(np.abs(rdw).max())
    assert np.abs(got - rdw).max() <= (2e-3 if dtype != 'f32' else 2e-5) * scale, np.abs(got - rdw).max() / scale


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('ci', [2, 3, 4])
def test_first_layer_with_several_image_channels(dtype, ci):
    """IMG_CHANNELS > 1 (Unets.py:77): rvip_conv3x3_cn_fwd / rvip_conv3x3_cn_wgrad against the oracle's conv (ragged tile edges)."""
    n, h, w, co = 3, 20, 40, 16
    rng = np.random.default_rng(50 + ci)
    x = rnd(rng.random((n, h, w, ci)), dtype)
    wt = (rng.standard_normal((3, 3, ci, co)) * 0.3).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    dy = rnd(rng.standard_normal((n, h, w, co)), dtype)
    xd, dyd, wdv, bd = up(x, dtype), up(dy, dtype), f32(wt), f32(b)
    y = torch.empty((n, h, w, co), dtype=tdt(dtype), device=dev())
    N.call('rvip_conv3x3_cn_fwd', P(xd), P(wdv), P(bd), P(y), n, h, w, ci, co, N.ACT['relu'], ndt(dtype), stream())
    ref = O.act_fwd(O.conv2d_same_fwd(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64)), 'relu')
    close(down(y), ref, dtype, 'cn fwd')
    L = N.lib()
    wsb = L.rvip_reduce_workspace(n * h * w, 16 * co)
    ws = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev())
    dw = torch.full((3, 3, ci, co), 7.0, dtype=torch.float32, device=dev())
    N.call('rvip_conv3x3_cn_wgrad', P(xd), P(dyd), P(dw), n, h, w, ci, co, ndt(dtype), P(ws), C.c_size_t(wsb), stream())
    _, rdw, _ = O.conv2d_same_bwd(x.astype(np.float64), wt.astype(np.float64), dy.astype(np.float64))
    assert np.abs(down(dw) - rdw).max() <= 2e-5 * float(np.abs(rdw).max())
    assert L.rvip_conv3x3_cn_fwd(P(xd), P(wdv), P(bd), P(y), n, h, w, 5, co, N.ACT['relu'], ndt(dtype), stream()) == -1


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
def test_conv3d_first_layer(dtype):
    nb, dep, h, w, co = 2, 3, 20, 40, 16
    n = nb * dep
    rng = np.random.default_rng(5)
    x = rnd(rng.random((nb, dep, h, w, 1)), dtype)
    wt = (rng.standard_normal((3, 3, 3, 1, co)) * 0.3).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    dy = rnd(rng.standard_normal((nb, dep, h, w, co)), dtype)
    xd, dyd, wdv, bd = up(x, dtype), up(dy, dtype), f32(wt), f32(b)
    y = torch.empty((n, h, w, co), dtype=tdt(dtype), device=dev())
    N.call('rvip_conv3d_c1_fwd', P(xd), P(wdv), P(bd), P(y), n, dep, h, w, co, N.ACT['relu'], ndt(dtype), stream())
    ref = O.act_fwd(O.conv3d_same_fwd(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64)), 'relu')
    close(down(y).reshape(ref.shape), ref, dtype, 'c1 3-D fwd')
    L = N.lib()
    wsb = L.rvip_reduce_workspace(n * h * w, 16 * co)
    ws = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev())
    dw = torch.empty((3, 3, 3, 1, co), dtype=torch.float32, device=dev())
    N.call('rvip_conv3d_c1_wgrad', P(xd), P(dyd), P(dw), n, dep, h, w, co, ndt(dtype), P(ws), C.c_size_t(wsb), stream())
    _, rdw, _ = O.conv3d_same_bwd(x.astype(np.float64), wt.astype(np.float64), dy.astype(np.float64))
    assert np.abs(down(dw) - rdw).max() <= 2e-5 * float(np.abs(rdw).max())


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', [(2, 40, 72, 32, 32), (3, 16, 16, 32, 24), (2, 24, 40, 8, 8),
                                   (2, 40, 72, 32, 64), (1, 16, 16, 64, 72), (2, 8, 40, 16, 128), (1, 32, 32, 256, 96)])   # two channel tiles per workgroup
def test_conv3x3_fused_batchnorm_statistics(shape, dtype):
    """conv + bias + ReLU with the BatchNormalization statistics of the STORED output folded into the epilogue, then
    rvip_bn_stats_finalize: same mean / invstd / moving statistics as the two-pass kernels and the oracle."""
    n, h, w, ci, co = shape
    rng = np.random.default_rng(11)
    x = rnd(rng.standard_normal((n, h, w, ci)), dtype)
    wt = rnd(rng.standard_normal((3, 3, ci, co)) * 0.2, dtype)
    b = rng.standard_normal(co).astype(np.float32)
    xd, bd = up(x, dtype), f32(b)
    wf, _ = pack(wt, dtype)
    y = torch.empty((n, h, w, co), dtype=tdt(dtype), device=dev())
    d = conv_desc(xd, ci, 0, None, 0, wf, bd, y, None, 0, n, h, w, co, N.ACT['relu'], dtype)
    L = N.lib()
    rows = L.rvip_conv3x3_fwd_stats_rows(C.byref(d))
    assert rows > 0
    wsb = rows * 2 * co * 4
    ws = torch.full((rows * 2 * co + 16,), 123.0, dtype=torch.float32, device=dev())
    N.call('rvip_conv3x3_fwd_stats', C.byref(d), P(ws), C.c_size_t(wsb), stream())
    gamma = (1 + 0.3 * rng.standard_normal(co)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(co)).astype(np.float32)
    gd, btd, mm, mv = f32(gamma), f32(beta), f32(np.zeros(co)), f32(np.ones(co))
    mean, invstd, scale, shift = (torch.empty(co, dtype=torch.float32, device=dev()) for _ in range(4))
    N.call('rvip_bn_stats_finalize', P(ws), rows, C.c_longlong(n * h * w), co, P(gd), P(btd), P(mm), P(mv), 0.99, 1e-3, 1,
           P(mean), P(invstd), P(scale), P(shift), stream())
    yq = down(y).astype(np.float64)                                   # statistics are those of the stored tensor
    ref = O.act_fwd(O.conv2d_same_fwd(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64)), 'relu')
    close(yq, ref, dtype, 'fwd (stats variant)')
    _, cache = O.bn_train_fwd(yq, gamma.astype(np.float64), beta.astype(np.float64))
    np.testing.assert_allclose(down(mean), cache[2], atol=3e-6 * max(1.0, np.abs(cache[2]).max()))
    np.testing.assert_allclose(down(invstd), cache[1], rtol=1e-5)
    rm, rv = O.bn_moving_update(np.zeros(co), np.ones(co), cache[2], cache[3], n * h * w)
    np.testing.assert_allclose(down(mv), rv, rtol=1e-5)
    np.testing.assert_allclose(down(scale), gamma * cache[1], rtol=1e-5)


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
def test_conv3x3_virtual_upsample_concat_and_split(dtype):
    """UpSampling2D + Conv2D and Concatenate + Conv2D as addressing modes; dgrad of the concat conv writes the
    two halves of the gradient to separate tensors."""
    n, h, w = 2, 16, 32
    c0, c1, co = 16, 8, 24
    rng = np.random.default_rng(3)
    lo = rnd(rng.standard_normal((n, h // 2, w // 2, c0)), dtype)
    sk = rnd(rng.standard_normal((n, h, w, c1)), dtype)
    wt = rnd(rng.standard_normal((3, 3, c0 + c1, co)) * 0.2, dtype)
    wf, wd = pack(wt, dtype)
    y = torch.empty((n, h, w, co), dtype=tdt(dtype), device=dev())
    lod, skd = up(lo, dtype), up(sk, dtype)
    d = conv_desc(lod, c0, 1, skd, c1, wf, None, y, None, 0, n, h, w, co, 0, dtype)
    N.call('rvip_conv3x3_fwd', C.byref(d), stream())
    xcat = np.concatenate([O.upsample_nearest_fwd(lo), sk], -1).astype(np.float64)
    close(down(y), O.conv2d_same_fwd(xcat, wt.astype(np.float64)), dtype, 'up+concat fwd')
    dy = rnd(rng.standard_normal((n, h, w, co)), dtype)
    g0 = torch.empty((n, h, w, c0), dtype=tdt(dtype), device=dev())
    g1 = torch.empty((n, h, w, c1), dtype=tdt(dtype), device=dev())
    dyd = up(dy, dtype)
    d2 = conv_desc(dyd, co, 0, None, 0, wd, None, g0, g1, c0, n, h, w, c0 + c1, 0, dtype)
    N.call('rvip_conv3x3_fwd', C.byref(d2), stream())
    rdx, rdw, _ = O.conv2d_same_bwd(xcat, wt.astype(np.float64), dy.astype(np.float64))
    close(down(g0), rdx[..., :c0], dtype, 'split dgrad 0')
    close(down(g1), rdx[..., c0:], dtype, 'split dgrad 1')
    glo = torch.empty((n, h // 2, w // 2, c0), dtype=tdt(dtype), device=dev())
    N.call('rvip_upsample2x_bwd', P(g0), P(glo), n, h // 2, w // 2, c0, ndt(dtype), stream())
    close(down(glo), O.upsample_nearest_bwd(rnd(down(g0), dtype).astype(np.float64)), dtype, 'upsample bwd')
    L = N.lib()
    wsb = L.rvip_conv3x3_wgrad_workspace(n, h, w, c0 + c1, co)
    ws = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev())
    dw = torch.empty((3, 3, c0 + c1, co), dtype=torch.float32, device=dev())
    g = N.Wgrad3x3Desc()
    g.x0, g.c0, g.up0, g.x1, g.c1 = lod.data_ptr(), c0, 1, skd.data_ptr(), c1
    g.dy, g.dw = dyd.data_ptr(), dw.data_ptr()
    g.n, g.h, g.w, g.cout, g.dtype = n, h, w, co, ndt(dtype)
    g.workspace, g.workspace_bytes = ws.data_ptr(), wsb
    N.call('rvip_conv3x3_wgrad', C.byref(g), stream())
    scale = float(np.abs(rdw).max())
    assert np.abs(down(dw) - rdw).max() <= (2e-3 if dtype != 'f32' else 2e-5) * scale
    ups = torch.empty((n, h, w, c0), dtype=tdt(dtype), device=dev())
    N.call('rvip_upsample2x_fwd', P(lod), P(ups), n, h // 2, w // 2, c0, ndt(dtype), stream())
    np.testing.assert_array_equal(down(ups), O.upsample_nearest_fwd(lo))


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', [(2, 24, 40, 32, 64), (1, 16, 16, 16, 40), (2, 8, 72, 8, 8),       # igemm v2: 512 px, TW=16, 256 px
                                   (1, 32, 64, 16, 256), (2, 16, 16, 8, 256), (1, 8, 40, 8, 264)])   # igemm v3 (K >= 256): same three tilings
def test_conv3x3_dgrad_with_fused_2x2_sum(shape, dtype):
    """down2: the data gradient of an UpSampling2D -> Conv2D pair written directly at half resolution (sum of each 2x2
    block, KerasLayers.py:756-758 autodiff) = conv on dy with the rotated kernel followed by the 2x2 sum."""
    n, h, w, ci, co = shape                     # forward conv ci -> co at h x w; its dgrad contracts over co, outputs ci
    rng = np.random.default_rng(sum(shape))
    wt = rnd(rng.standard_normal((3, 3, ci, co)) * 0.2, dtype)
    dy = rnd(rng.standard_normal((n, h, w, co)), dtype)
    _, wd = pack(wt, dtype)
    dyd = up(dy, dtype)
    glo = torch.full((n, h // 2, w // 2, ci), 5.0, dtype=tdt(dtype), device=dev())
    d = conv_desc(dyd, co, 0, None, 0, wd, None, glo, None, 0, n, h, w, ci, 0, dtype)
    d.down2 = 1
    N.call('rvip_conv3x3_fwd', C.byref(d), stream())
    rdx = O.conv2d_same_bwd(np.zeros((n, h, w, ci)), wt.astype(np.float64), dy.astype(np.float64))[0]
    close(down(glo), O.upsample_nearest_bwd(rdx), dtype, 'down2 dgrad')


@pytest.mark.parametrize('dtype', ['bf16', 'f16'])
@pytest.mark.parametrize('shape', [(2, 48, 80, 32, 64), (1, 32, 32, 16, 40), (2, 16, 144, 8, 8), (1, 64, 64, 64, 32), (3, 32, 32, 128, 64),
                                   (1, 64, 128, 40, 72), (2, 8, 8, 256, 128)])
def test_upsample_conv_dgrad_subpixel_form(shape, dtype):
    """subpix = 2: the data gradient of UpSampling2D -> Conv2D in its sub-pixel form (K loop over the four source phases of the
    full-resolution gradient x 2x2 summed taps; result on the low-resolution grid, no 2x2-sum epilogue) against the float64 oracle
    (conv with the rotated kernel, then the 2x2 sum), with and without the column sums of rvip_conv3x3_fwd_sums -- and the nine-tap
    down2 launch next to it.  bf16 / f16: the summed taps are rounded once (as in the forward form)."""
    n, h, w, ci, co = shape                     # forward conv ci -> co at the up-sampled size h x w; the gradient dy is [n, h, w, co]
    rng = np.random.default_rng(sum(shape) + 1)
    wt = (rng.standard_normal((3, 3, ci, co)) * 0.2).astype(np.float32)
    dy = rnd(rng.standard_normal((n, h, w, co)), dtype)
    wm, dyd = f32(wt), up(dy, dtype)
    wph = torch.empty(16 * ci * co, dtype=tdt(dtype), device=dev())
    N.call('rvip_pack_subpixel_dgrad_weights', P(wm), ci, co, ndt(dtype), P(wph), stream())
    L = N.lib()

    def desc(out):
        d = conv_desc(dyd, co, 0, None, 0, wph, None, out, None, 0, n, h, w, ci, 0, dtype)
        d.subpix = 2
        return d
    g0 = torch.full((n, h // 2, w // 2, ci), 5.0, dtype=tdt(dtype), device=dev())
    N.call('rvip_conv3x3_fwd', C.byref(desc(g0)), stream())
    g1 = torch.full_like(g0, 6.0)
    d1 = desc(g1)
    nr = L.rvip_conv3x3_fwd_sums_rows(C.byref(d1))
    assert nr > 0
    rows = torch.full((nr, ci), 7.0, dtype=torch.float32, device=dev())
    N.call('rvip_conv3x3_fwd_sums', C.byref(d1), P(rows), C.c_size_t(rows.numel() * 4), stream())
    torch.cuda.synchronize()
    assert torch.equal(g0, g1)
    # oracle: with the kernel as the launch sees it (summed taps rounded to the storage type) the result is exact up to accumulation order
    ref = O.upsample_nearest_bwd(O.conv2d_same_bwd(np.zeros((n, h, w, ci)), wt.astype(np.float64), dy.astype(np.float64))[0])
    scale = float(np.abs(ref).max())
    got = down(g0).astype(np.float64)
    assert np.abs(got - ref).max() <= 2.0 ** -6 * scale, np.abs(got - ref).max() / scale
    # the nine-tap form of the same launch (taps rounded one by one) agrees within the same bound
    wr = rnd(wt, dtype)
    _, wd9 = pack(wr, dtype)
    g9 = torch.full_like(g0, 4.0)
    d9 = conv_desc(dyd, co, 0, None, 0, wd9, None, g9, None, 0, n, h, w, ci, 0, dtype)
    d9.down2 = 1
    N.call('rvip_conv3x3_fwd', C.byref(d9), stream())
    assert np.abs(got - down(g9).astype(np.float64)).max() <= 2.0 ** -6 * scale
    # column sums: of the fp32 values in front of the storage rounding
    sums = down(rows).astype(np.float64).sum(0)
    want = got.sum((0, 1, 2))
    tol = {'bf16': 2.0 ** -8, 'f16': 2.0 ** -11}[dtype] * np.abs(got).sum((0, 1, 2)).max() + 1e-6
    assert np.abs(sums - want).max() <= tol, (np.abs(sums - want).max(), tol)
    # exact check of the packed phase kernels: what the launch multiplies with
    wph_h = down(wph).reshape(4, 4, ci, co).astype(np.float64)
    sets = {(0, 0): (1, 2), (0, 1): (0,), (1, 0): (2,), (1, 1): (0, 1)}
    for al in range(2):
        for be in range(2):
            for u in range(2):
                for v in range(2):
                    acc = sum(wt[kh, kw].astype(np.float32) for kh in sets[(al, u)] for kw in sets[(be, v)])
                    np.testing.assert_array_equal(wph_h[2 * al + be, 2 * u + v], rnd(acc, dtype).astype(np.float64))
    # refused: f32, an up-sampling read, a bias
    d = desc(g0)
    d.up0 = 1
    assert L.rvip_conv3x3_fwd(C.byref(d), stream()) == -1
    d = desc(g0)
    d.dtype = N.F32
    assert L.rvip_conv3x3_fwd(C.byref(d), stream()) == -1


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', [(2, 48, 80, 32, 32), (1, 32, 32, 16, 40), (2, 16, 72, 8, 8), (1, 64, 64, 72, 64)])
def test_upsample_conv_subpixel_form(shape, dtype):
    """UpSampling2D(2) -> Conv2D(3x3, same) (KerasLayers.py:756-758) as four 2x2-tap phase convolutions on the
    low-resolution input: same result as the conv on the materialised up-sampled tensor (fp32: summation order
    only; bf16: the phase kernels are sums of taps rounded once)."""
    n, h, w, ci, co = shape                     # h, w = output (up-sampled) size
    rng = np.random.default_rng(sum(shape))
    lo = rnd(rng.standard_normal((n, h // 2, w // 2, ci)), dtype)
    wt = rnd(rng.standard_normal((3, 3, ci, co)) * 0.2, dtype)
    b = rng.standard_normal(co).astype(np.float32)
    wm, bd, lod = f32(wt), f32(b), up(lo, dtype)
    wph = torch.empty(16 * ci * co, dtype=tdt(dtype), device=dev())
    N.call('rvip_pack_subpixel_weights', P(wm), ci, co, ndt(dtype), P(wph), stream())
    y = torch.full((n, h, w, co), 9.0, dtype=tdt(dtype), device=dev())
    d = conv_desc(lod, ci, 1, None, 0, wph, bd, y, None, 0, n, h, w, co, N.ACT['relu'], dtype)
    d.subpix = 1
    N.call('rvip_conv3x3_fwd', C.byref(d), stream())
    ref = O.act_fwd(O.conv2d_same_fwd(O.upsample_nearest_fwd(lo).astype(np.float64), wt.astype(np.float64), b.astype(np.float64)), 'relu')
    got = down(y)
    scale = float(np.abs(ref).max())
    tol = (2.0 ** -6 if dtype != 'f32' else 2e-5) * scale      # bf16: + one rounding of the summed taps
    assert np.abs(got - ref).max() <= tol, np.abs(got - ref).max() / scale


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', [(2, 16, 32, 32, 32), (3, 24, 40, 64, 32), (1, 32, 32, 128, 64), (2, 64, 64, 64, 32), (1, 64, 128, 72, 40),
                                   (5, 8, 8, 8, 8), (2, 32, 64, 256, 128)])
def test_upsample_conv_wgrad_subpixel_form(shape, dtype):
    """Weight gradient of UpSampling2D(2) -> Conv2D(3x3, same) in its sub-pixel form (16-bit types: wgrad3x3_ws<..., TAPS = 4>, four
    2x2-tap phase contractions of the low-resolution X with the stride-2 phase images of dY, each workgroup writing its four blocks into
    the nine-tap slab positions they belong to) against the same entry point with RVIP_SUBPIX_WGRAD=0 (the nine-tap contraction over
    the virtually up-sampled X) -- fp32 summation order only -- and against the float64 oracle.  f32 does not take the form."""
    n, h, w, ci, co = shape                     # h, w = up-sampled size = size of dy
    rng = np.random.default_rng(sum(shape) + 5)
    lo = rnd(rng.standard_normal((n, h // 2, w // 2, ci)), dtype)
    dy = rnd(rng.standard_normal((n, h, w, co)), dtype)
    wt = (rng.standard_normal((3, 3, ci, co)) * 0.2).astype(np.float32)
    lod, dyd, wm = up(lo, dtype), up(dy, dtype), f32(wt)
    L = N.lib()
    wsb = L.rvip_conv3x3_wgrad_workspace(n, h, w, ci, co)
    wsd = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev())

    def run(flag, dot):
        old = os.environ.get('RVIP_SUBPIX_WGRAD')
        os.environ['RVIP_SUBPIX_WGRAD'] = flag
        try:
            wsd.fill_(float('nan'))
            dw = torch.full((3, 3, ci, co), 7.0, dtype=torch.float32, device=dev())
            g = N.Wgrad3x3Desc()
            g.x0, g.c0, g.up0, g.x1, g.c1 = lod.data_ptr(), ci, 1, None, 0
            g.dy, g.dw = dyd.data_ptr(), dw.data_ptr()
            g.n, g.h, g.w, g.cout, g.dtype = n, h, w, co, ndt(dtype)
            g.workspace, g.workspace_bytes = wsd.data_ptr(), wsb
            rows = None
            if dot:
                nr = L.rvip_conv3x3_wgrad_dot_rows(C.byref(g))
                rows = torch.full((nr, ci), 7.0, dtype=torch.float64, device=dev())
                g.w_master, g.dot_rows, g.dot_rows_bytes = wm.data_ptr(), rows.data_ptr(), rows.numel() * 8
            ns = L.rvip_conv3x3_wgrad_splits(C.byref(g))
            N.call('rvip_conv3x3_wgrad', C.byref(g), stream())
            torch.cuda.synchronize()
            return down(dw).astype(np.float64), (rows.cpu().numpy().sum(0) if dot else None), ns
        finally:
            if old is None:
                del os.environ['RVIP_SUBPIX_WGRAD']
            else:
                os.environ['RVIP_SUBPIX_WGRAD'] = old
    dw9, rows9, ns9 = run('0', True)
    dw4, rows4, ns4 = run('2', True)              # 2: the form wherever it is eligible (the default takes it for 64 x 64 blocks only)
    dw4b, _, _ = run('2', False)
    np.testing.assert_array_equal(dw4, dw4b)                 # plain fold and dot-row fold of the same slabs
    assert 0 < ns4 * 9 * ci * co * 4 <= wsb and 0 < ns9 * 9 * ci * co * 4 <= wsb
    if dtype == 'f32':
        np.testing.assert_array_equal(dw4, dw9)
        assert ns4 == ns9
    else:
        assert ns4 % 4 == 0
        scale = np.abs(dw9).max()
        assert np.abs(dw4 - dw9).max() <= 2e-5 * scale, np.abs(dw4 - dw9).max() / scale
        assert np.abs(rows4 - rows9).max() <= 2e-5 * np.abs(rows9).max() + 1e-9
    _, rdw, _ = O.conv2d_same_bwd(O.upsample_nearest_fwd(lo).astype(np.float64), wt.astype(np.float64), dy.astype(np.float64))
    close(dw4, rdw, 'f32', 'sub-pixel wgrad vs oracle')        # operands are exact in the storage type: fp32 accumulation is the only error


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', [(2, 12, 20, 16, 8), (1, 8, 8, 8, 24)])
def test_conv2d_transpose_as_zero_stuffed_conv(shape, dtype):
    """Conv2DTranspose(3, strides=2, 'same') = the 3x3 igemm over the zero-stuffed read (up0 = 2) with the kernel in its
    equivalent forward form Weq[t][ci][co] = W_hwoi[2-t][co][ci]; forward, input gradient (odd-position gather of the
    data gradient) and weight gradient against the oracle's transpose-conv."""
    n, hl, wl, ci, co = shape                       # low-resolution input
    h, w = 2 * hl, 2 * wl
    rng = np.random.default_rng(21)
    x = rnd(rng.standard_normal((n, hl, wl, ci)), dtype)
    wt = rnd(rng.standard_normal((3, 3, co, ci)) * 0.2, dtype)          # Keras HWOI
    b = rng.standard_normal(co).astype(np.float32)
    weq = np.ascontiguousarray(wt[::-1, ::-1].transpose(0, 1, 3, 2))    # [3][3][ci][co]
    wf, wd = pack(weq, dtype)
    xd, bd = up(x, dtype), f32(b)
    y = torch.empty((n, h, w, co), dtype=tdt(dtype), device=dev())
    d = conv_desc(xd, ci, 2, None, 0, wf, bd, y, None, 0, n, h, w, co, 0, dtype)
    N.call('rvip_conv3x3_fwd', C.byref(d), stream())
    ref = O.conv2d_transpose_same_fwd(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    close(down(y), ref, dtype, 'transpose fwd')
    dy = rnd(rng.standard_normal((n, h, w, co)), dtype)
    rdx, rdw, _ = O.conv2d_transpose_same_bwd(x.astype(np.float64), wt.astype(np.float64), dy.astype(np.float64))
    dyd = up(dy, dtype)
    gfull = torch.empty((n, h, w, ci), dtype=tdt(dtype), device=dev())
    d2 = conv_desc(dyd, co, 0, None, 0, wd, None, gfull, None, 0, n, h, w, ci, 0, dtype)
    N.call('rvip_conv3x3_fwd', C.byref(d2), stream())
    dx = torch.empty((n, hl, wl, ci), dtype=tdt(dtype), device=dev())
    N.call('rvip_subsample_odd', P(gfull), P(dx), n, hl, wl, ci, ndt(dtype), stream())
    close(down(dx), rdx, dtype, 'transpose dgrad')
    L = N.lib()
    wsb = L.rvip_conv3x3_wgrad_workspace(n, h, w, ci, co)
    ws = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev())
    dw = torch.empty((3, 3, ci, co), dtype=torch.float32, device=dev())
    g = N.Wgrad3x3Desc()
    g.x0, g.c0, g.up0, g.x1, g.c1 = xd.data_ptr(), ci, 2, None, 0
    g.dy, g.dw = dyd.data_ptr(), dw.data_ptr()
    g.n, g.h, g.w, g.cout, g.dtype = n, h, w, co, ndt(dtype)
    g.workspace, g.workspace_bytes = ws.data_ptr(), wsb
    N.call('rvip_conv3x3_wgrad', C.byref(g), stream())
    dw_hwoi = down(dw).transpose(0, 1, 3, 2)[::-1, ::-1]                  # back to Keras' layout
    assert np.abs(dw_hwoi - rdw).max() <= (2e-3 if dtype != 'f32' else 2e-5) * np.abs(rdw).max()


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', [(2, 20, 36, 16), (3, 256, 256, 32), (1, 9, 70, 64)])
def test_first_layer_c1_fused_batchnorm_statistics(shape, dtype):
    """rvip_conv3x3_c1_fwd_stats + rvip_bn_stats_finalize = rvip_conv3x3_c1_fwd + the statistics of the stored tensor
    (ragged tiles, more tiles than workgroups at 256 x 256)."""
    n, h, w, co = shape
    rng = np.random.default_rng(41)
    x = rnd(rng.random((n, h, w, 1)), dtype)
    wt = (rng.standard_normal((3, 3, 1, co)) * 0.5).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    xd, wtd, bd = up(x, dtype), f32(wt), f32(b)
    y = torch.empty((n, h, w, co), dtype=tdt(dtype), device=dev())
    y2 = torch.empty_like(y)
    L = N.lib()
    rows = L.rvip_conv3x3_c1_fwd_stats_rows(n, h, w, co, ndt(dtype))
    assert rows > 0
    ws = torch.full((rows * 2 * co + 16,), 123.0, dtype=torch.float32, device=dev())
    N.call('rvip_conv3x3_c1_fwd_stats', P(xd), P(wtd), P(bd), P(y), n, h, w, co, N.ACT['relu'], ndt(dtype), P(ws), C.c_size_t(rows * 2 * co * 4), stream())
    N.call('rvip_conv3x3_c1_fwd', P(xd), P(wtd), P(bd), P(y2), n, h, w, co, N.ACT['relu'], ndt(dtype), stream())
    assert torch.equal(y, y2)                                            # the stored tensor is the plain kernel's, bit for bit
    gamma = (1 + 0.3 * rng.standard_normal(co)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(co)).astype(np.float32)
    gd, btd, mm, mv = f32(gamma), f32(beta), f32(np.zeros(co)), f32(np.ones(co))
    mean, invstd, scale, shift = (torch.empty(co, dtype=torch.float32, device=dev()) for _ in range(4))
    N.call('rvip_bn_stats_finalize', P(ws), rows, C.c_longlong(n * h * w), co, P(gd), P(btd), P(mm), P(mv), 0.99, 1e-3, 1,
           P(mean), P(invstd), P(scale), P(shift), stream())
    yq = down(y).astype(np.float64)
    _, cache = O.bn_train_fwd(yq, gamma.astype(np.float64), beta.astype(np.float64))
    np.testing.assert_allclose(down(mean), cache[2], atol=3e-6 * max(1.0, np.abs(cache[2]).max()))
    np.testing.assert_allclose(down(invstd), cache[1], rtol=2e-5)
    assert N.lib().rvip_conv3x3_c1_fwd_stats(P(xd), P(wtd), P(bd), P(y), n, h, w, co, 1, ndt(dtype), P(ws), C.c_size_t(16), stream()) == -3   # RVIP_EWORKSPACE


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
def test_first_layer_c1(dtype):
    n, h, w, co = 2, 20, 36, 16
    rng = np.random.default_rng(4)
    x = rnd(rng.random((n, h, w, 1)), dtype)
    wt = (rng.standard_normal((3, 3, 1, co)) * 0.5).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    xd = up(x, dtype)
    y = torch.empty((n, h, w, co), dtype=tdt(dtype), device=dev())
    wtd, bd = f32(wt), f32(b)                 # keep device operands alive across the async launch
    N.call('rvip_conv3x3_c1_fwd', P(xd), P(wtd), P(bd), P(y), n, h, w, co, N.ACT['elu'], ndt(dtype), stream())
    ref = O.act_fwd(O.conv2d_same_fwd(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64)), 'elu')
    close(down(y), ref, dtype, 'c1 fwd')
    dy = rnd(rng.standard_normal((n, h, w, co)), dtype)
    L = N.lib()
    wsb = L.rvip_reduce_workspace(n * h * w, 16 * co)
    ws = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev())
    dw = torch.empty((3, 3, 1, co), dtype=torch.float32, device=dev())
    dyd = up(dy, dtype)
    N.call('rvip_conv3x3_c1_wgrad', P(xd), P(dyd), P(dw), n, h, w, co, ndt(dtype), P(ws), C.c_size_t(wsb), stream())
    _, rdw, _ = O.conv2d_same_bwd(x.astype(np.float64), wt.astype(np.float64), dy.astype(np.float64))
    assert np.abs(down(dw) - rdw).max() <= 2e-5 * np.abs(rdw).max()


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('act_after', [0, 1])
def test_batchnorm_dropout_pool_forward_backward(dtype, act_after):
    n, h, w, c = 3, 12, 20, 16
    rows = n * h * w
    rng = np.random.default_rng(5)
    pre = rng.standard_normal((n, h, w, c)) * 1.5 + 0.3
    z = rnd(pre if act_after else np.maximum(pre, 0), dtype)          # act before BN: z is the post-ReLU tensor
    gamma = (1 + 0.3 * rng.standard_normal(c)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(c)).astype(np.float32)
    mm0, mv0 = rng.standard_normal(c).astype(np.float32), (1 + rng.random(c)).astype(np.float32)
    L = N.lib()
    wsb = L.rvip_reduce_workspace(rows, 16 * c)
    ws = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev())
    zd, gd, bd, mm, mv = up(z, dtype), f32(gamma), f32(beta), f32(mm0), f32(mv0)
    mean, invstd, scale, shift = (torch.empty(c, dtype=torch.float32, device=dev()) for _ in range(4))
    N.call('rvip_bn_train_stats', P(zd), C.c_longlong(rows), c, ndt(dtype), P(gd), P(bd), P(mm), P(mv), 0.99, 1e-3, 1,
           P(mean), P(invstd), P(scale), P(shift), P(ws), C.c_size_t(wsb), stream())
    z64 = z.astype(np.float64)
    ybn, cache = O.bn_train_fwd(z64, gamma.astype(np.float64), beta.astype(np.float64))
    np.testing.assert_allclose(down(mean), cache[2], rtol=0, atol=2e-6)
    np.testing.assert_allclose(down(invstd), cache[1], rtol=2e-6)
    rm, rv = O.bn_moving_update(mm0.astype(np.float64), mv0.astype(np.float64), cache[2], cache[3], rows)
    np.testing.assert_allclose(down(mm), rm, atol=1e-6)
    np.testing.assert_allclose(down(mv), rv, atol=1e-6)
    # apply: BN affine (+ReLU after BN) + dropout(stream) + 2x2 max-pool in one pass
    state = torch.zeros(8, dtype=torch.int32, device=dev())
    state[N.STATE_SEED], state[N.STATE_STEP] = 1234, 7
    rate, lid = 0.4, 3
    y = torch.empty((n, h, w, c), dtype=tdt(dtype), device=dev())
    pooled = torch.empty((n, h // 2, w // 2, c), dtype=tdt(dtype), device=dev())
    a = N.ApplyDesc()
    a.z, a.y, a.pooled = zd.data_ptr(), y.data_ptr(), pooled.data_ptr()
    a.scale, a.shift, a.act = scale.data_ptr(), shift.data_ptr(), (N.ACT['relu'] if act_after else 0)
    a.drop_rate, a.mask, a.state, a.layer_id = rate, None, state.data_ptr(), lid
    a.n, a.h, a.w, a.c, a.dtype = n, h, w, c, ndt(dtype)
    N.call('rvip_bn_apply', C.byref(a), stream())
    keep = ds.keep_mask((n, h, w, c), rate, 1234, 7, lid).astype(np.float64)
    yact = np.maximum(ybn, 0) if act_after else ybn
    yref = yact * keep / np.float32(1 - rate)
    close(down(y), yref, dtype, 'apply')
    yq = down(y).astype(np.float64)
    pref, idx = O.maxpool2x2_fwd(yq)
    np.testing.assert_array_equal(down(pooled), pref)                   # pool of what was stored: exact
    # injected mask gives the same result as the stream
    md = torch.from_numpy(keep.astype(np.uint8)).to(dev())
    y2 = torch.empty_like(y)
    a.y, a.pooled, a.mask = y2.data_ptr(), None, md.data_ptr()
    N.call('rvip_bn_apply', C.byref(a), stream())
    assert torch.equal(y, y2)
    # backward: maxpool route (+skip add) -> BN backward
    dp = rnd(rng.standard_normal((n, h // 2, w // 2, c)), dtype)
    addg = rnd(rng.standard_normal((n, h, w, c)), dtype)
    gy = torch.empty((n, h, w, c), dtype=tdt(dtype), device=dev())
    dpd, addd = up(dp, dtype), up(addg, dtype)
    N.call('rvip_maxpool2x2_bwd', P(y), P(dpd), P(addd), P(gy), n, h, w, c, ndt(dtype), stream())
    gref = O.maxpool2x2_bwd(dp.astype(np.float64), idx, yq.shape) + addg
    close(down(gy), gref, dtype, 'maxpool bwd')
    g_in = rnd(down(gy), dtype).astype(np.float64)                      # what the BN backward kernels read
    dz = torch.empty((n, h, w, c), dtype=tdt(dtype), device=dev())
    dgamma, dbeta, dbias = (torch.empty(c, dtype=torch.float32, device=dev()) for _ in range(3))
    coef = torch.empty(3 * c, dtype=torch.float32, device=dev())
    b = N.BnBwdDesc()
    b.dy, b.z, b.dz = gy.data_ptr(), zd.data_ptr(), dz.data_ptr()
    b.gamma, b.mean, b.invstd = gd.data_ptr(), mean.data_ptr(), invstd.data_ptr()
    b.scale, b.shift = scale.data_ptr(), shift.data_ptr()
    b.dgamma, b.dbeta, b.dbias, b.coef = dgamma.data_ptr(), dbeta.data_ptr(), dbias.data_ptr(), coef.data_ptr()
    b.act, b.act_after_bn = N.ACT['relu'], act_after
    b.drop_rate, b.mask, b.state, b.layer_id = rate, None, state.data_ptr(), lid
    b.rows, b.c, b.dtype = rows, c, ndt(dtype)
    b.workspace, b.workspace_bytes = ws.data_ptr(), wsb
    N.call('rvip_bn_bwd_reduce', C.byref(b), stream())
    N.call('rvip_bn_bwd_apply', C.byref(b), stream())
    g = g_in * keep / np.float32(1 - rate)
    if act_after:
        g = g * (yact > 0)
    dxr, dgr, dbr = O.bn_train_bwd(g, gamma.astype(np.float64), cache)
    dconv = dxr if act_after else dxr * (z64 > 0)
    tolr = 1e-2 if dtype != 'f32' else 1e-4
    np.testing.assert_allclose(down(dgamma), dgr, atol=tolr * np.abs(dgr).max())
    np.testing.assert_allclose(down(dbeta), dbr, atol=tolr * np.abs(dbr).max())
    close(down(dz), dconv, dtype, 'bn bwd dz')
    np.testing.assert_allclose(down(dbias), rnd(down(dz), dtype).astype(np.float64).sum((0, 1, 2)), rtol=1e-4, atol=1e-3)
    # no-BN path (up-conv / BATCH_NORMALISATION False): dconv = g * act'(z)
    b.gamma, b.coef, b.act_after_bn = None, None, 0
    N.call('rvip_bn_bwd_apply', C.byref(b), stream())
    z_act = np.maximum(z64, 0)
    close(down(dz), g_in * keep / np.float32(1 - rate) * (z64 > 0), dtype, 'act bwd (no BN)')
    # inference coefficients
    N.call('rvip_bn_infer_coeffs', P(gd), P(bd), P(mm), P(mv), 1e-3, c, P(scale), P(shift), stream())
    sc = gamma / np.sqrt(down(mv) + 1e-3)
    np.testing.assert_allclose(down(scale), sc, rtol=1e-5)
    np.testing.assert_allclose(down(shift), beta - down(mm) * sc, rtol=1e-4, atol=1e-6)
    assert z_act is not None


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', [(2, 12, 20, 16), (2, 16, 32, 32), (1, 6, 10, 8)], ids=['12x20', '16x32', '6x10'])
@pytest.mark.parametrize('rate', [0.0, 0.4])
def test_window_argmax_replaces_the_maxpool_backward_pass(dtype, shape, rate):
    """rvip_bn_apply(argmax) + rvip_bn_bwd_reduce / _apply(dpooled, argmax, skip gradient) must give, bit for bit, what
    rvip_maxpool2x2_bwd + the plain passes give: ties (post-ReLU zeros become one constant after BN) go to the first maximum in
    row-major window order; power-of-two extents take the shift path for the pixel coordinates, the others the division."""
    n, h, w, c = shape
    rows = n * h * w
    rng = np.random.default_rng(31)
    z = rnd(np.maximum(rng.standard_normal((n, h, w, c)) * 1.5 - 0.4, 0), dtype)          # ~60 % exact zeros
    gamma = (1 + 0.3 * rng.standard_normal(c)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(c)).astype(np.float32)
    L = N.lib()
    assert L.rvip_bn_apply_argmax_ok(c, ndt(dtype)) == 1
    wsb = L.rvip_reduce_workspace(rows, 16 * c)
    ws = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev())
    zd, gd, bd = up(z, dtype), f32(gamma), f32(beta)
    mm, mv = f32(np.zeros(c)), f32(np.ones(c))
    mean, invstd, scale, shift = (torch.empty(c, dtype=torch.float32, device=dev()) for _ in range(4))
    N.call('rvip_bn_train_stats', P(zd), C.c_longlong(rows), c, ndt(dtype), P(gd), P(bd), P(mm), P(mv), 0.99, 1e-3, 1,
           P(mean), P(invstd), P(scale), P(shift), P(ws), C.c_size_t(wsb), stream())
    state = torch.zeros(8, dtype=torch.int32, device=dev())
    state[N.STATE_SEED], state[N.STATE_STEP] = 77, 3
    oh, ow = h // 2, w // 2
    ve = 4 if dtype == 'f32' else 8
    y = torch.empty((n, h, w, c), dtype=tdt(dtype), device=dev())
    pooled = torch.empty((n, oh, ow, c), dtype=tdt(dtype), device=dev())
    arg = torch.full((n * oh * ow * (c // ve),), -1, dtype=torch.int16, device=dev())
    a = N.ApplyDesc()
    a.z, a.y, a.pooled, a.argmax = zd.data_ptr(), y.data_ptr(), pooled.data_ptr(), arg.data_ptr()
    a.scale, a.shift, a.act = scale.data_ptr(), shift.data_ptr(), 0
    a.drop_rate, a.mask, a.state, a.layer_id = rate, None, state.data_ptr(), 5
    a.n, a.h, a.w, a.c, a.dtype = n, h, w, c, ndt(dtype)
    N.call('rvip_bn_apply', C.byref(a), stream())
    # the argmax words against the oracle's first-maximum index of the stored y
    yq = down(y).astype(np.float64)
    _, idx = O.maxpool2x2_fwd(yq)                                       # [n, oh, ow, c] in 0..3
    words = arg.cpu().numpy().astype(np.uint16).reshape(n, oh, ow, c // ve)
    got = np.stack([(words >> (2 * e)) & 3 for e in range(ve)], -1).reshape(n, oh, ow, c)
    np.testing.assert_array_equal(got, idx)
    dp = rnd(rng.standard_normal((n, oh, ow, c)), dtype)
    addg = rnd(rng.standard_normal((n, h, w, c)), dtype)
    dpd, addd = up(dp, dtype), up(addg, dtype)

    def passes(fused, skip):
        dz = torch.full((n, h, w, c), 7.0, dtype=tdt(dtype), device=dev())
        dgamma, dbeta, dbias = (torch.empty(c, dtype=torch.float32, device=dev()) for _ in range(3))
        coef = torch.empty(3 * c, dtype=torch.float32, device=dev())
        b = N.BnBwdDesc()
        b.z, b.dz = zd.data_ptr(), dz.data_ptr()
        if fused:
            b.dy = addd.data_ptr() if skip else None
            b.dpooled, b.argmax, b.h, b.w = dpd.data_ptr(), arg.data_ptr(), h, w
        else:
            gy = torch.zeros((n, h, w, c), dtype=tdt(dtype), device=dev())
            N.call('rvip_maxpool2x2_bwd', P(y), P(dpd), P(addd) if skip else None, P(gy), n, h, w, c, ndt(dtype), stream())
            b.dy = gy.data_ptr()
            b._gy = gy
        b.gamma, b.mean, b.invstd = gd.data_ptr(), mean.data_ptr(), invstd.data_ptr()
        b.scale, b.shift = scale.data_ptr(), shift.data_ptr()
        b.dgamma, b.dbeta, b.dbias, b.coef = dgamma.data_ptr(), dbeta.data_ptr(), dbias.data_ptr(), coef.data_ptr()
        b.act, b.act_after_bn = N.ACT['relu'], 0
        b.drop_rate, b.mask, b.state, b.layer_id = rate, None, state.data_ptr(), 5
        b.rows, b.c, b.dtype = rows, c, ndt(dtype)
        b.workspace, b.workspace_bytes = ws.data_ptr(), wsb
        N.call('rvip_bn_bwd_reduce', C.byref(b), stream())
        N.call('rvip_bn_bwd_apply', C.byref(b), stream())
        torch.cuda.synchronize()
        return [t.clone() for t in (dz, dgamma, dbeta, dbias)]
    for skip in (True, False):
        for r0, r1 in zip(passes(False, skip), passes(True, skip)):
            assert torch.equal(r0, r1), (skip, (r0.float() - r1.float()).abs().max())


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('loss', ['mse', 'bce_dice'])
@pytest.mark.parametrize('k', [2, 4])               # 4: the loss sums [1]..[4] and BCE-Dice drop the background channel (Loss_and_metrics.py:240-242)
def test_head_loss_and_backward(dtype, loss, k):
    n, h, w, cin = 3, 16, 24, 16
    rows = n * h * w
    rng = np.random.default_rng(6)
    x = rnd(rng.standard_normal((n, h, w, cin)), dtype)
    wt = (rng.standard_normal((1, 1, cin, k)) * 0.4).astype(np.float32)
    b = (rng.standard_normal(k) * 0.1).astype(np.float32)
    _, yt = O.synthetic_batch(n, (h, w), k, seed=2)
    L = N.lib()
    wsb = L.rvip_reduce_workspace(rows, 8 * cin)
    ws = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev())
    xd, wd_, bd, ytd = up(x, dtype), f32(wt), f32(b), f32(yt)
    pred = torch.empty((n, h, w, k), dtype=torch.float32, device=dev())
    sums = torch.zeros(16, dtype=torch.float32, device=dev())
    N.call('rvip_head_fwd', P(xd), P(wd_), P(bd), P(pred), P(ytd), P(sums), C.c_longlong(rows), cin, k, ndt(dtype), P(ws),
           C.c_size_t(wsb), stream())
    logits = O.conv2d_same_fwd(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    pref = O.act_fwd(logits, 'sigmoid')
    np.testing.assert_allclose(down(pred), pref, atol=2e-6)
    s = down(sums).astype(np.float64)
    t64 = yt.astype(np.float64)
    np.testing.assert_allclose(s[0], ((pref - t64) ** 2).sum(), rtol=1e-4)
    np.testing.assert_allclose(s[2:5], [(t64[..., -3:] * pref[..., -3:]).sum(), t64[..., -3:].sum(), pref[..., -3:].sum()], rtol=1e-4)
    np.testing.assert_allclose(s[8:11], [(t64[..., -1] * pref[..., -1]).sum(), t64[..., -1].sum(), pref[..., -1].sum()], rtol=1e-4)
    world = 2.0                                                         # pretend 2 ranks: global-batch scaling
    dlogit = torch.empty((n, h, w, k), dtype=torch.float32, device=dev())
    lossd = torch.zeros(1, dtype=torch.float32, device=dev())
    kind = N.LOSS_MSE if loss == 'mse' else N.LOSS_BCE_DICE
    N.call('rvip_head_grad', P(pred), P(ytd), P(sums), P(dlogit), P(lossd), C.c_longlong(rows), k, kind,
           1.0 / (rows * (k if loss == 'mse' else min(k, 3)) * world), 1.0 / world, 0.5, 1.0, stream())
    if loss == 'mse':
        lv, dpred = O.mse_loss(t64, pref, global_batch=int(n * world))
        dl_ref = dpred * pref * (1 - pref)
    else:
        lv, dl_ref = O.bce_dice_loss(t64, pref, logits=logits, global_batch=int(n * world))
    np.testing.assert_allclose(float(down(lossd)[0]), lv, rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(down(dlogit), dl_ref, atol=1e-4 * np.abs(dl_ref).max())
    if loss != 'mse' and k == 4:
        assert not down(dlogit)[..., 0].any()
    dx = torch.empty((n, h, w, cin), dtype=tdt(dtype), device=dev())
    dw = torch.empty((1, 1, cin, k), dtype=torch.float32, device=dev())
    db = torch.empty(k, dtype=torch.float32, device=dev())
    N.call('rvip_head_bwd', P(xd), P(wd_), P(dlogit), P(dx), P(dw), P(db), C.c_longlong(rows), cin, k, ndt(dtype), P(ws),
           C.c_size_t(wsb), stream())
    dl = down(dlogit).astype(np.float64)
    rdx, rdw, rdb = O.conv2d_same_bwd(x.astype(np.float64), wt.astype(np.float64), dl)
    close(down(dx), rdx, dtype, 'head dx')
    np.testing.assert_allclose(down(dw), rdw, atol=1e-4 * np.abs(rdw).max())
    np.testing.assert_allclose(down(db), rdb, atol=1e-4 * np.abs(rdb).max())
    # landmarks: argmax bit-exact incl. ties, > 0.5 mask
    pr = down(pred)
    pr[0, 3, 5, 0] = pr[0, 9, 2, 0] = 2.0                               # tie -> first in row-major order
    prd = f32(pr)
    idx = torch.zeros((n, k), dtype=torch.int64, device=dev())
    mask = torch.zeros((n, h, w, k), dtype=torch.uint8, device=dev())
    N.call('rvip_landmarks', P(prd), P(idx), P(mask), n, h * w, k, 0.5, stream())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(idx.cpu().numpy(), O.landmark_argmax(pr))
    np.testing.assert_array_equal(mask.cpu().numpy().astype(bool), O.threshold_mask(pr))


def test_pack_all_with_step_tick_is_pack_all_plus_state_tick():
    rng = np.random.default_rng(8)
    w = rng.standard_normal((3, 3, 16, 24)).astype(np.float32)
    wf0, wd0 = pack_all(w, 'bf16')
    wm = f32(w)
    wf, wd = torch.empty_like(wf0), torch.empty_like(wd0)
    tab = (N.PackEntry * 1)()
    tab[0].w_off, tab[0].f_off, tab[0].d_off, tab[0].cin, tab[0].cout, tab[0].taps = 0, 0, 0, 16, 24, 9
    tabd = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev())
    state = torch.zeros(N.STATE_WORDS, dtype=torch.int32, device=dev())
    state[N.STATE_STEP] = 41
    for _ in range(2):
        N.call('rvip_pack_all_conv3x3_weights_tick', P(wm), P(tabd), 1, 9 * 16 * 24, N.BF16, P(wf), P(wd), P(state), stream())
    torch.cuda.synchronize()
    assert int(state[N.STATE_STEP]) == 43 and torch.equal(wf, wf0) and torch.equal(wd, wd0)
    assert N.lib().rvip_pack_all_conv3x3_weights_tick(P(wm), P(tabd), 1, 9 * 16 * 24, N.BF16, P(wf), P(wd), None, stream()) == -1


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', [(16, 24), (64, 32), (72, 40), (256, 128), (8, 8)])
def test_pack_all_mode_1_writes_both_phase_kernel_sets(shape, dtype):
    """A mode-1 entry of the pack table (an UpSampling2D -> conv layer): the forward phase kernels at f_off and the data-gradient phase
    kernels at d_off, bit for bit what rvip_pack_subpixel_weights / rvip_pack_subpixel_dgrad_weights write; a mode-0 entry beside it is
    untouched by it."""
    ci, co = shape
    rng = np.random.default_rng(ci + co)
    w = rng.standard_normal((3, 3, ci, co)).astype(np.float32)
    w2 = rng.standard_normal((3, 3, 8, 16)).astype(np.float32)
    theta = f32(np.concatenate([w.reshape(-1), w2.reshape(-1)]))
    k1, k0 = 16 * ci * co, 9 * 8 * 16
    wf = torch.full((k1 + k0 + 64,), 3.0, dtype=tdt(dtype), device=dev())
    wd = torch.full((k1 + k0 + 64,), 5.0, dtype=tdt(dtype), device=dev())
    tab = (N.PackEntry * 2)()
    tab[0].w_off, tab[0].f_off, tab[0].d_off, tab[0].cin, tab[0].cout, tab[0].taps, tab[0].mode = 0, 32, 32, ci, co, 9, 1
    tab[1].w_off, tab[1].f_off, tab[1].d_off, tab[1].cin, tab[1].cout, tab[1].taps, tab[1].mode = 9 * ci * co, 32 + k1, 32 + k1, 8, 16, 9, 0
    tabd = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev())
    N.call('rvip_pack_all_conv3x3_weights', P(theta), P(tabd), 2, max(k1, k0), ndt(dtype), P(wf), P(wd), stream())
    rf = torch.empty(k1, dtype=tdt(dtype), device=dev())
    rd = torch.empty(k1, dtype=tdt(dtype), device=dev())
    wm = f32(w)
    N.call('rvip_pack_subpixel_weights', P(wm), ci, co, ndt(dtype), P(rf), stream())
    N.call('rvip_pack_subpixel_dgrad_weights', P(wm), ci, co, ndt(dtype), P(rd), stream())
    f0, d0 = pack(w2, dtype)
    torch.cuda.synchronize()
    assert torch.equal(wf[32:32 + k1], rf) and torch.equal(wd[32:32 + k1], rd)
    assert torch.equal(wf[32 + k1:32 + k1 + k0], f0) and torch.equal(wd[32 + k1:32 + k1 + k0], d0)
    assert bool((wf[:32] == 3.0).all()) and bool((wd[:32] == 5.0).all()) and bool((wf[32 + k1 + k0:] == 3.0).all()) and bool((wd[32 + k1 + k0:] == 5.0).all())


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', [(8, 6), (12, 10), (40, 70), (7, 3)])
@pytest.mark.parametrize('mode', [0, 1])
def test_pack_all_stays_inside_its_rows_for_any_cout(shape, dtype, mode):
    """both modes moved four output channels at a time; with Cout % 4 != 0 the last group of a row ran into
    the next row.  rvip_pack_table_check (host side) refuses such a table, and the kernel takes an element path for it: the result
    is the NumPy re-layout bit for bit and the canaries around the two outputs are intact."""
    ci, co = shape
    rng = np.random.default_rng(100 * ci + co)
    w = rng.standard_normal((3, 3, ci, co)).astype(np.float32)
    k = (16 if mode else 9) * ci * co
    guard = 256
    wf = torch.full((k + 2 * guard,), 3.0, dtype=tdt(dtype), device=dev())
    wd = torch.full((k + 2 * guard,), 5.0, dtype=tdt(dtype), device=dev())
    tab = (N.PackEntry * 1)()
    tab[0].w_off, tab[0].f_off, tab[0].d_off, tab[0].cin, tab[0].cout, tab[0].taps, tab[0].mode = 0, guard, guard, ci, co, 9, mode
    assert N.lib().rvip_pack_table_check(C.cast(tab, C.c_void_p), 1, ndt(dtype)) == (-1 if co % 4 else 0)
    tabd = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev())
    theta = f32(np.concatenate([w.reshape(-1), np.full(64, 1e30, np.float32)]))       # what lies behind the master must never be read into the result
    N.call('rvip_pack_all_conv3x3_weights', P(theta), P(tabd), 1, k, ndt(dtype), P(wf), P(wd), stream())
    torch.cuda.synchronize()
    for buf, fill in ((wf, 3.0), (wd, 5.0)):
        assert bool((buf[:guard] == fill).all()) and bool((buf[guard + k:] == fill).all()), 'canary overwritten'
    gf, gd = wf[guard:guard + k].float().cpu().numpy(), wd[guard:guard + k].float().cpu().numpy()
    if mode == 0:
        rf = rnd(np.transpose(w.reshape(9, ci, co), (0, 2, 1)), dtype)                # [t][co][ci]
        rd = rnd(w.reshape(9, ci, co)[::-1], dtype)                                   # [8 - t][ci][co]
        assert np.array_equal(gf.reshape(9, co, ci), rf) and np.array_equal(gd.reshape(9, ci, co), rd)
    else:
        rf_t = torch.empty(k, dtype=tdt(dtype), device=dev())
        rd_t = torch.empty(k, dtype=tdt(dtype), device=dev())
        wm = f32(w)
        N.call('rvip_pack_subpixel_weights', P(wm), ci, co, ndt(dtype), P(rf_t), stream())          # the element-per-thread kernels
        N.call('rvip_pack_subpixel_dgrad_weights', P(wm), ci, co, ndt(dtype), P(rd_t), stream())
        torch.cuda.synchronize()
        assert np.array_equal(gf, rf_t.float().cpu().numpy()) and np.array_equal(gd, rd_t.float().cpu().numpy())


def test_adam_state_and_convert():
    rng = np.random.default_rng(7)
    cnt = 10007
    th, g = rng.standard_normal(cnt).astype(np.float32), rng.standard_normal(cnt).astype(np.float32)
    m, v = np.zeros(cnt, np.float32), np.zeros(cnt, np.float32)
    thd, gd, md, vd = f32(th), f32(g), f32(m), f32(v)
    state = torch.zeros(8, dtype=torch.int32, device=dev())
    state.view(torch.float32)[N.STATE_LR] = 1e-3
    rt, rm, rv = th.astype(np.float64), m.astype(np.float64), v.astype(np.float64)
    for t in range(1, 4):
        N.call('rvip_adam_step', P(thd), P(gd), P(md), P(vd), C.c_longlong(cnt), 0.9, 0.999, 1e-7, 1.0, P(state), stream())
        N.call('rvip_state_tick', P(state), stream())
        rt, rm, rv = O.adam_step(rt, g.astype(np.float64), rm, rv, t, 1e-3)
    assert int(state[0].item()) == 3
    np.testing.assert_allclose(down(thd), rt, atol=2e-6)
    # (1 - beta2) is formed in fp32 on the device (as in TF): 1.f - 0.999f differs from 1e-3 by 1.3e-5 relative
    np.testing.assert_allclose(down(vd), rv, rtol=5e-5, atol=1e-9)
    a = rng.standard_normal(1000).astype(np.float32)
    bd = torch.empty(1000, dtype=torch.bfloat16, device=dev())
    ad = f32(a)
    N.call('rvip_convert', P(ad), N.F32, P(bd), N.BF16, C.c_longlong(1000), stream())
    torch.cuda.synchronize()
    assert torch.equal(bd.cpu(), torch.from_numpy(a).to(torch.bfloat16))     # RNE like torch


def test_bad_arguments_are_refused_not_launched():
    L = N.lib()
    x = torch.zeros(64, dtype=torch.float32, device=dev())
    d = conv_desc(x, 6, 0, None, 0, x, None, x, None, 0, 1, 4, 4, 8, 0, 'f32')       # c0 % 4 != 0
    assert L.rvip_conv3x3_fwd(C.byref(d), stream()) == -1
    assert L.rvip_conv3x3_fwd(None, stream()) == -1
    g = N.Wgrad3x3Desc()
    g.x0, g.c0, g.dy, g.dw = x.data_ptr(), 8, x.data_ptr(), x.data_ptr()
    g.n, g.h, g.w, g.cout, g.dtype = 1, 4, 4, 8, N.F32
    g.workspace, g.workspace_bytes = x.data_ptr(), 16
    assert L.rvip_conv3x3_wgrad(C.byref(g), stream()) == -3                          # workspace too small


@pytest.mark.parametrize('cc', [0, 1])
def test_postprocess_flat_labels_cc_filter_and_points(cc):
    """rvip_postprocess vs the restatement of predict_model.py:149-156, Postprocess.py:108-120 (largest 4-connected
    component, first on ties) and evaluate_cv.py:418-442 (mean y/x per label): labels bit-exact, points to fp32."""
    rng = np.random.default_rng(7)
    n, h, w, k = 5, 40, 56, 2
    pred = (rng.random((n, h, w, k)) * 0.3).astype(np.float32)
    for i in range(n):                                   # a few blobs per channel, some touching only diagonally, one snake
        for c in range(k):
            for _ in range(3 + i % 2):
                y0, x0, r = int(rng.integers(2, h - 8)), int(rng.integers(2, w - 8)), int(rng.integers(1, 5))
                pred[i, y0:y0 + r, x0:x0 + r, c] = 0.9
    pred[0, 10:12, 10:12, 0] = 0.9; pred[0, 12:14, 12:14, 0] = 0.9          # diagonal neighbours: separate under 4-connectivity
    pred[1, 5, 3:50, 1] = 0.8; pred[1, 5:30, 49, 1] = 0.8; pred[1, 29, 3:50, 1] = 0.8   # long snake (many sweeps)
    pred[2, :, :, 1] = 0.1                               # label 2 absent on slice 2
    pred[3, 20:24, 20:24, 0] = 0.9; pred[3, 20:24, 30:34, 0] = 0.9          # equal sizes: the first in raster order wins
    pred[4, :, :, 0] = np.maximum(pred[4, :, :, 0], 0.6)  # no background on slice 4: np.unique(s)[1:] then skips label 1 (reference quirk)
    pd = f32(pred)
    flat = torch.empty((n, h, w), dtype=torch.uint8, device=dev())
    pts = torch.empty((n, k, 2), dtype=torch.float32, device=dev())
    sizes = torch.empty((n, k), dtype=torch.int32, device=dev())
    L = N.lib()
    wsb = L.rvip_postprocess_workspace(n, h, w, k)
    ws = torch.empty(wsb // 4 + 16, dtype=torch.int32, device=dev())
    N.call('rvip_postprocess', P(pd), P(flat), P(pts), P(sizes), n, h, w, k, 0.5, cc, P(ws), C.c_size_t(wsb), stream())
    torch.cuda.synchronize()
    ref = O.flat_labels(pred)
    if cc:
        ref = O.clean_2d_cc(ref)
    np.testing.assert_array_equal(flat.cpu().numpy(), ref)
    rp = O.mean_rvip_points(ref, k)
    got = pts.cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(rp)) and np.isnan(rp[2, 1]).all() and np.isnan(rp[4, 0]).all()
    np.testing.assert_allclose(got[~np.isnan(rp)], rp[~np.isnan(rp)], rtol=1e-6)
    np.testing.assert_array_equal(sizes.cpu().numpy(), np.stack([(ref == v + 1).sum((1, 2)) for v in range(k)], 1))


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('act_after', [0, 1])
def test_last_stage_fused_with_head_matches_the_separate_kernels(dtype, act_after):
    """rvip_bn_apply_head / rvip_bn_bwd_reduce_head / rvip_bn_bwd_apply_head (the stage's BN output and its gradient
    live in registers only) against rvip_bn_apply + rvip_head_fwd and rvip_head_bwd + rvip_bn_bwd_reduce/apply."""
    n, h, w, c, k = 3, 12, 20, 16, 2
    rows = n * h * w
    rng = np.random.default_rng(11)
    pre = rng.standard_normal((n, h, w, c)) * 1.5 + 0.2
    z = rnd(pre if act_after else np.maximum(pre, 0), dtype)
    gamma = (1 + 0.3 * rng.standard_normal(c)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(c)).astype(np.float32)
    hw = (rng.standard_normal((c, k)) * 0.3).astype(np.float32)
    hb = (rng.standard_normal(k) * 0.1).astype(np.float32)
    _, yt = O.synthetic_batch(n, (h, w), k, seed=3)
    L = N.lib()
    wsb = L.rvip_reduce_workspace(rows, 16 * c)
    ws = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev())
    zd, gd, bd, hwd, hbd, ytd = up(z, dtype), f32(gamma), f32(beta), f32(hw), f32(hb), f32(yt)
    mm, mv = f32(np.zeros(c)), f32(np.ones(c))
    mean, invstd, scale, shift = (torch.empty(c, dtype=torch.float32, device=dev()) for _ in range(4))
    N.call('rvip_bn_train_stats', P(zd), C.c_longlong(rows), c, ndt(dtype), P(gd), P(bd), P(mm), P(mv), 0.99, 1e-3, 1,
           P(mean), P(invstd), P(scale), P(shift), P(ws), C.c_size_t(wsb), stream())
    state = torch.zeros(8, dtype=torch.int32, device=dev())
    y = torch.empty((n, h, w, c), dtype=tdt(dtype), device=dev())
    a = N.ApplyDesc()
    a.z, a.y, a.pooled = zd.data_ptr(), y.data_ptr(), None
    a.scale, a.shift, a.act = scale.data_ptr(), shift.data_ptr(), (N.ACT['relu'] if act_after else 0)
    a.drop_rate, a.mask, a.state, a.layer_id = 0.0, None, state.data_ptr(), 0
    a.n, a.h, a.w, a.c, a.dtype = n, h, w, c, ndt(dtype)
    # separate kernels
    N.call('rvip_bn_apply', C.byref(a), stream())
    pred0 = torch.empty((n, h, w, k), dtype=torch.float32, device=dev())
    sums0 = torch.zeros(16, dtype=torch.float32, device=dev())
    N.call('rvip_head_fwd', P(y), P(hwd), P(hbd), P(pred0), P(ytd), P(sums0), C.c_longlong(rows), c, k, ndt(dtype), P(ws), C.c_size_t(wsb), stream())
    dl = (rng.standard_normal((n, h, w, k)) * 1e-3).astype(np.float32)
    dld = f32(dl)
    gy = torch.empty((n, h, w, c), dtype=tdt(dtype), device=dev())
    hdw0, hdb0 = torch.empty((c, k), dtype=torch.float32, device=dev()), torch.empty(k, dtype=torch.float32, device=dev())
    N.call('rvip_head_bwd', P(y), P(hwd), P(dld), P(gy), P(hdw0), P(hdb0), C.c_longlong(rows), c, k, ndt(dtype), P(ws), C.c_size_t(wsb), stream())

    def bwd_desc(dz, dgamma, dbeta, dbias, coef, dy):
        b = N.BnBwdDesc()
        b.dy, b.z, b.dz = (dy.data_ptr() if dy is not None else None), zd.data_ptr(), dz.data_ptr()
        b.gamma, b.mean, b.invstd = gd.data_ptr(), mean.data_ptr(), invstd.data_ptr()
        b.scale, b.shift = scale.data_ptr(), shift.data_ptr()
        b.dgamma, b.dbeta, b.dbias, b.coef = dgamma.data_ptr(), dbeta.data_ptr(), dbias.data_ptr(), coef.data_ptr()
        b.act, b.act_after_bn = N.ACT['relu'], act_after
        b.drop_rate, b.mask, b.state, b.layer_id = 0.0, None, state.data_ptr(), 0
        b.rows, b.c, b.dtype = rows, c, ndt(dtype)
        b.workspace, b.workspace_bytes = ws.data_ptr(), wsb
        return b
    out0 = [torch.empty((n, h, w, c), dtype=tdt(dtype), device=dev())] + [torch.empty(c, dtype=torch.float32, device=dev()) for _ in range(3)] + \
           [torch.empty(3 * c, dtype=torch.float32, device=dev())]
    b0 = bwd_desc(*out0, gy)
    N.call('rvip_bn_bwd_reduce', C.byref(b0), stream())
    N.call('rvip_bn_bwd_apply', C.byref(b0), stream())
    # fused
    pred1 = torch.empty_like(pred0)
    sums1 = torch.zeros(16, dtype=torch.float32, device=dev())
    a.y = None
    N.call('rvip_bn_apply_head', C.byref(a), P(hwd), P(hbd), k, P(pred1), P(ytd), P(sums1), P(ws), C.c_size_t(wsb), stream())
    out1 = [torch.empty((n, h, w, c), dtype=tdt(dtype), device=dev())] + [torch.empty(c, dtype=torch.float32, device=dev()) for _ in range(3)] + \
           [torch.empty(3 * c, dtype=torch.float32, device=dev())]
    hdw1, hdb1 = torch.empty_like(hdw0), torch.empty_like(hdb0)
    b1 = bwd_desc(*out1, None)
    N.call('rvip_bn_bwd_reduce_head', C.byref(b1), P(hwd), P(dld), k, P(hdw1), P(hdb1), stream())
    N.call('rvip_bn_bwd_apply_head', C.byref(b1), P(hwd), P(dld), k, stream())
    torch.cuda.synchronize()
    np.testing.assert_allclose(down(pred1), down(pred0), atol=2e-6)
    np.testing.assert_allclose(down(sums1)[:11], down(sums0)[:11], rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(down(hdw1), down(hdw0), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(down(hdb1), down(hdb0), rtol=1e-4, atol=1e-7)
    for t1, t0, nm in zip(out1[1:], out0[1:], ('dgamma', 'dbeta', 'dbias', 'coef')):
        sc_ = float(np.abs(down(t0)).max())
        np.testing.assert_allclose(down(t1), down(t0), atol=(2e-3 if dtype != 'f32' else 1e-5) * sc_ + 1e-9, err_msg=nm)
    close(down(out1[0]), down(out0[0]).astype(np.float64), dtype, 'dz fused vs separate')


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', [(3, 12, 20, 16, 2, 1.0), (2, 64, 96, 32, 2, 4096.0), (2, 40, 24, 8, 1, 1.0), (1, 64, 64, 64, 2, 1.0)])
def test_mse_head_gradients_from_the_forward_pass(dtype, shape):
    """rvip_bn_apply_head_mse + rvip_head_mse_coef (ABI 6) against rvip_bn_apply_head + rvip_head_grad (+ rvip_scale_f32) +
    rvip_bn_bwd_reduce_head: same pred / loss sums / dlogit bit for bit, the head's and the stage's BN gradients up to the
    summation order -- and up to the storage rounding of the rebuilt gradient, which the algebraic sums do not carry; the
    in-kernel exact route (every block declared ill-conditioned) carries it and is held to the tight bound."""
    n, h, w, c, k, dscale = shape
    rows = n * h * w
    rng = np.random.default_rng(17 + c)
    z = rnd(np.maximum(rng.standard_normal((n, h, w, c)) * 1.5 + 0.2, 0), dtype)
    gamma = (1 + 0.3 * rng.standard_normal(c)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(c)).astype(np.float32)
    hw = (rng.standard_normal((c, k)) * 0.3).astype(np.float32)
    hb = (rng.standard_normal(k) * 0.1).astype(np.float32)
    _, yt = O.synthetic_batch(n, (h, w), k, seed=3)
    L = N.lib()
    wsb = L.rvip_reduce_workspace(rows, 16 * c)
    ws = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev())
    zd, gd, bd, hwd, hbd, ytd = up(z, dtype), f32(gamma), f32(beta), f32(hw), f32(hb), f32(yt)
    mm, mv = f32(np.zeros(c)), f32(np.ones(c))
    mean, invstd, scale, shift = (torch.empty(c, dtype=torch.float32, device=dev()) for _ in range(4))
    N.call('rvip_bn_train_stats', P(zd), C.c_longlong(rows), c, ndt(dtype), P(gd), P(bd), P(mm), P(mv), 0.99, 1e-3, 1,
           P(mean), P(invstd), P(scale), P(shift), P(ws), C.c_size_t(wsb), stream())
    state = torch.zeros(8, dtype=torch.int32, device=dev())
    a = N.ApplyDesc()
    a.z, a.y, a.pooled = zd.data_ptr(), None, None
    a.scale, a.shift, a.act = scale.data_ptr(), shift.data_ptr(), 0
    a.drop_rate, a.mask, a.state, a.layer_id = 0.0, None, state.data_ptr(), 0
    a.n, a.h, a.w, a.c, a.dtype = n, h, w, c, ndt(dtype)
    inv_count = 1.0 / (rows * k)

    def outputs():
        o = {'dz': torch.empty((n, h, w, c), dtype=tdt(dtype), device=dev())}
        for nm in ('dgamma', 'dbeta', 'dbias'):
            o[nm] = torch.full((c,), 7.0, dtype=torch.float32, device=dev())
        o['coef'] = torch.full((3 * c,), 7.0, dtype=torch.float32, device=dev())
        o['hdw'] = torch.full((c, k), 7.0, dtype=torch.float32, device=dev())
        o['hdb'] = torch.full((k,), 7.0, dtype=torch.float32, device=dev())
        o['pred'] = torch.empty((n, h, w, k), dtype=torch.float32, device=dev())
        o['sums'] = torch.zeros(16, dtype=torch.float32, device=dev())
        o['dlogit'] = torch.full((n, h, w, k), 7.0, dtype=torch.float32, device=dev())
        o['loss'] = torch.full((1,), 7.0, dtype=torch.float32, device=dev())
        return o

    def bwd_desc(o):
        b = N.BnBwdDesc()
        b.dy, b.z, b.dz = None, zd.data_ptr(), o['dz'].data_ptr()
        b.gamma, b.mean, b.invstd = gd.data_ptr(), mean.data_ptr(), invstd.data_ptr()
        b.scale, b.shift = scale.data_ptr(), shift.data_ptr()
        b.dgamma, b.dbeta, b.dbias, b.coef = o['dgamma'].data_ptr(), o['dbeta'].data_ptr(), o['dbias'].data_ptr(), o['coef'].data_ptr()
        b.act, b.act_after_bn = N.ACT['relu'], 0
        b.drop_rate, b.mask, b.state, b.layer_id = 0.0, None, state.data_ptr(), 0
        b.rows, b.c, b.dtype = rows, c, ndt(dtype)
        b.workspace, b.workspace_bytes = ws.data_ptr(), wsb
        return b
    # classic
    o0 = outputs()
    N.call('rvip_bn_apply_head', C.byref(a), P(hwd), P(hbd), k, P(o0['pred']), P(ytd), P(o0['sums']), P(ws), C.c_size_t(wsb), stream())
    N.call('rvip_head_grad', P(o0['pred']), P(ytd), P(o0['sums']), P(o0['dlogit']), P(o0['loss']), C.c_longlong(rows), k, N.LOSS_MSE,
           C.c_float(inv_count), C.c_float(1.0), C.c_float(0.5), C.c_float(1.0), stream())
    if dscale != 1.0:
        N.call('rvip_scale_f32', P(o0['dlogit']), C.c_longlong(rows * k), C.c_float(dscale), stream())
    b0 = bwd_desc(o0)
    N.call('rvip_bn_bwd_reduce_head', C.byref(b0), P(hwd), P(o0['dlogit']), k, P(o0['hdw']), P(o0['hdb']), stream())
    # forward-side
    nr = L.rvip_bn_apply_head_mse_rows(C.c_longlong(rows), c, ndt(dtype), k)
    chunk = -(-rows // min(1024, -(-rows // 1024)))
    assert nr == -(-rows // chunk)

    def forward_side(min_gamma):
        o = outputs()
        mrows = torch.full((nr * 3 * c,), 7.0, dtype=torch.float32, device=dev())
        N.call('rvip_bn_apply_head_mse', C.byref(a), P(hwd), P(hbd), P(bd), k, P(o['pred']), P(ytd), P(o['sums']), P(o['dlogit']),
               C.c_float(inv_count), C.c_float(dscale), P(mrows), C.c_size_t(mrows.numel() * 4), P(ws), C.c_size_t(wsb), stream())
        b = bwd_desc(o)
        hc = N.HeadCoefDesc()
        hc.bn, hc.beta = C.pointer(b), bd.data_ptr()
        hc.head_w, hc.dlogit, hc.k, hc.nrows = hwd.data_ptr(), o['dlogit'].data_ptr(), k, nr
        hc.mse_rows, hc.head_dw, hc.head_db = mrows.data_ptr(), o['hdw'].data_ptr(), o['hdb'].data_ptr()
        hc.sums, hc.loss_out, hc.inv_count = o['sums'].data_ptr(), o['loss'].data_ptr(), inv_count
        o['flags'] = torch.full((-(-c // 32),), 7, dtype=torch.int32, device=dev())
        hc.flags, hc.min_gamma, hc.max_beta_ratio = o['flags'].data_ptr(), min_gamma, 64.0
        N.call('rvip_head_mse_coef', C.byref(hc), stream())
        torch.cuda.synchronize()
        return o, (a, b, hc, mrows)
    # The BN terms of the algebraic route are held to the float64 straight-through values (g = sum_k d w unrounded, xhat from z):
    # it sums g * y with y as the head read it (rounded to the storage type), the classic route sums round(g) * xhat -- two
    # different zero-mean roundings per pixel, whose share of a sum shrinks with sqrt(rows).  The exact route of an ill-conditioned
    # block repeats the classic arithmetic and is held to the classic launch.
    dl64, z64 = down(o0['dlogit']).astype(np.float64).reshape(rows, k), down(zd).astype(np.float64).reshape(rows, c)
    g64 = dl64 @ hw.astype(np.float64).T
    xh = (z64 - down(mean).astype(np.float64)) * down(invstd).astype(np.float64)
    want_db, want_dg = g64.sum(0), (g64 * xh).sum(0)
    gm_, is_, mu_ = gamma.astype(np.float64), down(invstd).astype(np.float64), down(mean).astype(np.float64)
    c2 = -gm_ * is_ * is_ * want_dg / rows
    exact = {'dbeta': want_db, 'dgamma': want_dg, 'coef': np.concatenate([gm_ * is_, c2, -gm_ * is_ * want_db / rows - c2 * mu_])}
    classic = {nm: down(o0[nm]).astype(np.float64) for nm in exact}
    loose = {'f32': 2e-5, 'bf16': 8e-3, 'f16': 1.2e-3}[dtype] * max(1.0, (12288.0 / rows) ** 0.5)
    for min_gamma, want, tol in ((1.0 / 64, exact, loose), (1e9, classic, 2e-5)):
        o1, keep = forward_side(min_gamma)
        assert torch.equal(o1['pred'], o0['pred']) and torch.equal(o1['sums'], o0['sums'])
        assert torch.equal(o1['dlogit'], o0['dlogit'])
        assert torch.equal(o1['loss'], o0['loss'])
        assert bool(down(o1['flags']).all()) == (min_gamma > 1) and bool(down(o1['flags']).any()) == (min_gamma > 1)
        for nm in ('hdw', 'hdb'):
            sc_ = float(np.abs(down(o0[nm])).max())
            np.testing.assert_allclose(down(o1[nm]), down(o0[nm]), atol=2e-5 * sc_ + 1e-12, err_msg=nm)
        for nm in ('dgamma', 'dbeta', 'coef'):
            got, ref = down(o1[nm]).reshape(-1, c).astype(np.float64), want[nm].reshape(-1, c)
            for i in range(got.shape[0]):
                assert np.abs(got[i] - ref[i]).max() <= tol * np.abs(ref[i]).max() + 1e-12, (nm, i, min_gamma, np.abs(got[i] - ref[i]).max(), np.abs(ref[i]).max())
    for nm in exact:                           # (and the classic launch sits within the same distance of the float64 values)
        got, ref = classic[nm].reshape(-1, c), exact[nm].reshape(-1, c)
        for i in range(got.shape[0]):
            assert np.abs(got[i] - ref[i]).max() <= loose * np.abs(ref[i]).max() + 1e-12, (nm, i, 'classic')
    # what the entry points refuse
    a2, b2, hc2, mrows2 = keep
    args = lambda kk, nbytes: (C.byref(a2), P(hwd), P(hbd), P(bd), kk, P(o1['pred']), P(ytd), P(o1['sums']), P(o1['dlogit']), C.c_float(inv_count),  # noqa: E731
                               C.c_float(1.0), P(mrows2), C.c_size_t(nbytes), P(ws), C.c_size_t(wsb), stream())
    assert L.rvip_bn_apply_head_mse(*args(3, mrows2.numel() * 4)) == -2
    assert L.rvip_bn_apply_head_mse(*args(k, mrows2.numel() * 4 - 4)) == -3
    a2.act = N.ACT['relu']
    assert L.rvip_bn_apply_head_mse(*args(k, mrows2.numel() * 4)) == -2
    a2.act = 0
    b2.act_after_bn = 1
    assert L.rvip_head_mse_coef(C.byref(hc2), stream()) == -2
    b2.act_after_bn = 0
    hc2.min_gamma = 0.0
    assert L.rvip_head_mse_coef(C.byref(hc2), stream()) == -1
    assert L.rvip_bn_apply_head_mse_rows(C.c_longlong(rows), c, ndt(dtype), 3) == 0


@pytest.mark.parametrize('dtype', ['bf16', 'f16'])
@pytest.mark.parametrize('shape', [(3, 12, 20, 16, 2, 1.0, 1.0), (2, 64, 96, 32, 2, 4096.0, 0.5), (2, 40, 24, 8, 1, 1.0, 1.0), (1, 64, 64, 64, 2, 2.0, 1.0)])
def test_bce_dice_head_gradients_from_the_forward_pass(dtype, shape):
    """The BCE-Dice form of the fused last stage (round 4): rvip_bn_apply_head_bcedice (three row sets, no logit gradient written) +
    rvip_head_mse_coef(loss_kind BCE_DICE) + rvip_bn_bwd_apply_head_lazy (the gradient rebuilt per pixel from the heat-map, the target and
    three coefficients) against the classic launches rvip_bn_apply_head + rvip_head_grad + rvip_scale_f32 + rvip_bn_bwd_reduce_head +
    rvip_bn_bwd_apply_head (Loss_and_metrics.py:229-245 as restated there), on the same inputs: pred / loss sums bit for bit, the loss value
    and every gradient to summation order; the BN terms of the algebraic route are held to the float64 straight-through values like the
    MSE form's, the in-kernel exact route to the classic launch."""
    n, h, w, c, k, dscale, lg = shape
    rows = n * h * w
    rng = np.random.default_rng(29 + c)
    z = rnd(np.maximum(rng.standard_normal((n, h, w, c)) * 1.5 + 0.2, 0), dtype)
    gamma = (1 + 0.3 * rng.standard_normal(c)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(c)).astype(np.float32)
    hw = (rng.standard_normal((c, k)) * 0.3).astype(np.float32)
    hb = (rng.standard_normal(k) * 0.1).astype(np.float32)
    _, yt = O.synthetic_batch(n, (h, w), k, seed=3)
    L = N.lib()
    wsb = L.rvip_reduce_workspace(rows, 16 * c)
    ws = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev())
    zd, gd, bd, hwd, hbd, ytd = up(z, dtype), f32(gamma), f32(beta), f32(hw), f32(hb), f32(yt)
    mm, mv = f32(np.zeros(c)), f32(np.ones(c))
    mean, invstd, scale, shift = (torch.empty(c, dtype=torch.float32, device=dev()) for _ in range(4))
    N.call('rvip_bn_train_stats', P(zd), C.c_longlong(rows), c, ndt(dtype), P(gd), P(bd), P(mm), P(mv), 0.99, 1e-3, 1,
           P(mean), P(invstd), P(scale), P(shift), P(ws), C.c_size_t(wsb), stream())
    state = torch.zeros(8, dtype=torch.int32, device=dev())
    a = N.ApplyDesc()
    a.z, a.y, a.pooled = zd.data_ptr(), None, None
    a.scale, a.shift, a.act = scale.data_ptr(), shift.data_ptr(), 0
    a.drop_rate, a.mask, a.state, a.layer_id = 0.0, None, state.data_ptr(), 0
    a.n, a.h, a.w, a.c, a.dtype = n, h, w, c, ndt(dtype)
    inv_count = 1.0 / (rows * k)
    w_bce, w_dice = 0.5, 1.0

    def outputs():
        o = {'dz': torch.full((n, h, w, c), 3.0, dtype=tdt(dtype), device=dev())}
        for nm in ('dgamma', 'dbeta', 'dbias'):
            o[nm] = torch.full((c,), 7.0, dtype=torch.float32, device=dev())
        o['coef'] = torch.full((3 * c,), 7.0, dtype=torch.float32, device=dev())
        o['hdw'] = torch.full((c, k), 7.0, dtype=torch.float32, device=dev())
        o['hdb'] = torch.full((k,), 7.0, dtype=torch.float32, device=dev())
        o['pred'] = torch.empty((n, h, w, k), dtype=torch.float32, device=dev())
        o['sums'] = torch.zeros(16, dtype=torch.float32, device=dev())
        o['dlogit'] = torch.full((n, h, w, k), 7.0, dtype=torch.float32, device=dev())
        o['loss'] = torch.full((1,), 7.0, dtype=torch.float32, device=dev())
        return o

    def bwd_desc(o):
        b = N.BnBwdDesc()
        b.dy, b.z, b.dz = None, zd.data_ptr(), o['dz'].data_ptr()
        b.gamma, b.mean, b.invstd = gd.data_ptr(), mean.data_ptr(), invstd.data_ptr()
        b.scale, b.shift = scale.data_ptr(), shift.data_ptr()
        b.dgamma, b.dbeta, b.dbias, b.coef = o['dgamma'].data_ptr(), o['dbeta'].data_ptr(), o['dbias'].data_ptr(), o['coef'].data_ptr()
        b.act, b.act_after_bn = N.ACT['relu'], 0
        b.drop_rate, b.mask, b.state, b.layer_id = 0.0, None, state.data_ptr(), 0
        b.rows, b.c, b.dtype = rows, c, ndt(dtype)
        b.workspace, b.workspace_bytes = ws.data_ptr(), wsb
        return b
    # classic
    o0 = outputs()
    N.call('rvip_bn_apply_head', C.byref(a), P(hwd), P(hbd), k, P(o0['pred']), P(ytd), P(o0['sums']), P(ws), C.c_size_t(wsb), stream())
    N.call('rvip_head_grad', P(o0['pred']), P(ytd), P(o0['sums']), P(o0['dlogit']), P(o0['loss']), C.c_longlong(rows), k, N.LOSS_BCE_DICE,
           C.c_float(inv_count), C.c_float(lg), C.c_float(w_bce), C.c_float(w_dice), stream())
    if dscale != 1.0:
        N.call('rvip_scale_f32', P(o0['dlogit']), C.c_longlong(rows * k), C.c_float(dscale), stream())
    b0 = bwd_desc(o0)
    N.call('rvip_bn_bwd_reduce_head', C.byref(b0), P(hwd), P(o0['dlogit']), k, P(o0['hdw']), P(o0['hdb']), stream())
    N.call('rvip_bn_bwd_apply_head', C.byref(b0), P(hwd), P(o0['dlogit']), k, stream())
    nr = L.rvip_bn_apply_head_mse_rows(C.c_longlong(rows), c, ndt(dtype), k)

    def forward_side(min_gamma):
        o = outputs()
        frows = torch.full((nr * 7 * c,), 7.0, dtype=torch.float32, device=dev())
        dcoef = torch.full((4,), 7.0, dtype=torch.float32, device=dev())
        N.call('rvip_bn_apply_head_bcedice', C.byref(a), P(hwd), P(hbd), P(bd), k, P(o['pred']), P(ytd), P(o['sums']),
               P(frows), C.c_size_t(frows.numel() * 4), P(ws), C.c_size_t(wsb), stream())
        b = bwd_desc(o)
        hc = N.HeadCoefDesc()
        hc.bn, hc.beta = C.pointer(b), bd.data_ptr()
        hc.head_w, hc.dlogit, hc.k, hc.nrows = hwd.data_ptr(), None, k, nr
        hc.mse_rows, hc.head_dw, hc.head_db = frows.data_ptr(), o['hdw'].data_ptr(), o['hdb'].data_ptr()
        hc.sums, hc.loss_out, hc.inv_count = o['sums'].data_ptr(), o['loss'].data_ptr(), inv_count
        o['flags'] = torch.full((-(-c // 32),), 7, dtype=torch.int32, device=dev())
        hc.flags, hc.min_gamma, hc.max_beta_ratio = o['flags'].data_ptr(), min_gamma, 64.0
        hc.loss_kind, hc.w_bce, hc.w_dice, hc.local_over_global, hc.dscale = N.LOSS_BCE_DICE, w_bce, w_dice, lg, dscale
        hc.pred, hc.y_true, hc.dcoef = o['pred'].data_ptr(), ytd.data_ptr(), dcoef.data_ptr()
        N.call('rvip_head_mse_coef', C.byref(hc), stream())
        N.call('rvip_bn_bwd_apply_head_lazy', C.byref(b), P(hwd), P(o['pred']), P(ytd), P(dcoef), k, stream())
        torch.cuda.synchronize()
        o['dcoef'] = dcoef
        return o, (a, b, hc, frows)
    dl64, z64 = down(o0['dlogit']).astype(np.float64).reshape(rows, k), down(zd).astype(np.float64).reshape(rows, c)
    g64 = dl64 @ hw.astype(np.float64).T
    xh = (z64 - down(mean).astype(np.float64)) * down(invstd).astype(np.float64)
    want_db, want_dg = g64.sum(0), (g64 * xh).sum(0)
    gm_, is_, mu_ = gamma.astype(np.float64), down(invstd).astype(np.float64), down(mean).astype(np.float64)
    c2 = -gm_ * is_ * is_ * want_dg / rows
    exact = {'dbeta': want_db, 'dgamma': want_dg, 'coef': np.concatenate([gm_ * is_, c2, -gm_ * is_ * want_db / rows - c2 * mu_])}
    classic = {nm: down(o0[nm]).astype(np.float64) for nm in exact}
    loose = {'bf16': 8e-3, 'f16': 1.2e-3}[dtype] * max(1.0, (12288.0 / rows) ** 0.5)
    # the coefficients against the closed form from the folded sums
    sm = down(o0['sums']).astype(np.float64)
    den = sm[3] + sm[4] + 1.0
    want_coef = np.array([w_bce * inv_count, -w_dice * lg * 2.0 / den, w_dice * lg * (2.0 * sm[2] + 1.0) / den ** 2]) * dscale
    for min_gamma, want, tol in ((1.0 / 64, exact, loose), (1e9, classic, 2e-5)):
        o1, keep = forward_side(min_gamma)
        assert torch.equal(o1['pred'], o0['pred']) and torch.equal(o1['sums'], o0['sums'])
        assert bool((o1['dlogit'] == 7.0).all())                                  # no logit gradient is written in this form
        np.testing.assert_allclose(down(o1['loss']), down(o0['loss']), rtol=2e-6)
        np.testing.assert_allclose(down(o1['dcoef'])[:3], want_coef, rtol=3e-6)
        assert bool(down(o1['flags']).all()) == (min_gamma > 1) and bool(down(o1['flags']).any()) == (min_gamma > 1)
        for nm in ('hdw', 'hdb'):
            sc_ = float(np.abs(down(o0[nm])).max())
            np.testing.assert_allclose(down(o1[nm]), down(o0[nm]), atol=3e-5 * sc_ + 1e-12, err_msg=nm)
        for nm in ('dgamma', 'dbeta', 'coef'):
            got, ref = down(o1[nm]).reshape(-1, c).astype(np.float64), want[nm].reshape(-1, c)
            for i in range(got.shape[0]):
                assert np.abs(got[i] - ref[i]).max() <= tol * np.abs(ref[i]).max() + 1e-12, (nm, i, min_gamma, np.abs(got[i] - ref[i]).max(), np.abs(ref[i]).max())
        # the apply pass on the rebuilt gradient: dz and the bias-gradient rows against the classic launch (same coefficients up to `tol`)
        dz1, dz0 = down(o1['dz']).astype(np.float64), down(o0['dz']).astype(np.float64)
        scd = float(np.abs(dz0).max())
        assert np.abs(dz1 - dz0).max() <= (3 * tol + {'bf16': 2.0 ** -7, 'f16': 2.0 ** -10}[dtype]) * scd, (min_gamma, np.abs(dz1 - dz0).max() / scd)
        # (the bias gradient is the column sum of dz, which BatchNormalization's backward makes cancel: the two launches' per-element
        #  differences, zero-mean roundings, add up like a random walk over the rows)
        np.testing.assert_allclose(down(o1['dbias']), down(o0['dbias']), atol=(3 * tol + {'bf16': 2.0 ** -7, 'f16': 2.0 ** -10}[dtype]) * scd * np.sqrt(rows))
    # what the entry points refuse
    a2, b2, hc2, frows2 = keep
    args = lambda kk, nbytes: (C.byref(a2), P(hwd), P(hbd), P(bd), kk, P(o1['pred']), P(ytd), P(o1['sums']), P(frows2), C.c_size_t(nbytes),  # noqa: E731
                               P(ws), C.c_size_t(wsb), stream())
    assert L.rvip_bn_apply_head_bcedice(*args(3, frows2.numel() * 4)) == -2
    assert L.rvip_bn_apply_head_bcedice(*args(k, frows2.numel() * 4 - 4)) == -3
    a2.dtype = N.F32
    assert L.rvip_bn_apply_head_bcedice(*args(k, frows2.numel() * 4)) == -2
    a2.dtype = ndt(dtype)
    hc2.dcoef = None
    assert L.rvip_head_mse_coef(C.byref(hc2), stream()) == -1
    hc2.loss_kind = 5
    assert L.rvip_head_mse_coef(C.byref(hc2), stream()) == -1
    assert L.rvip_bn_bwd_apply_head_lazy(C.byref(b2), P(hwd), None, P(ytd), P(o1['dcoef']), k, stream()) == -1
    assert L.rvip_bn_bwd_apply_head_lazy(C.byref(b2), P(hwd), P(o1['pred']), P(ytd), P(o1['dcoef']), 3, stream()) == -2


@pytest.mark.parametrize('dtype', ['bf16', 'f16'])
@pytest.mark.parametrize('shape', [(2, 20, 36), (3, 37, 70), (1, 256, 256), (5, 8, 32), (2, 1, 3)], ids=lambda s: 'x'.join(map(str, s)))
def test_first_layer_weight_gradient_on_the_matrix_cores(shape, dtype, monkeypatch):
    """rvip_conv3x3_c1_wgrad with Cout = 32 and a 16-bit type runs c1_wgrad_mfma (im2col(x) x dy over the pixels on v_mfma_f32_32x32x16):
    against the float64 oracle (the products of two 16-bit values are exact in fp32, so the bound is the fp32 summation's: relative
    to sum |x| |dy| per element), ragged tiles and tiny maps included; the rows-only form (dw = NULL) leaves exactly the rows whose
    sum that is.  (The VALU form, RVIP_C1_WGRAD_MFMA=0, is what every other Cout takes: test_first_layer_c1.)"""
    n, h, w = shape
    co = 32
    rng = np.random.default_rng(14)
    x = rnd(rng.random((n, h, w, 1)) - 0.3, dtype)
    dy = rnd(rng.standard_normal((n, h, w, co)), dtype)
    xd, dyd = up(x, dtype), up(dy, dtype)
    L = N.lib()
    wsb = L.rvip_reduce_workspace(n * h * w, 16 * co)
    ws = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev())
    dw = torch.full((3, 3, 1, co), 7.0, dtype=torch.float32, device=dev())
    N.call('rvip_conv3x3_c1_wgrad', P(xd), P(dyd), P(dw), n, h, w, co, ndt(dtype), P(ws), C.c_size_t(wsb), stream())
    x64, dy64 = x.astype(np.float64), dy.astype(np.float64)
    xp = np.pad(x64[..., 0], ((0, 0), (1, 1), (1, 1)))
    ref = np.zeros((3, 3, 1, co))
    mag = np.zeros((3, 3, 1, co))
    for ky in range(3):
        for kx in range(3):
            sh = xp[:, ky:ky + h, kx:kx + w]
            ref[ky, kx, 0] = np.einsum('nhw,nhwc->c', sh, dy64)
            mag[ky, kx, 0] = np.einsum('nhw,nhwc->c', np.abs(sh), np.abs(dy64))
    got = down(dw).astype(np.float64)
    assert np.all(np.abs(got - ref) <= 2e-6 * mag + 1e-30), float(np.max(np.abs(got - ref) / (mag + 1e-30)))
    nr = L.rvip_conv3x3_c1_wgrad_rows(n, h, w, co, ndt(dtype))
    rows = torch.full((nr * 9 * co,), 5.0, dtype=torch.float32, device=dev())
    N.call('rvip_conv3x3_c1_wgrad', P(xd), P(dyd), None, n, h, w, co, ndt(dtype), P(rows), C.c_size_t(rows.numel() * 4), stream())
    torch.cuda.synchronize()
    np.testing.assert_allclose(rows.cpu().numpy().reshape(nr, 3, 3, 1, co).astype(np.float64).sum(0), got, rtol=0, atol=2e-6 * mag.max())


def test_half_storage_saturates_instead_of_overflowing():
    """IEEE half stores clamp at +-65504 (rvip_common.h: sat_f16): with unconverged moving statistics an fp16 evaluation pass can exceed
    the half range, and one infinity would turn the next layer's sums into NaN (tools/soak_probe.py).  bf16 has the range and rounds as
    before.  BN apply pass with a large scale, then the values stored."""
    n, h, w, c = 1, 4, 8, 16
    z = np.zeros((n, h, w, c), np.float32)
    z[0, 0, 0, :4] = [1000.0, -1000.0, 3.0, 0.5]
    scale = np.full(c, 100.0, np.float32)
    shift = np.zeros(c, np.float32)
    want = {'f16': [65504.0, -65504.0, 300.0, 50.0], 'bf16': [99840.0, -99840.0, 300.0, 50.0]}
    for dtype in ('f16', 'bf16'):
        zd = up(z, dtype)
        y = torch.empty((n, h, w, c), dtype=tdt(dtype), device=dev())
        a = N.ApplyDesc()
        a.z, a.y, a.pooled = zd.data_ptr(), y.data_ptr(), None
        a.act = 0
        a.drop_rate, a.mask, a.state, a.layer_id = 0.0, None, None, 0
        a.n, a.h, a.w, a.c, a.dtype = n, h, w, c, ndt(dtype)
        sc, sh = f32(scale), f32(shift)
        a.scale, a.shift = sc.data_ptr(), sh.data_ptr()
        N.call('rvip_bn_apply', C.byref(a), stream())
        got = down(y)[0, 0, 0, :4]
        assert np.isfinite(down(y)).all()
        np.testing.assert_array_equal(got, np.asarray(want[dtype], np.float32))
