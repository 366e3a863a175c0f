"""Guard bands around every OUTPUT of the kernels that address global memory through raw pointers.

The LDS-DMA loaders and the igemm epilogues go through bounded buffer resources (an out-of-range lane reads zero / is dropped);
the element-wise passes, the folds, the first-layer kernels, the pack re-layout, the slab stores of the weight-gradient kernels
and the bit-plane / argmax side channels use plain global stores.  An out-of-bounds store of one of those only faults when the
neighbouring page happens to be unmapped -- box-dependent.  Here every output lives INSIDE a larger canary-filled allocation, at
the real layer shapes of config 2 (batch cut to 1-2 images) and a ragged small shape, and the canaries must be intact after the
launch: an out-of-bounds write becomes a deterministic assertion on any box.  (Results themselves are checked against the oracle
in tests/test_gpu_ops.py / test_gpu_real_shapes.py; this file only watches the fences.)"""
import ctypes as C

import numpy as np
import pytest
import torch

from test_gpu_ops import N, P, conv_desc, dev, f32, ndt, pack, stream, tdt, up

pytestmark = pytest.mark.gpu
GUARD = 4096          # elements on either side


class Fenced:
    """a tensor of `shape` in the middle of a canary-filled buffer"""

    def __init__(self, shape, dtype, canary):
        self.n = int(np.prod(shape))
        self.buf = torch.full((self.n + 2 * GUARD,), canary, dtype=dtype, device=dev())
        self.t = self.buf[GUARD:GUARD + self.n].view(*shape)
        self.canary = canary
        assert self.t.data_ptr() == self.buf.data_ptr() + GUARD * self.buf.element_size()

    def ok(self):
        torch.cuda.synchronize()
        lo, hi = self.buf[:GUARD], self.buf[GUARD + self.n:]
        return bool((lo == self.canary).all()) and bool((hi == self.canary).all())


def _fences_ok(**named):
    bad = [k for k, f in named.items() if not f.ok()]
    assert not bad, 'out-of-bounds store into the guard band of: %s' % bad


SHAPES = [(1, 256, 256, 32), (2, 128, 128, 64), (2, 32, 32, 256), (3, 14, 22, 24)]


@pytest.mark.parametrize('dtype', ['bf16', 'f32'])
@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('pool', [0, 1])
def test_bn_apply_outputs_stay_inside_their_tensors(shape, dtype, pool):
    """rvip_bn_apply: y, pooled, the 2-bit window argmax and the Dropout keep-bit planes"""
    n, h, w, c = shape
    if pool and ((h | w) & 1):
        pytest.skip('odd map: not pooled')
    rng = np.random.default_rng(1)
    ve = 4 if dtype == 'f32' else 8
    z = up(rng.standard_normal(shape), dtype)
    scale, shift = f32(1 + 0.1 * rng.standard_normal(c)), f32(0.1 * rng.standard_normal(c))
    y = Fenced(shape, tdt(dtype), 3.0)
    pooled = Fenced((n, h // 2, w // 2, c), tdt(dtype), 3.0) if pool else None
    L = N.lib()
    argmax = Fenced((n * (h // 2) * (w // 2) * (c // ve),), torch.int16, 21845) if pool and L.rvip_bn_apply_argmax_ok(c, ndt(dtype)) else None
    kb = Fenced((-(-c // 32) * n * h * w,), torch.int32, 0x55555555) if (not pool and (c % 32 == 0 or c in (8, 16))) else None
    state = torch.zeros(8, dtype=torch.int32, device=dev())
    state[N.STATE_SEED] = 7
    a = N.ApplyDesc()
    a.z, a.y, a.pooled = z.data_ptr(), y.t.data_ptr(), (pooled.t.data_ptr() if pool else None)
    a.scale, a.shift, a.act = scale.data_ptr(), shift.data_ptr(), 0
    a.drop_rate, a.mask, a.state, a.layer_id = (0.0 if pool else 0.3), None, state.data_ptr(), 3
    a.n, a.h, a.w, a.c, a.dtype = n, h, w, c, ndt(dtype)
    if argmax is not None:
        a.argmax = argmax.t.data_ptr()
    if kb is not None:
        a.keep_bits = kb.t.data_ptr()
    N.call('rvip_bn_apply', C.byref(a), stream())
    fences = dict(y=y)
    if pool:
        fences['pooled'] = pooled
    if argmax is not None:
        fences['argmax'] = argmax
    if kb is not None:
        fences['keep_bits'] = kb
    _fences_ok(**fences)


@pytest.mark.parametrize('dtype', ['bf16', 'f32'])
@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_bn_backward_passes_stay_inside_their_tensors(shape, dtype):
    """rvip_bn_bwd_reduce + rvip_bn_bwd_apply: dz, the bias-gradient rows, dgamma / dbeta / coef"""
    n, h, w, c = shape
    rows = n * h * w
    rng = np.random.default_rng(2)
    L = N.lib()
    z, dy = up(np.maximum(rng.standard_normal(shape), 0), dtype), up(rng.standard_normal(shape), dtype)
    gamma, mean, invstd = f32(1 + 0.1 * rng.standard_normal(c)), f32(0.3 * np.ones(c)), f32(np.ones(c))
    dz = Fenced(shape, tdt(dtype), 3.0)
    nr = L.rvip_bn_bwd_rows(C.c_longlong(rows), c, ndt(dtype))
    brows = Fenced((nr * c,), torch.float32, 3.0)
    dgamma, dbeta, coef = Fenced((c,), torch.float32, 3.0), Fenced((c,), torch.float32, 3.0), Fenced((3 * c,), torch.float32, 3.0)
    wsb = L.rvip_reduce_workspace(rows, 16 * c)
    ws = Fenced((wsb // 4,), torch.float32, 3.0)
    state = torch.zeros(8, dtype=torch.int32, device=dev())
    b = N.BnBwdDesc()
    b.dy, b.z, b.dz = dy.data_ptr(), z.data_ptr(), dz.t.data_ptr()
    b.gamma, b.mean, b.invstd = gamma.data_ptr(), mean.data_ptr(), invstd.data_ptr()
    b.dgamma, b.dbeta, b.coef = dgamma.t.data_ptr(), dbeta.t.data_ptr(), coef.t.data_ptr()
    b.dbias = None
    b.act, b.act_after_bn = N.ACT['relu'], 0
    b.drop_rate, b.mask, b.state, b.layer_id = 0.0, None, state.data_ptr(), 0
    b.rows, b.c, b.dtype = rows, c, ndt(dtype)
    b.workspace, b.workspace_bytes = ws.t.data_ptr(), wsb
    b.bias_rows, b.bias_rows_bytes = brows.t.data_ptr(), brows.n * 4
    N.call('rvip_bn_bwd_reduce', C.byref(b), stream())
    N.call('rvip_bn_bwd_apply', C.byref(b), stream())
    _fences_ok(dz=dz, bias_rows=brows, dgamma=dgamma, dbeta=dbeta, coef=coef, workspace=ws)


WG_SHAPES = [  # n, h, c0, up0, c1, cout: the forms rvip_conv3x3_wgrad chooses among
    (1, 256, 32, 0, 0, 32), (1, 256, 64, 1, 0, 32), (1, 256, 32, 0, 32, 32), (1, 128, 128, 1, 0, 64), (2, 32, 512, 1, 0, 256),
    (2, 16, 512, 0, 0, 512), (1, 24, 24, 0, 0, 40),
]


@pytest.mark.parametrize('dtype', ['bf16', 'f32'])
@pytest.mark.parametrize('shape', WG_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_wgrad_slabs_folds_and_dot_rows_stay_inside_their_buffers(shape, dtype):
    """rvip_conv3x3_wgrad in all its forms (nine-tap LDS-DMA, sub-pixel four-phase, sub-pixel pair PB = 1, register-staged fallback):
    the split-K slabs in the caller's workspace, dw, and the double-precision dot rows"""
    n, h, c0, up0, c1, co = shape
    rng = np.random.default_rng(3)
    L = N.lib()
    hs = h // 2 if up0 else h
    x0 = up(rng.standard_normal((n, hs, hs, c0)), dtype)
    x1 = up(rng.standard_normal((n, h, h, c1)), dtype) if c1 else None
    dy = up(rng.standard_normal((n, h, h, co)), dtype)
    wm = f32(rng.standard_normal((3, 3, c0 + c1, co)) * 0.1)
    wsb = L.rvip_conv3x3_wgrad_workspace(n, h, h, c0 + c1, co)
    ws = Fenced((wsb // 4,), torch.float32, 3.0)
    dw = Fenced((3, 3, c0 + c1, co), torch.float32, 3.0)
    g = N.Wgrad3x3Desc()
    g.x0, g.c0, g.up0 = x0.data_ptr(), c0, up0
    g.x1, g.c1 = (x1.data_ptr(), c1) if c1 else (None, 0)
    g.dy, g.dw = dy.data_ptr(), dw.t.data_ptr()
    g.n, g.h, g.w, g.cout, g.dtype = n, h, h, co, ndt(dtype)
    g.workspace, g.workspace_bytes = ws.t.data_ptr(), wsb
    nd = L.rvip_conv3x3_wgrad_dot_rows(C.byref(g))
    drows = Fenced((nd * (c0 + c1),), torch.float64, 3.0)
    g.w_master, g.dot_rows, g.dot_rows_bytes = wm.data_ptr(), drows.t.data_ptr(), drows.n * 8
    form = L.rvip_conv3x3_wgrad_form(C.byref(g))
    assert form in (0, 1, 2, 3)
    N.call('rvip_conv3x3_wgrad', C.byref(g), stream())
    _fences_ok(workspace=ws, dw=dw, dot_rows=drows)
    assert bool(torch.isfinite(dw.t).all())
    # deferred fold: exactly rvip_conv3x3_wgrad_splits() slabs in a private region, then the table-driven batch fold
    ns = L.rvip_conv3x3_wgrad_splits(C.byref(g))
    slabs = Fenced((ns * 9 * (c0 + c1) * co,), torch.float32, 3.0)
    dw2 = Fenced((3, 3, c0 + c1, co), torch.float32, 3.0)
    g.workspace, g.workspace_bytes, g.defer_fold = slabs.t.data_ptr(), slabs.n * 4, 1
    g.w_master, g.dot_rows, g.dot_rows_bytes = None, None, 0
    N.call('rvip_conv3x3_wgrad', C.byref(g), stream())
    tab = (N.FoldEntry * 1)()
    tab[0].src, tab[0].dst, tab[0].nrows, tab[0].stride, tab[0].width = slabs.t.data_ptr(), dw2.t.data_ptr(), ns, 0, 9 * (c0 + c1) * co
    tabd = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev())
    N.call('rvip_fold_rows_batch', P(tabd), 1, C.c_longlong(9 * (c0 + c1) * co), 1, stream())
    _fences_ok(slabs=slabs, dw_deferred=dw2)
    np.testing.assert_allclose(dw2.t.cpu().numpy(), dw.t.cpu().numpy(), rtol=1e-4, atol=1e-4 * float(dw.t.abs().max()))


@pytest.mark.parametrize('dtype', ['bf16', 'f32'])
@pytest.mark.parametrize('shape', [(1, 256, 256, 32), (2, 64, 96, 16), (3, 10, 34, 8)], ids=lambda s: 'x'.join(map(str, s)))
def test_first_layer_kernels_stay_inside_their_tensors(shape, dtype):
    """rvip_conv3x3_c1_fwd(_stats) / rvip_conv3x3_c1_wgrad: y, the statistics rows, dw and the reduction workspace"""
    n, h, w, co = shape
    rng = np.random.default_rng(4)
    L = N.lib()
    x = up(rng.random((n, h, w, 1)), dtype)
    wm, bias = f32(rng.standard_normal((3, 3, 1, co)) * 0.3), f32(0.1 * rng.standard_normal(co))
    y = Fenced((n, h, w, co), tdt(dtype), 3.0)
    wsb = L.rvip_reduce_workspace(n * h * w, 16 * co)
    ws = Fenced((wsb // 4,), torch.float32, 3.0)
    N.call('rvip_conv3x3_c1_fwd', P(x), P(wm), P(bias), P(y.t), n, h, w, co, N.ACT['relu'], ndt(dtype), stream())
    _fences_ok(y=y)
    if L.rvip_conv3x3_c1_fwd_stats_rows(n, h, w, co, ndt(dtype)) > 0:
        N.call('rvip_conv3x3_c1_fwd_stats', P(x), P(wm), P(bias), P(y.t), n, h, w, co, N.ACT['relu'], ndt(dtype), P(ws.t), C.c_size_t(wsb), stream())
        _fences_ok(y=y, stats_rows=ws)
    dy = up(rng.standard_normal((n, h, w, co)), dtype)
    dw = Fenced((3, 3, 1, co), torch.float32, 3.0)
    N.call('rvip_conv3x3_c1_wgrad', P(x), P(dy), P(dw.t), n, h, w, co, ndt(dtype), P(ws.t), C.c_size_t(wsb), stream())
    _fences_ok(dw=dw, workspace=ws)
    # rows-only form (dw = NULL): exactly rvip_conv3x3_c1_wgrad_rows() rows of 9 * cout floats in a private region, folded by the batch fold
    nr = L.rvip_conv3x3_c1_wgrad_rows(n, h, w, co, ndt(dtype))
    assert nr > 0
    rows = Fenced((nr * 9 * co,), torch.float32, 3.0)
    dw2 = Fenced((3, 3, 1, co), torch.float32, 3.0)
    N.call('rvip_conv3x3_c1_wgrad', P(x), P(dy), None, n, h, w, co, ndt(dtype), P(rows.t), C.c_size_t(rows.n * 4), stream())
    assert L.rvip_conv3x3_c1_wgrad(P(x), P(dy), None, n, h, w, co, ndt(dtype), P(rows.t), C.c_size_t(rows.n * 4 - 4), stream()) == -3
    tab = (N.FoldEntry * 1)()
    tab[0].src, tab[0].dst, tab[0].nrows, tab[0].stride, tab[0].width = rows.t.data_ptr(), dw2.t.data_ptr(), nr, 0, 9 * co
    tabd = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev())
    N.call('rvip_fold_rows_batch', P(tabd), 1, C.c_longlong(9 * co), 0, stream())
    _fences_ok(rows=rows, dw_batched=dw2)
    np.testing.assert_allclose(dw2.t.cpu().numpy(), dw.t.cpu().numpy(), rtol=2e-6, atol=2e-6 * float(dw.t.abs().max()))


@pytest.mark.parametrize('dtype', ['bf16', 'f16'])
def test_conv3d_first_layer_stays_inside_its_tensor(dtype):
    """rvip_conv3d_c1_fwd (rewritten in round 3: four pixels per thread) at config 5's plane size, two volumes of four frames"""
    n, d, h, w, co = 8, 4, 256, 256, 32
    rng = np.random.default_rng(5)
    x = up(rng.random((n, h, w, 1)), dtype)
    wm, bias = f32(rng.standard_normal((3, 3, 3, 1, co)) * 0.2), f32(0.1 * rng.standard_normal(co))
    y = Fenced((n, h, w, co), tdt(dtype), 3.0)
    N.call('rvip_conv3d_c1_fwd', P(x), P(wm), P(bias), P(y.t), n, d, h, w, co, N.ACT['relu'], ndt(dtype), stream())
    _fences_ok(y=y)
    assert bool(torch.isfinite(y.t.float()).all())


@pytest.mark.parametrize('dtype', ['bf16'])
@pytest.mark.parametrize('shape', [(1, 256, 32, 32), (2, 64, 128, 128), (1, 24, 24, 40)], ids=lambda s: 'x'.join(map(str, s)))
def test_dgrad_column_sum_rows_and_gated_results_stay_inside_their_buffers(shape, dtype):
    """rvip_conv3x3_fwd_sums: the result tensor and the partial rows of its column sums"""
    n, h, ci, co = shape
    rng = np.random.default_rng(6)
    L = N.lib()
    x = up(rng.standard_normal((n, h, h, ci)), dtype)
    wt = (rng.standard_normal((3, 3, ci, co)) * 0.1).astype(np.float32)
    wf, _ = pack(wt, dtype)
    y = Fenced((n, h, h, co), tdt(dtype), 3.0)
    d = conv_desc(x, ci, 0, None, 0, wf, None, y.t, None, 0, n, h, h, co, 0, dtype)
    nr = L.rvip_conv3x3_fwd_sums_rows(C.byref(d))
    if nr <= 0:
        pytest.skip('shape served by the register-staged kernel (no fused sums)')
    rows = Fenced((nr * co,), torch.float32, 3.0)
    N.call('rvip_conv3x3_fwd_sums', C.byref(d), P(rows.t), C.c_size_t(rows.n * 4), stream())
    _fences_ok(y=y, sum_rows=rows)
