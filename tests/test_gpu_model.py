"""End-to-end parity of the hot path on a real MI355X: get_model(config) -> train steps / predict through the
engine (every FLOP a HIP kernel behind the C ABI) against the NumPy oracle on identical seeded inputs, with the
device's own dropout stream reproduced on the host.

north_star bar: fp32 heat-maps within 1e-3 max-abs of the CPU reference, argmax landmark indices bit-exact,
>0.5 masks identical.  The bf16 path reports its own (looser) error."""
import importlib
import os

import numpy as np
import pytest
import torch

import cmr_landmark_detection_amd as rvip
from oracle import rvip_oracle as O

pytestmark = pytest.mark.gpu
M = rvip.Loss_and_metrics
ds = importlib.import_module('cmr-landmark-detection_amd.dropout_stream')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(**kw):
    c = dict(DIM=[32, 32], FILTERS=8, DEPTH=2, BATCH_NORMALISATION=True, ACTIVATION='relu', MASK_CLASSES=2,
             LEARNING_RATE=1e-3, RVIP_PRECISION='fp32', LOSS_FUNCTION=M.mse, SEED=11)
    c.update(kw)
    return c


def _oracle_from(model, cfg, dtype=np.float64):
    layers = O.build_graph(cfg)
    it = iter(model.get_weights())
    params = {}
    for l in layers:
        if l['type'].startswith('Conv'):
            params[l['name']] = [next(it), next(it)]
        elif l['type'] == 'BatchNormalization':
            params[l['name']] = [next(it) for _ in range(4)]
    return O.OracleUNet(cfg, params, dtype=dtype), layers


def _masks(layers, batch, seed, step):
    drops = [l for l in layers if l['type'] == 'Dropout']
    return {l['name']: ds.keep_mask((batch,) + l['shape'], l['rate'], seed, step, i + 1) for i, l in enumerate(drops)}


def _batch(B, cfg, seed):
    """Synthetic (x, y) of the config's shapes; further image channels are further seeded draws."""
    x, y = O.synthetic_batch(B, cfg['DIM'], cfg['MASK_CLASSES'], seed=seed)
    extra = [O.synthetic_batch(B, cfg['DIM'], 1, seed=1000 * c + seed)[0] for c in range(1, cfg.get('IMG_CHANNELS', 1))]
    return (np.concatenate([x] + extra, -1) if extra else x), y


def _flat_grads(grads):
    return {(k, i): g for k, gs in grads.items() for i, g in enumerate(gs)}


# Data seeds whose FIRST training step has no knife edge (no ReLU / pooling decision within fp32 noise; found on the CPU with the
# float64 oracle alone by tools/find_clean_seed.py at 1.5x the test's margin, up to 4 000 seeds per variant).  The chance of such a
# step falls exponentially with the number of pre-activations (about 1.4e-5 per element at the 4.5e-6 relative margin): the
# 48 x 40 / F = 12 / depth-5 / 3-D variants have none among 4 000 seeds at their size (12 - 40 knife elements per step) and keep
# seed 3 there; each has a SMALLER twin below (same layer types and code paths, fewer elements) whose first step IS knife-free, so
# that the tight bound is exercised -- and asserted, with clean_total > 0 -- for every graph family.
CLEAN_SEEDS = {
    'default': 325,
    'BN_FIRST=True': 994,
    'BATCH_NORMALISATION=False,ACTIVATION=elu': 0,
    'MASK_CLASSES=4,LOSS_FUNCTION=bce_dice_loss': 240,
    'USE_UPSAMPLE=False': 238,
    'IMG_CHANNELS=3': 52,
    'DEPTH=3,DIM=[24, 40],LOSS_FUNCTION=bce_dice_loss': 1535,
    'LOSS_FUNCTION=BcdDiceLoss_w_1.0_1.0,FILTERS=12,DIM=[16, 32]': 57,
    'DEPTH=5,DIM=[32, 32],FILTERS=4,RVIP_PRECISION=fp32': 35,
    'DIM=[2, 8, 8],M_POOL=[1, 2, 2],F_SIZE=[3, 3, 3],FILTERS=32,USE_UPSAMPLE=False': 2,
    'DIM=[2, 8, 8],M_POOL=[1, 2, 2],F_SIZE=[3, 3, 3],FILTERS=32': 0,
}

VARIANTS = [
    dict(),
    dict(BN_FIRST=True),
    dict(BATCH_NORMALISATION=False, ACTIVATION='elu'),
    dict(DEPTH=3, DIM=[48, 40], LOSS_FUNCTION=M.bce_dice_loss),
    dict(LOSS_FUNCTION=M.BceDiceLoss(), FILTERS=12),
    dict(MASK_CLASSES=4, LOSS_FUNCTION=M.bce_dice_loss),   # 4-class head: loss and dice_coef_labels drop the background channel (Loss_and_metrics.py:240-242, :158-159)
    dict(USE_UPSAMPLE=False),                       # Conv2DTranspose decoder (KerasLayers.py:761-765)
    dict(IMG_CHANNELS=3),                           # Input((*dim, IMG_CHANNELS)), Unets.py:77
    dict(DEPTH=5, DIM=[64, 64], FILTERS=4, RVIP_PRECISION='fp32'),   # cfg 4's depth (bottleneck 2x2), fp32: F % 4 == 0
    dict(DIM=[4, 32, 32], M_POOL=[1, 2, 2], F_SIZE=[3, 3, 3], FILTERS=32, USE_UPSAMPLE=False),   # Conv3DTranspose(3, strides (1, 2, 2)) decoder
    dict(DIM=[4, 32, 32], M_POOL=[1, 2, 2], F_SIZE=[3, 3, 3], FILTERS=32),   # cfg 5's graph (Conv3D, MaxPooling3D, UpSampling3D); 3-D runs on the LDS-DMA kernels only: concat halves must be whole 128-byte rows (F % 32 in fp32)
    # smaller twins of the variants above that have no knife-free first step at their size (see CLEAN_SEEDS)
    dict(DEPTH=3, DIM=[24, 40], LOSS_FUNCTION=M.bce_dice_loss),
    dict(LOSS_FUNCTION=M.BceDiceLoss(), FILTERS=12, DIM=[16, 32]),
    dict(DEPTH=5, DIM=[32, 32], FILTERS=4, RVIP_PRECISION='fp32'),
    dict(DIM=[2, 8, 8], M_POOL=[1, 2, 2], F_SIZE=[3, 3, 3], FILTERS=32, USE_UPSAMPLE=False),
    dict(DIM=[2, 8, 8], M_POOL=[1, 2, 2], F_SIZE=[3, 3, 3], FILTERS=32),
]


BRANCH_REPORT = {}            # variant id -> {'tight': n, 'f32': n, 'knife': n, 'worst_tight_ratio': x, 'detail': [...]}; written by the last test of the family


def _variant_id(v):
    return ','.join('%s=%s' % (k, getattr(x, '__name__', type(x).__name__ if not isinstance(x, (int, float, str, list, bool)) else x)) for k, x in v.items()) or 'default'


@pytest.mark.parametrize('variant', VARIANTS, ids=_variant_id)
def test_fp32_training_steps_match_oracle(variant):
    """Three training steps on the fp32 device path against the float64 oracle: loss, heat-maps, EVERY parameter gradient, Adam
    and the BN moving statistics.  Gradient bound per tensor, and WHICH bound admitted it is recorded (BRANCH_REPORT, asserted
    below and dumped to gpurun_out/r03_tolerance_branches.json):
      tight  |g_dev - g_64|max <= max(3e-4 * |g_64|max, 5e-8)
      f32    ... <= |g_32 - g_64|max: no worse than the float32 CPU evaluation of the same graph (ill-conditioned BN backward)
      knife  ... <= 25 % of |g_64|max, only when the float64 oracle finds a ReLU / pooling decision inside fp32 noise
    In every step the oracle finds free of knife edges the knife bound is not available and the tight bound must hold for >= 90 % of
    the tensors (the rest may take the f32 bound); the data seeds of three variants (CLEAN_SEEDS, chosen on the CPU with the oracle
    alone: tools/find_clean_seed.py) make their first step such a step, which is asserted."""
    cfg = _cfg(**variant)
    kind = M.resolve_loss(cfg['LOSS_FUNCTION'])
    red = M.loss_reduction(cfg['LOSS_FUNCTION'])       # 'sum' for the BceDiceLoss class form (oracle/rvip_oracle.py::bce_dice_loss)
    loss_name = kind[0]
    B = 4
    model = rvip.get_model(cfg, metrics=[M.dice_coef_labels, M.dice_coef_lower, M.dice_coef_upper])
    ref, layers = _oracle_from(model, cfg)
    ref32, _ = _oracle_from(model, cfg, dtype=np.float32)     # conditioning probe: the same graph evaluated in float32
    x, y = _batch(B, cfg, CLEAN_SEEDS.get(_variant_id(variant), 3))
    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    eng = model._engine(B)
    wname = {0: 'kernel', 1: 'bias'}
    specs = model.plan.weight_specs()
    rep = BRANCH_REPORT.setdefault(_variant_id(variant), dict(tight=0, f32=0, knife=0, clean_steps=0, clean_tight=0, clean_total=0,
                                                              worst_tight_ratio=0.0, knife_steps=[], detail=[],
                                                              landmark_near_ties=0, mask_near_threshold=0, landmarks_total=0))
    for step in range(3):
        # Adam turns fp32 noise on near-zero gradients into O(lr) weight differences, so the oracle restarts every
        # step from the DEVICE weights; the optimiser arithmetic is checked separately on the device's gradients.
        ref.set_weights(model.get_weights())
        ref32.set_weights(model.get_weights())
        masks = _masks(layers, B, model.seed, step)
        eng.load_input(x, y)
        eng.forward(training=True)
        eng.backward()
        torch.cuda.synchronize()
        if loss_name == 'mse':
            lv, rgrads, rpred, cache = ref.loss_and_grads(x64, y64, 'mse', masks)
        else:
            # reference objects differ in w_bce (0.5 for bce_dice_loss, 1 for BceDiceLoss) and in the reduction Keras applies
            lv, rgrads, rpred, cache = ref.loss_and_grads(x64, y64, 'bce_dice', masks, w_bce=kind[1], w_dice=kind[2], reduction=red)
        # float32 evaluation of the oracle: BN backward cancels the common-mode part of the incoming gradient (large
        # with BCE-Dice), so fp32 results scatter around the float64 truth by far more than 1e-7 on some inputs
        p32, c32 = ref32.forward(x, True, masks)
        if loss_name == 'mse':
            g32 = ref32.backward(c32, O.mse_loss(y, p32)[1])
        else:
            g32 = ref32.backward(c32, O.bce_dice_loss(y, p32, w_bce=kind[1], w_dice=kind[2], logits=c32['logits'], reduction=red)[1].astype(np.float32),
                                 d_is_logit_grad=True)
        assert abs(float(eng.loss.item()) - lv) <= 2e-5 * max(1.0, abs(lv)), (step, float(eng.loss.item()), lv)
        np.testing.assert_allclose(eng.pred.cpu().numpy().reshape(rpred.shape), rpred, atol=1e-4)
        # A ReLU pre-activation (or a pooling near-tie) within fp32 noise of the kink makes the gradient of ANY float32
        # evaluation a coin flip at that element (one flipped element moves a layer gradient of this tiny net by several
        # per cent).  The float64 oracle tells us when that is the case; only then the bound is flip-tolerant.
        knife = O.knife_edges(layers, cache)
        if knife:
            rep['knife_steps'].append(step)
        if _variant_id(variant) in CLEAN_SEEDS and step == 0:
            assert not knife, 'the clean seed no longer avoids knife edges in the first step: %s' % (knife,)
        rep['clean_steps'] += 0 if knife else 1
        got = model._params.grads_host()
        dev_grads = {}
        for (lname, i), g in _flat_grads(rgrads).items():
            wn = wname[i] if lname.startswith('conv') or lname == 'unet' else ('gamma', 'beta')[i]
            gg = got[(lname, wn)]
            gmax = float(np.abs(g).max())
            err = float(np.abs(gg - g).max())
            tight = max(3e-4 * gmax, 5e-8 * max(1.0, eng.grad_factor))       # absolute floor: a conv bias in front of BN has an exactly-zero gradient
            f32b = float(np.abs(g32[lname][i] - g).max())
            if err <= tight:
                branch = 'tight'
                rep['worst_tight_ratio'] = max(rep['worst_tight_ratio'], err / tight)
            elif err <= f32b:
                branch = 'f32'
            elif knife and err <= 0.25 * gmax:
                branch = 'knife'
            else:
                raise AssertionError((step, lname, wn, err, dict(tight=tight, f32=f32b, knife=bool(knife), gmax=gmax)))
            rep[branch] += 1
            if not knife:
                rep['clean_total'] += 1
                rep['clean_tight'] += branch == 'tight'
            if branch != 'tight':
                rep['detail'].append([step, '%s/%s' % (lname, wn), branch, err, tight, f32b, gmax])
            dev_grads.setdefault(lname, [None, None])[i] = gg.astype(np.float64)
        eng.optimizer_step()
        model.optimizer.iterations += 1
        torch.cuda.synchronize()
        # Keras-Adam + BN moving statistics, driven by the device's own gradients: must agree to fp32 rounding
        ref.apply_bn_moving(cache)
        ref.apply_adam({k: dev_grads[k] for k in rgrads})
        for a_, b_, (ln, wn, _, _, _) in zip(model.get_weights(), ref.get_weights(), specs):
            np.testing.assert_allclose(a_, b_, atol=3e-6, rtol=1e-5, err_msg='step %d %s/%s' % (step, ln, wn))
    # wherever the float64 oracle finds the comparison well-posed (no knife edge in that step), the tight bound must be the rule
    assert rep['clean_tight'] >= 0.9 * rep['clean_total'], 'tight gradient bound held on %d of %d tensors of knife-free steps: %s' % (
        rep['clean_tight'], rep['clean_total'], [d for d in rep['detail'] if d[2] != 'knife'][:8])
    if _variant_id(variant) in CLEAN_SEEDS:
        assert rep['clean_total'] > 0                                      # ... and the rule was really exercised for this graph
    torch.cuda.synchronize()
    assert model._params.step_count() == 3
    # inference after training: heat-maps 1e-3, argmax bit-exact, >0.5 masks identical
    xt, _ = _batch(6, cfg, 9)
    pg = model.predict(xt, batch_size=3)
    pr = ref.predict(xt.astype(np.float64))
    assert pg.dtype == np.float32 and pg.shape == pr.shape
    assert np.abs(pg - pr).max() < 1e-3
    nd, nm = _assert_landmarks_and_masks(pg, pr)
    # mismatches tolerated because the ORACLE's own two candidates tie within 2e-5 (every other mismatch fails inside the helper): reported
    # per variant (profiles/r03_tolerance_branches.json).  After three steps from a random initialisation the tiny 3-D nets still predict
    # almost flat heat-maps, so a near-tie is common there; the bit-exact claim is asserted on BASELINE config 1 (nd == 0, below).
    rep['landmark_near_ties'], rep['mask_near_threshold'] = nd, nm
    rep['landmarks_total'] = int(np.prod(pg.shape[:-3])) * pg.shape[-1]
    assert nd <= rep['landmarks_total'] // 2, (nd, rep['landmarks_total'])
    idx, mask = model.predict_landmarks(xt[:3])
    np.testing.assert_array_equal(idx, O.landmark_argmax(pg[:3]))
    np.testing.assert_array_equal(mask.astype(bool), O.threshold_mask(pg[:3]))
    if len(cfg['DIM']) == 2:                 # the reference's post-threshold (flat labels, CC filter, mean RVIP points) on the device
        flat, pts, sizes = model.predict_rvip(xt[:3], cc_filter=True)
        rflat = O.clean_2d_cc(O.flat_labels(pg[:3]))
        np.testing.assert_array_equal(flat, rflat)
        rp = O.mean_rvip_points(rflat, pg.shape[-1])
        assert np.array_equal(np.isnan(pts), np.isnan(rp))
        np.testing.assert_allclose(pts[~np.isnan(rp)], rp[~np.isnan(rp)], rtol=1e-6)


@pytest.mark.parametrize('classes', [2, 4])
def test_train_on_batch_logs_and_metrics(classes):
    cfg = _cfg(LOSS_FUNCTION=M.bce_dice_loss, MASK_CLASSES=classes)
    model = rvip.get_model(cfg, metrics=[M.dice_coef_labels, M.dice_coef_lower, M.dice_coef_upper])
    ref, layers = _oracle_from(model, cfg)
    x, y = O.synthetic_batch(4, cfg['DIM'], classes, seed=4)
    logs = model.train_on_batch(x, y, return_dict=True)
    rpred, cache = ref.forward(x.astype(np.float64), True, _masks(layers, 4, model.seed, 0))
    lv, _ = O.bce_dice_loss(y.astype(np.float64), rpred, logits=cache['logits'])
    dm = O.dice_metrics(y.astype(np.float64), rpred)
    assert abs(logs['loss'] - lv) < 1e-4
    for k in ('dice_coef_labels', 'dice_coef_lower', 'dice_coef_upper'):
        assert abs(logs[k] - dm[k]) < 1e-5, k
    ev = model.evaluate(x, y, return_dict=True)
    assert np.isfinite(ev['loss'])


def test_reference_default_config_224_fp32_forward():
    """BASELINE.json configs[0]: template config (224x224, FILTERS 32, depth 4, batch 2), fp32, synthetic slice."""
    cfg = dict(DIM=[224, 224], FILTERS=32, DEPTH=4, BATCH_NORMALISATION=True, BN_FIRST=False, ACTIVATION='relu',
               MASK_CLASSES=2, M_POOL=[2, 2], F_SIZE=[3, 3], LEARNING_RATE=1e-4, RVIP_PRECISION='fp32',
               LOSS_FUNCTION=M.BceDiceLoss(), SEED=42)
    model = rvip.get_model(cfg)
    assert model.count_params() == 8641730
    ref, _ = _oracle_from(model, cfg, dtype=np.float32)
    x, y = O.synthetic_batch(2, cfg['DIM'], 2, seed=42)
    pg = model.predict(x)
    pr = ref.predict(x)
    assert np.abs(pg - pr).max() < 1e-3
    nd, nm = _assert_landmarks_and_masks(pg, pr.astype(np.float64), eps=1e-4)
    assert nd == 0                                             # landmark indices bit-exact on this input
    loss = model.train_on_batch(x, y)[0]
    assert np.isfinite(loss)


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
def test_reference_default_config_224_low_precision_training(precision):
    """The template config's shape (224 -> 112 -> 56 -> 28 -> 14: ragged against the 32 x 16 / 16 x 16 pixel tiles at every
    level) on the 16-bit paths: heat-maps within the storage precision of the fp32 path's, a repeated step is bit-identical,
    and a few Adam steps on one batch lower the loss."""
    cfg = dict(DIM=[224, 224], FILTERS=32, DEPTH=4, BATCH_NORMALISATION=True, BN_FIRST=False, ACTIVATION='relu',
               MASK_CLASSES=2, M_POOL=[2, 2], F_SIZE=[3, 3], LEARNING_RATE=1e-3, RVIP_PRECISION=precision,
               LOSS_FUNCTION=M.mse, SEED=42)
    model = rvip.get_model(cfg, metrics=[])
    ref32 = rvip.get_model(dict(cfg, RVIP_PRECISION='fp32'), metrics=[])
    ref32.set_weights(model.get_weights())
    x, y = O.synthetic_batch(2, cfg['DIM'], 2, seed=42)
    d = np.abs(model.predict(x) - ref32.predict(x))
    assert d.max() < (0.08 if precision == 'bf16' else 0.02) and d.mean() < (8e-3 if precision == 'bf16' else 2e-3), (d.max(), d.mean())
    eng = model._engine(2)
    outs = []
    for _ in range(2):
        eng.load_input(x, y)
        eng.forward(training=True)
        eng.backward()
        torch.cuda.synchronize()
        outs.append((eng.loss.clone(), eng.pred.clone(), model._params.grad.clone()))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    assert torch.isfinite(outs[0][2]).all()
    ls = [model.train_on_batch(x, y)[0] for _ in range(6)]
    assert np.all(np.isfinite(ls)) and ls[-1] < ls[0], ls


def test_bce_dice_class_form_gradient_is_the_sum_reduction():
    """BceDiceLoss() (Loss_and_metrics.py:207-226, overrides Loss.__call__) against the function form with the same weights: same
    logged loss, every parameter gradient B_local*H*W times larger (oracle/rvip_oracle.py::bce_dice_loss gives the Keras argument)."""
    fn = M._tag('loss', loss='bce_dice', w_bce=1.0, w_dice=1.0)(lambda t, p: M.bce_dice_loss(t, p, 1.0, 1.0))
    x, y = O.synthetic_batch(4, [32, 32], 2, seed=3)
    out = {}
    for name, loss in (('class', M.BceDiceLoss()), ('function', fn)):
        model = rvip.get_model(_cfg(LOSS_FUNCTION=loss), metrics=[])
        eng = model._engine(4)
        eng.load_input(x, y)
        eng.forward(training=True)
        eng.backward()
        torch.cuda.synchronize()
        out[name] = (float(eng.loss.item()), model._params.grads_host(), eng.grad_factor)
    assert out['class'][2] == 4 * 32 * 32 and out['function'][2] == 1.0
    assert out['class'][0] == out['function'][0]
    for k, g in out['function'][1].items():
        np.testing.assert_allclose(out['class'][1][k], g * 4096.0, rtol=2e-5, atol=1e-6 * float(np.abs(g).max()) * 4096.0, err_msg=str(k))


def test_fit_logs_with_a_ragged_last_batch():
    """A user Sequence whose last batch is smaller: the epoch's logged loss is the mean of the per-batch losses, each over ITS
    element count (Keras' fit semantics; the logs drive ModelCheckpoint / ReduceLROnPlateau / EarlyStopping, KerasCallbacks.py:54-98)."""
    cfg = _cfg(DIM=[32, 32], FILTERS=8)
    x, y = O.synthetic_batch(10, cfg['DIM'], 2, seed=8)

    class Seq:
        def __len__(self):
            return 3

        def __getitem__(self, i):
            sl = slice(4 * i, min(4 * i + 4, 10))
            return x[sl], y[sl]
    a, b = rvip.get_model(cfg, metrics=[]), rvip.get_model(cfg, metrics=[])
    b.set_weights(a.get_weights())
    hist = a.fit(Seq(), epochs=1, shuffle=False, verbose=0)
    ref = [b.train_on_batch(*Seq()[i])[0] for i in range(3)]
    assert abs(hist.history['loss'][0] - float(np.mean(ref))) < 1e-6, (hist.history['loss'], ref)


def test_zz_tolerance_branch_report():
    """Dumps which gradient bound admitted every (variant, step, tensor) of test_fp32_training_steps_match_oracle."""
    if not BRANCH_REPORT:
        pytest.skip('the family did not run in this session')
    import json
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'r03_tolerance_branches.json'), 'w') as f:
        json.dump(BRANCH_REPORT, f, indent=1)
    tot = {k: sum(v[k] for v in BRANCH_REPORT.values()) for k in ('tight', 'f32', 'knife', 'clean_steps', 'clean_tight', 'clean_total',
                                                                  'landmark_near_ties', 'mask_near_threshold', 'landmarks_total')}
    print('gradient tolerance branches:', tot)
    assert tot['clean_steps'] >= 3 and tot['clean_tight'] >= 0.9 * tot['clean_total']


def _assert_landmarks_and_masks(pg, pr, eps=2e-5):
    """argmax indices and >0.5 masks must be IDENTICAL, except where the reference itself is within `eps` (fp32
    rounding of a different summation order) of a tie / of the threshold."""
    if pr.ndim == 5:                                           # volumes: frame by frame
        pg, pr = pg.reshape((-1,) + pg.shape[2:]), pr.reshape((-1,) + pr.shape[2:])
    pr32 = pr.astype(np.float32)
    ig, ir = O.landmark_argmax(pg), O.landmark_argmax(pr32)
    n, h, w, c = pr.shape
    flat = pr.transpose(0, 3, 1, 2).reshape(n, c, h * w)
    for (i, k) in zip(*np.where(ig != ir)):
        assert abs(flat[i, k, ig[i, k]] - flat[i, k, ir[i, k]]) < eps, ('argmax', i, k, ig[i, k], ir[i, k])
    bad = O.threshold_mask(pg) != O.threshold_mask(pr)
    assert (np.abs(pr[bad] - 0.5) < eps).all(), ('mask', int(bad.sum()))
    return int((ig != ir).sum()), int(bad.sum())


def test_bf16_path_matches_bf16_storage_emulation():
    """bf16 device path vs the oracle with bf16 rounding applied at every tensor the device materialises in bf16
    (fp32 accumulation everywhere): the two differ only by summation order, i.e. by rare 1-ulp bf16 flips."""
    cfg = _cfg(RVIP_PRECISION='bf16', FILTERS=16, DIM=[64, 64])
    model = rvip.get_model(cfg, metrics=[])
    _, layers = _oracle_from(model, cfg)
    params = _oracle_from(model, cfg)[0].params
    emu = O.OracleUNet(cfg, params, dtype=np.float64, quant=O.bf16_round)
    exact = O.OracleUNet(cfg, params, dtype=np.float64)
    B = 4
    x, y = O.synthetic_batch(B, cfg['DIM'], 2, seed=5)
    eng = model._engine(B)
    eng.load_input(x, y)
    eng.forward(training=True)
    eng.backward()
    torch.cuda.synchronize()
    masks = _masks(layers, B, model.seed, 0)
    lv, egrads, epred, _ = emu.loss_and_grads(x.astype(np.float64), y.astype(np.float64), 'mse', masks)
    _, xgrads, xpred, _ = exact.loss_and_grads(x.astype(np.float64), y.astype(np.float64), 'mse', masks)
    pred = eng.pred.cpu().numpy()
    err_emu, err_exact = np.abs(pred - epred), np.abs(pred - xpred)
    # Residual vs the emulation = 1-ulp bf16 flips where the fp32 accumulation order moves a value across a rounding
    # boundary (a fraction of a percent of the elements per tensor); vs the exact oracle EVERY element carries 2^-9.
    assert abs(float(eng.loss.item()) - lv) < 5e-3 * lv
    assert err_emu.mean() < 5e-3 and err_emu.max() < 0.1, (err_emu.mean(), err_emu.max())
    assert err_emu.mean() < 0.7 * err_exact.mean(), (err_emu.mean(), err_exact.mean())
    got = model._params.grads_host()
    for lname in ('conv2d', 'conv2d_1', 'conv2d_3', 'conv2d_5', 'conv2d_8', 'unet'):
        g = egrads[lname][0]
        rel = np.linalg.norm(got[(lname, 'kernel')] - g) / np.linalg.norm(g)
        rel_x = np.linalg.norm(got[(lname, 'kernel')] - xgrads[lname][0]) / np.linalg.norm(xgrads[lname][0])
        assert rel < 0.7 * rel_x + 0.02 and rel < 0.25, (lname, rel, rel_x)


def test_bf16_path_at_the_real_channel_widths_matches_bf16_storage_emulation():
    """The same comparison at config 2's channel schedule (F=32, depth 4: 32 ... 512 channels, the (256+256) -> 256 concat conv, the
    512 -> 256 up-conv, split-K weight gradients with the real slab sizes) on 128 x 128 slices (8 x 8 bottleneck), where the float64 oracle is still
    affordable: end to end, every kernel variant the benchmark launches takes part except the 256^2-only tile counts."""
    cfg = _cfg(RVIP_PRECISION='bf16', FILTERS=32, DEPTH=4, DIM=[128, 128])
    model = rvip.get_model(cfg, metrics=[])
    _, layers = _oracle_from(model, cfg)
    params = _oracle_from(model, cfg)[0].params
    emu = O.OracleUNet(cfg, params, dtype=np.float64, quant=O.bf16_round)
    exact = O.OracleUNet(cfg, params, dtype=np.float64)
    B = 4
    x, y = O.synthetic_batch(B, cfg['DIM'], 2, seed=5)
    eng = model._engine(B)
    eng.load_input(x, y)
    eng.forward(training=True)
    eng.backward()
    torch.cuda.synchronize()
    masks = _masks(layers, B, model.seed, 0)
    lv, egrads, epred, _ = emu.loss_and_grads(x.astype(np.float64), y.astype(np.float64), 'mse', masks)
    _, xgrads, xpred, _ = exact.loss_and_grads(x.astype(np.float64), y.astype(np.float64), 'mse', masks)
    pred = eng.pred.cpu().numpy()
    err_emu, err_exact = np.abs(pred - epred), np.abs(pred - xpred)
    print('loss dev %.6f emu %.6f; |pred - emu| mean %.2e max %.2e; |pred - exact| mean %.2e max %.2e' % (
        float(eng.loss.item()), lv, err_emu.mean(), err_emu.max(), err_exact.mean(), err_exact.max()))
    # (4 x 4 bottleneck at batch 4: BN normalises over 64 values per channel there, which amplifies every 1-ulp bf16 flip)
    assert abs(float(eng.loss.item()) - lv) < 1e-2 * lv
    assert err_emu.mean() < 2e-2 and err_emu.max() < 0.3, (err_emu.mean(), err_emu.max())
    assert err_emu.mean() < 0.8 * err_exact.mean(), (err_emu.mean(), err_exact.mean())
    got = model._params.grads_host()
    names = [l['name'] for l in layers if l['type'] == 'Conv2D']
    rels = {}
    for lname in names + ['unet']:
        g = egrads[lname][0]
        rel = np.linalg.norm(got[(lname, 'kernel')] - g) / np.linalg.norm(g)
        rel_x = np.linalg.norm(got[(lname, 'kernel')] - xgrads[lname][0]) / np.linalg.norm(xgrads[lname][0])
        rel_ex = np.linalg.norm(g - xgrads[lname][0]) / np.linalg.norm(xgrads[lname][0])
        rels[lname] = (round(float(rel), 3), round(float(rel_x), 3), round(float(rel_ex), 3))
    print('kernel-gradient relative errors (device vs emulation, device vs exact, emulation vs exact):', rels)
    for lname, (rel, rel_x, rel_ex) in rels.items():
        # At initialisation the loss gradient is almost common-mode (sigmoid outputs ~0.5 everywhere) and every BN backward cancels that
        # part, so the rounding of the stored ACTIVATIONS (profiles/r03_gradient_fidelity.txt) leaves an error as large as the gradient itself in the early layers
        # (emulation vs exact ~1.0 from conv2d_1 to conv2d_13, 0.03 at the last conv).  What the kernels can be held to: the device
        # sits closer to the emulation than the emulation's own rounding noise is large, and closer to it than to the exact oracle.
        assert rel < 0.8 * rel_x + 0.03 and rel < 0.9 * rel_ex + 0.03, (lname, rel, rel_x, rel_ex)


FIDELITY = {}


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
def test_gradient_fidelity_at_the_real_channel_widths(precision):
    """Per-layer relative error of the kernel gradients against the EXACT float64 oracle at config 2's channel schedule (F=32, depth 4,
    128 x 128, first step from the initialisation), for the two 16-bit storage types.  At initialisation the network's output is almost
    constant, the loss gradient almost common-mode, every BN backward cancels that part, and the rest is smaller than the rounding noise
    of the stored ACTIVATIONS (profiles/r03_gradient_fidelity.txt: the gradient tensors' rounding is 1-2 % of the error; it fades over
    the first tens of steps).  The table goes to gpurun_out/r03_gradient_fidelity.json; asserted: fp16's error is the smaller one."""
    cfg = _cfg(RVIP_PRECISION=precision, FILTERS=32, DEPTH=4, DIM=[128, 128])
    model = rvip.get_model(cfg, metrics=[])
    exact, layers = _oracle_from(model, cfg)
    B = 4
    x, y = O.synthetic_batch(B, cfg['DIM'], 2, seed=5)
    eng = model._engine(B)
    eng.load_input(x, y)
    eng.forward(training=True)
    eng.backward()
    torch.cuda.synchronize()
    masks = _masks(layers, B, model.seed, 0)
    _, xgrads, _, _ = exact.loss_and_grads(x.astype(np.float64), y.astype(np.float64), 'mse', masks)
    got = model._params.grads_host()
    names = [l['name'] for l in layers if l['type'] == 'Conv2D']
    rel = {n_: round(float(np.linalg.norm(got[(n_, 'kernel')] - xgrads[n_][0]) / np.linalg.norm(xgrads[n_][0])), 4) for n_ in names}
    FIDELITY[precision] = rel
    print(precision, 'kernel-gradient relative error vs the exact oracle:', rel)
    if len(FIDELITY) == 2:
        import json
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'r03_gradient_fidelity.json'), 'w') as f:
            json.dump(FIDELITY, f, indent=1)
        med = lambda d: float(np.median(list(d.values())))                     # noqa: E731
        assert med(FIDELITY['fp16']) < med(FIDELITY['bf16']), (med(FIDELITY['fp16']), med(FIDELITY['bf16']))


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_backward_routes_agree(precision, monkeypatch):
    """One training step's gradients through the three backward schedules the engine can build: the default (BN backward from the
    consumers' by-products, Dropout / up-conv ReLU backward as bit-plane gates), the same with every channel block forced down the
    exact in-kernel route of rvip_bn_bwd_coef (RVIP_BNBWD_MIN_GAMMA huge), and round 2's schedule (RVIP_BNBWD_ALGEBRAIC=0: reduction
    pass, separate apply pass for the up-convs).  Same weights, same batch, same dropout stream: fp32 agrees to summation order,
    bf16 to its storage rounding."""
    cfg = _cfg(RVIP_PRECISION=precision, FILTERS=32, DEPTH=3, DIM=[64, 64])
    x, y = O.synthetic_batch(4, cfg['DIM'], 2, seed=12)
    got = {}
    for route, env in (('default', {}), ('exact', {'RVIP_BNBWD_MIN_GAMMA': '1e9'}), ('round2', {'RVIP_BNBWD_ALGEBRAIC': '0'})):
        for k in ('RVIP_BNBWD_MIN_GAMMA', 'RVIP_BNBWD_ALGEBRAIC'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        model = rvip.get_model(cfg, metrics=[])
        eng = model._engine(4)
        assert bool(eng.algebraic) == (route != 'round2') and bool(eng.upact) == (route != 'round2')
        eng.load_input(x, y)
        eng.forward(training=True)
        eng.backward()
        torch.cuda.synchronize()
        flags = [int(f.sum().item()) for f in eng.bn_flags.values()]
        assert (all(flags) if route == 'exact' else not any(flags)), (route, flags)
        got[route] = (float(eng.loss.item()), model._params.grads_host())
    assert got['default'][0] == got['exact'][0] == got['round2'][0]                 # the forward pass is the same launch list
    tol = 2e-4 if precision == 'fp32' else 0.06
    for other in ('exact', 'round2'):
        for k, g in got['default'][1].items():
            ref = got[other][1][k]
            scale = float(np.abs(ref).max())
            if scale < 1e-12:
                continue
            assert np.abs(g - ref).max() <= tol * scale + 1e-9, (precision, other, k, float(np.abs(g - ref).max()), scale)


@pytest.mark.parametrize('targets', ['heatmaps', 'zeros'])
def test_bn_backward_in_front_of_subpixel_layers_with_trained_like_parameters(monkeypatch, targets):
    """the stage in front of an UpSampling2D -> conv layer.  That layer's data gradient runs in sub-pixel form
    (phase kernels = summed taps rounded once), so the algebraic BN backward of the stage would mix  <Wr, dW>  (nine-tap weights) with
    column sums of a gradient made from OTHER weights: an error of 2^-9 |beta / gamma| |sum g| in dgamma that every other test misses
    because beta = 0 at initialisation.  Here gamma, beta are trained-like (|beta| up to 8 |gamma|) and the loss gradient is the
    initialisation's (almost common-mode: |sum g| >> |sum g xhat|, the worst case).  Reference = round 2's schedule (reduction pass
    over the g and z the apply pass uses) on the bit-identical forward pass.  The default ('exact' for these stages) and 'ninetap'
    must sit on it; round 3's behaviour ('algebraic') is measured and reported beside them."""
    cfg = _cfg(RVIP_PRECISION='bf16', FILTERS=32, DEPTH=3, DIM=[64, 64], DROPOUT_MIN=0.0, DROPOUT_MAX=0.0)
    x, y = O.synthetic_batch(4, cfg['DIM'], 2, seed=12)
    if targets == 'zeros':
        y = np.zeros_like(y)            # 2 (p - 0) p (1 - p) > 0 everywhere: the gradient is as common-mode as it gets (|sum g| >> |sum g xhat|)
    base = rvip.get_model(cfg, metrics=[])
    names = base.weight_names()
    rng = np.random.default_rng(3)
    w = base.get_weights()
    for i, nm in enumerate(names):
        if nm.endswith('/gamma:0'):
            w[i] = (rng.uniform(0.5, 1.5, w[i].shape) * rng.choice([-1.0, 1.0], w[i].shape)).astype(np.float32)
            g_ = w[i]
        elif nm.endswith('/beta:0'):
            w[i] = (g_ * rng.uniform(1.0, 8.0, w[i].shape) * rng.choice([-1.0, 1.0], w[i].shape)).astype(np.float32)
    got, fronts = {}, None
    for route, env in (('round2', {'RVIP_BNBWD_ALGEBRAIC': '0'}), ('exact', {'RVIP_BNBWD_SUBPIX_CONSUMER': 'exact'}), ('ninetap', {'RVIP_BNBWD_SUBPIX_CONSUMER': 'ninetap'}),
                       ('algebraic', {'RVIP_BNBWD_SUBPIX_CONSUMER': 'algebraic'}), ('auto', {})):
        for k in ('RVIP_BNBWD_ALGEBRAIC', 'RVIP_BNBWD_SUBPIX_CONSUMER'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        model = rvip.get_model(cfg, metrics=[])
        model.set_weights(w)
        eng = model._engine(4)
        stages = model.plan.stages
        by_y = {st.y: st for st in stages}
        fronts = [by_y[st.src0] for st in stages if st.up0 == 1 and not st.src1 and st.src0 in by_y]
        assert len(fronts) == 3 and all(f.bn for f in fronts)
        if route == 'exact':
            assert not any(f.conv in eng.algebraic for f in fronts) and eng.algebraic           # the other stages keep the algebraic route
        elif route in ('ninetap', 'algebraic'):
            assert all(f.conv in eng.algebraic for f in fronts)
        elif route == 'auto':                      # 'phase' (consistent dot rows) where the weight gradient runs in the four-phase form
            assert 'phase' in eng.sp_modes.values() and all(f.conv in eng.algebraic for f in fronts), eng.sp_modes
            modes = dict(eng.sp_modes)
        eng.load_input(x, y)
        eng.forward(training=True)
        eng.backward()
        torch.cuda.synchronize()
        assert not any(int(f.sum().item()) for f in eng.bn_flags.values())                      # no block took the in-kernel exact route
        got[route] = (float(eng.loss.item()), model._params.grads_host())
    assert len({v[0] for v in got.values()}) == 1                                               # one and the same forward pass
    report = {}
    for f in fronts:
        ref = got['round2'][1][(f.bn, 'gamma')].astype(np.float64)
        scale = float(np.abs(ref).max())
        report[f.bn] = {r: float(np.abs(got[r][1][(f.bn, 'gamma')] - ref).max() / scale) for r in ('exact', 'ninetap', 'algebraic', 'auto')}
        big = np.abs(ref) >= 0.05 * scale                                   # per-channel relative deviation where dgamma is not itself noise
        for r in ('exact', 'ninetap', 'algebraic', 'auto'):
            rel = np.abs(got[r][1][(f.bn, 'gamma')] - ref)[big] / np.abs(ref)[big]
            report[f.bn][r + '_rel_median_p95'] = [float(np.median(rel)), float(np.percentile(rel, 95))]
        # how common-mode the gradient reaching the stage is: |sum g| / |sum g xhat| per channel (dbeta / dgamma), median
        report[f.bn]['dbeta_over_dgamma_median'] = float(np.median(np.abs(got['round2'][1][(f.bn, 'beta')]) / (np.abs(ref) + 1e-30)))
    print('dgamma of the stages in front of sub-pixel layers, max deviation from the reduction-pass schedule / max |dgamma|:', report, modes)
    for f in fronts:
        # 'exact' differs from round 2 only downstream of the OTHER stages' algebraic sums (bf16 storage noise), 'ninetap' likewise
        assert report[f.bn]['exact'] <= 0.02 and report[f.bn]['ninetap'] <= 0.02, report
        assert report[f.bn]['algebraic'] <= 0.03 and report[f.bn]['auto'] <= 0.02, report
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    import json
    with open(os.path.join(ROOT, 'gpurun_out', 'r04_subpixel_consumer_dgamma_%s.json' % targets), 'w') as fh:
        json.dump(report, fh, indent=1)


TRAINED_REPORT = {}


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
def test_trained_state_gradients_against_the_storage_emulation(precision):
    """the 16-bit gradients of the DEFAULT backward schedule (algebraic BN backward, bit-plane gates, sub-pixel
    consumers: `auto`) against the ORACLE with the same storage rounding, at a TRAINED state -- gamma != 1, beta != 0, logits with
    real spatial structure -- instead of the initialisation, where the loss gradient is common-mode and the emulation's own rounding
    noise is as large as the gradient (test_bf16_path_at_the_real_channel_widths_...).  The state comes from 60 Adam steps of the
    fp32 device path on the batch (any weights are a valid input; that path is held to the float64 oracle elsewhere).  Reference
    semantics: conv -> activation -> BatchNormalization -> Dropout (KerasLayers.py:684,691), TF's fused batch-norm backward.
    Measured: emulation - exact = 0.13 median / 0.38 max of |exact| in bf16 (0.04 / 0.11 in fp16) -- bf16 activations cost that much
    even at a trained state (profiles/r03_gradient_fidelity.txt), so 'emulation noise < 0.05' is not a state this graph has -- and
    device - emulation = 0.06 median / 0.20 max (0.02 / 0.05).
    Every trainable tensor: the device sits within `DEV_OVER_EMU` x the emulation's own distance from the exact float64 oracle -- of
    the emulation AND of the exact gradient (both round the same tensors; the device differs by summation order, i.e. rare one-ulp
    flips, and by the algebraic form of the BN backward, which is what this bounds) -- plus a floor for the tensors whose emulation
    noise is itself tiny.  (VERDICT asked for 2 x; a bound relative to the device-emulation distance ALONE is not stable: at a
    second, equally valid trained state the device is the closer of the two to the exact gradient and 0.9 x the noise away from
    the emulation.)"""
    # measured (MI355X, round 5) over two trained states (the second: the same recipe under another split-K partition of the fp32 steps):
    # device - emulation at most 1.08 x the emulation's noise, device - exact at most 1.38 x (first layer's bias; the device is usually the CLOSER of the two)
    DEV_OVER_EMU, FLOOR = 1.5, 0.02
    cfg = _cfg(FILTERS=32, DEPTH=3, DIM=[64, 64])
    B = 4
    x, y = O.synthetic_batch(B, cfg['DIM'], 2, seed=12)
    trainer = rvip.get_model(dict(cfg, RVIP_PRECISION='fp32'), metrics=[])
    for _ in range(60):
        trainer.train_on_batch(x, y)
    w = trainer.get_weights()
    trainer.close()
    names = trainer.weight_names()
    dev_gb = [float(np.abs(a - 1.0).max()) if n_.endswith('/gamma:0') else float(np.abs(a).max()) for n_, a in zip(names, w) if n_.endswith(('/gamma:0', '/beta:0'))]
    assert max(dev_gb) > 0.02, 'the state is not a trained one (gamma, beta still at their initial values)'
    model = rvip.get_model(dict(cfg, RVIP_PRECISION=precision), metrics=[])
    model.set_weights(w)
    exact, layers = _oracle_from(model, cfg)
    eng = model._engine(B)
    assert eng.algebraic and eng.keep_bits, 'the default backward schedule (algebraic BN backward, keep-bit gates) is what this test is about'
    q = O.bf16_round if precision == 'bf16' else O.f16_round
    qg = q if precision == 'bf16' else (lambda a_: O.f16_round(a_, eng.loss_scale))      # f16 gradient tensors are stored at the loss scale
    emu = O.OracleUNet(cfg, exact.params, dtype=np.float64, quant=q, quant_grad=qg)
    eng.load_input(x, y)
    eng.forward(training=True)
    eng.backward()
    torch.cuda.synchronize()
    assert not any(int(f.sum().item()) for f in eng.bn_flags.values())          # no channel block took the in-kernel exact route
    masks = _masks(layers, B, model.seed, 0)
    lv, egrads, epred, _ = emu.loss_and_grads(x.astype(np.float64), y.astype(np.float64), 'mse', masks)
    _, xgrads, _, _ = exact.loss_and_grads(x.astype(np.float64), y.astype(np.float64), 'mse', masks)
    assert abs(float(eng.loss.item()) - lv) < 5e-3 * lv
    got = model._params.grads_host()
    table, bad = {}, []
    for lname, gs in egrads.items():
        kinds = ('kernel', 'bias') if lname.startswith(('conv', 'unet')) else ('gamma', 'beta')
        for kind, ge, gx in zip(kinds, gs, xgrads[lname]):
            gd = got[(lname, kind)].astype(np.float64)
            nx = np.linalg.norm(gx) + 1e-300
            dev_emu, emu_x, dev_x = (float(np.linalg.norm(gd - ge) / nx), float(np.linalg.norm(ge - gx) / nx), float(np.linalg.norm(gd - gx) / nx))
            table['%s/%s' % (lname, kind)] = (round(dev_emu, 4), round(emu_x, 4), round(dev_x, 4))
            if dev_emu > DEV_OVER_EMU * emu_x + FLOOR or dev_x > DEV_OVER_EMU * emu_x + FLOOR:
                bad.append((lname, kind, dev_emu, emu_x, dev_x))
    TRAINED_REPORT[precision] = table
    print(precision, 'trained state: (device - emulation, emulation - exact, device - exact) / |exact| per tensor:', table)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    import json
    with open(os.path.join(ROOT, 'gpurun_out', 'r05_trained_state_gradients_%s.json' % precision), 'w') as fh:
        json.dump(table, fh, indent=1)
    assert not bad, bad


def test_full_size_step_is_deterministic_and_finite():
    """BASELINE.json configs[1] shape (256x256, F=32, depth 4, batch 32, bf16): size-independent properties --
    two identical steps from identical state give bit-identical loss, heat-maps and gradients; a further step
    lowers nothing to NaN."""
    cfg = dict(DIM=[256, 256], FILTERS=32, DEPTH=4, BATCH_NORMALISATION=True, ACTIVATION='relu', MASK_CLASSES=2,
               LEARNING_RATE=1e-4, RVIP_PRECISION='bf16', LOSS_FUNCTION=M.mse, SEED=1)
    model = rvip.get_model(cfg, metrics=[])
    G = rvip.Generators.SyntheticSAXGenerator(32, dict(DIM=[256, 256], BATCHSIZE=32, GAUS=True, SIGMA=2, SHUFFLE=False))
    x, y = G[0]
    eng = model._engine(32)
    outs = []
    for _ in range(2):
        eng.load_input(x, y)
        eng.forward(training=True)
        eng.backward()
        torch.cuda.synchronize()
        outs.append((eng.loss.clone(), eng.pred.clone(), model._params.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    assert torch.isfinite(outs[0][2]).all() and float(outs[0][2].abs().max()) > 0
    l0 = model.train_on_batch(x, y)[0]
    l1 = model.train_on_batch(x, y)[0]
    assert np.isfinite(l0) and np.isfinite(l1)


def test_cfg4_shape_step_is_deterministic_and_finite():
    """BASELINE.json configs[3] shape (512x512, F=64, depth 5, fp16 MFMA path with static loss scaling, batch 8):
    parameter count of SURVEY.md 8(d) (138 376 578), bit-identical repeat of fwd+bwd from identical state, finite
    gradients well inside the f16 range, finite and decreasing loss over optimizer steps."""
    cfg = dict(DIM=[512, 512], FILTERS=64, DEPTH=5, BATCH_NORMALISATION=True, ACTIVATION='relu', MASK_CLASSES=2,
               LEARNING_RATE=1e-4, RVIP_PRECISION='fp16', LOSS_FUNCTION=M.mse, SEED=1)
    model = rvip.get_model(cfg, metrics=[])
    assert model.count_params() == 138376578
    G = rvip.Generators.SyntheticSAXGenerator(8, dict(DIM=[512, 512], BATCHSIZE=8, GAUS=True, SIGMA=2, SHUFFLE=False))
    x, y = G[0]
    eng = model._engine(8)
    outs = []
    for _ in range(2):
        eng.load_input(x, y)
        eng.forward(training=True)
        eng.backward()
        torch.cuda.synchronize()
        outs.append((eng.loss.clone(), eng.pred.clone(), model._params.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    assert torch.isfinite(outs[0][2]).all() and float(outs[0][2].abs().max()) > 0
    # the scaled activation gradients the f16 tensors carry sit in the middle of the f16 range (tools/diag_f16_range.py)
    gmax = max(float(t.float().abs().max()) for t in eng.grd.values() if t is not None and t.dtype == torch.float16)
    assert eng.loss_scale == 2.0 ** 22 and 1e-2 < gmax < 1e3, gmax
    ls = [model.train_on_batch(x, y)[0] for _ in range(4)]
    assert np.all(np.isfinite(ls)) and ls[-1] < ls[0], ls


def test_cfg5_full_size_step_is_deterministic_and_finite():
    """BASELINE.json configs[4] at its real size (16 x 256 x 256 cine volumes, Conv3D 3x3x3, pool (1,2,2), F=32, depth 4, batch 4 per
    GPU, bf16): parameter count of SURVEY.md 8(d) (25 894 658), bit-identical repeat of fwd+bwd from identical state, finite non-zero
    gradients, volumes of the batch do not leak into each other (a changed volume changes only its own heat-maps in inference), and
    the loss falls over optimizer steps through the captured step."""
    cfg = dict(DIM=[16, 256, 256], M_POOL=[1, 2, 2], F_SIZE=[3, 3, 3], FILTERS=32, DEPTH=4, BATCH_NORMALISATION=True, ACTIVATION='relu',
               MASK_CLASSES=2, LEARNING_RATE=1e-4, RVIP_PRECISION='bf16', LOSS_FUNCTION=M.mse, SEED=1)
    model = rvip.get_model(cfg, metrics=[])
    assert model.count_params() == 25894658
    G = rvip.Generators.SyntheticSAXGenerator(4, dict(DIM=cfg['DIM'], BATCHSIZE=4, GAUS=True, SIGMA=2, SHUFFLE=False))
    x, y = G[0]
    assert x.shape == (4, 16, 256, 256, 1) and y.shape == (4, 16, 256, 256, 2)
    eng = model._engine(4)
    outs = []
    for _ in range(2):
        eng.load_input(x, y)
        eng.forward(training=True)
        eng.backward()
        torch.cuda.synchronize()
        outs.append((eng.loss.clone(), eng.pred.clone(), model._params.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    assert torch.isfinite(outs[0][2]).all() and float(outs[0][2].abs().max()) > 0
    p0 = model.predict_on_batch(x)
    x2 = x.copy()
    x2[2] = x[2][::-1]                                               # volume 2 played backwards
    p1 = model.predict_on_batch(x2)
    assert p0.shape == (4, 16, 256, 256, 2) and np.isfinite(p0).all()
    for b in (0, 1, 3):
        np.testing.assert_array_equal(p0[b], p1[b])                  # inference: no cross-volume coupling (BN on the moving statistics)
    assert not np.array_equal(p0[2], p1[2])
    ls = [model.train_on_batch(x, y)[0] for _ in range(4)]
    assert eng.launch_mode == 'hipGraph' and np.all(np.isfinite(ls)) and ls[-1] < ls[0], ls


def test_f16_path_matches_f16_storage_emulation():
    """f16 device path (RVIP_PRECISION='fp16': IEEE half activations / packed weights, fp32 accumulation and master weights,
    static loss scale) vs the oracle with binary16 rounding applied at every tensor the device materialises in f16 --
    gradient tensors rounded at the loss scale.  Differences: summation order only (rare 1-ulp flips); against the exact
    oracle every element carries 2^-12."""
    cfg = _cfg(RVIP_PRECISION='fp16', FILTERS=16, DIM=[64, 64])
    model = rvip.get_model(cfg, metrics=[])
    _, layers = _oracle_from(model, cfg)
    params = _oracle_from(model, cfg)[0].params
    B = 4
    eng = model._engine(B)
    S = eng.loss_scale
    assert S == 2.0 ** 15                               # 2^floor(log2(4*64*64*2))
    emu = O.OracleUNet(cfg, params, dtype=np.float64, quant=O.f16_round, quant_grad=lambda a: O.f16_round(a, S))
    exact = O.OracleUNet(cfg, params, dtype=np.float64)
    x, y = O.synthetic_batch(B, cfg['DIM'], 2, seed=5)
    eng.load_input(x, y)
    eng.forward(training=True)
    eng.backward()
    torch.cuda.synchronize()
    masks = _masks(layers, B, model.seed, 0)
    lv, egrads, epred, _ = emu.loss_and_grads(x.astype(np.float64), y.astype(np.float64), 'mse', masks)
    _, xgrads, xpred, _ = exact.loss_and_grads(x.astype(np.float64), y.astype(np.float64), 'mse', masks)
    pred = eng.pred.cpu().numpy()
    err_emu, err_exact = np.abs(pred - epred), np.abs(pred - xpred)
    assert abs(float(eng.loss.item()) - lv) < 1e-3 * lv
    assert err_emu.mean() < 1e-3 and err_emu.max() < 0.05, (err_emu.mean(), err_emu.max())
    assert err_emu.mean() < 0.7 * err_exact.mean(), (err_emu.mean(), err_exact.mean())
    got = model._params.grads_host()                    # unscaled by ParamStore.grad_unscale
    for lname in ('conv2d', 'conv2d_1', 'conv2d_3', 'conv2d_5', 'conv2d_8', 'unet'):
        rel = np.linalg.norm(got[(lname, 'kernel')] - egrads[lname][0]) / np.linalg.norm(egrads[lname][0])
        rel_x = np.linalg.norm(got[(lname, 'kernel')] - xgrads[lname][0]) / np.linalg.norm(xgrads[lname][0])
        assert rel < 0.7 * rel_x + 0.02 and rel < 0.1, (lname, rel, rel_x)     # rel_x: rounding moves ReLU masks / max-pool winners
    # one optimizer step: Adam sees the UNSCALED gradient.  First Keras-Adam step: dtheta = -lr * g / (|g| + eps / sqrt(1 - beta2))
    # -- a gradient still carrying the loss scale would give -lr * sign(g) everywhere.
    w0 = model.get_weights()[0].copy()
    model.train_on_batch(x, y)
    dw = model.get_weights()[0] - w0
    g0 = got[('conv2d', 'kernel')].astype(np.float64)               # the device's own gradient of that step, unscaled
    want = -cfg['LEARNING_RATE'] * g0 / (np.abs(g0) + 1e-7 / np.sqrt(1e-3))
    assert (np.abs(want) < 0.9 * cfg['LEARNING_RATE']).sum() > 5     # some elements are not saturated at +-lr
    np.testing.assert_allclose(dw, want, rtol=2e-3, atol=1e-7)


def test_f16_bce_dice_loss_curve_tracks_the_float64_oracle():
    """The f16 path with the reference's other loss (bce_dice_loss: the dice term's gradient is not a per-pixel 1/count
    quantity, the static loss scale multiplies the whole dlogit tensor): ragged 48 x 40 maps, depth 3, eight Adam steps."""
    cfg = _cfg(RVIP_PRECISION='fp16', FILTERS=16, DEPTH=3, DIM=[48, 40], LEARNING_RATE=1e-3, LOSS_FUNCTION=M.bce_dice_loss)
    model = rvip.get_model(cfg, metrics=[])
    ref, layers = _oracle_from(model, cfg)
    B = 4
    x, y = O.synthetic_batch(B, cfg['DIM'], 2, seed=21)
    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    dev_losses, ref_losses = [], []
    for step in range(8):
        dev_losses.append(model.train_on_batch(x, y)[0])
        lv, _ = ref.train_step(x64, y64, 'bce_dice', _masks(layers, B, model.seed, step))
        ref_losses.append(lv)
    np.testing.assert_allclose(np.array(dev_losses), np.array(ref_losses), rtol=2e-2, atol=2e-3)
    assert ref_losses[-1] < ref_losses[0]


def test_f16_loss_curve_tracks_the_float64_oracle():
    """Fifteen Adam steps, f16 device path vs the float64 oracle (same start, same dropout masks): same loss curve."""
    cfg = _cfg(RVIP_PRECISION='fp16', FILTERS=16, DIM=[64, 64], LEARNING_RATE=1e-3)
    model = rvip.get_model(cfg, metrics=[])
    ref, layers = _oracle_from(model, cfg)
    B = 4
    x, y = O.synthetic_batch(B, cfg['DIM'], 2, seed=12)
    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    dev_losses, ref_losses = [], []
    for step in range(15):
        dev_losses.append(model.train_on_batch(x, y)[0])
        lv, _ = ref.train_step(x64, y64, 'mse', _masks(layers, B, model.seed, step))
        ref_losses.append(lv)
    np.testing.assert_allclose(np.array(dev_losses), np.array(ref_losses), rtol=1e-2)


def test_capture_survives_garbage_models_and_collections(monkeypatch):
    """GPUTEST_r03's abort, turned into a test.  A model that died inside a reference cycle still owns its hipGraphs; if the cyclic
    collector frees it while another engine captures, at::cuda::CUDAGraph::~CUDAGraph synchronises the device, fails, throws from
    the destructor -> SIGABRT (tools/repro_graph_gc_abort.py shows that on the unguarded capture).  Engine.capture() collects before
    it captures and keeps the automatic collector off until the capture has ended; models no longer sit in cycles at all."""
    import gc
    import weakref
    cfg = _cfg(RVIP_PRECISION='bf16', DIM=[32, 32], FILTERS=8)
    x, y = O.synthetic_batch(4, cfg['DIM'], 2, seed=3)
    a = rvip.get_model(cfg, metrics=[])
    for _ in range(3):
        a.train_on_batch(x, y)
    assert a._engine(4).launch_mode == 'hipGraph'
    ref = weakref.ref(a)
    was = gc.isenabled()
    gc.disable()
    try:
        del a
        assert ref() is None, 'a trained model must die by reference count (no cycle through optimizer.lr / fit callbacks)'
        g = rvip.get_model(cfg, metrics=[])                 # a model some user code DID tie into a cycle
        for _ in range(3):
            g.train_on_batch(x, y)
        g._user_cycle = g
        gref = weakref.ref(g)
        del g
        assert gref() is not None                           # garbage, graphs and all, waiting for the collector
    finally:
        if was:
            gc.enable()
    b = rvip.get_model(cfg, metrics=[])
    b.train_on_batch(x, y)
    eng = b._engine(4)
    orig, seen = eng._step_parts, {}

    def spying():
        parts, buckets = orig()

        def first():
            seen['gc_enabled_inside_capture'] = gc.isenabled()
            seen['garbage_alive_inside_capture'] = gref() is not None
            parts[0]()
        return [first] + list(parts[1:]), buckets
    monkeypatch.setattr(eng, '_step_parts', spying)
    b.train_on_batch(x, y)                                  # captures: collects first (frees g legally), collector off while capturing
    assert eng.launch_mode == 'hipGraph'
    assert seen == {'gc_enabled_inside_capture': False, 'garbage_alive_inside_capture': False}, seen
    assert gc.isenabled() == was
    monkeypatch.undo()
    l3 = b.train_on_batch(x, y)[0]
    assert np.isfinite(l3)
    b.close()                                               # explicit release: graphs, ring and parameter blocks go now
    assert b._engines == {} and b._params is None
    assert np.isfinite(b.train_on_batch(x, y)[0])           # ... and the model rebuilds its device state from the weights it kept


def test_fit_under_garbage_collection_pressure():
    """The conditions of GPUTEST_r03's abort, made as hostile as they get: models that die inside USER-made reference cycles (graphs and
    all), a thread that churns cyclic garbage so that the collector runs all the time on whatever thread allocates, generator pool threads,
    and one fit() after the other each capturing its step.  With an unguarded capture some collection lands inside a capture window
    sooner or later and the process aborts (tools/repro_graph_gc_abort.py); with Engine.capture's guard nothing can."""
    import gc
    import threading
    cfg = _cfg(RVIP_PRECISION='bf16', DIM=[32, 32], FILTERS=8, LEARNING_RATE=1e-3)
    gcfg = dict(DIM=[32, 32], BATCHSIZE=4, GAUS=True, SIGMA=2, SHUFFLE=True, SEED=3)
    stop = threading.Event()
    collections = [0]

    def churn():
        class Node:
            pass
        while not stop.is_set():
            for _ in range(2000):
                a_, b_ = Node(), Node()
                a_.o, b_.o = b_, a_                     # a cycle: only the collector frees it
            collections[0] = sum(st['collections'] for st in gc.get_stats())
    th = threading.Thread(target=churn, daemon=True)
    old_thr = gc.get_threshold()
    gc.set_threshold(200, 3, 3)                         # every generation collects often
    th.start()
    try:
        before = sum(st['collections'] for st in gc.get_stats())
        for i in range(3):
            gen = rvip.Generators.SyntheticSAXGenerator(16, dict(gcfg, SEED=3 + i), in_memory=False)
            model = rvip.get_model(cfg, metrics=[])
            hist = model.fit(x=gen, epochs=2, verbose=0, max_queue_size=2, workers=2)
            assert np.isfinite(hist.history['loss']).all()
            assert model._engine(4).launch_mode == 'hipGraph'
            model._user_cycle = model                   # dies in the collector, whenever that runs, with its graphs
            del model
        after = sum(st['collections'] for st in gc.get_stats())
        assert after - before > 30, 'the collector was meant to be busy during this test'
    finally:
        stop.set()
        th.join(10)
        gc.set_threshold(*old_thr)
    gc.collect()
    torch.cuda.synchronize()


def test_fit_with_generator_and_callbacks(tmp_path):
    cfg = _cfg(RVIP_PRECISION='bf16', DIM=[64, 64], FILTERS=8, LEARNING_RATE=2e-3, MODEL_PATH=str(tmp_path),
               DROPOUT_MIN=0.0, DROPOUT_MAX=0.0)
    gcfg = dict(DIM=[64, 64], BATCHSIZE=8, GAUS=True, SIGMA=2, SHUFFLE=True)
    train = rvip.Generators.SyntheticSAXGenerator(32, gcfg, in_memory=True)
    val = rvip.Generators.SyntheticSAXGenerator(8, dict(gcfg, SHUFFLE=False), in_memory=True)
    model = rvip.get_model(cfg, metrics=[M.dice_coef_labels])
    cbs = rvip.KerasCallbacks.get_callbacks(cfg, train, val)
    hist = model.fit(x=train, validation_data=val, epochs=6, callbacks=cbs, initial_epoch=0, max_queue_size=4, verbose=0)
    h = hist.history
    assert set(h) >= {'loss', 'dice_coef_labels', 'val_loss', 'val_dice_coef_labels', 'lr'}
    assert len(h['loss']) == 6 and h['loss'][-1] < h['loss'][0]
    assert model._engine(8).launch_mode == 'hipGraph'                  # fit replays the captured step (the one bench.py measures)
    assert model.optimizer.iterations == 24 and model._params.step_count() == 24
    assert (tmp_path / 'model.h5').exists()                            # the reference's checkpoint name and format (KerasCallbacks.py:55)
    m2 = rvip.get_model(cfg)
    m2.load_weights(str(tmp_path / 'model.h5'))
    xb, _ = val[0]
    assert m2.predict(xb).shape == (8, 64, 64, 2)


class _MarkedSlices(rvip.Generators.SyntheticSAXGenerator):
    """A task the network can learn: the two landmarks of a slice are where the image carries a bright and a dark disc."""

    def _slice(self, rng, h, w):
        img, mask = super()._slice(rng, h, w)
        yy, xx = np.mgrid[0:h, 0:w]
        img = 0.5 * img[..., 0]
        for c, sign in ((0, 1.0), (1, -1.0)):
            cy, cx = np.unravel_index(int(mask[..., c].argmax()), (h, w))
            img = img + sign * 0.5 * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 3.0 ** 2))
        img = (img - img.min()) / (img.max() - img.min())
        return img[..., None].astype(np.float32), mask


@pytest.mark.parametrize('precision,loss,extra', [
    ('bf16', 'mse', dict(DROPOUT_MIN=0.1, DROPOUT_MAX=0.1)),
    ('fp16', 'bce_dice_loss', {}),                                   # default dropout schedule 0.3 .. 0.5 (Unets.py:105-106)
    ('bf16', 'BceDiceLoss', {}),                                     # the class form (sum-reduction gradient)
], ids=['bf16-mse', 'fp16-bce_dice', 'bf16-BceDiceLoss'])
def test_fit_learns_a_landmark_task_end_to_end(tmp_path, precision, loss, extra):
    """The whole product path on a learnable task: generator -> Model.fit (captured step, pinned input ring) with the reference's
    callback list -> best-only model.h5 -> a fresh model restores it -> predict -> landmarks.  The held-out landmarks must be
    found (tools/diag_learn.py: median error 0 px with BCE-Dice, 1.0-1.4 px with MSE after 30 epochs of 30 steps)."""
    lf = dict(mse=M.mse, bce_dice_loss=M.bce_dice_loss, BceDiceLoss=M.BceDiceLoss())[loss]
    cfg = _cfg(RVIP_PRECISION=precision, DIM=[64, 64], FILTERS=8, DEPTH=3, LEARNING_RATE=3e-3, MODEL_PATH=str(tmp_path), SEED=5,
               LOSS_FUNCTION=lf, **extra)
    gcfg = dict(DIM=[64, 64], BATCHSIZE=16, GAUS=True, SIGMA=2, SHUFFLE=True, SEED=7)
    train = _MarkedSlices(480, gcfg, in_memory=True)
    val = _MarkedSlices(64, dict(gcfg, SHUFFLE=False, SEED=8), in_memory=True)
    model = rvip.get_model(cfg, metrics=[])
    hist = model.fit(x=train, validation_data=val, epochs=30, callbacks=rvip.KerasCallbacks.get_callbacks(cfg, train, val), verbose=0, workers=2)
    h = hist.history
    assert np.isfinite(h['loss']).all() and h['loss'][-1] < h['loss'][0] and h['val_loss'][-1] < h['val_loss'][0], h
    assert model._engine(16).launch_mode == 'hipGraph'
    best = rvip.get_model(cfg)
    best.load_weights(str(tmp_path / 'model.h5'))
    err = []
    for i in range(len(val)):
        xb, yb = val[i]
        idx, _ = best.predict_landmarks(xb)
        ti = yb.reshape(yb.shape[0], -1, 2).argmax(1)
        err.append(np.hypot(idx // 64 - ti // 64, idx % 64 - ti % 64))
    err = np.concatenate(err).ravel()
    assert np.median(err) <= (1.5 if loss == 'mse' else 1.0) and (err <= 2.0).mean() >= 0.9, (np.median(err), (err <= 2.0).mean())


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_graph_replay_equals_eager_launches(precision, monkeypatch):
    """Engine.train_step: step 1 eager, step 2 captures, steps 3.. replay.  The replayed step must be the eager step bit for bit
    (fresh dropout stream, Adam bias correction and learning rate come from device words the kernels read), including a
    learning-rate change between replays and fit()'s pinned-ring input path."""
    cfg = _cfg(RVIP_PRECISION=precision, DIM=[32, 32], FILTERS=8, LEARNING_RATE=1e-3)
    x, y = O.synthetic_batch(4, cfg['DIM'], 2, seed=21)
    x2, y2 = O.synthetic_batch(4, cfg['DIM'], 2, seed=22)

    def run(graph):
        monkeypatch.setenv('RVIP_GRAPH', '1' if graph else '0')
        model = rvip.get_model(cfg, metrics=[])
        losses = []
        for i in range(6):
            if i == 4:
                model.optimizer.lr = 2.5e-4
            losses.append(model.train_on_batch(x if i % 2 == 0 else x2, y if i % 2 == 0 else y2)[0])
        return model, losses
    mg, lg = run(True)
    me, le = run(False)
    assert mg._engine(4).launch_mode == 'hipGraph' and me._engine(4).launch_mode == 'eager'
    assert lg == le, (lg, le)
    for a_, b_ in zip(mg.get_weights(), me.get_weights()):
        np.testing.assert_array_equal(a_, b_)
    # fit() through the pinned ring + copy stream == train_on_batch on the same batches in the same order
    monkeypatch.setenv('RVIP_GRAPH', '1')
    xs, ys = np.concatenate([x, x2, x, x2]), np.concatenate([y, y2, y, y2])
    mf = rvip.get_model(cfg, metrics=[])
    hist = mf.fit(xs, ys, batch_size=4, epochs=2, shuffle=False, verbose=0, max_queue_size=2)
    mt = rvip.get_model(cfg, metrics=[])
    lt = [mt.train_on_batch(xs[i:i + 4], ys[i:i + 4])[0] for _ in range(2) for i in range(0, 16, 4)]
    np.testing.assert_allclose(hist.history['loss'], [np.mean(lt[:4]), np.mean(lt[4:])], rtol=1e-6)
    for a_, b_ in zip(mf.get_weights(), mt.get_weights()):
        np.testing.assert_array_equal(a_, b_)


def _dp_gpu_worker(rank, world, port, overlap, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                      RVIP_OVERLAP_ALLREDUCE=overlap)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        cfg = _cfg(BATCH_NORMALISATION=False, DROPOUT_MIN=0.0, DROPOUT_MAX=0.0, DIM=[32, 32], FILTERS=8)
        model = rvip.get_model(cfg, metrics=[])
        x, y = O.synthetic_batch(8, cfg['DIM'], 2, seed=5)
        losses = [model.train_on_batch(x, y)[0] for _ in range(2)]        # global batch 8 -> 4 per rank (Model._shard)
        q.put((rank, losses, [w.copy() for w in model.get_weights()], bool(model._engine(4).overlap_ok())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('overlap', ['1', '0'])
def test_two_rank_data_parallel_step_matches_single_rank(overlap):
    """Two ranks (gloo, both on this one GPU) through the product's sharding + bucketed / single all-reduce + Adam give
    the weights of one rank training on the whole batch (no BN: per-replica statistics would differ by design)."""
    import torch.multiprocessing as mp
    cfg = _cfg(BATCH_NORMALISATION=False, DROPOUT_MIN=0.0, DROPOUT_MAX=0.0, DIM=[32, 32], FILTERS=8)
    single = rvip.get_model(cfg, metrics=[])
    x, y = O.synthetic_batch(8, cfg['DIM'], 2, seed=5)
    ref_losses = [single.train_on_batch(x, y)[0] for _ in range(2)]
    ref_w = single.get_weights()
    torch.cuda.synchronize()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000) + (1 if overlap == '1' else 0)
    procs = [ctx.Process(target=_dp_gpu_worker, args=(r, 2, port, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
    for rank, losses, w, ov in res:
        assert ov == (overlap == '1')
        for a_, b_ in zip(w, ref_w):
            np.testing.assert_allclose(a_, b_, atol=2e-6, rtol=2e-5)
    for rank, losses, _, _ in res:                       # the logged loss is the global one (loss sums are all-reduced)
        for step in range(2):
            assert abs(losses[step] - ref_losses[step]) <= 2e-5 * max(1.0, abs(ref_losses[step])), (rank, step)


def _rccl_worker(port, q):
    """One rank, backend 'nccl' (= RCCL on ROCm), data-parallel schedule forced: three captured segments, two asynchronous
    all-reduce buckets on the flat gradient block between them."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                      RVIP_FORCE_DP_SCHEDULE='1', RVIP_OVERLAP_ALLREDUCE='1')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        cfg = _cfg(RVIP_PRECISION='bf16', DIM=[64, 64], FILTERS=32, DEPTH=4)
        model = rvip.get_model(cfg, metrics=[])
        x, y = O.synthetic_batch(4, cfg['DIM'], 2, seed=5)
        x2, y2 = O.synthetic_batch(4, cfg['DIM'], 2, seed=6)
        losses = [model.train_on_batch(x if i % 2 == 0 else x2, y if i % 2 == 0 else y2)[0] for i in range(5)]
        eng = model._engine(4)
        t = torch.ones(1 << 20, device='cuda')
        dist.all_reduce(t)                                        # the communicator itself: sum over one rank is the identity
        weights = [w.copy() for w in model.get_weights()]
        marks = []                                                # one more step with bench.py's segment marks
        eng.train_step(marks=marks)
        torch.cuda.synchronize()
        spans = [(b[0], a[1].elapsed_time(b[1])) for a, b in zip(marks[:-1], marks[1:])]
        q.put((losses, weights, eng.launch_mode, bool(eng.overlap_ok()), eng.P.grad.numel() * 4,
               dist.get_backend(), float(t.sum().item()), spans))
    finally:
        dist.destroy_process_group()


def test_rccl_one_rank_data_parallel_schedule_equals_the_single_graph_step():
    """RCCL executes (VERDICT r3 missing #2): a world-size-1 'nccl' process group on this one GPU runs the product's data-parallel
    schedule -- head/decoder segment, all_reduce(async) of bucket 0, encoder segment, all_reduce(async) of bucket 1, optimiser
    segment, each segment a replayed hipGraph -- and must give the single-graph step bit for bit (a sum over one rank is the
    identity; what is exercised is the communicator, the stream ordering around the graphs and the capture of the segments
    against the real backend).  Reference behaviour: MirroredStrategy's gradient all-reduce, Unets.py:70-75."""
    import torch.multiprocessing as mp
    cfg = _cfg(RVIP_PRECISION='bf16', DIM=[64, 64], FILTERS=32, DEPTH=4)
    single = rvip.get_model(cfg, metrics=[])
    x, y = O.synthetic_batch(4, cfg['DIM'], 2, seed=5)
    x2, y2 = O.synthetic_batch(4, cfg['DIM'], 2, seed=6)
    ref_losses = [single.train_on_batch(x if i % 2 == 0 else x2, y if i % 2 == 0 else y2)[0] for i in range(5)]
    assert single._engine(4).launch_mode == 'hipGraph'
    ref_w = single.get_weights()
    torch.cuda.synchronize()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(29900 + os.getpid() % 2000, q))
    p.start()
    losses, w, mode, overlap, grad_bytes, backend, tsum, spans = q.get(timeout=300)
    p.join(60)
    assert backend == 'nccl' and tsum == float(1 << 20)
    # Engine.train_step(marks=): the step's segments as bench.py's `dp_segments` reports them at N > 1
    assert [k for k, _ in spans] == ['segment 0', 'segment 1', 'collectives landed', 'segment 2'], spans
    assert all(ms >= 0.0 for _, ms in spans) and spans[0][1] > spans[3][1]
    assert mode == 'hipGraph x3 + RCCL between' and overlap, mode
    assert grad_bytes > 30e6                                      # the 34.5 MB gradient block of the F = 32 / depth 4 graph
    assert losses == ref_losses, (losses, ref_losses)
    for a_, b_ in zip(w, ref_w):
        np.testing.assert_array_equal(a_, b_)


def _rccl_reserve_worker(port, reserve, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                      RVIP_FORCE_DP_SCHEDULE='1', RVIP_OVERLAP_ALLREDUCE='1', RVIP_RCCL_CU_RESERVE=str(reserve))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        cfg = _cfg(RVIP_PRECISION='fp32', DIM=[128, 128], FILTERS=32, DEPTH=3)
        model = rvip.get_model(cfg, metrics=[])
        x, y = O.synthetic_batch(8, cfg['DIM'], 2, seed=5)
        eng = model._engine(8)
        eng.load_input(x, y)
        eng.forward(training=True)
        eng.backward()
        torch.cuda.synchronize()
        grads = {k: v.copy() for k, v in model._params.grads_host().items()}
        losses = [model.train_on_batch(x, y)[0] for _ in range(3)]
        n_enc = 2 * model.plan.depth
        limits = [(int(th[1][0]._obj.cu_limit), i < eng.bwd_split) for i, th in enumerate(eng.bwd)
                  if getattr(th[0], '__name__', '') in ('rvip_conv3x3_wgrad', 'rvip_conv3x3_fwd', 'rvip_conv3x3_fwd_sums')]
        q.put((losses, grads, eng.cu_reserve, limits, eng.launch_mode, n_enc))
    finally:
        dist.destroy_process_group()


def test_rccl_cu_reserve_leaves_compute_units_to_the_collective():
    """RVIP_RCCL_CU_RESERVE = n sizes the grids of the contraction launches that run while gradient bucket 0 is in
    flight (the encoder's backward pass) for 256 - n compute units, so that the first multi-GPU run can A/B whether RCCL's kernels --
    which cannot co-reside with the persistent one-workgroup-per-CU kernels -- profit from CUs of their own.  One rank against RCCL:
    the launches of bucket 1 carry the reduced limits and those of bucket 0 do not, and a step's gradients and three training steps'
    losses agree up to the fp32 summation order of the partial rows / split-K slabs, whose partition follows the grid (the tensors
    themselves do not depend on it: rvip_hip.h, cu_limit).  Reference: MirroredStrategy's all-reduce, Unets.py:70-75."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    res = {}
    for reserve in (0, 8):
        q = ctx.Queue()
        p = ctx.Process(target=_rccl_reserve_worker, args=(30100 + os.getpid() % 1500 + reserve, reserve, q))
        p.start()
        res[reserve] = q.get(timeout=300)
        p.join(60)
    l0, w0, r0, lim0, mode0, _ = res[0]
    l8, w8, r8, lim8, mode8, _ = res[8]
    assert r0 == 0 and r8 == 8 and mode8 == 'hipGraph x3 + RCCL between', (r0, r8, mode8)
    # (fp32: no layer runs as a weight / data gradient pair -- that is a 16-bit kernel, whose halves would read 128 / 124 here --
    #  so every contraction launch of the backward pass has the chip to itself, minus the reserve while bucket 0 travels)
    assert all(lim == 0 for lim, _ in lim0), lim0
    assert all(lim == 248 for lim, head in lim8 if not head) and all(lim == 0 for lim, head in lim8 if head), lim8
    assert any(not head for _, head in lim8)
    np.testing.assert_allclose(l8, l0, rtol=2e-5)
    for k, g0 in w0.items():        # the gradients of one step (not weights after Adam: where a gradient element is ~0 its last bits decide a step of the learning rate)
        np.testing.assert_allclose(w8[k], g0, atol=2e-5 * float(np.abs(g0).max()) + 1e-12, rtol=0, err_msg=str(k))


def _dp_fit_worker(rank, world, port, path, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        np.random.seed(1000 + rank)                                     # the process-global NumPy stream DIFFERS per rank on purpose
        cfg = _cfg(DIM=[32, 32], FILTERS=8, LEARNING_RATE=1e-3, MODEL_PATH=os.path.join(path, 'rank%d' % rank if rank else 'chief'))
        gcfg = dict(DIM=[32, 32], BATCHSIZE=8, GAUS=True, SIGMA=2, SHUFFLE=True, SEED=7)
        train = rvip.Generators.SyntheticSAXGenerator(32, gcfg, in_memory=True)
        val = rvip.Generators.SyntheticSAXGenerator(8, dict(gcfg, SHUFFLE=False), in_memory=True)
        seen, mine = [], []
        orig = type(train).batch_slice

        class Spy(type(train)):
            def __getitem__(self, i):                                   # fit() must not ask for whole batches when it can ask for slices
                raise AssertionError('a rank generated a whole global batch')

            def batch_slice(self, i, lo, hi):
                seen.append(tuple(int(v) for v in self.INDICES[i * self.BATCHSIZE:(i + 1) * self.BATCHSIZE]))
                mine.append(tuple(int(v) for v in self.INDICES[i * self.BATCHSIZE + lo:i * self.BATCHSIZE + hi]))
                return orig(self, i, lo, hi)
        train.__class__ = Spy
        model = rvip.get_model(cfg, metrics=[M.dice_coef_labels])
        cbs = rvip.KerasCallbacks.get_callbacks(cfg, train, val)
        hist = model.fit(x=train, validation_data=val, epochs=2, callbacks=cbs, verbose=0, max_queue_size=2)
        w = model.get_weights()                                          # collective: replica mean of the BN moving statistics
        local_mv = model._params.moving.detach().cpu().numpy().copy()
        q.put((rank, seen, hist.history, [a.copy() for a in w], local_mv, os.path.exists(os.path.join(cfg['MODEL_PATH'], 'model.h5')),
               mine, train.samples_generated))
    finally:
        dist.destroy_process_group()


def test_two_rank_fit_is_rank_consistent(tmp_path):
    """fit() for 2 epochs on 2 ranks (gloo, both on this GPU) with BN and dropout on: both ranks draw the same global batches in the
    same order although their process-global NumPy streams differ, train on different halves -- each rank GENERATES only its half
    (B / world samples per step: the reference's one MirroredStrategy process produces every sample once, Generators.py:175-228) --
    and end with identical weights, identical (replica-mean) BN moving statistics and identical epoch logs; only rank 0 writes the
    checkpoint."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_fit_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
    (_, seen0, h0, w0, mv0, ck0, mine0, n0), (_, seen1, h1, w1, mv1, ck1, mine1, n1) = res
    assert seen0 == seen1 and len(seen0) == 8 and sorted(sum(seen0[:4], ())) == list(range(32))      # same global batches, each sample once per epoch
    assert seen0[:4] != seen0[4:]                                                                     # ... reshuffled between the epochs
    assert n0 == n1 == 8 * 4                                                                          # 8 steps x (batch 8 / 2 ranks) samples generated per rank
    assert all(len(a) == 4 and a + b == g for a, b, g in zip(mine0, mine1, seen0))                    # the two halves of every global batch
    assert set(h0) == set(h1) and all(h0[k] == h1[k] for k in h0), (h0, h1)
    for a_, b_ in zip(w0, w1):
        np.testing.assert_array_equal(a_, b_)
    np.testing.assert_array_equal(mv0, mv1)
    assert ck0 and not ck1


@pytest.mark.parametrize('variant', [dict(DIM=[4, 64, 64], M_POOL=[1, 2, 2], F_SIZE=[3, 3, 3], FILTERS=32),
                                     dict(USE_UPSAMPLE=False, FILTERS=16, DIM=[64, 64])],
                         ids=['cine-3d', 'conv2d-transpose'])
def test_bf16_variants_are_deterministic_and_learn(variant):
    """bf16 path of the 3-D cine graph (cfg 5's layer types) and of the Conv2DTranspose decoder: two identical steps
    from identical state are bit-identical, and a few Adam steps on one batch lower the loss."""
    cfg = _cfg(RVIP_PRECISION='bf16', LEARNING_RATE=2e-3, **variant)
    model = rvip.get_model(cfg, metrics=[])
    x, y = O.synthetic_batch(4, cfg['DIM'], 2, seed=6)
    eng = model._engine(4)
    outs = []
    for _ in range(2):
        eng.load_input(x, y)
        eng.forward(training=True)
        eng.backward()
        torch.cuda.synchronize()
        outs.append((eng.loss.clone(), eng.pred.clone(), model._params.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    assert torch.isfinite(outs[0][2]).all()
    losses = [model.train_on_batch(x, y)[0] for _ in range(12)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    p = model.predict(x)
    assert p.shape == (4,) + tuple(cfg['DIM']) + (2,) and np.isfinite(p).all()


def test_bf16_loss_curve_tracks_the_float64_oracle():
    """Fifteen Adam steps on one batch: the bf16 device path (dropout on, BN batch statistics, Keras-Adam) and the float64
    oracle started from the same weights and fed the same dropout masks follow the same loss curve (bf16 storage noise only)."""
    cfg = _cfg(RVIP_PRECISION='bf16', FILTERS=16, DIM=[64, 64], LEARNING_RATE=1e-3)
    model = rvip.get_model(cfg, metrics=[])
    ref, layers = _oracle_from(model, cfg)
    B = 4
    x, y = O.synthetic_batch(B, cfg['DIM'], 2, seed=12)
    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    dev_losses, ref_losses = [], []
    for step in range(15):
        dev_losses.append(model.train_on_batch(x, y)[0])
        lv, _ = ref.train_step(x64, y64, 'mse', _masks(layers, B, model.seed, step))
        ref_losses.append(lv)
    dev_losses, ref_losses = np.array(dev_losses), np.array(ref_losses)
    assert ref_losses[-1] < 0.8 * ref_losses[0]                        # the oracle is actually learning on this batch
    np.testing.assert_allclose(dev_losses, ref_losses, rtol=3e-2)
    # Inference after training.  The two weight sets drift apart in the directions the training-mode loss cannot see (a conv
    # bias ahead of relu -> BN has a near-zero gradient, and Adam turns the rounding noise of a near-zero gradient into a
    # full-size +-lr update), and inference on the barely-moved moving statistics does see those directions
    # (tools/diag_losscurve.py: fp32 shows the same drift with the loss equal to 6e-8).  So compare like with like: the
    # oracle's trained weights, moving statistics included, loaded into the device model.
    refw = [w for l in layers if l['name'] in ref.params for w in ref.params[l['name']]]
    model.set_weights([np.asarray(w, np.float32) for w in refw])
    xt, _ = O.synthetic_batch(2, cfg['DIM'], 2, seed=13)
    diff = np.abs(model.predict(xt) - ref.predict(xt.astype(np.float64)))
    assert diff.max() < 3e-2 and diff.mean() < 3e-3, (diff.mean(), diff.max())


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_pipelined_predict_and_evaluate_equal_the_batch_calls(precision):
    """Model.predict / Model.evaluate run through pinned rings with the copies of batch k +- 1 under batch k (round 4): the result
    must be what predict_on_batch / test_on_batch give batch by batch -- ragged last batch, ndarray and Sequence inputs, several
    calls in a row (the rings are reused), and a training step in between (the staging buffers are shared with fit)."""
    cfg = _cfg(RVIP_PRECISION=precision, DIM=[32, 32], FILTERS=8, DEPTH=2)
    model = rvip.get_model(cfg, metrics=[M.dice_coef_labels])
    x, y = O.synthetic_batch(70, cfg['DIM'], 2, seed=3)
    model.train_on_batch(x[:16], y[:16])
    ref = np.concatenate([model.predict_on_batch(x[i:i + 16]) for i in range(0, 70, 16)], 0)
    for _ in range(2):
        got = model.predict(x, batch_size=16)
        assert got.shape == ref.shape and got.dtype == np.float32
        np.testing.assert_array_equal(got, ref)

    class Seq:
        def __len__(self):
            return 5

        def __getitem__(self, i):
            return x[i * 16:(i + 1) * 16], y[i * 16:(i + 1) * 16]
    np.testing.assert_array_equal(model.predict(Seq()), ref)
    want = np.mean([model.test_on_batch(*Seq()[i]) for i in range(5)], 0)
    for _ in range(2):
        np.testing.assert_allclose(model.evaluate(Seq()), want, rtol=2e-6, atol=1e-9)
    model.train_on_batch(x[:16], y[:16])                   # the weights move: the pipelined calls must see the new ones
    ref2 = np.concatenate([model.predict_on_batch(x[i:i + 16]) for i in range(0, 70, 16)], 0)
    assert np.abs(ref2 - ref).max() > 0
    np.testing.assert_array_equal(model.predict(x, batch_size=16), ref2)
    np.testing.assert_allclose(model.evaluate(x[:64], y[:64]), model.test_on_batch(x[:64], y[:64]), rtol=2e-6, atol=1e-9)


_REV_CHILD = r'''
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
import cmr_landmark_detection_amd as rvip
M = rvip.Loss_and_metrics
cfg = dict(DIM=[48, 40], FILTERS=16, DEPTH=2, BATCH_NORMALISATION=True, ACTIVATION='relu', MASK_CLASSES=2, LEARNING_RATE=1e-3,
           DROPOUT_MIN=0.3, DROPOUT_MAX=0.5, RVIP_PRECISION=%r, LOSS_FUNCTION=M.mse, SEED=5)
rng = np.random.default_rng(3)
x = rng.random((6, 48, 40, 1)).astype(np.float32)
y = rng.random((6, 48, 40, 2)).astype(np.float32)
m = rvip.get_model(cfg, metrics=[])
losses = [float(np.ravel(m.train_on_batch(x, y))[0]) for _ in range(3)]
h = hashlib.sha256()
for w in m.get_weights():
    h.update(np.ascontiguousarray(w).tobytes())
h.update(np.asarray(m.predict(x, batch_size=6)).tobytes())
print('RESULT', h.hexdigest(), ' '.join('%%.9g' %% l for l in losses))
'''


@pytest.mark.parametrize('precision', ['bf16', 'fp32'])
def test_direction_of_the_apply_passes_does_not_change_a_bit(precision):
    """RVIP_REV (csrc/rvip_pointwise.hip: the element-wise passes walk their tensor from its end, for the Infinity Cache) only reorders
    work: three training steps (dropout, pooling with argmax, keep bits, deferred bias rows) and a prediction are the same bits with the
    forward passes reversed (default), nothing reversed, and the backward pass reversed as well.  The switch is read once per process,
    hence the children (one at a time)."""
    import subprocess
    import sys
    seen = {}
    for rev in ('0', '1', '3'):
        env = dict(os.environ, RVIP_REV=rev)
        out = subprocess.run([sys.executable, '-c', _REV_CHILD % (ROOT, precision)], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [l for l in out.stdout.splitlines() if l.startswith('RESULT')]
        assert line, out.stdout[-2000:]
        seen[rev] = line[-1]
    assert seen['0'] == seen['1'] == seen['3'], seen
