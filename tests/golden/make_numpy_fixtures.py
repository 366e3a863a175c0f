"""Generates tests/golden/ref_numpy_fixtures.npz by EXECUTING the reference's own pure-NumPy / SciPy functions on seeded inputs.

The modules that hold them cannot be imported in the build container (their top-level imports need SimpleITK, cv2,
albumentations, scikit-learn, TensorFlow: ordinary ModuleNotFoundError), but the functions themselves only use NumPy, SciPy,
`sys`, `logging` and `math`.  This script reads the reference source as text, takes the named function definitions (and, for
code that lives inside a method, the named statements) out of the `ast` of the file, compiles exactly those nodes and runs them.
Nothing of the reference's text is stored: the fixture holds inputs and outputs only (data), and this script is committed next to
it so the vectors can be regenerated.  Runs in the build container only (/root/reference does not travel to the GPU box).

One alias is supplied: `np.bool` (removed in NumPy 1.24; the reference pins numpy 1.18 where it IS the builtin `bool`,
environment.yml:82).

Reference sites executed:
  src/data/Preprocess.py:425-437  transform_to_binary_mask      :440-455 from_channel_to_flat     :458-468 clip_quantile
  src/data/Preprocess.py:471-491  normalise_image (minmax / standard)      :494-541 pad_and_crop
  src/data/Generators.py:385-391  the GAUS branch of DataGenerator.__preprocess_one_image__ (Gaussian heat-map targets)
  src/models/predict_model.py:153-156  heat-maps -> flat labels (> 0.5 -> 1 / 2)
  src/models/evaluate_cv.py:418-442    get_mean_rvip_2d (the definition in force: it shadows :48)   :508-536 get_angle2x   :538-546 get_dist
"""
import ast
import logging
import math
import os
import sys
import types

import numpy as np

REF = os.environ.get('RVIP_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
if not hasattr(np, 'bool'):
    np.bool = bool                                                       # numpy 1.18 alias (see module docstring)


def _functions(relpath, names, extra=None):
    """Compile the LAST top-level definition of each name in `names` from the reference file (later definitions shadow earlier ones)."""
    path = os.path.join(REF, relpath)
    tree = ast.parse(open(path).read(), filename=path)
    picked = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            picked[node.name] = node
    missing = set(names) - set(picked)
    assert not missing, (relpath, missing)
    ns = dict(np=np, sys=sys, logging=logging, math=math, atan2=math.atan2, degrees=math.degrees)
    ns.update(extra or {})
    mod = ast.Module(body=[picked[n] for n in names], type_ignores=[])
    exec(compile(mod, path, 'exec'), ns)
    return [ns[n] for n in names]


def _method_statements(relpath, cls, method, pick):
    """Statements of `cls.method` selected by `pick(stmt) -> bool`, compiled as a code object to exec in a caller namespace."""
    path = os.path.join(REF, relpath)
    tree = ast.parse(open(path).read(), filename=path)
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for f in node.body:
                if isinstance(f, ast.FunctionDef) and f.name == method:
                    stmts = [s for s in ast.walk(f) if isinstance(s, ast.stmt) and pick(s)]
                    assert stmts, (relpath, cls, method)
                    return compile(ast.Module(body=stmts[:1], type_ignores=[]), path, 'exec')
    raise AssertionError((relpath, cls, method))


def _function_statements(relpath, func, pick):
    path = os.path.join(REF, relpath)
    tree = ast.parse(open(path).read(), filename=path)
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == func:
            stmts = [s for s in ast.walk(node) if isinstance(s, ast.stmt) and pick(s)]
            assert stmts, (relpath, func)
            return compile(ast.Module(body=stmts, type_ignores=[]), path, 'exec')
    raise AssertionError((relpath, func))


def main(out_path):
    rng = np.random.default_rng(20261004)
    out = {}
    tbm, fctf, clipq, norm, pac = _functions('src/data/Preprocess.py', ['transform_to_binary_mask', 'from_channel_to_flat', 'clip_quantile',
                                                                        'normalise_image', 'pad_and_crop'])
    # ---- transform_to_binary_mask / from_channel_to_flat
    lab2 = rng.integers(0, 4, (17, 13)).astype(np.uint8)
    lab3 = rng.integers(0, 4, (3, 9, 11)).astype(np.int32)
    out['tbm_in_2d'], out['tbm_in_3d'] = lab2, lab3
    out['tbm_out_2d_0123'] = tbm(lab2, [0, 1, 2, 3])
    out['tbm_out_2d_12'] = tbm(lab2, [1, 2])
    out['tbm_out_3d_12'] = tbm(lab3, [1, 2])
    soft = rng.random((6, 12, 10, 3)).astype(np.float32)
    soft[0, 0, 0] = [0.5, 0.5, 0.49999]                                  # the >= 0.5 boundary, later channels win
    out['fctf_in'] = soft
    out['fctf_out_c0'], out['fctf_out_c1'] = fctf(soft, 0), fctf(soft, 1)
    # ---- clip_quantile / normalise_image
    img = (rng.gamma(2.0, 50.0, (40, 37)) - 5.0).astype(np.float32)
    img[3, 4], img[20, 20] = 9000.0, -40.0
    out['clip_in'] = img
    out['clip_out_999'], out['clip_out_95_lb10'] = clipq(img, .999), clipq(img, .95, 10)
    out['norm_in'] = img
    out['norm_out_minmax'], out['norm_out_standard'] = norm(img, 'MinMax'), norm(img, 'Standard')
    const = np.full((5, 5), 3.25, np.float64)
    out['norm_const_in'] = const
    out['norm_const_minmax'], out['norm_const_standard'] = norm(const, 'minmax'), norm(const, 'standard')
    # ---- pad_and_crop: even / odd pad and crop on every axis, 2-D and 3-D, mixed
    cases = [((7, 10), (12, 12)), ((7, 10), (4, 5)), ((7, 10), (10, 7)), ((9, 9), (9, 9)), ((5, 6), (8, 3)), ((3, 8, 9), (4, 5, 12)),
             ((16, 31, 30), (16, 32, 27)), ((1, 1), (4, 4))]
    out['pac_n'] = np.int64(len(cases))
    for i, (src, dst) in enumerate(cases):
        a = rng.standard_normal(src)
        out['pac_%d_in' % i], out['pac_%d_target' % i], out['pac_%d_out' % i] = a, np.array(dst), pac(a, dst)
    # ---- GAUS branch (Generators.py:385-391): executed with a stand-in `self` that only carries GAUS / SIGMA
    code = _method_statements('src/data/Generators.py', 'DataGenerator', '__preprocess_one_image__',
                              lambda s: isinstance(s, ast.If) and isinstance(s.test, ast.Attribute) and s.test.attr == 'GAUS'
                              and any(isinstance(n, ast.Attribute) and n.attr == 'gaussian_filter' for n in ast.walk(s)))
    pts = np.zeros((24, 20), np.uint8)
    pts[5, 6], pts[17, 12], pts[0, 19] = 1, 2, 2                          # one point per label + a border point
    onehot = tbm(pts, [1, 2])
    out['gaus_in_onehot'] = onehot
    for sigma in (1, 2, 4):
        ns = dict(self=types.SimpleNamespace(GAUS=True, SIGMA=sigma), mask_nda=onehot.copy(), np=np, normalise_image=norm)
        exec(code, ns)
        out['gaus_out_sigma%d' % sigma] = ns['mask_nda']
    empty = tbm(np.zeros((8, 8), np.uint8), [1, 2])                       # no landmark at all: 0 / eps
    ns = dict(self=types.SimpleNamespace(GAUS=True, SIGMA=2), mask_nda=empty.copy(), np=np, normalise_image=norm)
    exec(code, ns)
    out['gaus_in_empty'], out['gaus_out_empty'] = empty, ns['mask_nda']
    # ---- heat-maps -> flat labels (predict_model.py:153-156) and mean RVIP points / angle / distance (evaluate_cv.py)
    code = _function_statements('src/models/predict_model.py', 'pred_fold',
                                lambda s: isinstance(s, ast.Assign) and any(isinstance(n, ast.Name) and n.id == 'preds_flat' for n in ast.walk(s.targets[0]))
                                and not any(isinstance(n, ast.Name) and n.id == 'clean_3d_prediction_2d_cc' for n in ast.walk(s.value)))   # the CC_FILTER line needs cv2
    preds = rng.random((5, 16, 14, 2)).astype(np.float32) ** 3
    preds[0, 0, 0] = [0.5, 0.5]                                           # not > 0.5
    preds[0, 0, 1] = [0.9, 0.8]                                           # both: label 2 wins
    ns = dict(np=np, preds=preds, gts=np.zeros_like(preds))
    exec(code, ns)
    out['flat_in_preds'], out['flat_out'] = preds, ns['preds_flat']
    mean2d, angle2x, dist = _functions('src/models/evaluate_cv.py', ['get_mean_rvip_2d', 'get_angle2x', 'get_dist'])
    slices = [ns['preds_flat'][i] for i in range(5)]
    slices.append(np.zeros((16, 14)))                                     # nothing predicted
    only2 = np.zeros((16, 14)); only2[3:5, 7] = 2; slices.append(only2)   # background + label 2 only
    nobg = np.ones((16, 14)); nobg[8:, :] = 2; slices.append(nobg)        # NO background: np.unique(x)[1:] drops label 1
    full1 = np.ones((16, 14)); slices.append(full1)                       # one value only -> no labels
    out['rvip_in'] = np.stack(slices)
    pts_out, pts_both = [], []
    for s in slices:
        for store, both in ((pts_out, False), (pts_both, True)):
            a, b = mean2d(s, both_only=both)
            store.append([a if a is not None else [np.nan, np.nan], b if b is not None else [np.nan, np.nan]])
    out['rvip_out'], out['rvip_out_both_only'] = np.array(pts_out, np.float64), np.array(pts_both, np.float64)
    pairs = np.array([[[3.0, 4.0], [10.0, 9.0]], [[0.0, 0.0], [0.0, 5.0]], [[7.5, 2.25], [1.5, 8.0]], [[2.0, 2.0], [2.0, 2.0]]])
    out['angle_in'] = pairs
    out['angle_out'] = np.array([angle2x(p[0], p[1]) for p in pairs], np.float64)
    out['dist_out'] = np.array([dist(p[0], p[1]) for p in pairs], np.float64)
    np.savez_compressed(out_path, **out)
    print(out_path, os.path.getsize(out_path), 'bytes;', len(out), 'arrays')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, 'ref_numpy_fixtures.npz'))
