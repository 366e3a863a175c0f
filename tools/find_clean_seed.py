"""Pick the data seeds of tests/test_gpu_model.py::test_fp32_training_steps_match_oracle on the CPU with the float64 oracle alone:
for every variant of that test, the first seed whose FIRST training step has no knife edge (no ReLU pre-activation / pooling
decision within fp32 noise of its kink: `oracle.knife_edges` at 1.5x the test's margin, KNIFE_REL), so that the tight gradient
bound is the one that applies there.  The initial weights are the model's own seeded initialisation (no GPU needed).
    python tools/find_clean_seed.py [variant index ...]      -> prints CLEAN_SEEDS entries"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_model as T   # noqa: E402  (importable without a GPU: only its tests need one)
from oracle import rvip_oracle as O   # noqa: E402

REL = float(os.environ.get('KNIFE_REL', '4.5e-6'))
LIMIT = int(os.environ.get('SEED_LIMIT', '600'))
which = [int(a) for a in sys.argv[1:]] or range(len(T.VARIANTS))
for vi in which:
    variant = T.VARIANTS[vi]
    cfg = T._cfg(**variant)
    model = T.rvip.get_model(cfg, metrics=[])
    ref, layers = T._oracle_from(model, cfg)
    kind = T.M.resolve_loss(cfg['LOSS_FUNCTION'])
    red = T.M.loss_reduction(cfg['LOSS_FUNCTION'])
    B = 4
    masks = T._masks(layers, B, model.seed, 0)
    found = None
    for seed in range(LIMIT):
        x, y = T._batch(B, cfg, seed)                    # the test's own batches (image channels / mask classes of the variant)
        if kind[0] == 'mse':
            _, _, _, cache = ref.loss_and_grads(x.astype(np.float64), y.astype(np.float64), 'mse', masks)
        else:
            _, _, _, cache = ref.loss_and_grads(x.astype(np.float64), y.astype(np.float64), 'bce_dice', masks, w_bce=kind[1], w_dice=kind[2], reduction=red)
        if not O.knife_edges(layers, cache, rel=REL):
            found = seed
            break
    print('    %r: %s,' % (T._variant_id(variant), found), flush=True)
