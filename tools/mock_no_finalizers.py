"""Upper bound of what merging the single-workgroup finalisers into their neighbours could save: the captured
step timed as it is, and with (a) the 18 rvip_bn_stats_finalize launches, (b) also the 17 rvip_bn_bwd_coef launches REMOVED from the
launch lists after two eager steps (their outputs -- scale / shift, BN-backward coefficients -- stay those of step 2: the arithmetic
is stale, the timing is that of a step whose finalisers cost nothing at all).  Not a product path.
    python tools/mock_no_finalizers.py [steps]"""
import os, sys, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cmr_landmark_detection_amd as rvip
M = rvip.Loss_and_metrics
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = dict(DIM=[256, 256], FILTERS=32, DEPTH=4, BATCH_NORMALISATION=True, BN_FIRST=False, ACTIVATION='relu', MASK_CLASSES=2, M_POOL=[2, 2], F_SIZE=[3, 3],
           LEARNING_RATE=1e-4, RVIP_PRECISION='bf16', LOSS_FUNCTION=M.mse, SEED=42)
B = 32
gen = rvip.Generators.SyntheticSAXGenerator(B, dict(DIM=cfg['DIM'], BATCHSIZE=B, GAUS=True, SIGMA=2, SHUFFLE=False, SEED=42))
x, y = gen[0]


def run(drop):
    model = rvip.get_model(cfg, metrics=[])
    eng = model._engine(B)
    eng.load_input(x, y)
    os.environ['RVIP_GRAPH'] = '0'
    eng.train_step(); eng.train_step()
    torch.cuda.synchronize()
    n0 = len(eng.fwd_train) + len(eng.bwd)
    for name in drop:
        eng.fwd_train[:] = [t for t in eng.fwd_train if getattr(t[0], '__name__', '') != name]
        eng.bwd[:] = [t for t in eng.bwd if getattr(t[0], '__name__', '') != name]
    n1 = len(eng.fwd_train) + len(eng.bwd)
    os.environ['RVIP_GRAPH'] = '1'
    for _ in range(25):
        eng.train_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.train_step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    model.close()
    return ms, n0 - n1, eng.launch_mode


for rep in range(2):
    for tag, drop in (('as shipped', ()), ('no rvip_bn_stats_finalize', ('rvip_bn_stats_finalize',)),
                      ('no finalize, no rvip_bn_bwd_coef', ('rvip_bn_stats_finalize', 'rvip_bn_bwd_coef'))):
        ms, removed, mode = run(drop)
        print('%-36s %d launches removed: %.4f ms per step (%s)' % (tag, removed, ms, mode), flush=True)
