"""Times single launches of the benchmark configuration's step, back to back, with HIP events (eager, on the launch stream).

    python tools/time_launch.py bn_apply_head_mse [more name fragments ...] [--reps 50] [--batch 32] [--dim 256] [--loss mse]

Every entry of the engine's launch lists whose C-ABI function name contains one of the fragments is timed `reps` times in a row
(so cache state is "warm from itself": use for A/B of one kernel on one box, not as the in-step duration)."""
import argparse
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('frags', nargs='+')
    ap.add_argument('--reps', type=int, default=50)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--dim', type=int, default=256)
    ap.add_argument('--filters', type=int, default=32)
    ap.add_argument('--depth', type=int, default=4)
    ap.add_argument('--precision', default='bf16')
    ap.add_argument('--loss', default='mse')
    a = ap.parse_args()
    import torch
    rvip = importlib.import_module('cmr-landmark-detection_amd')
    M = rvip.Loss_and_metrics
    cfg = dict(DIM=[a.dim, a.dim], FILTERS=a.filters, DEPTH=a.depth, BATCH_NORMALISATION=True, ACTIVATION='relu', MASK_CLASSES=2,
               LEARNING_RATE=1e-4, RVIP_PRECISION=a.precision, LOSS_FUNCTION=M.mse if a.loss == 'mse' else M.bce_dice_loss, SEED=42)
    model = rvip.get_model(cfg, metrics=[])
    gen = rvip.Generators.SyntheticSAXGenerator(a.batch, dict(DIM=cfg['DIM'], BATCHSIZE=a.batch, GAUS=True, SIGMA=2, SHUFFLE=False, SEED=42))
    x, y = gen[0]
    eng = model._engine(a.batch)
    eng.load_input(x, y)
    os.environ['RVIP_GRAPH'] = '0'
    eng.train_step()
    torch.cuda.synchronize()
    s = torch.cuda.current_stream()
    cs = C.c_void_p(s.cuda_stream)
    for seq in (eng.fwd_train, eng.bwd, eng.opt):
        for th in seq:
            fn, args = th[0], th[1]
            lab = th[2] if len(th) > 2 else ''
            if not any(f in fn.__name__ for f in a.frags):
                continue
            for _ in range(3):
                assert fn(*args, cs) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(a.reps):
                fn(*args, cs)
            e1.record(s)
            torch.cuda.synchronize()
            print('%-34s %-52s %8.2f us' % (fn.__name__, lab[:52], 1e3 * e0.elapsed_time(e1) / a.reps))


if __name__ == '__main__':
    main()
