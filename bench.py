"""Headline benchmark: SAX slices/sec of the full training step (fwd + loss + bwd + [RCCL all-reduce] + Adam) of
the 256x256 4-level U-Net with 2 heat-maps, bf16 activations / fp32 accumulate, batch 32 per GPU
(BASELINE.json configs[1]; configs[2] = the same per GPU on N GPUs, weak scaling).

    python bench.py --gpus 1 --steps 200 --warmup 20          (the defaults)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
           bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  ``value`` = slices all ranks processed / max-over-ranks wall time of exactly K steps
with the synthetic batch already resident in HBM.  ``roofline`` is for the dominant kernel family -- every MFMA contraction launch
of the step: the implicit-GEMM conv (forward with the fused BN statistics, data gradient), the weight gradient (with its slab fold)
and the pair launches that run a layer's weight and data gradient as the two parts of one grid: achieved = the algorithmic conv
FLOPs of those launches / their duration INSIDE the step, measured live with HIP events around every launch of eager steps after the
timed region (each launch behind its producer, as in the captured step), minus the event pair's own cost calibrated on a trivial
kernel in the same run; ``roofline.parts`` splits it by entry point, ``roofline.profile`` holds the same figure from the committed
rocprofv3 kernel trace when it was taken on these kernel sources, ``kernels`` the per-entry-point eager table.  ``other_configs``:
BASELINE.json configs[3] (fp16, 512^2, F = 64, depth 5), configs[4] per GPU (Conv3D cine volumes) and the Train notebook's BCE-Dice
loss through the same product path, a few dozen steps each after everything the headline reports.  ``cpu_baseline`` (rank 0, N=1
only) times the same training step on the host cores with the PyTorch-CPU port in oracle/ (the reference's own TF2-CPU path cannot
run here: no TensorFlow) on a bounded sample.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0       # dense MFMA bf16, MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0
PMC_SUMMARY = 'r05_pmc_summary.json'
KERNEL_STATS = 'r05_bench_kernel_stats.csv'          # rocprofv3 --kernel-trace --stats of this command (tools/profile_round.sh), + .meta.json


def csrc_sha16():
    """Identity of the kernel sources a PMC summary was taken on (tools/pmc_summary.py stores the same digest)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, 'cmr-landmark-detection_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(ROOT, 'cmr-landmark-detection_amd', 'csrc', '*.h'))
                    + [os.path.join(ROOT, 'include', 'rvip_hip.h')]):
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def workload_key(args):
    return 'dim%d_f%d_d%d_b%d_t%d_%s' % (args.dim, args.filters, args.depth, args.batch, args.frames, args.precision)


def make_cfg(M, dim, filters, depth, frames, precision, loss):
    cfg = dict(DIM=[dim, dim], FILTERS=filters, DEPTH=depth, BATCH_NORMALISATION=True, BN_FIRST=False, ACTIVATION='relu',
               MASK_CLASSES=2, M_POOL=[2, 2], F_SIZE=[3, 3], LEARNING_RATE=1e-4, RVIP_PRECISION=precision,
               LOSS_FUNCTION=M.mse if loss == 'mse' else M.bce_dice_loss, SEED=42)
    if frames > 0:
        cfg.update(DIM=[frames, dim, dim], M_POOL=[1, 2, 2], F_SIZE=[3, 3, 3])
    return cfg


OTHER_CONFIGS = (      # (key, BASELINE.json config, dim, filters, depth, frames, precision, loss, batch per GPU)
    ('cfg4', 'configs[3]: 5-level U-Net, 64 base filters, 512x512, fp16 MFMA path', 512, 64, 5, 0, 'fp16', 'mse', 8),
    ('cfg5', 'configs[4] per GPU: 3D cine U-Net, 16x256x256 volumes, Conv3D implicit GEMM, batch 4', 256, 32, 4, 16, 'bf16', 'mse', 4),
    ('cfg2_bce_dice', "configs[1] with the Train notebook's bce_dice_loss (Train_tests.ipynb:219)", 256, 32, 4, 0, 'bf16', 'bce_dice', 32),
)


def measure_other(rvip, spec, steps, warmup=5):
    """One more configuration through the same product path (get_model -> Engine.train_step: eager, capture, replay), timed like the
    headline: inputs resident, `steps` replays between two device synchronisations.  Its engine and parameters are freed afterwards."""
    import gc
    import numpy as np
    import torch
    key, what, dim, filters, depth, frames, precision, loss, B = spec
    M = rvip.Loss_and_metrics
    cfg = make_cfg(M, dim, filters, depth, frames, precision, loss)
    model = rvip.get_model(cfg, metrics=[])
    try:
        gen = rvip.Generators.SyntheticSAXGenerator(B, dict(DIM=cfg['DIM'], BATCHSIZE=B, GAUS=True, SIGMA=2, SHUFFLE=False, SEED=42))
        x, y = gen[0]
        eng = model._engine(B)
        eng.load_input(x, y)
        for _ in range(3 + warmup):
            eng.train_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.train_step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        lossv = float(eng.loss.item())
        assert np.isfinite(lossv), 'training diverged in other_configs[%s]' % key
        step_flops = model.plan.flops_per_slice()[1]
        value = B * steps / el
        return {'config': what, 'value': round(value, 2), 'unit': 'volumes/s' if frames > 0 else 'slices/s', 'ms_per_step': round(1e3 * el / steps, 4),
                'steps': steps, 'warmup': warmup, 'dtype': {'bf16': 'bf16', 'fp16': 'f16', 'fp32': 'f32'}[precision],
                'workload': '%d-level %s U-Net F=%d, %s, batch %d, fwd+loss(%s)+bwd+Adam' % (
                    depth, '3D cine (Conv3D 3x3x3, pool 1x2x2)' if frames > 0 else '2D', filters,
                    ('%dx%dx%d' % (frames, dim, dim)) if frames > 0 else '%dx%d' % (dim, dim), B, 'MSE' if loss == 'mse' else 'BCE-Dice'),
                'launch': eng.launch_mode, 'gflop_per_unit_fwd_bwd': round(step_flops / 1e9, 3),
                'mfma_util_whole_step': round(value * step_flops / (PEAK_BF16_TFLOPS * 1e12), 4), 'loss': lossv}
    finally:
        model.close()
        del model
        gc.collect()
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200, help='timed steps (default: ~1.1 s of device time at cfg 2)')
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=32, help='slices per GPU (BASELINE.json: 32)')
    ap.add_argument('--dim', type=int, default=256)
    ap.add_argument('--filters', type=int, default=32, help='base filters (BASELINE.json headline: 32; cfg 4: 64)')
    ap.add_argument('--depth', type=int, default=4, help='U-Net levels (headline: 4; cfg 4: 5)')
    ap.add_argument('--frames', type=int, default=0, help='> 0: 3-D cine graph on [frames, dim, dim] volumes (cfg 5: 16), Conv3D 3x3x3, pool (1,2,2)')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp16', 'fp32'], help='fp16: BASELINE.json configs[3] (static loss scaling)')
    ap.add_argument('--loss', default='mse', choices=['mse', 'bce_dice'], help="bce_dice: the Train notebook's loss (Train_tests.ipynb:219, Loss_and_metrics.py:229-245); the headline is MSE (train_model.py:184)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying a captured hipGraph')
    ap.add_argument('--no-fit', action='store_true', help='skip the Model.fit throughput measurement')
    ap.add_argument('--no-aux', action='store_true', help='skip the informational predict pass (PMC runs: only training-step kernels)')
    ap.add_argument('--cpu-batch', type=int, default=8)
    ap.add_argument('--detail', default=None, help='write a per-launch timing table (conv / wgrad shapes) to this file')
    ap.add_argument('--dump-labels', default=None, help='write the launch labels of one step, in launch order, as JSON (tools/step_timeline.py joins them with a kernel trace)')
    ap.add_argument('--no-roofline-pass', action='store_true', help='skip the eager per-launch and family-graph passes (profiling runs: only the captured step in the trace)')
    ap.add_argument('--no-other-configs', action='store_true', help="skip the `other_configs` legs (BASELINE.json configs[3], configs[4] per GPU, the Train notebook's BCE-Dice loss)")
    ap.add_argument('--other-steps', type=int, default=30, help='timed steps of every `other_configs` leg')
    args = ap.parse_args()

    import numpy as np
    import torch
    rvip = importlib.import_module('cmr-landmark-detection_amd')
    M = rvip.Loss_and_metrics

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('--gpus %d needs the torch.distributed.run launcher (one rank per GPU)' % args.gpus)
    backend = os.environ.get('RVIP_BENCH_BACKEND', 'nccl')       # 'gloo' only to rehearse the N>1 code path on one GPU
    ndev = max(torch.cuda.device_count(), 1)
    os.environ['LOCAL_RANK'] = str(local % ndev)                 # the engine picks its device from LOCAL_RANK
    torch.cuda.set_device(local % ndev)
    one_rank_pg = world == 1 and os.environ.get('RVIP_FORCE_DP_SCHEDULE') == '1'   # rehearsal on one GPU: the data-parallel schedule against a one-rank group
    if world > 1 or one_rank_pg:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29512')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local % ndev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    cfg = make_cfg(M, args.dim, args.filters, args.depth, args.frames, args.precision, args.loss)
    model = rvip.get_model(cfg, metrics=[])
    plan = model.plan
    B = args.batch
    gen = rvip.Generators.SyntheticSAXGenerator(B, dict(DIM=cfg['DIM'], BATCHSIZE=B, GAUS=True, SIGMA=2, SHUFFLE=False, SEED=42 + rank))
    x, y = gen[0]
    eng = model._engine(B)
    eng.load_input(x, y)                       # synthetic batch resident in HBM before the timed region
    torch.cuda.synchronize()
    if args.dump_labels and rank == 0:
        labs = ['rvip_convert (stage_input)'] + ['%s %s' % (th[0].__name__, th[2] if len(th) > 2 else '') for seq in (eng.fwd_train, eng.bwd, eng.opt) for th in seq]
        json.dump(labs, open(args.dump_labels, 'w'))

    # The step is the product's: Engine.train_step (what Model.fit / train_on_batch call) runs eagerly once, captures the
    # step into a hipGraph on its second call (around the RCCL collectives when N > 1) and replays it from then on.
    if args.no_graph:
        os.environ['RVIP_GRAPH'] = '0'
    run = eng.train_step
    for _ in range(max(2, min(args.warmup, 3))):
        run()
    torch.cuda.synchronize()
    launch = eng.launch_mode
    overlap = eng.overlap_ok()
    if (world > 1 or one_rank_pg) and launch.startswith('hipGraph'):
        launch += ' (two overlapped all-reduce buckets)' if overlap else ' (one all-reduce)'

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(eng.loss.item())
    assert np.isfinite(loss) or os.environ.get('RVIP_DBG'), 'training diverged in the benchmark'     # RVIP_DBG: timing ablations compute garbage

    # ---- data-parallel runs: where the step's time goes, segment by segment (every rank runs the same ten steps after the timed
    # region; events on the launch stream, Engine.train_step(marks=)).  'segment 1' is the encoder's backward pass with gradient
    # bucket 0 in flight beside it, 'collectives landed' what the launch stream then still waits for both buckets.
    dp_segments = None
    if world > 1 or one_rank_pg:
        acc = {}
        n_seg_steps = 10
        for _ in range(n_seg_steps):
            marks = []
            eng.train_step(marks=marks)
            torch.cuda.synchronize()
            for (_, e0), (what, e1) in zip(marks[:-1], marks[1:]):
                acc[what] = acc.get(what, 0.0) + e0.elapsed_time(e1)
            acc['step'] = acc.get('step', 0.0) + marks[0][1].elapsed_time(marks[-1][1])
        dp_segments = {k + ' ms': round(v / n_seg_steps, 4) for k, v in acc.items()}
        dp_segments['note'] = 'rank 0, mean of %d steps after the timed region; a host synchronisation between steps' % n_seg_steps
        barrier()

    # ---- inference (Model.predict: BN on the moving statistics, no dropout), forward only, eager launches; and the HBM the
    # whole training state of this configuration occupies.  Informational keys beside `value`.
    predict_rate = hbm_gb = fit_rate = fit_info = None
    if rank == 0 and world == 1 and not args.no_aux:
        for _ in range(2):
            eng.forward(training=False)
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        ps = max(3, min(args.steps, 10))
        for _ in range(ps):
            eng.stage_input()
            eng.forward(training=False)
        torch.cuda.synchronize()
        predict_rate = round(B * ps / (time.perf_counter() - tp0), 2)

    if rank == 0:
        hbm_gb = round(torch.cuda.max_memory_allocated() / 2 ** 30, 3)

    # ---- roofline pass: HIP events around every launch, eager, on the launch stream -------------------------
    roof = None
    per_kernel = {}
    hbm_step = None
    if rank == 0 and not args.no_roofline_pass:
        import ctypes as C
        s = torch.cuda.current_stream()
        L = rvip._native.lib()
        conv_fn, conv_stats_fn, conv_sums_fn, wgrad_fn = L.rvip_conv3x3_fwd, L.rvip_conv3x3_fwd_stats, L.rvip_conv3x3_fwd_sums, L.rvip_conv3x3_wgrad
        pair_fn = L.rvip_conv3x3_wgrad_dgrad           # weight + data gradient of a layer as the two parts of one grid
        reps = 5
        agg = {}
        detail = []
        # What the event pair adds to a launch's reading, calibrated on a trivial kernel in this run: its reading inside a busy
        # queue minus its back-to-back cost per launch (one event pair around 200 launches).  Subtracted per launch below.
        scratch = torch.zeros(64, dtype=torch.float32, device='cuda')
        cs_ = C.c_void_p(s.cuda_stream)

        def _triv():
            assert L.rvip_scale_f32(C.c_void_p(scratch.data_ptr()), C.c_longlong(1), C.c_float(1.0), cs_) == 0
        for _ in range(20):
            _triv()
        torch.cuda.synchronize()
        eb0, eb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        eb0.record(s)
        for _ in range(200):
            _triv()
        eb1.record(s)
        ovs = []
        for _ in range(100):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); _triv(); e1.record(s)
            ovs.append((e0, e1))
        torch.cuda.synchronize()
        triv_ms = eb0.elapsed_time(eb1) / 200
        ev_ovh_ms = max(sorted(a.elapsed_time(b) for a, b in ovs)[len(ovs) // 2] - triv_ms, 0.0)
        for _ in range(reps):
            eng.stage_input()
            for seq in (eng.fwd_train, eng.bwd, eng.opt):
                for th in seq:
                    fn, a = th[0], th[1]
                    lab = th[2] if len(th) > 2 else ''
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(s)
                    rc = fn(*a, C.c_void_p(s.cuda_stream))
                    e1.record(s)
                    assert rc == 0
                    fn = getattr(fn, 'fn', fn)              # (an entry of the second stream wraps its entry point)
                    flops = 0.0
                    if fn is conv_fn or fn is conv_stats_fn or fn is conv_sums_fn:      # the same igemm kernels; the others add the BN partial sums / the column sums
                        d = a[0]._obj
                        flops = 2.0 * d.n * d.h * d.w * 9 * max(d.kd, 1) * (d.c0 + d.c1) * d.cout
                    elif fn is wgrad_fn:
                        d = a[0]._obj
                        flops = 2.0 * d.n * d.h * d.w * 9 * max(d.kd, 1) * (d.c0 + d.c1) * d.cout
                    elif fn is pair_fn:                 # both gradients of the layer: twice the layer's forward FLOPs
                        d = a[0]._obj
                        flops = 2.0 * 2.0 * d.n * d.h * d.w * 9 * (d.c0 + d.c1) * d.cout
                    agg.setdefault(fn.__name__, []).append((e0, e1, flops))
                    if flops > 0:
                        detail.append((fn.__name__, (d.n, d.h, d.w, d.c0 + d.c1, d.cout, d.up0, 1 if d.c1 else 0), e0, e1, flops))
                    else:
                        detail.append((fn.__name__, (lab,), e0, e1, 0.0))
        torch.cuda.synchronize()
        if args.detail:
            rows = {}
            for name, shp, a, b, f in detail:
                r = rows.setdefault((name, shp), [0.0, 0, f])
                r[0] += a.elapsed_time(b); r[1] += 1
            with open(args.detail, 'w') as fh:
                fh.write('kernel n h w cin cout up cat launches avg_us tflops\n')
                for (name, shp), (ms, cnt, f) in sorted(rows.items(), key=lambda kv: -kv[1][0]):
                    fh.write('%s %s %d %.1f %.1f\n' % (name, ' '.join(map(str, shp)), max(cnt // reps, 1), 1e3 * ms / cnt, f / (ms / cnt * 1e-3) / 1e12))
        for name, evs in agg.items():
            ms = sum(a.elapsed_time(b) for a, b, _ in evs)
            fl = sum(f for _, _, f in evs)
            per_kernel[name] = dict(launches_per_step=len(evs) // reps, ms_per_step=round(ms / reps, 4),
                                    tflops=round(fl / (ms * 1e-3) / 1e12, 2) if (fl > 0 and ms > 0) else None)
        # THE family of the roofline: every MFMA contraction launch of the step -- the implicit-GEMM conv (forward, with the fused BN
        # statistics; data gradient), the weight gradient, and the pair launches that hold a layer's weight AND data gradient as the two
        # parts of one grid (a weight-gradient entry point includes its slab fold: the event pair brackets the C call).  Time = the
        # launches INSIDE the step (every launch behind its producer, as in the captured step and in the rocprofv3 kernel trace): HIP
        # events around each launch of the eager pass above, minus the calibrated event-pair cost.
        FAMILY = ('rvip_conv3x3_fwd', 'rvip_conv3x3_fwd_stats', 'rvip_conv3x3_fwd_sums', 'rvip_conv3x3_wgrad', 'rvip_conv3x3_wgrad_dgrad')
        TRACE_NAMES = ('conv3x3_igemm', 'wgrad3x3_', 'wgrad_dgrad_pair', 'wgrad_fold')       # the same launches in a kernel trace

        def part(names):
            evs = [e for nm in names for e in agg.get(nm, [])]
            if not evs:
                return None
            ms_ = max(sum(x.elapsed_time(y) for x, y, _ in evs) / reps - ev_ovh_ms * (len(evs) // reps), 1e-6)
            fl_ = sum(f for _, _, f in evs) / reps
            return {'launches_per_step': len(evs) // reps, 'ms_per_step': round(ms_, 4), 'gflop_per_step': round(fl_ / 1e9, 2),
                    'achieved': round(fl_ / (ms_ * 1e-3) / 1e12, 2), 'frac': round(fl_ / (ms_ * 1e-3) / 1e12 / (PEAK_BF16_TFLOPS if args.precision != 'fp32' else 157.3), 4)}
        cv = [e for nm in FAMILY for e in agg.get(nm, [])]
        fl = sum(f for _, _, f in cv)
        fam_ms = max(sum(x.elapsed_time(y) for x, y, _ in cv) / reps - ev_ovh_ms * (len(cv) // reps), 1e-6)
        achieved = (fl / reps) / (fam_ms * 1e-3) / 1e12
        parts = {nm: part((nm,)) for nm in FAMILY if agg.get(nm)}          # per entry point: forward with fused BN statistics / forward without
        #                                                                       (up-convs) / data gradient alone / weight gradient alone + fold / pair
        traffic, traffic_src, hbm_step, pmc_note = None, None, None, None
        try:                                   # HBM bytes per launch of this kernel family from the committed PMC passes of THESE kernels
            pm = json.load(open(os.path.join(ROOT, 'profiles', PMC_SUMMARY)))
            if pm.get('csrc_sha16') != csrc_sha16():
                pmc_note = '%s was taken on other kernel sources (%s, tree is %s): traffic not reported' % (PMC_SUMMARY, pm.get('csrc_sha16'), csrc_sha16())
            elif pm.get('workload') != workload_key(args):
                pmc_note = '%s is for workload %s' % (PMC_SUMMARY, pm.get('workload'))
            else:
                fam = [v for k, v in pm['kernels'].items() if any(k.startswith(t) for t in TRACE_NAMES)]
                if not fam:
                    pmc_note = '%s holds no kernel of the family' % PMC_SUMMARY
                else:
                    traffic = round(sum(f['hbm_bytes_per_launch'] * f['launches_per_step'] for f in fam) / max(len(cv) // reps, 1))
                    traffic_src = 'profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH x2, KiB units; bytes of the family per step / its entry-point launches)' % PMC_SUMMARY
                    hbm_step = pm.get('hbm_bytes_per_step')
        except FileNotFoundError:
            pmc_note = 'profiles/%s missing' % PMC_SUMMARY
        except (OSError, ValueError, KeyError, TypeError, ZeroDivisionError) as e:      # truncated / stale-schema file: report, do not lose the line
            traffic, traffic_src, hbm_step = None, None, None
            pmc_note = 'profiles/%s unusable (%s: %s)' % (PMC_SUMMARY, type(e).__name__, e)
        # the same family in the committed rocprofv3 kernel trace of this command (captured step), when taken on THESE sources
        prof = None
        try:
            import csv
            meta = json.load(open(os.path.join(ROOT, 'profiles', KERNEL_STATS + '.meta.json')))
            if meta.get('csrc_sha16') != csrc_sha16() or meta.get('workload') != workload_key(args):
                prof = {'file': 'profiles/' + KERNEL_STATS, 'note': 'taken on other kernel sources / workload (%s, %s)' % (meta.get('csrc_sha16'), meta.get('workload'))}
            else:
                rows_ = list(csv.DictReader(open(os.path.join(ROOT, 'profiles', KERNEL_STATS))))
                steps_p = max(int(r['Calls']) for r in rows_ if 'adam_kernel' in r['Name'])
                fam_ns = sum(float(r['TotalDurationNs']) for r in rows_ if any(t in r['Name'] for t in TRACE_NAMES))
                pa = (fl / reps) / (fam_ns / steps_p * 1e-9) / 1e12
                all_ns = sum(float(r['TotalDurationNs']) for r in rows_ if 'at::native' not in r['Name'] and 'rocclr' not in r['Name'])
                # boxes differ by several per cent in clock: besides the absolute rate, compare the family's SHARE of the step
                prof = {'file': 'profiles/' + KERNEL_STATS, 'steps': steps_p, 'family_ms_per_step': round(fam_ns / steps_p * 1e-6, 4),
                        'achieved': round(pa, 2), 'live_over_profile': round(achieved / pa, 4),
                        'step_ms_kernel_sum': round(all_ns / steps_p * 1e-6, 4), 'family_share_of_step': round(fam_ns / all_ns, 4),
                        'live_family_share_of_step': round(fam_ms / (1e3 * elapsed / args.steps), 4)}
        except FileNotFoundError:
            prof = {'file': 'profiles/' + KERNEL_STATS, 'note': 'missing'}
        except (OSError, ValueError, KeyError, TypeError, ZeroDivisionError) as e:
            prof = {'file': 'profiles/' + KERNEL_STATS, 'note': 'unusable (%s: %s)' % (type(e).__name__, e)}
        peak_ = PEAK_BF16_TFLOPS if args.precision != 'fp32' else 157.3
        esz_ = 2 if args.precision != 'fp32' else 4
        roof = dict(bound='mfma', kernel='MFMA contraction launches of the step: conv3x3_igemm_ws16 (forward incl. fused BN statistics, data gradient), wgrad3x3_ws '
                                         '(+ slab fold) and wgrad_dgrad_pair (a layer\'s weight and data gradient as the two parts of one grid)',
                    achieved=round(achieved, 2), peak=peak_, unit='TFLOP/s', frac=round(achieved / peak_, 4),
                    traffic=traffic, traffic_source=traffic_src, traffic_note=pmc_note,
                    algorithmic_bytes_per_launch=round(plan.ideal_bytes_per_slice(esz_) * B / max(len(cv) // reps, 1)),
                    launches_per_step=len(cv) // reps, avg_launch_ms=round(fam_ms / max(len(cv) // reps, 1), 4),
                    flops_per_launch=fl / max(len(cv), 1), family_ms_per_step=round(fam_ms, 4), gflop_per_step=round(fl / reps / 1e9, 2),
                    timing='HIP events around every launch of the family inside %d eager steps (each launch behind its producer, as in the captured step), '
                           'minus what the event pair adds to a reading (%.2f us, calibrated on a trivial kernel in this run) per launch' % (reps, 1e3 * ev_ovh_ms),
                    parts=parts, profile=prof)

    # ---- the product's own loop: Model.fit on a SyntheticSAXGenerator held in memory (the reference trains with in_memory=True,
    # train_model.py:199-203): generator -> rank shard -> pinned ring -> copy stream -> replayed step, per-epoch logs.  Reported
    # beside `value`, never as it (the contract times resident inputs).  Runs last: it trains on other batches.
    if rank == 0 and world == 1 and not args.no_fit:
        nb, ep = 44, max(2, min(4, args.steps // 5))             # 44 steps per epoch: the reference's fold, 1 426 slices / 32
        fgen = rvip.Generators.SyntheticSAXGenerator(nb * B, dict(DIM=cfg['DIM'], BATCHSIZE=B, GAUS=True, SIGMA=2, SHUFFLE=True, SEED=1), in_memory=True)
        model.fit(fgen, epochs=1, verbose=0, max_queue_size=4, workers=4)          # fills the generator's sample cache (untimed)
        torch.cuda.synchronize()
        tf0 = time.perf_counter()
        hist = model.fit(fgen, epochs=ep, verbose=0, max_queue_size=4, workers=4)
        torch.cuda.synchronize()
        fit_rate = round(nb * B * ep / (time.perf_counter() - tf0), 2)
        fit_info = {'epochs': ep, 'steps_per_epoch': nb, 'launch': eng.launch_mode, 'loss_first_last': [hist.history['loss'][0], hist.history['loss'][-1]],
                    'input': 'host float32 batches (x %.1f MB + y %.1f MB per step) through pinned ring + copy stream' % (
                        eng.x_stage.numel() * 4 / 1e6, eng.y_true.numel() * 4 / 1e6)}

    # ---- the other configurations BASELINE.json names that fit one GPU, and the Train notebook's loss: the same product path, a few
    # dozen steps each, AFTER everything the headline line reports (its `metric` / `value` / `config` are untouched by these legs)
    other = None
    headline = (args.dim, args.filters, args.depth, args.frames, args.precision, args.loss, B) == (256, 32, 4, 0, 'bf16', 'mse', 32)
    if rank == 0 and world == 1 and headline and not (args.no_other_configs or args.no_aux):
        other = {}
        for spec in OTHER_CONFIGS:
            try:
                other[spec[0]] = measure_other(rvip, spec, args.other_steps)
            except Exception as e:                                          # the headline line must survive a failing leg
                other[spec[0]] = {'config': spec[1], 'error': '%s: %s' % (type(e).__name__, e)}

    fwd_flops, step_flops = plan.flops_per_slice()
    slices = B * world * args.steps
    value = slices / elapsed
    out = None
    if rank == 0:
        out = {
            'metric': 'SAX slices/sec (fwd+bwd), 256x256 U-Net 2-heatmap' if args.frames <= 0 else 'cine volumes/sec (fwd+bwd), %dx%dx%d 3-D U-Net' % (args.frames, args.dim, args.dim),
            'value': round(value, 2), 'unit': 'slices/s' if args.frames <= 0 else 'volumes/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': {'bf16': 'bf16', 'fp16': 'f16', 'fp32': 'f32'}[args.precision], 'data': 'synthetic',
            'config': {'workload': '%d-level %s U-Net F=%d, %s, batch %d per GPU, fwd+loss(%s)+bwd+Adam%s' % (
                args.depth, '3D cine (Conv3D 3x3x3, pool 1x2x2)' if args.frames > 0 else '2D', args.filters,
                ('%dx%dx%d' % (args.frames, args.dim, args.dim)) if args.frames > 0 else '%dx%d' % (args.dim, args.dim), B,
                'MSE' if args.loss == 'mse' else 'BCE-Dice',
                (' + %s grad all-reduce%s' % ('RCCL' if backend == 'nccl' else backend, ' (ONE-rank group: schedule rehearsal)' if one_rank_pg else '')) if (world > 1 or one_rank_pg) else ''),
                'global_batch': B * world, 'parallelism': 'dp%d' % world, 'launch': launch, 'collective': ('%s all-reduce of %d fp32 gradients' % (backend, model._params.count)) if world > 1 else None,
                'gflop_per_slice_fwd_bwd': round(step_flops / 1e9, 3)},
            'fit_slices_per_s': fit_rate, 'fit': fit_info, 'predict_slices_per_s_eager': predict_rate, 'hbm_allocated_gib': hbm_gb,
            'hbm_bytes_per_step': hbm_step, 'algorithmic_bytes_per_step': round(plan.ideal_bytes_per_slice(2 if args.precision != 'fp32' else 4) * B),
            'wasted_traffic_ratio': round(hbm_step / (plan.ideal_bytes_per_slice(2 if args.precision != 'fp32' else 4) * B), 3) if hbm_step else None,
            'mfma_util_whole_step': round(value / world * step_flops / (PEAK_BF16_TFLOPS * 1e12), 4),
            'loss': loss, 'steps_executed_total': max(2, min(args.warmup, 3)) + args.warmup + args.steps + (0 if args.no_roofline_pass else 3) + (10 if dp_segments else 0),
            'other_configs': other,
            'dp_segments': dp_segments,
            'roofline': roof,
            'kernels': per_kernel,
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg, args.cpu_batch)
    if world > 1 or one_rank_pg:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


def cpu_baseline(cfg, batch):
    """The same step on the host cores: PyTorch-CPU (oneDNN) fp32 port in oracle/torch_ref.py -- the stand-in for the
    reference's TF2-CPU path, which cannot be imported here.  Bounded sample: `batch` slices per step, 1 warm-up + 8
    timed steps (about 12 s at config 2 on 16 cores)."""
    import torch
    from oracle import torch_ref
    cores = len(os.sched_getaffinity(0))
    try:                                       # respect the cgroup CPU quota (the GPU box grants a share of its cores)
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    cores = min(cores, int(os.environ.get('RVIP_CPU_BASELINE_THREADS', 16)))
    cpu_cfg = {k: v for k, v in cfg.items() if k not in ('LOSS_FUNCTION', 'RVIP_PRECISION')}
    nsteps = 8
    sec = torch_ref.time_train_steps(cpu_cfg, batch, steps=nsteps, warmup=1, threads=cores)
    return {'value': round(batch / sec, 3), 'unit': 'slices/s', 'cores': cores, 'kind': 'port',
            'sample': '%d slices/step x %d timed steps of the same %dx%d training step, PyTorch-CPU fp32 (stand-in for TF2-CPU)' % (
                batch, nsteps, cfg['DIM'][0], cfg['DIM'][1]), 'threads': torch.get_num_threads()}


if __name__ == '__main__':
    main()
