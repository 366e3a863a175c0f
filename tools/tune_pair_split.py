"""Per paired layer of config 2: the pair launch (rvip_conv3x3_wgrad_dgrad + its slab fold) timed for several splits of the 256 CUs
between the weight gradient and the data gradient (cu_limit W / 256 - W), back to back (cache state: warm from itself).
    python tools/tune_pair_split.py [reps]"""
import ctypes as C, os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rvip = importlib.import_module('cmr-landmark-detection_amd')
N = rvip._native
M = rvip.Loss_and_metrics
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = dict(DIM=[256, 256], FILTERS=32, DEPTH=4, BATCH_NORMALISATION=True, ACTIVATION='relu', MASK_CLASSES=2, LEARNING_RATE=1e-4, RVIP_PRECISION='bf16', LOSS_FUNCTION=M.mse, SEED=42)
B = 32
model = rvip.get_model(cfg, metrics=[])
gen = rvip.Generators.SyntheticSAXGenerator(B, dict(DIM=cfg['DIM'], BATCHSIZE=B, GAUS=True, SIGMA=2, SHUFFLE=False, SEED=42))
x, y = gen[0]
eng = model._engine(B)
eng.load_input(x, y)
os.environ['RVIP_GRAPH'] = '0'
eng.train_step()
torch.cuda.synchronize()
L = N.lib()
s = torch.cuda.current_stream()
cs = C.c_void_p(s.cuda_stream)
splits = [96, 104, 112, 120, 128, 136, 144]
print('layer | ' + ' '.join('%6d' % w for w in splits))
tot = {w: 0.0 for w in splits}
best_tot = 0.0
for th in eng.bwd:
    if getattr(th[0], '__name__', '') != 'rvip_conv3x3_wgrad_dgrad':
        continue
    wg0, dg0 = th[1][0]._obj, th[1][1]._obj
    row = []
    for w in splits:
        wg, dg = type(wg0).from_buffer_copy(wg0), type(dg0).from_buffer_copy(dg0)
        wg.cu_limit, dg.cu_limit = w, 256 - w
        if not L.rvip_conv3x3_wgrad_dgrad_ok(C.byref(wg), C.byref(dg)):
            row.append(float('nan')); continue
        rows = L.rvip_conv3x3_fwd_sums_rows(C.byref(dg))
        sums = torch.zeros(max(rows, 1) * dg.cout + 64, dtype=torch.float32, device='cuda')
        ns = L.rvip_conv3x3_wgrad_splits(C.byref(wg))
        need = ns * 9 * (wg.c0 + wg.c1) * wg.cout * 4
        wsb = torch.empty(need // 4 + 64, dtype=torch.float32, device='cuda')
        wg.workspace, wg.workspace_bytes = wsb.data_ptr(), need
        if wg.dot_rows:
            nd = L.rvip_conv3x3_wgrad_dot_rows(C.byref(wg))
            dots = torch.zeros(nd * (wg.c0 + wg.c1), dtype=torch.float64, device='cuda')
            wg.dot_rows, wg.dot_rows_bytes = dots.data_ptr(), dots.numel() * 8
        args = (C.byref(wg), C.byref(dg), C.c_void_p(sums.data_ptr()), C.c_size_t(sums.numel() * 4))
        for _ in range(3):
            assert L.rvip_conv3x3_wgrad_dgrad(*args, cs) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            L.rvip_conv3x3_wgrad_dgrad(*args, cs)
        e1.record(s)
        torch.cuda.synchronize()
        row.append(1e3 * e0.elapsed_time(e1) / reps)
    lab = th[2] if len(th) > 2 else ''
    print('%-44s | %s' % (lab[:44], ' '.join('%6.1f' % v for v in row)), flush=True)
    for w, v in zip(splits, row):
        tot[w] += v
    best_tot += min(v for v in row if v == v)
print('%-44s | %s' % ('sum', ' '.join('%6.1f' % tot[w] for w in splits)))
print('sum of per-layer minima %.1f' % best_tot)
