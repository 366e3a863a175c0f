"""Which of fit()'s per-step stream operations costs the time between consecutive replays of the captured step.

    python tools/probe_fit_gap.py [--steps 150]

Variants (each: ms per step over `steps` replays, device-synchronised at both ends):
  replay          the bench loop
  +d2d            the two device-to-device input copies in front of every replay (torch copy_ on the launch stream)
  +d2d+hist       and the 64-byte log-row copy behind it
  +h2d            pinned -> device on the copy stream, event, wait on the launch stream, then the two copies (InputRing.feed without the slot protocol)
  feed            InputRing.feed itself (events handed back, slots released)"""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=150)
    a = ap.parse_args()
    import torch
    rvip = importlib.import_module('cmr-landmark-detection_amd')
    M = rvip.Loss_and_metrics
    cfg = dict(DIM=[256, 256], FILTERS=32, DEPTH=4, BATCH_NORMALISATION=True, ACTIVATION='relu', MASK_CLASSES=2,
               LEARNING_RATE=1e-4, RVIP_PRECISION='bf16', LOSS_FUNCTION=M.mse, SEED=42)
    model = rvip.get_model(cfg, metrics=[])
    gen = rvip.Generators.SyntheticSAXGenerator(32, dict(DIM=cfg['DIM'], BATCHSIZE=32, GAUS=True, SIGMA=2, SHUFFLE=False, SEED=42))
    x, y = gen[0]
    eng = model._engine(32)
    eng.load_input(x, y)
    for _ in range(4):
        eng.train_step()
    torch.cuda.synchronize()
    eng.alloc_input_ring(6)
    for s in range(6):
        eng.stage_host_batch(s, x, y)
    hist = torch.zeros((a.steps, eng.sums.numel()), dtype=torch.float32, device='cuda')
    dx, dy = eng.dev_in[0]
    cs = eng.copy_stream
    main_s = torch.cuda.current_stream()

    def v_replay(i):
        eng.train_step()

    def v_d2d(i):
        eng.x_stage.copy_(dx, non_blocking=True); eng.y_true.copy_(dy, non_blocking=True)
        eng.train_step()

    def v_d2d_hist(i):
        v_d2d(i)
        hist[i].copy_(eng.sums, non_blocking=True)

    def v_h2d(i):
        d = i & 1
        with torch.cuda.stream(cs):
            eng.dev_in[d][0].copy_(eng.pin_x[i % 6], non_blocking=True)
            eng.dev_in[d][1].copy_(eng.pin_y[i % 6], non_blocking=True)
            ev = torch.cuda.Event(); ev.record(cs)
        main_s.wait_event(ev)
        eng.x_stage.copy_(eng.dev_in[d][0], non_blocking=True); eng.y_true.copy_(eng.dev_in[d][1], non_blocking=True)
        eng.train_step()
        hist[i].copy_(eng.sums, non_blocking=True)

    st = {'free': [None, None]}

    def mk(record_pos, wait_free, reuse_events=False):
        pool = [torch.cuda.Event() for _ in range(8)]

        def fn(i):
            d = i & 1
            if wait_free and st['free'][d] is not None:
                cs.wait_event(st['free'][d])
            with torch.cuda.stream(cs):
                eng.dev_in[d][0].copy_(eng.pin_x[i % 6], non_blocking=True)
                eng.dev_in[d][1].copy_(eng.pin_y[i % 6], non_blocking=True)
                ev = pool[i % 8] if reuse_events else torch.cuda.Event()
                ev.record(cs)
            main_s.wait_event(ev)
            eng.x_stage.copy_(eng.dev_in[d][0], non_blocking=True); eng.y_true.copy_(eng.dev_in[d][1], non_blocking=True)
            if record_pos == 'before':
                evf = torch.cuda.Event(); evf.record(main_s); st['free'][d] = evf
            eng.train_step()
            if record_pos == 'after':
                evf = torch.cuda.Event(); evf.record(main_s); st['free'][d] = evf
            hist[i].copy_(eng.sums, non_blocking=True)
        return fn

    def mk_host(D, devwait=True, hostwait_h2d=False):
        devs = [(torch.empty_like(dx), torch.empty_like(dy)) for _ in range(D)]
        evfs = [None] * D

        def fn(i):
            d = i % D
            if evfs[d] is not None:
                evfs[d].synchronize()                      # HOST waits until the copies that read this pair have run
            with torch.cuda.stream(cs):
                devs[d][0].copy_(eng.pin_x[i % 6], non_blocking=True)
                devs[d][1].copy_(eng.pin_y[i % 6], non_blocking=True)
                ev = torch.cuda.Event(); ev.record(cs)
            if hostwait_h2d:
                ev.synchronize()
            elif devwait:
                main_s.wait_event(ev)
            eng.x_stage.copy_(devs[d][0], non_blocking=True); eng.y_true.copy_(devs[d][1], non_blocking=True)
            evf = torch.cuda.Event(); evf.record(main_s); evfs[d] = evf
            eng.train_step()
            hist[i].copy_(eng.sums, non_blocking=True)
        return fn

    def v_feed(i):
        eng.feed(i % 6)
        eng.train_step()
        hist[i].copy_(eng.sums, non_blocking=True)

    for name, fn in (('replay', v_replay), ('+d2d', v_d2d), ('+d2d+hist', v_d2d_hist), ('+h2d', v_h2d), ('feed', v_feed), ('rec-before', mk('before', False)), ('rec-before+waitfree', mk('before', True)), ('rec-after+waitfree', mk('after', True)), ('norec+h2d', mk('none', False)), ('hostwait D=2', mk_host(2)), ('hostwait D=2 no main wait (racy)', mk_host(2, devwait=False)), ('hostwait D=2, host waits h2d', mk_host(2, hostwait_h2d=True)), ('replay', v_replay)):
        for i in range(5):
            fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            fn(i)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('%-12s %.4f ms/step   (host loop %.4f ms/step)' % (name, 1e3 * dt / a.steps, 1e3 * host / a.steps), flush=True)


if __name__ == '__main__':
    main()
