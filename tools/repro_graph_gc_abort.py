"""Round 3's GPU-suite abort (GPUTEST_r03: rc 134, SIGABRT) reproduced deterministically, and the fix shown to hold.

Mechanism: a Model that sits in a Python reference cycle is destroyed by the cyclic collector; its engines own hipGraphs;
at::cuda::CUDAGraph::~CUDAGraph calls hipDeviceSynchronize and THROWS on its error from the destructor -> std::terminate ->
abort().  hipDeviceSynchronize fails whenever any stream is capturing.  So: garbage model + a collection that happens to run
while the next model captures its step = SIGABRT.  Which allocation triggers the collection depends on thread timing (fit() runs a
stager thread and pool workers) -- hence box-dependent.

    python tools/repro_graph_gc_abort.py            # parent: runs the two children below, prints their exit codes
    child 'old': cycle + a collection inside an UNGUARDED capture (what round 3 shipped)   -> expected rc -6 / 134
    child 'new': the same garbage, Engine.capture() of this round (collect first, collector off while capturing) -> rc 0
"""
import gc
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(mode):
    import numpy as np
    import torch
    import cmr_landmark_detection_amd as rvip
    from oracle import rvip_oracle as O
    M = rvip.Loss_and_metrics
    cfg = dict(DIM=[32, 32], FILTERS=8, DEPTH=2, BATCH_NORMALISATION=True, ACTIVATION='relu', MASK_CLASSES=2,
               LEARNING_RATE=1e-3, RVIP_PRECISION='bf16', LOSS_FUNCTION=M.mse, SEED=11)
    x, y = O.synthetic_batch(4, cfg['DIM'], 2, seed=1)
    gc.disable()                                   # the collector runs only where this script says so
    a = rvip.get_model(cfg)
    for _ in range(3):
        a.train_on_batch(x, y)                     # third call replays a captured graph
    assert next(iter(a._engines.values()))._graphs is not None
    a._cycle = a                                   # what Model -> optimizer -> lr listener -> Model was until round 3
    del a                                          # garbage now, alive until a collection
    b = rvip.get_model(cfg)
    b.train_on_batch(x, y)                         # eager warm-up step
    eng = next(iter(b._engines.values()))
    if mode == 'old':
        parts, _ = eng._step_parts()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            parts[0]()
            gc.collect()                           # stands for "some thread's allocation triggered a collection right now"
        print('old: survived (hypothesis refuted on this runtime)')
    else:
        gc.enable()
        orig = eng._step_parts

        def spying():
            parts, buckets = orig()

            def first():
                assert not gc.isenabled(), 'automatic collection must be off inside the capture'
                parts[0]()
            return [first] + parts[1:], buckets
        eng._step_parts = spying
        assert eng.capture()
        eng._step_parts = orig
        b.train_on_batch(x, y)
        assert gc.isenabled()
        print('new: captured with the garbage model released beforehand, loss', b.train_on_batch(x, y)[0])
    torch.cuda.synchronize()


if __name__ == '__main__':
    if len(sys.argv) > 1:
        child(sys.argv[1])
        sys.exit(0)
    for mode in ('old', 'new'):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), mode], capture_output=True, text=True, timeout=300)
        print('=== child %s: rc %d' % (mode, r.returncode))
        out = r.stdout + r.stderr
        print(out if len(out) < 2400 else out[:1200] + '\n...\n' + out[-800:])
