"""Host logic of fit()'s input pipeline and of object lifetime, no GPU.

Round 3's GPU suite aborted (SIGABRT) on the driver's box: a Model in a reference cycle was destroyed by the cyclic collector
while another engine was capturing its step; the hipGraph destructor synchronises the device, fails during a capture and throws
from a destructor (tools/repro_graph_gc_abort.py reproduces it on a GPU).  These tests hold the host-side
invariants of the fix:
  * a Model is NOT part of a reference cycle -- it dies by reference count, on the thread that drops it;
  * the stager thread of fit() is host-only and is JOINED by close(), pool included;
  * the pinned-slot hand-back protocol (engine.InputRing) cannot deadlock at slots = depth + 4 even when no upload ever reports
    completion early, and delivers every batch intact and in order.
"""
import gc
import importlib
import threading
import time
import weakref

import numpy as np
import pytest

import cmr_landmark_detection_amd as rvip

M = rvip.Loss_and_metrics
E = importlib.import_module('cmr-landmark-detection_amd.engine')
K = importlib.import_module('cmr-landmark-detection_amd.keras_model')
G = importlib.import_module('cmr-landmark-detection_amd.Generators')


def _cfg(**kw):
    c = dict(DIM=[32, 32], FILTERS=8, DEPTH=2, BATCH_NORMALISATION=True, ACTIVATION='relu', MASK_CLASSES=2, LOSS_FUNCTION=M.mse)
    c.update(kw)
    return c


def test_model_is_not_in_a_reference_cycle():
    was = gc.isenabled()
    gc.disable()
    try:
        m = rvip.get_model(_cfg())                         # compiled: optimizer.lr holds a listener of the model
        m.compile(optimizer=m.optimizer, loss=M.mse)       # compiling again must not stack listeners either
        assert len(m.optimizer.lr._listeners) == 1
        cb = m.history_callback()
        seen = []
        m._params = type('P', (), {'set_lr': lambda self, v: seen.append(v)})()
        m.optimizer.lr = 5e-4                              # the listener still works ...
        assert seen == [5e-4]
        m._params = None
        ref = weakref.ref(m)
        opt = m.optimizer
        del m, cb
        assert ref() is None, 'the model survived its last reference: it sits in a cycle (gc is off in this test)'
        opt.lr = 1e-4                                      # ... and a dead model's listener is skipped, not called
        assert all(r() is None for r in opt.lr._listeners)
    finally:
        if was:
            gc.enable()


class _FakeEvent:
    """an upload that never reports completion early: query() False, synchronize() returns"""
    def __init__(self, log):
        self.log = log

    def query(self):
        return False

    def synchronize(self):
        self.log.append('sync')


class _FakeRing(E.InputRing):
    """engine.InputRing with host arrays in place of pinned / device memory: the protocol itself is the product's code"""

    def __init__(self, shape_x, shape_y):
        self.shape_x, self.shape_y = shape_x, shape_y
        self.device_x = None
        self.log = []
        self.main_thread = threading.get_ident()
        self.loaded = []

    def _ring_alloc(self, slots):
        assert threading.get_ident() == self.main_thread
        self.pin_x = [np.zeros(self.shape_x, np.float32) for _ in range(slots)]
        self.pin_y = [np.zeros(self.shape_y, np.float32) for _ in range(slots)]
        self.pin_x_np, self.pin_y_np = self.pin_x, self.pin_y

    def _ring_upload(self, slot, d):
        assert threading.get_ident() == self.main_thread, 'uploads belong to the training thread'
        self.device_x = self.pin_x[slot].copy()
        self.device_y = self.pin_y[slot].copy()
        return _FakeEvent(self.log)

    def load_input(self, x, y):                            # the unpinned first batch
        assert threading.get_ident() == self.main_thread
        self.device_x, self.device_y = np.array(x, np.float32), np.array(y, np.float32)


class _FakeModel:
    def __init__(self):
        self._rings, self.engines = {}, {}

    def _dist(self):
        return 0, 1

    def _shard(self, x, y):
        return x, y

    def _engine(self, batch):
        if batch not in self.engines:
            self.engines[batch] = _FakeRing((batch, 4, 4, 1), (batch, 4, 4, 2))
        return self.engines[batch]


class _Gen:
    """batch i is filled with the value 1000 * epoch + i; records which threads called it"""
    BATCHSIZE = 3

    def __init__(self, n, fail_at=None):
        self.n, self.epoch, self.fail_at = n, 0, fail_at
        self.threads = set()
        self.epoch_ends = 0

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        self.threads.add(threading.get_ident())
        if self.fail_at is not None and i == self.fail_at:
            raise RuntimeError('generator failed at %d' % i)
        time.sleep(0.001 * (i % 3))
        v = 1000 * self.epoch + i
        return np.full((3, 4, 4, 1), v, np.float32), np.full((3, 4, 4, 2), -v, np.float32)

    def on_epoch_end(self):
        self.epoch += 1
        self.epoch_ends += 1


def _orders(gen, epochs):
    for ep in range(epochs):
        yield np.random.default_rng(ep).permutation(len(gen))


def _threads():
    return {t.name for t in threading.enumerate()}


@pytest.mark.parametrize('depth,workers', [(1, 1), (2, 3), (12, 2)])
def test_stager_delivers_every_batch_in_order_and_joins_its_threads(depth, workers):
    before = threading.active_count()
    model, gen = _FakeModel(), _Gen(17)
    st = K._Stager(model, gen, _orders(gen, 3), depth, workers)
    assert st.slots == depth + 4
    try:
        for ep in range(3):
            want = np.random.default_rng(ep).permutation(17)
            got = []
            for step, (eng, slot, steps) in enumerate(st.epoch()):
                assert steps == 17
                if slot is not None:
                    eng.feed(slot)
                got.append(int(eng.device_x.flat[0]))
                assert (eng.device_x == eng.device_x.flat[0]).all() and (eng.device_y == -eng.device_x.flat[0]).all()
                if step % 5 == 0:
                    time.sleep(0.003)                      # a slow consumer now and then: the stager runs into the full ring
            assert got == [1000 * ep + int(i) for i in want]
    finally:
        st.close()
    assert not st.alive() and threading.active_count() == before, _threads()
    assert threading.get_ident() not in gen.threads        # the generator ran on the stager / its pool only
    assert gen.epoch_ends == 3
    eng = model.engines[3]
    assert all(f.is_set() for f in eng.slot_free) and not eng._uploads      # close() left the ring reusable
    # a second fit() on the same model reuses the ring from its first batch on
    gen2 = _Gen(5)
    st2 = K._Stager(model, gen2, _orders(gen2, 1), depth, workers)
    try:
        slots = []
        for eng2, slot, _ in st2.epoch():
            slots.append(slot)
            eng2.feed(slot)
        assert len(slots) == 5 and all(s is not None for s in slots)
    finally:
        st2.close()
    assert threading.active_count() == before


def test_stager_close_in_mid_epoch_joins_and_frees_the_ring():
    before = threading.active_count()
    model, gen = _FakeModel(), _Gen(200)
    st = K._Stager(model, gen, _orders(gen, 5), 4, 3)
    it = st.epoch()
    for _ in range(7):
        eng, slot, _ = next(it)
        if slot is not None:
            eng.feed(slot)
    st.close()                                             # what fit() does when a callback stops the training or raises
    assert not st.alive() and threading.active_count() == before, _threads()
    assert all(f.is_set() for f in model.engines[3].slot_free)


def test_generator_error_surfaces_in_the_training_thread():
    before = threading.active_count()
    model, gen = _FakeModel(), _Gen(10, fail_at=4)
    st = K._Stager(model, gen, iter([np.arange(10)]), 2, 1)
    with pytest.raises(RuntimeError, match='generator failed at 4'):
        try:
            for eng, slot, _ in st.epoch():
                if slot is not None:
                    eng.feed(slot)
        finally:
            st.close()
    assert threading.active_count() == before


def test_second_fit_while_a_stager_lives_is_refused():
    m = rvip.get_model(_cfg())
    gate = threading.Event()

    class _Alive:
        def alive(self):
            return not gate.is_set()
    a = _Alive()
    m._stager = weakref.ref(a)
    with pytest.raises(RuntimeError, match='still alive'):
        m._staged_batches(_Gen(1), iter([]), 1, 1)
    gate.set()


def test_augmentation_draws_belong_to_the_sample_not_to_the_call_order(tmp_path):
    """2 ranks x half a batch == 1 process x the whole batch, with AUGMENT on (a shared sequential stream gave every
    rank the same parameter sequence for different samples)."""
    rng = np.random.default_rng(0)
    files_x, files_y = [], []
    for i in range(8):
        img = rng.random((40, 36)).astype(np.float32)
        lab = np.zeros((40, 36), np.int16)
        lab[10 + i, 12] = 1
        lab[20, 8 + i] = 2
        fx, fy = str(tmp_path / ('x%d.npy' % i)), str(tmp_path / ('y%d.npy' % i))
        np.save(fx, img)
        np.save(fy, lab)
        files_x.append(fx)
        files_y.append(fy)
    cfg = dict(DIM=[32, 32], BATCHSIZE=4, MASK_VALUES=[1, 2], AUGMENT=True, AUGMENT_PROB=1.0, SHUFFLE=True, SEED=7, GAUS=True, SIGMA=2)
    whole = G.DataGenerator(files_x, files_y, cfg)
    r0, r1 = G.DataGenerator(files_x, files_y, cfg), G.DataGenerator(files_x, files_y, cfg)
    for epoch in range(2):
        for i in range(len(whole)):
            x, y = whole[i]
            x1, y1 = r1.batch_slice(i, 2, 4)               # rank 1 first: call order must not matter
            x0, y0 = r0.batch_slice(i, 0, 2)
            np.testing.assert_array_equal(np.concatenate([x0, x1]), x)
            np.testing.assert_array_equal(np.concatenate([y0, y1]), y)
        first = whole[0][0].copy()
        for g in (whole, r0, r1):
            g.on_epoch_end()
        assert not np.array_equal(whole[0][0], first)      # a new epoch draws new augmentations


def test_a_fetch_keeps_the_epoch_it_started_under(tmp_path):
    """the stager thread calls on_epoch_end() as soon as an epoch's last batch is produced; a fetch from another thread that
    overlaps it (evaluate / predict on the same generator, gen[i] in user code) must see ONE epoch's order AND augmentation seed."""
    rng = np.random.default_rng(1)
    files_x, files_y = [], []
    for i in range(8):
        lab = np.zeros((40, 36), np.int16)
        lab[10 + i, 12] = 1
        lab[20, 8 + i] = 2
        fx, fy = str(tmp_path / ('x%d.npy' % i)), str(tmp_path / ('y%d.npy' % i))
        np.save(fx, rng.random((40, 36)).astype(np.float32))
        np.save(fy, lab)
        files_x.append(fx)
        files_y.append(fy)
    cfg = dict(DIM=[32, 32], BATCHSIZE=4, MASK_VALUES=[1, 2], AUGMENT=True, AUGMENT_PROB=1.0, SHUFFLE=True, SEED=7, GAUS=True, SIGMA=2)
    ref, gen = G.DataGenerator(files_x, files_y, cfg), G.DataGenerator(files_x, files_y, cfg)
    want_x, want_y = ref[1]

    class _Racing(type(gen)):
        def __data_generation__(self, idxs):                   # the other thread's on_epoch_end() lands inside the fetch
            type(gen).on_epoch_end(self)
            return super().__data_generation__(idxs)
    gen.__class__ = _Racing
    x, y = gen[1]
    np.testing.assert_array_equal(x, want_x)
    np.testing.assert_array_equal(y, want_y)
    ref.on_epoch_end()
    gen.__class__ = type(ref)
    np.testing.assert_array_equal(gen[1][0], ref[1][0])         # and the next epoch is the next epoch


def test_every_generator_class_serves_fetches_and_slices():
    """The fetch path calls __data_generation__ with the reference's one-argument signature, also where a subclass overrides it."""
    x = np.arange(6 * 4 * 4, dtype=np.float32).reshape(6, 4, 4, 1)
    y = np.stack([x[..., 0], -x[..., 0]], -1)
    for shuffle in (False, True):
        g = G.ArrayGenerator(x, y, 2, shuffle=shuffle)
        xb, yb = g[1]
        order = list(g.INDICES[2:4])
        np.testing.assert_array_equal(xb, x[order])
        np.testing.assert_array_equal(yb, y[order])
        np.testing.assert_array_equal(g.batch_slice(1, 1, 2)[0], x[order[1:]])
    s = G.SyntheticSAXGenerator(4, dict(DIM=[16, 16], BATCHSIZE=2, MASK_VALUES=[1, 2], GAUS=True, SIGMA=2))
    xb, yb = s[0]
    assert xb.shape == (2, 16, 16, 1) and yb.shape == (2, 16, 16, 2)


def test_single_threaded_ring_of_four_slots_never_waits_forever():
    """evaluate() / predict() drive a ring from ONE thread (engine.EvalRing: stage, feed, next batch) with four slots: staging batch
    k needs slot k % 4, which feeding batch k - 2 handed back -- also when no upload ever reports completion early, and for inputs
    without targets (predict)."""
    assert E.EvalRing.SLOTS == 4
    ring = _FakeRing((2, 4, 4, 1), (2, 4, 4, 2))
    ring.alloc_input_ring(E.EvalRing.SLOTS)
    done = threading.Event()
    seen = []

    def drive():
        for k in range(23):
            x = np.full((2, 4, 4, 1), k, np.float32)
            y = None if k % 3 == 0 else np.full((2, 4, 4, 2), -k, np.float32)
            slot = ring.next_slot()
            assert ring.stage_host_batch(slot, x, y)
            assert ring.slot_has_y[slot] == (y is not None)
            ring.feed(slot)
            seen.append(float(ring.device_x.ravel()[0]))
        done.set()
    ring.main_thread = None                                 # (the twin's thread assertions are for fit(): set below)
    th = threading.Thread(target=lambda: (setattr(ring, 'main_thread', threading.get_ident()), drive()), daemon=True)
    th.start()
    assert done.wait(20), 'a slot was never handed back: the single-threaded protocol deadlocked'
    assert seen == [float(k) for k in range(23)]
    ring.reset_input_ring()
    assert all(f.is_set() for f in ring.slot_free)
