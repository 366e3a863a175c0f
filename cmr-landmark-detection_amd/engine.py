"""HIP execution engine of the fused U-Net plan (one process = one MI355X).

PyTorch is plumbing here: it owns device memory (``torch.empty``), the stream and the process group; every
FLOP of the step is a kernel of librvip_hip.so launched through the C ABI (``_native``).  No tensor op of
torch touches the hot path, and there is no CPU fallback: without the library or a GPU this module raises.

Data layout in HBM (per rank)
  * parameters: ONE flat fp32 block ``theta`` (Keras HWIO kernels, biases, gamma, beta; every tensor starts on
    a 256-byte boundary) with twin blocks ``grad``, ``adam_m``, ``adam_v``: the optimiser and the RCCL
    all-reduce see a single contiguous buffer;
  * BN moving statistics: a second flat fp32 block (not trained, not all-reduced per step);
  * packed conv operands in the activation dtype (bf16/f32): [9][Cout][Cin] forward, [9][Cin][Cout] dgrad;
  * activations NHWC in the activation dtype, one buffer per tensor of the fused plan (z = conv output kept
    for backward, y = BN/dropout output consumed by the next conv, pooled); gradients mirror them;
  * ``state``: 8 device words (step, lr, dropout seed) read by the kernels so a captured hipGraph replays
    with fresh values.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import gc
import os
import threading
from collections import OrderedDict, deque

import numpy as np

from . import _native as N

BN_MOMENTUM, BN_EPS = 0.99, 1e-3          # Keras BatchNormalization defaults (KerasLayers.py:684 passes none)
ALIGN = 64                                 # floats: every parameter tensor starts 256-byte aligned


_CAPTURE_LOCK = threading.RLock()            # one stream capture at a time per process; graph release never overlaps one
_GRAVEYARD = []                              # captured graphs of engines that died without release(): see Engine.__del__


def _bury_dead_graphs():
    """(under _CAPTURE_LOCK, device idle, no capture running on this thread) destroys the graphs dead engines left behind"""
    while _GRAVEYARD:
        _GRAVEYARD.pop()


def _torch():
    import torch
    return torch


@contextlib.contextmanager
def capture_guard():
    """Everything that captures a hipGraph in this process does it inside this guard (Engine.capture, bench.py's family graphs):
    one capture at a time; the device idle and Python's garbage collected BEFORE it (dead engines release their graphs here, on
    this thread, legally); the automatic collector off -- it is process-global, so for every thread -- until the capture has ended.
    See Engine.capture for why."""
    torch = _torch()
    with _CAPTURE_LOCK:
        torch.cuda.synchronize()
        gc.collect()
        _bury_dead_graphs()
        was_on = gc.isenabled()
        gc.disable()
        try:
            yield
        finally:
            if was_on:
                gc.enable()


def _dist_ready():
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()
    except Exception:
        return False


def require_gpu():
    torch = _torch()
    N.lib()
    if not torch.cuda.is_available():
        raise N.RvipError('no HIP device visible: the RVIP engine has no CPU path (oracle/ is test-only)')
    return torch


def _ptr(t, offset_elems=0):
    return C.c_void_p(t.data_ptr() + offset_elems * t.element_size())


class _OnSide:
    """A launch-list entry that goes to the engine's SECOND stream (between a _Fork and a _Join): called like the C entry point it
    wraps -- (*args, stream) -- and ignores the stream it is handed."""

    def __init__(self, eng, fn):
        self.eng, self.fn, self.__name__ = eng, fn, fn.__name__

    def __call__(self, *a):
        if os.environ.get('RVIP_BWD_OVERLAP_SERIAL') == '1':          # (diagnosis: the same grids, one stream)
            return self.fn(*a)
        return self.fn(*a[:-1], C.c_void_p(self.eng.side_stream.cuda_stream))


class _Fork:
    """second stream <- everything queued so far on the current one (capturable: an event edge)"""
    __name__ = 'rvip_fork'

    def __init__(self, eng):
        self.eng = eng

    def __call__(self, *a):
        torch = _torch()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.eng.side_stream.wait_event(ev)
        return 0


class _Join:
    """current stream <- everything queued so far on the second one"""
    __name__ = 'rvip_join'

    def __init__(self, eng):
        self.eng = eng

    def __call__(self, *a):
        torch = _torch()
        ev = torch.cuda.Event()
        ev.record(self.eng.side_stream)
        torch.cuda.current_stream().wait_event(ev)
        return 0


class ParamStore:
    """Flat device parameter blocks + packed conv operands, shared by every per-batch-size Engine."""

    def __init__(self, plan, host_weights, dtype, device, seed=42, lr=1e-3):
        torch = require_gpu()
        self.plan, self.device = plan, device
        self.dt = {'bf16': N.BF16, 'f16': N.F16, 'f32': N.F32}[dtype]
        self.tdtype = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[dtype]
        self.grad_unscale = 1.0                   # 1 / loss scale of the engine that last wrote self.grad (f16 only)
        self.off, self.moff = OrderedDict(), OrderedDict()
        n_tr = n_mv = 0
        for (lname, wname, shape, trainable, _), arr in zip(plan.weight_specs(), host_weights):
            size = int(np.prod(shape))
            if trainable:
                self.off[(lname, wname)] = (n_tr, tuple(shape))
                n_tr += -(-size // ALIGN) * ALIGN
            else:
                self.moff[(lname, wname)] = (n_mv, tuple(shape))
                n_mv += -(-size // ALIGN) * ALIGN
        self.count = n_tr
        f32 = dict(dtype=torch.float32, device=device)
        self.theta = torch.zeros(n_tr, **f32)
        self.grad = torch.zeros(n_tr, **f32)
        self.adam_m = torch.zeros(n_tr, **f32)
        self.adam_v = torch.zeros(n_tr, **f32)
        self.moving = torch.zeros(max(n_mv, ALIGN), **f32)
        self.state = torch.zeros(N.STATE_WORDS, dtype=torch.int32, device=device)
        self.set_lr(lr)
        self.set_seed(seed)
        # packed MFMA operands of every 3x3 conv in two flat blocks + a device table for the one-launch re-layout
        self.packed = {}
        entries, off = [], 0
        # Conv2DTranspose(3, strides=2, 'same') layers (USE_UPSAMPLE=False) are kept on the device as the EQUIVALENT forward
        # conv over the zero-stuffed input: Weq[t][ci][co] = W_hwoi[2-t][co][ci] (taps reversed, last two axes swapped)
        self.transposed = {st.conv for st in plan.stages if st.transpose}
        self.taps = 27 if plan.ndims == 3 else 9          # Conv3D(3x3x3): taps (kd, kh, kw) row-major
        # UpSampling2D -> conv layers run in their sub-pixel form (four 2x2-tap phase convolutions on the low-resolution
        # input, 2.25x fewer multiply-adds in the forward pass): they also get phase kernels [4][4][Cout][Cin] (mode 1)
        sub_on = plan.ndims == 2 and os.environ.get('RVIP_SUBPIX', '1') != '0'
        for st in plan.stages:
            if st.src0 != 'input_1':
                k = self.taps * st.cin * st.cout
                entries.append((st.conv, self.off[(st.conv, 'kernel')][0], off, st.cin, st.cout, 0))
                off += -(-k // ALIGN) * ALIGN
                if sub_on and st.up0 == 1 and not st.src1:
                    entries.append((st.conv, self.off[(st.conv, 'kernel')][0], off, st.cin, st.cout, 1))
                    off += -(-16 * st.cin * st.cout // ALIGN) * ALIGN
        self.wf_all = torch.empty(max(off, ALIGN), dtype=self.tdtype, device=device)
        self.wd_all = torch.empty(max(off, ALIGN), dtype=self.tdtype, device=device)
        tab = (N.PackEntry * max(len(entries), 1))()
        self.pack_max = 1
        self.subpix, self.subpix_d = {}, {}
        for i, (name, w_off, p_off, cin, cout, mode) in enumerate(entries):
            tab[i].w_off, tab[i].f_off, tab[i].d_off, tab[i].cin, tab[i].cout = w_off, p_off, p_off, cin, cout
            tab[i].taps, tab[i].mode = self.taps, mode
            k = (16 if mode else self.taps) * cin * cout
            if mode:
                self.subpix[name] = self.wf_all[p_off:p_off + k]
                self.subpix_d[name] = self.wd_all[p_off:p_off + k]      # the phase kernels of the layer's data gradient (subpix = 2)
            else:
                self.packed[name] = (self.wf_all[p_off:p_off + k], self.wd_all[p_off:p_off + k])
            self.pack_max = max(self.pack_max, k)
        self.pack_entries = len(entries)
        if entries:                                # the launch only sees the device copy: refuse a bad table here
            N.check(N.lib().rvip_pack_table_check(C.cast(tab, C.c_void_p), len(entries), self.dt), 'rvip_pack_table_check')
        self.pack_table = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(device)
        self.upload(host_weights)

    # -- host <-> device ------------------------------------------------------------------------------
    def upload(self, host_weights):
        torch = _torch()
        th = np.zeros(self.count, np.float32)
        mv = np.zeros(self.moving.numel(), np.float32)
        for (lname, wname, shape, trainable, _), arr in zip(self.plan.weight_specs(), host_weights):
            a = np.asarray(arr, np.float32)
            if a.size != int(np.prod(shape)):
                raise ValueError('weight %s/%s: expected shape %s' % (lname, wname, (shape,)))
            if wname == 'kernel' and lname in self.transposed:
                a = self._to_equivalent(a.reshape(shape))
            a = np.ascontiguousarray(a).reshape(-1)
            if trainable:
                o = self.off[(lname, wname)][0]
                th[o:o + a.size] = a
            else:
                o = self.moff[(lname, wname)][0]
                mv[o:o + a.size] = a
        self.theta.copy_(torch.from_numpy(th))
        self.moving.copy_(torch.from_numpy(mv))
        self.repack(torch.cuda.current_stream().cuda_stream)

    def download(self):
        th = self.theta.detach().cpu().numpy()
        mv = self.moving.detach().cpu().numpy()
        out = []
        for (lname, wname, shape, trainable, _) in self.plan.weight_specs():
            size = int(np.prod(shape))
            src, o = (th, self.off[(lname, wname)][0]) if trainable else (mv, self.moff[(lname, wname)][0])
            out.append(self._from_device(lname, wname, src[o:o + size], shape))
        return out

    @staticmethod
    def _to_equivalent(w):
        """[k.., Cout, Cin] of a transposed conv -> [k.., Cin, Cout] of the forward conv over the zero-stuffed input: every spatial
        axis reversed (also the stride-1 depth axis of Conv3DTranspose), last two axes swapped."""
        nd = w.ndim - 2
        flip = (slice(None, None, -1),) * nd
        return np.ascontiguousarray(np.swapaxes(w[flip], -1, -2))

    def _from_device(self, lname, wname, flat, shape):
        if wname == 'kernel' and lname in self.transposed:
            k, (co, ci) = tuple(shape[:-2]), shape[-2:]
            flip = (slice(None, None, -1),) * len(k)
            return np.ascontiguousarray(np.swapaxes(flat.reshape(k + (ci, co)), -1, -2)[flip])
        return flat.reshape(shape).copy()

    def grads_host(self):
        g = self.grad.detach().cpu().numpy() * np.float32(self.grad_unscale)
        return OrderedDict(((ln, wn), self._from_device(ln, wn, g[o:o + int(np.prod(s))], s)) for (ln, wn), (o, s) in self.off.items())

    def p(self, lname, wname):
        return _ptr(self.theta, self.off[(lname, wname)][0])

    def g(self, lname, wname):
        return _ptr(self.grad, self.off[(lname, wname)][0])

    def mv(self, lname, wname):
        return _ptr(self.moving, self.moff[(lname, wname)][0])

    def set_lr(self, lr):
        self.state.view(_torch().float32)[N.STATE_LR] = float(lr)

    def set_seed(self, seed):
        self.state[N.STATE_SEED] = int(seed) & 0x7fffffff

    def set_step(self, step):
        self.state[N.STATE_STEP] = int(step)

    def step_count(self):
        return int(self.state[N.STATE_STEP].item())

    def pack_call(self):
        """(fn, args) of the one-launch re-layout of every 3x3 kernel (fp32 HWIO master -> packed MFMA operands)."""
        L = N.lib()
        return (L.rvip_pack_all_conv3x3_weights, (_ptr(self.theta), _ptr(self.pack_table), self.pack_entries, self.pack_max,
                                                  self.dt, _ptr(self.wf_all), _ptr(self.wd_all)))

    def repack(self, stream):
        if self.pack_entries:
            fn, args = self.pack_call()
            N.check(fn(*args, C.c_void_p(stream)), 'rvip_pack_all_conv3x3_weights')


class InputRing:
    """Host -> device input pipeline of fit(): a ring of pinned slots + a copy stream, so that the H2D of batch k+1 runs under step k.

    Threading contract (keras_model._Stager): the generator thread only waits on host-side flags and writes pinned memory through
    NumPy views (next_slot, stage_host_batch); every HIP call (allocation, copies, event record / query / synchronise: the _ring_*
    hooks, feed, reset_input_ring) belongs to the training thread.  Slot protocol: item i uses slot i % slots; feeding item k hands
    back every slot up to item k - 2 (their uploads were queued two steps ago) and newer ones whose upload happens to be complete.
    With a queue of `depth` items between the threads the generator thread can be at most depth + 2 items ahead of the last feed, so
    slots >= depth + 4 never deadlocks (tests/test_fit_pipeline_cpu.py drives exactly that worst case)."""

    pin_x = None

    # -- hooks (HIP side; overridden by the host-only twin in the tests) ---------------------------------------------
    def _ring_alloc(self, slots):
        torch = _torch()
        self.pin_x = [torch.empty(self.x_stage.shape, dtype=torch.float32).pin_memory() for _ in range(slots)]
        with_y = getattr(self, 'with_y', True)              # (EvalRing of predict(): inputs only -- no pinned / device target buffers)
        self.pin_y = [torch.empty(self.y_true.shape, dtype=torch.float32).pin_memory() if with_y else None for _ in range(slots)]
        self.pin_x_np = [t.numpy() for t in self.pin_x]    # host views, created here so that the generator thread never enters torch
        self.pin_y_np = [t.numpy() if t is not None else None for t in self.pin_y]
        self.dev_in = [(torch.empty_like(self.x_stage), torch.empty_like(self.y_true) if with_y else None) for _ in range(2)]
        self.copy_stream = torch.cuda.Stream(device=self.P.device)
        self.ev_free = [None, None]                        # device staging pair -> event after which it may be refilled

    def _ring_upload(self, slot, d):
        """pinned slot -> device staging pair d on the copy stream, then device-to-device into the buffers the captured step reads, on
        the compute stream (only that last copy is ordered with the step).  Returns the event recorded after the H2D copy."""
        torch = _torch()
        cs, main = self.copy_stream, torch.cuda.current_stream()
        # The HOST waits until the copies that last read this staging pair have run (two feeds ago: the training thread stays at most
        # two steps ahead of the device, which it was anyway through _release_slots).  A device-side wait here -- the copy stream
        # waiting for an event of the compute stream that is still pending when it is queued -- cost 0.10 ms of idle device time at
        # every step boundary (tools/probe_fit_gap.py: 4.84 -> 4.74 ms per fit step against 4.70 for bare replays).
        if self.ev_free[d] is not None:
            self.ev_free[d].synchronize()
        with torch.cuda.stream(cs):
            has_y = self.slot_has_y[slot]
            self.dev_in[d][0].copy_(self.pin_x[slot], non_blocking=True)
            if has_y:
                self.dev_in[d][1].copy_(self.pin_y[slot], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(cs)
        main.wait_event(ev)
        self.x_stage.copy_(self.dev_in[d][0], non_blocking=True)
        if has_y:
            self.y_true.copy_(self.dev_in[d][1], non_blocking=True)
        evf = torch.cuda.Event()
        evf.record(main)
        self.ev_free[d] = evf
        return ev

    # -- protocol ----------------------------------------------------------------------------------------------------
    def alloc_input_ring(self, slots):
        if self.pin_x is not None and len(self.pin_x) >= slots:
            return
        self._ring_alloc(slots)
        self.slot_free = [threading.Event() for _ in range(slots)]     # set: the generator thread may overwrite the pinned slot
        self.slot_has_y = [True] * slots
        self._uploads = deque()
        self.reset_input_ring()

    def reset_input_ring(self):
        """(training thread, no generator thread running) waits for the uploads in flight and marks every slot free."""
        if self.pin_x is None:
            return
        for _, ev in self._uploads:
            ev.synchronize()
        self._uploads = deque()                            # (slot, event recorded after its H2D copy), in feed order
        for f in self.slot_free:
            f.set()
        self._fed = 0
        self._staged = 0

    def ring_slots(self):
        return len(self.pin_x) if self.pin_x is not None else 0

    def next_slot(self):
        """(generator thread) pinned slots are used round-robin"""
        slot = self._staged % len(self.pin_x)
        self._staged += 1
        return slot

    def stage_host_batch(self, slot, x, y, cancel=None):
        """(generator thread; host memory only) pageable NumPy batch -> pinned slot, once the training thread has handed the slot
        back (its previous upload has completed).  Returns False if `cancel` was set while waiting."""
        flag = self.slot_free[slot]
        while not flag.wait(0.1):
            if cancel is not None and cancel.is_set():
                return False
        flag.clear()
        np.copyto(self.pin_x_np[slot], np.asarray(x, np.float32).reshape(self.pin_x_np[slot].shape))
        if y is not None:
            np.copyto(self.pin_y_np[slot], np.asarray(y, np.float32).reshape(self.pin_y_np[slot].shape))
        self.slot_has_y[slot] = y is not None              # (predict: inputs only)
        return True

    def _release_slots(self, keep):
        up = self._uploads
        while up and (len(up) > keep or up[0][1].query()):
            slot, ev = up.popleft()
            ev.synchronize()
            self.slot_free[slot].set()

    def feed(self, slot):
        """(training thread) queue the upload of a staged slot; hand older slots back to the generator thread."""
        d = self._fed & 1
        self._fed += 1
        self._uploads.append((slot, self._ring_upload(slot, d)))
        self._release_slots(keep=2)


class EvalRing(InputRing):
    """A second input ring in front of the SAME engine's staging buffers, for Model.evaluate(): validation runs on the training thread
    between two epochs of fit(), while fit()'s stager thread may already hold batches of the next epoch in the engine's own ring.
    Used from one thread (stage, feed, forward, next batch): four slots are enough -- feeding batch k hands back slot k - 2."""

    SLOTS = 4

    def __init__(self, eng, with_y=True):
        self.x_stage, self.y_true, self.P = eng.x_stage, eng.y_true, eng.P
        self.with_y = with_y
        self.alloc_input_ring(self.SLOTS)
        self.out = None

    def ensure_targets(self):
        """(training thread, nothing staged) a ring made for predict() gets its target buffers when evaluate() comes to use it"""
        if not self.with_y:
            self.reset_input_ring()
            self.with_y = True
            self.pin_x = None
            self.alloc_input_ring(self.SLOTS)

    def drop(self):
        """(training thread) waits for what is in flight and lets the pinned / device buffers go (rings of batch sizes a
        model no longer predicts with kept several hundred MB pinned)"""
        self.reset_input_ring()
        if self.out is not None:
            for ev in self.out['done']:
                if ev is not None:
                    ev.synchronize()
        self.pin_x = self.pin_y = self.pin_x_np = self.pin_y_np = self.dev_in = self.out = None

    def download(self, src):
        """(predict) queue the device-to-host copy of `src` (the engine's heat-maps of the batch just launched) and return a handle
        for `fetch`.  `src` is first copied into one of two device staging tensors on the compute stream -- the next forward pass may
        overwrite it at once -- and from there into pinned memory on a stream of its own, next to the input stream."""
        torch = _torch()
        if self.out is None:
            self.out = dict(dev=[torch.empty_like(src) for _ in range(2)],
                            pin=[torch.empty(src.shape, dtype=src.dtype).pin_memory() for _ in range(2)],
                            done=[None, None], k=0, stream=torch.cuda.Stream(device=self.P.device))
            self.out['np'] = [t.numpy() for t in self.out['pin']]
        o = self.out
        d = o['k'] & 1
        o['k'] += 1
        if o['done'][d] is not None:
            o['done'][d].synchronize()                     # HOST wait (see _ring_upload): the pair's previous download has left it
        main = torch.cuda.current_stream()
        o['dev'][d].copy_(src, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(main)
        o['stream'].wait_event(ev)
        with torch.cuda.stream(o['stream']):
            o['pin'][d].copy_(o['dev'][d], non_blocking=True)
            done = torch.cuda.Event()
            done.record(o['stream'])
        o['done'][d] = done
        return d, done

    def fetch(self, handle, dst):
        """wait for a download and copy it out of the pinned buffer into `dst` (a NumPy view of the caller's result)"""
        d, done = handle
        done.synchronize()
        np.copyto(dst, self.out['np'][d].reshape(dst.shape))


class Engine(InputRing):
    """Activation/gradient buffers and pre-built launch lists for ONE (batch size, loss) configuration."""

    def __init__(self, params, batch, loss_kind='mse', w_bce=0.5, w_dice=1.0, world=1, masks=None, sum_reduction=False):
        torch = require_gpu()
        L = N.lib()
        self.P, self.plan, self.batch = params, params.plan, int(batch)
        plan, P = self.plan, params
        # 3-D graphs (Conv3D on [B,T,H,W,C] volumes, cfg 5): every tensor is held as B*T images [B*T,H,W,C]; the depth axis
        # only enters the 3x3x3 convolutions (depth taps read the neighbouring images of the same volume), the pooling and
        # up-sampling of the reference's 3-D template act in-plane (M_POOL (1,2,2)).
        self.depth = int(plan.dim[0]) if plan.ndims == 3 else 1
        self.kd = 3 if plan.ndims == 3 else 1
        if plan.ndims == 3:
            if tuple(plan.f_size) != (3, 3, 3) or tuple(plan.m_pool)[0] != 1 or tuple(plan.m_pool)[1:] != (2, 2):
                raise NotImplementedError('3-D graphs: F_SIZE (3,3,3) and M_POOL (1,2,2) (the reference template) are built')
        elif tuple(plan.f_size) != (3, 3) or tuple(plan.m_pool) != (2, 2):
            raise NotImplementedError('2-D graphs: F_SIZE (3,3) and M_POOL (2,2) are built')
        self.n = n = self.batch * self.depth
        ve = 4 if P.dt == N.F32 else 8
        for st in plan.stages:
            if st.cout % ve or (st.src0 != 'input_1' and (st.c0 % ve or st.c1 % ve)):
                raise ValueError('channel counts must be multiples of %d for %s activations (layer %s: %d -> %d)'
                                 % (ve, '16-bit' if ve == 8 else 'f32', st.conv, st.cin, st.cout))
        self.cimg = ci = int(plan.img_channels)
        if ci != 1:
            # IMG_CHANNELS = 2..4 (Unets.py:77; every config of the reference uses 1): rvip_conv3x3_cn_fwd / _wgrad, separate statistics pass
            f0 = plan.stages[0].cout
            cg0 = f0 // ve
            if not (2 <= ci <= 4) or self.kd != 1 or cg0 & (cg0 - 1) or cg0 > 256 or 9 * ci * f0 * 4 > 48 * 1024:
                raise NotImplementedError('IMG_CHANNELS = %d: built for 2..4 channels on 2-D graphs with FILTERS / %d a power of two' % (ci, ve))
        dev, T = P.device, P.tdtype
        self.world = world
        # the data-parallel schedule (segments with the gradient all-reduce between them) -- also for a process group of ONE rank
        # when RVIP_FORCE_DP_SCHEDULE=1: the 1-GPU lease's way to run communicator, stream ordering and segment capture against RCCL
        self.dp = world > 1 or (os.environ.get('RVIP_FORCE_DP_SCHEDULE') == '1' and _dist_ready())
        self.loss_kind = N.LOSS_MSE if loss_kind == 'mse' else N.LOSS_BCE_DICE
        self.w_bce, self.w_dice = float(w_bce), float(w_dice)
        self.sum_reduction = bool(sum_reduction)     # BceDiceLoss class form: objective = SUM over the replica's B*H*W elements
        self.masks = masks or {}                  # dropout layer name -> uint8 device tensor (parity runs)
        H, W = plan.dim[-2:]
        K = plan.mask_classes
        self.out_shape = (self.batch,) + tuple(plan.dim) + (K,)
        self.act, self.grd, self.gskip = {}, {}, {}
        shape = {}

        def alloc(name, h, w, c, store):
            if name not in store:
                store[name] = torch.empty((n, h, w, c), dtype=T, device=dev)
                shape[name] = (h, w, c)
            return store[name]

        alloc('input_1', H, W, self.cimg, self.act)
        skip_names = {st.src1 for st in plan.stages if st.src1}
        for st in plan.stages:
            alloc(st.z, st.h, st.w, st.cout, self.act)
            st_needs_apply = bool(st.bn or st.act_post or st.drop or st.pool)
            if st_needs_apply:
                alloc(st.y, st.h, st.w, st.cout, self.act)
            else:
                self.act[st.y] = self.act[st.z]
                shape[st.y] = shape[st.z]
            if st.pool:
                alloc(st.pooled, st.h // 2, st.w // 2, st.cout, self.act)
        # 2-bit window argmax of the pooled stages (written by the pooled rvip_bn_apply, read by the stage's BN-backward passes
        # in place of a materialised MaxPooling gradient); stages the column-split kernel does not cover keep rvip_maxpool2x2_bwd
        self.argmax = {}
        if os.environ.get('RVIP_FUSE_POOLBWD', '1') != '0':
            for st in plan.stages:
                if st.pool and L.rvip_bn_apply_argmax_ok(st.cout, P.dt):
                    self.argmax[st.conv] = torch.zeros(n * (st.h // 2) * (st.w // 2) * (st.cout // ve), dtype=torch.int16, device=dev)
        # gradients: d(y) per stage, d(pooled), d(z) (= grad of the conv output), skip-branch grads
        self.dz = {}
        for st in plan.stages:
            if st.conv not in self.argmax:            # a fused pooled stage never materialises the gradient of its output
                alloc(st.y, st.h, st.w, st.cout, self.grd)
            alloc(st.z, st.h, st.w, st.cout, self.dz)
            if st.pool:
                alloc(st.pooled, st.h // 2, st.w // 2, st.cout, self.grd)
            if st.y in skip_names:
                alloc(st.y, st.h, st.w, st.cout, self.gskip)
        self.up_tmp = {}                   # full-resolution data gradient of an up-conv, when it is not summed in the epilogue
        for st in plan.stages:
            if st.up0 == 2 or (st.up0 == 1 and os.environ.get('RVIP_FUSE_DOWN2', '1') == '0'):
                self.up_tmp[st.conv] = torch.empty((n, st.h, st.w, st.c0), dtype=T, device=dev)

        f32 = dict(dtype=torch.float32, device=dev)
        self.y_true = torch.zeros((n, H, W, K), **f32)
        self.pred = torch.empty((n, H, W, K), **f32)
        self.dlogit = torch.empty((n, H, W, K), **f32)
        self.sums = torch.zeros(16, **f32)
        self.loss = torch.zeros(1, **f32)
        self.x_stage = torch.empty((n, H, W, self.cimg), **f32)
        self.lm_idx = torch.zeros((n, K), dtype=torch.int64, device=dev)
        # per-BN scratch: mean, invstd, scale, shift, coef[3]  -> 7*C floats per stage
        cmax_tot = sum(7 * (-(-st.cout // ALIGN) * ALIGN) for st in plan.stages)
        self.bn_scratch = torch.zeros(max(cmax_tot, ALIGN), **f32)
        self.bn_off = {}
        o = 0
        for st in plan.stages:
            ca = -(-st.cout // ALIGN) * ALIGN
            self.bn_off[st.conv] = (o, ca)
            o += 7 * ca
        # workspaces: max over every kernel's need.  The weight-gradient kernels may run on a side stream during the
        # backward pass (see backward()), so they get a workspace of their own.
        need, need_wg = 0, 0
        for st in plan.stages:
            rows = n * st.h * st.w
            need = max(need, L.rvip_reduce_workspace(rows, 16 * st.cout))
            if st.src0 != 'input_1':
                need_wg = max(need_wg, L.rvip_conv3x3_wgrad_workspace(n, st.h, st.w, st.cin, st.cout))
            else:
                need_wg = max(need_wg, L.rvip_reduce_workspace(rows, 16 * st.cout))
        need = max(need, L.rvip_reduce_workspace(n * H * W, 8 * plan.head['cin']))
        self.ws = torch.empty(need // 4 + 64, **f32)
        self.ws_bytes = need
        self.ws_wg = torch.empty(need_wg // 4 + 64, **f32)
        self.ws_wg_bytes = need_wg
        self._graphs, self._eager_steps, self.launch_mode = None, 0, 'eager'
        self._build_lists()

    # -- helpers --------------------------------------------------------------------------------------
    def _bn(self, st, which):
        o, ca = self.bn_off[st.conv]
        idx = {'mean': 0, 'invstd': 1, 'scale': 2, 'shift': 3, 'coef': 4}[which]
        return _ptr(self.bn_scratch, o + idx * ca)

    def _build_lists(self):
        torch = _torch()
        L, P, plan, n = N.lib(), self.P, self.plan, self.n
        dt = P.dt
        A = N.ACT
        ws, wsb = _ptr(self.ws), C.c_size_t(self.ws_bytes)
        state = _ptr(P.state)
        self._keep = []                 # keep ctypes structs alive

        class _Labelled(list):
            """launch list whose entries carry a (stage, bytes moved) label for bench.py --detail"""
            label = ''

            def append(self, item):
                list.append(self, (item[0], item[1], self.label))
        fwd_t, fwd_i, bwd = _Labelled(), _Labelled(), _Labelled()
        esz = 4 if dt == N.F32 else 2
        unbiased = 1 if self.kd == 1 else 0     # TF 2.3: fused 4-D BN feeds the unbiased variance to the moving average, 5-D does not

        # The last conv stage's BN output feeds only the 1x1 head: in training it is never materialised (rvip_bn_apply_head,
        # rvip_bn_bwd_*_head rebuild it / its gradient in registers).  Inference keeps the separate launches.
        last = plan.stages[-1]
        ve_ = 4 if dt == N.F32 else 8
        cg_ = last.cout // ve_
        self.fuse_head = bool(last.bn and not last.pool and not (last.drop and last.drop[1] > 0) and last.y == plan.head['src']
                              and cg_ <= 64 and (cg_ & (cg_ - 1)) == 0 and os.environ.get('RVIP_FUSE_HEAD', '1') != '0')
        last_apply = None
        # MSE head: the logit gradient and the sums of the head's / the stage's BN backward leave the forward pass
        # (rvip_bn_apply_head_mse + rvip_head_mse_coef replace rvip_head_grad, rvip_bn_bwd_reduce_head and their finalisers)
        # BCE-Dice (round 4): the same with three row sets and a logit gradient rebuilt per pixel in the backward apply pass
        # (rvip_bn_apply_head_bcedice + rvip_head_mse_coef(loss_kind BCE_DICE) + rvip_bn_bwd_apply_head_lazy); 4-class heads keep the classic launches
        self.head_alg = bool(self.fuse_head and plan.head['k'] <= 2 and not last.act_post
                             and N.ACT[last.act_conv] == N.ACT['relu'] and os.environ.get('RVIP_BNBWD_ALGEBRAIC', '1') != '0'
                             and os.environ.get('RVIP_HEAD_ALGEBRAIC', '1') != '0')
        if self.loss_kind == N.LOSS_BCE_DICE and dt == N.F32:
            self.head_alg = False                   # (rvip_bn_apply_head_bcedice: 16-bit types; the fp32 parity path keeps the classic launches)
        self.head_bcedice = self.head_alg and self.loss_kind == N.LOSS_BCE_DICE
        # who reads what: tensor name -> producing stage, stage -> [(consumer stage, 0 = as src0 / 1 = as the skip half)]
        producer = {}
        for st in plan.stages:
            producer[st.y] = st
            if st.pool:
                producer[st.pooled] = st
        consumers = {st.conv: [] for st in plan.stages}
        for st in plan.stages:
            for which, src in ((0, st.src0), (1, st.src1)):
                if src and src in producer:
                    consumers[producer[src].conv].append((st, which))
        alg_on = os.environ.get('RVIP_BNBWD_ALGEBRAIC', '1') != '0'
        upact_on = alg_on and os.environ.get('RVIP_FUSE_UPACT', '1') != '0'
        self.sign_bits, self.keep_bits = {}, {}
        apply_train = {}                        # stage -> its training-mode rvip_apply_desc (the backward plan may attach keep_bits)
        for st in plan.stages:
            rows = n * st.h * st.w
            z, y = self.act[st.z], self.act[st.y]
            first = st.src0 == 'input_1'
            fwd_t.label = fwd_i.label = '%s %dx%dx%d->%d tensor=%.1fMB' % (st.conv, st.h, st.w, st.cin, st.cout, rows * st.cout * esz / 1e6)
            bias = P.p(st.conv, 'bias')
            act_conv = A[st.act_conv]
            # ---- conv ----
            if first and self.kd == 3:
                call = (L.rvip_conv3d_c1_fwd, (_ptr(self.act['input_1']), P.p(st.conv, 'kernel'), bias, _ptr(z),
                                               n, self.depth, st.h, st.w, st.cout, act_conv, dt))
            elif first and self.cimg > 1:
                call = (L.rvip_conv3x3_cn_fwd, (_ptr(self.act['input_1']), P.p(st.conv, 'kernel'), bias, _ptr(z),
                                                n, st.h, st.w, self.cimg, st.cout, act_conv, dt))
            elif first:
                call = (L.rvip_conv3x3_c1_fwd, (_ptr(self.act['input_1']), P.p(st.conv, 'kernel'), bias, _ptr(z),
                                                n, st.h, st.w, st.cout, act_conv, dt))
            else:
                d = N.Conv3x3Desc()
                d.x0, d.c0, d.up0 = self.act[st.src0].data_ptr(), st.c0, st.up0
                d.x1, d.c1 = (self.act[st.src1].data_ptr(), st.c1) if st.src1 else (None, 0)
                d.w_packed = P.packed[st.conv][0].data_ptr()
                d.bias = bias.value
                d.y, d.y1, d.csplit = z.data_ptr(), None, 0
                d.n, d.h, d.w, d.cout, d.act, d.dtype = n, st.h, st.w, st.cout, act_conv, dt
                d.depth, d.kd = self.depth, self.kd
                d.stream_in = 1 if os.environ.get('RVIP_NT_FWD', '0') == '1' else 0
                if st.conv in P.subpix:
                    d.w_packed, d.subpix = P.subpix[st.conv].data_ptr(), 1
                self._keep.append(d)
                call = (L.rvip_conv3x3_fwd, (C.byref(d),))
                # A stage without BatchNormalization (the up-conv: conv -> ReLU, KerasLayers.py:758) whose only reader is a conv: its
                # ReLU backward can ride in that reader's data-gradient epilogue if the forward launch leaves the sign bits of what it
                # stored (1/16 of the tensor).  The training launch gets them; whether they are used is decided with the backward plan.
                if (upact_on and not st.bn and st.act_conv == 'relu' and not st.act_post and not st.pool and not (st.drop and st.drop[1] > 0)
                        and st.cout % 8 == 0 and len(consumers[st.conv]) == 1 and consumers[st.conv][0][1] == 0 and consumers[st.conv][0][0].src1):
                    dtr = N.Conv3x3Desc.from_buffer_copy(d)
                    sb = torch.zeros(-(-st.cout // 32) * rows, dtype=torch.int32, device=self.ws.device)
                    dtr.sign_bits = sb.data_ptr()
                    if L.rvip_conv3x3_sign_bits_ok(C.byref(dtr)):
                        self._keep.append(dtr)
                        self.sign_bits[st.conv] = sb
                        call_train = (L.rvip_conv3x3_fwd, (C.byref(dtr),))
            # ---- BN statistics / coefficients ----
            fused_rows = 0
            fuse_stats = st.bn and os.environ.get('RVIP_FUSE_STATS', '1') != '0'
            if fuse_stats and not first:
                fused_rows = L.rvip_conv3x3_fwd_stats_rows(C.byref(d))     # > 0: the LDS-DMA igemm folds them in its epilogue
            elif fuse_stats and self.kd == 1 and self.cimg == 1:
                fused_rows = L.rvip_conv3x3_c1_fwd_stats_rows(n, st.h, st.w, st.cout, dt)       # first layer (Cin = 1), tiled kernel
            if fused_rows > 0:
                if first:
                    fwd_t.append((L.rvip_conv3x3_c1_fwd_stats, (_ptr(self.act['input_1']), P.p(st.conv, 'kernel'), bias, _ptr(z),
                                                                n, st.h, st.w, st.cout, act_conv, dt, ws, wsb)))
                else:
                    fwd_t.append((L.rvip_conv3x3_fwd_stats, (C.byref(d), ws, wsb)))
                fwd_t.append((L.rvip_bn_stats_finalize, (
                    ws, fused_rows, C.c_longlong(rows), st.cout, P.p(st.bn, 'gamma'), P.p(st.bn, 'beta'),
                    P.mv(st.bn, 'moving_mean'), P.mv(st.bn, 'moving_variance'), C.c_float(BN_MOMENTUM), C.c_float(BN_EPS), unbiased,
                    self._bn(st, 'mean'), self._bn(st, 'invstd'), self._bn(st, 'scale'), self._bn(st, 'shift'))))
            else:
                fwd_t.append(call_train if st.conv in self.sign_bits else call)
            fwd_i.append(call)
            if st.bn:
                if fused_rows <= 0:
                    fwd_t.append((L.rvip_bn_train_stats, (
                        _ptr(z), C.c_longlong(rows), st.cout, dt, P.p(st.bn, 'gamma'), P.p(st.bn, 'beta'),
                        P.mv(st.bn, 'moving_mean'), P.mv(st.bn, 'moving_variance'), C.c_float(BN_MOMENTUM), C.c_float(BN_EPS), unbiased,
                        self._bn(st, 'mean'), self._bn(st, 'invstd'), self._bn(st, 'scale'), self._bn(st, 'shift'), ws, wsb)))
                fwd_i.append((L.rvip_bn_infer_coeffs, (
                    P.p(st.bn, 'gamma'), P.p(st.bn, 'beta'), P.mv(st.bn, 'moving_mean'), P.mv(st.bn, 'moving_variance'),
                    C.c_float(BN_EPS), st.cout, self._bn(st, 'scale'), self._bn(st, 'shift'))))
            # ---- apply (BN affine, act-after-BN, dropout, pool) ----
            # (a conv that feeds MaxPooling directly shares one tensor name for z and y: the pass then runs in
            #  place as an identity and only produces the pooled tensor)
            if st.bn or st.act_post or st.drop or st.pool:
                for training in (True, False):
                    a = N.ApplyDesc()
                    a.z, a.y = z.data_ptr(), y.data_ptr()
                    a.pooled = self.act[st.pooled].data_ptr() if st.pool else None
                    a.scale = self._bn(st, 'scale').value if st.bn else None
                    a.shift = self._bn(st, 'shift').value if st.bn else None
                    a.act = A[st.act_post]
                    a.drop_rate, a.mask, a.state, a.layer_id = 0.0, None, state.value, 0
                    if training and st.drop and st.drop[1] > 0:
                        a.drop_rate, a.layer_id = st.drop[1], st.drop[2]
                        if st.drop[0] in self.masks:
                            a.mask = self.masks[st.drop[0]].data_ptr()
                    a.n, a.h, a.w, a.c, a.dtype = n, st.h, st.w, st.cout, dt
                    if training and st.pool and st.conv in self.argmax:
                        a.argmax = self.argmax[st.conv].data_ptr()        # MaxPooling backward folds into the BN-backward passes
                    self._keep.append(a)
                    if training:
                        apply_train[st.conv] = a
                    if training and st is last and self.fuse_head:
                        last_apply = a                      # consumed by rvip_bn_apply_head below
                    elif not training and not (st.bn or st.act_post or st.drop or st.pool):
                        pass                                # inference: y is z (no pass needed)
                    else:
                        (fwd_t if training else fwd_i).append((L.rvip_bn_apply, (C.byref(a),)))

        hd = plan.head
        fwd_t.label = fwd_i.label = bwd.label = 'head'
        hrows = C.c_longlong(n * hd['h'] * hd['w'])
        hx = self.act[hd['src']]
        hw_, hb_ = P.p(hd['conv'], 'kernel'), P.p(hd['conv'], 'bias')
        head_eval = None
        if self.fuse_head:
            head_eval = (L.rvip_bn_apply_head, (C.byref(last_apply), hw_, hb_, hd['k'], _ptr(self.pred), _ptr(self.y_true),
                                                _ptr(self.sums), ws, wsb), 'head')
        if self.head_alg:
            pass                                        # the launch is appended once the loss scale is known (below)
        elif self.fuse_head:
            fwd_t.append(head_eval[:2])
        else:
            fwd_t.append((L.rvip_head_fwd, (_ptr(hx), hw_, hb_, _ptr(self.pred), _ptr(self.y_true), _ptr(self.sums), hrows,
                                            hd['cin'], hd['k'], dt, ws, wsb)))
        fwd_i.append((L.rvip_head_fwd, (_ptr(hx), hw_, hb_, _ptr(self.pred), None, None, hrows, hd['cin'], hd['k'], dt,
                                        None, C.c_size_t(0))))
        # inference-mode network + loss sums (validation)
        self.fwd_eval = list(fwd_i[:-1]) + [head_eval if head_eval is not None else fwd_t[-1]]

        # ---------------- backward ----------------
        per_rank = float(n * hd['h'] * hd['w'] * hd['k'])
        if self.loss_kind == N.LOSS_BCE_DICE:     # the background channel of a 4-class head is sliced off (Loss_and_metrics.py:240-242)
            self._inv_count = 1.0 / (float(n * hd['h'] * hd['w'] * min(hd['k'], 3)) * self.world)
        else:
            self._inv_count = 1.0 / (per_rank * self.world)
        # f16 activations: the loss gradient 2(p - y) p(1-p) / count is ~1e-7 at the benchmark shapes, below the f16 normal
        # range.  Static loss scaling (the reference has no f16 path; this is the usual mixed-precision recipe): dlogit is
        # multiplied by a power of two ~ count (so |dlogit| <= 0.5), every gradient of the step carries the factor, and the
        # optimiser removes it exactly (rvip_adam_step grad_scale).  RVIP_LOSS_SCALE overrides.
        # BceDiceLoss class form (Loss_and_metrics.py:220-226 overrides Loss.__call__): the gradient is that of the SUM over the
        # replica's B*H*W elements = the mean form's times grad_factor (oracle/rvip_oracle.py::bce_dice_loss); the loss VALUE the
        # kernel writes stays the mean.  Both factors ride on one rvip_scale_f32 pass over dlogit.
        self.grad_factor = float(n * hd['h'] * hd['w']) if self.sum_reduction else 1.0
        if P.dt == N.F16:
            env = os.environ.get('RVIP_LOSS_SCALE')
            self.loss_scale = float(env) if env else float(2 ** int(np.floor(np.log2(max(per_rank * self.world / self.grad_factor, 1.0)))))
        else:
            self.loss_scale = 1.0
        P.grad_unscale = 1.0 / self.loss_scale
        if self.head_bcedice:
            nr = L.rvip_bn_apply_head_mse_rows(hrows, last.cout, dt, hd['k'])
            self.head_rows = torch.empty(nr * 7 * last.cout, dtype=torch.float32, device=self.ws.device)
            self.head_dcoef = torch.zeros(4, dtype=torch.float32, device=self.ws.device)
            fwd_t.append((L.rvip_bn_apply_head_bcedice, (C.byref(last_apply), hw_, hb_, P.p(last.bn, 'beta'), hd['k'], _ptr(self.pred), _ptr(self.y_true),
                                                         _ptr(self.sums), _ptr(self.head_rows), C.c_size_t(self.head_rows.numel() * 4), ws, wsb)))
        elif self.head_alg:
            nr = L.rvip_bn_apply_head_mse_rows(hrows, last.cout, dt, hd['k'])
            self.head_rows = torch.empty(nr * 3 * last.cout, dtype=torch.float32, device=self.ws.device)
            fwd_t.append((L.rvip_bn_apply_head_mse, (C.byref(last_apply), hw_, hb_, P.p(last.bn, 'beta'), hd['k'], _ptr(self.pred), _ptr(self.y_true),
                                                     _ptr(self.sums), _ptr(self.dlogit), C.c_float(self._inv_count),
                                                     C.c_float(self.loss_scale * self.grad_factor), _ptr(self.head_rows),
                                                     C.c_size_t(self.head_rows.numel() * 4), ws, wsb)))
        else:
            bwd.append((L.rvip_head_grad, (_ptr(self.pred), _ptr(self.y_true), _ptr(self.sums), _ptr(self.dlogit), _ptr(self.loss),
                                           hrows, hd['k'], self.loss_kind, C.c_float(self._inv_count), C.c_float(1.0 / self.world),
                                           C.c_float(self.w_bce), C.c_float(self.w_dice))))
            if self.loss_scale * self.grad_factor != 1.0:
                bwd.append((L.rvip_scale_f32, (_ptr(self.dlogit), C.c_longlong(self.dlogit.numel()), C.c_float(self.loss_scale * self.grad_factor))))
        if not self.fuse_head:
            bwd.append((L.rvip_head_bwd, (_ptr(hx), hw_, _ptr(self.dlogit), _ptr(self.grd[hd['src']]), P.g(hd['conv'], 'kernel'),
                                          P.g(hd['conv'], 'bias'), hrows, hd['cin'], hd['k'], dt, ws, wsb)))
        # Gradient buckets for the data-parallel all-reduce: the backward pass finishes head, decoder and bottleneck
        # first; their gradients are the tail of the flat block (creation order) and can travel while the encoder's
        # backward still runs.  bwd[:bwd_split] produces grad[grad_split:], bwd[bwd_split:] produces grad[:grad_split].
        n_enc = 2 * plan.depth
        self.bwd_split = None
        self.grad_split = P.off[(plan.stages[n_enc].conv, 'kernel')][0] if len(plan.stages) > n_enc else 0
        # Deferred folds: the partial rows of the bias gradients and the split-K slabs of the weight gradients (read only by
        # the optimiser) stay in private regions and are summed by two table-driven launches per gradient bucket instead of
        # one small fold launch per layer (~40 per step).
        defer = self.kd == 1 and os.environ.get('RVIP_DEFER_FOLDS', '1') != '0'
        narrow, wide = [], []                  # (src tensor, dst pointer, nrows, width) of the current bucket
        self._fold_bufs = []

        def flush_folds():
            for entries, is_wide in ((narrow, 0), (wide, 1)):
                if not entries:
                    continue
                tab = (N.FoldEntry * len(entries))()
                for i, ent in enumerate(entries):
                    src, dst, nrows, width = ent[:4]
                    tab[i].src, tab[i].dst, tab[i].nrows, tab[i].width = src.data_ptr(), dst.value, nrows, width
                    tab[i].stride = ent[4] if len(ent) > 4 else 0
                tabd = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(self.ws.device)
                self._fold_bufs.append(tabd)
                bwd.label = 'deferred folds (%s, %d layers)' % ('weight gradients' if is_wide else 'bias gradients', len(entries))
                bwd.append((L.rvip_fold_rows_batch, (_ptr(tabd), len(entries), C.c_longlong(max(e[3] for e in entries)), is_wide)))
                del entries[:]
        # ---- weight- and data-gradient descriptors of every igemm stage (built first: the BN-backward planning below queries them) ----
        fuse_down_on = os.environ.get('RVIP_FUSE_DOWN2', '1') != '0'
        # The sub-pixel data gradient (subpix = 2) multiplies with phase kernels = sums of 2 / 4 taps rounded ONCE to the activation
        # type: its result g' differs from the nine-tap gradient g of the rounded taps Wr by a systematic 2^-9 (bf16).  The algebraic BN
        # backward of the stage in FRONT of such a layer solves  sum g*xhat = (<Wr, dW> - beta * sum g) / gamma  with <Wr, dW> = sum g*y
        # for the nine-tap g but  sum g  = the column sums of g': an inconsistency of 2^-9 |beta / gamma| |sum g| that does not
        # average down over pixels and is invisible at initialisation (beta = 0) -- ADVICE r3.  RVIP_BNBWD_SUBPIX_CONSUMER says what
        # such a stage does: 'exact' -- the classic reduction pass over the (g', z) it really uses; 'ninetap' -- the layer keeps the
        # nine-tap data gradient with the 2x2 sums in its epilogue (consistent with <Wr, dW>); 'algebraic' -- round 3's behaviour.
        # 'auto' (default) picks per layer whichever of the two consistent forms is cheaper (measured, round 4: the nine-tap data gradient
        # costs +7 .. +20 us per layer, the reduction pass of the stage in front +8 .. +25 us, in opposite order of map size): 'ninetap'
        # for the HBM-bound full-resolution layer (c0 * cout <= 64 * 32), 'exact' for the others.
        # 'phase' (round 4, where the layer's weight gradient runs in the four-phase sub-pixel form): <W, dW> is formed against the PHASE
        # kernels from the phase-resolved slabs (rvip_wgrad3x3_desc.w_phase) -- consistent with the sub-pixel data gradient by construction,
        # at no cost: the stage in front keeps the algebraic route.  'auto' = 'phase' where it exists, else the cheaper of the other two.
        sp_consumer = os.environ.get('RVIP_BNBWD_SUBPIX_CONSUMER', 'auto')
        if sp_consumer not in ('auto', 'phase', 'exact', 'ninetap', 'algebraic'):
            raise ValueError('RVIP_BNBWD_SUBPIX_CONSUMER=%r' % sp_consumer)

        def sp_mode(st):
            if sp_consumer in ('auto', 'phase'):
                if dt != N.F32 and st.conv in wg_desc and L.rvip_conv3x3_wgrad_form(C.byref(wg_desc[st.conv])) == 1:
                    return 'phase'
                if sp_consumer == 'phase':
                    return 'exact'
                return 'ninetap' if st.c0 * st.cout <= 64 * 32 else 'exact'
            return sp_consumer
        self.sp_modes = {}
        wg_desc, dg_desc = {}, {}
        # Weight and data gradient of a layer side by side (RVIP_BWD_OVERLAP = compute units of the weight gradient, 1..255; 0 = one
        # after the other): both read the same gradient tensor and neither fills the chip's fixed costs -- launch, first tile, the
        # epilogue / slab store at the end, the boundary -- with work.  Each is a one-workgroup-per-CU persistent kernel that owns its
        # CU (LDS, registers), so "side by side" is a PARTITION of the CUs: the two grids are sized to W and 256 - W of them
        # (rvip_*_desc.cu_limit) and launched on two streams between a fork and a join of the captured graph.
        # RVIP_RCCL_CU_RESERVE = n (data-parallel runs): the contraction launches of the encoder's backward pass -- the ones that run
        # while gradient bucket 0 is in flight -- leave n CUs to RCCL's kernels (default 0).
        self.side_stream = None
        self.paired = []                                  # layers whose two gradients run as one launch (rvip_conv3x3_wgrad_dgrad)
        self.bwd_mode = {}                                # conv name -> 'pair' | 'forkjoin' | 'serial'
        # RVIP_BWD_PAIR (default 1): where rvip_conv3x3_wgrad_dgrad serves the layer (16-bit types, 2-D: all 21 layers of config 2) the two gradients are the two parts of ONE grid instead of two launches between a fork and a join of the graph --
        # the cross-queue synchronisation of that schedule costs ~18 us per layer, most of what running side by side hides.
        # Same box (round 5): one after the other 4.62, fork / join 4.60, pair kernel 4.52 ms; bit-identical gradients (tools/ab_pair.py).
        self.bwd_pair = os.environ.get('RVIP_BWD_PAIR', '1') != '0'
        # Measured (round 5, same box, alternating): W = 128 4.650 -> 4.54 ms at config 2 (-2.4 %), BCE-Dice -3.0 %, config 4 +0.3 %;
        # 96 / 112 / 144 are SLOWER than one after the other (4.76 / 4.69 / 4.89: the tile counts no longer divide by the grids).
        # RVIP_BWD_OVERLAP unset ('auto'): 128 / 128, except at the two full-resolution levels, where the data gradient is the longer half
        # by 10-30 % (tools/tune_pair_split.py, the pair launch back to back, us at W = 104 / 112 / 128: 64 -> 64 at 128^2 76 / 83 / 81,
        # (64 + 64) -> 64 at 128^2 137 / 144 / 156, (32 + 32) -> 32 at 256^2 188 / 174 / 186, 32 -> 32 at 256^2 121 / 114 / 118): the
        # weight gradient gets 104 (128^2 level, >= 64 input channels, nine-tap forms) or 112 (256^2 level) compute units there.
        ove = os.environ.get('RVIP_BWD_OVERLAP', 'auto')
        self.bwd_split_auto = ove == 'auto'
        ov = 128 if self.bwd_split_auto else int(ove or 0)
        self.bwd_overlap = ov if (0 < ov < 256 and self.kd == 1) else 0
        reserve = int(os.environ.get('RVIP_RCCL_CU_RESERVE', '0') or 0) if self.dp else 0
        self.cu_reserve = reserve if 0 < reserve < 128 else 0
        if self.bwd_overlap:
            self.side_stream = torch.cuda.Stream(device=P.device)
        n_enc_ = 2 * plan.depth
        for si_, st in enumerate(plan.stages):
            if st.src0 == 'input_1':
                continue
            dz = self.dz[st.z]
            wg = N.Wgrad3x3Desc()
            avail = 256 - (self.cu_reserve if si_ < n_enc_ else 0)      # (encoder stages run while bucket 0 travels)
            cu_w = cu_d = avail if avail < 256 else 0
            if self.bwd_overlap:
                w_of_256 = self.bwd_overlap
                if self.bwd_split_auto and not st.up0:
                    if st.h >= 256:
                        w_of_256 = 112
                    elif st.h >= 128 and st.cin >= 64:
                        w_of_256 = 104
                cu_w = max(8, min(avail - 8, w_of_256 * avail // 256))
                cu_d = avail - cu_w
            wg.x0, wg.c0, wg.up0 = self.act[st.src0].data_ptr(), st.c0, st.up0
            wg.x1, wg.c1 = (self.act[st.src1].data_ptr(), st.c1) if st.src1 else (None, 0)
            wg.dy, wg.dw = dz.data_ptr(), P.g(st.conv, 'kernel').value
            wg.n, wg.h, wg.w, wg.cout, wg.dtype = n, st.h, st.w, st.cout, dt
            wg.depth, wg.kd = self.depth, self.kd
            wg.workspace, wg.workspace_bytes = self.ws_wg.data_ptr(), self.ws_wg_bytes
            wg.cu_limit = cu_w
            wg_desc[st.conv] = wg
            dg = N.Conv3x3Desc()
            dg.cu_limit = cu_d
            dg.x0, dg.c0, dg.up0, dg.x1, dg.c1 = dz.data_ptr(), st.cout, 0, None, 0
            dg.w_packed, dg.bias = P.packed[st.conv][1].data_ptr(), None
            dg.n, dg.h, dg.w, dg.cout, dg.act, dg.dtype = n, st.h, st.w, st.cin, 0, dt
            dg.depth, dg.kd = self.depth, self.kd
            dg.y1, dg.csplit = None, 0
            # stream_in (non-temporal fetch of dz, whose other reader - the weight gradient - ran already): measured slower,
            # the Cout / 64 workgroup columns of the data gradient re-read the same tile (5.72 vs 5.69 ms); opt-in
            dg.stream_in = 1 if os.environ.get('RVIP_NT_DGRAD', '0') == '1' else 0
            if st.src1:
                dg.y, dg.y1, dg.csplit = self.grd[st.src0].data_ptr(), self.gskip[st.src1].data_ptr(), st.c0
            elif st.up0 == 1 and fuse_down_on:      # UpSampling2D: the 2x2 block sums leave the data-gradient epilogue directly
                dg.y, dg.down2 = self.grd[st.src0].data_ptr(), 1
                if (st.conv in P.subpix_d and dt != N.F32 and self.kd == 1 and os.environ.get('RVIP_SUBPIX_DGRAD', '1') != '0'
                        and not (alg_on and sp_mode(st) == 'ninetap')):
                    # ... or, 16-bit types, the gradient arrives on the low-resolution grid at once: the sub-pixel form of the same launch
                    # (four source phases x 2x2 summed taps, 16 instead of 36 multiply-adds per low-resolution pixel)
                    sp = N.Conv3x3Desc.from_buffer_copy(dg)
                    sp.down2, sp.subpix, sp.w_packed = 0, 2, P.subpix_d[st.conv].data_ptr()
                    if L.rvip_conv3x3_fwd_sums_rows(C.byref(sp)) > 0:
                        dg = sp
            elif st.up0:
                dg.y = self.up_tmp[st.conv].data_ptr()
            else:
                dg.y = self.grd[st.src0].data_ptr()
            # which schedule the layer's two gradients take (decided HERE: the partial-row / slab counts queried below follow cu_limit)
            mode = 'serial'
            if self.bwd_overlap:
                # (what the pair hides is the launches' fixed cost: nothing on layers that run for hundreds of microseconds -- config 4,
                #  same box: 516.6 slices/s paired against 524.9 one after the other -- so only layers below a FLOP count take it)
                small = 2.0 * n * st.h * st.w * 9 * st.cin * st.cout <= float(os.environ.get('RVIP_BWD_PAIR_MAX_GFLOP', '100')) * 1e9
                if self.bwd_pair and small and L.rvip_conv3x3_wgrad_dgrad_ok(C.byref(wg), C.byref(dg)):
                    mode = 'pair'
                elif os.environ.get('RVIP_BWD_UNPAIRED', 'serial') == 'forkjoin':
                    mode = 'forkjoin'
                else:                           # (the sub-pixel forms of the up-conv layers: measured no faster between a fork and a join,
                    wg.cu_limit = dg.cu_limit = avail if avail < 256 else 0      #  160 against 137 us for the 64 -> 32 layer at 256^2)
            self.bwd_mode[st.conv] = mode
            self._keep += [wg, dg]
            wg_desc[st.conv], dg_desc[st.conv] = wg, dg

        # ---- BatchNormalization backward WITHOUT its reduction pass (rvip_bn_bwd_coef, include/rvip_hip.h) ----
        # For a stage  z -> BN -> [Dropout] -> y  every consumer of y is a 3x3 conv (directly, through MaxPooling2D / UpSampling2D, or
        # as the skip half of a Concatenate), so  sum g  is the column sum of the consumers' data-gradient outputs (fused into their
        # epilogues, Dropout backward included) and  sum g*y = sum W * dW  of the consumers (read off their weight gradients while the
        # split-K slabs are folded).  The 2 x tensor re-read of rvip_bn_bwd_reduce disappears; a 32-channel block with an ill-conditioned
        # gamma / beta takes the exact route inside rvip_bn_bwd_coef itself.
        def gated_probe(c, channels):
            """rows of the consumer's data gradient when its first `channels` result channels are gated by bit planes (0: not served)"""
            probe = N.Conv3x3Desc.from_buffer_copy(dg_desc[c.conv])
            probe.mask_bits, probe.mask_channels, probe.mask_scale = self.ws.data_ptr(), channels, 1.0
            return L.rvip_conv3x3_fwd_sums_rows(C.byref(probe))

        def algebraic_ok(p):
            if not alg_on or not p.bn or p.act_post or (p is last and self.fuse_head):
                return False
            cl = consumers[p.conv]
            dropping = bool(p.drop and p.drop[1] > 0)
            if not cl or len(cl) > 2 or (dropping and (p.drop[0] in self.masks or p.pool or len(cl) != 1)):
                return False
            if dropping and not (p.cout % 32 == 0 or p.cout in (8, 16)):      # keep bits: whole 32-channel blocks, or one partial block
                return False
            for c, which in cl:
                if c.conv not in dg_desc or c.up0 == 2 or (c.up0 == 1 and not fuse_down_on) or (dropping and (c.src1 or c.up0)):
                    return False
                if dg_desc[c.conv].subpix == 2:
                    self.sp_modes[c.conv] = sp_mode(c)
                    if sp_mode(c) == 'exact':
                        return False
                if L.rvip_conv3x3_fwd_sums_rows(C.byref(dg_desc[c.conv])) <= 0:
                    return False
                if dropping and (c.cin % 8 or gated_probe(c, c.cin) <= 0):      # Dropout backward: the keep bits gate the consumer's data gradient
                    return False
            return True
        self.algebraic = {p.conv for p in plan.stages if algebraic_ok(p)}
        self._alg_bufs = {}
        self.bn_flags = {}                      # stage -> device flags of rvip_bn_bwd_coef (1 = that 32-channel block took the exact route)
        sums_rows, dot_rows = {}, {}            # consumer conv name -> (tensor, nrows)
        for p in plan.stages:
            if p.conv not in self.algebraic:
                continue
            for c, which in consumers[p.conv]:
                dg, wg = dg_desc[c.conv], wg_desc[c.conv]
                if c.conv not in sums_rows:
                    nr = L.rvip_conv3x3_fwd_sums_rows(C.byref(dg))
                    sums_rows[c.conv] = (torch.zeros(nr * c.cin, dtype=torch.float32, device=self.ws.device), nr)
                    nd = L.rvip_conv3x3_wgrad_dot_rows(C.byref(wg))
                    dbuf = torch.zeros(nd * c.cin, dtype=torch.float64, device=self.ws.device)
                    dot_rows[c.conv] = (dbuf, nd)
                    wg.w_master, wg.dot_rows, wg.dot_rows_bytes = P.p(c.conv, 'kernel').value, dbuf.data_ptr(), dbuf.numel() * 8
                    if dg.subpix == 2 and self.sp_modes.get(c.conv) == 'phase':
                        wg.w_phase = P.subpix_d[c.conv].data_ptr()
                if p.drop and p.drop[1] > 0:          # Dropout backward rides in the consumer's data-gradient epilogue: the forward
                    kb = torch.zeros(-(-p.cout // 32) * n * p.h * p.w, dtype=torch.int32, device=self.ws.device)      # pass leaves its keep bits
                    self.keep_bits[p.conv] = kb
                    apply_train[p.conv].keep_bits = kb.data_ptr()
                    dg.mask_bits, dg.mask_channels, dg.mask_scale = kb.data_ptr(), p.cout, 1.0 / (1.0 - p.drop[1])
        # ReLU backward of the BN-less stages whose forward launch left sign bits: gate the x0 half of the reader's data gradient, which
        # then IS the stage's conv-output gradient (no rvip_bn_bwd_apply pass for the stage; its bias gradient = the column sums)
        self.upact = {}
        for u in plan.stages:
            if u.conv not in self.sign_bits:
                continue
            c = consumers[u.conv][0][0]
            if c.conv not in dg_desc or c.up0 or not c.src1 or c.c0 != u.cout or gated_probe(c, c.c0) <= 0:
                continue
            dg = dg_desc[c.conv]
            if c.conv not in sums_rows:
                nr = L.rvip_conv3x3_fwd_sums_rows(C.byref(dg))
                sums_rows[c.conv] = (torch.zeros(nr * c.cin, dtype=torch.float32, device=self.ws.device), nr)
            dg.mask_bits, dg.mask_channels, dg.mask_scale = self.sign_bits[u.conv].data_ptr(), c.c0, 1.0
            dg.y = self.dz[u.z].data_ptr()
            self.upact[u.conv] = c
        for c in plan.stages:                   # a split result whose first half nobody sums (a BN-less up-conv whose ReLU backward is a pass of its own)
            if c.conv in sums_rows and c.src1 and not (c.src0 in producer and (producer[c.src0].conv in self.algebraic or producer[c.src0].conv in self.upact)):
                dg_desc[c.conv].sums_from = c.c0
        self._alg_bufs['sums'], self._alg_bufs['dots'] = sums_rows, dot_rows
        min_gamma = float(os.environ.get('RVIP_BNBWD_MIN_GAMMA', 1.0 / 64))
        max_beta_ratio = float(os.environ.get('RVIP_BNBWD_MAX_BETA_RATIO', 64.0))

        for si, st in reversed(list(enumerate(plan.stages))):
            if si == n_enc - 1:
                if self.dp or os.environ.get('RVIP_FOLD_BUCKETS') == '2':
                    flush_folds()              # bucket 0 (head, decoder, bottleneck) is complete here: its all-reduce starts behind it
                # (one process: nobody reads the gradients before the optimiser -- every deferred fold waits for the final flush,
                #  two launches instead of four)
                self.bwd_split = len(bwd)
            rows = n * st.h * st.w
            first = st.src0 == 'input_1'
            bwd.label = '%s %dx%dx%d->%d tensor=%.1fMB' % (st.conv, st.h, st.w, st.cin, st.cout, rows * st.cout * esz / 1e6)
            gy, dz, z = self.grd.get(st.y), self.dz[st.z], self.act[st.z]
            fuse_pool = st.pool and st.conv in self.argmax
            if st.pool and not fuse_pool:
                add = self.gskip.get(st.y)
                bwd.append((L.rvip_maxpool2x2_bwd, (_ptr(self.act[st.y]), _ptr(self.grd[st.pooled]),
                                                    _ptr(add) if add is not None else None, _ptr(gy), n, st.h, st.w, st.cout, dt)))
            elif st.y in self.gskip and not st.pool:
                raise NotImplementedError('skip tensor without pooling')
            b = N.BnBwdDesc()
            b.dy, b.z, b.dz = (gy.data_ptr() if gy is not None else None), z.data_ptr(), dz.data_ptr()
            if fuse_pool:                      # gy is never materialised: (pooled gradient, window argmax, skip gradient) instead
                add = self.gskip.get(st.y)
                b.dy = add.data_ptr() if add is not None else None
                b.dpooled, b.argmax, b.h, b.w = self.grd[st.pooled].data_ptr(), self.argmax[st.conv].data_ptr(), st.h, st.w
            if st.bn:
                b.gamma = P.p(st.bn, 'gamma').value
                b.mean, b.invstd = self._bn(st, 'mean').value, self._bn(st, 'invstd').value
                b.scale, b.shift = self._bn(st, 'scale').value, self._bn(st, 'shift').value
                b.dgamma, b.dbeta = P.g(st.bn, 'gamma').value, P.g(st.bn, 'beta').value
                b.coef = self._bn(st, 'coef').value
            b.dbias = P.g(st.conv, 'bias').value
            b.act = N.ACT[st.act_post] if st.act_post else N.ACT[st.act_conv]
            b.act_after_bn = 1 if st.act_post else 0
            b.drop_rate, b.mask, b.state, b.layer_id = 0.0, None, state.value, 0
            alg = st.conv in self.algebraic
            if st.drop and st.drop[1] > 0 and not alg:       # (algebraic: the gradient arrives with the Dropout backward applied)
                b.drop_rate, b.layer_id = st.drop[1], st.drop[2]
                if st.drop[0] in self.masks:
                    b.mask = self.masks[st.drop[0]].data_ptr()
            b.rows, b.c, b.dtype = rows, st.cout, dt
            b.workspace, b.workspace_bytes = self.ws.data_ptr(), self.ws_bytes
            if defer:
                if st is last and self.fuse_head:
                    nr = L.rvip_bn_bwd_apply_head_rows(C.c_longlong(rows), st.cout, dt, hd['k'])
                else:
                    nr = L.rvip_bn_bwd_rows(C.c_longlong(rows), st.cout, dt)
                rbuf = torch.empty(nr * st.cout, dtype=torch.float32, device=self.ws.device)
                self._fold_bufs.append(rbuf)
                b.bias_rows, b.bias_rows_bytes = rbuf.data_ptr(), rbuf.numel() * 4
                narrow.append((rbuf, P.g(st.conv, 'bias'), nr, st.cout))
            self._keep.append(b)
            if st is last and self.fuse_head:
                b.dy = None
                if self.head_alg:
                    hc = N.HeadCoefDesc()
                    hc.bn, hc.beta = C.pointer(b), P.p(st.bn, 'beta').value
                    hc.head_w, hc.dlogit, hc.k = hw_.value, self.dlogit.data_ptr(), hd['k']
                    hc.mse_rows, hc.nrows = self.head_rows.data_ptr(), self.head_rows.numel() // ((7 if self.head_bcedice else 3) * st.cout)
                    if self.head_bcedice:
                        hc.loss_kind, hc.w_bce, hc.w_dice, hc.local_over_global = N.LOSS_BCE_DICE, self.w_bce, self.w_dice, 1.0 / self.world
                        hc.dscale = self.loss_scale * self.grad_factor
                        hc.pred, hc.y_true, hc.dcoef = self.pred.data_ptr(), self.y_true.data_ptr(), self.head_dcoef.data_ptr()
                    hc.head_dw, hc.head_db = P.g(hd['conv'], 'kernel').value, P.g(hd['conv'], 'bias').value
                    hc.sums, hc.loss_out, hc.inv_count = self.sums.data_ptr(), self.loss.data_ptr(), self._inv_count
                    flags = torch.zeros(-(-st.cout // 32), dtype=torch.int32, device=self.ws.device)
                    self._fold_bufs.append(flags)
                    self.bn_flags[st.conv] = flags
                    hc.flags, hc.min_gamma, hc.max_beta_ratio = flags.data_ptr(), min_gamma, max_beta_ratio
                    self._keep.append(hc)
                    bwd.append((L.rvip_head_mse_coef, (C.byref(hc),)))
                else:
                    bwd.append((L.rvip_bn_bwd_reduce_head, (C.byref(b), hw_, _ptr(self.dlogit), hd['k'], P.g(hd['conv'], 'kernel'),
                                                            P.g(hd['conv'], 'bias'))))
                if self.head_bcedice:
                    bwd.append((L.rvip_bn_bwd_apply_head_lazy, (C.byref(b), hw_, _ptr(self.pred), _ptr(self.y_true), _ptr(self.head_dcoef), hd['k'])))
                else:
                    bwd.append((L.rvip_bn_bwd_apply_head, (C.byref(b), hw_, _ptr(self.dlogit), hd['k'])))
            elif st.conv in self.upact:              # dz was written, ReLU backward applied, by the reader's data gradient; bias gradient = its column sums
                c = self.upact[st.conv]
                sbuf_, nr = sums_rows[c.conv]
                if defer:
                    narrow[-1] = (sbuf_, P.g(st.conv, 'bias'), nr, st.cout, c.cin)
                else:
                    tab1 = (N.FoldEntry * 1)()
                    tab1[0].src, tab1[0].dst, tab1[0].nrows, tab1[0].stride, tab1[0].width = sbuf_.data_ptr(), P.g(st.conv, 'bias').value, nr, c.cin, st.cout
                    tabd1 = torch.frombuffer(bytearray(bytes(tab1)), dtype=torch.uint8).to(self.ws.device)
                    self._fold_bufs.append(tabd1)
                    bwd.append((L.rvip_fold_rows_batch, (_ptr(tabd1), 1, C.c_longlong(st.cout), 0)))
            else:
                if alg:
                    cd = N.BnCoefDesc()
                    for q, (c, which) in enumerate(consumers[st.conv]):
                        off = 0 if which == 0 else c.c0
                        sbuf_, nr = sums_rows[c.conv]
                        dbuf_, nd = dot_rows[c.conv]
                        cd.t1[q].rows, cd.t1[q].nrows, cd.t1[q].stride, cd.t1[q].offset = sbuf_.data_ptr(), nr, c.cin, off
                        cd.t2[q].rows, cd.t2[q].nrows, cd.t2[q].stride, cd.t2[q].offset = dbuf_.data_ptr(), nd, c.cin, off
                    cd.gamma, cd.beta = P.p(st.bn, 'gamma').value, P.p(st.bn, 'beta').value
                    cd.mean, cd.invstd = b.mean, b.invstd
                    cd.dgamma, cd.dbeta, cd.coef = b.dgamma, b.dbeta, b.coef
                    nflags = -(-st.cout // 32)
                    flags = torch.zeros(nflags, dtype=torch.int32, device=self.ws.device)
                    self._fold_bufs.append(flags)
                    cd.flags, cd.count, cd.c = flags.data_ptr(), rows, st.cout
                    cd.min_gamma, cd.max_beta_ratio = min_gamma, max_beta_ratio
                    cd.fallback = C.pointer(b)                   # the exact route of an ill-conditioned block reads what the apply pass reads
                    self._keep.append(cd)
                    self.bn_flags[st.conv] = flags
                    bwd.append((L.rvip_bn_bwd_coef, (C.byref(cd),)))
                elif st.bn:
                    bwd.append((L.rvip_bn_bwd_reduce, (C.byref(b),)))
                bwd.append((L.rvip_bn_bwd_apply, (C.byref(b),)))
            if first and self.kd == 3:
                bwd.append((L.rvip_conv3d_c1_wgrad, (_ptr(self.act['input_1']), _ptr(dz), P.g(st.conv, 'kernel'), n, self.depth,
                                                     st.h, st.w, st.cout, dt, _ptr(self.ws_wg), C.c_size_t(self.ws_wg_bytes))))
                continue
            if first and self.cimg > 1:
                bwd.append((L.rvip_conv3x3_cn_wgrad, (_ptr(self.act['input_1']), _ptr(dz), P.g(st.conv, 'kernel'), n, st.h, st.w,
                                                      self.cimg, st.cout, dt, _ptr(self.ws_wg), C.c_size_t(self.ws_wg_bytes))))
                continue
            if first:
                nr1 = L.rvip_conv3x3_c1_wgrad_rows(n, st.h, st.w, st.cout, dt) if (defer and os.environ.get('RVIP_C1_FOLD_BATCHED', '1') != '0') else 0
                if nr1 > 0:       # its partial rows join the step's batched fold of narrow rows (one launch fewer, and a 1 024-row fold off the end of the step)
                    rb1 = torch.empty(nr1 * 9 * st.cout, dtype=torch.float32, device=self.ws.device)
                    self._fold_bufs.append(rb1)
                    bwd.append((L.rvip_conv3x3_c1_wgrad, (_ptr(self.act['input_1']), _ptr(dz), None, n, st.h, st.w,
                                                          st.cout, dt, _ptr(rb1), C.c_size_t(rb1.numel() * 4))))
                    narrow.append((rb1, P.g(st.conv, 'kernel'), nr1, 9 * st.cout))
                else:
                    bwd.append((L.rvip_conv3x3_c1_wgrad, (_ptr(self.act['input_1']), _ptr(dz), P.g(st.conv, 'kernel'), n, st.h, st.w,
                                                          st.cout, dt, _ptr(self.ws_wg), C.c_size_t(self.ws_wg_bytes))))
                continue
            wg, dg = wg_desc[st.conv], dg_desc[st.conv]
            if defer and st.conv not in dot_rows:   # (a layer whose kernel gradient feeds rvip_bn_bwd_coef is folded at once, with the dot rows)
                ns = L.rvip_conv3x3_wgrad_splits(C.byref(wg))
                sbuf = torch.empty(ns * 9 * st.cin * st.cout, dtype=torch.float32, device=self.ws.device)
                self._fold_bufs.append(sbuf)
                wg.workspace, wg.workspace_bytes, wg.defer_fold = sbuf.data_ptr(), sbuf.numel() * 4, 1
                wide.append((sbuf, P.g(st.conv, 'kernel'), ns, 9 * st.cin * st.cout))
            fuse_down = st.up0 == 1 and fuse_down_on
            mode = self.bwd_mode[st.conv]
            if mode == 'pair':
                # both kernels as the two parts of ONE grid (rvip_pair.hip): no fork / join of the graph, ~18 us per layer
                if st.conv in sums_rows:
                    sb = sums_rows[st.conv][0]
                else:                           # (nobody reads this layer's column sums: a scratch row set)
                    sb = torch.zeros(max(L.rvip_conv3x3_fwd_sums_rows(C.byref(dg)), 1) * st.cin, dtype=torch.float32, device=self.ws.device)
                    self._fold_bufs.append(sb)
                bwd.append((L.rvip_conv3x3_wgrad_dgrad, (C.byref(wg), C.byref(dg), _ptr(sb), C.c_size_t(sb.numel() * 4))))
                self.paired.append(st.conv)
            else:
                if mode == 'forkjoin':      # weight gradient (+ its slab fold) on the second stream, beside the data gradient
                    bwd.append((_Fork(self), ()))
                    bwd.append((_OnSide(self, L.rvip_conv3x3_wgrad), (C.byref(wg),)))
                else:
                    bwd.append((L.rvip_conv3x3_wgrad, (C.byref(wg),)))
                if st.conv in sums_rows:    # the column sums of the result ride in the epilogue (sum g of the producers' BN backward)
                    sb = sums_rows[st.conv][0]
                    bwd.append((L.rvip_conv3x3_fwd_sums, (C.byref(dg), _ptr(sb), C.c_size_t(sb.numel() * 4))))
                else:
                    bwd.append((L.rvip_conv3x3_fwd, (C.byref(dg),)))
                if mode == 'forkjoin':
                    bwd.append((_Join(self), ()))
            if st.up0 and not fuse_down:    # 1: UpSampling2D -> 2x2 sum; 2: zero-stuffed (Conv2DTranspose) -> odd positions
                back = L.rvip_upsample2x_bwd if st.up0 == 1 else L.rvip_subsample_odd
                bwd.append((back, (_ptr(self.up_tmp[st.conv]), _ptr(self.grd[st.src0]), n, st.h // 2, st.w // 2, st.c0, dt)))
        flush_folds()
        self.fwd_train, self.fwd_infer, self.bwd = fwd_t, fwd_i, bwd
        self.opt = [(L.rvip_adam_step, (_ptr(P.theta), _ptr(P.grad), _ptr(P.adam_m), _ptr(P.adam_v), C.c_longlong(P.count),
                                        C.c_float(0.9), C.c_float(0.999), C.c_float(1e-7), C.c_float(1.0 / self.loss_scale), _ptr(P.state)))]
        if P.pack_entries:                                       # re-layout of the updated kernels; the same launch counts the step
            fn, args = P.pack_call()
            self.opt.append((L.rvip_pack_all_conv3x3_weights_tick, args + (_ptr(P.state),)))
        else:
            self.opt.append((L.rvip_state_tick, (_ptr(P.state),)))

    # -- execution ------------------------------------------------------------------------------------
    @staticmethod
    def _run(seq, stream):
        s = C.c_void_p(stream)
        for th in seq:
            fn, args = th[0], th[1]
            rc = fn(*args, s)
            if rc:
                N.check(rc, fn.__name__)

    def stream(self):
        return _torch().cuda.current_stream().cuda_stream

    def load_input(self, x, y=None):
        """Host float32 NHWC batch -> device buffers (the generator contract, Generators.py:97-98,228)."""
        torch = _torch()
        xs = torch.from_numpy(np.ascontiguousarray(x, np.float32)).reshape(self.x_stage.shape)
        self.x_stage.copy_(xs, non_blocking=True)
        self.stage_input()
        if y is not None:
            self.y_true.copy_(torch.from_numpy(np.ascontiguousarray(y, np.float32)).reshape(self.y_true.shape), non_blocking=True)

    def stage_input(self):
        """x_stage (fp32) -> network input in the activation dtype, on the current stream."""
        L = N.lib()
        N.check(L.rvip_convert(_ptr(self.x_stage), N.F32, _ptr(self.act['input_1']), self.P.dt,
                               C.c_longlong(self.x_stage.numel()), C.c_void_p(self.stream())), 'rvip_convert')

    def forward(self, training):
        self._run(self.fwd_train if training else self.fwd_infer, self.stream())

    def forward_eval(self):
        self._run(self.fwd_eval, self.stream())

    def backward(self):
        self._run(self.bwd, self.stream())

    def optimizer_step(self):
        self._run(self.opt, self.stream())

    def allreduce_grads(self):
        if self.dp:
            import torch.distributed as dist
            dist.all_reduce(self.P.grad)                   # RCCL sum over xGMI; loss is pre-divided by the global batch

    # -- overlapped data-parallel step: two gradient buckets ------------------------------------------------------
    def overlap_ok(self):
        return (self.dp and self.bwd_split and 0 < self.grad_split < self.P.grad.numel()
                and os.environ.get('RVIP_OVERLAP_ALLREDUCE', '1') != '0')

    def backward_part(self, part):
        """part 0: head + decoder + bottleneck (fills grad[grad_split:]); part 1: encoder (fills grad[:grad_split])"""
        seq = self.bwd[:self.bwd_split] if part == 0 else self.bwd[self.bwd_split:]
        self._run(seq, self.stream())

    def allreduce_bucket_async(self, part):
        """Start the sum all-reduce of the bucket `backward_part(part)` has just produced; returns the work handle.
        The collective runs on the backend's own stream, ordered after the work already queued on the current one."""
        import torch.distributed as dist
        g = self.P.grad[self.grad_split:] if part == 0 else self.P.grad[:self.grad_split]
        return dist.all_reduce(g, async_op=True)

    # -- the training step as the product runs it: captured once per engine, replayed ever after -------------------
    def _step_parts(self):
        """The step as a list of segments; a collective separates two segments (it stays outside the graphs)."""
        def a_all():
            self.stage_input()
            self._run(self.fwd_train, self.stream())
            self._run(self.bwd, self.stream())

        def a_head():
            self.stage_input()
            self._run(self.fwd_train, self.stream())
            self.backward_part(0)
        opt = lambda: self._run(self.opt, self.stream())                                   # noqa: E731
        if not self.dp:
            return [lambda: (a_all(), opt())], []
        if self.overlap_ok():
            return [a_head, lambda: self.backward_part(1), opt], [0, 1]
        return [a_all, opt], [None]

    def capture(self):
        """hipGraph capture of the step (one graph on one GPU; around the RCCL collectives when data-parallel).  Needs one
        eager step before it (lazy module / kernel loading is not capturable).  Returns False -- and the engine stays on eager
        launches, in-process -- if the runtime refuses the capture.

        Nothing may DESTROY a hipGraph while a stream is capturing: ``at::cuda::CUDAGraph::~CUDAGraph`` calls hipDeviceSynchronize
        and throws from the destructor on its error (hipErrorStreamCaptureUnsupported while any stream captures) -> std::terminate
        -> SIGABRT of the process.  A graph dies with its engine, an engine with its model, and a model that sits in a reference
        cycle dies in Python's cyclic collector -- on any thread, at any allocation (torch >= 2.9 no longer collects before a
        capture).  That was round 3's box-dependent abort of the GPU suite.  So: collect BEFORE the capture, on
        this thread, and keep the automatic collector off (it is process-global: all threads) until the capture has ended."""
        torch = _torch()
        parts, _ = self._step_parts()
        graphs = []
        try:
            with capture_guard():
                for fn in parts:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode='thread_local'):       # other threads' (host-only) work is not our capture's business
                        fn()
                    graphs.append(g)
        except Exception as e:                                                            # pragma: no cover (needs a failing runtime)
            import sys
            sys.stderr.write('rvip: hipGraph capture failed (%s: %s); the step runs on eager launches\n' % (type(e).__name__, e))
            torch.cuda.synchronize()
            del graphs[:]
            self._graphs, self.launch_mode = None, 'eager (capture failed)'
            return False
        torch.cuda.synchronize()
        coll = 'RCCL'
        if len(graphs) > 1 and _dist_ready():
            import torch.distributed as dist
            be = str(dist.get_backend())
            coll = 'RCCL' if be == 'nccl' else be          # ('gloo': the one-GPU rehearsals of the N > 1 path)
        self._graphs, self.launch_mode = graphs, 'hipGraph' if len(graphs) == 1 else 'hipGraph x%d + %s between' % (len(graphs), coll)
        return True

    def __del__(self):
        # an engine whose last reference goes on ANOTHER thread (or in an explicit gc.collect() of a callback / pool /
        # checkpoint thread) must not run ~CUDAGraph there -- it synchronises the device, which throws from the destructor while
        # any stream captures (SIGABRT).  The graphs are handed to a module-level list instead and destroyed by
        # whoever next holds _CAPTURE_LOCK outside a capture (capture_guard's entry, release()).
        g = getattr(self, '_graphs', None)
        gy = _GRAVEYARD                      # (None while the interpreter tears the module down: nothing captures any more)
        if g and gy is not None:
            gy.append(g)
            self._graphs = None

    def release(self):
        """Drops the captured graphs and the pinned ring now (training thread, outside any capture, device idle)."""
        torch = _torch()
        with _CAPTURE_LOCK:
            torch.cuda.synchronize()
            _bury_dead_graphs()
            self._graphs, self.launch_mode = None, 'eager'
            self._eager_steps = 0
            self.reset_input_ring()
            self.pin_x = self.pin_y = self.pin_x_np = self.pin_y_np = None

    def train_step(self, marks=None):
        """stage input + fwd + loss + bwd + [all-reduce] + Adam on the batch in ``x_stage`` / ``y_true``.  The first call runs
        eagerly (warm-up), the second captures the step, every later one replays it (RVIP_GRAPH=0: always eager).

        ``marks`` (a list; bench.py's ``dp_segments``): timing events recorded on the launch stream at the step's start, behind
        every segment and behind the wait for the collectives -- the same launches in the same order, nothing else changes."""
        if self._graphs is None and self._eager_steps >= 1 and self.launch_mode == 'eager' and os.environ.get('RVIP_GRAPH', '1') != '0':
            self.capture()
        parts, buckets = self._step_parts() if self._graphs is None else ([g.replay for g in self._graphs], self._step_parts()[1])
        if self._graphs is None:
            self._eager_steps += 1

        def mark(what):
            if marks is not None:
                torch = _torch()
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(torch.cuda.current_stream())
                marks.append((what, ev))
        pending = []
        mark('start')
        for i, run in enumerate(parts):
            run()
            mark('segment %d' % i)
            if i < len(buckets):
                if buckets[i] is None:
                    self.allreduce_grads()
                    mark('all-reduce')
                else:
                    pending.append(self.allreduce_bucket_async(buckets[i]))
            if i == len(parts) - 2 and pending:            # every collective must have landed before the optimiser segment
                for w in pending:
                    w.wait()
                mark('collectives landed')

    def landmarks(self, thr=0.5, want_mask=False):
        torch = _torch()
        L = N.lib()
        n, H, W, K = self.pred.shape
        mask = torch.empty((n, H, W, K), dtype=torch.uint8, device=self.pred.device) if want_mask else None
        N.check(L.rvip_landmarks(_ptr(self.pred), _ptr(self.lm_idx), _ptr(mask) if want_mask else None, n, H * W, K,
                                 C.c_float(thr), C.c_void_p(self.stream())), 'rvip_landmarks')
        return self.lm_idx, mask

    def metrics_from_sums(self):
        """loss + dice metrics of the last forward from the folded sums (one small D2H copy)."""
        s = self.sums.detach().cpu().numpy().astype(np.float64)
        return dict(sq=s[0], bce=s[1], inter=s[2], st=s[3], sp=s[4], lower=(s[5], s[6], s[7]), upper=(s[8], s[9], s[10]))
