"""Isolated timing of rvip_bn_apply at the full-resolution shape: plain, with dropout, and with the fused 2x2 max-pool, each after
(a) an unrelated 600 MB memset (cold caches) and (b) a kernel that has just WRITTEN z (the state the training step leaves)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import cmr_landmark_detection_amd as rvip
N = rvip._native
L = N.lib()
dev = torch.device('cuda', 0)
s = torch.cuda.current_stream(); sp = C.c_void_p(s.cuda_stream)
n, h, w, c = 32, 256, 256, 32
z = torch.randn((n, h, w, c), device=dev).to(torch.bfloat16); y = torch.empty_like(z)
z2 = torch.randn((n, h, w, c), device=dev).to(torch.bfloat16)
pooled = torch.empty((n, h // 2, w // 2, c), dtype=torch.bfloat16, device=dev)
scale = torch.ones(c, device=dev); shift = torch.zeros(c, device=dev)
state = torch.zeros(8, dtype=torch.int32, device=dev)
big = torch.empty(600 * 1000 * 1000, dtype=torch.uint8, device=dev)
for name, drop, pool in (('plain', 0.0, False), ('dropout', 0.3, False), ('pool', 0.0, True)):
    a = N.ApplyDesc()
    a.z, a.y, a.pooled = z.data_ptr(), y.data_ptr(), (pooled.data_ptr() if pool else None)
    a.scale, a.shift, a.act = scale.data_ptr(), shift.data_ptr(), 0
    a.drop_rate, a.mask, a.state, a.layer_id = drop, None, state.data_ptr(), 3
    a.n, a.h, a.w, a.c, a.dtype = n, h, w, c, N.BF16
    for prep in ('cold', 'z just written'):
        best = 1e9
        for rep in range(8):
            if prep == 'cold':
                big.zero_()
            else:
                big.zero_(); z.copy_(z2)
            torch.cuda.synchronize() if prep == 'cold' else None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); N.check(L.rvip_bn_apply(C.byref(a), sp), 'apply'); e1.record(s); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        mb = (2 * z.numel() * 2 + (pooled.numel() * 2 if pool else 0)) / 1e6
        print('%-8s %-15s %.1f us  %.2f TB/s' % (name, prep, best * 1e3, mb / best / 1e3))
