import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import cmr_landmark_detection_amd as rvip
N = rvip._native
L = N.lib()
dev = torch.device('cuda', 0)
s = torch.cuda.current_stream(); sp = C.c_void_p(s.cuda_stream)
for (n, h, w, c) in ((32, 256, 256, 32), (32, 128, 128, 64), (32, 64, 64, 128)):
    z = torch.randn((n, h, w, c), device=dev).to(torch.bfloat16); y = torch.empty_like(z)
    scale = torch.ones(c, device=dev); shift = torch.zeros(c, device=dev)
    state = torch.zeros(8, dtype=torch.int32, device=dev)
    big = torch.empty(600 * 1000 * 1000, dtype=torch.uint8, device=dev)
    for drop in (0.0, 0.3):
        a = N.ApplyDesc()
        a.z, a.y, a.pooled = z.data_ptr(), y.data_ptr(), None
        a.scale, a.shift, a.act = scale.data_ptr(), shift.data_ptr(), 0
        a.drop_rate, a.mask, a.state, a.layer_id = drop, None, state.data_ptr(), 3
        a.n, a.h, a.w, a.c, a.dtype = n, h, w, c, N.BF16
        best = 1e9
        for rep in range(8):
            big.zero_(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); N.check(L.rvip_bn_apply(C.byref(a), sp), 'apply'); e1.record(s); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        mb = 2 * z.numel() * 2 / 1e6
        print('blocks', os.environ.get('RVIP_APPLY_BLOCKS', '4096'), (n, h, w, c), 'drop', drop, '%.1f us  %.2f TB/s' % (best * 1e3, mb / best / 1e3))
