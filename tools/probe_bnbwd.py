"""Design probe: isolated timing of the BN-backward passes (reduce: reads gy, z; apply: reads gy, z, writes dz)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import cmr_landmark_detection_amd as rvip
N = rvip._native
L = N.lib()
dev = torch.device('cuda', 0)
s = torch.cuda.current_stream(); sp = C.c_void_p(s.cuda_stream)
for (n, h, w, c) in ((32, 256, 256, 32), (32, 128, 128, 64), (32, 32, 32, 256)):
    rows = n * h * w
    z = torch.randn((n, h, w, c), device=dev).to(torch.bfloat16); gy = torch.randn((n, h, w, c), device=dev).to(torch.bfloat16)
    dz = torch.empty_like(z)
    f = lambda: torch.ones(c, device=dev)
    gamma, mean, invstd, scale, shift, dgamma, dbeta, dbias = f(), f(), f(), f(), f(), f(), f(), f()
    coef = torch.ones(3 * c, device=dev)
    state = torch.zeros(8, dtype=torch.int32, device=dev)
    wsb = L.rvip_reduce_workspace(rows, 16 * c)
    ws = torch.empty(wsb // 4 + 16, dtype=torch.float32, device=dev)
    big = torch.empty(600 * 1000 * 1000, dtype=torch.uint8, device=dev)
    b = N.BnBwdDesc()
    b.dy, b.z, b.dz = gy.data_ptr(), z.data_ptr(), dz.data_ptr()
    b.gamma, b.mean, b.invstd, b.scale, b.shift = gamma.data_ptr(), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr()
    b.dgamma, b.dbeta, b.dbias, b.coef = dgamma.data_ptr(), dbeta.data_ptr(), dbias.data_ptr(), coef.data_ptr()
    b.act, b.act_after_bn = N.ACT['relu'], 0
    b.drop_rate, b.mask, b.state, b.layer_id = float(os.environ.get("DROP", "0.3")), None, state.data_ptr(), 2
    b.rows, b.c, b.dtype = rows, c, N.BF16
    b.workspace, b.workspace_bytes = ws.data_ptr(), wsb
    for name, fn, nbytes in (('reduce', L.rvip_bn_bwd_reduce, 2), ('apply', L.rvip_bn_bwd_apply, 3)):
        best = 1e9
        for rep in range(8):
            big.zero_(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); N.check(fn(C.byref(b), sp), name); e1.record(s); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        mb = nbytes * z.numel() * 2 / 1e6
        print((n, h, w, c), name, '%.1f us  %.2f TB/s (incl. fold launch)' % (best * 1e3, mb / best / 1e3))
