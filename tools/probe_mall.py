"""Does a consumer that runs right after its producer find the tensor in the 256 MB Infinity Cache?  (design probe)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import cmr_landmark_detection_amd as rvip
N = rvip._native
L = N.lib()
dev = torch.device('cuda', 0)
s = torch.cuda.current_stream()
sp = C.c_void_p(s.cuda_stream)

def copy(src, dst):
    N.check(L.rvip_convert(C.c_void_p(src.data_ptr()), N.BF16, C.c_void_p(dst.data_ptr()), N.BF16, C.c_longlong(src.numel()), sp), 'convert')

for mb in (16, 33, 67, 134, 268):
    n = mb * 1000 * 1000 // 2
    bufs = [torch.empty(n, dtype=torch.bfloat16, device=dev) for _ in range(8)]
    for b in bufs:
        b.zero_()
    big = torch.empty(600 * 1000 * 1000, dtype=torch.uint8, device=dev)
    res = {}
    for mode in ('cold', 'hot'):
        ts = []
        for rep in range(5):
            big.zero_()                                   # flush the cache with unrelated data
            torch.cuda.synchronize()
            if mode == 'hot':
                copy(bufs[0], bufs[1])                    # producer writes bufs[1]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            copy(bufs[1], bufs[2])                        # consumer reads bufs[1]
            e1.record(s)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[mode] = min(ts)
    print('%4d MB  consumer cold %.1f us (%.2f TB/s r+w)   right after producer %.1f us (%.2f TB/s)' % (
        mb, res['cold'] * 1e3, 2 * mb / res['cold'] / 1e3, res['hot'] * 1e3, 2 * mb / res['hot'] / 1e3))
    del bufs, big
