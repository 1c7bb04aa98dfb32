import ctypes as C, os, sys, torch
sys.path.insert(0, os.getcwd())
from nerf_amd._lib import NerfLevelSet, check, lib, ptr, stream
from nerf_amd.nets import HashGridEncoding
enc = HashGridEncoding(20, 2, 24, device='cuda', seed=0)
cfg, F = enc.cfg, 2
levels = [l for l in range(20) if int(cfg.entries[l]) * F > 16 * 32768]
ls = NerfLevelSet(); ls.count = len(levels)
for i, l in enumerate(levels): ls.level[i] = l
p = enc.params.detach().clone(); m, v = torch.zeros_like(p), torch.zeros_like(p)
ws = torch.empty(1 << 20, dtype=torch.uint8, device='cuda')
def upd():
    check(lib().nerf_table_update_binned(C.byref(cfg), C.byref(ls), 0, ptr(ws), ws.numel(), ptr(p), ptr(m), ptr(v), 1e-2, 0.9, 0.99, 1e-15, 0.0, 1, 1.0, stream()), 'upd')
upd(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): upd()
e1.record(); torch.cuda.synchronize()
print('update with no items: %.3f ms' % (e0.elapsed_time(e1) / 10))
