"""Which torch (non-library) kernels run in a train step, with their input shapes and python call sites
(torch.profiler): what is left of the "glue" around the HIP library calls."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.profiler import profile, ProfilerActivity

dev = torch.device('cuda:0')
model, renderer, loader, opt = bench.build_gpu(dev, 0, 4096)
for it in range(4):
    bench.train_step(model, renderer, loader, opt, 4096, it, 1)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for it in range(4, 8):
        bench.train_step(model, renderer, loader, opt, 4096, it, 1)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True, group_by_stack_n=6).table(sort_by='device_time_total', row_limit=40,
                                                                             max_name_column_width=50, max_src_column_width=90))
