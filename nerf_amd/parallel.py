"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm)
for GPUs, gloo for the CPU logic tests.  The reference has no distributed code at all; the
partitioning follows the natural shard axes of the path:

  training   rays are independent, the loss is a mean over rays -> every rank draws its own ray
             batch (seed = base + rank), all ranks hold a full parameter replica, gradients are
             SUMMED by one all-reduce per tensor and scaled by 1/world inside the Adam kernel;
  inference  pixels are independent -> contiguous slices of the flattened ray list, one
             all_gather of the finished slices;
  extraction voxels are independent -> contiguous slabs of the voxel index range, one
             variable-length all_gather.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, force=False):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).  A world of one
    rank gets no process group unless `force` (every helper below is a no-op without one)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            # NERF_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsals on a 1-GPU box)
            backend = os.environ.get('NERF_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            local = local % torch.cuda.device_count()
            torch.cuda.set_device(local)
        prefix = os.environ.get('NERF_STORE_PREFIX')
        if prefix:
            # a second generation of rank processes behind one launcher (bench.py's supervisor): same store as the
            # launcher's, under a key prefix of its own so that nothing of an earlier generation is read back
            from datetime import timedelta
            store = dist.TCPStore(os.environ['MASTER_ADDR'], int(os.environ['MASTER_PORT']), world, False,
                                  timedelta(seconds=600), wait_for_workers=False)
            dist.init_process_group(backend=backend, store=dist.PrefixStore(prefix, store), rank=rank, world_size=world)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        if backend == 'nccl':
            # create the RCCL communicator now, on the main thread, instead of lazily inside the first
            # collective (which the training loop issues from autograd's backward thread)
            dist.all_reduce(torch.zeros(1, device='cuda'))
            torch.cuda.synchronize()
    if torch.cuda.is_available():
        local = local % torch.cuda.device_count()
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def shard_range(total, rank_, world):
    """Contiguous [start, stop) slice of `total` items for `rank_`; sizes differ by at most one."""
    base, rem = divmod(total, world)
    start = rank_ * base + min(rank_, rem)
    return start, start + base + (1 if rank_ < rem else 0)


def all_reduce_sum_(t, async_op=False):
    """In-place SUM all-reduce.  Returns a work handle when async_op and the backend runs it asynchronously
    (None otherwise: gloo moves device tensors through the host, synchronously -- see _stage_cpu)."""
    if _stage_cpu(t):
        c = t.cpu()
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        t.copy_(c)
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)


class _PackedReduce(object):
    """Handle of an asynchronous packed all-reduce: wait() makes the CURRENT stream wait for the collective and then
    unpacks the flat buffer into the gradients / flags it was packed from."""

    def __init__(self, work, flat, small, flags):
        self.work, self.flat, self.small, self.flags = work, flat, small, flags

    def wait(self):
        if self.work is not None:
            self.work.wait()
        o = 0
        for p in self.small:
            p.grad.copy_(self.flat[o:o + p.numel()].view_as(p.grad))
            o += p.numel()
        for f in self.flags:
            f.copy_(self.flat[o:o + f.numel()].view_as(f))
            o += f.numel()


def all_reduce_grads(params, async_op=False, skip=(), flags=()):
    """SUM-all-reduce the .grad of every parameter (largest first so the big transfer starts early).
    `skip`: parameters whose gradient is reduced elsewhere (the pipelined hash table).
    `flags`: small counters (the non-finite flag of the loss) that ride in the same collective as the small
    gradients -- a collective of their own would cost one more latency per step; values must be exact in fp32.
    Returns the list of handles to wait() on when async_op (wait() makes the current stream wait and unpacks)."""
    if not dist.is_initialized():
        return []
    handles = []
    skip_ids = {id(p) for p in skip}
    todo = sorted((p for p in params if p is not None and p.grad is not None and id(p) not in skip_ids),
                  key=lambda t: -t.numel())
    small = [p for p in todo if p.numel() <= (1 << 20)]
    flags = [f for f in flags if f is not None]
    if len(small) + len(flags) > 1:
        # the MLP / latent gradients are a few KB each: one collective for all of them
        flat = torch.cat([p.grad.reshape(-1) for p in small] + [f.reshape(-1).to(torch.float32) for f in flags])
        h = _PackedReduce(all_reduce_sum_(flat, async_op=async_op), flat, small, flags)
        if async_op:
            handles.append(h)
        else:
            h.wait()
        todo = [p for p in todo if p.numel() > (1 << 20)]
        flags = []
    for f in flags:
        all_reduce_sum_(f)
    for p in todo:
        h = all_reduce_sum_(p.grad, async_op=async_op)
        if async_op and h is not None:
            handles.append(h)
    return handles


EARLY_SMALL_REDUCE = os.environ.get('NERF_EARLY_SMALL_REDUCE', '1') != '0'


def backward_and_reduce(loss, model, params, skip=(), flags=()):
    """loss.backward() followed by all_reduce_grads(params, skip, flags) (trainer.py:134 under data parallelism; `skip` may
    be a callable, evaluated after the backward pass).  With a
    level-sharded table and direct gradients the small collective is queued from INSIDE the backward pass, right behind the
    last ray group's d(features) exchange (LevelSharding.after_last_backward): its latency then hides under that exchange
    and the owner's bin fill instead of standing between the backward pass and the table update, which needs the summed
    non-finite flag.  One backward per step only (a second one before step() must use all_reduce_grads itself)."""
    if world_size() == 1:
        loss.backward()
        return
    sh = getattr(model, 'level_sharding', None)
    early = (EARLY_SMALL_REDUCE and sh is not None and getattr(model, 'direct_grads', False) and model._sharded())
    box = []
    _skip = lambda: skip() if callable(skip) else skip      # (may depend on what the backward pass did: evaluated late)
    if early:
        sh.after_last_backward = lambda: box.append(all_reduce_grads(params, async_op=True, skip=_skip(), flags=flags))
    try:
        loss.backward()
    finally:
        if early:
            sh.after_last_backward = None
    if box:
        for h in box[0]:
            h.wait()
    else:
        all_reduce_grads(params, skip=_skip(), flags=flags)


def gather_slices(local, total, dim=0):
    """all_gather contiguous slices produced with shard_range back into one tensor of length `total`."""
    world = world_size()
    if not dist.is_initialized():
        return local
    sizes = [shard_range(total, r, world) for r in range(world)]
    maxlen = max(b - a for a, b in sizes)
    pad_shape = list(local.shape)
    pad_shape[dim] = maxlen
    padded = local.new_zeros(pad_shape)
    padded.narrow(dim, 0, local.shape[dim]).copy_(local)
    bufs = _all_gather_list(padded)
    return torch.cat([b.narrow(dim, 0, e - s) for b, (s, e) in zip(bufs, sizes)], dim=dim)


def _all_gather_list(t):
    world = world_size()
    if _stage_cpu(t):
        c = t.cpu()
        bufs = [torch.empty_like(c) for _ in range(world)]
        dist.all_gather(bufs, c)
        return [b.to(t.device) for b in bufs]
    bufs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(bufs, t)
    return bufs


def gather_varlen(local):
    """all_gather of [P_r, C] tensors with per-rank row counts (dense extraction results)."""
    world = world_size()
    if not dist.is_initialized():
        return local
    count = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
    counts = [int(c.item()) for c in _all_gather_list(count)]
    maxlen = max(max(counts), 1)
    padded = local.new_zeros((maxlen,) + tuple(local.shape[1:]))
    padded[:local.shape[0]] = local
    bufs = _all_gather_list(padded)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def _stage_cpu(t):
    """gloo cannot move device tensors reliably (see tools/gloo_gpu_probe.py): stage through the host."""
    return dist.get_backend() == 'gloo' and t.is_cuda


def all_gather_rows(local, async_op=False):
    """[R, ...] on every rank -> [world*R, ...], rank-major (equal R on all ranks).
    async_op: returns (out, work); `out` is valid on the current stream after work.wait() (work is None when the backend
    ran the collective synchronously)."""
    world = world_size()
    if not dist.is_initialized():
        return (local, None) if async_op else local
    local = local.contiguous()
    if _stage_cpu(local):
        src = local.cpu()
        out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype)
        dist.all_gather_into_tensor(out, src)
        out = out.to(local.device)
        return (out, None) if async_op else out
    out = local.new_empty((world * local.shape[0],) + tuple(local.shape[1:]))
    work = dist.all_gather_into_tensor(out, local, async_op=async_op)
    return (out, work) if async_op else out


def all_to_all_blocks(recv_blocks, send_blocks, async_op=False):
    """send_blocks[r] -> rank r, recv_blocks[r] <- rank r (flat, contiguous views; sizes must match pairwise).
    The caller's OWN block is not transferred -- callers fill / consume it locally while the exchange is in flight.
    Returns a work handle when async_op and the backend is asynchronous (RCCL), else None."""
    if not dist.is_initialized():      # no process group: the only block is the caller's own, which it fills locally
        return None
    me, world = rank(), world_size()
    if dist.get_backend() == 'gloo':       # no list all-to-all in gloo: one flat exchange of host tensors
        ins = [b.reshape(-1).cpu() for i, b in enumerate(send_blocks) if i != me]
        out_sizes = [b.numel() if i != me else 0 for i, b in enumerate(recv_blocks)]
        in_sizes = [b.numel() if i != me else 0 for i, b in enumerate(send_blocks)]
        out = torch.empty(sum(out_sizes), dtype=send_blocks[0].dtype)
        dist.all_to_all_single(out, torch.cat(ins) if ins else out.new_empty(0), out_sizes, in_sizes)
        o = 0
        for i, b in enumerate(recv_blocks):
            if i != me:
                b.reshape(-1).copy_(out[o:o + b.numel()])
                o += b.numel()
        return None
    empty = send_blocks[0].new_empty(0)
    outs = [b.reshape(-1) if i != me else empty for i, b in enumerate(recv_blocks)]
    ins = [b.reshape(-1) if i != me else empty for i, b in enumerate(send_blocks)]
    work = dist.all_to_all(outs, ins, async_op=async_op)
    return work if async_op else None


def broadcast_(t, src):
    """In-place broadcast of a (contiguous view of a) tensor from rank `src`."""
    if not dist.is_initialized():
        return t
    if _stage_cpu(t):
        c = t.cpu()
        dist.broadcast(c, src)
        t.copy_(c)
        return t
    dist.broadcast(t, src)
    return t
