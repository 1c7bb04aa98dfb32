#!/usr/bin/env python3
"""Benchmark of the render-and-train hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Metric (BASELINE.json): rays/sec (train step + render) at 64 coarse + 128 fine samples/ray.
A "step" is one training step on a batch of `--rays` rays PER GPU (weak scaling): device-side
batch sampling, render (ray generation, coarse density pass, resampling, fine field pass,
compositing), fused losses, backward (compositing, field network, hash-table scatter), gradient
all-reduce over RCCL when N > 1, and the dense fused Adam step over every parameter (the
reference's torch.optim.Adam semantics, train.py:84-88).  Network and renderer are the metric
config: config_base network (20 levels x 2 features, 2^24 table, SH4, latent 48, 64-wide MLPs),
steps [64], z_bounds [1.6, 4.8], 128 importance samples; synthetic procedural scene
(100 poses, 800x800), random-init weights.  Inputs are resident in HBM when the timed region starts.

The JSON line also carries
  * "render": full 800x800 image inference (BASELINE configs[1]) in rays/s, measured after the
    training steps with the same weights;
  * "roofline" for the kernel that takes the largest share of the step, from live per-launch
    device timings (HIP events on the launch stream);
  * "kernels": the per-entry-point breakdown the roofline figure was picked from;
  * "cpu_baseline": the CPU oracle (oracle/ref_cpu.py, a port: the reference network cannot run on
    a CPU) timed on this box's host cores on a bounded sample of the same workload (rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # f32-input MFMA (v_mfma_f32_32x32x2_f32), dense

NET = dict(encoding_precision='float32', encoding_n_levels=20, encoding_n_features_per_level=2,
           encoding_log2_hashmap_size=24, geo_feat_dim=15, sigma_hidden_dim=64, sigma_num_layers=2,
           encoding_dir_precision='float32', encoding_dir_encoding='SphericalHarmonics', encoding_dir_degree=4,
           latent_embedding_dim=48, color_hidden_dim=64, color_num_layers=3)
STEPS, Z_BOUNDS, SF, ALPHA, INNER, OUTER = [64], [1.6, 4.8], 128, 0.2, 2, 2.5
N_IMAGES, H, W = 100, 800, 800


def algorithmic_bytes(name, n_rays, n_params, Sc=64, Sf=SF, L=20, F=2):
    """Algorithmic HBM bytes of ONE launch (SURVEY.md section 8d per-unit figures x units per launch)."""
    gather = L * 8 * F * 4                       # 1280 B / point
    return {
        'nerf_density_fwd': n_rays * Sc * (gather + 4),
        'nerf_field_fwd': n_rays * Sf * (gather + L * F * 4 + 4 + 12 + 12),
        'nerf_field_bwd': n_rays * Sf * (2 * L * F * 4 + 4 + 12),
        'nerf_hashgrid_scatter': n_rays * Sf * (2 * gather + L * F * 4),
        # level-sharded table (N > 1): every rank handles L/N levels of N x the points = the same totals
        'nerf_hashgrid_scatter_levels': n_rays * Sf * (2 * gather + L * F * 4),
        'nerf_hashgrid_encode_levels': n_rays * (Sc + Sf) * (gather + L * F * 4),
        'nerf_field_from_features': n_rays * Sf * (L * F * 4 + 4 + 12 + 12),
        'nerf_density_from_features': n_rays * Sc * (L * F * 4 + 4),
        'nerf_adam_step': None,                  # per-tensor launches: 32 B / parameter, summed below
        # per-ray kernels: sigma in, fine depths (log and linear) out, + the noise array unless drawn in-kernel
        'nerf_coarse_to_fine': n_rays * (4 * Sc + 12 * Sf),
        'nerf_coarse_to_fine_rng': n_rays * (4 * Sc + 8 * Sf),
        # sigma 4 + rgb 12 + z 4 + zlog 4 read, weights 4 + zlog_s 4 written per sample; image + invdepth per ray
        'nerf_composite_fwd': n_rays * (Sf * 32 + 16),
        # sigma 4 + rgb 12 + zlog 4 + dweights 4 read, dsigma 4 + drgb 12 written per sample; dimage per ray
        'nerf_composite_bwd': n_rays * (Sf * 40 + 12),
        'nerf_losses': n_rays * (Sf * 12 + 36),
        # the fused tail of the training step: compositing + losses (gt, image, invdepth per ray), and their backward
        # (sigma, rgb, zlog read, dsigma, drgb written; gt + image per ray)
        'nerf_composite_loss_fwd': n_rays * (Sf * 32 + 28),
        'nerf_composite_loss_bwd': n_rays * (Sf * 36 + 24),
        'nerf_raygen': n_rays * (16 + 36 + 64 + 24),
    }.get(name)


def algorithmic_flops(name, n_rays, Sc=64, Sf=SF):
    return {'nerf_density_fwd': n_rays * Sc * 7168, 'nerf_field_fwd': n_rays * Sf * 25856,
            'nerf_field_bwd': n_rays * Sf * 3 * 25856, 'nerf_field_from_features': n_rays * Sf * 25856,
            'nerf_density_from_features': n_rays * Sc * 7168}.get(name)


def build_gpu(device, rank, n_rays, pipeline=True, shard_levels=None):
    from nerf_amd.loaders.procedural import ProceduralSceneLoader
    from nerf_amd.nets import NeRFNetwork
    from nerf_amd.optim import FusedAdam
    from nerf_amd.renderer import NerfRenderer
    model = NeRFNetwork(N_IMAGES, device=device, seed=0, **NET)
    renderer = NerfRenderer(model, INNER, OUTER, Z_BOUNDS, STEPS, SF, ALPHA)
    loader = ProceduralSceneLoader(N_IMAGES, H, W, seed=0, device=device, batch_seed=1000 + rank)
    opt = FusedAdam([
        {'name': 'encoding', 'params': list(model.encoder.parameters()), 'lr': 2e-2},
        {'name': 'latent_emb', 'params': [model.latent_emb], 'lr': 2e-2},
        {'name': 'net', 'params': list(model.sigma_net.parameters()) + list(model.color_net.parameters()),
         'weight_decay': 1e-6, 'lr': 1e-3}], betas=(0.9, 0.99), eps=1e-15).attach(model, pipeline=pipeline, shard_levels=shard_levels)
    return model, renderer, loader, opt


_NEXT_BATCH = {}        # loader id -> the batch Trainer.train_epoch's prefetch drew for the coming step (N > 1)


def train_step(model, renderer, loader, opt, n_rays, it, world):
    from nerf_amd import helpers, parallel
    opt.zero_grad()
    batch = _NEXT_BATCH.pop(id(loader), None) or loader.get_random_batch(n_rays)
    n, h, w, K, E, rgb_gt, color_bg, _ = batch
    lam_dist = 10 ** (it / 10000 * 3 - 5)            # trainer.py:160 with lambda 1e-5 -> 1e-2 over 10k iters
    if os.environ.get('NERF_FUSED_TAIL', '1') != '0':
        loss = renderer.render_loss(n, h, w, K, E, color_bg, rgb_gt, lam_dist, 1e-1)[0]   # what Trainer.train_step calls
    else:
        rgb, weights, zs, _ = renderer.render(n, h, w, K, E, color_bg)
        loss, _ = helpers.fused_train_loss(rgb, rgb_gt, weights, zs, lam_dist, 1e-1)
    opt.skip_flag = loss.nonfinite_flag        # Trainer.train_epoch's non-finite guard (device side, no sync)
    if world > 1:
        skip = lambda: [opt._table_param] if opt.table_reduced_by_pipeline() else []
        parallel.backward_and_reduce(loss, model, [p for g in opt.param_groups for p in g['params']], skip=skip,
                                     flags=[opt.skip_flag])
        opt.grad_scale = 1.0 / world
        if os.environ.get('NERF_PREFETCH_RAYS', '1') != '0':
            # as Trainer.train_epoch: the next batch, its rays and the ray all-gather of its coarse pass before the optimizer step
            nxt = loader.get_random_batch(n_rays)
            renderer.prefetch_batch(*nxt[:5])
            _NEXT_BATCH[id(loader)] = nxt
    else:
        loss.backward()
    opt.step()
    return loss


def _cpu_steps(ora, R, opt, n_rays, steps, warm, seed):
    """`warm` + `steps` oracle train steps of `n_rays` rays; per timed step (render fwd, loss + backward, Adam) seconds."""
    from oracle import ref_cpu
    from nerf_amd.loaders.procedural import sphere_cameras
    K_all, E_all = sphere_cameras(N_IMAGES, 0)
    g = torch.Generator().manual_seed(seed)
    rows = []
    for it in range(warm + steps):
        n = torch.randint(0, N_IMAGES, (n_rays,), generator=g)
        h = torch.randint(0, H, (n_rays,), generator=g)
        w = torch.randint(0, W, (n_rays,), generator=g)
        bg, gt = torch.rand(3, generator=g), torch.rand(n_rays, 3, generator=g)
        t0 = time.perf_counter()
        opt.zero_grad()
        img, wts, zs, _ = R.render(n, h, w, K_all[n], E_all[n], bg)
        t1 = time.perf_counter()
        loss, _ = ref_cpu.train_loss(img, gt, wts, zs, it, 10000, 1e-5, 1e-2, 1e-1, 1e-1)
        loss.backward()
        t2 = time.perf_counter()
        opt.step()
        t3 = time.perf_counter()
        if it >= warm:
            rows.append((t1 - t0, t2 - t1, t3 - t2))
    with torch.no_grad():           # forward-only render of the last batch (the inference half of the metric)
        t0 = time.perf_counter()
        R.render(n, h, w, K_all[n], E_all[n], bg)
        t_render = time.perf_counter() - t0
    med = lambda xs: sorted(xs)[len(xs) // 2]
    tot = med([sum(r) for r in rows])
    return {'seconds_per_step': tot, 'render_fwd_s': med([r[0] for r in rows]), 'loss_backward_s': med([r[1] for r in rows]),
            'adam_s': med([r[2] for r in rows]), 'rays_per_s': n_rays / tot, 'render_only_rays_per_s': n_rays / t_render}


def per_ray_kernels_at(n_rays=65536, Sc=64, Sf=SF, reps=20):
    """The per-ray kernels (resampling, compositing, their fused training forms and backward passes) at a batch where
    bandwidth, not launch latency, is what is measured: `n_rays` rays x (Sc coarse, Sf fine) of synthetic rows, every entry
    point called through the C ABI `reps` times between two HIP events on the launch stream.  Returns a list of
    {name, avg_launch_ms, algorithmic_bytes_per_launch, algorithmic_GBps, hbm_frac}."""
    import ctypes as C
    import math
    from nerf_amd._lib import check, lib, ptr, stream
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(3)
    N = n_rays
    sig_c = torch.rand(N, Sc, device=dev, generator=g) * 20
    zlog_c = torch.linspace(math.log10(1.6), math.log10(4.8), Sc + 1, device=dev)[:Sc].contiguous()
    lin = torch.linspace(0, 1 - 1 / Sf, Sf).to(dev)
    steps = (C.c_int32 * 1)(Sc)
    zlog_f, z_f = torch.empty(N, Sf, device=dev), torch.empty(N, Sf, device=dev)
    sig = torch.rand(N, Sf, device=dev, generator=g) * 20
    rgb = torch.rand(N, Sf, 3, device=dev, generator=g)
    bg, gt = torch.rand(3, device=dev, generator=g), torch.rand(N, 3, device=dev, generator=g)
    image, invd = torch.empty(N, 3, device=dev), torch.empty(N, device=dev)
    wts, zs = torch.empty(N, Sf, device=dev), torch.empty(N, Sf, device=dev)
    dsig, drgb = torch.empty(N, Sf, device=dev), torch.empty(N, Sf, 3, device=dev)
    dimg, dw = torch.rand(N, 3, device=dev, generator=g), torch.rand(N, Sf, device=dev, generator=g)
    sums = torch.empty(8 + 4 * 4096, device=dev, dtype=torch.float64)
    out6 = torch.empty(6, device=dev)
    lo, rng = math.log10(1.6), math.log10(4.8) - math.log10(1.6)
    L = lib()
    calls = [
        ('nerf_coarse_to_fine_rng', lambda: L.nerf_coarse_to_fine_rng(ptr(sig_c), ptr(zlog_c), 0, 1, 2, ptr(lin), N, Sc, Sf, 0.2,
                                                                      steps, 1, ptr(zlog_f), ptr(z_f), stream())),
        ('nerf_composite_fwd', lambda: L.nerf_composite_fwd(ptr(sig), ptr(rgb), ptr(z_f), ptr(zlog_f), ptr(bg), N, Sf, lo, rng,
                                                            ptr(image), ptr(invd), ptr(wts), ptr(zs), stream())),
        ('nerf_composite_bwd', lambda: L.nerf_composite_bwd(ptr(sig), ptr(rgb), ptr(zlog_f), ptr(bg), ptr(dimg), ptr(dw), N, Sf,
                                                            ptr(dsig), ptr(drgb), stream())),
        ('nerf_composite_loss_fwd', lambda: L.nerf_composite_loss_fwd(ptr(sig), ptr(rgb), ptr(z_f), ptr(zlog_f), ptr(bg), ptr(gt), N,
                                                                      Sf, lo, rng, 1e-3, 1e-1, ptr(image), ptr(invd), ptr(wts),
                                                                      ptr(zs), ptr(sums), ptr(out6), stream())),
        ('nerf_composite_loss_bwd', lambda: L.nerf_composite_loss_bwd(ptr(sig), ptr(rgb), ptr(zlog_f), ptr(bg), ptr(gt), ptr(image),
                                                                      None, N, Sf, lo, rng, 1e-3, 1e-1, ptr(dsig), ptr(drgb),
                                                                      sums.data_ptr() + 40, stream())),
    ]
    rows = []
    for name, fn in calls:
        for _ in range(3):
            check(fn(), name)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        nbytes = algorithmic_bytes(name, N, 0, Sc=Sc, Sf=Sf)
        rows.append({'name': name, 'rays': N, 'avg_launch_ms': round(ms, 4), 'algorithmic_bytes_per_launch': nbytes,
                     'algorithmic_GBps': round(nbytes / (ms * 1e-3) / 1e9, 1),
                     'hbm_frac': round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
    return rows


def cpu_baseline(n_rays=4096, steps=10, warm=3, threads=16):
    """The CPU oracle (oracle/ref_cpu.py, fp32 torch CPU; a port -- the reference's network needs the CUDA-only tinycudann)
    on a bounded sample of the SAME workload: `steps` timed train steps (after `warm` warm-up steps; BASELINE.md section 3:
    3 + >= 10, median) of `n_rays` rays of the
    metric configuration, config_base network with the 2^24 table and torch.optim.Adam over all of it; then, for
    orientation, SURVEY's C1 case (1024 rays, 480 coarse + 128 fine samples).  Phases are reported separately: the dense
    Adam pass over 400 M parameters does not shrink with the batch."""
    from oracle import ref_cpu
    threads = min(threads, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    ora = ref_cpu.OracleNetwork(N_IMAGES, seed=0, **NET)
    opt = ref_cpu.make_optimizer(ora)
    R = ref_cpu.OracleRenderer(ora, INNER, OUTER, Z_BOUNDS, STEPS, SF, ALPHA)
    main = _cpu_steps(ora, R, opt, n_rays, steps, warm, 0)
    R1 = ref_cpu.OracleRenderer(ora, INNER, OUTER, [0.1, 1, 10, 100, 1000], [256, 128, 64, 32], SF, ALPHA)
    c1 = _cpu_steps(ora, R1, opt, 1024, 2, 1, 1)
    cpu_model = ''
    try:
        with open('/proc/cpuinfo') as f:
            cpu_model = next((l.split(':', 1)[1].strip() for l in f if l.startswith('model name')), '')
    except OSError:
        pass
    return {'value': main['rays_per_s'], 'unit': 'rays/s', 'cores': threads, 'kind': 'port',
            'render_rays_per_s': main['render_only_rays_per_s'], 'cpu_model': cpu_model, 'host_cpus': os.cpu_count(),
            'sample': '%d timed train steps of %d rays (%d warm-up, median), %d coarse + %d fine samples, oracle/ref_cpu.py fp32 '
                      'torch CPU, same network / renderer / Adam configuration' % (steps, n_rays, warm, sum(STEPS), SF),
            'seconds_per_step': main['seconds_per_step'],
            'phases_s': {k: main[k] for k in ('render_fwd_s', 'loss_backward_s', 'adam_s')},
            'c1_1024_rays_480_coarse': {k: c1[k] for k in ('rays_per_s', 'render_only_rays_per_s', 'seconds_per_step',
                                                           'render_fwd_s', 'loss_backward_s', 'adam_s')}}


def _beat():
    """Tell the supervisor (if any) that this worker is alive."""
    hb = os.environ.get('NERF_BENCH_HEARTBEAT')
    if hb:
        with open(hb, 'w') as f:
            f.write(str(time.time()))


def supervise(argv):
    """N > 1 under torch.distributed.run: every launched rank process becomes a supervisor that never touches the GPU
    and runs the real rank as a child process (the worker).  The level-sharded exchange has only ever run over gloo and
    over RCCL in a world of one rank (there is one GPU per test box): should its first multi-GPU run stall in a collective
    instead of raising, every supervisor sees its worker's heartbeat stop, ends that worker (its own child, by PID) and
    starts a second worker generation on the replicated table (dense all-reduce, the plainest RCCL path), which says so in
    config.parallelism.  A worker that succeeds is simply passed through: same stdout, same exit code."""
    import signal
    import subprocess
    import tempfile
    stall_s = float(os.environ.get('NERF_BENCH_STALL_S', '150'))
    first_s = float(os.environ.get('NERF_BENCH_FIRST_S', '420'))     # import torch on a fresh box + build + rendezvous
    rank = int(os.environ.get('RANK', '0'))
    attempts = [('sharded', {})]
    if os.environ.get('NERF_TABLE_PARALLEL') != '0':
        attempts.append(('replicated', {'NERF_TABLE_PARALLEL': '0'}))
    rc = 1
    for i, (mode, extra) in enumerate(attempts):
        hb = os.path.join(tempfile.gettempdir(), 'nerf_bench_hb_%d_%d_%d' % (os.getpid(), rank, i))
        env = dict(os.environ, NERF_BENCH_WORKER='1', NERF_BENCH_HEARTBEAT=hb, NERF_STORE_PREFIX='nerfgen%d' % i, **extra)
        if i > 0:
            env['NERF_BENCH_NOTE'] = ' [level-sharded worker generation stalled or failed; replicated table]'
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env)
        started, last = time.time(), None
        stalled = False
        while child.poll() is None:
            time.sleep(1.0)
            try:
                last = os.path.getmtime(hb)
            except OSError:
                pass
            quiet = time.time() - (last if last is not None else started)
            if quiet > (stall_s if last is not None else first_s):
                stalled = True
                sys.stderr.write('bench supervisor rank %d: %s worker silent for %.0f s, ending it\n' % (rank, mode, quiet))
                child.send_signal(signal.SIGKILL)
                child.wait()
        try:
            os.remove(hb)
        except OSError:
            pass
        rc = child.returncode
        if rc == 0 and not stalled:
            return 0
        sys.stderr.write('bench supervisor rank %d: %s worker ended with code %s\n' % (rank, mode, rc))
        time.sleep(5.0)      # let the driver reclaim the killed worker's queues before the next generation starts
    return rc if rc else 1


def main():
    if (int(os.environ.get('WORLD_SIZE', '1')) > 1 and os.environ.get('TORCHELASTIC_USE_AGENT_STORE') == 'True'
            and os.environ.get('NERF_BENCH_WORKER') != '1' and os.environ.get('NERF_BENCH_SUPERVISE', '1') != '0'):
        sys.exit(supervise(sys.argv[1:]))
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--rays', type=int, default=4096, help='rays per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-render', action='store_true')
    ap.add_argument('--no-pipeline', action='store_true', help='dense table all-reduce + one Adam pass (no per-level pipeline)')
    ap.add_argument('--profile-steps', type=int, default=5)
    ap.add_argument('--table-log2', type=int, default=24, help='REHEARSAL ONLY: smaller hash table (the metric config is 24)')
    ap.add_argument('--levels', type=int, default=20, help='REHEARSAL ONLY: hash-grid levels (the metric config is 20; '
                    "SURVEY's light variant is --levels 16 --table-log2 19)")
    ap.add_argument('--coarse', default='64', help="coarse sampling: '64' (metric config, z_bounds [1.6,4.8]) or "
                    "'480' (config_base: steps [256,128,64,32], z_bounds [.1,1,10,100,1000]; SURVEY C1/C4 parity sizes)")
    ap.add_argument('--table-precision', default='float32', choices=['float32', 'float16'],
                    help="REHEARSAL ONLY: 'float16' gathers from a half-precision image of the table (the metric config is float32)")
    ap.add_argument('--hang-trace', type=int, default=0, help='dump all Python stacks to stderr every N seconds')
    args = ap.parse_args()
    if args.hang_trace > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.hang_trace, repeat=True)

    _beat()
    from nerf_amd import _lib, parallel
    rank, world, local = parallel.init_from_env()
    _beat()
    if world != args.gpus and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    device = torch.device('cuda', local if world > 1 else 0)
    torch.cuda.set_device(device)
    import torch.distributed as dist

    NET['encoding_log2_hashmap_size'] = args.table_log2
    NET['encoding_n_levels'] = args.levels
    NET['encoding_precision'] = args.table_precision
    if args.coarse == '480':
        STEPS[:] = [256, 128, 64, 32]
        Z_BOUNDS[:] = [0.1, 1, 10, 100, 1000]
    model, renderer, loader, opt = build_gpu(device, rank, args.rays, pipeline=not args.no_pipeline)
    _beat()
    fallback_note = os.environ.get('NERF_BENCH_NOTE', '')
    if world > 1 and model.level_sharding is not None:
        # The level-sharded exchange has only ever run over gloo (one-GPU box); if its first step raises over RCCL --
        # an API-level error is raised on every rank alike -- fall back to the replicated table, and say so.
        try:
            if os.environ.get('NERF_TEST_BENCH_STALL') == '1':      # tests: a collective that never completes
                time.sleep(3600)
            train_step(model, renderer, loader, opt, args.rays, 0, world)
            torch.cuda.synchronize()
        except Exception as e:      # noqa: BLE001
            fallback_note = ' [level-sharded path failed (%s: %s); replicated table]' % (type(e).__name__, str(e)[:200])
            sys.stderr.write('rank %d:%s\n' % (rank, fallback_note))
            del model, renderer, loader, opt
            torch.cuda.empty_cache()
            model, renderer, loader, opt = build_gpu(device, rank, args.rays, pipeline=not args.no_pipeline,
                                                     shard_levels=False)
    n_params = sum(p.numel() for g in opt.param_groups for p in g['params'])
    model_sharded = model.level_sharding
    if model_sharded is not None:      # this rank's Adam pass only covers the levels it owns
        owned = sum(b - a for a, b in (model_sharded.level_slice(l) for l in model_sharded.mine))
        n_params = n_params - model.encoder.params.numel() + owned

    _beat()
    for it in range(args.warmup):
        train_step(model, renderer, loader, opt, args.rays, it, world)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    _beat()
    t0 = time.perf_counter()
    for it in range(args.steps):
        train_step(model, renderer, loader, opt, args.rays, args.warmup + it, world)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * args.rays * args.steps / elapsed
    _beat()

    # ---- per-kernel device times (live, HIP events on the launch stream) ----
    if args.profile_steps > 0:
        _lib.kernel_timing(True)
    for it in range(args.profile_steps):
        train_step(model, renderer, loader, opt, args.rays, it, world)
    prof = _lib.kernel_timing(False) if args.profile_steps > 0 else {}
    _beat()
    kernels = []
    n_pts_scatter = args.rays * SF * (world if model_sharded is not None else 1)
    pipe = model.table_pipeline
    # the backward may run as two ray chunks (nets._FieldFn.backward): every level is then scattered by two launches
    n_chunks = 2 if (pipe is not None and model_sharded is None and pipe.early_events) else 1
    for name, (launches, total_ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        per_step_ms = total_ms / args.profile_steps
        base = name.split('[')[0]
        cls = name[len(base):]
        # algorithmic bytes of ALL launches of this class in one step
        if base == 'nerf_adam_step':
            bytes_step = None
            if cls.startswith('[table slice'):          # 32 B / parameter: p, g, m, v read; p, m, v written, g cleared
                bytes_step = 32 * float(cls.split(',')[1].split('M')[0]) * 1e6 * (launches / args.profile_steps)
        elif base in ('nerf_hashgrid_scatter', 'nerf_hashgrid_scatter_levels'):
            lv = int(cls[1:].split(' ')[0])             # levels per launch
            per_level = n_pts_scatter * (2 * 8 * 2 * 4 + 2 * 4)     # RMW of 8 corners x F floats + F floats of dfeat
            bytes_step = per_level * lv * (launches / args.profile_steps) / n_chunks
        elif base in ('nerf_table_update_binned', 'nerf_table_bin_levels', 'nerf_table_bin_fill') and pipe is not None and pipe.binned_levels:
            # binned table update (csrc/table_update.hip): an item = one corner update = 2 B index + 4F B values;
            # the update pass moves 24 B per parameter (p, m, v read and written) and reads every item once; the bin
            # passes read d(features) once per level and write every item once
            cfg_ = model.encoder.cfg
            F_ = cfg_.n_features
            n_items = 8 * n_pts_scatter * len(pipe.binned_levels)
            n_binned_params = sum(int(cfg_.entries[l]) * F_ for l in pipe.binned_levels)
            if base == 'nerf_table_update_binned':
                bytes_step = 24 * n_binned_params + n_items * (2 + 4 * F_)
            else:
                bytes_step = n_items * (2 + 4 * F_) + n_pts_scatter * len(pipe.binned_levels) * 4 * F_
        elif base == 'nerf_hashgrid_encode_levels':
            bytes_step = algorithmic_bytes(base, args.rays, n_params, Sc=sum(STEPS), L=args.levels)   # both launches of one step together
        else:
            b = algorithmic_bytes(base, args.rays, n_params, Sc=sum(STEPS), L=args.levels)
            bytes_step = b * (launches / args.profile_steps) if b else None
        k = {'name': name, 'launches_per_step': launches / args.profile_steps, 'ms_per_step': round(per_step_ms, 4),
             'avg_launch_ms': round(total_ms / launches, 4)}
        if bytes_step:
            k['algorithmic_bytes_per_launch'] = bytes_step / (launches / args.profile_steps)
            k['algorithmic_GBps'] = round(bytes_step / (per_step_ms * 1e-3) / 1e9, 1)
            k['hbm_frac'] = round(k['algorithmic_GBps'] / HBM_PEAK_GBS, 4)
        fl = algorithmic_flops(base, args.rays, Sc=sum(STEPS))
        if fl:
            k['algorithmic_TFLOPs'] = round(fl / (per_step_ms * 1e-3) / 1e12, 2)
            k['mfma_f32_frac'] = round(k['algorithmic_TFLOPs'] / MFMA_F32_PEAK_TFLOPS, 4)
            k['mfma_busy_counter'] = None       # filled below from profiles/mfma.json when the run matches it
        kernels.append(k)
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    # (the PMC passes were taken at N = 1 on the default table and batch: per-launch traffic of any other run is unmeasured)
    default_workload = (args.rays == 4096 and args.table_log2 == 24 and args.levels == 20 and args.coarse == '64'
                        and args.table_precision == 'float32'
                        and os.environ.get('NERF_TABLE_BINNED', '1') != '0')
    pmc = json.load(open(tpath)) if os.path.exists(tpath) and world == 1 and default_workload else {}

    mpath = os.path.join(ROOT, 'profiles', 'mfma.json')
    mfma = json.load(open(mpath)) if os.path.exists(mpath) and world == 1 and default_workload else {}

    def mfma_counter(name):
        """Counter-based MFMA utilisation of the kernel behind an entry point, from the committed
        rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE pass (tools/mfma_summary.py)."""
        rec = mfma.get('per_kernel', {}).get(name.split('[')[0])
        if not rec:
            return None
        return dict(rec, source='committed PMC pass, not this run: profiles/mfma.json (build %s)' % mfma.get('build', '?'))

    def roofline_of(k):
        # HBM bytes per launch from the committed PMC passes (tools/pmc_summary.py), same launch class
        rec = pmc.get('per_launch', {}).get(k['name']) or pmc.get('per_launch', {}).get(k['name'].split(',')[0] + ']')
        if k['name'].split('[')[0] == 'nerf_field_bwd':
            # the MLP backward is bound by the matrix cores, not by its 336 B per point
            return {'kernel': k['name'], 'bound': 'mfma', 'achieved': k.get('algorithmic_TFLOPs'),
                    'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': k.get('mfma_f32_frac'),
                    'traffic': rec['hbm_bytes'] if rec else None,
                    'traffic_source': ('committed PMC passes, not this run: profiles/traffic.json (build %s)' % pmc.get('build', '?')) if rec else None,
                    'mfma_busy': mfma_counter(k['name']), 'avg_launch_ms': k['avg_launch_ms'],
                    'launches_per_step': k['launches_per_step'],
                    'algorithmic_flops_per_launch': algorithmic_flops('nerf_field_bwd', args.rays, Sc=sum(STEPS))
                    / max(1.0, k['launches_per_step']),
                    'note': ('algorithmic fp32 flops priced against the f32-input MFMA peak.  Two thirds of them (forward '
                             'recompute + back-propagation chains) run on v_mfma_f32_32x32x2_f32; the weight-gradient third '
                             'runs on v_mfma_f32_32x32x16_bf16 with split hi/lo operands (two MFMAs = four bf16 products per '
                             'fp32 product, 4x fewer matrix-core cycles than the f32-input form), so frac is not a busy '
                             'fraction: mfma_busy.util (counter pass) is, and it counts both kinds')}
        return {'kernel': k['name'], 'bound': 'hbm', 'achieved': k.get('algorithmic_GBps'), 'peak': HBM_PEAK_GBS,
                'unit': 'GB/s', 'frac': k.get('hbm_frac'), 'traffic': rec['hbm_bytes'] if rec else None,
                'traffic_source': (('committed PMC passes, not this run: profiles/traffic.json (build %s; rocprofv3 --pmc FETCH_SIZE '
                                    '/ WRITE_SIZE, per launch; FETCH_SIZE x2 for the wide p/m/v stream per MI355X_MICROARCH.md HBM)'
                                    % pmc.get('build', '?')) if rec else None),
                'avg_launch_ms': k['avg_launch_ms'], 'launches_per_step': k['launches_per_step'],
                'algorithmic_bytes_per_launch': k.get('algorithmic_bytes_per_launch')}
    for k in kernels:
        if 'mfma_busy_counter' in k:
            k['mfma_busy_counter'] = mfma_counter(k['name'])
    with_bytes = [k for k in kernels if k.get('algorithmic_GBps')]
    if pipe is not None and pipe.binned_levels and model_sharded is None:
        # binned table update: the LDS-privatised scatter of the small dense levels runs on its own stream BESIDE the
        # bin and update passes (its wall time is stretched by sharing the CUs and is not on the critical path), so
        # the dominant kernel is picked among the launches of the main stream
        # (so do the count + scan passes, issued from the forward pass: nerf_table_bin_count)
        with_bytes = [k for k in with_bytes if not k['name'].startswith(('nerf_hashgrid_scatter', 'nerf_table_bin_count'))]
    roofline = roofline_of(with_bytes[0]) if with_bytes else None
    if roofline and roofline['kernel'] == 'nerf_table_update_binned':
        roofline['note'] = ('scatter + Adam of the hashed and large dense levels in one pass: every bin of corner updates is '
                            'summed in LDS, then p, m, v stream through the Adam update (24 B per parameter)')
    if roofline and roofline['kernel'].startswith('nerf_hashgrid_scatter'):
        roofline['note'] = ('random 8-byte atomic adds: bounded by the memory-side atomic rate (one 64-B request per table '
                            'entry), not by bytes; the launches overlap the Adam pass of the previous level')
    # second: the kernel of the OTHER bound, the matrix-core-bound MLP backward with its counter-based MFMA utilisation
    # (else the second-largest class)
    bwd = [k for k in kernels if k['name'].split('[')[0] == 'nerf_field_bwd']
    roofline_2 = roofline_of(bwd[0]) if bwd else (roofline_of(with_bytes[1]) if len(with_bytes) > 1 else None)

    # ---- 800x800 inference (BASELINE configs[1]) ----
    render = None
    if not args.no_render:
        from nerf_amd.inference import Inferencer
        inf = Inferencer(renderer, 65536, 0, 0, 0.1, 0.05, 0.4, [1, 2, 3, 4], 20)
        n, h, w, K, E, _, _, _ = loader.get_image_batch(0)
        inf.render_image(n, h, w, K, E)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            inf.render_image(n, h, w, K, E)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / reps
        _beat()
        render = {'workload': '800x800 image, %d+128 samples/ray, chunks of 65536 rays, sharded over %d GPU(s)' % (sum(STEPS), world),
                  'rays_per_s': H * W / dt, 'ms_per_image': dt * 1e3}
        # the reference's own chunk size (inference.py:66 uses trainer.n_rays-sized chunks; SURVEY C2 lists 16 384 too)
        inf_small = Inferencer(renderer, 16384, 0, 0, 0.1, 0.05, 0.4, [1, 2, 3, 4], 20)
        inf_small.render_image(n, h, w, K, E)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        inf_small.render_image(n, h, w, K, E)
        torch.cuda.synchronize()
        render['rays_per_s_chunks_16384'] = H * W / (time.perf_counter() - t2)
        _beat()

    # the per-ray kernels at 65 536 rays: their bandwidth is invisible at 4096 rays (launch-latency-bound there)
    per_ray = per_ray_kernels_at(65536, Sc=sum(STEPS)) if (rank == 0 and world == 1 and not args.no_render) else None
    _beat()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        del model, renderer, loader, opt
        torch.cuda.empty_cache()
        cpu = cpu_baseline()

    if rank == 0:
        out = {
            'metric': 'rays/sec (train step + render) at 64+128 samples/ray', 'value': value, 'unit': 'rays/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': ('train step (render + backward + dense Adam), %d rays/GPU/step, %d coarse + 128 fine '
                                    'samples/ray, config_base network (L%d F2 T%d, SH4, latent 48, 64-wide MLPs), '
                                    'procedural scene 100 poses 800x800' % (args.rays, sum(STEPS), args.levels, args.table_log2))
                                   + ('' if args.coarse == '64' else ' [NOT the metric config: %s coarse samples]' % args.coarse)
                                   + ('' if args.table_precision == 'float32' else ' [NOT the metric config: half-precision table image]'),
                       'rays_per_gpu': args.rays, 'global_rays': args.rays * world,
                       'table_mode': ('single_gpu' if world == 1 else
                                      ('level_sharded' if getattr(model_sharded, 'world', 1) > 1 else 'replicated_fallback')),
                       'parallelism': ('dp%d over rays' % world) + (', hash table sharded by level (feature all-to-all)'
                                                                    if getattr(model_sharded, 'world', 1) > 1 else '') + fallback_note},
            'roofline': roofline, 'roofline_second': roofline_2, 'cpu_baseline': cpu, 'render': render,
            'per_ray_kernels_65536_rays': per_ray, 'kernels': kernels,
        }
    if world > 1:
        dist.barrier()       # every rank got here: only now may rank 0 report
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
