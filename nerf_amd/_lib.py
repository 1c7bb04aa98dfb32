"""ctypes binding of libnerf_hip.so (include/nerf_hip.h).  No torch types cross the ABI:
tensors are passed as raw device pointers plus sizes.  The library is REQUIRED: every product
op raises if it is missing -- there is no CPU or eager-PyTorch fallback."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (NERF_HIP_LIB: another build of the same library, for A/B timing of kernel variants on one box)
LIB_PATH = os.environ.get('NERF_HIP_LIB') or os.path.join(_HERE, 'lib', 'libnerf_hip.so')

NERF_MAX_LEVELS = 32


class NerfGridCfg(C.Structure):
    _fields_ = [('n_levels', C.c_int32), ('n_features', C.c_int32), ('log2_hashmap_size', C.c_int32),
                ('base_resolution', C.c_int32), ('scale', C.c_float * NERF_MAX_LEVELS),
                ('res', C.c_uint32 * NERF_MAX_LEVELS), ('entries', C.c_uint32 * NERF_MAX_LEVELS),
                ('hashed', C.c_uint32 * NERF_MAX_LEVELS), ('offset', C.c_uint64 * NERF_MAX_LEVELS),
                ('total_entries', C.c_uint64)]


class NerfNetCfg(C.Structure):
    _fields_ = [('hidden_dim', C.c_int32), ('sigma_layers', C.c_int32), ('geo_feat_dim', C.c_int32),
                ('color_layers', C.c_int32), ('sh_degree', C.c_int32), ('latent_dim', C.c_int32)]


class NerfPointSrc(C.Structure):
    _fields_ = [('mode', C.c_int32), ('n_samples', C.c_int32), ('n_points', C.c_int64), ('x', C.c_void_p),
                ('d', C.c_void_p), ('n', C.c_void_p), ('rays_o', C.c_void_p), ('rays_d', C.c_void_p),
                ('z', C.c_void_p), ('z_stride', C.c_int64), ('inner_bound', C.c_float), ('outer_bound', C.c_float),
                ('ray_block', C.c_int32), ('reserved_', C.c_int32)]


class NerfLevelSet(C.Structure):
    _fields_ = [('count', C.c_int32), ('level', C.c_int32 * NERF_MAX_LEVELS)]


NERF_MAX_FEATURE_COLS = 128


class NerfColMap(C.Structure):
    _fields_ = [('offset', C.c_int64 * NERF_MAX_FEATURE_COLS), ('stride', C.c_int32 * NERF_MAX_FEATURE_COLS)]


_P, _I64, _I, _F = C.c_void_p, C.c_int64, C.c_int, C.c_float
_G, _N, _S = C.POINTER(NerfGridCfg), C.POINTER(NerfNetCfg), C.POINTER(NerfPointSrc)
_LS = C.POINTER(NerfLevelSet)

# every symbol include/nerf_hip.h declares, with its argument types
SIGNATURES = {
    'nerf_abi_version': (C.c_int, []),
    'nerf_last_error': (C.c_char_p, []),
    'nerf_grid_cfg_init': (_I64, [_G, _I, _I, _I, _I, _F]),
    'nerf_sigma_param_count': (_I64, [_G, _N]),
    'nerf_color_param_count': (_I64, [_N]),
    'nerf_sample_batch': (_I, [_P, _P, _P, _P, _I64, _I, _I, _I, _I64, C.c_uint64, C.c_uint64, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'nerf_raygen': (_I, [_P, _P, _P, _P, _I64, _P, _P, _P]),
    'nerf_sample_points': (_I, [_P, _P, _P, _I64, _I64, _I, _F, _F, _P, _P, _P, _P]),
    'nerf_contract': (_I, [_P, _I64, _F, _F, _P, _P]),
    'nerf_hashgrid_fwd': (_I, [_G, _P, _P, _I64, _P, _P]),
    'nerf_hashgrid_bwd': (_I, [_G, _P, _P, _I64, _P, _P]),
    'nerf_sh_fwd': (_I, [_P, _I64, _I, _P, _P]),
    'nerf_mlp_fwd': (_I, [_P, C.POINTER(C.c_int32), _I, _P, _I64, _P, _P]),
    'nerf_mlp_bwd_workspace_floats': (_I64, [C.POINTER(C.c_int32), _I, _I64]),
    'nerf_mlp_bwd': (_I, [_P, C.POINTER(C.c_int32), _I, _P, _P, _I64, _P, _P, _P, _I64, _P]),
    'nerf_mlp_bwd_noalloc': (_I, [_P, C.POINTER(C.c_int32), _I, _P, _P, _I64, _P, _P, _P]),
    'nerf_density_fwd': (_I, [_G, _N, _P, _P, _S, _P, _P, _P]),
    'nerf_field_fwd': (_I, [_G, _N, _P, _P, _P, _P, _I64, _S, _P, _P, _P, _P, _P]),
    'nerf_field_bwd': (_I, [_G, _N, _P, _P, _P, _P, _I64, _S, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'nerf_density_bwd_input': (_I, [_G, _N, _P, _P, _S, _P, _P, _P]),
    'nerf_hashgrid_scatter': (_I, [_G, _S, _P, _P, _I, _I, _P]),
    'nerf_hashgrid_encode_levels': (_I, [_G, _P, _S, _LS, _P, _I64, _P]),
    'nerf_hashgrid_scatter_levels': (_I, [_G, _S, _LS, _P, _I64, _P, _P]),
    'nerf_density_from_features': (_I, [_G, _N, _P, _P, _I64, _P, _P]),
    'nerf_field_from_features': (_I, [_G, _N, _P, _P, _P, _I64, _S, _P, _P, _P, _P, _P]),
    'nerf_feature_shuffle': (_I, [_P, _P, _I64, _I, C.POINTER(NerfColMap), _I, _P]),
    'nerf_table_to_half': (_I, [_P, _P, _I64, _P]),
    'nerf_hashgrid_fwd_f16': (_I, [_G, _P, _P, _I64, _P, _P]),
    'nerf_density_fwd_f16': (_I, [_G, _N, _P, _P, _S, _P, _P, _P]),
    'nerf_field_fwd_f16': (_I, [_G, _N, _P, _P, _P, _P, _I64, _S, _P, _P, _P, _P, _P]),
    'nerf_density_bwd_input_f16': (_I, [_G, _N, _P, _P, _S, _P, _P, _P]),
    'nerf_hashgrid_encode_levels_f16': (_I, [_G, _P, _S, _LS, _P, _I64, _P]),
    'nerf_coarse_weights': (_I, [_P, _P, _I64, _I64, _I, _P, _P]),
    'nerf_reweight': (_I, [_P, _I64, _I, _F, C.POINTER(C.c_int32), _I, _P, _P]),
    'nerf_sample_pdf': (_I, [_P, _I64, _P, _P, _P, _I64, _I, _I, _P, _P, _P, _P]),
    'nerf_coarse_to_fine': (_I, [_P, _P, _I64, _P, _P, _I64, _I, _I, _F, C.POINTER(C.c_int32), _I, _P, _P, _P]),
    'nerf_coarse_to_fine_rng': (_I, [_P, _P, _I64, C.c_uint64, C.c_uint64, _P, _I64, _I, _I, _F, C.POINTER(C.c_int32), _I, _P, _P,
                                     _P]),
    'nerf_coarse_to_fine_rng_at': (_I, [_P, _P, _I64, C.c_uint64, C.c_uint64, _I64, _P, _I64, _I, _I, _F, C.POINTER(C.c_int32), _I, _P, _P,
                                     _P]),
    'nerf_uniform_noise': (_I, [C.c_uint64, C.c_uint64, _I64, _P, _P]),
    'nerf_composite_fwd': (_I, [_P, _P, _P, _P, _P, _I64, _I, _F, _F, _P, _P, _P, _P, _P]),
    'nerf_composite_bwd': (_I, [_P, _P, _P, _P, _P, _P, _I64, _I, _P, _P, _P]),
    'nerf_losses': (_I, [_P, _P, _P, _P, _I64, _I, _F, _F, _F, _I, _P, _P, _P, _P]),
    'nerf_composite_loss_fwd': (_I, [_P, _P, _P, _P, _P, _P, _I64, _I, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P, _P]),
    'nerf_composite_loss_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _I64, _I, _F, _F, _F, _F, _P, _P, _P, _P]),
    'nerf_adam_step': (_I, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _I64, _F, _I, _P]),
    'nerf_adam_step_guarded': (_I, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _I64, _F, _I, _P, _P]),
    'nerf_table_update_binned_guarded': (_I, [_G, _LS, _I64, _P, _I64, _P, _P, _P, _F, _F, _F, _F, _F, _I64, _F, _P, _P]),
    'nerf_adam_step_f16': (_I, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _I64, _F, _I, _P, _P, _P]),
    'nerf_table_update_binned_f16': (_I, [_G, _LS, _I64, _P, _I64, _P, _P, _P, _F, _F, _F, _F, _F, _I64, _F, _P, _P, _P]),
    'nerf_table_bin_workspace_bytes': (_I64, [_G, _LS, _I64]),
    'nerf_table_bin_levels': (_I, [_G, _S, _LS, _P, _I64, C.POINTER(C.c_int32), _P, _I64, _P]),
    'nerf_table_bin_count': (_I, [_G, _S, _LS, _P, _I64, _P]),
    'nerf_table_bin_fill': (_I, [_G, _S, _LS, _P, _I64, C.POINTER(C.c_int32), _P, _I64, _P]),
    'nerf_table_bin_fill_range': (_I, [_G, _S, _LS, _P, _I64, C.POINTER(C.c_int32), _P, _I64, _I64, _I64, _P]),
    'nerf_table_update_binned': (_I, [_G, _LS, _I64, _P, _I64, _P, _P, _P, _F, _F, _F, _F, _F, _I64, _F, _P]),
    'nerf_cum_weights_thresh': (_I, [_P, _I64, _I, _F, _P, _P]),
}

_lib = None
_prof = None   # name -> list of (start_event, end_event) while kernel timing is on


class _Timed(object):
    """Wraps one entry point; when kernel timing is on, brackets the call with events recorded on
    the stream the kernel is launched on (torch's current stream, which is what `stream()` passes)."""

    def __init__(self, name, fn):
        # (the guarded optimizer entry points are timed under the name of the plain ones: same kernels)
        self.name, self.fn = name.replace('_guarded', ''), fn

    def __call__(self, *args):
        if _prof is None:
            return self.fn(*args)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = self.fn(*args)
        b.record()
        _prof.setdefault(self.name + _launch_class(self.name, args), []).append((a, b))
        return rc


def _launch_class(name, args):
    """Entry points whose launches differ a lot in size are timed per size class, so that one class is one
    kernel at one problem size (what a roofline figure needs)."""
    if name == 'nerf_hashgrid_scatter':
        grid, l0, cnt = args[0]._obj, args[4], args[5]
        if cnt == 1:
            return '[1 hashed level]' if grid.hashed[l0] else '[1 dense level]'
        return '[%d dense levels]' % cnt
    if name == 'nerf_hashgrid_scatter_levels':
        grid, ls = args[0]._obj, args[2]._obj
        if ls.count == 1:
            return '[1 hashed level]' if grid.hashed[ls.level[0]] else '[1 dense level]'
        return '[%d levels]' % ls.count
    if name == 'nerf_adam_step':
        n = args[4]
        return '[table slice, %.1f M floats]' % (n / 1e6) if n >= (1 << 20) else '[small tensor]'
    return ''


def kernel_timing(on):
    """Start (True) or stop (False) per-entry-point device timing.  Stopping returns
    {name: (launches, total_ms)} after synchronising."""
    global _prof
    if on:
        _prof = {}
        return None
    rec, _prof = _prof, None
    torch.cuda.synchronize()
    return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in (rec or {}).items()}


class _Lib(object):
    pass


def lib():
    """Load libnerf_hip.so (once).  Raises if it has not been built: python __graft_entry__.py build."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libnerf_hip.so is missing (%s). Build it with `make -C nerf_amd/csrc`; '
                               'nerf_amd has no fallback path.' % LIB_PATH)
        dll = C.CDLL(LIB_PATH)
        l = _Lib()
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(dll, name)
            fn.restype = res
            fn.argtypes = args
            launches = res is _I and name not in ('nerf_abi_version',)
            setattr(l, name, _Timed(name, fn) if launches else fn)
        if l.nerf_abi_version() != 4:
            raise RuntimeError('libnerf_hip.so ABI version mismatch')
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().nerf_last_error().decode()
        if rc == -2:
            raise NotImplementedError('%s: %s' % (what, msg))
        raise RuntimeError('%s failed (%d): %s' % (what, rc, msg))


def ptr(t):
    """Device pointer of a contiguous CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('nerf_amd kernels need tensors on the GPU (got %s); there is no CPU path' % t.device)
    if not t.is_contiguous():
        raise RuntimeError('nerf_amd kernels need contiguous tensors')
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def f32(t):
    return t.contiguous() if t.dtype == torch.float32 else t.to(torch.float32).contiguous()


def i64(t):
    return t.contiguous() if t.dtype == torch.int64 else t.to(torch.int64).contiguous()
