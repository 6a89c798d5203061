"""ctypes binding of libttup.so (include/ttup.h).  Fails loudly when the HIP library is missing: there is
no CPU fallback anywhere in this package."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TTUP_LIB', os.path.join(_HERE, 'libttup.so'))     # TTUP_LIB: alternative build (ablation experiments)

OK, EINVAL, EFORMAT, EHIP, ENOMEM, EMASK = 0, 1, 2, 3, 4, 5
DTYPE_BF16, DTYPE_F32 = 0, 1
REFINE_BALL, REFINE_TABLE = 0, 1

_c = ctypes
_vp, _i, _sz = _c.c_void_p, _c.c_int, _c.c_size_t

# name -> (restype, argtypes): every symbol include/ttup.h declares
SIGNATURES = {
    'ttup_version': (_i, []),
    'ttup_build_id': (_c.c_char_p, []),
    'ttup_last_error': (_c.c_char_p, []),
    'ttup_device_count': (_i, []),
    'ttup_preprocess_triples': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'ttup_wasb_create': (_i, [_vp, _sz, _i, _i, _i, _i, _c.POINTER(_vp)]),
    'ttup_wasb_streams': (_i, [_vp, _c.POINTER(_vp), _i, _c.POINTER(_i)]),
    'ttup_wasb_create_ex': (_i, [_vp, _sz, _i, _i, _i, _i, _i, _i, _c.POINTER(_vp)]),
    'ttup_wasb_destroy': (None, [_vp]),
    'ttup_wasb_forward': (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp]),
    'ttup_wasb_forward_frames': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'ttup_wasb_read_tap': (_i, [_vp, _c.c_char_p, _i, _vp, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i), _vp]),
    'ttup_wasb_time_ops': (_i, [_vp, _i, _i, _i, _vp, _vp, _c.POINTER(_i), _vp]),
    'ttup_wasb_time_graph': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _c.POINTER(_i), _vp]),
    'ttup_wasb_time_replay': (_i, [_vp, _i, _i, _vp, _vp]),
    'ttup_peak_mfma_bf16': (_i, [_i, _i, _vp, _vp]),
    'ttup_peak_hbm': (_i, [_sz, _vp, _vp]),
    'ttup_wasb_set_certify': (_i, [_vp, _c.c_float, _i, _i]),
    'ttup_certify_scan': (_i, [_vp, _vp, _i, _i, _i, _c.c_float, _i, _vp, _vp, _vp, _vp]),
    'ttup_max_abs_diff': (_i, [_vp, _vp, _c.c_longlong, _vp, _vp]),
    'ttup_max_abs_diff_cols': (_i, [_vp, _vp, _c.c_longlong, _c.c_longlong, _c.c_longlong, _c.c_longlong, _vp, _i, _vp]),
    'ttup_slice_columns': (_i, [_vp, _c.c_longlong, _i, _i, _i, _vp, _vp]),
    'ttup_wasb_certify_budget': (_i, [_vp, _i]),
    'ttup_wasb_certify_exact_windows': (_i, [_vp, _i]),
    'ttup_wasb_certify_audit_crops': (_i, [_vp, _i, _i]),
    'ttup_wasb_certify_info': (_i, [_vp, _vp, _vp]),
    'ttup_wasb_certify_status': (_i, [_vp, _i, _vp, _vp]),
    'ttup_wasb_certify_flags': (_i, [_vp, _i, _vp, _vp]),
    'ttup_wasb_certify_margins': (_i, [_vp, _i, _vp, _vp]),
    'ttup_wasb_certify_stats': (_i, [_vp, _vp, _i]),
    'ttup_wasb_set_priority': (_i, [_vp, _i]),
    'ttup_wasb_micro_batch': (_i, [_vp]),
    'ttup_wasb_out_channels': (_i, [_vp]),
    'ttup_preprocess_frames': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'ttup_refine_workspace_bytes': (_sz, [_i, _i, _i]),
    'ttup_refine': (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    'ttup_refine_windows': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'ttup_uplift_create': (_i, [_vp, _sz, _i, _i, _c.POINTER(_vp)]),
    'ttup_uplift_destroy': (None, [_vp]),
    'ttup_uplift_graph_info': (_i, [_vp, _vp]),
    'ttup_uplift_stage_info': (_i, [_vp, _vp]),
    'ttup_uplift_forward': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp]),
    'ttup_transform_rotationaxes': (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    'ttup_trajgen_max_samples': (_i, []),
    'ttup_trajgen_workspace_bytes': (_sz, [_i]),
    'ttup_trajgen_simulate': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'ttup_calib_forward': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'ttup_odefit_forward': (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _c.c_double, _i, _c.c_double, _vp, _vp, _vp, _vp, _vp]),
    'ttup_odefit_integrate': (_i, [_vp, _vp, _vp, _i, _i, _i, _c.c_double, _vp, _vp, _vp]),
    'ttup_trajgen_select': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
}

_lib = None


def load():
    """Load libttup.so (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libttup.so not found at %s: build it with `python -m upliftingtabletennis_amd.build` '
                               '(hipcc, gfx950).  There is no CPU fallback.' % LIB_PATH)
        # torch first: its wheel bundles the HIP runtime the process must share (one libamdhip64 per process); loading
        # libttup.so before torch would pull in /opt/rocm's copy and leave the two runtimes with separate device state
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _check_build_id(lib)
        _lib = lib
    return _lib


def build_id():
    """The source hash compiled into the loaded library (ttup_build_id)."""
    return load().ttup_build_id().decode()


def _check_build_id(lib):
    """The library must have been built from THIS tree's sources (a stale libttup.so would otherwise pass for HEAD's
    kernels).  TTUP_LIB (ablation builds of a modified copy) and TTUP_ALLOW_STALE_LIB=1 skip the check."""
    if 'TTUP_LIB' in os.environ or os.environ.get('TTUP_ALLOW_STALE_LIB') == '1':
        return
    from . import build
    try:
        want = build.source_id()
    except OSError as e:
        # a deployed copy without the source tree (the package + libttup.so, no csrc/ or include/): nothing to compare with
        import warnings
        warnings.warn('libttup.so: the source tree is not next to the package (%s), so the library\'s build id (%s) cannot be checked against it; '
                      'set TTUP_ALLOW_STALE_LIB=1 to silence this' % (e, lib.ttup_build_id().decode()))
        return
    have = lib.ttup_build_id().decode()
    if have != want:
        raise RuntimeError('libttup.so was built from other sources (library %s, tree %s): rebuild it with '
                           '`python -m upliftingtabletennis_amd.build`' % (have, want))


def check(rc):
    """Map a TTUP_E* code to the exception type the reference raises at the same place."""
    if rc == OK:
        return
    msg = load().ttup_last_error().decode('utf-8', 'replace')
    if rc in (EINVAL, EFORMAT, EMASK):
        raise ValueError(msg)
    if rc == ENOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError('upliftingtabletennis_amd needs a HIP device (MI355X); torch.cuda.is_available() is False '
                           'and there is no CPU fallback')
    load()


def ptr(t):
    """data_ptr of a torch tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
