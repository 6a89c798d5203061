"""Weights: seeded generators in the reference's state_dict naming, and the flat blobs the C-ABI parses.

There are no trained checkpoints offline (the reference downloads them at import,
interface.py:29,78), so benchmarks and tests use seeded random weights.  The generators use
numpy's PCG64 so that the golden-fixture script (which loads them into the *reference* modules
with ``load_state_dict(strict=True)``) and the GPU tests reproduce identical tensors.

Checkpoint ingestion (reference format): ``load_checkpoint_state_dict`` accepts the
dict saved by balldetection/helper_balldetection.py:510-529 / uplifting/helper.py:371-391.
"""
import os
import struct
import numpy as np

from . import arch

WASB_MAGIC = b'TTUPWSB1'
UPLIFT_MAGIC = b'TTUPUPL1'


def _np(v):
    if hasattr(v, 'detach'):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32))


# --------------------------------------------------------------------------- generators
def random_wasb_state_dict(seed=0, planted=False, in_ch=9, head_out=3, eps=0.2, plant_all_heads=False):
    """Seeded WASB/HRNet weights.

    planted=False: Kaiming-scaled noise everywhere, BN running stats randomised so that BN
    folding is exercised (a fresh BatchNorm has mean 0 / var 1).
    planted=True: the same noise scaled by ``eps`` plus an identity path that carries the mean of
    the centre frame's three channels through channel 0 of the full-resolution trunk to head channel 1, so a
    bright blob in the frames produces a dominant heatmap peak (margin >> bf16 rounding) while
    every conv still contributes.  plant_all_heads=True routes the planted trunk channel to EVERY head channel (the 13-keypoint
    table detector of the end-to-end fixtures: all keypoints then follow the blob with the same margin).
    """
    rng = np.random.default_rng(seed)
    sd = {}
    gain = eps if planted else 1.0
    for s in arch.hrnet_convs(in_ch, head_out):
        fan_in = s.cin * s.k * s.k
        sd[s.conv + '.weight'] = (rng.standard_normal((s.cout, s.cin, s.k, s.k)) * np.sqrt(2.0 / fan_in) * gain).astype(np.float32)
        if s.has_bias:
            sd[s.conv + '.bias'] = (rng.standard_normal(s.cout) * 0.1 * gain).astype(np.float32)
        if s.bn:
            damp = 0.5 if s.bn.endswith('bn2') or s.bn.endswith('bn3') else 1.0
            sd[s.bn + '.weight'] = (rng.uniform(0.8, 1.2, s.cout) * damp).astype(np.float32)
            sd[s.bn + '.bias'] = (rng.standard_normal(s.cout) * 0.1 * gain).astype(np.float32)
            sd[s.bn + '.running_mean'] = (rng.standard_normal(s.cout) * 0.1 * gain).astype(np.float32)
            sd[s.bn + '.running_var'] = rng.uniform(0.5, 1.5, s.cout).astype(np.float32)
    if planted:
        def ident_bn(bn):
            sd[bn + '.weight'][0] = 1.0
            sd[bn + '.bias'][0] = 0.0
            sd[bn + '.running_mean'][0] = 0.0
            sd[bn + '.running_var'][0] = 1.0
        p = 'model'
        w = sd[p + '.conv1.weight']; w[0] = 0; w[0, in_ch // 3:2 * in_ch // 3, 1, 1] = 3.0 / in_ch; ident_bn(p + '.bn1')   # centre frame only
        w = sd[p + '.conv2.weight']; w[0] = 0; w[0, 0, 1, 1] = 1.0; ident_bn(p + '.bn2')
        w = sd[p + '.layer1.0.downsample.0.weight']; w[0] = 0; w[0, 0, 0, 0] = 1.0; ident_bn(p + '.layer1.0.downsample.1')
        w = sd[p + '.transition1.0.0.weight']; w[0] = 0; w[0, 0, 1, 1] = 1.0; ident_bn(p + '.transition1.0.1')
        w = sd[p + '.final_layers.0.weight']
        for k in (range(head_out) if plant_all_heads else [1]):
            w[k] *= 0.25; w[k, 0, 0, 0] = 1.0
    return sd


def random_uplift_state_dict(seed=0, size='large'):
    """Seeded uplift-transformer weights (xavier-like scale as model.py:22-28,117-121,178-184,243-249;
    biases and LayerNorm randomised so that every term is exercised)."""
    rng = np.random.default_rng(seed)
    d, depth, heads = arch.UPLIFT_SIZES[size]
    sd = {}
    for k, shape in arch.uplift_schema(size):
        if k.endswith('inv_freq'):
            hd = d // heads
            sd[k] = (1.0 / (10000 ** (np.arange(0, hd, 2, dtype=np.float32) / np.float32(hd)))).astype(np.float32)
        elif k.endswith('norm1.weight') or k.endswith('norm2.weight'):
            sd[k] = rng.uniform(0.8, 1.2, shape).astype(np.float32)
        elif k.endswith('.bias'):
            sd[k] = (rng.standard_normal(shape) * 0.05).astype(np.float32)
        elif k == 'cls_token':
            sd[k] = (rng.standard_normal(shape) * 0.1).astype(np.float32)
        else:
            fan_out, fan_in = shape
            a = np.sqrt(6.0 / (fan_in + fan_out))
            sd[k] = rng.uniform(-a, a, shape).astype(np.float32)
    return sd


# --------------------------------------------------------------------------- blobs
def pack_wasb_blob(state_dict, in_ch=9, head_out=3, prefix='model'):
    """Reference-format state_dict -> bytes for ``ttup_wasb_create`` (layout: include/ttup.h)."""
    convs = arch.hrnet_convs(in_ch, head_out, prefix)
    parts = [WASB_MAGIC, struct.pack('<4i', len(convs), in_ch, head_out, 0)]
    for s in convs:
        w = _np(state_dict[s.conv + '.weight'])
        if w.shape != (s.cout, s.cin, s.k, s.k):
            raise ValueError('%s: expected shape %s, got %s' % (s.conv, (s.cout, s.cin, s.k, s.k), w.shape))
        parts.append(struct.pack('<8i', s.cout, s.cin, s.k, s.stride, 1 if s.bn else 0, 1 if s.has_bias else 0, 0, 0))
        parts.append(w.tobytes())
        if s.has_bias:
            parts.append(_np(state_dict[s.conv + '.bias']).tobytes())
        if s.bn:
            for f in ('weight', 'bias', 'running_mean', 'running_var'):
                v = _np(state_dict['%s.%s' % (s.bn, f)])
                if v.shape != (s.cout,):
                    raise ValueError('%s.%s: bad shape %s' % (s.bn, f, v.shape))
                parts.append(v.tobytes())
    return b''.join(parts)


def pack_uplift_blob(state_dict, size='large'):
    """Reference-format state_dict -> bytes for ``ttup_uplift_create``."""
    d, depth, heads = arch.UPLIFT_SIZES[size]
    pos, first, second = arch.uplift_layers(size)
    parts = [UPLIFT_MAGIC, struct.pack('<8i', d, heads, len(pos), len(first), len(second), 13, 0, 0)]
    inv = _np(state_dict[(pos + first + second)[0] + '.attn.rotary_emb.inv_freq'])   # identical in every layer (model.py:51)
    parts += [struct.pack('<i', inv.size), inv.tobytes()]
    for k, shape in arch.uplift_schema(size):
        if k.endswith('inv_freq') or k.startswith('embed.'):
            continue
        v = _np(state_dict[k])
        if v.shape != tuple(shape):
            raise ValueError('%s: expected shape %s, got %s' % (k, shape, v.shape))
        parts.append(struct.pack('<i', v.size))
        parts.append(v.tobytes())
    return b''.join(parts)


def load_checkpoint_state_dict(path):
    """Read a reference checkpoint file (torch.save of {'model_state_dict', 'identifier',
    'additional_info'}) -> (state_dict, additional_info).

    A checkpoint folder is user input, so the file is read with ``weights_only=True`` (tensors + plain containers: the default of
    the torch 2.6 the reference pins, and what the reference's detector loaders use, inference_balldetection.py:49).  The
    reference reads the UPLIFTING checkpoint with ``weights_only=False`` (inference_uplifting.py:43) because its
    ``additional_info`` carries the training hyper-parameters, which may hold numpy scalars / dtypes: those are allow-listed on a
    second attempt.  Anything beyond that needs the explicit opt-in ``TTUP_UNSAFE_LOAD=1`` (full unpickling, trusted files only)."""
    import pickle
    import torch
    if os.environ.get('TTUP_UNSAFE_LOAD') == '1':
        d = torch.load(path, map_location='cpu', weights_only=False)
        return d['model_state_dict'], d.get('additional_info', {})
    try:
        d = torch.load(path, map_location='cpu', weights_only=True)
    except pickle.UnpicklingError as first:
        import numpy as np
        allow = [np.dtype, np.ndarray, type(np.dtype('float32')), type(np.dtype('float64')), type(np.dtype('int64')), type(np.dtype('int32')), type(np.dtype('bool'))]
        try:
            from numpy._core.multiarray import scalar, _reconstruct
        except ImportError:                                   # numpy < 2
            from numpy.core.multiarray import scalar, _reconstruct
        allow += [scalar, _reconstruct]
        try:
            with torch.serialization.safe_globals(allow):
                d = torch.load(path, map_location='cpu', weights_only=True)
        except pickle.UnpicklingError:
            raise RuntimeError('checkpoint %s holds pickled objects beyond tensors, containers and numpy scalars (%s); if the file is '
                               'trusted, set TTUP_UNSAFE_LOAD=1 to read it like the reference does (weights_only=False)' % (path, first)) from first
    return d['model_state_dict'], d.get('additional_info', {})
