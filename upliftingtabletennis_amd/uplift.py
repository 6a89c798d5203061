"""a6/a7: drop-in for the uplift network behind ``self.model(ball, table, mask, times) -> (rot, pos)``
(interface.py:235, inference/utils.py:254) and for ``transform_rotationaxes`` (uplifting/helper.py:394-420).
Reference model: uplifting/model.py:502-571 built by get_model('connectstage', size, 'dynamic', 'new')."""
import ctypes

import torch

from . import _lib, arch, weights


class MultiStageModel:
    def __init__(self, state_dict, size='large', max_batch=64, max_len=128, device='cuda:0'):
        _lib.require_gpu()
        self.device = torch.device(device)
        self.size = size
        self.dim, self.depth, self.heads = arch.UPLIFT_SIZES[size]
        self.max_batch, self.max_len = int(max_batch), int(max_len)
        self.time_rotation = 'new'
        self._lib = _lib.load()
        blob = weights.pack_uplift_blob(state_dict, size)
        self._handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.ttup_uplift_create(blob, len(blob), self.max_batch, self.max_len, ctypes.byref(self._handle)))

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def __del__(self):
        h, self._handle = getattr(self, '_handle', None), None
        if h:
            self._lib.ttup_uplift_destroy(h)

    def forward(self, ball_pos, table_pos, mask, times, check_mask=True):
        """ball (B,T,2), table (B,13,3), mask (B,T) in {0,1} with at least one 0, times (B,T) -> rot (B,3), pos (B,T,3).
        Raises ValueError for a mask that is not {0,1} with both values present (model.py:541-546)."""
        args = [t.to(self.device, torch.float32).contiguous() for t in (ball_pos, table_pos, mask, times)]
        ball, table, mask, times = args
        b, t, _ = ball.shape
        if table.shape != (b, 13, 3) or mask.shape != (b, t) or times.shape != (b, t):
            raise ValueError('inconsistent input shapes')
        if b == 0 or t == 0:      # the reference fails on mask.min() of an empty tensor (model.py:541)
            raise ValueError('empty batch: the uplift model needs at least one trajectory with one time step')
        rot = torch.empty((b, 3), dtype=torch.float32, device=self.device)
        pos = torch.empty((b, t, 3), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self._lib.ttup_uplift_forward(self._handle, _lib.ptr(ball), _lib.ptr(table), _lib.ptr(mask), _lib.ptr(times), b, t,
                                               _lib.ptr(rot), _lib.ptr(pos), 1 if check_mask else 0, _lib.stream_ptr())
        _lib.check(rc)
        return rot, pos

    __call__ = forward

    def graph_info(self):
        """{'graphs', 'off', 'replays', 'stage_launches'}: the small-batch path (hipGraph replay of a captured forward; all layers
        of a stage in one stage_x3_kernel launch, csrc/uplift.hip)."""
        out = (ctypes.c_int * 3)()
        _lib.check(self._lib.ttup_uplift_graph_info(self._handle, out))
        st = ctypes.c_longlong(0)
        _lib.check(self._lib.ttup_uplift_stage_info(self._handle, ctypes.byref(st)))
        return {'graphs': int(out[0]), 'off': bool(out[1]), 'replays': int(out[2]), 'stage_launches': int(st.value)}


def get_model(name='connectstage', size='large', mode='dynamic', time_rotation='new', state_dict=None, **kw):
    """Mirror of uplifting/model.py:574-603 for the shipped configuration."""
    assert time_rotation in ['old', 'new'], 'time_rotation should be either "old" or "new"'
    if name != 'connectstage' or mode != 'dynamic' or time_rotation != 'new':
        raise ValueError('only the shipped configuration connectstage/dynamic/new is built (see DESIGN.md)')
    if size not in arch.UPLIFT_SIZES:
        raise ValueError(f'Unknown model size {size}')
    if state_dict is None:
        raise ValueError('a state_dict is required (no weights can be downloaded offline)')
    return MultiStageModel(state_dict, size=size, **kw)


def transform_rotationaxes(rotation, r_gt):
    """uplifting/helper.py:394-420: spin from the global frame into the ball-local frame.  (B,3),(B,T,3) or (3,),(T,3)."""
    _lib.require_gpu()
    lib = _lib.load()
    single = rotation.dim() == 1
    if r_gt.dim() not in (2, 3):
        raise ValueError('Shape not supported.')
    rot = (rotation[None] if single else rotation).to(torch.float32).contiguous()
    pos = (r_gt[None] if single else r_gt).to(torch.float32).contiguous()
    if not rot.is_cuda:
        rot, pos = rot.cuda(), pos.cuda()
    out = torch.empty_like(rot)
    with torch.cuda.device(rot.device):
        _lib.check(lib.ttup_transform_rotationaxes(_lib.ptr(rot), _lib.ptr(pos), rot.shape[0], pos.shape[1], _lib.ptr(out), _lib.stream_ptr()))
    return out[0] if single else out
