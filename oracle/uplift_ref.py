"""Oracle (a6/a7): 2D->3D uplift transformer forward + spin frame change, fp32 torch-CPU.

Functional restatement of ``uplifting/model.py`` for the shipped configuration
``get_model('connectstage', size, mode='dynamic', time_rotation='new')`` (model.py:588-595,
training defaults uplifting/train.py:17-21) working on a reference-format ``state_dict``.

Reference map:
  mask conversion / ValueError         model.py:541-546
  FirstStage.forward                    model.py:335-390
  SimpleStaticLayer.forward             model.py:278-300
  attention (+RoPE, additive mask)      model.py:186-229 ; proj has NO bias (model.py:268 passes
                                        attn_drop_rate into the proj_bias slot of :162)
  RotaryPositionalEmbedding.forward     model.py:56-102  (pos = round(t*500), interleaved pairs)
  embeddings / head                     model.py:105-158, :232-261
  second stage + cls token              model.py:551-571
  transform_rotationaxes                uplifting/helper.py:394-420
"""
import math
import torch
import torch.nn.functional as F

MAX_FPS = 500            # uplifting/helper.py:27
KEYPOINT_VISIBLE = 1     # tabledetection/helper_tabledetection.py:37

SIZES = {'small': (32, 8, 4), 'base': (64, 12, 4), 'large': (128, 16, 4), 'huge': (192, 16, 8)}  # model.py:590-597


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.as_tensor(v)


def _linear(x, sd, p):
    b = p + '.bias'
    return F.linear(x, _t(sd, p + '.weight'), _t(sd, b) if b in sd else None)


def _mlp2(x, sd, p):
    """Linear-ReLU-Linear (BallEmbedding/TableEmbedding :151-158, Mlp :30-36 with act=ReLU)."""
    return _linear(F.relu(_linear(x, sd, p + '.fc1')), sd, p + '.fc2')


def _head(x, sd, p):
    """MyHead.forward model.py:251-261."""
    return _linear(F.relu(_linear(F.relu(_linear(x, sd, p + '.fc1')), sd, p + '.fc2')), sd, p + '.fc3')


def rope(x, times, head_dim):
    """model.py:56-102, time_rotation='new'.  x (B,h,T,D), times (B,T)."""
    inv_freq = 1.0 / (10000 ** (torch.arange(0, head_dim, 2).float() / head_dim))
    pos = torch.round(times / (1 / MAX_FPS))
    freqs = torch.einsum('bi,j->bij', pos, inv_freq).unsqueeze(1)
    cos, sin = torch.cos(freqs), torch.sin(freqs)
    a, b = x[..., 0::2], x[..., 1::2]
    out = torch.zeros_like(x)
    out[..., 0::2] = a * cos - b * sin
    out[..., 1::2] = a * sin + b * cos
    return out


def attention(x, sd, p, mask, times, num_cls, heads):
    """model.py:186-229."""
    B, N, C = x.shape
    qkv = _linear(x, sd, p + '.qkv').reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    if num_cls > 0:
        cq, q = q[:, :, :num_cls], q[:, :, num_cls:]
        ck, k = k[:, :, :num_cls], k[:, :, num_cls:]
    q, k = rope(q, times, C // heads), rope(k, times, C // heads)
    if num_cls > 0:
        q, k = torch.cat((cq, q), 2), torch.cat((ck, k), 2)
    add = mask[:, None, None, :] + mask[:, None, :, None]
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=add, dropout_p=0.0, is_causal=False)
    o = o.transpose(1, 2).reshape(B, N, C)
    return _linear(o, sd, p + '.proj')       # state_dict holds no proj.bias


def layer(x, sd, p, mask, times, num_cls, heads):
    """SimpleStaticLayer.forward model.py:278-300."""
    D = x.shape[-1]
    h = F.layer_norm(x, (D,), _t(sd, p + '.norm1.weight'), _t(sd, p + '.norm1.bias'))
    x = attention(h, sd, p + '.attn', mask, times, num_cls, heads) + x
    h = F.layer_norm(x, (D,), _t(sd, p + '.norm2.weight'), _t(sd, p + '.norm2.bias'))
    return _mlp2(h, sd, p + '.mlp1') + x


def _count(sd, prefix):
    n = 0
    while ('%s.%d.norm1.weight' % (prefix, n)) in sd:
        n += 1
    return n


def convert_mask(mask):
    """model.py:541-546."""
    if mask.min() == 0 and mask.max() == 1:
        return torch.where(mask == 0, torch.tensor(float('-inf')), torch.tensor(0.0))
    if mask.max() == 0 and mask.min() < -1e8:
        return mask
    raise ValueError('wrong format for masks. Should be 0, 1 or -1e9, 0.')


def uplift_forward(ball, table, mask, times, sd, heads=4):
    """MultiStageModel.forward (use_skipconnection=True, mode='dynamic'), model.py:529-571.
    ball (B,T,2), table (B,13,3), mask (B,T) in {0,1}, times (B,T) -> rot (B,3), pos (B,T,3)."""
    with torch.no_grad():
        ball, table, mask, times = [torch.as_tensor(a, dtype=torch.float32) for a in (ball, table, mask, times)]
        B, T, _ = ball.shape
        mask = convert_mask(mask)
        # ---- FirstStage.forward :335-390
        x = _mlp2(ball, sd, 'firststage.ball_embed')                       # (B,T,D)
        D = x.shape[-1]
        vis = table[:, :, 2]
        tmask = torch.where(vis == KEYPOINT_VISIBLE, 0.0, float('-inf'))
        tmask = torch.cat((torch.zeros((B, 1)), tmask), 1)                  # (B,14)
        tmask = tmask[:, None, :].expand(B, T, -1).reshape(B * T, -1)
        N = table.shape[1]
        ttimes = (torch.arange(N, dtype=torch.float32) / (MAX_FPS / 5))[None].expand(B * T, -1)
        tt = _mlp2(table[..., :2], sd, 'firststage.table_embed')            # (B,13,D)
        xx = torch.cat((x.unsqueeze(2), tt.unsqueeze(1).expand(B, T, N, D)), 2).reshape(B * T, N + 1, D)
        for i in range(_count(sd, 'firststage.pos_layers')):
            xx = layer(xx, sd, 'firststage.pos_layers.%d' % i, tmask, ttimes, 1, heads)
        x = xx.reshape(B, T, N + 1, D)[:, :, 0]
        for i in range(_count(sd, 'firststage.layers')):
            x = layer(x, sd, 'firststage.layers.%d' % i, mask, times, 0, heads)
        pos = _head(x, sd, 'firststage.position_head')
        # ---- second stage :551-571 (skip connection: tokens, not positions)
        x = torch.cat((_t(sd, 'cls_token').expand(B, 1, D), x), 1)
        m2 = torch.zeros((B, T + 1))
        m2[:, 1:] = mask
        for i in range(_count(sd, 'secondstage')):
            x = layer(x, sd, 'secondstage.%d' % i, m2, times, 1, heads)
        rot = _head(x[:, 0], sd, 'rotation_head')
        return rot, pos


def transform_rotationaxes(rot, pos):
    """uplifting/helper.py:394-420 for batched input: rot (B,3), pos (B,T,3) -> (B,3)."""
    rot, pos = torch.as_tensor(rot, dtype=torch.float32), torch.as_tensor(pos, dtype=torch.float32)
    v0 = torch.zeros((pos.shape[0], 3))
    v0[:, :2] = pos[:, 1, :2] - pos[:, 0, :2]
    ex = v0 / torch.linalg.norm(v0, dim=-1, keepdim=True)
    ez = torch.tensor([0.0, 0.0, 1.0]).expand_as(ex)
    ey = torch.cross(ez, ex, dim=-1)
    return torch.stack([(rot * ex).sum(-1), (rot * ey).sum(-1), (rot * ez).sum(-1)], -1)
