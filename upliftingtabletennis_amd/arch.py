"""Architecture tables for the two networks on the hot path.

Pure-Python description (names, shapes, order) of
  * the WASB / HRNet ball-heatmap CNN   (reference balldetection/models/wasb.py:514-573 config,
    module construction :255-313, state_dict order = construction order), and
  * the uplift transformer               (reference uplifting/model.py:502-527, :303-333).

The tables are used to (1) build seeded random weights with the reference's state_dict
names, (2) serialise a reference-format state_dict into the flat blob that the C-ABI
``ttup_wasb_create`` / ``ttup_uplift_create`` parse (csrc/wasb_net.hip, csrc/uplift.hip walk
the same order and verify every record header).
"""
from collections import namedtuple

ConvSpec = namedtuple('ConvSpec', 'conv bn cin cout k stride has_bias')

STAGE_CHANNELS = (16, 32, 64, 128)
BLOCKS_PER_BRANCH = 2


def _stage(convs, p, nb):
    ch = STAGE_CHANNELS
    for b in range(nb):
        for k in range(BLOCKS_PER_BRANCH):
            q = '%s.branches.%d.%d' % (p, b, k)
            convs.append(ConvSpec(q + '.conv1', q + '.bn1', ch[b], ch[b], 3, 1, False))
            convs.append(ConvSpec(q + '.conv2', q + '.bn2', ch[b], ch[b], 3, 1, False))
    for i in range(nb):
        for j in range(nb):
            q = '%s.fuse_layers.%d.%d' % (p, i, j)
            if j > i:
                convs.append(ConvSpec(q + '.0', q + '.1', ch[j], ch[i], 1, 1, False))
            elif j < i:
                for k in range(i - j):
                    last = k == i - j - 1
                    convs.append(ConvSpec('%s.%d.0' % (q, k), '%s.%d.1' % (q, k), ch[j], ch[i] if last else ch[j], 3, 2, False))


def hrnet_convs(in_ch=9, head_out=3, prefix='model'):
    """Ordered conv list of the WASB HRNet (72 convs for the ball detector)."""
    p = prefix
    c = []
    c.append(ConvSpec(p + '.conv1', p + '.bn1', in_ch, 64, 3, 1, False))
    c.append(ConvSpec(p + '.conv2', p + '.bn2', 64, 64, 3, 1, False))
    q = p + '.layer1.0'
    c.append(ConvSpec(q + '.conv1', q + '.bn1', 64, 32, 1, 1, False))
    c.append(ConvSpec(q + '.conv2', q + '.bn2', 32, 32, 3, 1, False))
    c.append(ConvSpec(q + '.conv3', q + '.bn3', 32, 128, 1, 1, False))
    c.append(ConvSpec(q + '.downsample.0', q + '.downsample.1', 64, 128, 1, 1, False))
    c.append(ConvSpec(p + '.transition1.0.0', p + '.transition1.0.1', 128, 16, 3, 1, False))
    c.append(ConvSpec(p + '.transition1.1.0.0', p + '.transition1.1.0.1', 128, 32, 3, 2, False))
    _stage(c, p + '.stage2.0', 2)
    c.append(ConvSpec(p + '.transition2.2.0.0', p + '.transition2.2.0.1', 32, 64, 3, 2, False))
    _stage(c, p + '.stage3.0', 3)
    c.append(ConvSpec(p + '.transition3.3.0.0', p + '.transition3.3.0.1', 64, 128, 3, 2, False))
    _stage(c, p + '.stage4.0', 4)
    c.append(ConvSpec(p + '.final_layers.0', None, 16, head_out, 1, 1, True))
    return c


def wasb_schema(in_ch=9, head_out=3, prefix='model'):
    """[(state_dict key, shape)] in reference order, without num_batches_tracked."""
    out = []
    for s in hrnet_convs(in_ch, head_out, prefix):
        out.append((s.conv + '.weight', (s.cout, s.cin, s.k, s.k)))
        if s.has_bias:
            out.append((s.conv + '.bias', (s.cout,)))
        if s.bn:
            for f in ('weight', 'bias', 'running_mean', 'running_var'):
                out.append(('%s.%s' % (s.bn, f), (s.cout,)))
    return out


# ---------------------------------------------------------------- uplift transformer
UPLIFT_SIZES = {'small': (32, 8, 4), 'base': (64, 12, 4), 'large': (128, 16, 4), 'huge': (192, 16, 8)}
N_POS_LAYERS = 4       # model.py:323-326
N_SECOND = 4           # model.py:506


def _mlp_embed(p, din, d):
    return [(p + '.fc1.weight', (d, din)), (p + '.fc1.bias', (d,)), (p + '.fc2.weight', (d, d)), (p + '.fc2.bias', (d,))]


def _layer(p, d):
    return [(p + '.attn.qkv.weight', (3 * d, d)), (p + '.attn.qkv.bias', (3 * d,)),
            (p + '.attn.proj.weight', (d, d)),
            (p + '.attn.rotary_emb.inv_freq', None),          # shape filled by caller (head_dim/2)
            (p + '.mlp1.fc1.weight', (d, d)), (p + '.mlp1.fc1.bias', (d,)),
            (p + '.mlp1.fc2.weight', (d, d)), (p + '.mlp1.fc2.bias', (d,)),
            (p + '.norm1.weight', (d,)), (p + '.norm1.bias', (d,)),
            (p + '.norm2.weight', (d,)), (p + '.norm2.bias', (d,))]


def _head(p, d):
    return [(p + '.fc1.weight', (d // 2, d)), (p + '.fc1.bias', (d // 2,)),
            (p + '.fc2.weight', (d // 4, d // 2)), (p + '.fc2.bias', (d // 4,)),
            (p + '.fc3.weight', (3, d // 4)), (p + '.fc3.bias', (3,))]


def uplift_layers(size='large'):
    """Layer-prefix lists (pos_layers, layers, secondstage) for a 'connectstage' model."""
    d, depth, heads = UPLIFT_SIZES[size]
    return (['firststage.pos_layers.%d' % i for i in range(N_POS_LAYERS)],
            ['firststage.layers.%d' % i for i in range(depth - N_SECOND)],
            ['secondstage.%d' % i for i in range(N_SECOND)])


def uplift_schema(size='large'):
    """[(state_dict key, shape)] of get_model('connectstage', size, 'dynamic', 'new') in reference order."""
    d, depth, heads = UPLIFT_SIZES[size]
    pos, first, second = uplift_layers(size)
    out = [('cls_token', (1, 1, d))]
    out += _mlp_embed('embed', 3, d)                       # unused by connectstage but present (model.py:513)
    out += _mlp_embed('firststage.ball_embed', 2, d)
    out += _mlp_embed('firststage.table_embed', 2, d)
    for p in pos + first:
        out += _layer(p, d)
    out += _head('firststage.position_head', d)
    for p in second:
        out += _layer(p, d)
    out += _head('rotation_head', d)
    return [(k, s if s is not None else (d // heads // 2,)) for k, s in out]
