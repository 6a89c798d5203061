"""Oracle (a2): WASB / HRNet ball-heatmap CNN forward, fp32 torch-CPU.

Table-driven restatement of ``balldetection/models/wasb.py`` working directly on a
reference-format ``state_dict`` (keys ``model.conv1.weight`` ...), eval-mode BN.

Reference map:
  stem                wasb.py:446-451
  Bottleneck          wasb.py:85-105      (layer1, wasb.py:452, cfg :532-539)
  BasicBlock          wasb.py:48-64
  transition layers   wasb.py:362-396, forward :454-475
  HighResolutionModule.forward (branches + fuse)  wasb.py:227-245, fuse build :179-222
  head                wasb.py:328-333, :484 ; WASBNet.forward keeps channel 1  :596-608
  config              wasb.py:514-573  (stage channels 16/32/64/128, 2 BasicBlocks per branch)
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm2d default, wasb.py:41 passes only momentum

STAGE_CHANNELS = (16, 32, 64, 128)  # wasb.py:545,553,561
BLOCKS_PER_BRANCH = 2               # wasb.py:544,552,560


def _t(sd, key):
    v = sd[key]
    return v if isinstance(v, torch.Tensor) else torch.as_tensor(v)


def _bn(x, sd, p):
    return F.batch_norm(x, _t(sd, p + '.running_mean'), _t(sd, p + '.running_var'),
                        _t(sd, p + '.weight'), _t(sd, p + '.bias'), False, 0.0, BN_EPS)


def _conv(x, sd, p, stride=1):
    w = _t(sd, p + '.weight')
    return F.conv2d(x, w, None, stride, w.shape[-1] // 2)


def _conv_bn(x, sd, conv, bn, stride=1, relu=False):
    y = _bn(_conv(x, sd, conv, stride), sd, bn)
    return F.relu(y) if relu else y


def basic_block(x, sd, p):
    """wasb.py:48-64 (no downsample inside stages: in==out channels)."""
    out = _conv_bn(x, sd, p + '.conv1', p + '.bn1', relu=True)
    out = _conv_bn(out, sd, p + '.conv2', p + '.bn2')
    return F.relu(out + x)


def bottleneck(x, sd, p):
    """wasb.py:85-105 with the 1x1 downsample of wasb.py:400-405."""
    out = _conv_bn(x, sd, p + '.conv1', p + '.bn1', relu=True)
    out = _conv_bn(out, sd, p + '.conv2', p + '.bn2', relu=True)
    out = _conv_bn(out, sd, p + '.conv3', p + '.bn3')
    res = _conv_bn(x, sd, p + '.downsample.0', p + '.downsample.1')
    return F.relu(out + res)


def _fuse_term(xj, sd, p, i, j):
    """One term fuse_layers[i][j](x[j]) of wasb.py:179-222."""
    q = '%s.fuse_layers.%d.%d' % (p, i, j)
    if j > i:      # 1x1 conv + BN + nearest upsample by 2**(j-i)   (:190-198)
        y = _conv_bn(xj, sd, q + '.0', q + '.1')
        return F.interpolate(y, scale_factor=2 ** (j - i), mode='nearest')
    y = xj         # chain of i-j stride-2 3x3 convs; ReLU on all but the last (:202-219)
    for k in range(i - j):
        y = _conv_bn(y, sd, '%s.%d.0' % (q, k), '%s.%d.1' % (q, k), stride=2, relu=(k != i - j - 1))
    return y


def hr_module(xs, sd, p, n_out=None):
    """HighResolutionModule.forward, wasb.py:227-245.  n_out limits how many fused outputs
    are produced (reference always produces all: multi_scale_output=True, :306)."""
    nb = len(xs)
    xs = list(xs)
    for b in range(nb):
        for k in range(BLOCKS_PER_BRANCH):
            xs[b] = basic_block(xs[b], sd, '%s.branches.%d.%d' % (p, b, k))
    outs = []
    for i in range(nb if n_out is None else n_out):
        y = xs[0] if i == 0 else _fuse_term(xs[0], sd, p, i, 0)
        for j in range(1, nb):
            y = y + (xs[j] if i == j else _fuse_term(xs[j], sd, p, i, j))
        outs.append(F.relu(y))
    return outs


def hrnet_features(x, sd, prefix='model', return_taps=False):
    """HRNet.forward up to the stage-4 outputs, wasb.py:445-477."""
    p = prefix
    taps = {}
    x = _conv_bn(x, sd, p + '.conv1', p + '.bn1', relu=True)
    taps['stem1'] = x
    x = _conv_bn(x, sd, p + '.conv2', p + '.bn2', relu=True)
    taps['stem2'] = x
    x = bottleneck(x, sd, p + '.layer1.0')
    taps['layer1'] = x
    # transition1 (:454-459): branch0 3x3 s1 128->16, branch1 3x3 s2 128->32
    xs = [_conv_bn(x, sd, p + '.transition1.0.0', p + '.transition1.0.1', relu=True),
          _conv_bn(x, sd, p + '.transition1.1.0.0', p + '.transition1.1.0.1', stride=2, relu=True)]
    taps['trans1_0'], taps['trans1_1'] = xs
    ys = hr_module(xs, sd, p + '.stage2.0')
    taps['stage2_0'], taps['stage2_1'] = ys
    # transition2 (:462-467): new branch from y_list[-1]
    xs = [ys[0], ys[1], _conv_bn(ys[-1], sd, p + '.transition2.2.0.0', p + '.transition2.2.0.1', stride=2, relu=True)]
    ys = hr_module(xs, sd, p + '.stage3.0')
    taps['stage3_0'], taps['stage3_1'], taps['stage3_2'] = ys
    xs = [ys[0], ys[1], ys[2],
          _conv_bn(ys[-1], sd, p + '.transition3.3.0.0', p + '.transition3.3.0.1', stride=2, relu=True)]
    ys = hr_module(xs, sd, p + '.stage4.0')
    for i, y in enumerate(ys):
        taps['stage4_%d' % i] = y
    return (ys, taps) if return_taps else ys


def hrnet_forward(x, sd, prefix='model'):
    """HRNet.forward, wasb.py:445-486: head 1x1 conv (+bias) on stage-4 output 0."""
    ys = hrnet_features(x, sd, prefix)
    w = _t(sd, prefix + '.final_layers.0.weight')
    b = _t(sd, prefix + '.final_layers.0.bias')
    return F.conv2d(ys[0], w, b), ys


def wasb_forward(x, sd):
    """WASBNet.forward (classify_invisible=False), wasb.py:596-608 -> (B,1,H,W) fp32."""
    with torch.no_grad():
        heat, _ = hrnet_forward(torch.as_tensor(x, dtype=torch.float32), sd, 'model')
        return heat[:, 1:2]
