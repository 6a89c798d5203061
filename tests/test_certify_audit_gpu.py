"""The certified argmax's supporting machinery on the MI355X box (-m gpu): the continuous eps audit (side-stream fp32 re-runs,
candidate-level errors from the crops, widening + re-certification), the repair path of pipelined clips, and the per-handle
serialisation of consecutive calls issued on different streams."""
import os

import numpy as np
import pytest
import torch

from conftest import has_gpu
from upliftingtabletennis_amd import synth, weights

pytestmark = pytest.mark.gpu
if has_gpu():
    from upliftingtabletennis_amd import pipeline, refine, wasb, _lib

H, W = 96, 160            # small network: the full-frame fp32 twin is cheap


def _fp32_peaks(sd, frames_dev, res):
    """(idx, win, xyv) of every triple on the full-frame fp32 path."""
    twin = wasb.WASBNet(sd, resolution=res, max_batch=1, dtype='f32')
    x = wasb.preprocess_triples(frames_dev, res)
    idx, win = [], []
    for k in range(x.shape[0]):
        _, i1, w1 = twin.forward(x[k:k + 1], want_heatmap=False, want_peaks=True)
        idx.append(i1); win.append(w1)
    idx, win = torch.cat(idx), torch.cat(win)
    return idx, win, refine.refine_windows_device(idx, win, res[1], res[0], 1920, 1080, _lib.REFINE_TABLE)


def test_audit_twin_recomputes_the_production_heatmap_bit_for_bit():
    """The audit compares the fp32 twin with a ONE-sample bf16 handle instead of touching the production handle: both bf16 handles
    must give the same heatmap for the same triple (per-tile deterministic kernels, any micro-batch / lane count)."""
    sd = weights.random_wasb_state_dict(5)
    frames, _ = synth.synth_frames(20, 720, 1280, seed=5)
    fr = torch.from_numpy(frames).cuda()
    net = wasb.WASBNet(sd, resolution=(1280, 704), max_batch=18, dtype='bf16')
    heat, _, _ = net.forward_frames(fr, want_heatmap=True)
    for t in (0, 7, 8, 17):
        h1, _, _ = net._audit_twin().forward_frames(fr[t:t + 3], want_heatmap=True)
        assert torch.equal(h1[0], heat[t]), t
    e = float(net.heatmap_error(fr, 3).item())
    assert 0 < e < 0.1 * float(heat.abs().max())
    # the running audit looks at a quarter-width strip of the frame: a smaller sample of the same error distribution
    strips = [float(net.heatmap_error(fr, 3, x0).item()) for x0 in (0, 480, 960)]
    print('\nfull-frame error %.4g, strip errors %s' % (e, ['%.4g' % v for v in strips]))
    assert all(0.25 * e < v <= 1.25 * e for v in strips), (e, strips)


def test_max_abs_diff_equals_the_torch_expression():
    """ttup_max_abs_diff (the audit's error measure) == (a - b).abs().max(), bit for bit; ragged sizes, zero size, NaN."""
    g = torch.Generator(device='cuda').manual_seed(5)
    for n in (1, 63, 4097, 704 * 1280, 3 * 704 * 1280 + 5):
        a = torch.randn((n,), device='cuda', generator=g)
        b = a + 1e-3 * torch.randn((n,), device='cuda', generator=g)
        assert torch.equal(wasb.max_abs_diff(a, b), (a - b).abs().max())
    assert float(wasb.max_abs_diff(a[:0], b[:0])) == 0.0
    assert float(wasb.max_abs_diff(a, a)) == 0.0
    b[1234] = float('nan')
    assert torch.isnan(wasb.max_abs_diff(a, b))
    b[1234] = float('inf')
    assert float(wasb.max_abs_diff(a, b)) == float('inf')
    with pytest.raises(ValueError):
        wasb.max_abs_diff(a, b[:-1])
    # column ranges (the strip audit leaves out the columns near an artificial border), running maximum, column slices
    x = torch.randn((3, 5, 40, 64), device='cuda', generator=g)
    y = x + 1e-2 * torch.randn(x.shape, device='cuda', generator=g)
    for c0, c1 in ((0, 64), (9, 50), (0, 1), (63, 64), (17, 17)):
        want = (x[..., c0:c1] - y[..., c0:c1]).abs().max() if c1 > c0 else torch.zeros((), device='cuda')
        assert torch.equal(wasb.max_abs_diff(x, y, cols=(c0, c1)), want), (c0, c1)
    run = torch.empty((1,), dtype=torch.float32, device='cuda')
    wasb.max_abs_diff(x[0], y[0], out=run)
    wasb.max_abs_diff(x[1], y[1], cols=(3, 60), out=run, accumulate=True)
    assert torch.equal(run[0], torch.maximum((x[0] - y[0]).abs().max(), (x[1, ..., 3:60] - y[1, ..., 3:60]).abs().max()))
    assert torch.equal(wasb.slice_columns(x, 8, 24), x[..., 8:32].contiguous())
    with pytest.raises(ValueError):
        wasb.slice_columns(x, 50, 24)


def test_eps_audit_widens_on_brighter_frames_and_recertifies():
    """eps is calibrated on a dark clip; a later, much brighter clip has a larger bf16 error.  The side-stream audit (here one in
    two triples) must notice, widen eps and re-run the clip, after which every index is the fp32 argmax and the observed error sits
    inside the safety margin again."""
    sd = weights.random_wasb_state_dict(9)                      # noise weights: near-ties everywhere
    base, _ = synth.synth_frames(14, H, W, seed=9)
    # "dark" = a flat mid-grey clip (normalised inputs near 0: small activations, small bf16 error); "bright" = the same scene at
    # four times the contrast, saturating both ends
    dark = np.clip(115.0 + (base.astype(np.float32) - 70.0) * 0.1, 0, 255).astype(np.uint8)
    bright = np.clip(128.0 + (base.astype(np.float32) - 70.0) * 4.0, 0, 255).astype(np.uint8)
    usd = weights.random_uplift_state_dict(9, 'large')
    worker = pipeline.StreamWorker('cuda:0', sd, usd, net_wh=(W, H), max_triples=12, traj_len=32, seq_len=50, audit_every=2, audit_seed=1)
    table_px = np.concatenate([np.random.default_rng(0).uniform(100, 900, (13, 2)), np.ones((13, 1))], 1)
    fd, fb = torch.from_numpy(dark).cuda(), torch.from_numpy(bright).cuda()
    worker.process_clip(fd, table_px, 60.0)
    eps_dark = worker.certify_eps
    err_bright = max(float(worker.net.heatmap_error(fb, t).item()) for t in range(12))
    assert err_bright * 1.5 > eps_dark, 'premise: the bright clip must break the dark clip\'s bound (%.3g vs eps %.3g)' % (err_bright, eps_dark)
    # pipelined: the bright clip is submitted under the stale eps and must come back re-certified
    t1 = worker.submit(fb)
    t2 = worker.submit(fb)
    o1 = worker.collect(t1, table_px, 60.0)
    o2 = worker.collect(t2, table_px, 60.0)
    a = worker.audit
    print('\neps %.4g (dark clip) -> %.4g; bright clip error %.4g; audited %d frames, widened %d times, %d clips re-certified, max err / eps %.3f'
          % (eps_dark, worker.certify_eps, err_bright, a['audited_frames'], a['widened'], a['recertified_clips'], a['max_err_over_eps']))
    assert worker.certify_eps > eps_dark and a['widened'] >= 1 and a['recertified_clips'] >= 1
    assert a['max_err_over_eps'] <= 1 / 1.5 + 1e-6
    ref_idx, _, _ = _fp32_peaks(sd, fb, (W, H))
    for o in (o1, o2):
        _, idx, win = worker.net.forward_frames(fb)
        worker.net.fix_uncertified(idx, win, frames_u8=fb)
        assert torch.equal(idx, ref_idx)
    assert torch.equal(o1['xyv'], o2['xyv'])


class _FixedPick:
    """Stands in for the worker's audit rng: the side-stream audit then re-runs the triple the test names."""
    def __init__(self, t):
        self.t = t

    def integers(self, n):
        return min(self.t, n - 1)


def test_one_out_of_bound_frame_between_audits_missed_is_stated_caught_is_recertified():
    """What the sampled audit does with ONE frame whose bf16 error exceeds eps inside an otherwise quiet clip.
    A dark clip calibrates eps; the next clip is the same dark scene with a single high-contrast frame planted in the middle (the
    three triples that contain it have errors above the bound).
      (a) the side-stream audit samples a triple that does NOT contain the frame: the strip audit cannot see it.  Either the
          candidate-level audit (the fp32 crops that the frame's own heatmaps needed) catches it -- then eps is widened and every
          index is the fp32 path's -- or nobody does: then the bound still holds on every OTHER triple (their indices equal the
          fp32 path's), `audit` states what share of the frames was audited (< 1), and max_err_seen is below the planted error.
      (b) the audit samples a triple that contains it: eps is widened by the strip audit, the clip is re-certified, every index
          equals the fp32 path's and the observed error sits inside the safety factor again.
    Also covers the adaptive rate: the fast rate until eps has stood for `audit_settle_clips` clips, the steady rate after."""
    sd = weights.random_wasb_state_dict(9)
    usd = weights.random_uplift_state_dict(9, 'large')
    base, _ = synth.synth_frames(14, H, W, seed=9)
    dark = np.clip(115.0 + (base.astype(np.float32) - 70.0) * 0.1, 0, 255).astype(np.uint8)
    mixed = dark.copy()
    mixed[7] = np.clip(128.0 + (base[7].astype(np.float32) - 70.0) * 4.0, 0, 255).astype(np.uint8)
    fd, fm = torch.from_numpy(dark).cuda(), torch.from_numpy(mixed).cuda()
    table_px = np.concatenate([np.random.default_rng(0).uniform(100, 900, (13, 2)), np.ones((13, 1))], 1)
    ref_idx, _, _ = _fp32_peaks(sd, fm, (W, H))

    def run(pick):
        # one audited triple per 12-triple clip at both rates, so that the pick is the only thing that differs between (a) and (b)
        worker = pipeline.StreamWorker('cuda:0', sd, usd, net_wh=(W, H), max_triples=12, traj_len=32, seq_len=50, audit_every=12, audit_every_fast=12,
                                       audit_settle_clips=2, audit_seed=1, audit_crops_every=0)
        worker.process_clip(fd, table_px, 60.0)
        eps0 = worker.certify_eps
        errs = np.array([float(worker.net.heatmap_error(fm, t).item()) for t in range(12)])
        assert errs[[5, 6, 7]].max() * 1.5 > eps0, 'premise: the planted frame must break the dark clip\'s bound (%.3g vs eps %.3g)' % (errs.max(), eps0)
        worker._rng = _FixedPick(pick)
        out = worker.process_clip(fm, table_px, 60.0)
        _, idx, win = worker.net.forward_frames(fm)
        worker.net.fix_uncertified(idx, win, frames_u8=fm)
        return worker, eps0, errs, out, idx
    # (a) the audit looks elsewhere
    worker, eps0, errs, out, idx = run(0)
    a = worker.audit
    assert a['frames_seen'] == 24 and 0 < a['audited_share'] <= 1 and a['audit_every_now'] == 12          # (a clip re-run after a widening is counted once)
    inbound = errs <= worker.certify_eps
    assert torch.equal(idx.cpu()[torch.from_numpy(inbound)], ref_idx.cpu()[torch.from_numpy(inbound)]), 'the bound holds on these triples: their indices must be the fp32 path\'s'
    if worker.certify_eps > eps0:
        assert a['widen_sources']['candidates'] >= 1 and a['max_err_over_eps'] <= 1 / 1.5 + 1e-6
        caught_by = 'the candidate-level audit'
    else:
        assert a['max_err_seen'] < errs.max() and a['audited_share'] < 1, a
        caught_by = 'nobody (stated: audited share %.3f)' % a['audited_share']
    n_bad = int((idx.cpu() != ref_idx.cpu()).sum())
    print('\n(a) audit elsewhere: eps %.4g -> %.4g, planted errors %s, caught by %s, %d of 12 indices differ from the fp32 path'
          % (eps0, worker.certify_eps, ['%.3g' % e for e in errs[[5, 6, 7]]], caught_by, n_bad))
    # (b) the audit samples the planted frame
    worker, eps0, errs, out, idx = run(6)
    a = worker.audit
    assert worker.certify_eps > eps0 and a['widened'] >= 1 and a['max_err_over_eps'] <= 1 / 1.5 + 1e-6, a
    inbound = errs <= worker.certify_eps
    assert inbound[6] and a['max_err_seen'] >= errs[6] * (1 - 1e-6)
    assert torch.equal(idx.cpu()[torch.from_numpy(inbound)], ref_idx.cpu()[torch.from_numpy(inbound)])
    assert inbound.all(), 'the neighbours of the audited triple carry the same planted frame: 1.5 x its error should cover them (%s vs eps %.3g)' % (errs[[5, 6, 7]], worker.certify_eps)
    print('(b) audit on the planted frame: eps %.4g -> %.4g, widened %d (sources %s), %d heatmaps / %d clips re-certified'
          % (eps0, worker.certify_eps, a['widened'], a['widen_sources'], a['recertified_heatmaps'], a['recertified_clips']))
    # (c) audit crops (round 6): the strip audit looks elsewhere, but one single-candidate heatmap per 4 triples gets an fp32 crop that
    # reports |bf16 - fp32| at ITS WINNER -- the pixel the detection rests on, not the frame's worst pixel (that is the strip audit's
    # measure, `errs`).  For every phase: the largest error seen afterwards is at least the winner error of every picked triple, and a
    # picked triple whose winner error breaks the old bound widens eps.  The chance that a given frame is looked at this way is (triples
    # that contain it) / audit_crops_every per clip -- 3/4 here, 3/16 in production, on top of the strip audit's 1/64 ... 1/256.
    twin = wasb.WASBNet(sd, resolution=(W, H), max_batch=1, dtype='f32')
    xs = wasb.preprocess_triples(fm, (W, H))
    hf = torch.cat([twin.forward(xs[k:k + 1], want_heatmap=True, want_peaks=True)[0] for k in range(12)])
    n_widened = 0
    for phase in range(4):
        w3 = pipeline.StreamWorker('cuda:0', sd, usd, net_wh=(W, H), max_triples=12, traj_len=32, seq_len=50, audit_every=12, audit_every_fast=12,
                                   audit_settle_clips=2, audit_seed=1, audit_crops_every=4)
        w3.process_clip(fd, table_px, 60.0)
        e0, seen0 = w3.certify_eps, w3.audit['max_err_seen']
        hb, ib, _ = w3.net.forward_frames(fm, want_heatmap=True)
        werr = (hb.reshape(12, -1).gather(1, ib.reshape(12, 1)) - hf.reshape(12, -1).gather(1, ib.reshape(12, 1))).abs().reshape(12).cpu().numpy()
        w3._rng = _FixedPick(0)                                # strip audit on triple 0
        w3._rng_crops = _FixedPick(phase)
        w3.process_clip(fm, table_px, 60.0)
        a3 = w3.audit
        picked = [t for t in range(12) if (t + phase) % 4 == 0]
        assert a3['audit_crop_frames'] == 2 * 3 and a3['audited_share'] > a3['strip_audited_share'], a3
        assert a3['max_err_seen'] >= max(werr[picked].max(), seen0) * (1 - 1e-5), (phase, a3['max_err_seen'], werr[picked], seen0)
        if werr[picked].max() * 1.5 > e0 * (1 + 1e-6):
            assert w3.certify_eps > e0 and a3['widen_sources']['candidates'] >= 1, (phase, a3)
            n_widened += 1
        print('(c) phase %d: audit crops on triples %s, winner errors %s, eps %.4g -> %.4g' % (phase, picked, ['%.3g' % v for v in werr[picked]], e0, w3.certify_eps))
    print('(c) %d of 4 phases widened eps from an audit crop' % n_widened)
    # adaptive rate: fast until eps has stood for audit_settle_clips clips in a row, steady afterwards, fast again after a widening
    w2 = pipeline.StreamWorker('cuda:0', sd, usd, net_wh=(W, H), max_triples=12, traj_len=32, seq_len=50, audit_every=48, audit_every_fast=6, audit_settle_clips=2, audit_crops_every=0)
    assert w2.audit_rate() == 6
    w2.process_clip(fd, table_px, 60.0)                    # calibration (8 frames) + 12 triples at one audit per 6
    assert w2.audit['audited_frames'] == 8 + 2 and w2.audit['frames_seen'] == 12
    for _ in range(20):                                    # the same clip until eps has stood for two clips in a row
        if w2.audit['quiet_clips'] >= 2:
            break
        w2.process_clip(fd, table_px, 60.0)
    assert w2.audit['quiet_clips'] >= 2 and w2.audit_rate() == 48
    n0 = w2.audit['audited_frames']
    for _ in range(3):
        w2.process_clip(fd, table_px, 60.0)
    assert w2.audit['audited_frames'] - n0 <= 1            # 36 triples at one per 48
    assert abs(w2.audit['audited_share'] - w2.audit['audited_frames'] / w2.audit['frames_seen']) < 1e-12
    w2._quiet_clips = 0
    assert w2.audit_rate() == 6


def test_audit_crops_give_single_candidate_heatmaps_an_fp32_check():
    """`certify_audit_crops(every, phase)` on content whose heatmaps have ONE candidate each (planted-peak weights: nothing would get an
    fp32 crop): exactly the frames with (f + phase) % every == 0 get one, the handle's running |bf16 - fp32| maximum at candidates
    becomes non-zero -- the error at the winners -- and NOTHING else changes: status, indices and 3x3 windows are the bf16 path's
    whatever the phase (an audit measures; it does not decide what a frame returns)."""
    sd = weights.random_wasb_state_dict(21, planted=True)
    frames, _ = synth.synth_frames(14, H, W, seed=21)
    fr = torch.from_numpy(frames).cuda()
    net = wasb.WASBNet(sd, resolution=(W, H), max_batch=12, dtype='bf16')
    net.calibrate(fr, n=4)
    _, idx0, win0 = net.forward_frames(fr)
    st0 = net.certify_status(12).cpu().numpy() & 3
    s0 = net.certify_stats(reset=True)
    if not (st0 == 0).all():
        pytest.skip('premise: every heatmap of the planted-peak clip has a single candidate (status %s)' % st0)
    assert s0['max_candidate_err'] == 0.0 or s0['crops'] == 0
    for phase in (1, 2):
        net.certify_audit_crops(4, phase)
        _, idx1, win1 = net.forward_frames(fr)
        st1 = net.certify_status(12).cpu().numpy() & 3
        s1 = net.certify_stats(reset=True)
        assert (st1 == 0).all() and torch.equal(idx0, idx1) and torch.equal(win0, win1), (phase, st1)
        assert s1['exact_singles'] == 3 and s1['crops'] == 3 and s1['max_candidate_err'] > 0, s1
    net.certify_audit_crops(0)
    _, idx2, _ = net.forward_frames(fr)
    assert (net.certify_status(12).cpu().numpy() & 3 == 0).all() and torch.equal(idx0, idx2)


def test_class2_crops_reproduce_the_fp32_path_at_every_alignment():
    """Crops of class 2 (round 6: the candidates of a crop fit a 14-position core, the crop is centred on THAT core and its fp32 pass
    pruned to the core's cone -- the crop's first 160 rows / columns): in parity mode every heatmap of a planted-peak clip gets one, at
    whatever (y mod 8, x mod 8) the ball sits; index and fp32 3x3 window must be the full-frame fp32 path's, bit for bit, and the
    counters must show that the class was used."""
    res = (640, 352)
    sd = weights.random_wasb_state_dict(23, planted=True)
    net = wasb.WASBNet(sd, resolution=res, max_batch=12, dtype='bf16')
    phases, n_small, n_crops = set(), 0, 0
    for k in range(4):
        fr = torch.from_numpy(synth.synth_frames(14, 720, 1280, seed=60 + k)[0]).cuda()
        if k == 0:
            net.calibrate(fr, n=4, exact_windows=True)
            net.certify_stats(reset=True)
        _, idx, win = net.forward_frames(fr)
        st = net.certify_status(12).cpu().numpy() & 3
        ref_idx, ref_win, _ = _fp32_peaks(sd, fr, res)
        assert (st == 1).all(), st
        assert torch.equal(idx, ref_idx) and torch.equal(win, ref_win), (k, (idx != ref_idx).nonzero().flatten().tolist())
        for v in idx.cpu().tolist():
            phases.add(((v // res[0]) % 8, (v % res[0]) % 8))
        s1 = net.certify_stats(reset=True)
        n_small += s1['small_crops']; n_crops += s1['crops']
    print('\n%d crops, %d of class 2, %d (y, x) alignments mod 8' % (n_crops, n_small, len(phases)))
    assert n_crops == 48 and n_small >= 40 and len(phases) >= 16, (n_crops, n_small, len(phases))          # (a single candidate always fits class 2)


def test_pipelined_repair_uses_the_tickets_own_status():
    """collect() of clip k runs after submit() of clip k+1 has flipped the handle's per-call slot: the heatmaps of clip k that the
    crop budget could not settle (status 2) must be repaired from clip k's OWN status.  A tiny budget and a wide eps on noise
    weights force status 2 on many heatmaps of three different clips; every returned detection must equal the fp32 path's."""
    sd = weights.random_wasb_state_dict(3)
    usd = weights.random_uplift_state_dict(3, 'large')
    clips = [torch.from_numpy(synth.synth_frames(14, H, W, seed=40 + k)[0]).cuda() for k in range(3)]
    worker = pipeline.StreamWorker('cuda:0', sd, usd, net_wh=(W, H), max_triples=12, traj_len=32, seq_len=50, audit_every=0)
    table_px = np.concatenate([np.random.default_rng(0).uniform(100, 900, (13, 2)), np.ones((13, 1))], 1)
    worker.process_clip(clips[0], table_px, 60.0)
    worker.certify_eps = worker.net.widen_eps(worker.certify_eps * 4)        # more candidates than the 256 kept per heatmap on some maps
    tickets = [worker.submit(c) for c in clips[:2]]
    outs = [worker.collect(tickets[0], table_px, 60.0)]
    tickets.append(worker.submit(clips[2]))
    outs += [worker.collect(tickets[1], table_px, 60.0), worker.collect(tickets[2], table_px, 60.0)]
    assert worker.fp32_reruns > 0, 'the repair path did not run: no heatmap was flagged'
    n_flagged = 0
    for c, o, t in zip(clips, outs, tickets):
        ref_idx, ref_win, ref_xyv = _fp32_peaks(sd, c, (W, H))
        st = o['status']
        n_flagged += int((st == 2).sum())
        assert torch.equal(t['idx'], ref_idx)
        fp32_win = torch.from_numpy(st != 0).cuda()
        assert torch.equal(t['win'][fp32_win], ref_win[fp32_win])
        assert torch.equal(o['xyv'][fp32_win], ref_xyv[fp32_win])
    print('\n%d heatmaps flagged over 3 pipelined clips, %d full-frame fp32 re-runs' % (n_flagged, worker.fp32_reruns))
    assert n_flagged == worker.fp32_reruns


def test_slot_reuse_after_a_whole_clip_rerun_keeps_every_tickets_status():
    """Round-3 advisor (medium): a whole-clip re-run inside collect() is ONE extra forward on the production handle, which flips the
    per-call slot: from then on clip m on stream X reuses the slot of clip m-1 on stream Y while m-1's status / info copies may still
    be pending.  The library orders the slot's reset behind those copies (per-slot read events).  Here: a collect-time whole-clip
    re-run, then four more clips two deep in the pipeline, many of their heatmaps over the candidate budget (status 2): every
    ticket must carry its OWN status, every flagged heatmap must be repaired, every index must be the fp32 path's."""
    sd = weights.random_wasb_state_dict(3)
    usd = weights.random_uplift_state_dict(3, 'large')
    clips = [torch.from_numpy(synth.synth_frames(14, H, W, seed=60 + k)[0]).cuda() for k in range(5)]
    worker = pipeline.StreamWorker('cuda:0', sd, usd, net_wh=(W, H), max_triples=12, traj_len=32, seq_len=50, audit_every=0)
    table_px = np.concatenate([np.random.default_rng(0).uniform(100, 900, (13, 2)), np.ones((13, 1))], 1)
    worker.process_clip(clips[0], table_px, 60.0)
    worker.certify_eps = worker.net.widen_eps(worker.certify_eps * 2)
    t0 = worker.submit(clips[0])
    worker.certify_eps = worker.net.widen_eps(worker.certify_eps * 2 / worker.net.HEADROOM)      # past the guard factor: whole clip again
    o0 = worker.collect(t0, table_px, 60.0)
    assert worker.recertified_clips >= 1
    refs = [_fp32_peaks(sd, c, (W, H)) for c in clips]
    assert torch.equal(t0['idx'], refs[0][0])
    for rep in range(3):
        tickets, outs = [worker.submit(clips[1]), worker.submit(clips[2])], []
        for k in (3, 4):
            outs.append(worker.collect(tickets[len(outs)], table_px, 60.0))
            tickets.append(worker.submit(clips[k]))
        outs += [worker.collect(t, table_px, 60.0) for t in tickets[2:]]
        n_flagged = 0
        for k, (t, o) in enumerate(zip(tickets, outs)):
            ref_idx, ref_win, ref_xyv = refs[k + 1]
            n_flagged += int((o['status'] == 2).sum())
            assert torch.equal(t['idx'], ref_idx), (rep, k)
            fp32_win = torch.from_numpy(o['status'] != 0).cuda()
            assert torch.equal(t['win'][fp32_win], ref_win[fp32_win]) and torch.equal(o['xyv'][fp32_win], ref_xyv[fp32_win])
        assert n_flagged > 0, 'premise: some heatmaps must overflow the candidate budget'
    print('\n%d heatmaps flagged in the last round, %d fp32 re-runs in all, %d whole-clip re-runs' % (n_flagged, worker.fp32_reruns, worker.recertified_clips))


def test_widening_within_the_guard_factor_reruns_only_guarded_heatmaps():
    """eps is widened by 15 % between submit and collect (what a new error maximum does).  Heatmaps whose guard band -- the pixels
    between 2 eps and 2.5 eps below the maximum -- is empty have the same candidate set under the new eps and keep their result;
    only the others are run again (on the one-sample certified handle).  Every index must equal the fp32 path's, the count of
    re-run heatmaps must equal the count of guard bits, and the whole clip must NOT be re-run."""
    sd = weights.random_wasb_state_dict(21, planted=True, eps=1.0)          # planted peak + unscaled noise: a mix of clear peaks and near-ties
    usd = weights.random_uplift_state_dict(3, 'large')
    fr = torch.from_numpy(synth.synth_frames(14, H, W, seed=21)[0]).cuda()
    worker = pipeline.StreamWorker('cuda:0', sd, usd, net_wh=(W, H), max_triples=12, traj_len=32, seq_len=50, audit_every=0)
    table_px = np.concatenate([np.random.default_rng(0).uniform(100, 900, (13, 2)), np.ones((13, 1))], 1)
    worker.process_clip(fr, table_px, 60.0)
    worker.net.SUBSET_MAX_SHARE = 1.0           # this weight set guards most heatmaps: keep the subset path (default: whole call above 25 %)
    e0 = worker.certify_eps
    t = worker.submit(fr)
    torch.cuda.synchronize()
    raw = t['status'].numpy().copy()
    worker.certify_eps = worker.net.widen_eps(1.15 * e0 / worker.net.HEADROOM)
    assert abs(worker.certify_eps - 1.15 * e0) < 1e-6 * e0
    before = worker.recertified_heatmaps
    o = worker.collect(t, table_px, 60.0)
    n_guard = int(((raw & 4) != 0).sum())
    print('\n%d of 12 heatmaps had a non-empty guard band and were run again; status %s' % (n_guard, raw.tolist()))
    assert worker.recertified_heatmaps - before == n_guard and worker.recertified_clips == 0
    ref_idx, ref_win, ref_xyv = _fp32_peaks(sd, fr, (W, H))
    assert torch.equal(t['idx'], ref_idx)
    # the status of a re-run heatmap is the RE-RUN's: 1 only where its window holds fp32 values (crops or the full-frame repair), 0
    # where it is a single candidate under the widened eps and keeps its bf16 window (a 60-clip soak on noisier weights found two
    # such heatmaps labelled 1: tools/soak_debug.py)
    st = np.asarray(o['status'])
    assert set(np.unique(st)) <= {0, 1}
    n_fp32 = 0
    for k in range(st.shape[0]):
        if st[k] != 0:
            n_fp32 += 1
            assert torch.equal(t['win'][k].reshape(-1), ref_win[k].reshape(-1)), (k, int(raw[k]))
    print('%d of %d windows hold fp32 values and equal the fp32 path\'s' % (n_fp32, st.shape[0]))
    # the same clip run from scratch under the widened eps gives the same detections
    again = worker.process_clip(fr, table_px, 60.0)
    assert torch.equal(again['xyv'], o['xyv'])
    # with the default share the same situation re-runs the whole clip in one batched pass instead: same detections
    if n_guard > 3:
        worker.net.SUBSET_MAX_SHARE = 0.25
        e1 = worker.certify_eps
        t = worker.submit(fr)
        worker.certify_eps = worker.net.widen_eps(1.15 * e1 / worker.net.HEADROOM)
        clips_before = worker.recertified_clips
        o2 = worker.collect(t, table_px, 60.0)
        assert worker.recertified_clips == clips_before + 1
        assert torch.equal(t['idx'], ref_idx)              # the ticket describes the pass that produced its detections
        assert torch.equal(o2['xyv'], worker.process_clip(fr, table_px, 60.0)['xyv'])


def test_uplift_beside_the_cnn_is_bit_stable():
    """The uplift transformer of clip k runs on a side stream while the detector of clip k+1 is busy.  Measured on MI355X: packed
    fp32 instructions with operand swizzles (the compiler's choice for the RoPE / softmax arithmetic) return wrong values while
    another kernel's waves on the same CU feed MFMAs from LDS reads -- the CNN's chain kernels -- and pos3d moved by up to 4e-3
    (csrc/common.h, tools/pk_coresidency_repro.hip).  The uplift's device code is built without packed fp32 instructions; its
    results beside a busy CNN must be bit-identical to the ones of an idle GPU.  HIP streams share a few hardware queues
    round-robin, and a stream that lands on the CNN's queue is serialised behind it, so five side streams are tried."""
    from upliftingtabletennis_amd import uplift
    usd = weights.random_uplift_state_dict(0, 'large')
    up = uplift.get_model('connectstage', 'large', 'dynamic', 'new', state_dict=usd, max_batch=64, max_len=50)
    args = [torch.from_numpy(a).cuda() for a in synth.synth_trajectories(3, 40, seed=3, pad=10)]
    r0, p0 = up(*args)
    r0, p0 = r0.clone(), p0.clone()
    net = wasb.WASBNet(weights.random_wasb_state_dict(0, planted=True), resolution=(1280, 704), max_batch=16, dtype='bf16')
    fr = torch.from_numpy(synth.synth_frames(18, 720, 1280, seed=1)[0]).cuda()
    net.forward_frames(fr)
    s1 = torch.cuda.Stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); up(*args); e1.record()
    torch.cuda.synchronize()
    alone_ms = e0.elapsed_time(e1)
    bad, worst, slowest = 0, 0.0, 0.0
    for side in [torch.cuda.Stream() for _ in range(5)]:
        for k in range(6):
            with torch.cuda.stream(s1):
                for _ in range(2):
                    net.forward_frames(fr)
            with torch.cuda.stream(side):
                e0.record(); r, p = up(*args); e1.record()
            side.synchronize()
            if not (torch.equal(r, r0) and torch.equal(p, p0)):
                bad += 1
                worst = max(worst, float((p - p0).abs().max()))
            torch.cuda.synchronize()
            slowest = max(slowest, e0.elapsed_time(e1))
    print('\nuplift beside the CNN: %d of 30 calls differ from the idle result (worst |dpos| %.2e); %.2f ms alone, up to %.2f ms beside the CNN'
          % (bad, worst, alone_ms, slowest))
    assert bad == 0


def test_soak_on_changing_content_matches_the_fp32_path():
    """tools/soak_audit.py with 8 clips: new background / noise / trajectory / blob size / brightness per clip, the continuous audit
    on (one triple per 16), every triple compared with the full-frame fp32 path: no index and no fp32-window mismatch, whatever the
    audits widen on the way (profiles/r3_soak_audit.json holds a 160-clip run)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ); e['TTUP_SOAK_CLIPS'] = '8'
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'soak_audit.py')], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print('\nsoak: %d triples, eps %.4f -> %.4f (%d widenings), %d clips re-run, %.2f crops per heatmap' % (
        d['triples_checked_against_fp32'], d['eps_first'], d['eps_last'], d['eps_widened'], d['recertified_clips'], d['crops_per_heatmap']))
    assert d['triples_checked_against_fp32'] == 8 * 64 and d['argmax_mismatches'] == 0 and d['fp32_window_mismatches'] == 0


def test_packed_fp32_reproducer_victims_without_swizzles_are_clean(tmp_path):
    """tools/pk_coresidency_repro.hip, the stand-alone reproducer behind csrc/common.h's TTUP_NO_PACKED_FP32_*: compiled and run
    here.  The two kinds of code the library ships -- packed fp32 disabled (uplift, refine), packed fp32 without operand swizzles
    (convolution epilogues) -- must come out clean beside every neighbour; the count for the swizzled forms is reported (160 runs
    in 4 x 4 x 10; 40 wrong on this pool: every run beside the LDS-fed MFMA neighbour)."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not found')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / 'pk_repro'
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-fno-fast-math', '-o', str(exe), os.path.join(root, 'tools', 'pk_coresidency_repro.hip')],
                   check=True, capture_output=True, timeout=600)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=600).stdout
    m = re.search(r'swizzled packed fp32: (\d+), packed fp32 disabled: (\d+), packed fp32 without swizzles: (\d+)', out)
    assert m, out[-500:]
    print('\nreproducer: runs with wrong results -- swizzled packed fp32 %s, packed fp32 disabled %s, packed fp32 without swizzles %s (of 160 each)' % m.groups())
    assert int(m.group(2)) == 0 and int(m.group(3)) == 0


@pytest.mark.parametrize('lanes', ['1', '2'])
def test_pipelined_small_clips_on_alternating_streams_do_not_share_buffers(monkeypatch, lanes):
    """Clips of <= one micro-batch run on ONE lane; consecutive submits alternate between two caller streams.  Different clips
    back to back must give what they give one at a time (TTUP_LANES=1: the micro-batches run on the caller's streams and the
    handle orders them itself; 2: on the lane's stream)."""
    monkeypatch.setenv('TTUP_LANES', lanes)
    sd = weights.random_wasb_state_dict(0, planted=True)
    usd = weights.random_uplift_state_dict(0, 'large')
    worker = pipeline.StreamWorker('cuda:0', sd, usd, net_wh=(1280, 704), max_triples=16 if lanes == '2' else 8, traj_len=32, seq_len=50, audit_every=0)
    table_px = np.concatenate([np.random.default_rng(0).uniform(100, 900, (13, 2)), np.ones((13, 1))], 1)
    clips = [torch.from_numpy(synth.synth_frames(8, 720, 1280, seed=70 + k)[0]).cuda() for k in range(4)]
    alone = [worker.process_clip(c, table_px, 60.0) for c in clips]
    assert not torch.equal(alone[0]['xyv'], alone[1]['xyv'])
    for rep in range(3):
        tickets = [worker.submit(c) for c in clips]
        outs = [worker.collect(t, table_px, 60.0) for t in tickets]
        for a, o in zip(alone, outs):
            assert torch.equal(a['xyv'], o['xyv']) and torch.equal(a['spin'], o['spin'])
