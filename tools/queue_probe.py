#!/usr/bin/env python3
"""Which streams of the timed pipeline share a hardware queue?  HIP maps the streams of a process onto GPU_MAX_HW_QUEUES (4) queues
(the least-used one when a stream is first used); kernels of two streams on one queue do not overlap.  Builds bench.py's pipeline
after TTUP_DUMMY_STREAMS extra streams (which shift the mapping), times a few pipelined steps, then groups {submit streams, CNN
lanes, fp32 crop stream, audit stream, default stream} by pairwise probes with two single-thread spin kernels (co-resident when
both take the time of one).  DESIGN.md 12: the headline moves by up to 6 % with the mapping.  Run under `timeout`."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device('cuda:0')
dummy = [torch.cuda.Stream(dev) for _ in range(int(os.environ.get('TTUP_DUMMY_STREAMS', '0')))]
for s in dummy:
    with torch.cuda.stream(s):
        torch.zeros(1, device=dev)
pipe = bench.Pipeline(dev, seed=0)
for _ in range(2):
    pipe.step()
torch.cuda.synchronize()
steps = int(os.environ.get('TTUP_PROBE_STEPS', '8'))
t0 = time.perf_counter()
ticket = None
for _ in range(steps):
    nxt = pipe.submit()
    if ticket is not None:
        pipe.collect(ticket)
    ticket = nxt
pipe.collect(ticket)
torch.cuda.synchronize()
fps = bench.TRIPLES * steps / (time.perf_counter() - t0)

w = pipe.worker
named = [('submit0', w._sub['streams'][0]), ('submit1', w._sub['streams'][1])]
ints = w.net.internal_streams()
named += [('lane%d' % k, s) for k, s in enumerate(ints[:-1])] + [('crops', ints[-1])]
if getattr(w.net, '_audit_stream', None) is not None:
    named.append(('audit', w.net._audit_stream))
named.append(('default', torch.cuda.default_stream(dev)))
CYC = 20_000_000


def spin_ms(streams):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    cur = torch.cuda.current_stream(dev)
    e0.record(cur)
    for s in streams:
        s.wait_event(e0)
        with torch.cuda.stream(s):
            torch.cuda._sleep(CYC)
        ev = torch.cuda.Event(); ev.record(s)
        cur.wait_event(ev)
    e1.record(cur)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


one = spin_ms([named[0][1]])
groups = []
for name, s in named:
    for g in groups:
        if spin_ms([g[0][1], s]) > 1.6 * one:
            g.append((name, s)); break
    else:
        groups.append([(name, s)])
print('dummy streams %s: %.1f frames/s; hardware queues: %s' % (os.environ.get('TTUP_DUMMY_STREAMS', '0'), fps, ' | '.join('+'.join(n for n, _ in g) for g in groups)), flush=True)
