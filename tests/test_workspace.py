"""vps_amd.nhwc.Workspace / Pool on the CPU (no kernel runs): liveness-based reuse of activation blocks (round 5). What the GPU tests
check end to end (pooled == one buffer per activation, bitwise) rests on these rules:
  * a block goes back to the free list of the stream it was taken on and is handed out again only there;
  * best fit within Pool.SLACK, exact repetition of a take / give sequence returns the same blocks (stable addresses);
  * padded maps (ld > C), `temp=False` and `pooling = False` give persistent named buffers;
  * `scope()` returns what was taken inside and not released; `out=True` routes to the `out` workspace; `with_out` shares everything else."""
import ctypes

import pytest
import torch

from vps_amd import hip, nhwc


@pytest.fixture
def streams(monkeypatch):
    cur = {'s': 0}
    monkeypatch.setattr(hip, 'stream_ptr', lambda: ctypes.c_void_p(cur['s']) if cur['s'] else None)
    return cur


def test_pool_is_per_stream_best_fit_and_repeatable(streams):
    pool = nhwc.Pool('cpu')
    a = pool.take(0, 1000); b = pool.take(0, 4000)
    assert pool.blocks == 2 and pool.total == 4 * 5000
    pool.give(0, a); pool.give(0, b)
    assert pool.take(0, 900) is a                      # best fit within the slack (1000 <= 1.5 * 900)
    assert pool.take(0, 3000) is b
    c = pool.take(0, 100)                              # nothing within 1.5x: a block of its own
    assert c is not a and c is not b and pool.blocks == 3
    pool.give(0, a); pool.give(0, b); pool.give(0, c)
    d = pool.take(7, 1000)                             # another stream never sees stream 0's blocks
    assert all(d is not x for x in (a, b, c)) and pool.blocks == 4
    # the same sequence again returns the same blocks: a frame's activations keep their addresses
    seq1 = [pool.take(0, n) for n in (1000, 100, 4000)]
    for x in seq1:
        pool.give(0, x)
    seq2 = [pool.take(0, n) for n in (1000, 100, 4000)]
    assert [x.data_ptr() for x in seq1] == [x.data_ptr() for x in seq2] and pool.blocks == 4


def test_workspace_temporaries_release_scope_and_persistence(streams):
    ws = nhwc.Workspace('cpu')
    ws.pooling = True
    x = ws.fmap('x', 1, 8, 8, 16, temp=True)
    y = ws.fmap('y', 1, 8, 8, 16, temp=True)
    assert x.t.data_ptr() != y.t.data_ptr() and len(ws._live) == 2 and 'x' not in ws.bufs
    ws.release(x, None, x)                             # None and a second release are ignored
    z = ws.fmap('z', 1, 8, 8, 16, temp=True)           # x's block, immediately
    assert z.t.data_ptr() == x.t.data_ptr() and ws.pool.blocks == 2
    ws.release(y.window(4, 8))                         # a window releases the map it belongs to
    assert len(ws._live) == 1
    with ws.scope():
        a = ws.fmap('a', 1, 4, 4, 32, temp=True)
        with ws.scope():
            b = ws.fmap('b', 1, 4, 4, 32, temp=True)
        assert id(b.t) not in ws._live and id(a.t) in ws._live
    assert id(a.t) not in ws._live and id(z.t) in ws._live            # z was taken outside the scopes
    ws.release(z)
    assert not ws._live
    # padded layouts, temp=False and pooling off: persistent by name, zero-initialised, same tensor every time
    p = ws.fmap('padded', 1, 8, 8, 18, temp=True)
    assert p.ld == 20 and 'padded' in ws.bufs and float(p.t.abs().sum()) == 0 and ws.fmap('padded', 1, 8, 8, 18, temp=True).t is p.t
    q = ws.fmap('plain', 1, 8, 8, 16)
    assert ws.fmap('plain', 1, 8, 8, 16).t is q.t
    ws.pooling = False
    r = ws.fmap('unpooled', 1, 8, 8, 16, temp=True)
    assert 'unpooled' in ws.bufs and not ws._live
    ws.release(r)                                      # harmless


def test_workspace_out_target_and_stream_keys(streams):
    pool = nhwc.Pool('cpu')
    lane, slot = nhwc.Workspace('cpu', pool=pool), nhwc.Workspace('cpu', pool=pool)
    view = lane.with_out(slot)
    o = view.fmap('levels.p2', 1, 8, 8, 16, out=True)
    assert 'levels.p2' in slot.bufs and 'levels.p2' not in lane.bufs
    assert lane.fmap('levels.p2', 1, 8, 8, 16, out=True).t is not o.t                # without an out target: the workspace's own name
    t0 = view.fmap('t', 1, 8, 8, 16, temp=True)        # temporaries and persistent internals are the lane's
    assert id(t0.t) in lane._live
    k = view.fmap('internal', 1, 8, 8, 18)
    assert 'internal' in lane.bufs
    streams['s'] = 0x1000                              # another stream: another free list
    t1 = view.fmap('t1', 1, 8, 8, 16, temp=True)
    view.release(t0)                                   # goes back to stream 0's list ...
    t2 = view.fmap('t2', 1, 8, 8, 16, temp=True)       # ... so stream 0x1000 cannot get it
    assert t2.t.data_ptr() != t0.t.data_ptr()
    streams['s'] = 0
    t3 = view.fmap('t3', 1, 8, 8, 16, temp=True)
    assert t3.t.data_ptr() == t0.t.data_ptr()
    view.release(t1, t2, t3)
    assert not lane._live and pool.blocks == 3
