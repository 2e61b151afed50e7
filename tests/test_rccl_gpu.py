"""RCCL (torch.distributed backend "nccl") as far as ONE GPU allows: communicator creation with `device_id`,
an all-reduce, the point-to-point path of ClipShardRunner._post (a send to self inside one batch: the same ncclSend / ncclRecv group
the hand-off and the record streaming use), the runner on a process group, and the RCCL branch of bench.py at world size 1.
Every case runs in a child process under a timeout: a wedged communicator must not take the test session (or the box) with it."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _child(code, port, timeout=240, extra_env=None):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, 'tests'))
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, '-c', textwrap.dedent(code)], env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    return p.stdout


def test_rccl_world_1_communicator_allreduce_and_self_p2p(dev):
    out = _child('''
        import torch, torch.distributed as dist
        from vps_amd.clip_shard import ClipShardRunner
        dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
        dist.init_process_group('nccl', device_id=dev)
        assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1
        ones = torch.ones(1, dtype=torch.int32, device=dev); dist.all_reduce(ones); assert int(ones.item()) == 1
        dist.barrier()
        # the runner's p2p primitive: a batch with a send and the matching receive (to self: one ncclGroup). The source is produced on
        # the current stream right before the post - RCCL orders the send behind it (what `_post` relies on for the 134 MB hand-off)
        r = ClipShardRunner(None, 0, 1, dist, dev)
        src = torch.empty(1 << 22, device=dev); dst = torch.zeros(1 << 22, device=dev)
        for i in range(3):
            src.normal_()                               # enqueued, not synchronised
            want = src.clone()
            for q in r._post([dist.P2POp(dist.isend, src, 0), dist.P2POp(dist.irecv, dst, 0)]):
                q.wait()
            assert torch.equal(dst, want), i
        dist.destroy_process_group()
        print('RCCL_OK')
    ''', 29541)
    assert 'RCCL_OK' in out


def test_clip_shard_runner_on_an_rccl_group_equals_the_plain_run(dev):
    """ClipShardRunner(..., dist=<nccl group>) at world 1 == ClipShardRunner(..., dist=None): the detector's streams next to a live
    communicator (its own stream, its registered buffers) leave the results bitwise alone"""
    out = _child('''
        import numpy as np, torch, torch.distributed as dist
        import vps_amd
        from vps_amd import hip, nhwc, synth
        from vps_amd.clip_shard import ClipShardRunner, DetectorBackend
        dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
        nhwc.DEFAULT_PREC = hip.PREC_F16X3
        cfg = vps_amd.Config.fromfile('configs/cityscapes/fusetrack.py')
        m = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
        synth.load_synth(m, 0)
        H, W, n = 128, 256, 4
        fr = [f.to(dev) for f in synth.synth_clip(H, W, n, 0)]
        def run(d):
            m._cache = None; m._pf = None; m._handoff = None; m.reset_tracker()
            outs = ClipShardRunner(DetectorBackend(m, H, W), 0, 1, d, dev).run(lambda t: fr[t], n)
            return [(np.asarray(o['panoptic_det_obj_ids']).copy(), o['panoptic_outputs'].cpu().numpy().copy()) for o in outs]
        a = run(None)
        dist.init_process_group('nccl', device_id=dev)
        b = run(dist)
        dist.barrier()
        dist.destroy_process_group()
        for (ia, pa), (ib, pb) in zip(a, b):
            assert np.array_equal(ia, ib) and np.array_equal(pa, pb)
        print('RUNNER_OK', len(a))
    ''', 29542)
    assert 'RUNNER_OK 4' in out


def test_bench_rccl_branch_at_world_1(dev):
    """bench.py with VPS_BENCH_DIST=1: init_process_group('nccl', device_id=...), the barriers and max-over-ranks all-reduce around the
    timed region, `rccl_ranks` from an all-reduce of ones - the code path of an N-GPU run, on the one GPU of this box (small frames)"""
    env = dict(os.environ, VPS_BENCH_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29543', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--height', '128', '--width', '256',
                        '--no-extras', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=400, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1, (p.stdout[-1500:], p.stderr[-1500:])           # ONE JSON line (library chatter may surround it)
    j = json.loads(lines[0])
    assert j['n_gpus'] == 1 and j['rccl_ranks'] == 1 and j['steps'] == 3 and j['value'] > 0
    assert 'RCCL' in j['config']['backend']
