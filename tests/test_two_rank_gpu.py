"""GPU: the multi-rank PRODUCT path of SURVEY 8(e) inside the driver's suite. A 1-GPU box cannot run two RCCL
ranks, so two PROCESSES share the one GPU over gloo (the transport differs, everything above it - ClipShardRunner with the real
DetectorBackend, the feature hand-off, deferred tracking, the streamed records, the sequential replay on rank 0 - is the code the
8-GPU run uses). tools/check_two_rank.py does the comparison; here it is launched the way the driver launches bench.py
(python -m torch.distributed.run, 127.0.0.1) under a hard timeout."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _launch(extra, env_extra, timeout):
    env = dict(os.environ, VPS_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', **env_extra)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tools', 'check_two_rank.py')] + extra
    t0 = time.time()
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    return p.returncode, p.stdout.decode(errors='replace'), time.time() - t0


@pytest.mark.timeout(900)
def test_two_ranks_sharing_the_gpu_equal_the_sequential_run():
    """8 frames at 256x512 in f16x3 (the benchmarked arithmetic), shards (4, 4): per frame the track ids, classes, scores and both maps
    of the 2-rank pipeline are array_equal to the sequential single-process run, and the gathered feature rank 1 received at the shard
    boundary is bitwise the one rank 0 computes for its last frame"""
    rc, out, dt = _launch(['--height', '256', '--width', '512', '--frames', '8', '--prec', 'f16x3'], {}, 800)
    print(out[-3000:])
    assert rc == 0, 'tools/check_two_rank.py exit code %d' % rc
    assert '2-rank pipeline EQUALS the sequential run' in out
    assert 'hand-off feature rank 0 -> 1 (frame 3' in out and 'bitwise equal' in out
    assert out.count("'ids': True") == 8 and "False" not in out.split('2-rank pipeline')[0].split('hand-off feature')[-1]


@pytest.mark.timeout(600)
def test_a_stalled_peer_ends_the_clip_with_one_json_error_line_not_a_hang():
    """rank 1 withholds its shard (alive, silent): rank 0 - whose hand-off send finds no receiver and whose posted receives of rank
    1's frame records never complete - gives up after VPS_CLIP_TIMEOUT_S with ONE JSON error line naming the peer and what it waited
    for, exit code 3, well inside the test's timeout"""
    rc, out, dt = _launch(['--height', '128', '--width', '256', '--frames', '4', '--withhold', '1'], {'VPS_CLIP_TIMEOUT_S': '20'}, 500)
    print(out[-3000:])
    assert rc != 0
    lines = [l for l in out.splitlines() if l.startswith('{') and '"error"' in l]
    assert len(lines) == 1, lines
    err = json.loads(lines[0])
    assert err['rank'] == 0 and 'rank 1' in err['error'] and ('hand-off' in err['error'] or 'frame 2' in err['error']), err
    assert dt < 400, dt


@pytest.mark.timeout(600)
def test_a_stalled_sender_of_the_hand_off_is_reported_by_the_receiver():
    """rank 0 withholds everything: rank 1 waits for the feature hand-off of frame 1 and reports it"""
    rc, out, dt = _launch(['--height', '128', '--width', '256', '--frames', '4', '--withhold', '0'], {'VPS_CLIP_TIMEOUT_S': '20'}, 500)
    print(out[-3000:])
    assert rc != 0
    lines = [l for l in out.splitlines() if l.startswith('{') and '"error"' in l]
    assert len(lines) == 1, lines
    err = json.loads(lines[0])
    assert err['rank'] == 1 and 'hand-off' in err['error'], err
    assert dt < 400, dt
