"""bench.py — FuseTrack inference throughput on synthetic 1024x2048 clips (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one frame of PanopticFuseTrack.simple_test per GPU (target frame + previous frame: FlowNet2, ResNet-50-FPN,
BFP-TCEA fusion, UPSNet semantic head, RPN, RoIAlign, bbox / track / mask heads, MaskRemoval, panoptic combine, instance-id
assignment) with inputs already resident in HBM. Weights are synthetic (vps_amd.synth; no checkpoints offline).

The timed region drives the product's clip pipeline, vps_amd.clip_shard.ClipShardRunner (BASELINE config 4): ONE synthetic
clip of N*K frames is sharded contiguously over the N ranks (K frames each: per-GPU work fixed as N grows -> "weak"), every
shard boundary costs one point-to-point RCCL send/recv of the 134 MB gathered pre-neck feature, and the sequential tracker
replay on rank 0 plus the gather of all per-frame results are INSIDE the timed region. At N=1 it is the plain sequential
path. After the timed region (untimed extras, rank 0): an instrumented frame for the roofline numbers, micro-timings of the
HBM-bound kernels, the fixed 30-frame clip of config 4 (strong scaling, field `clip30`), and the CPU baseline.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import statistics
import sys
import time
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
warnings.simplefilter('ignore')

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: bf16 MFMA dense peak (not the 2:1-sparse headline)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured with a float4 copy)
H, W = 1024, 2048
DEFAULT_PREC = 'f16x3'
WORKLOADS = {
    'fusetrack': 'FlowNet2 + ResNet50-FPN + BFP-TCEA + UPSNet panoptic head + RPN / bbox / track / mask heads',
    'fuse': 'PanopticFuse: FlowNet2 + ResNet50-FPN (both frames) + BFP-TCEA + UPSNet panoptic head + RPN / bbox / mask heads, no track head',
    'track': 'PanopticTrack: ResNet50-FPN + UPSNet panoptic head + RPN / bbox / track / mask heads, no FlowNet2 / temporal fusion',
}


def cpu_baseline(seed):
    """the oracle (CPU restatement of the reference path, kind "port") on the host cores of this box, rank 0 only, bounded:
    ONE real 1024x2048 frame (the benched size; the first frame of the clip: FlowNet2 + two ResNet/FPN passes + every head, only
    the tracker comparison against an empty memory is missing) timed once, and the median of 3 runs of a 256x512 frame pair for
    the spread (1 warm-up + 3 timed, as BASELINE.md promises)."""
    from oracle.fusetrack import FuseTrackOracle
    import vps_amd
    from vps_amd import synth
    cfg = vps_amd.Config.fromfile(os.path.join(ROOT, 'configs', 'cityscapes', 'fusetrack.py'))
    model = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    sd = synth.synth_state_dict({k: v.shape for k, v in model.state_dict().items()}, seed)
    o = FuseTrackOracle(sd)
    old = torch.get_num_threads()
    cores = max(1, min(64, (os.cpu_count() or 2) // 2))        # one thread per physical core; SMT siblings slow the small convs
    torch.set_num_threads(cores)
    try:
        h, w = 256, 512
        frames = synth.synth_clip(h, w, 2, seed)
        small = []
        with torch.no_grad():
            o.simple_test(frames[0], frames[0], True)                 # warm-up; also the tracker memory of the timed frame
            mem = (o.prev_bboxes.clone(), o.prev_roi_feats.clone(), o.prev_det_labels.clone())
            for _ in range(3):
                o.prev_bboxes, o.prev_roi_feats, o.prev_det_labels = (t.clone() for t in mem)
                t0 = time.perf_counter()
                o.simple_test(frames[1], frames[0], False)
                small.append(time.perf_counter() - t0)
            frames = synth.synth_clip(H, W, 1, seed)
            o.prev_bboxes = None
            t0 = time.perf_counter()
            o.simple_test(frames[0], frames[0], True)
            dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(old)
    med = statistics.median(small)
    return dict(value=round(1.0 / dt, 5), unit='frames/s', cores=cores, kind='port', n=1,
                sample='1 real FuseTrack frame at %dx%d (the benched size), oracle/ on PyTorch-CPU fp32, %d threads, timed once: %.1f s' % (H, W, cores, dt),
                small_frame_spread=dict(size=[h, w], n=3, warmup=1, seconds=[round(v, 3) for v in small], median_s=round(med, 3),
                                        frames_per_s_scaled_to_the_benched_size=round((h * w) / float(H * W) / med, 5),
                                        note='run-to-run spread of the same oracle on a 256x512 frame pair (the full-size leg is timed once: ~45 s)'))


class _TraceLib:
    """proxy of the loaded libvpship: HIP events (torch's current stream = the launch stream) around EVERY C-ABI launch, by symbol.
    Used for the instrumented single-stream frame only: in-frame durations of the non-conv kernels (the micro-benchmark below
    times the same kernels alone, on synthetic operands)."""

    def __init__(self, lib):
        self._lib, self.trace = lib, []

    def __getattr__(self, name):
        f = getattr(self._lib, name)
        if not name.startswith('vps_') or name in ('vps_abi_version', 'vps_build_info', 'vps_conv2d'):      # convs: nhwc.CONV_TRACE times them
            return f

        def call(*a):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = f(*a)
            e1.record()
            self.trace.append((name, e0, e1))
            return rc
        return call


def hbm_kernels(dev):
    """achieved GB/s of the HBM-bound warp / gather kernels at the BASELINE shapes: algorithmic bytes divided by
    the average launch duration, HIP events on the launch stream (torch's current stream), 20 launches after 3 warm-ups."""
    from vps_amd import hip, nhwc
    from vps_amd.pipeline import DeviceImagePrep
    ws = nhwc.Workspace(dev)
    torch.manual_seed(0)

    def fm(name, h, w, c, ld=None, scale=1.0):
        m = ws.fmap(name, 1, h, w, c, ld)
        m.t.copy_(torch.randn(m.t.shape, device=dev) * scale)
        return m

    def timed(fn, n=20):
        for _ in range(3):
            fn()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3

    out = {}

    def add(name, nbytes, fn, note):
        dt = timed(fn)
        out[name] = dict(us=round(dt * 1e6, 1), algorithmic_MB=round(nbytes / 1e6, 1), achieved_GBs=round(nbytes / dt / 1e9, 1),
                         frac_of_8TBs=round(nbytes / dt / 1e9 / PEAK_HBM_GBS, 4), shape=note)

    h4, w4 = H // 4, W // 4
    # flow-guided feature warp (flow_modules.py:126-148) @P2: read 256 ch + 2 ch flow, write 256 ch
    x = fm('x', h4, w4, 256); fl = fm('fl', h4, w4, 2, 4, 3.0); o = fm('o', h4, w4, 256)
    add('flow_warp', 4.0 * h4 * w4 * (256 + 2 + 256), lambda: nhwc.flow_warp(x, fl, o), '256ch @%dx%d' % (h4, w4))
    # correlation, LiteFlowNetCorr: 2x256 ch in, 81 ch out @P2; FlowNetC: 2x256 ch in, 441 out @128x256
    b = fm('b', h4, w4, 256); c81 = ws.fmap('c81', 1, h4, w4, 81)
    add('correlation_lite_81ch', 4.0 * h4 * w4 * (512 + 81), lambda: nhwc.correlation(x, b, c81, 4, 1), '2x256ch -> 81ch @%dx%d' % (h4, w4))
    h8, w8 = H // 8, W // 8
    a8 = fm('a8', h8, w8, 256); b8 = fm('b8', h8, w8, 256); c441 = ws.fmap('c441', 1, h8, w8, 441)
    add('correlation_flownetc_441ch', 4.0 * h8 * w8 * (512 + 441), lambda: nhwc.correlation(a8, b8, c441, 20, 2), '2x256ch -> 441ch @%dx%d' % (h8, w8))
    add('correlation_flownetc_441ch_split_fp16_mfma', 4.0 * h8 * w8 * (512 + 441), lambda: nhwc.correlation(a8, b8, c441, 20, 2, prec=hip.PREC_F16X3),
        '2x256ch -> 441ch @%dx%d, vps_correlation_f16 (what the f16x3 frame runs)' % (h8, w8))
    # FlowNet2 stage kernel (upsample x4 + resample2d + 2x channelnorm + concat, flownet2.py:142-187), whole-pixel form: the 8-float
    # image pixel + the quarter-resolution flow read, the 12-float pixel of the next network's input written @full res
    lib = hip.load()
    x6 = fm('x6', H, W, 6, 8); flo = fm('flo', h4, w4, 2, 4, 0.2); cc = ws.fmap('cc', 1, H, W, 12)
    add('flow_stage', 4.0 * (H * W * (8 + 12) + h4 * w4 * 2), lambda: hip.check(lib.vps_flow_stage_full(
        x6.ptr(), x6.ld, flo.ptr(), flo.ld, flo.coff, flo.ptr(), flo.ld, flo.coff, H, W, 0, 20.0, cc.ptr(), cc.ld, hip.stream_ptr()), 'stage'),
        'x6 + flow/4 -> 12-channel FlowNetS input @%dx%d' % (H, W))
    # panoptic combine (panoptic_fusetrack.py:588-597): 20-channel quarter-resolution logits + 45 instances (28x28 mask logits) -> two uint8 maps
    from vps_amd import hip as _h
    k = 45
    sc = fm('sc', h4, w4, 19, 20)
    inst = (_h.PanInst * k)()
    rgi = np.random.default_rng(1)
    for j in range(k):
        bw, bh = int(rgi.integers(40, 400)), int(rgi.integers(40, 300))
        x1, y1 = int(rgi.integers(0, W - bw)), int(rgi.integers(0, H - bh))
        inst[j].sx0, inst[j].sy0, inst[j].sx1, inst[j].sy1 = x1, y1, x1 + bw + 1, y1 + bh + 1
        inst[j].seg_ch = 11 + j % 8
        inst[j].bx1, inst[j].by1, inst[j].bx2, inst[j].by2 = x1, y1, x1 + bw, y1 + bh
        inst[j].mask_idx = j
    inst_dev = torch.frombuffer(bytearray(bytes(inst)), dtype=torch.uint8).to(dev)
    ml = torch.randn(k, 28, 28, device=dev)
    pan = torch.empty(H, W, dtype=torch.uint8, device=dev); sem = torch.empty(H, W, dtype=torch.uint8, device=dev)
    add('panoptic_combine', 4.0 * h4 * w4 * 20 + 4.0 * k * 28 * 28 + 2.0 * H * W, lambda: hip.check(lib.vps_panoptic_combine(
        sc.ptr(), sc.ld, h4, w4, 19, 11, hip.ptr(inst_dev), k, hip.ptr(ml), 28, hip.ptr(pan), hip.ptr(sem), H, W, hip.stream_ptr()), 'combine'),
        '19 classes @%dx%d + %d instances -> 2 x uint8 @%dx%d' % (h4, w4, k, H, W))
    # RoIAlign 1000 x 7x7 x 256 over P2..P5 (output-bound)
    lv = [fm('l%d' % s, H // s, W // s, 256) for s in (4, 8, 16, 32)]
    rg = np.random.default_rng(0)
    cx = rg.uniform(0, W, 1000); cy = rg.uniform(0, H, 1000); s = np.exp(rg.uniform(np.log(16), np.log(512), 1000))
    rois = torch.tensor(np.stack([np.zeros(1000), np.clip(cx - s / 2, 0, W - 1), np.clip(cy - s / 2, 0, H - 1), np.clip(cx + s / 2, 0, W - 1),
                                  np.clip(cy + s / 2, 0, H - 1)], 1), dtype=torch.float32, device=dev)
    ro = torch.empty(1000, 7, 7, 256, device=dev)
    add('roi_align_1000x7x7', 4.0 * 1000 * 49 * 256, lambda: nhwc.roi_align(lv, [4, 8, 16, 32], rois, 7, out=ro), '1000 rois x 7x7 x 256ch, 4 levels')
    # BFP gather (bfp_tcea.py:96-109): 5 levels read, 1 written @P2
    lv5 = lv + [fm('l64', H // 64, W // 64, 256)]
    go = ws.fmap('go', 1, h4, w4, 256)
    add('bfp_gather', 4.0 * 256 * (sum((H // s) * (W // s) for s in (4, 8, 16, 32, 64)) + h4 * w4), lambda: nhwc.bfp_gather(lv5, go), '5 levels -> 256ch @%dx%d' % (h4, w4))
    # image prep (Normalize + Pad + ImageToTensor): uint8 HWC in, fp32 CHW out
    img = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device=dev)
    prep = DeviceImagePrep(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True, size_divisor=32, device=dev)
    add('image_prep', 1.0 * H * W * 3 + 4.0 * H * W * 3, lambda: prep.prep(img), 'uint8 %dx%dx3 -> fp32 3x%dx%d' % (H, W, H, W))
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this command under torch.distributed.run, one rank per GPU over RCCL
    (the launcher the driver uses for N > 1: same environment contract), rendezvous on 127.0.0.1 at a free port. A box with fewer
    than N devices gets ONE JSON line with an `error` field and exit code 2 - nothing touches a GPU (VPS_BENCH_BACKEND=gloo lets
    the ranks share devices: functional runs of the multi-rank path on a 1-GPU box, never a scaling number)."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < n and os.environ.get('VPS_BENCH_BACKEND', 'nccl') == 'nccl':
        print(json.dumps({'error': 'bench.py --gpus %d needs %d GPUs on this node (one rank per GPU over RCCL); torch sees %d' % (n, n, ndev),
                          'n_gpus': n, 'devices_visible': ndev}))
        return 2
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the untimed extras (hbm kernel table, 30-frame clip)')
    ap.add_argument('--height', type=int, default=H)
    ap.add_argument('--width', type=int, default=W)
    ap.add_argument('--prec', default=DEFAULT_PREC, choices=['f32', 'bf16', 'bf16x3', 'bf16x6', 'f16x3'], help='arithmetic of the dense contractions')
    ap.add_argument('--variant', default='fusetrack', choices=['fusetrack', 'fuse', 'track'],
                    help='detector (SURVEY 8(f) row 4): the headline metric is fusetrack; the variants are single-GPU only')
    ap.add_argument('--model-config', default=None, help='model config file (default configs/cityscapes/<variant>.py); BASELINE config 5: '
                    'configs/viper/fusetrack_r101.py --height 1088 --width 1920 --prec bf16')
    ap.add_argument('--conv-table', default=None, help='write the per-layer-shape conv timing table of one frame here')
    ap.add_argument('--no-prefetch', action='store_true', help='do not enqueue the next frame behind the current one (A/B of the clip pipelining)')
    ap.add_argument('--single-stream', action='store_true', help='run every frame on one stream (for kernel traces whose durations add up)')
    ap.add_argument('--png-workers', type=int, default=6, help='decode threads of the `from_png` leg (frames read from PNG files through vps_amd.pipeline.ClipFeeder)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(args.gpus)
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if rank == 0:
            print(json.dumps({'error': 'bench.py --gpus %d was started under a launcher with WORLD_SIZE=%d: use --nproc-per-node == --gpus '
                                       '(or start it without a launcher: it re-launches itself)' % (args.gpus, world), 'n_gpus': args.gpus}))
        return 2
    if not torch.cuda.is_available():
        print(json.dumps({'error': 'bench.py measures the HIP path and needs the MI355X (no CPU fallback); torch sees no GPU here', 'n_gpus': args.gpus}))
        return 2
    ndev = torch.cuda.device_count()
    if local >= ndev:
        # only for smoke-testing the multi-rank code path on a 1-GPU box (VPS_BENCH_BACKEND=gloo): ranks share a device
        assert os.environ.get('VPS_BENCH_BACKEND', 'nccl') != 'nccl', 'one GPU per rank is required with RCCL'
        local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    comm_ranks, backend = 1, None
    # VPS_BENCH_DIST=1: create the communicator at world size 1 too (tests/test_rccl_gpu.py: the RCCL branch of this file on a 1-GPU box)
    use_dist = world > 1 or os.environ.get('VPS_BENCH_DIST', '0') == '1'
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        backend = os.environ.get('VPS_BENCH_BACKEND', 'nccl')      # 'nccl' IS RCCL on ROCm
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
        # proof that the communicator spans the job: an all-reduce of ones over the backend the timed region uses
        ones = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(ones)
        comm_ranks = int(ones.item())

    import vps_amd
    from vps_amd import hip, nhwc, synth
    from vps_amd.clip_shard import ClipShardRunner, DetectorBackend, partition
    nhwc.DEFAULT_PREC = nhwc.PREC_NAMES[args.prec]
    Hh, Ww = args.height, args.width
    assert args.variant == 'fusetrack' or world == 1
    cfg = vps_amd.Config.fromfile(args.model_config or os.path.join(ROOT, 'configs', 'cityscapes', args.variant + '.py'))
    model = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    synth.load_synth(model, args.seed)
    depth = getattr(model.backbone, 'depth', 50)
    if args.single_stream:
        model.overlap_streams = False

    # the synthetic clip: 8 distinct frames resident in HBM before timing, cycled (frame t = pool[t % 8])
    pool = [f.to(dev) for f in synth.synth_clip(Hh, Ww, 8, args.seed)]

    def load_frame(t):
        return pool[t % len(pool)]

    def plain_step(t, vid):
        img = load_frame(t)
        ref = load_frame(t - 1) if t else img
        return model(return_loss=False, rescale=True, img=[img], img_meta=[[synth.img_meta(Hh, Ww, 10000 * vid + t + 1)]], ref_img=[ref])

    use_runner = args.variant == 'fusetrack'
    runner = ClipShardRunner(DetectorBackend(model, Hh, Ww, prefetch=not args.no_prefetch), rank, world, dist, dev) if use_runner else None

    def reset():
        model._cache = None; model._handoff = None; model._pf = None
        model.reset_tracker()

    # ---- warm-up: W frames per rank through the same pipeline (also sets up the RCCL p2p communicators) -------------------------
    if use_runner:
        reset()
        runner.run(load_frame, max(args.warmup, 1) * world, video_id=1)
    else:
        for t in range(args.warmup):
            plain_step(t, 1)
    reset()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    # ---- timed region: exactly K frames per rank --------------------------------------------------------------------------------
    t0 = time.perf_counter()
    ndet = 0
    if use_runner:
        outs = runner.run(load_frame, args.steps * world, video_id=2)
        if rank == 0:
            assert len(outs) == args.steps * world
            ndet = sum(int(o['panoptic_cls_inds'].numel()) for o in outs)
    else:
        for t in range(args.steps):
            out = plain_step(t, 2)
            ndet += int(out[2]['panoptic_cls_inds'].numel())
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    total_frames = args.steps * world
    ws_timed_gb = round(model.workspace_bytes() / 1e9, 2) if hasattr(model, 'workspace_bytes') else None      # what the timed configuration holds

    # ---- untimed: BASELINE config 4 as stated — the fixed 30-frame clip over N GPUs (strong scaling) ----------------------------
    clip30 = None
    if use_runner and not args.no_extras:
        reset()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        c0 = time.perf_counter()
        o30 = runner.run(load_frame, 30, video_id=3)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        c30 = time.perf_counter() - c0
        if use_dist:
            tm = torch.tensor([c30], dtype=torch.float64, device=dev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            c30 = float(tm.item())
        if rank == 0:
            sig = [int(np.asarray(o['panoptic_det_obj_ids']).astype(np.int64).sum()) for o in o30]
            clip30 = dict(frames=30, seconds=round(c30, 4), frames_per_s=round(30.0 / c30, 3), scaling='strong',
                          shards=[b - a for a, b in partition(30, world)],
                          note='one 30-frame synthetic clip over %d GPU(s): %d feature hand-off(s), tracker replay on rank 0 and result gather inside' % (world, world - 1),
                          id_checksum=int(sum(sig)))

    # ---- untimed: the same pipeline in the other fp32-grade arithmetic (bf16x6: 6 bf16 MFMAs per product), for comparison ----------
    other = None
    if use_runner and not args.no_extras and world == 1 and args.prec == 'f16x3' and (Hh, Ww) == (H, W):
        old = nhwc.DEFAULT_PREC
        nhwc.DEFAULT_PREC = hip.PREC_BF16X6
        try:
            m2 = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
            synth.load_synth(m2, args.seed)
            m2.ensure_packed(dev)
        finally:
            nhwc.DEFAULT_PREC = old
        r2 = ClipShardRunner(DetectorBackend(m2, Hh, Ww, prefetch=not args.no_prefetch), 0, 1, None, dev)
        r2.run(load_frame, 3, video_id=5)
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        r2.run(load_frame, 10, video_id=6)
        torch.cuda.synchronize()
        c1 = time.perf_counter() - c0
        other = dict(prec='bf16x6', frames=10, frames_per_s=round(10 / c1, 3), ms_per_frame=round(100 * c1, 3),
                     note='same clip pipeline with 3 bf16 planes per operand and 6 MFMAs per fp32 product (error ~2^-23); f16x3: 2+3 fp16 planes, '
                          '3 MFMAs (error <= 3*2^-22, operands within the fp16 range)')
        del m2, r2
        torch.cuda.empty_cache()

    # ---- untimed: what the UNMODIFIED tools/test_vpq.py:41-63 loop gets after the import switch of INTEGRATION.md Level 1 (build_detector,
    # build_dataloader and MMDataParallel from vps_amd): one model(...) call per frame over a data loader, the two maps and the instance
    # vectors fetched to the host after every frame. The loader stand-in yields the reference's test batches (img / img_meta in a
    # cpu_only DataContainer / ref_img, ref_img a tensor of its own like cityscapes_vps.py:137-148 loads it); vps_amd.LookaheadLoader
    # keeps two batches ahead and announces them to the detector (round 6). `no_lookahead`: the same loop over the bare loader (round 5).
    plain = None
    if rank == 0 and not args.no_extras and args.variant == 'fusetrack':
        import vps_amd as _va
        nfr = 20

        class _Loader:
            dataset = list(range(3 + nfr))

            def __init__(self, lo, hi, vid):
                self.lo, self.hi, self.vid = lo, hi, vid

            def __len__(self):
                return self.hi - self.lo

            def __iter__(self):
                for t in range(self.lo, self.hi):
                    img = load_frame(t)
                    ref = (load_frame(t - 1) if t else img).clone()         # the dataset loads the reference frame again: another tensor
                    yield dict(img=[img], img_meta=[_va.DataContainer([[synth.img_meta(Hh, Ww, 10000 * self.vid + t + 1)]], cpu_only=True)], ref_img=[ref])

        def vpq_loop(loader):
            n = 0
            for data in loader:                                             # tools/test_vpq.py:41-63
                with torch.no_grad():
                    r = wrapped(return_loss=False, rescale=True, **data)
                r[2]['fcn_outputs'].cpu(); r[2]['panoptic_outputs'].cpu(); r[2]['panoptic_cls_inds'].cpu(); r[2]['panoptic_det_obj_ids'].cpu()
                n += 1
            return n
        wrapped = _va.MMDataParallel(model, device_ids=[dev.index or 0])
        res = {}
        for name, wrap in (('lookahead', lambda l: _va.LookaheadLoader(l, depth=2, device=dev)), ('no_lookahead', lambda l: l)):
            reset()
            vpq_loop(wrap(_Loader(0, 3, 7)))
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            vpq_loop(wrap(_Loader(3, 3 + nfr, 7)))
            torch.cuda.synchronize()
            res[name] = time.perf_counter() - c0
        c1 = res['lookahead']
        plain = dict(frames=nfr, frames_per_s=round(nfr / c1, 3), ms_per_frame=round(1e3 * c1 / nfr, 3),
                     no_lookahead_frames_per_s=round(nfr / res['no_lookahead'], 3),
                     note='per-frame model(...) calls as tools/test_vpq.py:41-63 makes them, over vps_amd.LookaheadLoader + vps_amd.MMDataParallel '
                          '(the import switch of INTEGRATION.md Level 1: the loader announces the next two frames to the detector), D2H of the two '
                          'uint8 maps and the instance vectors after every frame; no_lookahead: the same loop over the bare loader')
        reset()

    # ---- untimed: the same clip pipeline fed from PNG FILES: vps_amd.pipeline.ClipFeeder decodes with a pool of host
    # threads ahead of the detector, uploads the uint8 frame through pinned memory and normalises / pads on the device -----------------
    from_png = None
    if rank == 0 and not args.no_extras and use_runner and world == 1:
        import shutil
        import tempfile
        from PIL import Image
        from vps_amd.pipeline import ClipFeeder, DeviceImagePrep
        tmpd = tempfile.mkdtemp(prefix='vps_bench_png_')
        try:
            names = []
            for i in range(8):
                fr = synth.synth_frame(Hh, Ww, seed=i % 4, shift=(2 * i, i), noise=2.0).astype(np.uint8)       # BGR, camera-like noise
                fn = os.path.join(tmpd, 'f%02d_city_newImg8bit.png' % i)
                Image.fromarray(np.ascontiguousarray(fr[:, :, ::-1])).save(fn, compress_level=6)
                names.append(fn)
            prep = DeviceImagePrep(**cfg.img_norm_cfg, size_divisor=32, img_scale=(max(Hh, Ww), min(Hh, Ww)), device=dev)
            nfr = 32
            reset()
            runner.run(ClipFeeder([names[t % 8] for t in range(4)], prep, workers=args.png_workers), 4, video_id=8)
            reset()
            feeder = ClipFeeder([names[t % 8] for t in range(nfr)], prep, workers=args.png_workers).start()
            time.sleep(0.2)          # the loader of a running pipeline is ahead of its consumer: the pool fills its window (2 x workers frames)
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            runner.run(feeder, nfr, video_id=9)
            torch.cuda.synchronize()
            c1 = time.perf_counter() - c0
            feeder.close()
            # control: the SAME frames, decoded and prepared before the clock starts (the detector's work depends on the content: the
            # detections of these camera-noise frames differ from the timed clip's) - what the feeder is to be compared with
            reset()
            from vps_amd.pipeline import imread as _imread
            res_frames = [prep.prep(_imread(f))[0].unsqueeze(0) for f in names]
            runner.run(lambda t: res_frames[t % 8], 4, video_id=13)
            reset()
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            runner.run(lambda t: res_frames[t % 8], nfr, video_id=14)
            torch.cuda.synchronize()
            c2 = time.perf_counter() - c0
            from_png = dict(same_frames_resident_frames_per_s=round(nfr / c2, 3), ratio_to_resident=round(c2 / c1, 3),
                            frames=nfr, decode_threads=args.png_workers, frames_per_s=round(nfr / c1, 3), ms_per_frame=round(1e3 * c1 / nfr, 3),
                            png_MB_per_frame=round(sum(os.path.getsize(f) for f in names) / 8e6, 2), decodes=feeder.decodes,
                            consumer_ms_per_frame={k[:-2] + '_ms': round(1e3 * v / nfr, 3) for k, v in feeder.stats.items()},
                            note='ClipShardRunner.run(ClipFeeder(files, DeviceImagePrep)): PNG decode on %d host threads with the native decoder of libvpship (every file once), 6 MB pinned upload, '
                                 'Normalize + Pad + ImageToTensor on the device; the reference decodes and normalises both images of every pair on 2 workers' % args.png_workers)
        finally:
            shutil.rmtree(tmpd, ignore_errors=True)
        reset()

    # ---- untimed: inputs of the multi-GPU cost model (vps_amd.clip_shard.predict_clip_time) measured on this GPU, and its prediction for
    # 2 / 4 / 8 GPUs - no RCCL run exists from this build (one GPU per box), so the first real one has something to be judged against
    scaling_model = None
    if rank == 0 and not args.no_extras and use_runner and world == 1:
        from vps_amd.clip_shard import predict_clip_time

        def timed(fn, n):
            ts = []
            for _ in range(n):
                torch.cuda.synchronize()
                c0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - c0)
            return float(np.median(ts))
        reset()
        t_first = timed(lambda: (reset(), runner.run(load_frame, 1, video_id=11)), 3)          # a frame with nothing to hide its image-only stages behind
        t_hand = timed(lambda: model.gathered_feature(load_frame(3)), 5)
        model._handoff = None
        reset()
        plain_step(0, 12); plain_step(1, 12)
        rec = dict(model._track_record)
        t_assign = timed(lambda: model.track_assign(rec, False), 5)
        reset()
        feat_mb = 4.0 * (Hh // 4) * (Ww // 4) * model.extra_neck.in_channels / 1e6
        t_xfer = feat_mb * 1e6 / 48e9                  # ASSUMED: one xGMI link at ~48 GB/s effective for a single p2p stream (153 GB/s raw)
        t_frame = dt / total_frames
        pred = {}
        for nn in (1, 2, 4, 8):
            c = predict_clip_time(30, nn, t_frame, t_first, t_hand, t_xfer, t_assign)
            w = predict_clip_time(args.steps * nn, nn, t_frame, t_first, t_hand, t_xfer, t_assign)
            pred[str(nn)] = dict(clip30_frames_per_s=round(c['frames_per_s'], 1), clip30_critical=c['critical'],
                                 bench_weak_frames_per_s=round(w['frames_per_s'], 1))
        scaling_model = dict(inputs_ms=dict(frame_steady=round(1e3 * t_frame, 3), frame_without_prefetch_partner=round(1e3 * t_first, 3),
                                            handoff_resnet_fpn_gather=round(1e3 * t_hand, 3), handoff_transfer_assumed=round(1e3 * t_xfer, 3),
                                            tracker_step_replay=round(1e3 * t_assign, 3)),
                             handoff_MB=round(feat_mb, 1), predicted=pred,
                             note='critical-path model of ClipShardRunner (vps_amd/clip_shard.py:predict_clip_time), NOT a measurement: no multi-GPU run exists '
                                  'from this build. transfer time assumes ~48 GB/s for one point-to-point stream over one xGMI link')

    # ---- instrumented extra frame (outside the timed region): per-stage and per-conv-launch HIP events, ONE stream ---------------
    roof, stages, hbm = None, None, None
    if rank == 0:
        reset()
        plain_step(0, 4); plain_step(1, 4)
        # INSTR_FRAMES consecutive instrumented frames; every launch (and stage) is reported with the MEDIAN of its durations over
        # them: a single frame can catch a multi-ms stall of the box (seen once: FlowNet2 42 ms instead of 11) and would then put
        # roofline.frac at a third of its value. A frame whose launch sequence differs from the first one's is dropped.
        INSTR_FRAMES = 3
        real_lib = hip.load()
        runs = []
        for fi in range(INSTR_FRAMES):
            model.profile = {}
            nhwc.CONV_TRACE = []
            hip._lib = tl = _TraceLib(real_lib)
            try:
                plain_step(2 + fi, 4)
                torch.cuda.synchronize()
            finally:
                hip._lib = real_lib
            run = dict(convs=[(c[0], c[1].elapsed_time(c[2]), c[3], c[4]) for c in nhwc.CONV_TRACE],
                       lib=[(name, e0.elapsed_time(e1)) for name, e0, e1 in tl.trace], stages=list(model.stage_times_ms()))
            # (the shape labels of the mask head carry the detection count of the frame: the SEQUENCE must agree, not the labels)
            if not runs or len(run['convs']) == len(runs[0]['convs']):
                runs.append(run)
        med = lambda vals: float(sorted(vals)[len(vals) // 2]) if len(vals) % 2 else float(sum(sorted(vals)[len(vals) // 2 - 1:len(vals) // 2 + 1]) / 2)
        # per-launch records of the conv family: (flops, median ms, shape label, algorithmic bytes)
        convs = [(c[0], med([r['convs'][i][1] for r in runs]), c[2], c[3]) for i, c in enumerate(runs[0]['convs'])]
        in_frame = {}
        lib_runs = [r for r in runs if [x[0] for x in r['lib']] == [x[0] for x in runs[0]['lib']]]
        for i, (name, _) in enumerate(runs[0]['lib']):
            in_frame.setdefault(name, []).append(round(1e3 * med([r['lib'][i][1] for r in lib_runs]), 1))
        st_runs = [r for r in runs if [x[0] for x in r['stages']] == [x[0] for x in runs[0]['stages']]]
        stages = {k: round(med([r['stages'][i][1] for r in st_runs]), 3) for i, (k, _) in enumerate(runs[0]['stages'])}
        fl = sum(c[0] for c in convs)
        ms = sum(c[1] for c in convs)
        nl = len(convs)
        abytes = sum(c[3] for c in convs)
        if args.conv_table:
            agg = {}
            for c in convs:
                a = agg.setdefault(c[2], [0, 0.0, 0.0]); a[0] += 1; a[1] += c[1]; a[2] += c[0]
            with open(args.conv_table, 'w') as f:
                f.write('%-58s %5s %9s %9s %8s\n' % ('layer shape', 'calls', 'ms', 'GFLOP', 'TFLOP/s'))
                for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                    f.write('%-58s %5d %9.3f %9.2f %8.2f\n' % (k, a[0], a[1], a[2] / 1e9, a[2] / a[1] / 1e9))
            # launch order of the instrumented frame (tools/pmc_per_layer.py joins it with the per-dispatch PMC rows)
            with open(args.conv_table + '.ordered.json', 'w') as f:
                json.dump([dict(layer=c[2], flops=c[0], ms=c[1], algorithmic_bytes=c[3]) for c in convs], f)
        nhwc.CONV_TRACE = None
        model.profile = None
        ach = fl / (ms * 1e-3) / 1e12
        # achieved = ALGORITHMIC FLOPs (2*MAC, unpadded: SURVEY 8(d)) / summed launch time of the conv family (HIP events around
        # every launch of the instrumented single-stream frame). peak = what the matrix pipe the kernels ISSUE TO can deliver for
        # that arithmetic: an fp32-grade product costs `nprod` bf16 MFMA products, so peak = dense bf16 MFMA peak / nprod
        # (2500 / 6 = 416.7 for bf16x6, 2500 / 3 for bf16x3); the exact mode issues to the fp32 MFMA pipe (157.3).
        # frac therefore equals executed MFMA TFLOP/s / 2500 (matrix_pipe_*).
        nprod = nhwc.MFMA_PRODUCTS[nhwc.PREC_NAMES[args.prec]]
        pipe_peak = PEAK_FP32_MFMA_TFLOPS if args.prec == 'f32' else PEAK_BF16_MFMA_TFLOPS
        peak = pipe_peak / nprod
        roof = dict(bound='mfma', kernel='conv_mfma_f32_kernel' if args.prec == 'f32' else 'conv_mfma_{h8p,h8,h8s2,bf16h,bf16q,bf16p,n16}_kernel + conv_pw_kernel + conv_thin_kernel <%s> (vps_conv2d family)' % args.prec,
                    achieved=round(ach, 2), peak=round(peak, 1), unit='TFLOP/s', frac=round(ach / peak, 4), traffic=None,
                    mfma_products_per_fp32_product=nprod, matrix_pipe_executed_tflops=round(ach * nprod, 1),
                    matrix_pipe_peak=pipe_peak, matrix_pipe_frac=round(ach * nprod / pipe_peak, 4),
                    algorithmic_bytes_per_frame=round(abytes), launches_per_frame=nl, gflop_per_frame=round(fl / 1e9, 1),
                    conv_ms_per_frame=round(ms, 3), avg_launch_us=round(1e3 * ms / max(nl, 1), 2),
                    scope='SINGLE-STREAM figures: achieved / frac / conv_ms_per_frame are the conv launches of an instrumented one-stream frame summed '
                          '(per-launch medians of 3 frames); the timed frame overlaps five streams, so conv_ms_per_frame may exceed ms_per_step')
        # HBM traffic of the conv kernels per frame: separate rocprofv3 --pmc passes (tools/pmc_traffic.py), not collectable from
        # inside this process; the newest committed measurement for this arithmetic mode is attached
        for rnd in ('r06', 'r05', 'r04', 'r03', 'r02', 'r01'):
            pmc = os.path.join(ROOT, 'profiles', '%s_pmc_traffic_%s.json' % (rnd, args.prec))
            if os.path.exists(pmc):
                pj = json.load(open(pmc))
                roof['traffic'] = round(pj['conv_hbm_bytes_per_frame'])
                roof['traffic_over_algorithmic'] = round(roof['traffic'] / max(abytes, 1), 3)
                roof['traffic_source'] = 'profiles/' + os.path.basename(pmc) + ' (bytes per frame over all conv launches, like algorithmic_bytes_per_frame)'
                # measured in a separate rocprofv3 pass, not in this run: say whether it was taken on the kernel sources this run uses
                roof['traffic_measured_on_these_kernel_sources'] = (pj.get('csrc_sha16') == hip.csrc_sha16()) if pj.get('csrc_sha16') else 'unknown (measured before the stamp existed)'
                break
        # every non-conv C-ABI launch of the instrumented frame, by symbol: [us per call] (HIP events, one stream)
        roof['instrumented_frames'] = dict(conv_launches=len(runs), other_launches=len(lib_runs), stages=len(st_runs))   # per-launch medians over this many single-stream frames
        roof['in_frame_launch_us'] = {k: v for k, v in sorted(in_frame.items())}
        roof['in_frame_non_conv_ms'] = round(sum(sum(v) for v in in_frame.values()) * 1e-3, 3)
        if not args.no_extras and (Hh, Ww) == (H, W):
            hbm = hbm_kernels(dev)
            # the same kernels INSIDE the frame (real operands, neighbours in the caches): achieved GB/s from the in-frame duration
            csym = 'vps_correlation_f16' if 'vps_correlation_f16' in in_frame else 'vps_correlation'       # f16x3 mode: split fp16 on MFMA where an instance exists
            frame_calls = {'flow_warp': ('vps_flow_warp', 0), 'correlation_flownetc_441ch': (csym, 0),
                           'correlation_lite_81ch': (csym, 1), 'flow_stage': ('vps_flow_stage_full', 0),
                           'panoptic_combine': ('vps_panoptic_combine_dev', 0), 'roi_align_1000x7x7': ('vps_roi_align', 0),
                           'bfp_gather': ('vps_bfp_gather', 0)}
            for k, (sym, idx) in frame_calls.items():
                calls = in_frame.get(sym, [])
                if k in hbm and idx < len(calls) and calls[idx] > 0:
                    hbm[k]['in_frame_us'] = calls[idx]
                    hbm[k]['in_frame_GBs'] = round(hbm[k]['algorithmic_MB'] * 1e6 / (calls[idx] * 1e-6) / 1e9, 1)
                    hbm[k]['in_frame_frac_of_8TBs'] = round(hbm[k]['in_frame_GBs'] / PEAK_HBM_GBS, 4)
            roof['hbm_kernels'] = hbm
            roof['hbm_kernels_note'] = ('us / achieved_GBs: the kernel alone on synthetic operands (20 launches); in_frame_*: the same launch inside the '
                                        'instrumented frame (real flows / features) - quote these')
    if rank == 0:
        fps = total_frames / dt
        line = {
            'metric': 'frames/sec %s%s %dx%d' % ({'fusetrack': 'FuseTrack', 'fuse': 'PanopticFuse', 'track': 'PanopticTrack'}[args.variant],
                                                 '' if depth == 50 else ' ResNet-%d' % depth, Hh, Ww),
            'value': round(fps, 3), 'unit': 'frames/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * dt / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'f32': 'f32', 'bf16': 'bf16 operands (rounded at staging) on MFMA, f32 accumulate, f32 activations in HBM', 'bf16x6': 'f32-grade: bf16x6 split operands on MFMA, f32 accumulate',
                      'f16x3': 'f32-grade: f16x3 split operands (fp16 pair with scaled residual, 22 significand bits) on MFMA, f32 accumulate',
                      'bf16x3': 'bf16x3 split operands on MFMA, f32 accumulate'}[args.prec], 'data': 'synthetic',
            'config': {'workload': '2-frame pair %s (%s), synthetic %dx%d clip of %d frames, batch 1' % (
                           {'fusetrack': 'FuseTrack', 'fuse': 'PanopticFuse', 'track': 'PanopticTrack'}[args.variant], WORKLOADS[args.variant].replace('ResNet50', 'ResNet%d' % depth),
                           Hh, Ww, total_frames),
                       'weights': 'synthetic (vps_amd.synth seed %d)' % args.seed,
                       'detections_per_frame': round(ndet / max(total_frames, 1), 1),
                       'parallelism': ('clip-shard x%d (contiguous shards of %d frames), 1 p2p feature hand-off per shard boundary, tracker replay '
                                       'on rank 0 + result gather inside the timed region' % (world, args.steps)) if use_runner else 'single GPU',
                       'pipelining': ('two HIP streams per frame (neck + detection heads || semantic head) + %d for the image-only stages of the frames ahead '
                                      '(FlowNetC-S-S chain | ResNet + FPN + gather | FlowNetSD; ring of three output slots); %d frame(s) announced: frame t+1 at the start of call t '
                                      'if not in flight, frame t+2 behind neck(t)' % (getattr(model, 'pre_streams', 1), DetectorBackend.prefetch_depth)) if (use_runner and not args.no_prefetch and not args.single_stream) else 'two HIP streams per frame' if not args.single_stream else 'one stream',
                       'workspace_GB': ws_timed_gb,
                       'workspace_GB_after_the_untimed_extras': round(model.workspace_bytes() / 1e9, 2),       # + the one-stream instrumented frame, the per-call loop
                       'timed_region': 'inputs resident in HBM; excludes the H2D of the two 25 MB frames and the D2H of the two uint8 maps that '
                                       'tools/test_vpq.py:46-56 pays (~0.4 ms per frame over PCIe 5 x16 when not overlapped)'},
            'roofline': roof, 'stage_ms': stages,
            'stage_ms_note': 'median of 3 instrumented extra frames on ONE stream; the timed frames overlap the semantic head with the detection heads on two streams and the next frame\'s FlowNet2 + backbone + FPN on a third',
        }
        if os.environ.get('VPS_S2_HALO', '1')[0] == '0':
            line['config']['experimental'] = 'VPS_S2_HALO=0: the phase-split stride-2 halo kernel switched off (A/B run, not the default configuration)'
        if plain is not None:
            line['test_vpq_loop'] = plain
        if from_png is not None:
            line['from_png'] = from_png
        if use_dist:
            line['rccl_ranks' if backend == 'nccl' else backend + '_ranks'] = comm_ranks        # from an all_reduce of ones
            line['config']['backend'] = 'RCCL (torch.distributed "nccl")' if backend == 'nccl' else backend + ' (functional run, ranks may share a GPU: not a scaling number)'
        if scaling_model is not None:
            scaling_model['measured'] = False
            line['extra'] = {'scaling_model': scaling_model}       # a critical-path MODEL (assumed link rate), kept out of the headline fields
        line['config']['host_reads_per_frame'] = '2 (the detection list after MaskROI: 8 KB; kept list + track ids + range report + status words at the end: 2 KB)'
        line['f16_fallbacks'] = int(nhwc.F16_FALLBACKS[0])        # layers switched from f16x3 to bf16x6 by the fp16 range report (0 here)
        if clip30 is not None:
            line['clip30'] = clip30
        if other is not None:
            line['other_arithmetic'] = other
        if not args.no_cpu_baseline and world == 1:          # the CPU leg is reported at N = 1 only (it takes about a minute of host time)
            line['cpu_baseline'] = cpu_baseline(args.seed)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the ONE JSON line is the last thing this process writes (communicator teardown / library chatter comes before it)
        sys.stdout.flush(); sys.stderr.flush()
        print(json.dumps(line), flush=True)


if __name__ == '__main__':
    sys.exit(main() or 0)
