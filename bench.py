"""bench.py — FuseTrack inference throughput on synthetic 1024x2048 clips (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one frame of PanopticFuseTrack.simple_test (target frame + previous frame: FlowNet2, ResNet-50-FPN, BFP-TCEA
fusion, UPSNet semantic head, RPN, RoIAlign, bbox / track / mask heads, MaskRemoval, panoptic combine) with inputs
already resident in HBM. Weights are synthetic (vps_amd.synth; no checkpoints offline) and scaled so the heads emit the
configured maximum of ~100 detections per frame. One process per GPU; each rank runs its own contiguous shard of the clip
(weak scaling, no data-path collective inside a shard; the shard-boundary feature hand-off is in vps_amd/clip_shard.py).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
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
H, W = 1024, 2048


def cpu_baseline(seed):
    """the oracle (CPU restatement of the reference path) on a bounded sample of the same workload, rank 0 only"""
    from oracle.fusetrack import FuseTrackOracle
    import vps_amd
    from vps_amd import synth
    h, w = 256, 512
    cfg = vps_amd.Config.fromfile(os.path.join(ROOT, 'configs', 'cityscapes', 'fusetrack.py'))
    model = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    sd = synth.synth_state_dict({k: v.shape for k, v in model.state_dict().items()}, seed)
    o = FuseTrackOracle(sd)
    frames = synth.synth_clip(h, w, 2, seed)
    cores = torch.get_num_threads()
    with torch.no_grad():
        o.simple_test(frames[0], frames[0], True)                 # warm-up (first frame of the clip)
        t0 = time.perf_counter()
        o.simple_test(frames[1], frames[0], False)
        dt = time.perf_counter() - t0
    frac = (h * w) / float(H * W)
    return dict(value=round(frac / dt, 5), unit='frames/s', cores=cores, kind='port',
                sample='1 FuseTrack frame pair at %dx%d (=%.4f of the %dx%d frame, scaled by pixel count), oracle/ on '
                       'PyTorch-CPU fp32, %.1f s' % (h, w, frac, H, W, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--height', type=int, default=H)
    ap.add_argument('--width', type=int, default=W)
    ap.add_argument('--prec', default='bf16x6', choices=['f32', 'bf16x3', 'bf16x6'], help='arithmetic of the dense contractions')
    ap.add_argument('--variant', default='fusetrack', choices=['fusetrack', 'fuse', 'track'],
                    help='detector (SURVEY 8(f) row 4): the headline metric is fusetrack; the variants are single-GPU only')
    ap.add_argument('--conv-table', default=None, help='write the per-layer-shape conv timing table of one frame here')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, 'launch with --nproc-per-node == --gpus'
    assert torch.cuda.is_available(), 'bench.py measures the HIP path and needs the MI355X (no CPU fallback)'
    ndev = torch.cuda.device_count()
    if local >= ndev:
        # only for smoke-testing the multi-rank code path on a 1-GPU box (VPS_BENCH_BACKEND=gloo): ranks share a device
        assert os.environ.get('VPS_BENCH_BACKEND', 'nccl') != 'nccl', 'one GPU per rank is required with RCCL'
        local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get('VPS_BENCH_BACKEND', 'nccl')      # 'nccl' IS RCCL on ROCm
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    import vps_amd
    from vps_amd import hip, nhwc, synth
    nhwc.DEFAULT_PREC = {'f32': hip.PREC_F32, 'bf16x3': hip.PREC_BF16X3, 'bf16x6': hip.PREC_BF16X6}[args.prec]
    Hh, Ww = args.height, args.width
    assert args.variant == 'fusetrack' or world == 1
    cfg = vps_amd.Config.fromfile(os.path.join(ROOT, 'configs', 'cityscapes', args.variant + '.py'))
    model = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    synth.load_synth(model, args.seed)

    # this rank's shard of the synthetic clip: warmup + steps consecutive frames, resident in HBM before timing
    nfr = args.warmup + args.steps + 1
    base = rank * 100
    frames = [f.to(dev) for f in synth.synth_clip(Hh, Ww, min(nfr, 8), args.seed + rank)]
    metas = [synth.img_meta(Hh, Ww, 10000 * (rank + 1) + t + 1) for t in range(nfr)]

    def step(t):
        img = frames[t % len(frames)]
        ref = frames[(t - 1) % len(frames)] if t else frames[0]
        return model(return_loss=False, rescale=True, img=[img], img_meta=[[metas[t]]], ref_img=[ref])

    C = model.extra_neck.in_channels if model.extra_neck is not None else 0

    def handoff():
        """clip sharding (vps_amd/clip_shard.py): every rank computes the gathered pre-neck feature of its LAST frame first
        and passes it to the next rank with ONE point-to-point send/recv (RCCL over the direct xGMI link); the receiver
        uses it as ref_bsf of its first frame instead of recomputing ResNet+FPN on the previous image."""
        if world == 1:
            return None
        ops, buf, feat = [], None, None
        if rank < world - 1:
            feat = model.gathered_feature(frames[(nfr - 1) % len(frames)])
            ops.append(dist.P2POp(dist.isend, feat, rank + 1))
        if rank > 0:
            buf = torch.empty(1, Hh // 4, Ww // 4, C, dtype=torch.float32, device=dev)
            ops.append(dist.P2POp(dist.irecv, buf, rank - 1))
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        return buf

    t = 0
    handoff()                                       # untimed: RCCL p2p communicator setup
    for _ in range(args.warmup):
        step(t); t += 1
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ndet = 0
    ref_feature = handoff()                         # timed: one hand-off per shard (rank > 0 receives)
    for i in range(args.steps):
        if i == 0 and ref_feature is not None:
            img = frames[t % len(frames)]
            out = model.simple_test(img, [metas[t]], ref_img=[frames[(t - 1) % len(frames)]], ref_feature=ref_feature)
        else:
            out = step(t)
        t += 1
        ndet += int(out[2]['panoptic_cls_inds'].numel())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- instrumented extra frame (outside the timed region): per-stage and per-conv-launch HIP events ----------
    roof, stages = None, None
    if rank == 0:
        model.profile = {}
        nhwc.CONV_TRACE = []
        step(t)
        torch.cuda.synchronize()
        stages = {k: round(v, 3) for k, v in model.stage_times_ms()}
        fl = sum(c[0] for c in nhwc.CONV_TRACE)
        ms = sum(c[1].elapsed_time(c[2]) for c in nhwc.CONV_TRACE)
        nl = len(nhwc.CONV_TRACE)
        abytes = sum(c[4] for c in nhwc.CONV_TRACE)
        if args.conv_table:
            agg = {}
            for c in nhwc.CONV_TRACE:
                a = agg.setdefault(c[3], [0, 0.0, 0.0]); a[0] += 1; a[1] += c[1].elapsed_time(c[2]); a[2] += c[0]
            with open(args.conv_table, 'w') as f:
                f.write('%-58s %5s %9s %9s %8s\n' % ('layer shape', 'calls', 'ms', 'GFLOP', 'TFLOP/s'))
                for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                    f.write('%-58s %5d %9.3f %9.2f %8.2f\n' % (k, a[0], a[1], a[2] / 1e9, a[2] / a[1] / 1e9))
        nhwc.CONV_TRACE = None
        model.profile = None
        ach = fl / (ms * 1e-3) / 1e12
        # achieved = algorithmic FLOPs (2*MAC, unpadded) / summed launch time of the conv kernel.
        # f32 and bf16x6 deliver fp32-grade arithmetic, so they are priced against the fp32 matrix peak (157.3): that is the
        # roofline of an fp32 convolution on this chip. bf16x6 reaches it by executing 6 bf16 MFMA products per fp32 product;
        # the utilisation of the bf16 matrix pipe itself is reported next to it (matrix_pipe_*). bf16x3 is not fp32-grade
        # and is priced against the bf16 peak directly.
        nprod = {'f32': 1, 'bf16x3': 3, 'bf16x6': 6}[args.prec]
        peak = PEAK_BF16_MFMA_TFLOPS if args.prec == 'bf16x3' else PEAK_FP32_MFMA_TFLOPS
        pipe_peak = PEAK_FP32_MFMA_TFLOPS if args.prec == 'f32' else PEAK_BF16_MFMA_TFLOPS
        roof = dict(bound='mfma', kernel='conv_mfma_f32_kernel' if args.prec == 'f32' else 'conv_mfma_bf16{h,p,s}_kernel (vps_conv2d family)',
                    achieved=round(ach, 2), peak=peak, unit='TFLOP/s', frac=round(ach / peak, 4), traffic=None,
                    mfma_products_per_fp32_product=nprod, matrix_pipe_executed_tflops=round(ach * nprod, 1),
                    matrix_pipe_peak=pipe_peak, matrix_pipe_frac=round(ach * nprod / pipe_peak, 4),
                    algorithmic_bytes_per_frame=round(abytes), launches_per_frame=nl, gflop_per_frame=round(fl / 1e9, 1), conv_ms_per_frame=round(ms, 3))

    if rank == 0 and roof is not None:
        # HBM traffic of the conv kernels per frame: measured by separate rocprofv3 --pmc passes (tools/pmc_traffic.py), not
        # collectable from inside this process; the committed measurement is attached when it is for this arithmetic mode
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_pmc_traffic_%s.json' % args.prec)
        if os.path.exists(pmc):
            roof['traffic'] = round(json.load(open(pmc))['conv_hbm_bytes_per_frame'])
            roof['traffic_source'] = 'profiles/' + os.path.basename(pmc) + ' (bytes per frame over all conv launches, like algorithmic_bytes_per_frame)'
    if rank == 0:
        fps = world * args.steps / dt
        line = {
            'metric': 'frames/sec %s 1024x2048' % {'fusetrack': 'FuseTrack', 'fuse': 'PanopticFuse', 'track': 'PanopticTrack'}[args.variant], 'value': round(fps, 3), 'unit': 'frames/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * dt / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'f32': 'f32', 'bf16x6': 'f32-grade: bf16x6 split operands on MFMA, f32 accumulate',
                      'bf16x3': 'bf16x3 split operands on MFMA, f32 accumulate'}[args.prec], 'data': 'synthetic',
            'config': {'workload': '2-frame pair FuseTrack (FlowNet2 + ResNet50-FPN + BFP-TCEA + UPSNet panoptic + track head), '
                                   'synthetic %dx%d clip, batch 1, one clip shard per GPU' % (Hh, Ww),
                       'weights': 'synthetic (vps_amd.synth seed %d)' % args.seed,
                       'detections_per_frame': round(ndet / max(args.steps, 1), 1),
                       'parallelism': 'clip-shard x%d, 1 p2p feature hand-off per shard boundary' % world},
            'roofline': roof, 'stage_ms': stages,
            'stage_ms_note': 'instrumented extra frame on ONE stream; the timed frames overlap FlowNet2 with backbone+FPN and the semantic head with the detection heads on two streams',
        }
        if not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args.seed)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
