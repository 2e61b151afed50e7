"""Frame-sharded inference of ONE video clip over the GPUs of a node (one process per GPU, torch.distributed, RCCL/xGMI).

The reference is single-GPU and strictly sequential (tools/test_vpq.py:28-69, `distributed = False` at :106); this is
new design for BASELINE config 4:

  * rank r owns the contiguous frame range partition(nframes, world)[r].
  * every per-frame stage except instance-id assignment depends only on (frame_t, frame_{t-1}): FlowNet2 needs the two
    raw images (input data, each rank loads frame s_r-1 itself), the BFP-TCEA neck needs the gathered pre-neck feature of
    frame t-1 (bfp_tcea.py:117 `ref_bsf = gather(ref_inputs)`).
  * the ONLY data-path exchange: rank r computes gather(FPN(frame e_r-1)) of its LAST frame first (it depends on that
    image alone) and sends it point-to-point to rank r+1 (134 MB fp32 at 1024x2048: one direct xGMI link, ~0.9 ms,
    overlapped with everything rank r+1 does before its neck). One send/recv per shard boundary, no ring, no all-reduce.
  * instance ids depend on the whole history (panoptic_fusetrack.py:400-469): each rank ships every finished frame's detection
    record (boxes, labels, scores, 1024-d track embeddings, kept list: ONE fixed-layout fp32 tensor of ~1 MB) and its two maps
    (one uint8 tensor) to rank 0 point-to-point, as soon as the frame is done; rank 0 runs the greedy assignment frame by frame
    in clip order while the other ranks are still computing — exactly the sequential algorithm, so ids are identical to the
    single-GPU run. No object collectives (nothing is pickled), no compute/replay barrier.

`backend` protocol (product: DetectorBackend below; CPU tests inject an oracle-backed one):
    record_layout() -> [(key, width, dtype)]       columns of the tracker record (the runner's `track_keys`), map_shape() -> (H, W)
    ref_feature(img) -> contiguous tensor          gathered pre-neck feature of a frame
    process(img, ref_img, ref_feature, iid, is_first) -> dict record (no ids yet)
    assign(record, is_first) -> np.ndarray ids     sequential tracker step (stateful)
    finalize(record, ids) -> dict                  per-frame outputs with 'panoptic_det_obj_ids'
A backend with `supports_prefetch` also takes `next_img=` in process(): the next frame of the shard, whose image-only stages
(FlowNet2, ResNet/FPN) it may enqueue behind the current frame's (vps_amd.detector: `prefetch`).
"""
import os

import numpy as np
import torch


def partition(nframes, world):
    """contiguous chunks, sizes differ by at most one, larger chunks first: 30/8 -> 4,4,4,4,4,4,3,3"""
    base, rem = divmod(nframes, world)
    out, s = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((s, s + n))
        s += n
    return out


class ClipShardError(RuntimeError):
    """a peer of the clip pipeline did not deliver (died, stalled) - raised instead of waiting forever"""


class ClipShardRunner:
    """Streams the clip: every rank works through its contiguous shard; a finished frame's detection record (ONE fixed-layout fp32
    tensor) and its two maps (ONE uint8 tensor) go to rank 0 point-to-point as soon as the frame is done; rank 0 assigns the
    track ids frame by frame in clip order — its own frames right away, the others' as their records arrive (it posted the
    receives up front) — and assembles the outputs. No object collectives, no pickling, no barrier between "compute" and
    "replay": the only serial work is the tracker step itself (two small kernels per frame)."""

    # seconds a rank waits for ONE point-to-point operation of a peer (hand-off feature, a frame's record) before it gives up with a
    # ClipShardError naming the peer and the frame - a peer that died or stalled must end the clip with an error, not hang the node
    wait_timeout_s = float(os.environ.get('VPS_CLIP_TIMEOUT_S', '600'))
    recv_window = 3        # positions per peer whose receives rank 0 keeps posted, at least (see `_window`)
    recv_bytes_cap = 2 << 30      # memory rank 0 may hold in posted receive buffers + early-unpacked records of the other ranks

    def __init__(self, backend, rank=0, world=1, dist=None, device=None, track_keys=('det_bboxes', 'det_labels', 'cls_prob', 'emb')):
        self.backend, self.rank, self.world, self.dist = backend, rank, world, dist
        self.device = device
        self.track_keys = track_keys

    def _wait(self, q, what):
        """wait for one posted point-to-point operation, bounded by `wait_timeout_s`"""
        import datetime
        try:
            q.wait(datetime.timedelta(seconds=self.wait_timeout_s))
        except Exception as ex:           # gloo / RCCL raise RuntimeError (timeout, connection closed by a dead peer)
            raise ClipShardError('rank %d: %s did not complete within %.0f s (%s: %s)' % (
                self.rank, what, self.wait_timeout_s, type(ex).__name__, str(ex).splitlines()[0][:200] if str(ex) else '')) from ex

    # ---- fixed record layout: [K, k, keep_inds[cap], per-instance vectors [cap] x3, then the tracker columns [cap, width] ...]
    VEC_KEYS = ('panoptic_cls_inds', 'panoptic_cls_prob', 'panoptic_det_labels')

    def _layout(self):
        cap = int(getattr(self.backend, 'max_det', 256))
        lay = list(self.backend.record_layout())                       # [(key, width, torch dtype)] for self.track_keys
        assert [k for k, _, _ in lay] == list(self.track_keys), (lay, self.track_keys)
        n = 2 + cap * (1 + len(self.VEC_KEYS)) + cap * sum(w for _, w, _ in lay)
        return cap, lay, n

    def _pack(self, rec, out):
        cap, lay, n = self._layout()
        K = int(rec[self.track_keys[0]].shape[0]); keep = torch.as_tensor(np.asarray(rec['keep_inds']).astype(np.int64))
        k = int(keep.numel())
        assert K <= cap and k <= cap, (K, k, cap)
        out[0], out[1] = float(K), float(k)
        o = 2
        out[o:o + k] = keep.to(out.device, torch.float32); o += cap
        for key in self.VEC_KEYS:
            out[o:o + k] = torch.as_tensor(rec[key]).reshape(-1).to(out.device, torch.float32); o += cap
        for key, w, _ in lay:
            out[o:o + K * w] = rec[key].reshape(-1).to(torch.float32); o += cap * w
        return out

    def _unpack(self, buf, maps, t):
        cap, lay, n = self._layout()
        head = buf[:2].cpu()
        K, k = int(head[0]), int(head[1])
        rec = {'t': t}
        o = 2
        rec['keep_inds'] = buf[o:o + k].to(torch.int64).cpu().numpy(); o += cap
        for key, dt in zip(self.VEC_KEYS, (torch.int64, torch.float32, torch.int64)):
            rec[key] = buf[o:o + k].to(dt, copy=True); o += cap      # a copy: `buf` is recycled by the runner (a same-dtype .to() is a view)
        for key, w, dt in lay:
            v = buf[o:o + K * w].to(dt, copy=True)        # own allocation: the kernels need 16-byte aligned operands
            rec[key] = v.reshape(K, w) if (w > 1 or key == 'det_bboxes') else v.reshape(K); o += cap * w
        rec['fcn_outputs'], rec['panoptic_outputs'] = maps[1][None].clone(), maps[0][None].clone()     # `maps` is recycled by the runner
        return rec

    def _post(self, ops):
        """batch_isend_irecv. RCCL sends are ordered behind the current stream by the backend. The gloo backend (functional runs of
        the multi-rank path on one box, CPU tests) hands the raw pointer of a device tensor to the transport, whose host thread reads
        it whenever it gets to it - NOT stream-ordered (measured: the 134 MB hand-off arrived partly stale at 1024x2048 while the
        128x256 run was bitwise right): the producing stream is drained before a send is posted."""
        dist = self.dist
        if dist.get_backend() == 'gloo' and any(op.op is dist.isend and op.tensor.is_cuda for op in ops):
            torch.cuda.current_stream(ops[0].tensor.device).synchronize()
        return dist.batch_isend_irecv(ops)

    def run(self, load_frame, nframes, video_id=1):
        """load_frame(t) -> normalised frame tensor [1,3,H,W] on this rank's device (data loading is not the sharded path).
        Returns on rank 0 the list of per-frame outputs for the WHOLE clip (frame order), elsewhere []."""
        dist, rank, world = self.dist, self.rank, self.world
        # every frame is loaded ONCE and the same tensor object is handed to the backend wherever the frame is needed (as the
        # hand-off frame, as `img`, as the next call's `ref_img`, as the announced `next_img`): the detector matches its
        # prefetched / handed-off work by tensor identity, and a loader that decodes or uploads per call would otherwise never
        # match (every frame's image-only stages would run twice)
        user_load, memo = load_frame, {}

        def load_frame(t):
            if t not in memo:
                memo[t] = user_load(t)
            return memo[t]
        parts = partition(nframes, world)
        s, e = parts[rank]
        be = self.backend
        if e > s and hasattr(user_load, 'set_range'):
            user_load.set_range(max(s - 1, 0), e)        # a read-ahead loader (pipeline.ClipFeeder) stops at the end of this rank's shard
        if hasattr(be, 'inline_ids'):
            be.inline_ids = rank == 0        # rank 0 owns the head of the clip: its frames get their ids inside the detector call
        recv_buf = None
        reqs = []
        # 1) hand-off: last frame's gathered feature -> next rank (computed first: it only needs that image).
        # A clip shorter than the node (nframes < world) leaves the trailing ranks without frames: a send is posted only when
        # the NEXT rank owns frames (it is the one that posts the matching recv), otherwise the last busy rank would wait forever.
        if world > 1 and e > s:
            ops = []
            if rank < world - 1 and parts[rank + 1][1] > parts[rank + 1][0]:
                feat = be.ref_feature(load_frame(e - 1))
                ops.append(dist.P2POp(dist.isend, feat, rank + 1))
            if rank > 0:
                recv_buf = be.ref_feature_buffer(load_frame(s))
                ops.append(dist.P2POp(dist.irecv, recv_buf, rank - 1))
            reqs = self._post(ops) if ops else []
            if rank > 0 and hasattr(be, 'prime'):
                # the shard's first frame has no predecessor on this rank whose call could announce it: its image-only stages are
                # enqueued now, beside the hand-off, not behind the wait for it
                be.prime(load_frame(s), load_frame(s - 1))
        # 2) rank 0 posts the receives of every other rank's records + maps: per peer in the order of its sends (frame order), and
        # across peers by position in the shard, not peer by peer — the ranks finish their j-th frames at about the same time, and on
        # RCCL the point-to-point operations of a rank run in posting order on one communicator stream: posted peer by peer, the
        # sends of ranks 2..N-1 would spin (on a few CUs of their GPU) until rank 1's whole shard had arrived
        # Only a WINDOW of positions is posted at a time (`recv_window` per peer, buffers recycled as step 4 consumes them): a long
        # video would otherwise pin ~5 MB per remote frame on rank 0 and queue O(clip) receive kernels on the communicator stream
        # (ADVICE r3). A peer's sends complete in order, so its position j + window is posted when its position j has been consumed.
        inbox, pool, stash = {}, [], {}
        post_recv = None
        if world > 1 and rank == 0:
            cap, lay, n = self._layout()
            dev = self.device if self.device is not None else load_frame(0).device
            Hm, Wm = be.map_shape()

            def post_recv(r, j):
                t = parts[r][0] + j
                if t >= parts[r][1]:
                    return
                if pool:
                    buf, maps = pool.pop()
                    buf.zero_()
                else:
                    buf = torch.zeros(n, dtype=torch.float32, device=dev)
                    maps = torch.empty(2, Hm, Wm, dtype=torch.uint8, device=dev)
                inbox[t] = (buf, maps, self._post([dist.P2POp(dist.irecv, buf, r), dist.P2POp(dist.irecv, maps, r)]))
            # window = as many positions per peer as the memory cap allows (a record + its maps are ~5 MB), at least `recv_window`: a
            # shard that fits is posted completely (no send of a peer ever waits for its receive: an unmatched RCCL send spins on CUs
            # of the sender beside its compute); a longer one is topped up while rank 0 works through its own shard (step 3) and in step 4
            per_frame = 4 * n + 2 * Hm * Wm
            # `recv_bytes_cap` bounds what rank 0 holds for the other shards ALTOGETHER: posted receive buffers plus the records that
            # were unpacked early into the stash (the two used to be budgeted separately, i.e. up to twice the cap). Each
            # peer keeps at least `recv_window` positions posted whatever the cap says (its sends complete in order: progress).
            budget = max(int(self.recv_bytes_cap // per_frame), 1)
            window = max(self.recv_window, budget // max(world - 1, 1))
            posted = {r: 0 for r in range(1, world)}          # positions posted so far, per peer
            taken = {r: 0 for r in range(1, world)}           # positions unpacked so far, per peer
            self.max_posted = 0
            self.max_held = 0                                  # posted + stashed frames, high-water mark (tests)
            held = lambda: sum(posted[r] - taken[r] for r in posted) + len(stash)

            def fill():
                # round robin over the peers, lowest position first: on RCCL a rank's point-to-point operations run in posting order
                more = True
                while more:
                    more = False
                    ok = lambda r: (posted[r] < parts[r][1] - parts[r][0] and posted[r] - taken[r] < window and
                                    (held() < budget or posted[r] - taken[r] < self.recv_window))
                    lo = min((posted[r] for r in posted if ok(r)), default=None)
                    for r in range(1, world):
                        if posted[r] == lo and ok(r):
                            post_recv(r, lo)
                            posted[r] = lo + 1
                            more = True
                self.max_posted = max(self.max_posted, max(posted[r] - taken[r] for r in posted))
                self.max_held = max(self.max_held, held())
            fill()
        # 3) this rank's frames
        outs, sent = [], []
        prev = None
        for t in range(s, e):
            img = load_frame(t)
            is_first = t == 0
            ref_img = img if is_first else (prev if prev is not None else load_frame(t - 1))
            ref_feature = None
            if t == s and rank > 0:
                for rq in reqs:
                    self._wait(rq, 'the feature hand-off with rank %d / %d (frame %d)' % (rank - 1, rank + 1, s - 1))
                reqs = []
                ref_feature = recv_buf
                self.last_ref_feature = recv_buf        # (tests compare it with the sender's gathered feature)
            if getattr(be, 'supports_prefetch', False):
                # the next frame of this shard is known: its image-only stages are enqueued behind this frame's (detector.simple_test)
                nxt = load_frame(t + 1) if t + 1 < e else None
                if getattr(be, 'prefetch_depth', 1) >= 2:
                    # the frame after next is announced too: the detector enqueues its image-only stages before it blocks on this frame's
                    # end-of-frame read, so the GPU never runs out of queued work at a frame boundary
                    rec = be.process(img, ref_img, ref_feature, video_id * 10000 + t + 1, is_first, next_img=nxt,
                                     next2_img=load_frame(t + 2) if t + 2 < e else None)
                else:
                    rec = be.process(img, ref_img, ref_feature, video_id * 10000 + t + 1, is_first, next_img=nxt)
            else:
                rec = be.process(img, ref_img, ref_feature, video_id * 10000 + t + 1, is_first)
            rec['t'] = t
            if rank == 0:
                outs.append(be.finalize(rec, be.assign(rec, is_first)))       # rank 0 owns the head of the clip: ids right away
            else:
                cap, lay, n = self._layout()
                buf = self._pack(rec, torch.zeros(n, dtype=torch.float32, device=rec[self.track_keys[0]].device))
                maps = torch.stack([torch.as_tensor(rec['panoptic_outputs'])[0], torch.as_tensor(rec['fcn_outputs'])[0]]).to(torch.uint8).contiguous()
                sent.append((buf, maps, self._post([dist.P2POp(dist.isend, buf, 0), dist.P2POp(dist.isend, maps, 0)])))
            prev = img
            memo.pop(t - 1, None)            # frame t-1 is no longer needed (frame t stays: it is frame t+1's reference)
            if post_recv is not None:
                # records that have arrived meanwhile are unpacked (copied out of the pooled buffers) and their buffers reposted, so a
                # peer whose shard is longer than the window is never left with unmatched sends until step 4
                for r in range(1, world):
                    while taken[r] < posted[r]:
                        tt = parts[r][0] + taken[r]
                        buf, maps, rq = inbox[tt]
                        if not all(q.is_completed() for q in rq):
                            break
                        for q in rq:
                            self._wait(q, 'the record of frame %d from rank %d' % (tt, r))             # completed: returns at once, and orders the unpack behind the receive on every backend
                        inbox.pop(tt)
                        stash[tt] = self._unpack(buf, maps, tt)
                        pool.append((buf, maps))
                        taken[r] += 1
                fill()
        for rq in reqs:
            self._wait(rq, 'the feature hand-off to rank %d' % (rank + 1))
        # 4) rank 0: the other shards' frames in clip order, each as soon as its record has arrived
        if rank == 0:
            for r in range(1, world):
                for j in range(parts[r][1] - parts[r][0]):
                    t = parts[r][0] + j
                    if t in stash:                         # arrived and unpacked while rank 0 was still computing
                        rec = stash.pop(t)
                    else:
                        buf, maps, rq = inbox.pop(t)
                        for q in rq:
                            self._wait(q, 'the record of frame %d from rank %d' % (t, r))
                        rec = self._unpack(buf, maps, t)       # copies everything it keeps: the buffers go back to the pool
                        pool.append((buf, maps))
                        taken[r] += 1
                        fill()
                    outs.append(be.finalize(rec, be.assign(rec, t == 0)))
            return outs
        for buf, maps, rq in sent:
            for q in rq:
                self._wait(q, 'sending a frame record to rank 0')
        return []


class DetectorBackend:
    """adapts vps_amd.detector.PanopticFuseTrack to the ClipShardRunner protocol"""

    supports_prefetch = True
    prefetch_depth = int(os.environ.get('VPS_PREFETCH_DEPTH', '2'))      # 2: the frame after next is announced as well (see ClipShardRunner.run)
    inline_ids = True

    def __init__(self, detector, H, W, prefetch=True):
        self.det, self.H, self.W = detector, H, W
        self.prefetch = prefetch
        # the runner hands frame t-1 to frame t as its reference by construction: the probe check is skipped per call (the
        # detector's own setting is restored after every call, the model object may be shared with other callers)

    max_det = 256

    def record_layout(self):
        E = self.det.track_head.fcs[1].out_features
        return [('det_bboxes', 4, torch.float32), ('det_labels', 1, torch.int64), ('cls_prob', 1, torch.float32), ('emb', E, torch.float32)]

    def map_shape(self):
        return self.H, self.W

    def ref_feature(self, img):
        """gathered pre-neck feature of the shard's last frame, for the hand-off to the next rank. f16x3: the ResNet + FPN pass behind
        it reports its fp16 range like every frame does, but no end-of-frame read follows before the send - so the report is read
        here (one 4-byte D2H ahead of the isend): an overflowed layer is switched to bf16x6 and the feature is computed again, the
        receiver never sees an fp16-overflowed `ref_bsf`"""
        from . import hip, nhwc
        for _ in range(4):
            feat = self.det.gathered_feature(img)
            if nhwc.DEFAULT_PREC != hip.PREC_F16X3 and nhwc._F16_NEXT[0] <= 1:
                return feat
            if int(nhwc.f16_status(img.device).amax().item()) == 0:
                return feat
            nhwc.f16_fallback(img.device)
            self.det._handoff = None
        raise hip.VpsHipError('f16x3: the hand-off feature still overflows the fp16 range after three rounds of per-layer bf16x6 fallback')

    def ref_feature_buffer(self, img):
        C = self.det.extra_neck.in_channels
        return torch.empty(1, self.H // 4, self.W // 4, C, dtype=torch.float32, device=img.device)

    def process(self, img, ref_img, ref_feature, iid, is_first, next_img=None, next2_img=None):
        from . import synth
        meta = synth.img_meta(self.H, self.W, iid)
        pf = None
        if self.prefetch and next_img is not None:
            pf = [(next_img, img)]                                # the next frame's reference is this frame
            if next2_img is not None:
                pf.append((next2_img, next_img))
        keep, self.det.verify_ref_frame = self.det.verify_ref_frame, False
        try:
            out = self.det.simple_test(img, [meta], ref_img=[ref_img], ref_feature=ref_feature, defer_tracking=not self.inline_ids, prefetch=pf)
        finally:
            self.det.verify_ref_frame = keep
        rec = dict(out[2])
        rec.update(self.det._track_record)
        if self.inline_ids:
            rec['ids'] = self.det._aux['det']['det_obj_ids']        # assigned inside the call (read with its end-of-frame read)
        return rec

    def prime(self, img, ref_img):
        if self.prefetch and os.environ.get('VPS_NO_PRIME', '0') == '0':
            self.det.prime(img, ref_img)

    def assign(self, rec, is_first):
        if 'ids' in rec:
            return rec['ids']
        return self.det.track_assign(rec, is_first)

    def finalize(self, rec, ids):
        keep = rec['keep_inds']
        out = {k: rec[k] for k in ('fcn_outputs', 'panoptic_outputs', 'panoptic_cls_inds', 'panoptic_cls_prob',
                                   'panoptic_det_labels')}
        out['panoptic_det_obj_ids'] = np.asarray(ids)[np.asarray(keep)]
        out['t'] = rec['t']
        return out


def predict_clip_time(nframes, world, t_frame, t_first, t_handoff, t_xfer, t_assign, primed=True):
    """Critical-path estimate of `ClipShardRunner.run` on `world` GPUs from single-GPU stage times (seconds): a prediction to judge
    the first real multi-GPU run against (no such run exists from this build: one GPU per box). Per rank r with n_r frames:

        sender part    t_handoff   ResNet + FPN + gather of the shard's LAST frame, first thing on every rank but the last (its
                                   levels are reused by that frame's own call, so only the early start costs: it delays frame 0)
        wait           the shard's first neck needs the previous rank's feature: it arrives at t_handoff + t_xfer; the first
                       frame's image-only stages run meanwhile when `primed` (DetectorBackend.prime), else behind the wait
        frames         t_first (a frame whose image-only stages had no frame to hide behind) + (n_r - 1) * t_frame
        rank 0         after its own shard, one tracker step (t_assign) per remote frame, each as soon as the record is in

    -> dict(seconds, frames_per_s, per_rank_finish, critical). t_first - t_frame is what the pipelining hides in steady state."""
    parts = partition(nframes, world)
    n = [b - a for a, b in parts]
    fin = []
    for r in range(world):
        if n[r] == 0:
            fin.append(0.0)
            continue
        sends = r < world - 1 and n[r + 1] > 0
        start = t_handoff if sends else 0.0                   # this rank's own hand-off work comes first on its streams
        if r > 0:
            arrive = t_handoff + t_xfer                         # every sender starts at 0: all hand-offs arrive at about the same time
            image_only = max(t_first - t_frame, 0.0) + 0.7 * t_frame     # share of a frame that needs the images alone (~16 of 22 ms)
            if primed:
                first_done = max(start + image_only, arrive) + (t_first - image_only)
            else:
                first_done = max(start, arrive) + t_first
        else:
            first_done = start + t_first
        fin.append(first_done + (n[r] - 1) * t_frame)
    # rank 0 replays the remote frames in clip order: frame j of rank r is available when that rank has finished it
    t = fin[0]
    for r in range(1, world):
        for j in range(n[r]):
            avail = fin[r] - (n[r] - 1 - j) * t_frame
            t = max(t, avail) + t_assign
    total = max(t, max(fin))
    return dict(seconds=total, frames_per_s=nframes / total, per_rank_finish=[round(x, 5) for x in fin],
                critical='rank 0 replay' if t >= max(fin) else 'rank %d compute' % int(np.argmax(fin)))
