"""Device-side panoptic post-processing behind the reference's own method signature.

`PanopticUnifier.get_unified_pan_result(segs, pans, cls_inds, obj_ids, stuff_area_limit, names)` takes what
`tools/test_vpq.py:51-63` collects per frame (`fcn_outputs`, `panoptic_outputs`, `panoptic_cls_inds`,
`panoptic_det_obj_ids`) and returns what `tools/dataset/cityscapes_vps.py:162-226` returns — `{name: uint8 [H,W,3]}` with
channels (pan_seg, pan_ins, pan_obj) — but the maps stay on the GPU until the 3-channel result is ready: one histogram pass,
the reference's per-instance decisions on one wavefront, one table-lookup pass (`vps_unify_*`, `csrc/post_ops.hip`).
The only host work is the object-id de-duplication (`:170-181`, <= 100 integers, and stateful across frames).
No CPU path: maps must be (or are uploaded to) device tensors and the HIP library must load."""
import json
import os
from collections import Counter

import numpy as np
import torch

from . import hip


def dedup_obj_ids(obj_id, max_oid):
    """cityscapes_vps.py:170-181: of every repeated object id, the LAST occurrence keeps it and the others get fresh ids
    max_oid, max_oid+1, ... handed out from the end of the list backwards; repeated values are treated in ascending order.
    Returns (ids, max_oid)."""
    orig = [int(v) for v in obj_id]
    work = orig[::-1]                                    # the reference edits a reversed copy
    counts = Counter(orig)
    for red in sorted(v for v, c in counts.items() if c > 1):
        fresh = [red] + [max_oid + i for i in range(counts[red] - 1)]
        max_oid += counts[red] - 1
        where = [i for i, v in enumerate(work) if v == red]
        if len(where) != len(fresh):                     # a fresh id collided with an existing one: numpy raises here too
            raise ValueError('NumPy boolean array indexing assignment cannot assign %d input values to the %d output values '
                             'where the mask is true' % (len(fresh), len(where)))
        for i, v in zip(where, fresh):
            work[i] = v
    return np.asarray(work[::-1], dtype=np.int64), max_oid


# class tables of the datasets the reference's post-processing is configured for: `config.dataset.num_seg_classes / num_classes`
# (stuff classes = ids 0 .. num_seg_classes - num_classes - 1 of the semantic map, things behind them) and the categories the
# evaluation leaves out of the PQ average. tools/dataset/cityscapes_vps.py + configs/cityscapes/test_cityscapes_1gpu.yaml:7-8;
# tools/dataset/viper.py:93-130 (23 / 11: ids 0..12 stuff, 13..22 things + the void instance) and :74-76 (`#### exclude
# "mobilebarrier"`, category 11)
DATASETS = {
    'cityscapes_vps': dict(num_seg_classes=19, num_classes=9, pq_exclude=()),
    'viper': dict(num_seg_classes=23, num_classes=11, pq_exclude=(11,)),
}


class PanopticUnifier:
    def __init__(self, device='cuda', num_seg_classes=19, num_classes=9):
        self.device = torch.device(device)
        self.id_last_stuff = num_seg_classes - num_classes          # config.dataset.* (cityscapes_vps.py:186)
        self.hist = torch.empty(256 * 256, dtype=torch.int32, device=self.device)
        self.pan_count = torch.empty(256, dtype=torch.int32, device=self.device)
        self.tables = torch.empty(3 * 256, dtype=torch.uint8, device=self.device)
        self.status = torch.zeros(1, dtype=torch.int32, device=self.device)

    def _map(self, m):
        t = torch.from_numpy(np.ascontiguousarray(m)) if isinstance(m, np.ndarray) else m
        t = t.to(self.device)
        if t.dtype != torch.uint8:
            t = t.to(torch.uint8)                        # test_vpq.py:53,56 `.astype(np.uint8)`
        return t.contiguous()

    def unify_frame(self, seg, pan, cls_ind, obj_id, stuff_area_limit=4 * 64 * 64):
        """one frame, object ids already de-duplicated. Returns a device uint8 tensor [H,W,3]."""
        lib = hip.load()
        seg, pan = self._map(seg), self._map(pan)
        assert seg.shape == pan.shape and pan.dim() == 2
        npix = pan.numel()
        cls_t = torch.as_tensor(np.asarray(cls_ind.cpu() if torch.is_tensor(cls_ind) else cls_ind), dtype=torch.int32).to(self.device)
        obj_t = None
        if obj_id is not None:
            obj_t = torch.as_tensor(np.asarray(obj_id), dtype=torch.int32).to(self.device)
        out = torch.empty(pan.shape[0], pan.shape[1], 3, dtype=torch.uint8, device=self.device)
        s = hip.stream_ptr()
        hip.check(lib.vps_unify_hist(hip.ptr(pan), hip.ptr(seg), npix, self.id_last_stuff, hip.ptr(self.hist), hip.ptr(self.pan_count), s),
                  'vps_unify_hist')
        hip.check(lib.vps_unify_tables(hip.ptr(self.hist), hip.ptr(self.pan_count), hip.ptr(cls_t) if cls_t.numel() else None,
                                       cls_t.numel(), hip.ptr(obj_t) if (obj_t is not None and obj_t.numel()) else None,
                                       0 if obj_t is None else obj_t.numel(), self.id_last_stuff, int(stuff_area_limit),
                                       hip.ptr(self.tables), hip.ptr(self.status), s), 'vps_unify_tables')
        hip.check(lib.vps_unify_write(hip.ptr(pan), npix, hip.ptr(self.tables), hip.ptr(out), s), 'vps_unify_write')
        self._keep = (seg, pan, cls_t, obj_t)            # alive until the stream has consumed them
        return out

    def get_unified_pan_result(self, segs, pans, cls_inds, obj_ids=None, stuff_area_limit=4 * 64 * 64, names=None):
        if obj_ids is None:
            obj_ids = [None for _ in range(len(cls_inds))]
        results, outs = {}, []
        max_oid = 100
        for seg, pan, cls_ind, obj_id, name in zip(segs, pans, cls_inds, obj_ids, names):
            if obj_id is not None:
                obj_id = obj_id.cpu().numpy() if torch.is_tensor(obj_id) else np.asarray(obj_id)
                obj_id, max_oid = dedup_obj_ids(obj_id, max_oid)
            out = self.unify_frame(seg, pan, cls_ind, obj_id, stuff_area_limit)
            st = int(self.status.item())                 # also orders the reuse of hist / tables by the next frame
            if st:
                raise IndexError('instance id without %s entry (cityscapes_vps.py:%s)' % (('cls_ind', '197') if st == 1 else ('obj_id', '201')))
            outs.append((name, out))
        for name, out in outs:
            results[name] = out.cpu().numpy()
        return results


def _rgb2id(color):
    return int(color[0]) + 256 * int(color[1]) + 256 * 256 * int(color[2])        # panopticapi.utils.rgb2id for one colour


class TrackConverter:
    """Device-side body of `converter_2ch_track_core(proc_id, pan_2ch_set, color_generator)` (cityscapes_vps.py:97-159): same
    result — (annotations, pan_all) — from one statistics pass and one painting pass per frame instead of a boolean mask per
    segment. Colours come from the caller's generator exactly as in the reference (one call per stuff segment and frame, one
    per new (class, object id) pair and clip), in ascending order of 1000*seg + obj."""

    def __init__(self, device='cuda'):
        self.device = torch.device(device)
        self.stats = torch.empty(65536 * 5, dtype=torch.int32, device=self.device)
        self.lut = torch.zeros(65536 * 3, dtype=torch.uint8, device=self.device)

    def convert(self, pan_2ch_set, color_generator):
        lib = hip.load()
        annotations, pan_all = [], []
        inst2color = {}
        for pan_2ch in pan_2ch_set:
            t = torch.from_numpy(np.ascontiguousarray(pan_2ch)) if isinstance(pan_2ch, np.ndarray) else pan_2ch
            t = t.to(self.device).contiguous()
            assert t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == 3
            H, W = int(t.shape[0]), int(t.shape[1])
            hip.check(lib.vps_segment_stats(hip.ptr(t), H, W, hip.ptr(self.stats), hip.stream_ptr()), 'vps_segment_stats')
            st = self.stats.view(65536, 5)
            keys = torch.nonzero(st[:, 0] > 0).flatten()
            rows = st[keys].cpu().numpy()                       # a few dozen present segments
            keys = keys.cpu().numpy()
            seg, obj = keys >> 8, keys & 255
            order = np.argsort(1000 * seg + obj, kind='stable')  # np.unique(1000*seg + obj) ascending
            lut = np.zeros((65536, 3), dtype=np.uint8)
            segm_info, painted = {}, {}
            for i in order:
                sem, ob = int(seg[i]), int(obj[i])
                if sem == 255:
                    continue
                el = 1000 * sem + ob
                if ob > 0:
                    if el in inst2color:
                        color = inst2color[el]
                    else:
                        color = color_generator.get_color(sem)
                        inst2color[el] = color
                else:
                    color = color_generator.get_color(sem)
                lut[keys[i]] = color
                cnt, x0, y0, x1, y1 = (int(v) for v in rows[i])
                sid = _rgb2id(color)
                segm_info[sid] = {"category_id": sem, "iscrowd": 0, "id": sid, "bbox": [x0, y0, x1 - x0, y1 - y0], "area": cnt}
                painted[sid] = painted.get(sid, 0) + cnt    # areas are re-counted per COLOUR (two segments may share one)
            for sid, area in painted.items():
                if sid != 0:
                    segm_info[sid]["area"] = area
            self.lut.copy_(torch.from_numpy(lut.reshape(-1)), non_blocking=False)
            out = torch.empty(H, W, 3, dtype=torch.uint8, device=self.device)
            hip.check(lib.vps_segment_paint(hip.ptr(t), H * W, hip.ptr(self.lut), hip.ptr(out), hip.stream_ptr()), 'vps_segment_paint')
            pan_all.append(out.cpu().numpy())
            annotations.append({"segments_info": [v for k, v in segm_info.items()]})
        return annotations, pan_all


# ------------------------------------------------------------------------------------------------------------------
# Output side of tools/test_vpq.py:194-198 — `inference_panoptic_video` (cityscapes_vps.py:27-94): sample the labelled frames,
# convert the 2-channel maps (TrackConverter above), write pan_2ch / pan_pred PNGs and pred.json.
# ------------------------------------------------------------------------------------------------------------------
class AsyncPngWriter:
    """PNG encoding off the critical path: a small thread pool (zlib releases the GIL) fed with host arrays; `close()` joins.
    The reference encodes inside multiprocessing pools after ALL frames are done (base_dataset.py:434-447); here frame t is
    encoded while frame t+1 runs on the GPU. Same file bytes' CONTENT (PIL `Image.fromarray(image).save(name)`)."""

    def __init__(self, workers=4):
        from concurrent.futures import ThreadPoolExecutor
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.futures = []

    @staticmethod
    def _save(image, name):
        from PIL import Image
        os.makedirs(os.path.dirname(name) or '.', exist_ok=True)
        Image.fromarray(image).save(name)
        return name

    def submit(self, image, name):
        self.futures.append(self.pool.submit(self._save, np.ascontiguousarray(image), name))

    def close(self):
        names = [f.result() for f in self.futures]       # re-raises a worker's exception
        self.pool.shutdown()
        self.futures = []
        return names


def png_name(save_folder, name):
    """cityscapes_vps.py:73 (save_image): output file name of an input image name"""
    return os.path.join(save_folder, name.replace('_leftImg8bit', '').replace('_newImg8bit', '').replace('jpg', 'png').replace('jpeg', 'png'))


def inference_panoptic_video(pred_pans_2ch, output_dir, categories, names, n_video=0, color_generator=None, device='cuda',
                             labeled_fid=20, lambda_=5, nframes_per_video=6, writer=None):
    """`CityscapesVps.inference_panoptic_video` (cityscapes_vps.py:27-94) with the conversion on the device and asynchronous PNG
    writing: same arguments and return value `(pred_pans, pred_json)`, same files (`pan_2ch/`, `pan_pred/`, `pred.json`).
    `pred_pans_2ch`: per-frame uint8 [H,W,3] maps (host arrays or device tensors, e.g. straight from PanopticUnifier);
    `names`: the image file names of the SAMPLED frames (the reference passes them already sampled, test_vpq.py:186-197).
    color_generator: panopticapi's IdGenerator(categories) by default (imported lazily, like the reference); colours are handed
    out per video in the reference's order — one converter state per video, as `np.array_split(.., nprocs)` over whole videos
    gives when nprocs == number of videos."""
    pred_pans_2ch = pred_pans_2ch[(labeled_fid // lambda_)::lambda_]            # only frames with GT annotations (:36)
    if color_generator is None:
        from panopticapi.utils import IdGenerator
        color_generator = IdGenerator({el['id']: el for el in categories})
    own = writer is None
    writer = AsyncPngWriter() if own else writer
    conv = TrackConverter(device)
    annotations, pan_all = [], []
    for v0 in range(0, len(pred_pans_2ch), nframes_per_video):
        chunk = pred_pans_2ch[v0:v0 + nframes_per_video]
        ann, pans = conv.convert(chunk, color_generator)
        for j, (a, pan) in enumerate(zip(ann, pans)):
            i = v0 + j
            annotations.append(a)
            pan_all.append(pan)
            if names is not None:
                two = chunk[j]
                two = two.cpu().numpy() if torch.is_tensor(two) else np.asarray(two)
                writer.submit(two, png_name(os.path.join(output_dir, 'pan_2ch'), names[i]))
                writer.submit(pan, png_name(os.path.join(output_dir, 'pan_pred'), names[i]))
    pred_json = {'annotations': annotations}
    os.makedirs(output_dir, exist_ok=True)
    with open(os.path.join(output_dir, 'pred.json'), 'w') as f:
        json.dump(pred_json, f)
    if own:
        writer.close()
    return pan_all, pred_json
