"""CPU restatement (NumPy) of the reference's per-frame panoptic post-processing — TEST INFRASTRUCTURE ONLY.

Follows /root/reference/tools/dataset/cityscapes_vps.py:162-226 (`CityscapesVPS.get_unified_pan_result`), the step between
`tools/test_vpq.py:51-63` (device maps -> host) and the PNG / json writers. Pinned against the
real reference function: tests/golden/make_unify_golden.py imports it (easydict / cv2 import shims only) and stores its
outputs in tests/golden/unify_cases.npz; tests/test_postprocess.py checks this restatement against them bit for bit.

The restatement computes the same result from per-id tables instead of per-id boolean masks over the full map, which is
also how the device path (vps_amd/postprocess.py + vps_unify_*) is organised:
  hist[id][c]  = number of pixels with pan == id and seg == c                     (one pass)
  tables       = per pan id: the values written to the three output channels      (<= 245 instances, sequential)
  out[p]       = tables[:, pan[p]]                                                (one pass)
"""
import numpy as np


def dedup_obj_ids(obj_id, max_oid):
    """cityscapes_vps.py:170-181 — all but one occurrence of a repeated object id get fresh ids (max_oid, max_oid+1, ...).
    Returns (new obj_id array, new max_oid). The statements are the reference's, on copies."""
    obj_id = np.array(obj_id).copy()
    oid_unique, oid_cnt = np.unique(obj_id, return_counts=True)
    obj_id_ = obj_id[::-1].copy()
    if np.any(oid_cnt > 1):
        redundants = oid_unique[oid_cnt > 1]
        for red in redundants:
            part = obj_id[obj_id == red]
            for i in range(1, len(part)):
                part[i] = max_oid
                max_oid += 1
            obj_id_[obj_id_ == red] = part
        obj_id = obj_id_[::-1]
    return obj_id, max_oid


def unify_tables(hist, pan_count, cls_ind, obj_id, id_last_stuff, stuff_area_limit):
    """Per-pan-id output values. hist [256,256] int64, pan_count [256] int64. Returns uint8 tables [3,256] (seg, ins, obj).
    cityscapes_vps.py:183-219."""
    seg_t = np.arange(256, dtype=np.int64)          # pan_seg = pan.copy()                      (:183)
    ins_t = np.arange(256, dtype=np.int64)          # pan_ins = pan.copy(); <= id_last_stuff -> 0 (:184, :189)
    obj_t = np.arange(256, dtype=np.int64)          # pan_obj = pan.copy()                      (:185)
    ins_t[:id_last_stuff + 1] = 0
    ids_ins = [i for i in range(id_last_stuff + 1, 256) if pan_count[i] > 0]   # np.unique(pan) > id_last_stuff, ascending (:187-188)
    for idx, pid in enumerate(ids_ins):
        if pid == 255:                                                          # (:192-195)
            seg_t[pid] = 255
            ins_t[pid] = 0
            continue
        row = hist[pid]
        present = np.nonzero(row)[0]                                            # cls, cnt = np.unique(seg[region], return_counts=True)
        cnt = row[present]
        top = present[np.argmax(cnt)]                                           # first maximum in ascending class order
        inst_cls = int(cls_ind[pid - id_last_stuff - 1]) + id_last_stuff
        keep = True
        if top != inst_cls and cnt.max() / cnt.sum() >= 0.5 and top <= id_last_stuff:   # (:203-207)
            keep = False
        if keep:                                                                # (:197-201, :208-212)
            seg_t[pid] = inst_cls
            ins_t[pid] = idx + 1
            if obj_id is not None:
                obj_t[pid] = int(obj_id[idx]) + 1
        else:
            seg_t[pid] = top
            ins_t[pid] = 0
            obj_t[pid] = 0
    # stuff classes smaller than the limit become void, counted on the UPDATED semantic map (:214-219)
    area = np.zeros(256, dtype=np.int64)
    for pid in range(256):
        if pan_count[pid] > 0:
            area[seg_t[pid] & 255] += pan_count[pid]
    small = [c for c in range(0, id_last_stuff + 1) if 0 < area[c] < stuff_area_limit]
    for pid in range(256):
        if (seg_t[pid] & 255) in small:
            seg_t[pid] = 255
    return np.stack([seg_t, ins_t, obj_t]).astype(np.uint8)      # uint8 maps: values wrap modulo 256 like the in-place stores


def unify_frame(seg, pan, cls_ind, obj_id, id_last_stuff=10, stuff_area_limit=4 * 64 * 64):
    """One frame of get_unified_pan_result after the object-id de-duplication. seg, pan: uint8 [H,W]. Returns uint8 [H,W,3]."""
    seg = np.asarray(seg, dtype=np.uint8); pan = np.asarray(pan, dtype=np.uint8)
    hist = np.zeros((256, 256), dtype=np.int64)
    np.add.at(hist, (pan.reshape(-1).astype(np.int64), seg.reshape(-1).astype(np.int64)), 1)
    pan_count = hist.sum(1)
    tables = unify_tables(hist, pan_count, cls_ind, obj_id, id_last_stuff, stuff_area_limit)
    return np.stack([tables[0][pan], tables[1][pan], tables[2][pan]], axis=-1)


def get_unified_pan_result(segs, pans, cls_inds, obj_ids=None, stuff_area_limit=4 * 64 * 64, names=None, id_last_stuff=10):
    """cityscapes_vps.py:162-226: same arguments and return value (dict name -> uint8 [H,W,3]); `max_oid` runs across frames."""
    if obj_ids is None:
        obj_ids = [None for _ in range(len(cls_inds))]
    out = {}
    max_oid = 100
    for seg, pan, cls_ind, obj_id, name in zip(segs, pans, cls_inds, obj_ids, names):
        if obj_id is not None:
            obj_id, max_oid = dedup_obj_ids(obj_id, max_oid)
        out[name] = unify_frame(seg, pan, cls_ind, obj_id, id_last_stuff, stuff_area_limit)
    return out


# ------------------------------------------------------------------------------------------------------------------------
# cityscapes_vps.py:97-159 converter_2ch_track_core: (pan_seg, pan_ins, pan_obj) maps -> colour-coded panoptic PNG arrays +
# COCO-panoptic `segments_info`. `panopticapi` (rgb2id, IdGenerator) is a third-party dependency that is NOT in
# /root/reference: rgb2id is restated from its published source (panopticapi/utils.py: color[0] + 256*color[1] +
# 256*256*color[2]); the colour generator is the caller's object (the reference passes its IdGenerator). PARITY UNPINNED
# for those two; the statements below are the reference's.
# ------------------------------------------------------------------------------------------------------------------------
def rgb2id(color):
    if isinstance(color, np.ndarray) and len(color.shape) == 3:
        if color.dtype == np.uint8:
            color = color.astype(np.int32)
        return color[:, :, 0] + 256 * color[:, :, 1] + 256 * 256 * color[:, :, 2]
    return int(color[0] + 256 * color[1] + 256 * 256 * color[2])


def converter_2ch_track_core(pan_2ch_set, color_generator):
    OFFSET = 1000
    VOID = 255
    annotations, pan_all = [], []
    inst2color = {}
    for idx in range(len(pan_2ch_set)):
        pan_2ch = np.uint32(pan_2ch_set[idx])
        pan = OFFSET * pan_2ch[:, :, 0] + pan_2ch[:, :, 2]
        pan_format = np.zeros((pan_2ch.shape[0], pan_2ch.shape[1], 3), dtype=np.uint8)
        segm_info = {}
        for el in np.unique(pan):
            sem = el // OFFSET
            if sem == VOID:
                continue
            mask = pan == el
            if el % OFFSET > 0:                      # things: one colour per (class, object id) for the whole clip
                if el in inst2color:
                    color = inst2color[el]
                else:
                    color = color_generator.get_color(sem)
                    inst2color[el] = color
            else:
                color = color_generator.get_color(sem)
            pan_format[mask] = color
            index = np.where(mask)
            x = index[1].min(); y = index[0].min()
            width = index[1].max() - x; height = index[0].max() - y
            dt = {"category_id": sem.item(), "iscrowd": 0, "id": int(rgb2id(color)),
                  "bbox": [x.item(), y.item(), width.item(), height.item()], "area": mask.sum().item()}
            segm_info[int(rgb2id(color))] = dt
        pan_all.append(pan_format)
        gt_pan = np.uint32(pan_format)
        pan_gt = gt_pan[:, :, 0] + gt_pan[:, :, 1] * 256 + gt_pan[:, :, 2] * 256 * 256
        labels, labels_cnt = np.unique(pan_gt, return_counts=True)
        for label, area in zip(labels, labels_cnt):
            if label == 0:
                continue
            if label not in segm_info.keys():
                raise KeyError('label not in segm_info keys.')
            segm_info[label]["area"] = int(area)
        annotations.append({"segments_info": [v for k, v in segm_info.items()]})
    return annotations, pan_all
