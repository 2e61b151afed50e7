"""Device-side tail of the reference's test pipeline.

`configs/cityscapes/fusetrack.py:176-191` runs, per frame and per image of the (img, ref_img) pair, on 2 CPU workers:
LoadRefImageFromFile -> Resize(keep_ratio, (2048,1024)) -> RandomFlip(off) -> Normalize -> Pad(32) -> ImageToTensor -> Collect,
then ships two fp32 tensors (2 x 25 MB at 1024x2048) to the GPU. Here the decoded uint8 image is uploaded as it is (6 MB)
and `Normalize -> Pad -> ImageToTensor` is one kernel (`vps_image_prep`); `PairFeeder` keeps the previous frame's tensor on
the device, because frame t's `ref_img` IS frame t-1's `img` (`tools/dataset/cityscapes_vps.py:140`: the first frame of a
video is its own reference) — the reference decodes and normalises every image twice.

Resize: at the dataset's native 1024x2048 the keep-ratio rescale to (2048, 1024) has scale factor 1.0 and copies the image;
any other size goes through `vps_resize_u8`, OpenCV's 8-bit fixed-point bilinear (what mmcv.imrescale -> cv2.resize computes on
the decoded image; restated from the published resize.cpp, no cv2 here to pin it). Image decoding itself stays on the host:
`imread` / `LoadRefImageFromFile` below mirror `mmcv.imread` (BGR uint8) and `datasets/pipelines/loading.py:33-68` with PIL (PNG is
lossless, so the arrays equal cv2's for the datasets of the path) and decode every file ONCE — frame t's `ref_filename` is frame
t-1's `filename`. No CPU path for the transforms: the HIP library must load."""
import ctypes
import os.path as osp

import warnings

import numpy as np
import torch

from . import hip


class DeviceImagePrep:
    """Normalize(mean, std, to_rgb) + Pad(size_divisor, pad_val) + ImageToTensor, same constructor keywords as the reference
    transforms (`transforms.py:228-236`, `:300-303`). __call__(results) mirrors their effect on the `results` dict."""

    def __init__(self, mean, std, to_rgb=True, size_divisor=32, pad_val=0, img_scale=(2048, 1024), keep_ratio=True, device='cuda'):
        self.mean = np.array(mean, dtype=np.float32)
        self.std = np.array(std, dtype=np.float32)
        self.to_rgb = to_rgb
        self.size_divisor = size_divisor
        self.pad_val = pad_val
        self.img_scale = img_scale
        self.keep_ratio = keep_ratio
        self.device = torch.device(device)
        self._mean_c = (ctypes.c_float * 3)(*[float(v) for v in self.mean])
        self._std_c = (ctypes.c_float * 3)(*[float(v) for v in self.std])

    def scale_factor(self, h, w):
        """mmcv 0.2.14 imrescale: min(long_edge / max(h, w), short_edge / min(h, w)) for a (long, short) scale tuple."""
        max_long, max_short = max(self.img_scale), min(self.img_scale)
        return min(max_long / max(h, w), max_short / min(h, w))

    @staticmethod
    def _linear_table(dst, src):
        """resize.cpp, INTER_LINEAR: per destination index (source index, 2048*(1-f), 2048*f) with f in float like OpenCV"""
        d = np.arange(dst, dtype=np.float64)
        f = ((d + 0.5) * (float(src) / float(dst)) - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        lo = s < 0
        f[lo] = 0; s[lo] = 0
        hi = s >= src - 1
        f[hi] = 0; s[hi] = src - 1
        tab = np.stack([s, np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64), np.rint(f * np.float32(2048)).astype(np.int64)], 1)
        return torch.from_numpy(np.ascontiguousarray(tab.astype(np.int32)))

    def resize(self, t, new_w, new_h):
        """device uint8 [H,W,3] -> [new_h,new_w,3], cv2.resize(.., INTER_LINEAR) arithmetic"""
        H, W = int(t.shape[0]), int(t.shape[1])
        key = (H, W, new_h, new_w)
        tabs = self.__dict__.setdefault('_tabs', {})
        if key not in tabs:
            tabs[key] = (self._linear_table(new_w, W).to(self.device), self._linear_table(new_h, H).to(self.device))
        xt, yt = tabs[key]
        out = torch.empty(new_h, new_w, 3, dtype=torch.uint8, device=self.device)
        hip.check(hip.load().vps_resize_u8(hip.ptr(t), H, W, hip.ptr(out), new_h, new_w, 3, hip.ptr(xt), hip.ptr(yt), hip.stream_ptr()),
                  'vps_resize_u8')
        return out

    def prep(self, img):
        """uint8 [H,W,3] (numpy as cv2.imread returns it, or a device tensor) -> device fp32 [3,Hp,Wp]"""
        t = torch.from_numpy(np.ascontiguousarray(img)) if isinstance(img, np.ndarray) else img
        assert t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == 3, 'decoded uint8 HWC image expected'
        H, W = int(t.shape[0]), int(t.shape[1])
        sf = self.scale_factor(H, W)
        t = t.to(self.device).contiguous()
        nw, nh = int(W * float(sf) + 0.5), int(H * float(sf) + 0.5)              # mmcv 0.2.14 _scale_size
        if (nw, nh) != (W, H):
            src = t
            t = self.resize(t, nw, nh)
            self._keep_src = src
            H, W = nh, nw
        d = self.size_divisor
        Hp, Wp = (H + d - 1) // d * d, (W + d - 1) // d * d
        out = torch.empty(3, Hp, Wp, dtype=torch.float32, device=self.device)
        hip.check(hip.load().vps_image_prep(hip.ptr(t), H, W, Hp, Wp, self._mean_c, self._std_c, 1 if self.to_rgb else 0,
                                            float(self.pad_val), hip.ptr(out), hip.stream_ptr()), 'vps_image_prep')
        self._keep = t
        return out, (H, W, 3), (Hp, Wp, 3), sf

    def __call__(self, results):
        els = ['ref_img', 'img'] if 'ref_img' in results else ['img']          # transforms.py:108, :261
        for el in els:
            out, img_shape, pad_shape, sf = self.prep(results[el])
            results[el] = out
        results['img_shape'] = img_shape                                         # transforms.py:118-121
        results['pad_shape'] = pad_shape                                         # :270
        results['scale_factor'] = sf
        results['keep_ratio'] = self.keep_ratio
        results['pad_fixed_size'] = None
        results['pad_size_divisor'] = self.size_divisor
        results['img_norm_cfg'] = dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb)   # :316-317
        return results


class PairFeeder:
    """Feeds (img, ref_img) pairs of one video in frame order, preparing every decoded image once: the reference of frame t
    is the prepared tensor of frame t-1, the first frame is its own reference (cityscapes_vps.py:140)."""

    def __init__(self, prep):
        self.prep = prep
        self.prev = None

    def reset(self):
        self.prev = None

    def __call__(self, img_u8):
        cur = self.prep.prep(img_u8)[0]
        ref = cur if self.prev is None else self.prev
        self.prev = cur
        return cur.unsqueeze(0), ref.unsqueeze(0)            # [1,3,Hp,Wp] each, what model(img=[..], ref_img=[..]) takes


class ClipFeeder:
    """`load_frame(t)` of a clip from image FILES, for `ClipShardRunner.run` and test_vpq-style loops (the
    reference's side of it is `datasets/pipelines/loading.py:43-68` on `workers_per_gpu=2` loader processes, `configs/cityscapes/
    fusetrack.py:193-194`, each decoding and normalising BOTH images of every pair).

    `workers` decode THREADS run ahead of the consumer, every file ONCE: a worker reads the file into a reusable buffer and the
    library's own PNG decoder (`csrc/png_host.cpp`, a C-ABI call that runs without the interpreter lock) writes the BGR frame
    STRAIGHT INTO A PINNED staging buffer of a small ring - no per-frame host allocation, no copy on the consumer's thread. The
    consumer uploads the uint8 frame (6 MB instead of the 25 MB fp32 tensor) and `DeviceImagePrep.prep` makes the normalised,
    padded fp32 tensor on the device. Files the native decoder does not take (JPEG, 16-bit / palette PNG) go through `imread` and one
    copy into the staging buffer. What did NOT work (measured on the GPU box): PIL decode threads - they hold the interpreter lock
    for long stretches and beside a main thread that launches ~560 kernels per frame deliver 6 frames/s; forked decode processes -
    the fork of a process that maps 30 GB of device memory stalls (1.4 frames/s). A prepared frame is kept until the consumer asks
    for a frame two positions later (frame t is frame t+1's reference), so the same tensor OBJECT serves as `img` of frame t and
    `ref_img` of frame t+1 - what the detector's prefetch / hand-off matching by tensor identity needs.

        feeder = ClipFeeder(files, prep, workers=4)
        outs = ClipShardRunner(DetectorBackend(model, H, W)).run(feeder, len(files))
        feeder.close()
    """

    def __init__(self, files, prep, workers=4, ahead=None):
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        self.files, self.prep = list(files), prep
        self.workers = int(workers)
        self.ahead = int(ahead) if ahead is not None else 2 * self.workers      # decoded frames in flight beyond the consumer
        self._pool = ThreadPoolExecutor(self.workers)
        self._pending = {}          # t -> (future, staging slot)
        self._ready = {}            # t -> prepared device tensor [1,3,Hp,Wp]
        self._meta = {}             # t -> shapes of the prepared frame (img_meta entries)
        self._done = set()          # frames delivered so far: the window does not decode them a second time (consumers keep what they got)
        self._next = 0              # first index not yet submitted
        self._hi = len(self.files)  # the window never runs past this frame (`set_range`: the end of a rank's shard)
        self.decodes = 0
        self.out_of_window = 0      # requests that found their frame neither prepared nor in flight (decoded on the spot)
        self.stats = dict(wait_for_decode_s=0.0, upload_prep_s=0.0)      # where the consumer's time in __call__ went
        self._stage, self._fbuf, self._events = [], [], []
        self._free = deque()        # staging slots, recycled FIFO: the slot taken next is the one whose upload is oldest

    def __len__(self):
        return len(self.files)

    def set_range(self, lo, hi):
        """the consumer will ask for frames of [lo, hi) only (ClipShardRunner: a rank's shard + its reference frame): the read-ahead
        stops at hi instead of decoding frames another rank owns"""
        self._hi = min(int(hi), len(self.files))

    def _slots(self, nbytes):
        """the staging ring: `ahead + 2` pinned frame buffers (plain host memory without a device) + as many file buffers"""
        n = self.ahead + 2
        pin = self.prep.device.type == 'cuda'
        self._stage = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() if pin else torch.empty(nbytes, dtype=torch.uint8) for _ in range(n)]
        self._fbuf = [bytearray(0) for _ in range(n)]
        self._free.clear(); self._free.extend(range(n))
        self._events = [None] * n

    def _decode(self, path, slot):
        """worker thread: file -> staging slot; returns the frame's (H, W)"""
        lib = hip.load_host()
        size = osp.getsize(path)
        if len(self._fbuf[slot]) < size:
            self._fbuf[slot] = bytearray(size + (size >> 2))
        buf = self._fbuf[slot]
        with open(path, 'rb', buffering=0) as f:
            n = f.readinto(memoryview(buf)[:size])
        stage = self._stage[slot]
        cbuf = (ctypes.c_char * len(buf)).from_buffer(buf)
        H, W, C = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        if str(path).lower().endswith('.png') and lib.vps_png_info(cbuf, n, ctypes.byref(H), ctypes.byref(W), ctypes.byref(C)) == 0 \
                and H.value * W.value * 3 <= stage.numel():
            hip.check(lib.vps_png_decode_bgr8(cbuf, n, ctypes.c_void_p(stage.data_ptr()), stage.numel()), 'vps_png_decode_bgr8')
            return H.value, W.value
        img = imread(path)                         # the general decoder (cv2 / PIL) + one copy
        assert img.nbytes <= stage.numel(), 'frame larger than the staging buffer (%d > %d bytes)' % (img.nbytes, stage.numel())
        stage[:img.nbytes].copy_(torch.from_numpy(img).reshape(-1))
        return img.shape[0], img.shape[1]

    def _take_slot(self):
        slot = self._free.popleft()
        if self._events[slot] is not None:
            self._events[slot].synchronize()                                  # the upload that last read this slot has finished
            self._events[slot] = None
        return slot

    def _submit_until(self, t_hi):
        if not self._stage and self.files:
            from PIL import Image
            with Image.open(self.files[0]) as im:                                # header only: the frame size of the clip
                w, h = im.size
            self._slots(h * w * 3)
        t_hi = min(t_hi, self._hi)
        while self._next < t_hi:
            t = self._next
            if t in self._done or t in self._ready or t in self._pending:         # delivered / in flight already (a one-off request ahead of the window)
                self._next += 1
                continue
            if not self._free:
                break
            slot = self._take_slot()
            self._pending[t] = (self._pool.submit(self._decode, self.files[t], slot), slot)
            self._next += 1

    def start(self, t=0):
        """begin decoding the window that starts at frame t now (a loader in front of a running pipeline is `ahead` frames ahead)"""
        self._next = max(self._next, t)
        self._submit_until(t + 1 + self.ahead)
        return self

    def meta(self, t):
        """img_meta entries of frame t that depend on the file (`Collect` meta keys, datasets/pipelines/formating.py:185-186:
        filename, ori_shape, img_shape, pad_shape, scale_factor, flip); the shapes are known once the frame has been delivered"""
        return dict(self._meta.get(t, {}), filename=self.files[t])

    def _rewind(self, t):
        """a request BEHIND the window (ClipShardRunner asks for the hand-off frame e-1 first, then works through s .. e-1): the window
        restarts at t. Decodes in flight beyond the new window give their staging slots back (cancelled, or waited for when a worker
        already has them) - they would otherwise sit on the ring until the consumer gets there, and the frames in between would
        all be decoded on the spot (every shard frame took the out-of-window path)."""
        for u in [u for u in self._pending if u > t + self.ahead]:
            fut, slot = self._pending.pop(u)
            if not fut.cancel():
                try:
                    fut.result()
                except Exception:
                    pass
            self._free.append(slot)
        self._next = t
    def __call__(self, t):
        import time
        if t in self._ready:
            return self._ready[t]
        if t not in self._pending:
            if t >= self._next:                   # ahead of everything submitted (a shard that starts mid-clip): the window starts there
                self._next = t
            else:                                 # behind the window: it restarts at t
                self._rewind(t)
        self._submit_until(t + 1 + self.ahead)
        ent = self._pending.pop(t, None)
        if ent is None:                           # delivered before and dropped since, or the ring is full: decode it now
            self.out_of_window += 1
            if not self._free:
                raise RuntimeError('ClipFeeder: frame %d requested outside the decode window with every staging buffer in flight' % t)
            slot = self._take_slot()
            ent = (self._pool.submit(self._decode, self.files[t], slot), slot)
        fut, slot = ent
        c0 = time.perf_counter()
        H, W = fut.result()
        c1 = time.perf_counter()
        self.stats['wait_for_decode_s'] += c1 - c0
        self.decodes += 1
        src = self._stage[slot][:H * W * 3].view(H, W, 3)
        dev = self.prep.device
        if dev.type == 'cuda':
            d = src.to(dev, non_blocking=True)
            ev = torch.cuda.Event(); ev.record()
            self._events[slot] = ev
        else:
            d = src.numpy().copy()                 # host stand-in (tests): own copy, the slot goes back to the ring
        out, img_shape, pad_shape, sf = self.prep.prep(d)
        out = out.unsqueeze(0)
        self._meta[t] = dict(img_shape=tuple(img_shape), pad_shape=tuple(pad_shape), scale_factor=sf, ori_shape=(H, W, 3), flip=False)
        self.stats['upload_prep_s'] += time.perf_counter() - c1
        self._free.append(slot)                    # to the BACK of the ring: it is reused after every other free slot
        self._done.add(t)
        self._submit_until(t + 1 + self.ahead)
        self._ready[t] = out
        for old in [k for k in self._ready if k < t - 1]:      # frame t-1 stays: it is frame t's reference
            del self._ready[old]
        return out

    def close(self):
        for fut, slot in self._pending.values():
            fut.cancel()
        self._pool.shutdown(wait=True)
        self._pending = {}


def png_decode(data):
    """bytes of an 8-bit non-interlaced RGB / RGBA / grey PNG -> BGR uint8 [H,W,3] through the library's host decoder
    (csrc/png_host.cpp: zlib inflate + un-filter, no interpreter lock while it runs); None for any other flavour"""
    lib = hip.load_host()
    buf = (ctypes.c_char * len(data)).from_buffer_copy(data) if not isinstance(data, (bytearray, memoryview)) else (ctypes.c_char * len(data)).from_buffer(data)
    H, W, C = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    if lib.vps_png_info(buf, len(data), ctypes.byref(H), ctypes.byref(W), ctypes.byref(C)) != 0:
        return None
    out = np.empty((H.value, W.value, 3), dtype=np.uint8)
    hip.check(lib.vps_png_decode_bgr8(buf, len(data), out.ctypes.data_as(ctypes.c_void_p), out.nbytes), 'vps_png_decode_bgr8')
    return out


def imread(path, native=True):
    """mmcv.imread(path) / cv2.imread(path, IMREAD_COLOR): uint8 [H,W,3] in BGR order; grey images are replicated to three
    channels, an alpha channel is dropped. 8-bit PNGs (the Cityscapes-VPS frames) go through the library's own decoder, which
    runs without the interpreter lock (decode threads scale beside a busy main thread); everything else through cv2 / PIL."""
    if native and str(path).lower().endswith('.png'):
        with open(path, 'rb') as f:
            data = f.read()
        img = png_decode(bytearray(data))
        if img is not None:
            return img
    try:
        import cv2                          # the reference's own decoder when the host has it
        if not (callable(getattr(cv2, 'imread', None)) and getattr(cv2, '__version__', None)):
            raise ImportError('a cv2 stand-in without imread (import shims of the golden-vector scripts)')
        img = cv2.imread(path, cv2.IMREAD_COLOR)
        if img is None:
            raise IOError('cv2.imread failed: %s' % path)
        return img
    except ImportError:
        pass
    import io
    from PIL import Image
    # The file is read in one piece and handed to the decoder in ONE call (`decodermaxblock`): PIL's load loop otherwise feeds the
    # inflater 64 KB at a time from Python, and a decode THREAD then re-acquires the interpreter lock ~60 times per 1024x2048 frame -
    # each time waiting for a main thread that is busy launching kernels to give it up (ClipFeeder: 26 frames/s instead of 46)
    with open(path, 'rb') as f:
        data = f.read()
    with Image.open(io.BytesIO(data)) as im:
        # PIL equals cv2.imread bit for bit only for 8-bit PNG (lossless; the Cityscapes-VPS frames). A 16-bit PNG is not scaled
        # like cv2 does -> refused; a JPEG (VIPER) decodes with another IDCT / EXIF handling -> decoded, with a warning
        if im.mode in ('I;16', 'I;16B', 'I', 'F'):
            raise ValueError('imread: %s is not an 8-bit image (mode %s); cv2 is needed for cv2.imread semantics' % (path, im.mode))
        if im.format != 'PNG':
            warnings.warn('imread: %s decoded with PIL; pixels can differ from cv2.imread for lossy formats (%s)' % (path, im.format))
        im.decodermaxblock = len(data) + 1
        rgb = np.asarray(im.convert('RGB'))
    return np.ascontiguousarray(rgb[:, :, ::-1])


class LoadRefImageFromFile:
    """datasets/pipelines/loading.py:33-68, same `results` keys in and out (`img_prefix`, `ref_prefix`, `img_info{filename,
    ref_filename, id}` -> `filename`, `img`, `img_shape`, `ori_shape`, `ref_img`, `iid`). The reference decodes the image and its
    reference on every call; in a clip the reference of frame t is the image of frame t-1 (cityscapes_vps.py:137-148), so the last
    decoded image is kept and a matching `ref_filename` costs no second decode."""

    def __init__(self, sample=True, to_float32=False):
        self.to_float32, self.sample = to_float32, sample
        self._last = (None, None)
        self.decodes = 0

    def _read(self, path):
        if self._last[0] == path:
            return self._last[1]
        self.decodes += 1
        return imread(path)

    def __call__(self, results):
        assert results['ref_prefix'] is not None, 'ref_prefix must be specified.'
        filename = osp.join(results['img_prefix'], results['img_info']['filename'])
        if 'ref_filename' not in results['img_info']:
            raise NotImplementedError('We need this implementation.')             # loading.py:55
        ref_filename = osp.join(results['ref_prefix'], results['img_info']['ref_filename'])
        ref_img = self._read(ref_filename)                                         # before `img` replaces the kept image
        img = ref_img.copy() if ref_filename == filename else self._read(filename)    # independent arrays, like the reference
        self._last = (filename, img)
        img.setflags(write=False)          # the kept array is next frame's ref_img: an in-place transform must not corrupt it
        if self.to_float32:
            img, ref_img = img.astype(np.float32), ref_img.astype(np.float32)
        results['filename'] = filename
        results['img'] = img
        results['img_shape'] = img.shape
        results['ori_shape'] = img.shape
        results['ref_img'] = ref_img
        results['iid'] = results['img_info']['id']
        return results

    def __repr__(self):
        return self.__class__.__name__ + '(to_float32={})'.format(self.to_float32)
