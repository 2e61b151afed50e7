"""The caller side of `tools/test_vpq.py` for the import switch of INTEGRATION.md Level 1 (round 6): `build_dataloader` and
`MMDataParallel` stand-ins that give the UNMODIFIED `single_gpu_test` loop (tools/test_vpq.py:28-69: `for i, data in
enumerate(data_loader): result = model(return_loss=False, rescale=True, **data)`) the cross-frame pipeline the clip runner has.

The loop hands the detector one frame per call and nothing else, but the dataset knows frame t+1 (datasets/cityscapes_vps.py:137-148:
test mode walks the frames of a video in order, `ref_img` = the previous frame). `LookaheadLoader` wraps any iterable of test-mode
batches (`dict(img=[T], img_meta=[DC], ref_img=[T])`, datasets/pipelines/formating.py Collect + test_aug), keeps `depth` batches ahead,
moves their images to the device (so the tensors it announces are the tensors the model is later called with) and posts the pairs
of the NEXT frames on the detector's announcement board (`vps_amd.detector.announce`) right before it yields frame t. The detector
reads the board when its caller passed no `prefetch=` and enqueues those frames' image-only stages (FlowNet2, ResNet + FPN) beside
the current frame - exactly what `ClipShardRunner` does with its `prefetch=` argument.

No mmcv / mmdet in this image: `MMDataParallel` / `DataContainer` restate the behaviour the call relies on (mmcv 0.2.x
parallel/data_container.py, scatter_gather.py: a cpu_only container is unwrapped to its per-GPU list, tensors go to the device - a
tensor that already lives there is passed through AS IS -, lists / tuples / dicts are mapped element-wise)."""
import collections

import torch

from . import detector as _det


class DataContainer:
    """mmcv.parallel.DataContainer as far as the test pipeline uses it"""

    def __init__(self, data, stack=False, padding_value=0, cpu_only=False):
        self._data, self.stack, self.padding_value, self.cpu_only = data, stack, padding_value, cpu_only

    data = property(lambda s: s._data)


def _is_container(o):
    return hasattr(o, 'data') and hasattr(o, 'cpu_only') and not torch.is_tensor(o)


def scatter(obj, device):
    """mmcv scatter for ONE device: containers unwrapped, tensors moved (non-blocking; passed through when already there)"""
    if _is_container(obj):
        return obj.data[0] if obj.cpu_only else scatter(obj.data[0], device)
    if torch.is_tensor(obj):
        return obj if obj.device == device else obj.to(device, non_blocking=True)
    if isinstance(obj, (list, tuple)):
        return type(obj)(scatter(o, device) for o in obj)
    if isinstance(obj, dict):
        return {k: scatter(v, device) for k, v in obj.items()}
    return obj


class MMDataParallel(torch.nn.Module):
    """`MMDataParallel(model, device_ids=[gpu])` of tools/test_vpq.py:149 for one device: scatters the call's arguments and forwards"""

    def __init__(self, module, device_ids=None, dim=0):
        super().__init__()
        self.module = module
        self.device_ids = list(device_ids) if device_ids is not None else [torch.cuda.current_device() if torch.cuda.is_available() else 0]
        self.device = torch.device('cuda', self.device_ids[0]) if torch.cuda.is_available() else torch.device('cpu')

    def forward(self, *inputs, **kwargs):
        return self.module(*scatter(inputs, self.device), **scatter(kwargs, self.device))


class LookaheadLoader:
    """Iterates `loader` (test-mode batches), `depth` batches ahead; see the module docstring. len(), .dataset and every other attribute
    are the wrapped loader's. `device=None`: the images stay where the loader put them (they must then be the objects the model gets)."""

    def __init__(self, loader, depth=2, device=None):
        self.loader, self.depth = loader, max(int(depth), 0)
        self.device = torch.device(device) if device is not None else (torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None)

    def __len__(self):
        return len(self.loader)

    def __getattr__(self, name):
        return getattr(self.loader, name)

    def _stage(self, data):
        """the batch with its images on the device: NEW lists around the same containers, the loader's batch is not modified"""
        if self.device is None or not isinstance(data, dict):
            return data
        out = dict(data)
        for k in ('img', 'ref_img'):
            if k in out:
                out[k] = scatter(out[k], self.device)
        return out

    @staticmethod
    def _pair(data):
        """(img, ref_img) tensors of a single-scale test batch, or None (no reference image / multi-scale: nothing to announce)"""
        try:
            img, ref = data['img'], data['ref_img']
            img = img[0] if isinstance(img, (list, tuple)) else img
            ref = ref[0] if isinstance(ref, (list, tuple)) else ref
            if torch.is_tensor(img) and torch.is_tensor(ref) and len(data['img']) == 1:
                return img, ref
        except (KeyError, TypeError, IndexError):
            pass
        return None

    def __iter__(self):
        it = iter(self.loader)
        queue = collections.deque()
        done = False
        try:
            while True:
                while not done and len(queue) < self.depth + 1:
                    try:
                        queue.append(self._stage(next(it)))
                    except StopIteration:
                        done = True
                if not queue:
                    break
                cur = queue.popleft()
                ahead = [p for p in (self._pair(d) for d in queue) if p is not None]
                _det.announce(ahead)                    # the frames BEHIND the one the caller gets now
                yield cur
        finally:
            _det.announce([])


def build_dataloader(dataset, imgs_per_gpu=1, workers_per_gpu=0, num_gpus=1, dist=False, shuffle=False, lookahead=2, **kwargs):
    """`build_dataloader` of tools/test_vpq.py:125-128 (mmdet/datasets/loader/build_loader.py) for test mode: one sample per batch, the
    dataset's order, wrapped in a `LookaheadLoader`. With mmdet importable its own builder makes the inner loader (same collate)."""
    assert imgs_per_gpu == 1 and not shuffle, 'test mode: one image per batch, the dataset order (datasets/cityscapes_vps.py walks a video frame by frame)'
    inner = None
    try:                                                 # pragma: no cover - no mmdet in this image
        from mmdet.datasets import build_dataloader as _mm
        inner = _mm(dataset, imgs_per_gpu=imgs_per_gpu, workers_per_gpu=workers_per_gpu, num_gpus=num_gpus, dist=dist, shuffle=False, **kwargs)
    except ImportError:
        def collate(batch):                              # mmcv.parallel.collate for ONE sample: every leaf gains the batch dimension
            b = batch[0]

            def lift(v):
                if _is_container(v):
                    return DataContainer([[v.data]] if v.cpu_only else [lift(v.data)], v.stack, v.padding_value, v.cpu_only)
                if torch.is_tensor(v):
                    return v.unsqueeze(0)
                if isinstance(v, (list, tuple)):
                    return type(v)(lift(o) for o in v)
                return v
            return {k: lift(v) for k, v in b.items()} if isinstance(b, dict) else b
        inner = torch.utils.data.DataLoader(dataset, batch_size=1, shuffle=False, num_workers=workers_per_gpu, collate_fn=collate,
                                            pin_memory=torch.cuda.is_available())
    return LookaheadLoader(inner, depth=lookahead)
