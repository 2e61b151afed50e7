"""vps_amd — MI355X-native FuseTrack (VPSNet) inference path.

Python host code on PyTorch-ROCm (device memory, streams, torch.distributed only) over libvpship.so, a C-ABI library of
hand-written HIP kernels for gfx950 (include/vps_hip.h). Importing the package registers the reference's component
names (PanopticFuseTrack, ResNet, FPN, BFPTcea, UPSNetFPN, RPNHead, SingleRoIExtractor, SharedFCBBoxHead, TrackHead,
FCNMaskHead) so configs/cityscapes/fusetrack.py-style configs build unchanged.
"""
from . import hip  # noqa: F401
from .registry import (BACKBONES, DETECTORS, EXTRA_NECKS, HEADS, NECKS, PANOPTIC, ROI_EXTRACTORS, Config, ConfigDict,  # noqa: F401
                       Registry, build_backbone, build_detector, build_extra_neck, build_from_cfg, build_head, build_neck,
                       build_panoptic, build_roi_extractor)
from . import backbones, necks, heads, flownet2, panoptic_ops, detector  # noqa: F401,E402  (registration side effects)
from .detector import PanopticFuse, PanopticFuseTrack, PanopticTrack  # noqa: F401,E402
from .checkpoint import load_checkpoint  # noqa: F401,E402
from .dataloader import DataContainer, LookaheadLoader, MMDataParallel, build_dataloader  # noqa: F401,E402

__version__ = '0.1.0'
