"""CPU restatement (NumPy) of the tail of the reference's test pipeline — TEST INFRASTRUCTURE ONLY.

Normalize -> Pad(size_divisor) -> ImageToTensor as configured in /root/reference/configs/cityscapes/fusetrack.py:153-154,
:184-188, implemented by /root/reference/mmdet/datasets/pipelines/transforms.py:258-269 (Pad._pad_img), :310-318
(Normalize.__call__) and formating.py:52-67 (ImageToTensor) on top of mmcv==0.2.14 (requirements.txt:1), which is NOT in
/root/reference: its two functions are restated from the published 0.2.14 sources —
    imnormalize(img, mean, std, to_rgb):  img = img.astype(np.float32); if to_rgb: img = img[..., ::-1]; return (img - mean) / std
    impad_to_multiple(img, divisor, pad_val): zero-initialised (pad_val) array of the rounded-up shape, image in the top-left
PARITY UNPINNED for those two third-party functions (no mmcv in this image to execute); the call sites, argument order and
configuration values are the reference's."""
import numpy as np


def imnormalize(img, mean, std, to_rgb=True):
    img = img.astype(np.float32)
    if to_rgb:
        img = img[..., ::-1]                                  # cv2.cvtColor(img, COLOR_BGR2RGB) on a 3-channel image
    return (img - mean) / std


def impad_to_multiple(img, divisor, pad_val=0):
    h = int(np.ceil(img.shape[0] / divisor)) * divisor
    w = int(np.ceil(img.shape[1] / divisor)) * divisor
    pad = np.empty((h, w) + img.shape[2:], dtype=img.dtype)
    pad[...] = pad_val
    pad[:img.shape[0], :img.shape[1], ...] = img
    return pad


def prepare(img_u8, mean, std, to_rgb=True, size_divisor=32, pad_val=0):
    """uint8 HWC (BGR as decoded) -> float32 CHW, what the model receives for one image (transforms.py:310-318, :258-269,
    formating.py:64-67)."""
    mean = np.array(mean, dtype=np.float32); std = np.array(std, dtype=np.float32)      # transforms.py:301-302
    x = imnormalize(img_u8, mean, std, to_rgb)
    x = impad_to_multiple(x, size_divisor, pad_val)
    return np.ascontiguousarray(x.transpose(2, 0, 1))
