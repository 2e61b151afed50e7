"""CPU restatement (NumPy) of the tail of the reference's test pipeline — TEST INFRASTRUCTURE ONLY.

Normalize -> Pad(size_divisor) -> ImageToTensor as configured in /root/reference/configs/cityscapes/fusetrack.py:153-154,
:184-188, implemented by /root/reference/mmdet/datasets/pipelines/transforms.py:258-269 (Pad._pad_img), :310-318
(Normalize.__call__) and formating.py:52-67 (ImageToTensor) on top of mmcv==0.2.14 (requirements.txt:1), which is NOT in
/root/reference: its two functions are restated from the published 0.2.14 sources —
    imnormalize(img, mean, std, to_rgb):  img = img.astype(np.float32); if to_rgb: img = img[..., ::-1]; return (img - mean) / std
    impad_to_multiple(img, divisor, pad_val): zero-initialised (pad_val) array of the rounded-up shape, image in the top-left
PARITY UNPINNED for those two third-party functions (no mmcv in this image to execute); the call sites, argument order and
configuration values are the reference's.

Resize (transforms.py:107-122 -> mmcv.imrescale -> cv2.resize(img, new_size, interpolation=INTER_LINEAR) on the decoded uint8
image): OpenCV is a third-party dependency absent from /root/reference AND from this image; `cv2_resize_linear_u8` restates
the published algorithm of modules/imgproc/src/resize.cpp for 8-bit images (fixed-point coefficients with
INTER_RESIZE_COEF_BITS = 11, the HResizeLinear / VResizeLinear<uchar, int, short> passes, and the special case that an exact
2x shrink with INTER_LINEAR is computed as INTER_AREA). PARITY UNPINNED (no cv2 to execute)."""
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


# ------------------------------------------------------------------------------------------------ Resize
def rescale_size(h, w, scale):
    """mmcv 0.2.14 imrescale(img, scale=(long, short), return_scale=True): factor and new (w, h)"""
    max_long, max_short = max(scale), min(scale)
    f = min(max_long / max(h, w), max_short / min(h, w))
    return f, (int(w * float(f) + 0.5), int(h * float(f) + 0.5))


def cv_linear_tables(dst, src):
    """resize.cpp (cv::resize, INTER_LINEAR): source index and the two 11-bit fixed-point weights of every destination index"""
    scale = float(src) / float(dst)
    ofs = np.zeros(dst, dtype=np.int64); a0 = np.zeros(dst, dtype=np.int64); a1 = np.zeros(dst, dtype=np.int64)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if s < 0:
            f = np.float32(0); s = 0
        if s >= src - 1:
            f = np.float32(0); s = src - 1
        ofs[d] = s
        a0[d] = int(np.rint((np.float32(1) - f) * np.float32(2048))); a1[d] = int(np.rint(f * np.float32(2048)))
    return ofs, a0, a1


def cv2_resize_linear_u8(img, dsize):
    """cv2.resize(img uint8 [h0,w0,c], (w, h), interpolation=cv2.INTER_LINEAR)"""
    w, h = int(dsize[0]), int(dsize[1])
    img = np.asarray(img, dtype=np.uint8)
    h0, w0 = img.shape[:2]
    src = img.astype(np.int64)
    if (w0, h0) == (w, h):
        return img.copy()
    if w0 == 2 * w and h0 == 2 * h:
        # `interpolation == INTER_LINEAR && is_area_fast && iscale_x == 2 && iscale_y == 2` -> INTER_AREA (ResizeAreaFastVec)
        return ((src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    xo, xa0, xa1 = cv_linear_tables(w, w0)
    yo, yb0, yb1 = cv_linear_tables(h, h0)
    x1 = np.minimum(xo + 1, w0 - 1)
    shp = (1, w) + (1,) * (img.ndim - 2)
    rows = src[:, xo] * xa0.reshape(shp) + src[:, x1] * xa1.reshape(shp)          # HResizeLinear: int, scaled by 2^11
    y1 = np.minimum(yo + 1, h0 - 1)
    s0, s1 = rows[yo], rows[y1]
    shp = (h, 1) + (1,) * (img.ndim - 2)
    out = (((yb0.reshape(shp) * (s0 >> 4)) >> 16) + ((yb1.reshape(shp) * (s1 >> 4)) >> 16) + 2) >> 2      # VResizeLinear, FixedPtCast<.., 22>
    return np.clip(out, 0, 255).astype(np.uint8)
