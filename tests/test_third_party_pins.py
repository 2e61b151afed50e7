"""Pins of the oracle's restatements of THIRD-PARTY code the reference calls on the path:

    cv2.resize INTER_LINEAR float32    mask_removal.py:66-70        -> oracle/ops.py:cv2_resize_linear
    cv2.resize INTER_LINEAR uint8      mmcv imresize (Resize)       -> oracle/pipeline.py:cv2_resize_linear_u8
    mmcv.imnormalize / impad           transforms.py:176-191,:261   -> oracle/pipeline.py:imnormalize / impad_to_multiple

Neither cv2 nor mmcv is in this image (nor on the GPU box, which runs the same image): every test here SKIPS there and the
restatements stay "unpinned by the real library" as DESIGN §4 says. They run wherever the libraries exist (a maintainer's box),
which is the cheapest way to close the pin without vendoring anything.
"""
import numpy as np
import pytest

from oracle import ops as O
from oracle import pipeline as OP


@pytest.mark.parametrize('src,dst', [((14, 14), (37, 61)), ((28, 28), (28, 28)), ((28, 28), (9, 13)), ((28, 28), (301, 177)), ((7, 5), (1, 1))])
def test_cv2_resize_linear_float32(src, dst):
    cv2 = pytest.importorskip('cv2')
    rng = np.random.default_rng(src[0] * 131 + dst[1])
    a = rng.random(src, dtype=np.float32)
    ref = cv2.resize(a, (dst[1], dst[0]), interpolation=cv2.INTER_LINEAR)
    got = O.cv2_resize_linear(a, (dst[1], dst[0]))
    assert got.dtype == np.float32 and got.shape == ref.shape
    # cv2's float path is a separable fp32 lerp: the restatement follows the same order of operations; 1 ulp of slack for FMA contraction
    np.testing.assert_allclose(got, ref, rtol=0, atol=1.2e-7)


@pytest.mark.parametrize('src,dst', [((64, 96), (128, 192)), ((100, 60), (67, 41)), ((33, 47), (33, 47)), ((1024, 2048), (800, 1600))])
def test_cv2_resize_linear_uint8(src, dst):
    cv2 = pytest.importorskip('cv2')
    rng = np.random.default_rng(src[1])
    a = rng.integers(0, 256, size=src + (3,), dtype=np.uint8)
    ref = cv2.resize(a, (dst[1], dst[0]), interpolation=cv2.INTER_LINEAR)
    got = OP.cv2_resize_linear_u8(a, (dst[1], dst[0]))
    assert np.array_equal(got, ref), 'differs at %d of %d bytes' % (int((got != ref).sum()), ref.size)


def test_mmcv_normalize_and_pad():
    mmcv = pytest.importorskip('mmcv')
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    mean = np.array([123.675, 116.28, 103.53], np.float32); std = np.array([58.395, 57.12, 57.375], np.float32)
    ref = mmcv.impad_to_multiple(mmcv.imnormalize(img, mean, std, True), 32, pad_val=0)
    got = OP.impad_to_multiple(OP.imnormalize(img, mean, std, True), 32, 0)
    assert got.shape == ref.shape and np.array_equal(got, ref)
