"""Input preparation on the device against the NumPy restatement of Normalize -> Pad -> ImageToTensor."""
import numpy as np
import pytest
import torch

from oracle import pipeline as opl

NORM = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)     # configs/cityscapes/fusetrack.py:153-154


def test_oracle_normalize_pad_shapes_and_values():
    img = np.arange(5 * 7 * 3, dtype=np.uint8).reshape(5, 7, 3)
    x = opl.prepare(img, **NORM, size_divisor=4)
    assert x.shape == (3, 8, 8) and x.dtype == np.float32
    assert x[0, 0, 0] == np.float32((np.float32(2) - np.float32(123.675)) / np.float32(58.395))    # R = BGR channel 2
    assert float(np.abs(x[:, 5:, :]).max()) == 0.0 and float(np.abs(x[:, :, 7:]).max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('H,W', [(1024, 2048), (37, 53), (64, 96)])
@pytest.mark.parametrize('to_rgb', [True, False])
def test_device_image_prep_is_bit_exact(dev, H, W, to_rgb):
    from vps_amd import pipeline as pl
    rng = np.random.default_rng(H + W)
    img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    cfg = dict(NORM); cfg['to_rgb'] = to_rgb
    prep = pl.DeviceImagePrep(**cfg, size_divisor=32, img_scale=(max(H, W), min(H, W)), device=dev)
    out, img_shape, pad_shape, sf = prep.prep(img)
    ref = opl.prepare(img, **cfg, size_divisor=32)
    assert sf == 1.0 and img_shape == (H, W, 3) and pad_shape == (ref.shape[1], ref.shape[2], 3)
    assert np.array_equal(out.cpu().numpy(), ref)


@pytest.mark.gpu
def test_results_dict_and_pair_feeder(dev):
    from vps_amd import pipeline as pl
    rng = np.random.default_rng(1)
    f0, f1 = (rng.integers(0, 256, size=(64, 128, 3), dtype=np.uint8) for _ in range(2))
    prep = pl.DeviceImagePrep(**NORM, img_scale=(128, 64), device=dev)
    res = prep(dict(img=f1.copy(), ref_img=f0.copy()))
    assert res['pad_shape'] == (64, 128, 3) and res['pad_size_divisor'] == 32 and res['img_norm_cfg']['to_rgb']
    assert np.array_equal(res['ref_img'].cpu().numpy(), opl.prepare(f0, **NORM))
    feed = pl.PairFeeder(prep)
    a, a_ref = feed(f0)
    b, b_ref = feed(f1)
    assert a.data_ptr() == a_ref.data_ptr()                     # the first frame is its own reference
    assert b_ref.data_ptr() == a.data_ptr()                     # frame t's ref_img is frame t-1's tensor, prepared once
    assert np.array_equal(b[0].cpu().numpy(), opl.prepare(f1, **NORM))
    # a non-identity rescale: Resize (cv2 fixed-point bilinear) -> Normalize -> Pad on the device
    out, img_shape, pad_shape, sf = pl.DeviceImagePrep(**NORM, img_scale=(2048, 1024), device=dev).prep(f0)
    f, (nw, nh) = opl.rescale_size(64, 128, (2048, 1024))
    assert sf == f == 16.0 and img_shape == (nh, nw, 3) == (1024, 2048, 3)
    assert np.array_equal(out.cpu().numpy(), opl.prepare(opl.cv2_resize_linear_u8(f0, (nw, nh)), **NORM))


def test_oracle_resize_properties():
    """cv2.resize(INTER_LINEAR, uint8) restatement: identity, constant images, the exact-2x area rule, 11-bit weights"""
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(12, 20, 3), dtype=np.uint8)
    assert np.array_equal(opl.cv2_resize_linear_u8(img, (20, 12)), img)
    assert np.array_equal(opl.cv2_resize_linear_u8(np.full((9, 7, 3), 200, np.uint8), (31, 17)), np.full((17, 31, 3), 200, np.uint8))
    half = opl.cv2_resize_linear_u8(img, (10, 6))
    i64 = img.astype(np.int64)
    assert np.array_equal(half, ((i64[0::2, 0::2] + i64[0::2, 1::2] + i64[1::2, 0::2] + i64[1::2, 1::2] + 2) >> 2).astype(np.uint8))
    o, a0, a1 = opl.cv_linear_tables(31, 7)
    assert (a0 + a1 == 2048).all() and o.min() == 0 and o.max() == 6 and a1[0] == 0 and a1[-1] == 0
    up = opl.cv2_resize_linear_u8(img, (40, 24))            # x2 upscale: every output within the range of its 4 sources
    assert up.shape == (24, 40, 3) and up.min() >= img.min() and up.max() <= img.max()


@pytest.mark.gpu
@pytest.mark.parametrize('H0,W0,H,W', [(540, 960, 1024, 1820), (1080, 1920, 1024, 1820), (2048, 4096, 1024, 2048), (37, 53, 64, 91), (64, 96, 17, 29)])
def test_device_resize_matches_cv2_restatement(dev, H0, W0, H, W):
    from vps_amd import pipeline as pl
    img = np.random.default_rng(H0 + W).integers(0, 256, size=(H0, W0, 3), dtype=np.uint8)
    prep = pl.DeviceImagePrep(**NORM, device=dev)
    out = prep.resize(torch.from_numpy(img).to(dev), W, H)
    assert np.array_equal(out.cpu().numpy(), opl.cv2_resize_linear_u8(img, (W, H)))


def test_imread_and_load_ref_image_decode_every_file_once(tmp_path):
    """host side of SURVEY 8(f) row 1: mmcv.imread semantics (BGR uint8, grey -> 3 channels, alpha dropped) with PIL, and the
    LoadRefImageFromFile mirror (datasets/pipelines/loading.py:33-68: same keys) that does not decode frame t-1 a second time
    as frame t's reference"""
    from PIL import Image
    from vps_amd.pipeline import LoadRefImageFromFile, imread
    rg = np.random.default_rng(0)
    frames = [rg.integers(0, 256, (24, 40, 3), dtype=np.uint8) for _ in range(4)]           # RGB as stored in the file
    for i, f in enumerate(frames):
        Image.fromarray(f).save(str(tmp_path / ('f%d.png' % i)))
    assert np.array_equal(imread(str(tmp_path / 'f0.png')), frames[0][:, :, ::-1])
    Image.fromarray(frames[0][:, :, 0]).save(str(tmp_path / 'grey.png'))
    g = imread(str(tmp_path / 'grey.png'))
    assert g.shape == (24, 40, 3) and all(np.array_equal(g[:, :, c], frames[0][:, :, 0]) for c in range(3))
    rgba = np.concatenate([frames[1], np.full((24, 40, 1), 77, np.uint8)], axis=2)
    Image.fromarray(rgba).save(str(tmp_path / 'rgba.png'))
    assert np.array_equal(imread(str(tmp_path / 'rgba.png')), frames[1][:, :, ::-1])
    load = LoadRefImageFromFile()
    for t in range(4):
        ref = t - 1 if t else 0                                                             # the first frame is its own reference
        res = load(dict(img_prefix=str(tmp_path), ref_prefix=str(tmp_path), img_info=dict(filename='f%d.png' % t, ref_filename='f%d.png' % ref, id=10001 + t)))
        assert np.array_equal(res['img'], frames[t][:, :, ::-1]) and np.array_equal(res['ref_img'], frames[ref][:, :, ::-1])
        assert res['img_shape'] == res['ori_shape'] == (24, 40, 3) and res['iid'] == 10001 + t
        assert res['filename'] == str(tmp_path / ('f%d.png' % t))
    assert load.decodes == 4                                                                # the reference decodes 8 times here
    with pytest.raises(NotImplementedError):
        load(dict(img_prefix='.', ref_prefix='.', img_info=dict(filename='x.png', id=1)))


def test_clip_feeder_decodes_every_file_once_and_keeps_tensor_identity(tmp_path):
    """vps_amd.pipeline.ClipFeeder (the product's `load_frame` for clips on disk) with a host-side stand-in for DeviceImagePrep:
    files decoded once by the thread pool, the SAME tensor object for frame t as `img` and as frame t+1's reference, old frames
    dropped, a shard that starts mid-clip served, decoded pixels == imread"""
    from PIL import Image
    from vps_amd.pipeline import ClipFeeder, imread

    class HostPrep:
        device = torch.device('cpu')
        calls = 0

        def prep(self, img):
            HostPrep.calls += 1
            t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float()
            return t, tuple(img.shape), tuple(img.shape), 1.0

    rng = np.random.RandomState(0)
    files = []
    for i in range(7):
        a = rng.randint(0, 255, (12, 20, 3)).astype(np.uint8)
        fn = str(tmp_path / ('f%02d.png' % i))
        Image.fromarray(a).save(fn)
        files.append(fn)
    fd = ClipFeeder(files, HostPrep(), workers=3)
    assert len(fd) == 7
    prev = None
    for t in range(7):
        img = fd(t)
        assert img.shape == (1, 3, 12, 20)
        assert fd(t) is img                                   # asked twice (as img, then as the next frame's reference): one object
        if prev is not None:
            assert fd(t - 1) is prev                          # frame t-1 is still there while frame t is current
        assert torch.equal(img[0], torch.from_numpy(np.ascontiguousarray(imread(files[t]))).permute(2, 0, 1).float())
        prev = img
        assert set(fd._ready) <= {t - 1, t}
    assert fd.decodes == 7 and HostPrep.calls == 7
    fd.close()
    fd2 = ClipFeeder(files, HostPrep(), workers=2)
    x4 = fd2(4); x3 = fd2(3)                                  # a shard that starts at frame 4 loads its reference frame 3 afterwards
    assert torch.equal(x3[0], torch.from_numpy(np.ascontiguousarray(imread(files[3]))).permute(2, 0, 1).float())
    assert fd2(4) is x4 and fd2(5).shape == (1, 3, 12, 20)
    fd2.close()


def test_clip_feeder_keeps_its_read_ahead_in_the_runner_access_order(tmp_path):
    """ClipShardRunner asks a rank that hands a feature on for its LAST frame first (e-1), then for s, (s-1), s+1 .. e-1.
    The feeder treated the first request as the start of its window and never moved back: every shard frame was decoded on the
    spot. Now a request behind the window rewinds it (decodes in flight beyond it give their slots back), `set_range` keeps the
    read-ahead inside the shard, and every frame of the shard is decoded exactly once, all but the first two from the window."""
    from PIL import Image
    from vps_amd.clip_shard import partition
    from vps_amd.pipeline import ClipFeeder, imread

    class HostPrep:
        device = torch.device('cpu')

        def prep(self, img):
            return torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float(), tuple(img.shape), tuple(img.shape), 1.0

    rng = np.random.RandomState(3)
    files = []
    for i in range(24):
        fn = str(tmp_path / ('g%02d.png' % i))
        Image.fromarray(rng.randint(0, 255, (10, 16, 3)).astype(np.uint8)).save(fn)
        files.append(fn)
    want = [torch.from_numpy(np.ascontiguousarray(imread(f))).permute(2, 0, 1).float() for f in files]

    for rank, world in ((0, 2), (1, 2), (1, 3)):
        s, e = partition(len(files), world)[rank]
        fd = ClipFeeder(files, HostPrep(), workers=2, ahead=3)
        # the order ClipShardRunner.run produces on a rank that sends a hand-off (and, for rank > 0, primes with frame s-1)
        if hasattr(fd, 'set_range'):
            fd.set_range(max(s - 1, 0), e)
        order = ([e - 1] if rank < world - 1 else []) + [s] + ([s - 1] if rank > 0 else [])
        got = {t: fd(t) for t in order}
        for t in range(s, e):
            x = fd(t) if t not in got else got[t]
            assert torch.equal(x[0], want[t]), (rank, t)
            if t + 1 < e and t + 1 not in got:
                got[t + 1] = fd(t + 1)                      # the announced next frame
        nshard = e - max(s - 1, 0)
        assert fd.decodes == nshard, (rank, world, fd.decodes, nshard)          # every frame once, none of another rank's
        assert fd.out_of_window == 0, (rank, world, fd.out_of_window)            # nothing decoded on the spot
        assert all(u < e for u in fd._pending), fd._pending
        fd.close()


@pytest.mark.parametrize('mode,shape', [('RGB', (37, 53)), ('RGBA', (16, 40)), ('L', (21, 19)), ('RGB', (256, 512))])
def test_native_png_decoder_equals_pil(tmp_path, mode, shape):
    """csrc/png_host.cpp (host code of libvpship, no GPU needed): bit-identical to PIL on every filter type the encoder picks
    (smooth and noisy content, all compression levels), RGB / RGBA (alpha dropped) / grey (replicated); flavours outside its
    scope (16-bit, palette, interlaced) are handed back to the general decoder"""
    from PIL import Image
    from vps_amd.pipeline import imread, png_decode
    rng = np.random.RandomState(1)
    H, W = shape
    nch = {'RGB': 3, 'RGBA': 4, 'L': 1}[mode]
    yy, xx = np.mgrid[0:H, 0:W]
    for k, level in enumerate((0, 1, 6, 9)):
        a = (np.stack([(xx * (3 + c) + yy * 2 + 17 * c) % 256 for c in range(nch)], -1) if k % 2 == 0 else rng.randint(0, 256, (H, W, nch))).astype(np.uint8)
        a = a[..., 0] if nch == 1 else a
        fn = str(tmp_path / ('t%d.png' % k))
        Image.fromarray(a, mode).save(fn, compress_level=level)
        got = imread(fn)
        want = np.asarray(Image.open(fn).convert('RGB'))[:, :, ::-1]
        assert got.dtype == np.uint8 and got.shape == (H, W, 3) and np.array_equal(got, want), (mode, level)
    # out of scope -> None -> imread falls back to PIL
    pal = str(tmp_path / 'pal.png')
    Image.fromarray(rng.randint(0, 255, (8, 9)).astype(np.uint8), 'L').convert('P').save(pal)
    assert png_decode(bytearray(open(pal, 'rb').read())) is None
    assert imread(pal).shape == (8, 9, 3)
    assert png_decode(bytearray(b'not a png at all, just thirty-three bytes.')) is None


@pytest.mark.parametrize('C', [1, 3, 4])
def test_native_png_decoder_every_filter_type_on_every_row_position(C):
    """the un-filter loops of csrc/png_host.cpp (round 6: per-channel chains, branch-free Paeth predictor, thread-local scanline buffer)
    on files written HERE with a chosen filter type per scanline (PNG specification 9.2: None, Sub, Up, Average, Paeth) - every type as
    the FIRST row (no row above) and behind every other type, on noise (every predictor branch taken). Decoding twice through the same
    thread re-uses its buffer; a second, larger image grows it."""
    import struct
    import zlib
    from vps_amd.pipeline import png_decode

    def paeth(a, b, c):
        p = a + b - c
        pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
        return a if pa <= pb and pa <= pc else (b if pb <= pc else c)

    def encode(img, types):
        H, W = img.shape[:2]
        rows = img.reshape(H, W * C).astype(np.int64)
        raw = bytearray()
        for y in range(H):
            ft = types[y % len(types)]
            cur, up = rows[y], rows[y - 1] if y else np.zeros(W * C, np.int64)
            left = np.concatenate([np.zeros(C, np.int64), cur[:-C]])
            ul = np.concatenate([np.zeros(C, np.int64), up[:-C]])
            pred = {0: 0 * cur, 1: left, 2: up, 3: (left + up) >> 1, 4: np.array([paeth(a, b, c) for a, b, c in zip(left, up, ul)])}[ft]
            raw.append(ft)
            raw += ((cur - pred) & 255).astype(np.uint8).tobytes()
        def chunk(t, d):
            return struct.pack('>I', len(d)) + t + d + struct.pack('>I', zlib.crc32(t + d) & 0xFFFFFFFF)
        ctype = {1: 0, 3: 2, 4: 6}[C]
        return (b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', W, H, 8, ctype, 0, 0, 0)) +
                chunk(b'IDAT', zlib.compress(bytes(raw), 6)) + chunk(b'IEND', b''))

    rg = np.random.default_rng(C)
    for (H, W), order in (((11, 19), [0, 1, 2, 3, 4]), ((11, 19), [4, 3, 2, 1, 0]), ((11, 19), [3, 4, 4, 1, 2, 0]), ((40, 70), [2, 4, 1, 3, 0, 4])):
        img = rg.integers(0, 256, (H, W, C) if C > 1 else (H, W), dtype=np.uint8)
        want = (np.repeat(img[:, :, None], 3, 2) if C == 1 else img[:, :, 2::-1][:, :, :3] if C == 3 else img[:, :, [2, 1, 0]])
        for _ in range(2):
            got = png_decode(encode(img, order))
            assert got is not None and np.array_equal(got, want), (C, H, W, order)
