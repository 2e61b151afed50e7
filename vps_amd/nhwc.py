"""NHWC activation maps, persistent device workspace and packed convolution weights for libvpship.

Layout: every activation is a device tensor [N, H, W, ld] fp32; a feature map (`FMap`) is a channel window
[coff, coff+C) of it. Producers write directly into windows of concat buffers, so the reference's torch.cat calls
(FlowNet2 decoders, LiteFlowNet input, TCEA stack, UPSNet head) never materialise. Pad channels of a buffer are
zero (buffers come from `Workspace.get(..., zero=True)` once and pads are never written).
"""
import ctypes
import math
import os
import weakref
from ctypes import c_float, c_int, c_void_p

import torch

from . import hip


def _ceil(a, b):
    return (a + b - 1) // b * b


# arithmetic of every PackedConv created without an explicit `prec` (hip.PREC_F32 exact / PREC_BF16X3 / PREC_BF16X6 / PREC_F16X3).
# The library default is the exact fp32 mode; VPS_PREC=bf16x6 / f16x3 (fp32-grade split modes) / bf16x3 / bf16 / f32 selects it
# for an unmodified caller such as tools/test_vpq.py.
_PREC_NAMES = {'f32': hip.PREC_F32, 'bf16': hip.PREC_BF16, 'bf16x3': hip.PREC_BF16X3, 'bf16x6': hip.PREC_BF16X6, 'f16x3': hip.PREC_F16X3}
PREC_NAMES = _PREC_NAMES
# number of 16-bit weight planes / MFMA products per fp32 product of each split mode
_PLANES = {hip.PREC_BF16: 1, hip.PREC_BF16X3: 2, hip.PREC_BF16X6: 3, hip.PREC_F16X3: 3}
MFMA_PRODUCTS = {hip.PREC_F32: 1, hip.PREC_BF16: 1, hip.PREC_BF16X3: 3, hip.PREC_BF16X6: 6, hip.PREC_F16X3: 3}
# f16x3 range report: one device word PER LAYER (vps_conv_desc.status points at the layer's slot); a conv launch ORs bit 0 into it
# when it staged an activation beyond the fp16 range. Slot 0 is shared by anonymous layers (per-frame GEMMs built from device
# matrices), every persistent PackedConv owns one of the others and can fall back to bf16x6 on its own (`f16_fallback`).
F16_SLOTS = 4096
_F16_STATUS = {}
_F16_LAYERS = {}          # slot -> weakref of the PackedConv that owns it
_F16_NEXT = [1]
F16_FALLBACKS = [0]       # layers switched to bf16x6 so far (reported by bench.py)


def f16_status(device):
    """the range-report words of the f16x3 mode on `device` (int32 tensor [F16_SLOTS]); PanopticFuseTrack reads their maximum with
    its end-of-frame read"""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else (torch.cuda.current_device() if device.type == 'cuda' else 0))
    t = _F16_STATUS.get(key)
    if t is None:
        t = torch.zeros(F16_SLOTS, dtype=torch.int32, device=device)
        _F16_STATUS[key] = t
    return t


def f16_fallback(device):
    """f16x3: the layers that reported an activation beyond the fp16 range switch to bf16x6 (three bf16 planes, the fp32 exponent
    range: no range restriction, same fp32-grade error) FOR GOOD, the report is cleared. -> number of layers switched. An
    anonymous layer (slot 0) cannot switch: raises. The caller re-runs whatever those launches produced."""
    st = f16_status(device)
    flagged = torch.nonzero(st).flatten().cpu().tolist()
    st.zero_()
    n = 0
    if F16_CORR_SLOT in flagged:
        # a correlation operand left the fp16 range: the exact kernels from now on
        flagged.remove(F16_CORR_SLOT)
        CORR_F16[0] = False
        n += 1
    for slot in flagged:
        ref = _F16_LAYERS.get(slot)
        pc = ref() if ref is not None else None
        if pc is None:
            raise hip.VpsHipError('f16x3: an activation exceeded the fp16 range (|x| > 65504) in a per-frame GEMM that has no bf16x6 '
                                  'fallback. Run this checkpoint with VPS_PREC=bf16x6 (or f32).')
        n += pc.use_fallback(device)
    F16_FALLBACKS[0] += n
    return n


if os.environ.get('VPS_PREC', 'f32') not in _PREC_NAMES:
    raise ValueError('VPS_PREC must be one of %s' % sorted(_PREC_NAMES))
DEFAULT_PREC = _PREC_NAMES[os.environ.get('VPS_PREC', 'f32')]

# bench.py sets this to a list to time every vps_conv2d launch with HIP events on the launch stream:
# entries (algorithmic_flops, start_event, end_event, shape tag, algorithmic_bytes). None = no instrumentation (the default).
CONV_TRACE = None
# split-K partial sums are added up by a separate reduce launch. VPS_SPLITK_LAST_BLOCK=1 lets the block that finishes a tile's last
# split do it (vps_conv_desc.tile_counter): measured 31.2 instead of 42.2 frames/s (profiles/r03_bench_splitk_last_block_ab.json) -
# the device-scope release / acquire around the ticket (buffer_wbl2 / buffer_inv: the partials of a tile come from blocks on
# different XCDs, each with its own L2) writes back and invalidates a whole L2 per block. Kept as an option, OFF.
# blocks a split launch aims at. 512 (two per CU) until round 4, when a frame had at most two kernels in flight; with the image-only
# stages on three streams of their own the other streams fill the CUs a low-resolution layer leaves idle, and fewer, longer splits
# win (less partial-sum traffic, smaller reduce launches): 1024 / 512 / 256 / 128 -> 49.9 / 51.1 / 51.5 / 51.8 frames/s, conv time of
# the one-stream instrumented frame 20.5 / 20.0 / 19.9 / 21.0 ms (gpurun_out c5, round 5) -> 256
SPLITK_TARGET_BLOCKS = int(os.environ.get('VPS_SPLITK_TARGET', '256'))
SPLITK_LAST_BLOCK = os.environ.get('VPS_SPLITK_LAST_BLOCK', '0') == '1'
# 1: the 3x3 narrow-output layers with >= 64 input channels on large maps run on the MFMA tile kernel (a second packed copy). It won
# against the first vector kernel (predict_flow2 0.133 -> 0.080 ms); the round-3 vector kernel does 0.059 ms in exact fp32: default off.
SMALL_ON_MFMA = os.environ.get('VPS_SMALL_MFMA', '0') != '0'
THIN_KERNEL = os.environ.get('VPS_THIN', '1') != '0'      # thin-input layers (cin_pad <= 12 -> 64) get the packing of csrc/conv_thin.hip too
GN_REP = 32   # copies of the GroupNorm sums a conv epilogue spreads its atomics over (vps_conv_desc.gn_rep)


def check_f16_range(device):
    """f16x3 mode: raise if a convolution since the last check staged an activation beyond the fp16 range (|x| > 65504): its result
    is not fp32-grade (fp16 overflow). For callers of single layers; the detector falls back per layer instead (`f16_fallback`)."""
    device = torch.device(device)
    for key, t in _F16_STATUS.items():
        if key[0] == device.type and (device.index is None or key[1] == device.index):
            if int(t.amax().item()) != 0:
                t.zero_()
                raise hip.VpsHipError('f16x3: an activation exceeded the fp16 range (|x| > 65504) in a convolution; the fp16 split is '
                                      'not fp32-grade there. Run it with prec=bf16x6 (or f32).')


class FMap:
    __slots__ = ('t', 'C', 'coff')

    def __init__(self, t, C=None, coff=0):
        assert t.dim() == 4 and t.dtype == torch.float32 and t.is_contiguous()
        self.t = t
        self.C = t.shape[3] if C is None else C
        self.coff = coff
        assert self.coff + self.C <= t.shape[3]

    N = property(lambda s: s.t.shape[0])
    H = property(lambda s: s.t.shape[1])
    W = property(lambda s: s.t.shape[2])
    ld = property(lambda s: s.t.shape[3])
    npix = property(lambda s: s.t.shape[0] * s.t.shape[1] * s.t.shape[2])

    def window(self, coff, C):
        return FMap(self.t, C, self.coff + coff)

    def ptr(self):
        return hip.ptr(self.t)

    def to_nchw(self):
        """[N,C,H,W] contiguous copy through the HIP transpose kernel (API boundary / tests)."""
        out = torch.empty(self.N, self.C, self.H, self.W, dtype=torch.float32, device=self.t.device)
        hip.check(hip.load().vps_nhwc_to_nchw(self.ptr(), self.ld, self.coff, hip.ptr(out), self.N, self.C, self.H, self.W,
                                              hip.stream_ptr()), 'vps_nhwc_to_nchw')
        return out


POOLING = os.environ.get('VPS_WS_POOL', '1') != '0'     # 0: every activation keeps a buffer of its own (the pre-round-5 workspace; A/B and debugging)


class Pool:
    """Free lists of fp32 blocks, one list PER STREAM: a block given back after its last consumer was enqueued on stream s may be handed
    to the next producer enqueued on s (stream order makes the reuse safe without any event); it is never handed to work on another
    stream. Blocks are never returned to the allocator, and a frame repeats the same take / give sequence, so every activation
    sees the same address frame after frame (the cached conv descriptors stay valid)."""

    SLACK = 1.5        # a free block serves requests down to 1/SLACK of its size (best fit); smaller requests get a block of their own

    def __init__(self, device):
        self.device = torch.device(device)
        self.free = {}                # stream key -> list of flat fp32 tensors
        self.total = 0                # bytes ever allocated
        self.blocks = 0

    def take(self, key, numel):
        lst = self.free.setdefault(key, [])
        best = -1
        for i, blk in enumerate(lst):
            n = blk.numel()
            if n >= numel and n <= self.SLACK * numel and (best < 0 or n < lst[best].numel()):
                best = i
        if best >= 0:
            return lst.pop(best)
        self.total += 4 * numel
        self.blocks += 1
        return torch.empty(numel, dtype=torch.float32, device=self.device)

    def give(self, key, blk):
        self.free.setdefault(key, []).append(blk)


def _stream_key():
    return getattr(hip.stream_ptr(), 'value', None) or 0


class Workspace:
    """Device buffers of the frame graph, by name. Two kinds:

    * PERSISTENT (`get`, `fmap`): one buffer per name, reused every frame - what crosses a stream or a frame boundary (the outputs of
      the image-only stages, the neck outputs, fcn_score, the cached reference feature), every buffer with zero pad channels
      (`ld > C`: pads are zeroed once and never written) and all small index / scratch buffers;
    * TEMPORARY (`fmap(..., temp=True)`, `release`, `scope`): pad-free fp32 activations that are produced and consumed on ONE stream.
      They come from the `Pool` of the stream that is current when they are taken and go back to it when they are released - from
      then on the block may serve the next producer ON THAT STREAM (liveness-based reuse: round 5; 35.7 -> ~9 GB at 1024x2048).

    INVARIANTS of every caller (vps_amd/detector.py and the modules): a temporary buffer is written and read only by work enqueued on
    the stream it was taken on, between its `fmap` and its `release`; whatever another stream or the next frame reads is persistent;
    branches that run concurrently use disjoint persistent names, a stream is joined (wait_stream / wait_event) before another reads
    what it wrote and before the next frame rewrites it. `out`: a second workspace that receives the buffers a module marks
    `out=True` - the ring slot of a prefetched frame (its flow / FPN levels / gathered feature live until the frame after next) while
    the internals of the image-only stages are shared by all slots (the prefetch streams run them one frame after the other).
    tests: the pooled schedule is bitwise the unpooled one (tests/test_fusetrack_gpu.py, tests/test_fullsize_gpu.py)."""

    def __init__(self, device, pool=None, out=None):
        self.device = torch.device(device)
        self.bufs = {}
        self.pool = pool if pool is not None else Pool(device)
        self.out = out
        self.pooling = POOLING
        self._live = {}              # id(view tensor) -> (block, stream key, name)
        self._scopes = []

    def with_out(self, out):
        """a view of this workspace whose `out=True` buffers go to `out` (persistent names, temporaries and pool are shared)"""
        v = Workspace.__new__(Workspace)
        v.__dict__ = dict(self.__dict__)
        v.out = out
        return v

    def get(self, name, shape, dtype=torch.float32, zero=True):
        key = name
        t = self.bufs.get(key)
        shape = tuple(int(s) for s in shape)
        if t is None or tuple(t.shape) != shape or t.dtype != dtype:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            self.bufs[key] = t
        return t

    def fmap(self, name, N, H, W, C, ld=None, temp=False, out=False):
        if out and self.out is not None:
            return self.out.fmap(name, N, H, W, C, ld)
        ld = _ceil(C, 4) if ld is None else ld
        if temp and self.pooling and ld == C:
            numel = int(N) * int(H) * int(W) * int(ld)
            key = _stream_key()
            blk = self.pool.take(key, numel)
            t = blk[:numel].view(int(N), int(H), int(W), int(ld))
            self._live[id(t)] = (blk, key, name, t)
            if self._scopes:
                self._scopes[-1].append(id(t))
            return FMap(t, C, 0)
        return FMap(self.get(name, (N, H, W, ld)), C, 0)

    def release(self, *maps):
        """the last consumer of these temporary maps has been enqueued (on the stream they were taken on): their blocks may be reused.
        Persistent maps (and None) are ignored, a map is released once."""
        for m in maps:
            if m is None:
                continue
            ent = self._live.pop(id(m.t if isinstance(m, FMap) else m), None)
            if ent is not None:
                self.pool.give(ent[1], ent[0])

    def scope(self):
        """with ws.scope(): ...  - every temporary map taken inside and not released by then goes back at the end of the block"""
        return _Scope(self)

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self.bufs.values())


class _Scope:
    def __init__(self, ws):
        self.ws = ws

    def __enter__(self):
        self.ids = []
        self.ws._scopes.append(self.ids)
        return self.ws

    def __exit__(self, *exc):
        self.ws._scopes.pop()
        for i in self.ids:
            ent = self.ws._live.pop(i, None)
            if ent is not None:
                self.ws.pool.give(ent[1], ent[0])
        return False


def from_nchw(x, ws=None, name=None, Cpad=None):
    """NCHW device tensor -> FMap (HIP transpose, pads zeroed)."""
    N, C, H, W = x.shape
    ld = _ceil(C, 4) if Cpad is None else Cpad
    x = x.contiguous()
    t = ws.get(name, (N, H, W, ld)) if ws is not None else torch.zeros(N, H, W, ld, dtype=torch.float32, device=x.device)
    hip.check(hip.load().vps_nchw_to_nhwc(hip.ptr(x), hip.ptr(t), ld, 0, N, C, H, W, ld, hip.stream_ptr()), 'vps_nchw_to_nhwc')
    return FMap(t, C, 0)


# ------------------------------------------------------------------------------------------------------------
# packed convolution
# ------------------------------------------------------------------------------------------------------------
def pack_thin(g, KS, C4):
    """g: fp16 [2 planes][64][kpad], k = (j*KS + i)*C4 + c (tap-major) -> the packing of csrc/conv_thin.hip (vps_conv_desc.w_thin):
    [KS kernel rows][plane][ceil(KS*C4/16)][cout/32][lane = 32*(k/8 % 2) + cout % 32][8], k = i*C4 + c within the row, zero-padded"""
    RL = KS * C4
    nk = (RL + 15) // 16
    w = torch.zeros(2, 64, KS, nk * 16, dtype=g.dtype, device=g.device)
    w[..., :RL] = g[:, :, :KS * RL].reshape(2, 64, KS, RL)
    # [pl][nb][32][KS][nk][kh][8] -> [KS][pl][nk][nb][kh][32][8]
    return w.view(2, 2, 32, KS, nk, 2, 8).permute(3, 0, 4, 1, 5, 2, 6).contiguous()


# 256-column tile of the deformable layers (VPS_DCN256=0 keeps the 128-column tile); only where it still gives >= this many blocks
DCN256 = [os.environ.get('VPS_DCN256', '1') != '0']
DCN256_MIN_TILES = int(os.environ.get('VPS_DCN256_MIN_TILES', '256'))


def _tile_n(cout):
    return 32 if cout <= 32 else (64 if cout <= 64 else 128)


class PackedConv:
    """One conv / transposed conv / linear / deformable conv of the path, packed for vps_conv2d.

    weight: Conv2d [O,I,kh,kw] (transposed=False) or ConvTranspose2d [I,O,kh,kw] (transposed=True, stride 2).
    bn: optional dict(weight,bias,running_mean,running_var,eps) folded into the epilogue (eval BatchNorm).
    """

    def __init__(self, weight, bias=None, bn=None, stride=1, padding=0, act=hip.ACT_NONE, slope=0.1,
                 transposed=False, deform=False, device='cuda', prec=None, _twin=False):
        w = weight.detach().float().cpu()
        self.prec = DEFAULT_PREC if prec is None else prec
        nout = w.shape[1] if transposed else w.shape[0]
        # narrow outputs (predict_flow, the 2-channel flow up-convolutions, 3-channel heads) run on the exact-fp32 vector kernel in
        # every mode - except where the matrix cores win although 30 of the 32 output columns of their tile are padding: 3x3 layers
        # with >= 64 input channels on >= 100 000 pixels (measured per layer, profiles/r03_conv_table_small_on_mfma.txt: predict_flow2
        # 194->2 @256x512 0.133 -> 0.080 ms per call; the low-resolution ones are faster on the vector kernel). Such a layer is
        # packed both ways and `__call__` picks by the size of the map.
        nin = w.shape[0] if transposed else w.shape[1]
        self.small = nout <= 4 and not deform and not _twin
        self._mfma_twin = None
        if self.small and SMALL_ON_MFMA and not transposed and w.shape[2] == 3 and nin >= 64 and (DEFAULT_PREC if prec is None else prec) != hip.PREC_F32:
            self._mfma_twin = PackedConv(weight, bias, bn, stride, padding, act, slope, transposed, deform, device, prec, _twin=True)
        if self.small:
            self.prec = hip.PREC_F32
        self.stride = stride
        self.act, self.slope = act, float(slope)
        self.deform = deform
        self.transposed = transposed
        if not transposed:
            O, I, kh, kw = w.shape
            self.nclass, self.os = 1, 1
            self.KH, self.KW = kh, kw
            self.pad_y = (padding, padding)
            self.pad_x = (padding, padding)
            cin_pad = _ceil(I, 4)
            self.korder = 1 if (cin_pad >= 64 or cin_pad % 32 == 0) and kh * kw > 1 else 0
            if deform and self.korder == 0 and self.prec != hip.PREC_F32:
                # a deformable layer with few input channels (tap-major k order; none on the path): the split-operand kernels take
                # deformable layers in the chunk-major order only - such a layer runs on the exact-fp32 kernel (vps_conv2d, VPS_EARG 17)
                self.prec = hip.PREC_F32
            blocks = [self._pack_taps(w.permute(0, 2, 3, 1).reshape(O, kh * kw, I), cin_pad)]
        else:
            I, O, kh, kw = w.shape
            assert stride == 2 and kh == kw and kh % 2 == 0, 'only stride-2 even-kernel transposed convs are on the path'
            self.nclass, self.os = 4, 2
            self.stride = 1
            kc = kh // 2
            self.KH = self.KW = kc
            cin_pad = _ceil(I, 4)
            pads, taps = [], []
            for par in range(2):
                r = (par + padding) & 1
                d = (par + padding - r) // 2
                pads.append((kc - 1) - d)
                taps.append([r + 2 * ((kc - 1) - u) for u in range(kc)])  # kernel index used by tap u
            self.pad_y = tuple(pads)
            self.pad_x = tuple(pads)
            self.korder = 1 if (cin_pad >= 64 or cin_pad % 32 == 0) and kc * kc > 1 else 0
            blocks = []
            for py in range(2):
                for px in range(2):
                    wp = torch.zeros(O, kc * kc, I)
                    for uy in range(kc):
                        for ux in range(kc):
                            wp[:, uy * kc + ux, :] = w[:, :, taps[py][uy], taps[px][ux]].t()
                    blocks.append(self._pack_taps(wp, cin_pad))
        self.cin, self.cin_pad, self.cout = I, cin_pad, O
        self.tile_n = _tile_n(O)
        self.cout_pad = _ceil(O, self.tile_n)
        K = blocks[0].shape[1]
        self.kpad = _ceil(K, 32)
        packed = torch.zeros(self.nclass, self.cout_pad, self.kpad)
        for c, b in enumerate(blocks):
            packed[c, :O, :K] = b
        wscale = self._set_weights(packed, device)
        self._fb = None
        self.f16_slot = 0
        if self.prec == hip.PREC_F16X3:
            # what `use_fallback` needs to re-pack this layer for bf16x6: the fp32 weights in packed layout (host) — the epilogue
            # scale without the f16x3 pre-scaling is restored below
            self.f16_slot = _F16_NEXT[0] if _F16_NEXT[0] < F16_SLOTS - 1 else 0        # (the last slot is the correlations')
            if self.f16_slot:
                _F16_NEXT[0] += 1
                _F16_LAYERS[self.f16_slot] = weakref.ref(self)
                self._fb = dict(packed=packed)
        # epilogue: y = acc*scale + shift
        scale = torch.ones(O)
        shift = torch.zeros(O) if bias is None else bias.detach().float().cpu().clone()
        if bn is not None:
            g = bn['weight'].detach().float().cpu(); b = bn['bias'].detach().float().cpu()
            m = bn['running_mean'].detach().float().cpu(); v = bn['running_var'].detach().float().cpu()
            s = g / torch.sqrt(v + bn.get('eps', 1e-5))
            shift = (shift - m) * s + b
            scale = s
            self.has_scale = True
        else:
            self.has_scale = False
        if self._fb is not None:
            self._fb.update(scale=scale.clone() if self.has_scale else None, has_scale=self.has_scale)
        if wscale is not None:
            # f16x3: the weights were pre-scaled per output channel by a power of two; undone exactly here
            scale = scale * wscale[:O].cpu()
            self.has_scale = True
        self.scale = scale.to(device) if self.has_scale else None
        self.shift = shift.to(device) if (bias is not None or bn is not None) else None

    def use_fallback(self, device):
        """f16x3 -> bf16x6 for this layer, for good (an activation beyond the fp16 range was staged here). -> 1 if switched"""
        if self.prec != hip.PREC_F16X3 or self._fb is None:
            return 0
        fb, self._fb = self._fb, None
        self.prec = hip.PREC_BF16X6
        self._set_weights(fb['packed'], device)
        self.has_scale = fb['has_scale']
        self.scale = fb['scale'].to(device) if fb['has_scale'] else None
        self.__dict__.pop('_dcache', None)
        return 1

    def _pack_taps(self, wt, cin_pad):
        """wt [O, ntap, I] -> [O, K] in the kernel's k order (see vps_conv_desc.korder)"""
        O, ntap, I = wt.shape
        if self.korder == 0:                                   # tap-major: k = tap*cin_pad + ci
            wp = torch.zeros(O, ntap, cin_pad)
            wp[..., :I] = wt
            return wp.reshape(O, ntap * cin_pad)
        nch = (cin_pad + 31) // 32                             # chunk-major: k = (chunk*ntap + tap)*32 + c
        wp = torch.zeros(O, ntap, nch * 32)
        wp[..., :I] = wt
        return wp.view(O, ntap, nch, 32).permute(0, 2, 1, 3).reshape(O, nch * ntap * 32)

    def _set_weights(self, packed, device):
        """packed fp32 [nclass][cout_pad][kpad] (host or device) -> the operand format of the selected arithmetic.
        Returns None, or (f16x3) the per-output-channel factor [cout_pad] the epilogue scale has to be multiplied with."""
        if self.prec == hip.PREC_F32:
            self.w, self.w_split = packed.to(device), None
            return None
        planes, r = [], packed.to(device)
        wscale = None
        if self.prec == hip.PREC_F16X3:
            # per output channel: w' = w * 2^s with max|w'| in [2^11, 2^12) (exact), planes g0 = fp16(w'), g1 = fp16(w' - g0),
            # g2 = 2^-11 * g0 (exact): see VPS_PREC_F16X3 in include/vps_hip.h
            amax = r.abs().amax(dim=(0, 2))                                   # [cout_pad]
            e = torch.frexp(amax)[1] - 1                                      # floor(log2(amax)) for amax > 0
            sh = torch.where(amax > 0, 11 - e, torch.zeros_like(e)).clamp(-60, 60)
            r = torch.ldexp(r, sh.view(1, -1, 1).expand_as(r))
            wscale = torch.ldexp(torch.ones_like(amax), -sh)
            g0 = r.to(torch.float16)
            g1 = (r - g0.float()).to(torch.float16)
            g2 = torch.ldexp(g0.float(), torch.full_like(sh, -11).view(1, -1, 1).expand_as(r)).to(torch.float16)
            planes = [g0, g1, g2]
        else:
            for _ in range(_PLANES[self.prec]):         # term p = bf16 RNE of the residual after p terms
                h = r.to(torch.bfloat16)
                planes.append(h)
                r = r - h.float()
        ws = torch.stack(planes, 0)                     # [planes][class][cout_pad][kpad]
        self.w_thin = None
        if (THIN_KERNEL and self.prec == hip.PREC_F16X3 and not self.deform and not self.transposed and self.korder == 0 and self.KH == self.KW
                and self.KH in (3, 7) and self.cin_pad in (4, 8, 12) and self.cout_pad == 64 and self.tile_n == 64):
            self.w_thin = pack_thin(ws[:2, 0], self.KH, self.cin_pad)
        if not (self.deform and self.korder == 0):
            # MFMA-fragment order for the direct-to-register weight path (every kernel but the two-barrier deformable one, which
            # only takes the tap-major deformable layers - none on the path):
            # [plane][class][cout_pad/32][kpad/16][lane = 32*(k/8 % 2) + cout % 32][8 consecutive k]
            P, C, O, K = ws.shape
            ws = ws.view(P, C, O // 32, 32, K // 16, 2, 8).permute(0, 1, 2, 4, 5, 3, 6)
        self.w, self.w_split = None, ws.contiguous()
        return wscale

    @classmethod
    def from_matrix(cls, mat, prec=None):
        """GEMM against a device matrix mat [M, D] (rows = output channels) without a host round trip: out = x @ mat^T."""
        M, D = mat.shape
        assert D % 32 == 0
        self = cls.__new__(cls)
        self.stride, self.act, self.slope, self.deform, self.transposed = 1, hip.ACT_NONE, 0.1, False, False
        self.nclass, self.os, self.KH, self.KW = 1, 1, 1, 1
        self.pad_y = self.pad_x = (0, 0)
        self.cin = self.cin_pad = self.kpad = D
        self.korder = 0
        self.small = False
        self.cout = M
        self.tile_n = _tile_n(M)
        self.cout_pad = _ceil(M, self.tile_n)
        w = torch.zeros(1, self.cout_pad, D, dtype=torch.float32, device=mat.device)
        w[0, :M] = mat
        self.prec = DEFAULT_PREC if prec is None else prec
        self._fb, self.f16_slot = None, 0
        wscale = self._set_weights(w, mat.device)
        self.scale, self.shift, self.has_scale = None, None, False
        if wscale is not None:
            self.scale, self.has_scale = wscale[:M].contiguous(), True
        return self

    def out_hw(self, H, W):
        if self.transposed:
            return 2 * H, 2 * W
        p = self.pad_y[0]
        return (H + 2 * p - self.KH) // self.stride + 1, (W + 2 * p - self.KW) // self.stride + 1

    def __call__(self, x, out=None, ws=None, name=None, res=None, res_shift=0, offset=None, act=None, gn=None, temp=False, keep=False):
        """x: FMap. out: FMap window to write (allocated from `ws` under `name` if None; temp: as a temporary map the caller releases,
        keep: in the workspace's `out` target - see Workspace).
        gn = (stats, G): float64 tensor [GN_REP, 2*G] of zeros that receives the GroupNorm sums of the output from the epilogue
        (vps_conv_desc.gn_stats; deformable layers of the split-operand modes). `self.gn_fused` tells whether the launch took
        them (it does not when the layer is split over K) - the caller then runs the statistics pass itself."""
        if getattr(self, '_mfma_twin', None) is not None and x.N * x.H * x.W >= 100000:
            return self._mfma_twin(x, out, ws, name, res, res_shift, offset, act, gn, temp, keep)
        assert x.C == self.cin or (x.C >= self.cin and x.C <= self.cin_pad), (x.C, self.cin)
        assert x.coff % 4 == 0 and x.ld % 4 == 0 and x.coff + self.cin_pad <= x.ld, (x.coff, x.ld, self.cin_pad)
        Ho, Wo = self.out_hw(x.H, x.W)
        if out is None:
            out = ws.fmap(name, x.N, Ho, Wo, self.cout, temp=temp, out=keep)
        assert (out.N, out.H, out.W) == (x.N, Ho, Wo) and out.C == self.cout, ((out.N, out.H, out.W, out.C), (x.N, Ho, Wo, self.cout))
        # The workspace is persistent, so a layer sees the same operand addresses every frame: the filled descriptor (and the
        # split-K scratch it points to) is cached per operand set; a hit costs one dict lookup + the launch instead of ~50
        # ctypes field stores (the low-resolution layers run for 10-20 us, less than it takes to describe them).
        ckey = (x.t.data_ptr(), x.N, x.H, x.W, x.ld, x.coff, out.t.data_ptr(), out.ld, out.coff,
                None if res is None else (res.t.data_ptr(), res.ld, res.coff, res_shift),
                None if offset is None else (offset.t.data_ptr(), offset.ld), act, getattr(hip.stream_ptr(), 'value', None) or 0,
                None if gn is None else (gn[0].data_ptr(), gn[1]))
        cache = self.__dict__.setdefault('_dcache', {})
        hit = cache.get(ckey)
        if hit is not None and CONV_TRACE is None:
            self.gn_fused = bool(hit[0].gn_stats)
            hip.conv2d(hit[0])
            return out
        d = hip.ConvDesc()
        d.inp = x.t.data_ptr(); d.N, d.H, d.W = x.N, x.H, x.W
        d.in_ld, d.in_coff, d.cin_pad = x.ld, x.coff, self.cin_pad
        d.prec = self.prec
        d.korder = self.korder
        if self.prec == hip.PREC_F32:
            d.w = self.w.data_ptr()
        else:
            d.w_split = self.w_split.data_ptr()
            if getattr(self, 'w_thin', None) is not None and self.prec == hip.PREC_F16X3:
                d.w_thin = self.w_thin.data_ptr()
        d.cout, d.cout_pad, d.kpad = self.cout, self.cout_pad, self.kpad
        d.KH, d.KW, d.stride = self.KH, self.KW, self.stride
        d.pad_y[0], d.pad_y[1] = self.pad_y; d.pad_x[0], d.pad_x[1] = self.pad_x
        d.out = out.t.data_ptr(); d.Ho, d.Wo, d.out_ld, d.out_coff = Ho, Wo, out.ld, out.coff
        if self.transposed:
            d.Qh, d.Qw, d.os_y, d.os_x, d.nclass = x.H, x.W, 2, 2, 4
        else:
            d.Qh, d.Qw, d.os_y, d.os_x, d.nclass = Ho, Wo, 1, 1, 1
        d.scale = self.scale.data_ptr() if self.scale is not None else None
        d.shift = self.shift.data_ptr() if self.shift is not None else None
        if res is not None:
            d.res = res.t.data_ptr(); d.res_ld, d.res_coff, d.res_shift = res.ld, res.coff, res_shift
            assert res.C == self.cout and res.H == (Ho >> res_shift) and res.W == (Wo >> res_shift)
        d.act = self.act if act is None else act
        d.slope = self.slope
        if self.deform:
            assert offset is not None and offset.coff == 0 and offset.C >= 2 * self.KH * self.KW
            d.offset = offset.t.data_ptr(); d.off_ld = offset.ld
        d.tile_n = self.tile_n
        M = x.N * d.Qh * d.Qw
        # deformable layers with a multiple of 256 output channels on the large maps: one block computes all 256 columns of its
        # 128 pixels, so the bilinear loader runs once per pixel tile instead of once per 128 columns (conv_mfma.hip, tile_n 256)
        wide = (DCN256[0] and self.deform and self.prec == hip.PREC_F16X3 and self.korder == 1 and self.cout_pad % 256 == 0
                and (M + 127) // 128 * (self.cout_pad // 256) >= DCN256_MIN_TILES)
        tile_n = 256 if wide else self.tile_n
        d.tile_n = tile_n
        if self.prec == hip.PREC_F16X3:
            d.status = f16_status(x.t.device).data_ptr() + 4 * getattr(self, 'f16_slot', 0)
        # split-K for launches that cannot fill 256 CUs
        tiles = ((M + 127) // 128) * (self.cout_pad // tile_n) * d.nclass
        ksteps = self.kpad // 32
        ksplit = 1
        if tiles < 256 and ksteps >= 8 and not getattr(self, 'small', False):
            ksplit = max(1, min((SPLITK_TARGET_BLOCKS + tiles - 1) // tiles, ksteps // 4, 32))
            ntap = self.KH * self.KW
            if self.korder == 1:
                # chunk-major layers split over whole 32-channel chunks (vps_conv2d: `chunk_split`); the ranges may be uneven (the last
                # split is shorter) - the largest count <= target that ceil-division of the chunk count reproduces
                nch = ksteps // ntap
                ksplit = max(k for k in range(1, ksplit + 1) if -(-nch // -(-nch // k)) == k)
            else:
                per = (ksteps + ksplit - 1) // ksplit
                ksplit = (ksteps + per - 1) // per
        d.ksplit = ksplit
        self.gn_fused = False
        if gn is not None and self.deform and self.prec != hip.PREC_F32 and ksplit == 1 and res is None and not ((self.cout | out.ld | out.coff) & 3):
            cpg = self.cout // gn[1]
            if self.cout % gn[1] == 0 and (cpg == 4 or cpg % 8 == 0):
                assert gn[0].dtype == torch.float64 and gn[0].numel() == GN_REP * 2 * gn[1] and gn[0].is_contiguous()
                d.gn_stats, d.gn_cpg, d.gn_rep = gn[0].data_ptr(), cpg, GN_REP
                self.gn_fused = True
        if ksplit > 1:
            need = ksplit * d.nclass * M * self.cout_pad
            # tickets of the last-block reduction (vps_conv_desc.tile_counter): an upper bound of the tile count of every kernel family
            ntick = d.nclass * (self.cout_pad // tile_n) * max((M + 127) // 128, x.N * ((d.Qh + 7) // 8) * ((d.Qw + 15) // 16))
            if ws is not None:
                # one scratch buffer per stream: branches of the frame graph that run concurrently must not share it
                key = '__splitk_ws_%x' % (getattr(hip.stream_ptr(), 'value', None) or 0)
                wsb = ws.bufs.get(key)
                if wsb is None or wsb.numel() < need:
                    wsb = torch.empty(max(need, 1 << 24), dtype=torch.float32, device=x.t.device)
                    ws.bufs[key] = wsb
                tck = ws.bufs.get(key + '_tickets')
                if tck is None or tck.numel() < ntick:
                    tck = torch.zeros(max(ntick, 1 << 16), dtype=torch.int32, device=x.t.device)     # zero once: launches leave them zero
                    ws.bufs[key + '_tickets'] = tck
            else:
                wsb = torch.empty(need, dtype=torch.float32, device=x.t.device)
                tck = torch.zeros(ntick, dtype=torch.int32, device=x.t.device)
            d.ws = wsb.data_ptr()
            if SPLITK_LAST_BLOCK:
                d.tile_counter = tck.data_ptr()
            self._last_ws = (wsb, tck)  # keep alive until the next call
        if CONV_TRACE is not None:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            hip.conv2d(d)
            e1.record()
            CONV_TRACE.append((self.flops(x.N, x.H, x.W), e0, e1,
                               '%d->%d k%dx%d s%d %s%s n%d %dx%d tile%d ksplit%d p%d' % (self.cin, self.cout, self.KH, self.KW, self.stride,
                                                                                   'T' if self.transposed else '', 'D' if self.deform else '',
                                                                                   x.N, x.H, x.W, tile_n, ksplit, self.prec),
                               self.bytes(x.N, x.H, x.W, res is not None)))
        else:
            hip.conv2d(d)
        if len(cache) >= 16:
            cache.clear()
        cache[ckey] = (d, wsb if ksplit > 1 else None, x.t, out.t, None if res is None else res.t, None if offset is None else offset.t)
        return out

    def bytes(self, x_N, x_H, x_W, has_res=False):
        """algorithmic HBM bytes of one call: every input / weight / residual element read once, every output written once."""
        Ho, Wo = self.out_hw(x_H, x_W)
        wbytes = self.nclass * self.cout * self.KH * self.KW * self.cin * (4 if self.prec == hip.PREC_F32 else 2 * _PLANES[self.prec])
        return 4.0 * x_N * (x_H * x_W * self.cin + Ho * Wo * self.cout * (2 if has_res else 1)) + wbytes

    def flops(self, x_N, x_H, x_W):
        """algorithmic FLOPs (2*MACs, unpadded) of one call on an input of that size."""
        if self.transposed:
            return 2.0 * x_N * x_H * x_W * 4 * self.cout * self.cin * self.KH * self.KW
        Ho, Wo = self.out_hw(x_H, x_W)
        return 2.0 * x_N * Ho * Wo * self.cout * self.cin * self.KH * self.KW


def pack_conv_module(m, bn=None, act=hip.ACT_NONE, slope=0.1, device='cuda', deform=False, prec=None):
    """torch.nn.Conv2d / ConvTranspose2d (used purely as a parameter container) -> PackedConv."""
    import torch.nn as nn
    bnd = None
    if bn is not None:
        bnd = dict(weight=bn.weight, bias=bn.bias, running_mean=bn.running_mean, running_var=bn.running_var, eps=bn.eps)
    if isinstance(m, nn.ConvTranspose2d):
        return PackedConv(m.weight, m.bias, bnd, stride=m.stride[0], padding=m.padding[0], act=act, slope=slope,
                          transposed=True, device=device, prec=prec)
    return PackedConv(m.weight, m.bias, bnd, stride=m.stride[0], padding=m.padding[0], act=act, slope=slope,
                      device=device, deform=deform, prec=prec)


def pack_linear(weight, bias, act=hip.ACT_NONE, device='cuda', chw=None, prec=None):
    """nn.Linear as a 1x1 conv over `rows` pixels. chw=(C,S): the reference flattens NCHW [C, S=h*w] while the NHWC
    RoI features flatten as [S, C] -> permute the weight columns once at pack time."""
    w = weight.detach().float().cpu()
    if chw is not None:
        C, S = chw
        w = w.view(w.shape[0], C, S).permute(0, 2, 1).reshape(w.shape[0], C * S)
    return PackedConv(w.view(w.shape[0], w.shape[1], 1, 1), bias, None, 1, 0, act=act, device=device, prec=prec)


# ------------------------------------------------------------------------------------------------------------
# thin FMap-level wrappers of the other kernels
# ------------------------------------------------------------------------------------------------------------
def resize(x, out, mode='bilinear', alpha=1.0):
    hip.check(hip.load().vps_resize(x.ptr(), x.ld, x.coff, x.H, x.W, out.ptr(), out.ld, out.coff, out.H, out.W,
                                    x.N, x.C, 0 if mode == 'bilinear' else 1, float(alpha), hip.stream_ptr()), 'vps_resize')
    return out


def pool3x3s2(x, out, mode='max'):
    hip.check(hip.load().vps_pool3x3s2(x.ptr(), x.ld, x.coff, x.H, x.W, out.ptr(), out.ld, out.coff, x.N, x.C,
                                       0 if mode == 'max' else 1, hip.stream_ptr()), 'vps_pool3x3s2')
    return out


# f16x3 mode: the two correlations of the path run in split fp16 on the matrix cores (vps_correlation_f16, csrc/corr_mfma.hip) and
# report operands beyond the fp16 range in their own status slot; `f16_fallback` then switches them to the exact kernels for good.
# VPS_CORR_F16=0: always the exact vector-ALU kernels.
CORR_F16 = [os.environ.get('VPS_CORR_F16', '1') != '0']
F16_CORR_SLOT = F16_SLOTS - 1


def correlation(x1, x2, out, max_disp, stride2, act=hip.ACT_NONE, slope=0.1, prec=None):
    """prec: the arithmetic of the calling model's contractions (hip.PREC_*); PREC_F16X3 selects the split-fp16 MFMA kernel where one
    exists, everything else the exact fp32 kernels"""
    assert x1.C == x2.C
    if prec == hip.PREC_F16X3 and CORR_F16[0]:
        st = c_void_p(f16_status(x1.t.device).data_ptr() + 4 * F16_CORR_SLOT)
        hip.check(hip.load().vps_correlation_f16(x1.ptr(), x1.ld, x1.coff, x2.ptr(), x2.ld, x2.coff, out.ptr(), out.ld, out.coff,
                                                 x1.N, x1.H, x1.W, x1.C, max_disp, stride2, act, float(slope), st, hip.stream_ptr()),
                  'vps_correlation_f16')
        return out
    hip.check(hip.load().vps_correlation(x1.ptr(), x1.ld, x1.coff, x2.ptr(), x2.ld, x2.coff, out.ptr(), out.ld, out.coff,
                                         x1.N, x1.H, x1.W, x1.C, max_disp, stride2, act, float(slope), hip.stream_ptr()),
              'vps_correlation')
    return out


def flow_warp(x, flow, out):
    hip.check(hip.load().vps_flow_warp(x.ptr(), x.ld, x.coff, flow.ptr(), flow.ld, flow.coff, out.ptr(), out.ld, out.coff,
                                       x.N, x.H, x.W, x.C, hip.stream_ptr()), 'vps_flow_warp')
    return out


def bfp_gather(levels, out):
    n = len(levels)
    ptrs = (c_void_p * n)(*[l.t.data_ptr() for l in levels])
    lds = (c_int * n)(*[l.ld for l in levels])
    ratios = (c_int * n)(*[levels[0].H // l.H for l in levels])
    for l in levels:
        assert l.coff == 0
    hip.check(hip.load().vps_bfp_gather(ptrs, lds, ratios, n, out.ptr(), out.ld, out.coff, out.N, out.H, out.W, out.C,
                                        hip.stream_ptr()), 'vps_bfp_gather')
    return out


def bfp_scatter_all(bsf, levels, outs):
    """every level of the pyramid in one pass over `bsf` (vps_bfp_scatter_all); -> outs, or None when the shapes are not the kernel's
    (level l at ratio 2^l, whole 2^(L-1) cells, C % 32 == 0): the caller then scatters level by level"""
    L = len(levels)
    if not (2 <= L <= 5 and bsf.C % 32 == 0 and bsf.coff == 0 and all(lv.coff == 0 and o.coff == 0 for lv, o in zip(levels, outs))
            and bsf.H % (1 << (L - 1)) == 0 and bsf.W % (1 << (L - 1)) == 0
            and all(lv.H == bsf.H >> l and lv.W == bsf.W >> l and lv.C == bsf.C for l, lv in enumerate(levels))):
        return None
    lp = (ctypes.c_void_p * L)(*[lv.t.data_ptr() for lv in levels]); op = (ctypes.c_void_p * L)(*[o.t.data_ptr() for o in outs])
    ll = (ctypes.c_int * L)(*[lv.ld for lv in levels]); ol = (ctypes.c_int * L)(*[o.ld for o in outs])
    hip.check(hip.load().vps_bfp_scatter_all(bsf.ptr(), bsf.ld, lp, ll, op, ol, L, bsf.N, bsf.H, bsf.W, bsf.C, hip.stream_ptr()), 'vps_bfp_scatter_all')
    return outs


def bfp_scatter(bsf, level, out):
    assert bsf.coff == 0 and level.coff == 0 and out.coff == 0
    hip.check(hip.load().vps_bfp_scatter(bsf.ptr(), bsf.ld, level.ptr(), level.ld, out.ptr(), out.ld, bsf.N, bsf.H, bsf.W,
                                         bsf.C, bsf.H // level.H, hip.stream_ptr()), 'vps_bfp_scatter')
    return out


def groupnorm_relu(x, out, G, gamma, beta, eps, stats, relu=True, stats_ready=False):
    """stats: 2*G doubles of scratch; stats_ready: it already holds the [GN_REP, 2*G] partial sums the producing conv's epilogue took"""
    assert x.coff == 0 and x.N == 1
    args = (x.ptr(), x.ld, out.ptr(), out.ld, out.coff, x.npix, x.C, G, hip.ptr(gamma), hip.ptr(beta), float(eps), 1 if relu else 0, hip.ptr(stats))
    if stats_ready:
        hip.check(hip.load().vps_groupnorm_apply(*args, GN_REP, hip.stream_ptr()), 'vps_groupnorm_apply')
    else:
        hip.check(hip.load().vps_groupnorm_relu(*args, hip.stream_ptr()), 'vps_groupnorm_relu')
    return out


def roi_align(levels, strides, rois, P, sample_num=2, finest_scale=56.0, out=None):
    """levels: list of FMap (coff 0); rois: device [R,5] -> device tensor [R,P,P,C] (NHWC)."""
    n = len(levels)
    R = rois.shape[0]
    C = levels[0].C
    if out is None:
        out = torch.empty(R, P, P, C, dtype=torch.float32, device=rois.device)
    ptrs = (c_void_p * n)(*[l.t.data_ptr() for l in levels])
    lds = (c_int * n)(*[l.ld for l in levels])
    Hs = (c_int * n)(*[l.H for l in levels])
    Ws = (c_int * n)(*[l.W for l in levels])
    sc = (c_float * n)(*[1.0 / s for s in strides])
    rois = rois.contiguous()
    hip.check(hip.load().vps_roi_align(ptrs, lds, Hs, Ws, sc, n, float(finest_scale), hip.ptr(rois), R, C, P, sample_num,
                                       hip.ptr(out), hip.stream_ptr()), 'vps_roi_align')
    return out


def tcea_temporal(emb, emb_ref, fea0, fea1, out):
    C = fea0.C
    assert emb.C == 2 * C and emb.coff == 0 and emb_ref.coff == 0 and out.coff == 0 and out.C == 2 * C
    assert fea0.coff % 4 == 0 and fea1.coff % 4 == 0
    p0 = c_void_p(fea0.t.data_ptr() + 4 * fea0.coff)
    p1 = c_void_p(fea1.t.data_ptr() + 4 * fea1.coff)
    hip.check(hip.load().vps_tcea_temporal(emb.ptr(), emb.ld, emb_ref.ptr(), emb_ref.ld, p0, fea0.ld, p1, fea1.ld,
                                           out.ptr(), out.ld, out.npix, C, hip.stream_ptr()), 'vps_tcea_temporal')
    return out


def tcea_modulate(fea, att, att_add, out):
    if all(m.coff == 0 and m.C == m.ld for m in (fea, att, att_add, out)):
        hip.check(hip.load().vps_tcea_modulate(fea.ptr(), att.ptr(), att_add.ptr(), out.ptr(), out.t.numel(), hip.stream_ptr()),
                  'vps_tcea_modulate')
        return out
    # channel windows (the fused feature is one half of the merged fea_fusion | sAtt_1 output)
    w = lambda m: c_void_p(m.t.data_ptr() + 4 * m.coff)
    assert fea.C == att.C == att_add.C == out.C and all(m.coff % 4 == 0 for m in (fea, att, att_add, out))
    hip.check(hip.load().vps_tcea_modulate_ld(w(fea), fea.ld, w(att), att.ld, w(att_add), att_add.ld, w(out), out.ld, out.npix, out.C,
                                              hip.stream_ptr()), 'vps_tcea_modulate_ld')
    return out
