"""ctypes binding of libvpship.so (the C-ABI declared in include/vps_hip.h).

The library is the product: if it is missing or does not export a declared symbol this module raises at import
of the first kernel call — there is NO CPU / PyTorch fallback anywhere in `vps_amd`.
PyTorch is used for device memory (`tensor.data_ptr()`), streams and `torch.distributed` only.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, byref, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint64,
                    c_void_p)

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# VPS_HIP_LIB: developer override to load an experimental build of the same ABI (kernel A/B timing)
LIB_PATH = os.environ.get('VPS_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libvpship.so')
ABI_VERSION = 18

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
PREC_F32, PREC_BF16, PREC_BF16X3, PREC_BF16X6, PREC_F16X3 = 0, 1, 2, 3, 4

# every symbol include/vps_hip.h declares (checked by tests/test_cabi.py without a GPU)
SYMBOLS = [
    'vps_abi_version', 'vps_build_info', 'vps_conv2d', 'vps_resample2d', 'vps_channelnorm', 'vps_correlation',
    'vps_flow_warp', 'vps_nchw_to_nhwc', 'vps_nhwc_to_nchw', 'vps_resize', 'vps_pool3x3s2', 'vps_bfp_gather',
    'vps_bfp_scatter', 'vps_bfp_scatter_all', 'vps_axpb', 'vps_flow_prep', 'vps_flow_prep_pad', 'vps_flow_stage', 'vps_flow_stage_full', 'vps_groupnorm_relu', 'vps_groupnorm_apply', 'vps_tcea_temporal',
    'vps_tcea_modulate', 'vps_tcea_modulate_ld', 'vps_correlation_f16', 'vps_roi_align', 'vps_nms_batched', 'vps_delta2bbox', 'vps_bbox_overlaps',
    'vps_row_softmax', 'vps_mask_count', 'vps_mask_commit', 'vps_mask_removal', 'vps_mask_removal_dep', 'vps_mask_removal_hist', 'vps_frame_tail', 'vps_mask_level', 'vps_panoptic_combine',
    'vps_panoptic_combine_dev', 'vps_rpn_select', 'vps_rpn_collect', 'vps_maskroi_select', 'vps_maskroi_finish', 'vps_track_assign', 'vps_pan_instances',
    'vps_unify_hist', 'vps_unify_tables', 'vps_unify_write', 'vps_image_prep', 'vps_resize_u8', 'vps_segment_stats', 'vps_segment_paint', 'vps_pair_count',
    'vps_png_info', 'vps_png_decode_bgr8',
]


class ConvDesc(Structure):
    _fields_ = [
        ('inp', c_void_p), ('N', c_int32), ('H', c_int32), ('W', c_int32),
        ('in_ld', c_int32), ('in_coff', c_int32), ('cin_pad', c_int32),
        ('w', c_void_p), ('cout', c_int32), ('cout_pad', c_int32), ('kpad', c_int32),
        ('KH', c_int32), ('KW', c_int32), ('stride', c_int32),
        ('pad_y', c_int32 * 2), ('pad_x', c_int32 * 2),
        ('out', c_void_p), ('Ho', c_int32), ('Wo', c_int32), ('out_ld', c_int32), ('out_coff', c_int32),
        ('Qh', c_int32), ('Qw', c_int32), ('os_y', c_int32), ('os_x', c_int32), ('nclass', c_int32),
        ('scale', c_void_p), ('shift', c_void_p), ('res', c_void_p),
        ('res_ld', c_int32), ('res_coff', c_int32), ('res_shift', c_int32),
        ('act', c_int32), ('slope', c_float),
        ('offset', c_void_p), ('off_ld', c_int32),
        ('tile_n', c_int32), ('ksplit', c_int32), ('ws', c_void_p),
        ('prec', c_int32), ('w_split', c_void_p), ('korder', c_int32), ('status', c_void_p),
        ('gn_stats', c_void_p), ('gn_cpg', c_int32), ('gn_rep', c_int32), ('tile_counter', c_void_p), ('w_thin', c_void_p),
    ]


class Tensor4(Structure):
    _fields_ = [('p', c_void_p), ('sn', c_int64), ('sc', c_int64), ('sh', c_int64), ('sw', c_int64)]


class PanInst(Structure):
    _fields_ = [('sx0', c_int32), ('sy0', c_int32), ('sx1', c_int32), ('sy1', c_int32), ('seg_ch', c_int32),
                ('bx1', c_int32), ('by1', c_int32), ('bx2', c_int32), ('by2', c_int32), ('mask_idx', c_int32)]


_lib = None
_host = None


class VpsHipError(RuntimeError):
    pass


def csrc_sha16():
    """sha256[:16] over the CODE of the library's kernel sources (csrc/*.hip, *.h, *.cpp, Makefile; comments and white space are
    dropped first, so that a comment edit does not un-stamp a measurement): stamps measurements with what they ran on"""
    import glob
    import hashlib
    import re
    d = os.path.dirname(LIB_PATH)
    hsh = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(d, '*.hip')) + glob.glob(os.path.join(d, '*.h')) + glob.glob(os.path.join(d, '*.cpp')) + [os.path.join(d, 'Makefile')]):
        txt = open(fn, 'r', errors='replace').read()
        if not fn.endswith('Makefile'):
            txt = re.sub(r'/\*.*?\*/', ' ', txt, flags=re.S)
            txt = re.sub(r'//[^\n]*', ' ', txt)
        else:
            txt = re.sub(r'(?m)^#[^\n]*', ' ', txt)
        hsh.update(os.path.basename(fn).encode()); hsh.update(' '.join(txt.split()).encode())
    return hsh.hexdigest()[:16]


def load_host():
    """the handle whose calls release the interpreter lock: the host-side functions (PNG decode) that decode threads run in parallel"""
    load()
    return _host


def load():
    """Load libvpship.so (no GPU needed to load). Raises loudly if absent or ABI-mismatched."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VpsHipError('libvpship.so not found at %s — build it with `python -c "import __graft_entry__ as g; '
                          'g.build()"` or `make -C vps_amd/csrc`. There is no CPU fallback.' % LIB_PATH)
    # PyDLL: the calls are made WITH the interpreter lock held. Every entry point but the two host-side PNG functions only enqueues
    # work (microseconds); releasing the lock around each of the ~560 launches of a frame lets any other Python thread (the decode
    # pool of pipeline.ClipFeeder) take it in between, and getting it back costs a condition-variable round trip per launch: the
    # frame went from 21.6 to 38.5 ms beside six decode threads. The PNG functions are bound on a second, lock-releasing handle.
    lib = ctypes.PyDLL(LIB_PATH)
    missing = [s for s in SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise VpsHipError('libvpship.so is missing symbols: %s' % missing)
    lib.vps_abi_version.restype = c_int
    lib.vps_build_info.restype = c_char_p
    if lib.vps_abi_version() != ABI_VERSION:
        raise VpsHipError('libvpship.so ABI %d != binding ABI %d' % (lib.vps_abi_version(), ABI_VERSION))
    for s in SYMBOLS[2:]:
        getattr(lib, s).restype = c_int
    lib.vps_conv2d.argtypes = [POINTER(ConvDesc), c_void_p]
    lib.vps_resample2d.argtypes = [Tensor4, Tensor4, Tensor4, c_int, c_int, c_int, c_int, c_void_p]
    lib.vps_channelnorm.argtypes = [Tensor4, Tensor4, c_int, c_int, c_int, c_int, c_void_p]
    lib.vps_correlation.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int,
                                    c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]
    lib.vps_correlation_f16.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int,
                                        c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]
    lib.vps_flow_warp.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int,
                                  c_int, c_int, c_int, c_int, c_void_p]
    lib.vps_nchw_to_nhwc.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.vps_nhwc_to_nchw.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    lib.vps_resize.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int,
                               c_int, c_int, c_int, c_float, c_void_p]
    lib.vps_pool3x3s2.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                  c_void_p]
    lib.vps_bfp_gather.argtypes = [POINTER(c_void_p), POINTER(c_int), POINTER(c_int), c_int, c_void_p, c_int, c_int,
                                   c_int, c_int, c_int, c_int, c_void_p]
    lib.vps_bfp_scatter.argtypes = [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                    c_int, c_void_p]
    lib.vps_bfp_scatter_all.argtypes = [c_void_p, c_int, POINTER(c_void_p), POINTER(c_int), POINTER(c_void_p), POINTER(c_int), c_int, c_int, c_int,
                                        c_int, c_int, c_void_p]
    lib.vps_axpb.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int64, c_int, c_float, c_float,
                             c_void_p]
    lib.vps_flow_prep.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                  c_int, c_void_p, c_void_p]
    lib.vps_flow_prep_pad.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                      c_int, c_void_p, c_void_p]
    lib.vps_flow_stage.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
                                   c_void_p, c_int, c_int, c_float, c_int, c_int, c_int, c_int, c_void_p]
    lib.vps_flow_stage_full.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                        c_void_p, c_int, c_void_p]
    lib.vps_groupnorm_relu.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int, c_int64, c_int, c_int, c_void_p, c_void_p,
                                       c_float, c_int, c_void_p, c_void_p]
    lib.vps_groupnorm_apply.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int, c_int64, c_int, c_int, c_void_p, c_void_p,
                                       c_float, c_int, c_void_p, c_int, c_void_p]
    lib.vps_tcea_temporal.argtypes = [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                      c_int64, c_int, c_void_p]
    lib.vps_tcea_modulate.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]
    lib.vps_tcea_modulate_ld.argtypes = [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_void_p]
    lib.vps_roi_align.argtypes = [POINTER(c_void_p), POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_float),
                                  c_int, c_float, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.vps_nms_batched.argtypes = [c_void_p, c_int, c_int, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.vps_delta2bbox.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float,
                                   c_float, c_float, c_void_p]
    lib.vps_bbox_overlaps.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]
    lib.vps_row_softmax.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
    lib.vps_mask_count.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                   c_void_p]
    lib.vps_mask_commit.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                    c_double, c_void_p, c_void_p]
    lib.vps_mask_removal.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                     c_double, c_void_p, c_void_p]
    lib.vps_mask_removal_dep.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                         c_double, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.vps_mask_removal_hist.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, ctypes.c_size_t,
                                          c_double, c_void_p, c_void_p, c_void_p]
    lib.vps_frame_tail.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.vps_mask_level.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                   c_void_p, c_double, c_void_p, c_void_p]
    lib.vps_panoptic_combine.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int,
                                         c_void_p, c_void_p, c_int, c_int, c_void_p]
    lib.vps_panoptic_combine_dev.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                             c_void_p, c_void_p, c_int, c_int, c_void_p]
    global _host
    _host = ctypes.CDLL(LIB_PATH)                 # same library; these calls run WITHOUT the interpreter lock (long host work)
    for h in (lib, _host):
        h.vps_png_info.restype = h.vps_png_decode_bgr8.restype = c_int
        h.vps_png_info.argtypes = [c_void_p, c_int64, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32)]
        h.vps_png_decode_bgr8.argtypes = [c_void_p, c_int64, c_void_p, c_int64]
    lib.vps_rpn_select.argtypes = [POINTER(c_void_p), POINTER(c_int32), POINTER(c_void_p), POINTER(c_int32), POINTER(c_int32), POINTER(c_int32),
                                   POINTER(c_float), c_int, c_int, c_void_p, c_int, POINTER(c_float), c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.vps_rpn_collect.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]
    lib.vps_maskroi_select.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_float, POINTER(c_float), c_float, c_float,
                                       c_void_p, c_void_p, c_void_p, c_void_p]
    lib.vps_maskroi_finish.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.vps_track_assign.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p]
    lib.vps_pan_instances.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, POINTER(c_int32), c_int, c_void_p, c_void_p, c_void_p,
                                      c_void_p]
    lib.vps_unify_hist.argtypes = [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]
    lib.vps_unify_tables.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p]
    lib.vps_unify_write.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]
    lib.vps_image_prep.argtypes = [c_void_p, c_int, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), c_int,
                                   ctypes.c_float, c_void_p, c_void_p]
    lib.vps_resize_u8.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]
    lib.vps_segment_stats.argtypes = [c_void_p, c_int, c_int, c_void_p, c_void_p]
    lib.vps_segment_paint.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]
    lib.vps_pair_count.argtypes = [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]
    _lib = lib
    return lib


def build_info():
    return load().vps_build_info().decode()


def stream_ptr():
    """hipStream_t of torch's current stream (the kernels are launched on it; no hidden sync)."""
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def check(code, what):
    if code != 0:
        raise VpsHipError('%s failed with code %d (negative hipError_t, or -1000-x = bad argument x)' % (what, code))


def ptr(t):
    """device pointer of a CUDA fp32/.. tensor (or None)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise VpsHipError('libvpship kernels need device tensors (got a CPU tensor); there is no CPU path')
    return c_void_p(t.data_ptr())


def tensor4_nchw(t):
    """View an NCHW (any strides) 4-d tensor as a vps_tensor4."""
    assert t.dim() == 4 and t.dtype == torch.float32
    if not t.is_cuda:
        raise VpsHipError('libvpship kernels need device tensors')
    s = t.stride()
    return Tensor4(c_void_p(t.data_ptr()), s[0], s[1], s[2], s[3])


def conv2d(desc):
    check(load().vps_conv2d(byref(desc), stream_ptr()), 'vps_conv2d')
