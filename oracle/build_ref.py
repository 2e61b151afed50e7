"""TEST INFRASTRUCTURE ONLY — builds oracle/_ref/libvpsref.so from the REFERENCE's own native kernel sources.

    python oracle/build_ref.py            # needs /root/reference and hipcc; writes oracle/_ref/libvpsref.so (+ manifest)

The reference's CUDA extension ops cannot be built with their own build system here (CUDA 10 + ATen-CUDA
of torch 1.4), but the `__global__` kernel bodies are short plain CUDA C that hipcc compiles for gfx950. This recipe

  * reads the kernel bodies from the sources WHERE THEY LIE under /root/reference (line ranges below, each checked against
    the text expected on its first line so a moved file fails loudly), writes them as include fragments into a temporary
    directory that is deleted afterwards (no reference source is ever written into this repository, tracked or ignored),
  * compiles them together with oracle/ref_harness.hip — our ~150-line replacement for the ATen launchers: `extern "C"`
    entry points taking raw device pointers that launch the reference kernels with the reference's own grid/block shapes —
  * and compiles the UPSNet `nms_kernel.cu` (raw cudaMalloc/cudaMemcpy API, no ATen) as a whole translation unit from its
    original location with a force-included cuda->hip name map (oracle/ref_cuda_names.h).

Only the `.so` lands in oracle/_ref/ (git-ignored, NOT gpurun-ignored: it travels to the GPU box where /root/reference
does not exist). Only tests load it (oracle/ref_native.py); the product (vps_amd/) never does.

Wave-width note: correlation_cuda_kernel.cu hard-codes 32-lane warps (THREADS_PER_BLOCK 32, FULL_MASK, offsets 16..1,
`blockDim.x == warpSize`). The harness compiles it with `warpSize` = 32 and `__shfl_down_sync` -> `__shfl_down(.., 32)`, so
a 32-thread block reduces exactly like one CUDA warp (same partial-sum tree, same rounding) on the lower half of a wave64.
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('VPS_REFERENCE_ROOT', '/root/reference')
OUT = os.path.join(HERE, '_ref')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')

# fragment name -> (source relative to REF/mmdet, [(first_line, last_line, text the first line must start with)])
FRAGMENTS = {
    'ref_correlation.inc': ('models/flow_modules/correlation_package/correlation_cuda_kernel.cu', [
        (5, 7, '#define CUDA_NUM_THREADS 1024'),                 # CUDA_NUM_THREADS, THREADS_PER_BLOCK 32, FULL_MASK
        (16, 147, 'template<typename scalar_t>'),                # warpReduceSum, blockReduceSum, channels_first, correlation_forward
    ]),
    'ref_resample2d.inc': ('models/flow_modules/resample2d_package/resample2d_kernel.cu', [
        (5, 13, '#define CUDA_NUM_THREADS 512'),                 # DIM0..3 / DIM3_INDEX
        (15, 72, 'template <typename scalar_t>'),                # kernel_resample2d_update_output
    ]),
    'ref_channelnorm.inc': ('models/flow_modules/channelnorm_package/channelnorm_kernel.cu', [
        (18, 60, 'template <typename scalar_t>'),                # kernel_channelnorm_update_output (macros shared with resample2d)
    ]),
    'ref_roi_align.inc': ('ops/roi_align/src/roi_align_kernel.cu', [
        (4, 6, '#define CUDA_1D_KERNEL_LOOP(i, n)'),
        (16, 124, 'template <typename scalar_t>'),               # bilinear_interpolate, ROIAlignForward
    ]),
    'ref_deform.inc': ('ops/dcn/src/deform_conv_cuda_kernel.cu', [
        (71, 73, '#define CUDA_KERNEL_LOOP(i, n)'),
        (83, 114, 'template <typename scalar_t>'),               # deformable_im2col_bilinear
        (189, 242, 'template <typename scalar_t>'),              # deformable_im2col_gpu_kernel
    ]),
    'ref_nms.inc': ('ops/nms/src/nms_kernel.cu', [
        (11, 67, 'int const threadsPerBlock'),                   # threadsPerBlock, devIoU, nms_kernel
    ]),
    # host-side greedy reduce of nms_cuda (statements only; included INSIDE a harness function that declares the same locals)
    'ref_nms_reduce.inc': ('ops/nms/src/nms_kernel.cu', [
        (111, 123, '  int num_to_keep = 0;'),
    ]),
}
UPSNET_NMS = 'models/utils/upsnet/nms/nms_kernel.cu'


def _sha(path):
    with open(path, 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def build(verbose=True):
    if not os.path.isdir(REF):
        raise RuntimeError('%s not present: oracle/_ref can only be built where the reference is mounted' % REF)
    tmp = tempfile.mkdtemp(prefix='vps_ref_build_')
    manifest = {'sources': {}, 'hipcc': HIPCC}
    try:
        for frag, (rel, ranges) in FRAGMENTS.items():
            src = os.path.join(REF, 'mmdet', rel)
            with open(src) as f:
                lines = f.read().split('\n')
            out = []
            for a, b, head in ranges:
                assert lines[a - 1].startswith(head), '%s:%d is %r, expected %r — the reference moved' % (rel, a, lines[a - 1], head)
                out.extend(lines[a - 1:b])
            with open(os.path.join(tmp, frag), 'w') as f:
                f.write('\n'.join(out) + '\n')
            manifest['sources'][rel] = _sha(src)
        os.makedirs(OUT, exist_ok=True)
        ups = os.path.join(REF, 'mmdet', UPSNET_NMS)
        manifest['sources'][UPSNET_NMS] = _sha(ups)
        common = [HIPCC, '--offload-arch=gfx950', '-O2', '-std=c++17', '-fPIC', '-w']
        o1, o2 = os.path.join(tmp, 'harness.o'), os.path.join(tmp, 'upsnet_nms.o')
        cmds = [
            common + ['-I', tmp, '-c', os.path.join(HERE, 'ref_harness.hip'), '-o', o1],
            # the UPSNet kernel + host function (_nms) straight from its file: only cuda* runtime names need mapping
            common + ['-x', 'hip', '-include', os.path.join(HERE, 'ref_cuda_names.h'), '-I', os.path.dirname(ups), '-c', ups, '-o', o2],
            [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', o1, o2, '-o', os.path.join(OUT, 'libvpsref.so')],
        ]
        for c in cmds:
            if verbose:
                print(' '.join(c))
            subprocess.check_call(c)
        with open(os.path.join(OUT, 'MANIFEST.json'), 'w') as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return os.path.join(OUT, 'libvpsref.so')


if __name__ == '__main__':
    print('built', build())
    sys.exit(0)
