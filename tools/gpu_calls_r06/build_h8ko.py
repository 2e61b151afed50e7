"""Builds the measurement-only knock-out library of call_h8ko.sh: a copy of vps_amd/csrc/conv_h8.hip whose stride-1 kernel takes a bit mask
(VPS_H8_KO: 1 no activation loads, 2 no activation staging, 4 no weight loads, 8 no weight staging, 16 no barrier), linked with the other
objects of the product build into build/h8ko/libvpship.so (select it with VPS_HIP_LIB). Results are garbage; only the timing is used.
Note: with VPS_H8P=1 (default) the f16x3 3x3 layers run conv_h8p.hip - set VPS_H8P=0 to time this kernel.

    make -C vps_amd/csrc && python tools/gpu_calls_r06/build_h8ko.py
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
C = os.path.join(ROOT, 'vps_amd', 'csrc')
OUT = os.path.join(ROOT, 'build', 'h8ko')
os.makedirs(OUT, exist_ok=True)
s = open(os.path.join(C, 'conv_h8.hip')).read()
rep = [
    ("void conv_mfma_h8_kernel(const vps_conv_desc d, const int tiles_m, const int tiles_n, const int chunks_per_split) {",
     "void conv_mfma_h8_kernel(const vps_conv_desc d, const int tiles_m, const int tiles_n, const int chunks_per_split, const int ko) {"),
    ("    auto load_A = [&]() {\n        const bool kv = achunk * BK + k4 * 4 < cin_pad;", "    auto load_A = [&]() {\n        if (ko & 1) return;\n        const bool kv = achunk * BK + k4 * 4 < cin_pad;"),
    ("    auto store_A = [&](int i, int buf) {\n        x4 sp[NSA];\n        split_act<MODE>(areg[i], sp, amax);", "    auto store_A = [&](int i, int buf) {\n        if (ko & 2) return;\n        x4 sp[NSA];\n        split_act<MODE>(areg[i], sp, amax);"),
    ("    auto load_B = [&](int step) {\n#pragma unroll\n        for (int j = 0; j < NBL; ++j) {\n            const int f = __builtin_amdgcn_readfirstlane((t + 512 * j) >> 6)",
     "    auto load_B = [&](int step) {\n        if (ko & 4) return;\n#pragma unroll\n        for (int j = 0; j < NBL; ++j) {\n            const int f = __builtin_amdgcn_readfirstlane((t + 512 * j) >> 6)"),
    ("    auto store_B = [&](int buf) {\n#pragma unroll\n        for (int j = 0; j < NBL; ++j) *reinterpret_cast<x8*>(&Bs[buf * BBUF + (t + 512 * j) * 8]) = breg[j];",
     "    auto store_B = [&](int buf) {\n        if (ko & 8) return;\n#pragma unroll\n        for (int j = 0; j < NBL; ++j) *reinterpret_cast<x8*>(&Bs[buf * BBUF + (t + 512 * j) * 8]) = breg[j];"),
    ("            __syncthreads();                                    // weight buffers alternate per tap (and, after the last tap, the chunk's)",
     "            if (!(ko & 16)) __syncthreads();"),
    ("hipLaunchKernelGGL((conv_mfma_h8_kernel<MODE, K, K>), dim3((unsigned)nblk8), dim3(512), 0, s, d, tiles_m8, tiles_n, chunks_per_split)",
     "hipLaunchKernelGGL((conv_mfma_h8_kernel<MODE, K, K>), dim3((unsigned)nblk8), dim3(512), 0, s, d, tiles_m8, tiles_n, chunks_per_split, (getenv(\"VPS_H8_KO\") ? atoi(getenv(\"VPS_H8_KO\")) : 0))"),
]
for a, b in rep:
    assert a in s, a[:60]
    s = s.replace(a, b, 1)
src = os.path.join(C, '_conv_h8_ko.hip')
open(src, 'w').write(s)
try:
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function', '-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
    subprocess.check_call(['/opt/rocm/bin/hipcc'] + flags + ['-c', src, '-o', os.path.join(OUT, 'conv_h8_ko.o')], cwd=C)
finally:
    os.remove(src)
objs = [o for o in sorted(os.listdir(C)) if o.endswith('.o') and o != 'conv_h8.o']
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC'] + [os.path.join(C, o) for o in objs] +
                      [os.path.join(OUT, 'conv_h8_ko.o'), '-lz', '-o', os.path.join(OUT, 'libvpship.so')])
print('built', os.path.join(OUT, 'libvpship.so'))
