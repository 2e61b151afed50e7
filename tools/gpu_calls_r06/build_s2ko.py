"""Builds the measurement-only knock-out library of call_s2ko.sh: a copy of vps_amd/csrc/conv_h8.hip whose stride-2 phase-split kernel takes
a bit mask (VPS_S2_KO: 1 no activation loads, 2 no activation staging, 4 no weight loads, 8 no weight staging, 16 no barrier, 32 no MFMA),
linked with the other objects of the product build into build/s2ko/libvpship.so (select it with VPS_HIP_LIB). Results are garbage; only
the timing is used. Measured on `64->64 3x3 s2 @1024x2048` (0.283 ms in this build): everything off 0.095 ms (fragment reads, epilogue,
prologue), no MFMA 0.236, no activation loads 0.187, no loads and no staging 0.159 - the parts add up almost serially (one block of two
waves per SIMD, a barrier per tap of 12 MFMAs); an instance with activations two stages and weights three taps ahead timed the same.

    make -C vps_amd/csrc && python tools/gpu_calls_r06/build_s2ko.py
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
C = os.path.join(ROOT, 'vps_amd', 'csrc')
OUT = os.path.join(ROOT, 'build', 's2ko')
os.makedirs(OUT, exist_ok=True)
s = open(os.path.join(C, 'conv_h8.hip')).read()
i = s.index('void conv_mfma_h8s2_kernel')
head, tail = s[:i], s[i:]
KO = '(getenv("VPS_S2_KO") ? atoi(getenv("VPS_S2_KO")) : 0)'
rep = [
    ("void conv_mfma_h8s2_kernel(const vps_conv_desc d, const int tiles_m, const int tiles_n, const int chunks_per_split) {",
     "void conv_mfma_h8s2_kernel(const vps_conv_desc d, const int tiles_m, const int tiles_n, const int chunks_per_split, const int ko) {"),
    ("        if (++aph == 4) { aph = 0; ++achunk; }\n", "        if (++aph == 4) { aph = 0; ++achunk; }\n        if (ko & 1) return;\n"),
    ("    auto store_A = [&](int i, int buf) {\n", "    auto store_A = [&](int i, int buf) {\n        if (ko & 2) return;\n"),
    ("    auto load_B = [&](int wstep) {\n", "    auto load_B = [&](int wstep) {\n        if (ko & 4) return;\n"),
    ("    auto store_B = [&](int buf) {\n", "    auto store_B = [&](int buf) {\n        if (ko & 8) return;\n"),
    ("            __syncthreads();\n        });", "            if (!(ko & 16)) __syncthreads();\n        });"),
    ("                            acc[a][b] = split_mfma<MODE>(bcur[m][SM::PB[q]][b], af[m][SM::PA[q]][a], acc[a][b]);",
     "                            if (!(ko & 32)) acc[a][b] = split_mfma<MODE>(bcur[m][SM::PB[q]][b], af[m][SM::PA[q]][a], acc[a][b]);"),
]
for a, b in rep:
    assert a in tail, a[:70]
    tail = tail.replace(a, b, 1)
n = tail.count('tiles_m8, tiles_n, chunks_per_split);')
tail = tail.replace('tiles_m8, tiles_n, chunks_per_split);', 'tiles_m8, tiles_n, chunks_per_split, %s);' % KO)
assert n == 2, n
s = head + tail
src = os.path.join(C, '_conv_s2_ko.hip')
open(src, 'w').write(s)
try:
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function', '-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
    subprocess.check_call(['/opt/rocm/bin/hipcc'] + flags + ['-c', src, '-o', os.path.join(OUT, 'conv_s2_ko.o')], cwd=C)
finally:
    os.remove(src)
objs = [o for o in sorted(os.listdir(C)) if o.endswith('.o') and o != 'conv_h8.o']
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC'] + [os.path.join(C, o) for o in objs] +
                      [os.path.join(OUT, 'conv_s2_ko.o'), '-lz', '-o', os.path.join(OUT, 'libvpship.so')])
print('built', os.path.join(OUT, 'libvpship.so'))
