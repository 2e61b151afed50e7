"""CPU-side checks of the drop-in boundary: libvpship.so loads and exports every symbol include/vps_hip.h
declares; the ctypes structs match the C layout; the product fails loudly without a device."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

from vps_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'vps_hip.h')


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(vps_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    lib = hip.load()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for s in declared:
        assert hasattr(lib, s), 'libvpship.so does not export %s' % s
    assert sorted(hip.SYMBOLS) == declared
    # every entry point has its ctypes signature (a call through ctypes' default conversions truncates 64-bit arguments)
    for s in hip.SYMBOLS:
        f = getattr(lib, s)
        assert f.argtypes is not None or s in ('vps_abi_version', 'vps_build_info'), '%s has no argtypes in vps_amd/hip.py' % s


def test_abi_version_and_info():
    lib = hip.load()
    assert lib.vps_abi_version() == hip.ABI_VERSION
    assert 'gfx950' in hip.build_info()


def test_struct_layout_matches_c(tmp_path):
    """compile a tiny C program against the header and compare sizeof / offsetof of EVERY field with the ctypes mirrors"""
    cname = {'inp': 'in'}                                    # `in` is a Python keyword: the ctypes mirror calls it `inp`
    structs = (('vps_conv_desc', hip.ConvDesc), ('vps_tensor4', hip.Tensor4), ('vps_pan_inst', hip.PanInst))
    lines = []
    for cn, ct in structs:
        lines.append('printf("%%zu\\n", sizeof(%s));' % cn)
        lines += ['printf("%%zu\\n", offsetof(%s, %s));' % (cn, cname.get(f[0], f[0])) for f in ct._fields_]
    src = tmp_path / 'layout.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "vps_hip.h"\nint main(void) {\n%s\nreturn 0; }\n' % '\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    out = [int(v) for v in subprocess.check_output([str(exe)]).decode().split()]
    want = []
    for cn, ct in structs:
        want.append(ctypes.sizeof(ct))
        want += [getattr(ct, f[0]).offset for f in ct._fields_]
    assert out == want
    assert {'gn_stats', 'gn_cpg', 'gn_rep', 'status', 'w_split'} <= {f[0] for f in hip.ConvDesc._fields_}


def test_cpu_tensor_is_rejected_loudly():
    with pytest.raises(hip.VpsHipError):
        hip.ptr(torch.zeros(4))


def test_bad_arguments_return_error_codes_without_a_gpu():
    lib = hip.load()
    d = hip.ConvDesc()
    assert lib.vps_conv2d(ctypes.byref(d), None) <= -1000      # null pointers are rejected before any launch
    assert lib.vps_conv2d(None, None) <= -1000


def test_library_isa_is_free_of_the_packed_fp32_form_that_fails_beside_mfma_kernels():
    """tools/check_isa.py on the built library: no v_pk_{mul,add,fma}_f32 with op_sel = 1 on its second source (wrong results
    beside MFMA kernels on gfx950: tools/pkhazard); the Makefile builds without packed FP32 at all"""
    import importlib.util
    spec = importlib.util.spec_from_file_location('check_isa', os.path.join(ROOT, 'tools', 'check_isa.py'))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    if not os.path.exists(mod.OBJDUMP):
        pytest.skip('llvm-objdump not available')
    nobj, npk, bad = mod.scan(hip.LIB_PATH)
    assert nobj >= 1 and not bad, bad[:5]
    assert npk == 0, 'packed FP32 was re-enabled: keep the lint above green and re-validate the stream-invariance tests'


def test_kernel_resource_budget():
    """code-object metadata of every kernel in the built library (tools/check_isa.py --resources): no VGPR spills anywhere, no
    scratch memory in the MFMA conv kernels (a spilled accumulator would silently cost more than any tuning gains), LDS within the
    160 KB of a CU, the 8-wave kernels within one block per CU's register file (<= 256 VGPRs at 2 waves per SIMD)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location('check_isa', os.path.join(ROOT, 'tools', 'check_isa.py'))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    if not (os.path.exists(mod.OBJDUMP) and os.path.exists(mod.READELF)):
        pytest.skip('llvm-objdump / llvm-readelf not available')
    res = mod.resources(hip.LIB_PATH)
    assert len(res) > 100 and any('conv_mfma_h8_kernel' in n for n in res), len(res)
    for name, r in res.items():
        assert r['vgpr_spill'] == 0, (name, r)
        # (the correlation kernel's two accumulator tiles live in AGPRs - the compiler's choice for accumulators no VALU instruction
        # touches inside the loop; 164 + 32 registers, one wave per SIMD)
        # (the 256-column deformable instance is built for ONE block per CU: 2 x 4 accumulator tiles = 128 registers per lane in AGPRs,
        # 512 registers per lane in all)
        if 'conv_mfma_bf16p_kernel<2, 4, 2, 2' in name:
            assert r['lds_bytes'] <= 160 * 1024 and r['vgpr'] <= 512 and r['scratch_bytes'] == 0, (name, r)
            continue
        assert r['lds_bytes'] <= 160 * 1024 and r['vgpr'] <= 256 and (r['agpr'] == 0 or ('corr_mfma' in name and r['vgpr'] + r['agpr'] <= 256)), (name, r)
        if 'conv_mfma' in name:
            assert r['scratch_bytes'] == 0, (name, r)
