"""ISA lint of the built library: no packed-FP32 instruction may take the HIGH half of its second source for the LOW result
(`v_pk_{mul,add,fma}_f32 ... op_sel:[x,1,...]`). On gfx950 (ROCm 7.2) that operand form returns wrong values in ~2 % of its
executions while MFMA kernels share the compute units (tools/pkhazard/patterns.hip, profiles/r02_pkhazard_patterns.txt,
DESIGN.md 3.3). The Makefile builds without packed FP32 altogether; this check keeps it that way — or makes a build that
re-enables packed FP32 prove that it is free of the failing form.

    python tools/check_isa.py [path/to/libvpship.so]      # exit code 1 + the offending lines when the form is present
    python tools/check_isa.py --resources [lib]            # per-kernel registers / spills / scratch / LDS from the code-object notes
    python tools/check_isa.py --mix [lib]                  # static instruction mix of the MFMA kernels (whole kernel, per MFMA)
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = os.environ.get('LLVM_OBJDUMP', '/opt/rocm/lib/llvm/bin/llvm-objdump')
READELF = os.environ.get('LLVM_READELF', '/opt/rocm/lib/llvm/bin/llvm-readelf')
CXXFILT = os.environ.get('CXXFILT') or shutil.which('c++filt') or '/opt/rocm/lib/llvm/bin/llvm-cxxfilt'
# measured on the three FP32 forms; every packed (VOP3P) instruction with that operand selection is refused, to be on the safe side
BAD = re.compile(r'\bv_pk_\w+\b.*\bop_sel:\[[01],1')
PK = re.compile(r'v_pk_[a-z]+_f32\b')


def scan(lib):
    """-> (number of gfx950 code objects, packed-FP32 instructions, offending lines)"""
    tmp = tempfile.mkdtemp(prefix='vps_isa_')
    try:
        so = os.path.join(tmp, 'lib.so')
        shutil.copy(lib, so)
        subprocess.run([OBJDUMP, '--offloading', so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        objs = sorted(glob.glob(so + '.*gfx950*'))
        npk, bad = 0, []
        for o in objs:
            txt = subprocess.run([OBJDUMP, '-d', o], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode(errors='replace')
            for line in txt.splitlines():
                if PK.search(line):
                    npk += 1
                if BAD.search(line):
                    bad.append(line.strip())
        return len(objs), npk, bad
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def resources(lib):
    """-> {demangled kernel name: dict(vgpr, agpr, sgpr_spill, vgpr_spill, scratch_bytes, lds_bytes)} from the AMDGPU metadata notes"""
    tmp = tempfile.mkdtemp(prefix='vps_isa_')
    try:
        so = os.path.join(tmp, 'lib.so')
        shutil.copy(lib, so)
        subprocess.run([OBJDUMP, '--offloading', so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = {}
        for o in sorted(glob.glob(so + '.*gfx950*')):
            txt = subprocess.run([READELF, '--notes', o], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode(errors='replace')
            cur = {}
            for line in txt.splitlines():
                m = re.match(r'\s*-?\s*\.(\w+):\s*(.*)$', line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip()
                if k == 'agpr_count' or (k == 'group_segment_fixed_size' and 'name' in cur):
                    # a new kernel record starts (keys are emitted alphabetically: .agpr_count first, .args next ...)
                    pass
                if k == 'name' and not v.startswith("'") and ('kernel' in v or v.startswith('_Z')):
                    cur['name'] = v
                elif k in ('vgpr_count', 'agpr_count', 'sgpr_spill_count', 'vgpr_spill_count', 'private_segment_fixed_size', 'group_segment_fixed_size'):
                    cur[k] = int(v)
                if k == 'wavefront_size':                       # last key of a kernel record
                    if 'name' in cur:
                        out[cur['name']] = cur
                    cur = {}
        names = list(out)
        if names and os.path.exists(CXXFILT):
            dem = subprocess.run([CXXFILT], input='\n'.join(names).encode(), stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
            out = {d: out[n] for d, n in zip(dem, names)}
        return {n: dict(vgpr=r.get('vgpr_count', 0), agpr=r.get('agpr_count', 0), sgpr_spill=r.get('sgpr_spill_count', 0),
                        vgpr_spill=r.get('vgpr_spill_count', 0), scratch_bytes=r.get('private_segment_fixed_size', 0),
                        lds_bytes=r.get('group_segment_fixed_size', 0)) for n, r in out.items()}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _classify(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_read') or op.startswith('ds_load'): return 'lds_read'
    if op.startswith('ds_write') or op.startswith('ds_store'): return 'lds_write'
    if op.startswith('global_load') or op.startswith('buffer_load'): return 'gload'
    if op.startswith('global_store') or op.startswith('buffer_store'): return 'gstore'
    if op == 's_waitcnt': return 'waitcnt'
    if op == 's_barrier': return 'barrier'
    if op == 's_nop': return 'nop'
    return 'other'


def instruction_mix(lib, match='conv_mfma'):
    """-> {kernel: counts} of the kernel's MAIN LOOP: the backward-branch region that holds the most MFMAs (the k loops are unrolled
    over the taps / both k slabs, so this is one whole chunk or k-step pair). Keys: mfma, valu (other v_*), lds_read, lds_write,
    gload, gstore, waitcnt, barrier, nop, other."""
    tmp = tempfile.mkdtemp(prefix='vps_isa_')
    try:
        so = os.path.join(tmp, 'lib.so')
        shutil.copy(lib, so)
        subprocess.run([OBJDUMP, '--offloading', so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        kernels = {}
        for o in sorted(glob.glob(so + '.*gfx950*')):
            txt = subprocess.run([OBJDUMP, '-d', o], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode(errors='replace')
            cur = None
            for line in txt.splitlines():
                m = re.match(r'^([0-9a-f]+) <(.*)>:$', line)
                if m:
                    cur = kernels.setdefault(m.group(2), dict(start=int(m.group(1), 16), ins=[])) if match in m.group(2) else None
                    continue
                if cur is None:
                    continue
                m = re.match(r'^\s+(\S+).*//\s*([0-9A-F]+):', line)
                if not m:
                    continue
                op, addr = m.group(1), int(m.group(2), 16)
                tgt = None
                if op.startswith('s_cbranch') or op == 's_branch':
                    mt = re.search(r'<[^>]*\+0x([0-9a-f]+)>\s*$', line)
                    if mt:
                        tgt = cur['start'] + int(mt.group(1), 16)
                cur['ins'].append((addr, op, tgt))
        out = {}
        names = list(kernels)
        dem = names
        if names and os.path.exists(CXXFILT):
            dem = subprocess.run([CXXFILT], input='\n'.join(names).encode(), stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
        for n, d in zip(names, dem):
            ins = kernels[n]['ins']
            best = None
            for addr, op, tgt in ins:
                if tgt is not None and tgt <= addr:
                    region = [c for a, c, _ in ins if tgt <= a <= addr]
                    nm = sum(1 for c in region if c.startswith('v_mfma'))
                    if best is None or nm > best[0]:
                        best = (nm, region)
            if best is None or best[0] == 0:
                continue
            c = dict(mfma=0, valu=0, lds_read=0, lds_write=0, gload=0, gstore=0, waitcnt=0, barrier=0, nop=0, other=0)
            for op in best[1]:
                c[_classify(op)] += 1
            out[re.sub(r'\(anonymous namespace\)::', '', d).split('(')[0]] = c
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--mix':
        lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, 'vps_amd', 'csrc', 'libvpship.so')
        mix = instruction_mix(lib)
        print('static instruction mix of the MAIN LOOP of the MFMA conv kernels (the backward-branch region with the most MFMAs), per MFMA.')
        print('Issue budget beside one 32-cycle 32x32x16 MFMA (tools/gapbench.hip): <= 5 VALU, <= 2 LDS reads; a global load = 64 cycles of the')
        print('vector-memory pipe. The f32 kernels issue 32x32x2 MFMAs (64 cycles each).')
        print('%-58s %6s %7s %7s %7s %7s %7s %7s' % ('kernel', 'MFMA', 'VALU/M', 'LDSr/M', 'LDSw/M', 'gld/M', 'wait/M', 'barrier'))
        for n, c in sorted(mix.items()):
            if not c['mfma']:
                continue
            f = float(c['mfma'])
            print('%-58s %6d %7.2f %7.2f %7.2f %7.3f %7.2f %7d' % (n[:58], c['mfma'], c['valu'] / f, c['lds_read'] / f, c['lds_write'] / f, c['gload'] / f,
                                                                  c['waitcnt'] / f, c['barrier']))
        return 0
    if len(sys.argv) > 1 and sys.argv[1] == '--resources':
        lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, 'vps_amd', 'csrc', 'libvpship.so')
        res = resources(lib)
        print('%-110s %5s %6s %6s %8s %8s' % ('kernel', 'VGPR', 'vspill', 'sspill', 'scratch', 'LDS'))
        for n, r in sorted(res.items()):
            short = re.sub(r'\(anonymous namespace\)::', '', n).split('(')[0]
            print('%-110s %5d %6d %6d %8d %8d' % (short[:110], r['vgpr'], r['vgpr_spill'], r['sgpr_spill'], r['scratch_bytes'], r['lds_bytes']))
        return 0
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'vps_amd', 'csrc', 'libvpship.so')
    nobj, npk, bad = scan(lib)
    print('%s: %d gfx950 code objects, %d packed-FP32 instructions, %d with op_sel = 1 on src1' % (lib, nobj, npk, len(bad)))
    for b in bad[:20]:
        print('  ', b)
    return 1 if bad or not nobj else 0


if __name__ == '__main__':
    sys.exit(main())
