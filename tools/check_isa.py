"""ISA lint of the built library: no packed-FP32 instruction may take the HIGH half of its second source for the LOW result
(`v_pk_{mul,add,fma}_f32 ... op_sel:[x,1,...]`). On gfx950 (ROCm 7.2) that operand form returns wrong values in ~2 % of its
executions while MFMA kernels share the compute units (tools/pkhazard/patterns.hip, profiles/r02_pkhazard_patterns.txt,
DESIGN.md 3.3). The Makefile builds without packed FP32 altogether; this check keeps it that way — or makes a build that
re-enables packed FP32 prove that it is free of the failing form.

    python tools/check_isa.py [path/to/libvpship.so]      # exit code 1 + the offending lines when the form is present
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
BAD = re.compile(r'v_pk_(mul|add|fma)_f32\b.*\bop_sel:\[[01],1')
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


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'vps_amd', 'csrc', 'libvpship.so')
    nobj, npk, bad = scan(lib)
    print('%s: %d gfx950 code objects, %d packed-FP32 instructions, %d with op_sel = 1 on src1' % (lib, nobj, npk, len(bad)))
    for b in bad[:20]:
        print('  ', b)
    return 1 if bad or not nobj else 0


if __name__ == '__main__':
    sys.exit(main())
