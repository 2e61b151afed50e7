"""Aggregate the HBM-side traffic of the conv kernels from two rocprofv3 --pmc passes over bench.py (GPU box).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python $R/tools/pmc_traffic.py $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write 3 > $R/gpurun_out/pmc_traffic.json

(separate passes, no other trace domain: MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots"). FETCH_SIZE / WRITE_SIZE are
in KiB; on gfx950 FETCH_SIZE counts 128-byte requests of wide (16 B/lane) coalesced reads as 64 bytes, so it is doubled
(all loads of the conv kernels are global_load_dwordx4 / buffer_load_dwordx4); WRITE_SIZE is taken as reported (uncalibrated, see the guide).
The third argument is the number of frames the profiled command ran: warmup + steps + the 2 set-up frames and the 3 instrumented
frames of bench.py's single-stream section (7 for `--steps 1 --warmup 1`; 5 up to round 5's committed measurement, which ran one
instrumented frame)."""
import csv
import re
import glob
import json
import os
import sys


# every kernel of the vps_conv2d family, whatever translation unit it lives in. Round 6 found the list of names that stood here missing the
# kernels added since (conv_mfma_h8p_kernel, conv_pw_kernel, the n16t / n32 instances: a substring test on 'conv_mfma_h8_kernel' does not see
# 'conv_mfma_h8p_kernel') - the first r06 figure (23.8 GB) left their traffic out. A pattern cannot fall behind the sources.
CONV = re.compile(r'conv_(mfma_\w+|thin|small\w*|pw|splitk_reduce)_kernel')


def totals(directory, counter):
    """{short kernel name: [sum of the counter, dispatches]} over the conv family"""
    acc = {}
    for fn in glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(fn)):
            if row['Counter_Name'] != counter:
                continue
            m = CONV.search(row['Kernel_Name'])
            if m:
                a = acc.setdefault(m.group(0), [0.0, 0])
                a[0] += float(row['Counter_Value']); a[1] += 1
    return acc


def main():
    fetch_dir, write_dir, frames = sys.argv[1], sys.argv[2], int(sys.argv[3])
    out = {'frames': frames, 'kernels': {}}
    fetch, write = totals(fetch_dir, 'FETCH_SIZE'), totals(write_dir, 'WRITE_SIZE')
    for name in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(name, (0.0, 0))
        w, nw = write.get(name, (0.0, 0))
        out['kernels'][name] = {'launches_per_frame': nf / frames, 'fetch_bytes_per_frame': 2.0 * f * 1024 / frames,
                                'write_bytes_per_frame': w * 1024 / frames,
                                'raw': {'FETCH_SIZE_KiB_sum': f, 'fetch_dispatches': nf, 'WRITE_SIZE_KiB_sum': w, 'write_dispatches': nw}}
    out['conv_hbm_bytes_per_frame'] = sum(k['fetch_bytes_per_frame'] + k['write_bytes_per_frame'] for k in out['kernels'].values())
    # which kernel sources this was measured on: bench.py attaches the newest committed measurement to its line and says whether the
    # library it runs was built from the same sources (VERDICT r3 hygiene)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from vps_amd.hip import csrc_sha16
    out['csrc_sha16'] = csrc_sha16()
    out['note'] = 'FETCH_SIZE x2 (gfx950 wide-read correction), WRITE_SIZE as reported; Infinity-Cache hits are counted'
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
