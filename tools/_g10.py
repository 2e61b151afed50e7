import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from concurrent.futures import ThreadPoolExecutor
from vps_amd import synth
from vps_amd.pipeline import imread
tmp = tempfile.mkdtemp()
files = []
for i in range(8):
    fr = synth.synth_frame(1024, 2048, seed=i % 4, shift=(2 * i, i), noise=2.0).astype(np.uint8)
    fn = os.path.join(tmp, 'f%d.png' % i); Image.fromarray(np.ascontiguousarray(fr[:, :, ::-1])).save(fn, compress_level=6); files.append(fn)
for native in (True, False):
    t0 = time.perf_counter(); [imread(f, native=native) for f in files]; print('native' if native else 'PIL', 'single thread %.1f ms/frame' % (1e3 * (time.perf_counter() - t0) / 8))
    for nw in (2, 4, 6, 12):
        with ThreadPoolExecutor(nw) as ex:
            t0 = time.perf_counter(); list(ex.map(lambda f: imread(f, native=native), files * 4)); dt = time.perf_counter() - t0
        print('   %d threads: %.1f frames/s' % (nw, 32 / dt))
