"""VPQ tube statistics. CPU: the oracle restatement against golden statistics of the REAL reference
function (tests/golden/make_vpq_golden.py). GPU: the device-counted evaluator against the golden statistics, exactly."""
import json
import os

import numpy as np
import pytest

from oracle import evaluate as oev

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'vpq_cases.npz')
CATS = {c: {'id': c, 'isthing': 1 if c >= 11 else 0} for c in range(19)}


def _clips():
    z = np.load(GOLD)
    for ci in range(int(z['nclips'])):
        js = json.loads(bytes(z['clip%d_json' % ci]).decode())
        gt, pred = z['clip%d_gt' % ci], z['clip%d_pred' % ci]
        frames = [(js[f][0], js[f][1], gt[f], pred[f], {}) for f in range(len(js))]
        for nf in (1, 2, 3):
            yield ci, nf, frames, z['clip%d_nf%d_counts' % (ci, nf)], z['clip%d_nf%d_iou' % (ci, nf)]


def _check(stat, counts, iou, what):
    for row, v in zip(counts, iou):
        c = int(row[0])
        assert [stat[c].tp, stat[c].fp, stat[c].fn] == [int(row[1]), int(row[2]), int(row[3])], (what, c)
        assert stat[c].iou == float(v), (what, c, stat[c].iou, float(v))          # same sums in the same order: bitwise


def test_oracle_matches_reference_function():
    n = 0
    for ci, nf, frames, counts, iou in _clips():
        _check(oev.vpq_compute_single_core(frames, CATS, nframes=nf), counts, iou, (ci, nf)); n += 1
    assert n == 12 and int(np.load(GOLD)['clip1_nf2_counts'][:, 1].sum()) > 0        # the golden clips do contain matches


def _ev():
    from vps_amd import evaluate as ev
    return ev


# the per-frame device count (vps_pair_count) restated in NumPy: CPU stand-in for the tests of the HOST logic around it
def _numpy_count(self, gt_json, pred_json, gt_pan, pred_pan, categories, extra_gt_ids=None):
    ids = lambda q: (lambda u: u[:, :, 0] + u[:, :, 1] * 256 + u[:, :, 2] * 65536)(np.asarray(q).astype(np.int64))
    g, p = ids(gt_pan), ids(pred_pan)
    gt_segms, pred_segms = _ev()._merged(gt_json), _ev()._merged(pred_json)
    labels, cnt = np.unique(p, return_counts=True)
    pred_set = set(el['id'] for el in pred_json['segments_info'])
    for label, c in zip(labels, cnt):
        label = int(label)
        if label not in pred_segms:
            assert label == 0
            continue
        pred_segms[label]['area'] = int(c); pred_set.remove(label)
    assert not pred_set
    listed_g = set([0] + [el['id'] for el in gt_json['segments_info']] + list(extra_gt_ids or ()))
    lab, c2 = np.unique(g * (1 << 24) + p, return_counts=True)
    pairs = {(int(l >> 24), int(l & ((1 << 24) - 1))): int(c) for l, c in zip(lab, c2) if int(l >> 24) in listed_g}
    return gt_segms, pred_segms, pairs


def test_host_matching_logic_with_numpy_counts(monkeypatch):
    """CPU coverage of vps_amd/evaluate.py's host side (segment bookkeeping, window sums, matching): the per-frame device count
    is replaced by its NumPy definition and the result must still equal the reference function's statistics"""
    from vps_amd import evaluate as ev
    count = _numpy_count
    monkeypatch.setattr(ev.FrameCounts, 'count', count)
    monkeypatch.setattr(ev.FrameCounts, '__init__', lambda self, device='cuda': None)
    for ci, nf, frames, counts, iou in _clips():
        _check(ev.vpq_compute_single_core(frames, CATS, nframes=nf, device='cpu'), counts, iou, (ci, nf))


@pytest.mark.gpu
def test_device_counted_vpq_matches_reference_function(dev):
    from vps_amd import evaluate as ev
    for ci, nf, frames, counts, iou in _clips():
        _check(ev.vpq_compute_single_core(frames, CATS, nframes=nf, device=dev), counts, iou, (ci, nf))


@pytest.mark.gpu
def test_device_counts_full_size_and_error_paths(dev):
    import torch
    from vps_amd import evaluate as ev
    rng = np.random.default_rng(0)
    H, W = 1024, 2048
    ids = np.array([0, 7, 300, 70000, 1 << 20, (1 << 24) - 1], dtype=np.int64)
    gt = ids[rng.integers(0, len(ids), size=(H // 32, W // 32))].repeat(32, 0).repeat(32, 1)
    pr = ids[rng.integers(1, len(ids), size=(H // 16, W // 16))].repeat(16, 0).repeat(16, 1)
    rgb = lambda m: np.stack([m % 256, (m // 256) % 256, m // 65536], -1).astype(np.uint8)
    info = lambda m: {'segments_info': [{'id': int(i), 'category_id': int(i % 19), 'iscrowd': 0, 'area': int((m == i).sum())} for i in np.unique(m) if i]}
    g, p, pairs = ev.FrameCounts(dev).count(info(gt), info(pr), torch.from_numpy(rgb(gt)).to(dev), rgb(pr), CATS)
    key = gt.astype(np.uint64) * (1 << 24) + pr.astype(np.uint64)
    lab, cnt = np.unique(key, return_counts=True)
    assert pairs == {(int(l >> 24), int(l & ((1 << 24) - 1))): int(c) for l, c in zip(lab, cnt)}
    assert all(p[int(i)]['area'] == int((pr == i).sum()) for i in np.unique(pr))
    bad = info(pr); bad['segments_info'] = bad['segments_info'][1:]               # a PNG id that the JSON does not list
    with pytest.raises(KeyError):
        ev.FrameCounts(dev).count(info(gt), bad, rgb(gt), rgb(pr), CATS)
    extra = info(pr); extra['segments_info'].append({'id': 12345, 'category_id': 3, 'iscrowd': 0, 'area': 1})
    with pytest.raises(KeyError):
        ev.FrameCounts(dev).count(info(gt), extra, rgb(gt), rgb(pr), CATS)


def _write_dataset(tmp_path, clips):
    """the files tools/eval_vpq.py reads: <truth>/<name>_final_mask.png, <submit>/pan_pred/<id>.png, <submit>/pred.json, gt json"""
    from PIL import Image
    z = np.load(GOLD)
    truth, submit = tmp_path / 'truth', tmp_path / 'submit'
    (submit / 'pan_pred').mkdir(parents=True); truth.mkdir()
    images, gt_ann, pred_ann, sets = [], [], [], []
    for v, ci in enumerate(clips):
        js = json.loads(bytes(z['clip%d_json' % ci]).decode())
        gt, pred = z['clip%d_gt' % ci], z['clip%d_pred' % ci]
        sets.append([(js[f][0], js[f][1], gt[f], pred[f], {}) for f in range(len(js))])
        for f in range(len(js)):
            iid = '%04d_%04d_city_%06d' % (v, f, f)
            images.append({'id': iid, 'file_name': iid + '_newImg8bit.png'})
            Image.fromarray(gt[f]).save(str(truth / (iid + '_final_mask.png')))
            Image.fromarray(pred[f]).save(str(submit / 'pan_pred' / (iid + '.png')))
            gt_ann.append(js[f][0]); pred_ann.append(js[f][1])
    cats = [{'id': c, 'isthing': 1 if c >= 11 else 0, 'name': 'c%d' % c} for c in range(19)]
    gj = tmp_path / 'gt.json'
    gj.write_text(json.dumps({'images': images, 'annotations': gt_ann, 'categories': cats}))
    (submit / 'pred.json').write_text(json.dumps({'annotations': pred_ann}))
    return str(submit), str(truth), str(gj), sets


def _expected_vpq(sets):
    """the reference's averaging (eval_vpq.py:212-232, 303-330) over the oracle's per-clip statistics"""
    from vps_amd.evaluate import PQStat
    out = {'All': [], 'Things': [], 'Stuff': []}
    for nf in (1, 2, 3, 4):
        stat = PQStat()
        for frames in sets:
            o = oev.vpq_compute_single_core(frames, CATS, nframes=nf)
            for c, cat in o.pq_per_cat.items():
                stat[c].tp += cat.tp; stat[c].fp += cat.fp; stat[c].fn += cat.fn; stat[c].iou += cat.iou
        for name, isthing in (('All', None), ('Things', True), ('Stuff', False)):
            out[name].append(100 * stat.pq_average(CATS, isthing=isthing)[0]['pq'])
    return {k: sum(v) / 4 for k, v in out.items()}


def _load_cli():
    import importlib.util
    spec = importlib.util.spec_from_file_location('eval_vpq_device', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'eval_vpq_device.py'))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod


def test_eval_vpq_command_line_files_and_numbers(tmp_path, monkeypatch):
    """tools/eval_vpq_device.py = tools/eval_vpq.py:main on this package: reads the PNG / json files of the reference's layout,
    writes vpq-0/5/10/15.txt and vpq-final.txt; numbers equal the oracle's statistics pushed through the reference's averaging
    (the device count replaced by its NumPy definition on the CPU)"""
    ev = _ev()
    monkeypatch.setattr(ev.FrameCounts, 'count', _numpy_count)
    monkeypatch.setattr(ev.FrameCounts, '__init__', lambda self, device='cuda': None)
    submit, truth, gj, sets = _write_dataset(tmp_path, (0, 3))
    got = _load_cli().main(['--submit_dir', submit, '--truth_dir', truth, '--pan_gt_json_file', gj, '--nframes_per_video', '5', '--device', 'cpu'])
    exp = _expected_vpq(sets)
    for k in exp:
        assert abs(got[k] - exp[k]) < 1e-9, (k, got[k], exp[k])
    assert exp['All'] > 0
    final = open(os.path.join(submit, 'vpq-final.txt')).read().split()
    assert final == ['vpq_all:%.4f' % exp['All'], 'vpq_thing:%.4f' % exp['Things'], 'vpq_stuff:%.4f' % exp['Stuff']]
    for k in (0, 5, 10, 15):
        lines = open(os.path.join(submit, 'vpq-%d.txt' % k)).read().splitlines()
        assert lines[0] == '=' * 48 and lines[3].startswith('All       |') and lines[6].startswith('IDX |') and len(lines) == 7 + 19


@pytest.mark.gpu
def test_eval_vpq_command_line_on_the_device(dev, tmp_path):
    submit, truth, gj, sets = _write_dataset(tmp_path, (0, 3))
    got = _load_cli().main(['--submit_dir', submit, '--truth_dir', truth, '--pan_gt_json_file', gj, '--nframes_per_video', '5', '--device', str(dev)])
    exp = _expected_vpq(sets)
    for k in exp:
        assert abs(got[k] - exp[k]) < 1e-9, (k, got[k], exp[k])


def test_pq_average_viper_variant_excludes_mobilebarrier():
    """tools/dataset/viper.py:64-90 (PQStat.pq_average with `if label == 11: continue`) against vps_amd's `exclude=`: the REAL
    class, imported under the golden generator's shims when /root/reference is there, else the restated expectation"""
    import os
    from vps_amd import evaluate as ev
    from vps_amd.postprocess import DATASETS
    cats = {c: {'isthing': 1 if c >= 13 else 0} for c in range(23)}
    stat = ev.PQStat()
    rng = np.random.default_rng(3)
    vals = {}
    for c in (2, 5, 11, 13, 17, 22):
        tp, fp, fn = int(rng.integers(1, 9)), int(rng.integers(0, 5)), int(rng.integers(0, 5))
        iou = float(rng.uniform(0.5, 1.0)) * tp
        stat[c].tp, stat[c].fp, stat[c].fn, stat[c].iou = tp, fp, fn, iou
        vals[c] = (tp, fp, fn, iou)
    got, per = stat.pq_average(cats, None, exclude=DATASETS['viper']['pq_exclude'])
    assert 11 not in per and got['n'] == 5
    want = np.mean([vals[c][3] / (vals[c][0] + 0.5 * vals[c][1] + 0.5 * vals[c][2]) for c in (2, 5, 13, 17, 22)])
    assert abs(got['pq'] - want) < 1e-12
    full, _ = stat.pq_average(cats, None)
    assert full['n'] == 6
    if os.path.isdir('/root/reference/tools/dataset'):
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
        import make_unify_golden as mg
        before = set(sys.modules)
        try:
            mg.load_reference_function('tools.dataset.viper', 23, 11)
            ref_mod = sys.modules['tools.dataset.viper']
            rs = ref_mod.PQStat()
            for c, (tp, fp, fn, iou) in vals.items():
                rs[c].tp, rs[c].fp, rs[c].fn, rs[c].iou = tp, fp, fn, iou
            ref, ref_per = rs.pq_average(cats, None)
        finally:
            # the import shims (inert stand-ins for cv2, pycocotools, ... and the reference's `tools` package) must not outlive this
            # test: `pytest.importorskip('cv2')` of the third-party pins and `pipeline.imread` would take the stand-in for the library
            for k in set(sys.modules) - before:
                del sys.modules[k]
        assert ref['n'] == got['n'] and ref['pq'] == got['pq'] and ref['sq'] == got['sq'] and ref['rq'] == got['rq']
        assert set(ref_per) == set(per)
