"""The final line of bench.py must survive a log tail of a few KB (round 4's 17.5 KB line lost its head in the driver's record)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _full_record():
    """round 4's real full record (profiles/r04_bench_v3.json, 17.5 KB) re-keyed the way bench_inference now writes it"""
    with open(os.path.join(ROOT, 'profiles', 'r04_bench_v3.json')) as fh:
        full = json.loads([ln for ln in fh.read().splitlines() if ln.startswith('{')][-1])
    inf = full['inference']
    sb = inf['single_batch_pass']
    inf.update(value_steady=inf['value'], images_steady=inf['images'], seconds_per_pass_steady=inf['seconds_per_pass'],
               value=sb['value'], images=sb['images'], seconds_per_pass=sb['seconds_per_pass'], cold_shape_ms=123.4,
               value_unseen_shapes=111.1)
    full['dist'] = {'allreduce_ms': 1.234, 'overlap_frac': 0.5, 'rccl_ranks_seen': 8, 'backend': 'nccl', 'note': 'x' * 500}
    full['fit_path'] = {'value': 900.1, 'unit': 'chips/s', 'batches': 50, 'seconds': 1.1, 'what': 'y' * 900}
    return full


def test_final_line_is_small_and_carries_the_contract():
    import bench
    full = _full_record()
    assert len(json.dumps(full)) > 12000                 # the stub is as fat as the record that broke the driver's parse
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT <= 4096, len(text)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in line, k
    assert line['value'] == full['value'] and line['ms_per_step'] == full['ms_per_step']
    assert set(('workload', 'chips_per_gpu', 'global_batch', 'parallelism')) <= set(line['config'])
    r = line['roofline']
    assert r['bound'] == 'mfma' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and 'traffic' in r
    assert 'by_shape' not in r and 'roofline_hbm' not in r
    c = line['cpu_baseline']
    assert c['kind'] in ('reference', 'port') and c['cores'] >= 1 and c['value'] > 0 and len(c['sample']) <= 160
    i = line['inference']
    assert i['images'] == 8 and i['images_steady'] == 64 and i['value'] == full['inference']['value']
    assert 0 < i['roofline']['frac'] < 1 and i['cpu_baseline']['value'] > 0
    assert line['dist']['rccl_ranks_seen'] == 8 and 'note' not in line['dist']
    assert line['fit_path']['value'] == 900.1 and 'what' not in line['fit_path']


def test_line_sheds_optional_groups_before_the_contract():
    import bench
    full = _full_record()
    full['inference']['roofline']['bound'] = 'm' * 5000          # something absurd in an optional group
    line = bench.compact_line(full)
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    assert 'inference' not in line and 'roofline' in line and 'cpu_baseline' in line and line['value'] == full['value']
