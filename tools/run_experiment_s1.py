#!/usr/bin/env python
"""End-to-end drop-in run at ML-1M size (development aid): writes the synthetic ml1m(1) dataset in the reference's file
formats, then runs `src/experiment.py` on a basic-gnn grid1 experiment file (3 epochs) and reports the wall time."""
import json
import os
import subprocess
import sys
import tempfile
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yaml


def main():
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from tests.test_experiment_gpu import BASE_CONFIG
    tmp = tempfile.mkdtemp(prefix='amar_s1_')
    ds = synthetic.ml1m(1)
    paths = synthetic.write_dataset(ds, os.path.join(tmp, 'datasets', 'movielens'))
    cfg = json.loads(json.dumps(BASE_CONFIG))
    cfg['parameters']['epochs'] = 3
    cfg['dataset'].update({k: v for k, v in paths.items() if k != 'props_triples_filepath'})
    open(os.path.join(tmp, 'config.yaml'), 'w').write(yaml.safe_dump(cfg))
    open(os.path.join(tmp, 'exps.yaml'), 'w').write(
        "grid:\n  grid1:\n    model:\n      name: [basic.BasicGCN, basic.BasicGraphSage, basic.BasicGAT, basic.BasicLightGCN]\n"
        "      l2_regularizer: [1e-4]\n      dense_units: [[24, 24]]\n      clf_units: [[48, 48]]\n      embedding_dim: [8]\n"
        "      n_hiddens: [[8, 8]]\n      n_layers: [2]\n    dataset:\n      load_function_name: [load_user_item_graph]\n")
    t0 = time.perf_counter()
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'src', 'experiment.py'), '-c', 'config.yaml', '-e', 'exps.yaml',
                           '--exp_name', 's1'], cwd=tmp, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    lines = [l for l in proc.stdout.splitlines() if 'Epoch' in l or 'precision' in l or 'f1' in l or 'Experiment ' in l]
    print('\n'.join(lines[-24:]))
    print('return code', proc.returncode, '- 4 experiments (load, 3 epochs, evaluate, top-5/10, P/R/F1) in %.1f s' % dt)
    if proc.returncode:
        print(proc.stderr[-2000:])


if __name__ == '__main__':
    main()
