#!/usr/bin/env python
"""Round 4: ONE experiment of the reference's driver end to end at ML-1M size (synthetic ml1m(s=1) files on disk, config.yaml's 25 epochs): where
the wall time of `python src/experiment.py` goes — dataset load, model build, fit, evaluate + predict, top-k files + P/R/F1.
usage: python tools/exp_experiment_e2e.py <basic.BasicGCN|hybrid.HybridBertGCN|...> [epochs]"""
import json
import os
import sys
import tempfile
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import yaml


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'basic.BasicGCN'
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    from deep_cbrs_amar_renaissance_amd import experiment
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.utilities.utils import setup_mlflow
    from tests.test_experiment_gpu import BASE_CONFIG
    tmp = tempfile.mkdtemp(prefix='amar_e2e_')
    t0 = time.perf_counter()
    ds = synthetic.ml1m(1)
    hybrid = name.startswith('hybrid.') and name != 'hybrid.HybridCBRS'
    paths = synthetic.write_dataset(ds, os.path.join(tmp, 'datasets'), bert_dim=768 if name.startswith('hybrid.') else 32, kge_dim=32)
    t_files = time.perf_counter() - t0
    cfg = json.loads(json.dumps(BASE_CONFIG))
    cfg['dataset'].update({k: v for k, v in paths.items() if k != 'props_triples_filepath'})
    cfg['dataset']['load_function_name'] = 'load_user_item_graph_bert_embeddings' if hybrid else 'load_user_item_graph'
    head_only = name in ('basic.BasicRS', 'hybrid.HybridCBRS')          # the reference's baselines on pre-computed embedding rows (no graph)
    if head_only:
        cfg['dataset']['load_function_name'] = 'load_graph_embeddings' if name == 'basic.BasicRS' else 'load_hybrid_embeddings'
    if 'TS' in name or 'TW' in name:                                   # TwoStep / TwoWay stacks: (user-item, item-property[, user-property]) graphs
        cfg['dataset'].update({'type_adjacency': 'unary-kg', 'props_triples_filepath': paths['props_triples_filepath']})
        if 'TW' in name:
            cfg['dataset']['user_properties'] = True
            cfg['model']['user_item_node'] = 'concatenation'
    cfg['parameters']['epochs'] = epochs
    if head_only:
        cfg['model'].update({'name': name, 'dense_units': [64, 32] if name == 'basic.BasicRS' else [[64, 32], [64, 32], [32, 16]], 'clf_units': [16]})
    else:
        cfg['model'].update({'name': name, 'embedding_dim': 8, 'n_hiddens': [8, 8], 'n_layers': 2, 'clf_units': [64, 64] if hybrid else [48, 48],
                             'dense_units': [[24, 24], [256, 64], [64, 64]] if hybrid else [24, 24]})
    open(os.path.join(tmp, 'config.yaml'), 'w').write(yaml.safe_dump(cfg))
    os.chdir(tmp)
    run_log = setup_mlflow('e2e', os.path.join(tmp, 'mlruns'))
    exp = experiment.Experimenter(cfg, run_log)
    stages = {}

    def timed(label, fn):
        import torch
        t = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        stages[label] = time.perf_counter() - t
        return out
    timed('build_dataset', exp.build_dataset)
    timed('build_optimizer + build_model', lambda: (exp.build_optimizer(), exp.build_model()))
    timed('fit ({} epochs)'.format(epochs), lambda: exp.model.fit(exp.trainset, epochs=exp.parameters.epochs, workers=1))
    timed('evaluate (test loss, predict, top-5/10 files, P/R/F1)', exp.evaluate)
    print('%s at ml1m(s=1): synthetic files written in %.1f s (not part of an experiment)' % (name, t_files))
    for k, v in stages.items():
        print('   %-58s %7.2f s' % (k, v))
    print('   %-58s %7.2f s' % ('total', sum(stages.values())))


if __name__ == '__main__':
    main()
