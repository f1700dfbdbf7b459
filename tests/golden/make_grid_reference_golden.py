#!/usr/bin/env python
"""Golden vectors for the experiment-grid expansion of the drop-in boundary (`experiment.py` builds one experiment per
combination of an econfig's value lists), produced by the REFERENCE'S OWN functions run in the build container.

`/root/reference/src/utilities/utils.py` imports mlflow at its top and cannot be imported here (ModuleNotFoundError), but
`nested_dict_update`, `linearize`, `extract`, `delinearize`, `make_grid` (utils.py:19-99) and `mlflow_linearize` (the keys the run log
receives, utils.py:132-146) are plain Python: this script reads the
file as text, takes exactly these function definitions out of the syntax tree and executes THEM, unchanged, in a namespace that
holds `collections`, `groupby` and `product`.  Inputs (grids shaped like the reference's econfigs, written here) and the functions'
outputs go to tests/golden/grid_reference.json.

    python tests/golden/make_grid_reference_golden.py      (needs /root/reference; CPU only)
"""
import ast
import collections
import collections.abc                              # the reference reaches collections.abc through the bare module
import copy
import json
import os
from itertools import groupby, product

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = '/root/reference/src/utilities/utils.py'
NAMES = ['nested_dict_update', 'linearize', 'extract', 'delinearize', 'make_grid', 'mlflow_linearize']

GRIDS = {
    'basic_gnn_like': {'model': {'name': ['basic.BasicGCN', 'basic.BasicGAT'], 'embedding_dim': [8, 16], 'n_hiddens': [[8, 8], [16, 16]],
                                 'dense_units': [[24, 24]], 'clf_units': [[48, 48]]},
                       'dataset': {'load_function_name': ['load_user_item_graph']}},
    'hybrid_like': {'model': {'name': ['hybrid.HybridBertGCN'], 'dense_units': [[[24, 24], [256, 64], [64, 64]], [[48, 48], [256, 64], [64, 64]]],
                              'clf_units': [[64, 64]], 'feature_based': [True, False]},
                    'dataset': {'type_adjacency': ['unary-uip'], 'train_ratings_filepath': ['a.tsv']}, 'parameters': {'epochs': [25]}},
    'flat': {'seed': [1, 2, 3]},
    'three_levels': {'a': {'b': {'c': [1, 2], 'd': ['x']}, 'e': [0.5]}, 'f': [None, 'g']},
}
UPDATES = [
    ({'model': {'name': 'x', 'l2_regularizer': 1e-4, 'nested': {'p': 1, 'q': 2}}, 'parameters': {'epochs': 25, 'lr': 0.001}},
     {'model': {'name': 'y', 'nested': {'q': 3, 'r': 4}}, 'parameters': {'epochs': 5}, 'new': {'k': [1, 2]}}),
    ({}, {'a': {'b': 1}}),
]


def main():
    ns = {'collections': collections, 'groupby': groupby, 'product': product}
    tree = ast.parse(open(PATH).read(), filename=PATH)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in NAMES:
            exec(compile(ast.Module(body=[node], type_ignores=[]), PATH, 'exec'), ns)
    out = {'grids': {}, 'updates': [], 'log_keys': []}
    for name, grid in GRIDS.items():
        out['grids'][name] = {'input': grid, 'output': ns['make_grid'](copy.deepcopy(grid))}
    for d, u in UPDATES:
        out['updates'].append({'d': d, 'u': u, 'output': ns['nested_dict_update'](copy.deepcopy(d), copy.deepcopy(u))})
    for name, grid in GRIDS.items():
        for exp in out['grids'][name]['output'][:2]:
            out['log_keys'].append({'input': exp, 'output': ns['mlflow_linearize'](copy.deepcopy(exp))})
    json.dump(out, open(os.path.join(HERE, 'grid_reference.json'), 'w'), indent=1, sort_keys=False)
    print('wrote grid_reference.json:', {k: len(v['output']) for k, v in out['grids'].items()})


if __name__ == '__main__':
    main()
