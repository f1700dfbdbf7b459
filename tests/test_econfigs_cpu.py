"""CPU: the reference's own experiment files are consumed unmodified.

Every `econfigs/*.yaml` of the reference tree is expanded with the product's grid logic on top of the reference's
`config.yaml`, and every resulting model section is instantiated (weights built, no device work).  The reference tree only
exists in the build container: elsewhere the test skips (nothing of it is copied into this repository)."""
import glob
import os

import numpy as np
import pytest
from scipy import sparse

REFERENCE = '/root/reference'
ECONFIGS = sorted(glob.glob(os.path.join(REFERENCE, 'econfigs', '*.yaml')))


@pytest.mark.skipif(not ECONFIGS, reason="reference tree not present")
@pytest.mark.parametrize('path', ECONFIGS, ids=[os.path.basename(p) for p in ECONFIGS])
def test_every_reference_experiment_file_expands_and_builds(path):
    import copy
    from deep_cbrs_amar_renaissance_amd import experiment, models as models_pkg
    from deep_cbrs_amar_renaissance_amd.data import loaders
    from deep_cbrs_amar_renaissance_amd.models.basic import BasicGNN
    from deep_cbrs_amar_renaissance_amd.models.hybrid import HybridBertGNN
    from deep_cbrs_amar_renaissance_amd.utilities.utils import make_grid, nested_dict_update
    base = experiment.load_yaml(os.path.join(REFERENCE, 'config.yaml'))
    cfg = experiment.load_yaml(path) or {}
    experiments = dict(cfg.get('linear') or {})
    for grid in (cfg.get('grid') or {}).values():
        experiments.update({str(e): e for e in make_grid(grid)})
    assert experiments, "no experiment found in " + path
    rng = np.random.default_rng(0)
    n_users, n_items = 30, 20
    u, i = rng.integers(0, n_users, 200), rng.integers(0, n_items, 200) + n_users
    n = n_users + n_items
    adj = sparse.coo_matrix((np.ones(400, dtype=np.float32), (np.concatenate([u, i]), np.concatenate([i, u]))), shape=(n, n))
    built = set()
    for overrides in experiments.values():
        config = nested_dict_update(copy.deepcopy(base), overrides) if overrides else copy.deepcopy(base)
        model_cfg = dict(config['model'])
        module_name, class_name = model_cfg['name'].split('.')
        cls = getattr(__import__(models_pkg.__name__ + '.' + module_name, fromlist=[class_name]), class_name)
        assert callable(getattr(loaders, config['dataset']['load_function_name']))
        key = (class_name, str(sorted((k, str(v)) for k, v in model_cfg.items())))
        if key in built:
            continue
        built.add(key)
        if issubclass(cls, (BasicGNN, HybridBertGNN)):
            model = cls(adj, **model_cfg)
            g_dim = model.gnn.output_dim()
            if issubclass(cls, HybridBertGNN):
                model.rs.build_head(g_dim, 768)
            else:
                model.rs.build_head(g_dim, g_dim)
        else:                                                 # BasicRS / HybridCBRS on pre-computed embeddings
            model = cls(**model_cfg)
            model.build_head(768, 768)
        assert sum(p.numel() for p in model.parameters()) > 0
    assert built
