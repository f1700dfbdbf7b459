"""CPU oracle for the GNN-propagation + hybrid-scoring hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, in plain numpy/scipy, the arithmetic of the reference path
(swapUniba/Deep_CBRS_Amar_Renaissance, `src/models`, `src/layers`, `src/data/preprocess.py`,
`src/utilities/{math,metrics}.py`) plus the Spektral 1.x / Keras layer semantics those files
call into.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import it; the shipped package `deep_cbrs_amar_renaissance_amd` never does.

PARITY UNPINNED by reference tests for the layer arithmetic (rows A2-A8): the reference has no tests, fixtures or golden vectors
for this path, its data/weights live in an unreachable DVC remote, and its arithmetic sits in
third-party packages that are not vendored and not installed here:

    spektral   (unpinned, requirements.txt:8; API use implies 1.0.x-1.2)  GCNConv, GraphSageConv,
               GATConv, ops.modal_dot, utils.gcn_filter
    tensorflow/keras (unpinned, requirements.txt:4; API use implies 2.7-2.8)  Dense, Concatenate,
               sparse_dense_matmul, embedding_lookup, glorot_uniform

so those published algorithms are restated here and anchored on (a) the reference's own call
sites, (b) the trainable-parameter counts published in the reference's doc.pdf (SURVEY.md §8c
KAT table), (c) hand-computed tiny graphs, and (d) an independent second implementation
(`oracle/torch_ref.py`, dense torch-CPU) that must agree to 1e-6.

Three rows ARE pinned by the reference itself — its plain numpy / pandas / scipy functions were executed in the build
container and their inputs / outputs committed: `top_k` (models.py) against `top_k_predictions`
(`src/utilities/metrics.py:11-34`; tests/golden/topk_reference.npz), and graph.py's `remap_ratings`, `remap_props`,
`adjacency_unary*`, `user_properties` against `load_train_test_ratings`, `build_adjacency_matrix`,
`get_user_properties`, `symmetrize_matrix` (`src/data/loaders.py:11-82`, `src/data/preprocess.py:9-170`,
`src/utilities/math.py:6-21`; tests/golden/graph_reference.npz) — bit for bit, triplet order included.  The scripts
that made the fixtures are tests/golden/make_*_reference_golden.py.
"""
