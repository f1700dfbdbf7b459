"""Experiment driver — mirrors `/root/reference/src/experiment.py:29-318`.

Same CLI (``-c/--config``, ``-e/--experiments``, ``--exp_name``), same YAML inputs
(``config.yaml`` + ``experiments.yaml`` / ``econfigs/*.yaml`` with ``grid:`` and ``linear:``
sections), same resolution of model / loader classes from name strings, same per-experiment
catch-and-continue.  Differences, all forced by scope:

* tracking goes to a JSON-lines run log instead of MLflow (not installed);
* ``Model.fit`` is the HIP training step of training.py (every model class named in ``econfigs/`` and the TwoStep / TwoWay
  classes the factories generate); a model or reduction without a training recipe raises and the experiment fails — the
  grid goes on with the next one, nothing is evaluated on untrained weights;
* Precision/Recall/F1@k come from a host-side restatement of RiVal's holdout metrics instead of ``binaries/mimir.jar``
  (utilities/metrics.py: hand-computed vectors in tests/test_metrics_cpu.py).

Run from the directory that holds ``config.yaml`` and the ``datasets/`` tree:
``python -m deep_cbrs_amar_renaissance_amd.experiment -e econfigs/basic-gnn.yaml``
(or ``python src/experiment.py ...`` through the thin ``src/`` shim).
"""
import argparse
import copy
import inspect
import io
import os
import re
import time
import traceback
from os.path import join as path_join
from time import strftime

import numpy as np
import pandas as pd
import yaml

from deep_cbrs_amar_renaissance_amd import engine, models as models_pkg
from deep_cbrs_amar_renaissance_amd.data import loaders
from deep_cbrs_amar_renaissance_amd.models.basic import BasicRS, BasicGNN, BasicKnowledgeGCN, BasicTSGNN, BasicTWGNN
from deep_cbrs_amar_renaissance_amd.models.hybrid import HybridCBRS, HybridBertGNN
from deep_cbrs_amar_renaissance_amd.utilities.keras import get_total_parameters
from deep_cbrs_amar_renaissance_amd.utilities.metrics import top_k_predictions, top_k_metrics
from deep_cbrs_amar_renaissance_amd.utilities.utils import \
    get_experiment_logger, nested_dict_update, make_grid, mlflow_linearize, setup_mlflow

PARAMS_PATH = 'config.yaml'
EXPERIMENTS_PATH = 'experiments.yaml'
MLFLOW_PATH = './mlruns'
MLFLOW_EXP_NAME = 'SIS - Movielens-1M - BasicRS with Knowledge GNNs'
LOG_FREQUENCY = 100
METRICS_TOP_KS = [5, 10]

parser = argparse.ArgumentParser()
parser.add_argument("-c", "--config", dest='config', type=str, help="Config input file", default=PARAMS_PATH)
parser.add_argument("-e", "--experiments", dest='experiments', type=str,
                    help="Experiment (grid search) file", default=EXPERIMENTS_PATH)
parser.add_argument("--exp_name", dest='exp_name', type=str,
                    help="Name of the group of runs (used in the run log)", default=MLFLOW_EXP_NAME)


class _Yaml12Loader(yaml.SafeLoader):
    """PyYAML is YAML 1.1: '1e-4' (no dot) would load as a string. The reference reads its configs
    with ruamel (YAML 1.2), where it is a float — resolve floats the 1.2 way."""


_Yaml12Loader.add_implicit_resolver(
    'tag:yaml.org,2002:float',
    re.compile(r'^[-+]?(\.[0-9]+|[0-9]+(\.[0-9]*)?)([eE][-+]?[0-9]+)?$|^[-+]?\.(inf|Inf|INF)$|^\.(nan|NaN|NAN)$'),
    list('-+0123456789.'))


def load_yaml(path):
    with open(path, 'r') as fp:
        return yaml.load(fp, Loader=_Yaml12Loader)


class AttrDict(dict):
    """dict with attribute access, nested (the reference uses EasyDict)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __setitem__(self, key, value):
        if isinstance(value, dict) and not isinstance(value, AttrDict):
            value = AttrDict(value)
        super().__setitem__(key, value)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    __setattr__ = __setitem__


class Adam:
    """Optimizer placeholder with Keras' constructor (config.yaml:52-56); used once training lands."""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, **kwargs):
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon


OPTIMIZERS = {'Adam': Adam}


class Experimenter:
    def __init__(self, config, run_log):
        """Holds every object of one experiment and performs its training and evaluation."""
        self.run_log = run_log
        self.config = AttrDict(copy.deepcopy(config))
        engine.set_seed(self.config.seed)

        self.exp_name = strftime("%m_%d-%H_%M") + '-' + self.config.model.name
        if BasicRS.__name__ in self.config.model.name:
            pass
        elif HybridCBRS.__name__ in self.config.model.name:
            self.exp_name += '-' + ('feature' if self.config.model.feature_based else 'entity')
        else:
            self.exp_name += '-' + str(self.config.model.l2_regularizer) + '-' + self.config.model.final_node
        if self.config.get('details'):
            self.exp_name += '-' + self.config.details
        run_log.start_run(run_name=self.exp_name)
        run_log.log_params(mlflow_linearize(config))

        self.config.dest = path_join(run_log.run_dir, 'artifacts')
        self.predictions_dest = path_join(self.config.dest, "predictions")
        os.makedirs(self.predictions_dest, exist_ok=True)
        with open(path_join(self.config.dest, "config.yaml"), 'w') as fp:       # reproducibility copy
            yaml.safe_dump(config, fp)

        self.logger = get_experiment_logger(self.config.dest)
        buf = io.StringIO()
        yaml.safe_dump(config, buf)
        self.logger.info('CONFIG')
        self.logger.info(buf.getvalue())

        self._retrieve_classes()
        self.trainset = self.testset = self.model = self.optimizer = None
        self.parameters = self.config.parameters

    def _retrieve_classes(self):
        """Object classes from name strings (experiment.py:107-118)."""
        self.optimizer_class = OPTIMIZERS[self.config.parameters.optimizer.name]
        model_module, model_class = self.config.model.name.split('.')
        module = __import__(models_pkg.__name__ + '.' + model_module, fromlist=[model_class])
        self.model_class = getattr(module, model_class)
        self.load_function = getattr(loaders, self.config.dataset.load_function_name)

    def build_dataset(self):
        accepted = inspect.signature(self.load_function).parameters
        kwargs = {k: self.config.dataset[k] for k in self.config.dataset.keys() & accepted.keys()}
        self.trainset, self.testset = self.load_function(**kwargs)

    def build_optimizer(self):
        accepted = inspect.signature(self.optimizer_class).parameters
        opt_cfg = self.config.parameters.optimizer
        self.optimizer = self.optimizer_class(**{k: opt_cfg[k] for k in opt_cfg.keys() & accepted.keys()})

    def build_model(self):
        self.logger.info('Building model...')
        cls, model_cfg = self.model_class, dict(self.config.model)
        if issubclass(cls, (BasicKnowledgeGCN, BasicTSGNN, BasicTWGNN)):
            self.model = cls(len(self.trainset.users), len(self.trainset.items), self.trainset.adj_matrix, **model_cfg)
        elif issubclass(cls, (BasicGNN, HybridBertGNN)):
            self.model = cls(self.trainset.adj_matrix, **model_cfg)
        else:
            self.model = cls(**model_cfg)
        if hasattr(self.model, 'n_users'):                   # lets hoisted scoring run each tower on its own rows
            self.model.n_users, self.model.n_items = len(self.trainset.users), len(self.trainset.items)
        self.model.compile(loss=self.parameters.loss, optimizer=self.optimizer, metrics=self.parameters.metrics)
        self.model(self.trainset[0][0])                       # one prediction builds every weight
        self.model.summary(print_fn=self.logger.info, expand_nested=True)
        trainable, non_trainable = get_total_parameters(self.model)
        self.run_log.log_metrics({'trainable_params': trainable, 'non_trainable_params': non_trainable})

    def train(self):
        self.logger.info("Experiment folder: " + self.config.dest)
        self.build_dataset()
        self.build_optimizer()
        self.build_model()
        self.logger.info('Training:')
        t0 = time.perf_counter()
        # a model or reduction without a training recipe raises here: the experiment fails (MultiExperimenter.run_experiment
        # logs the traceback, ends the run and goes on with the grid, experiment.py:295-302) instead of evaluating random weights
        self.model.fit(self.trainset, epochs=self.parameters.epochs, workers=self.config.n_workers)
        # the reference's LogCallback reports the fit wall time as 'training_time' (utilities/keras.py:43-51, 69-85)
        self.run_log.log_metrics({'training_time': time.perf_counter() - t0})

    def evaluate(self):
        loss_acc = self.model.evaluate(self.testset)
        self.run_log.log_metrics({'test_loss': loss_acc[0], 'test_accuracy': loss_acc[1]})
        predictions = self.model.predict(self.testset)
        ratings_pred = np.concatenate([self.testset.ratings[:, [0, 1]], predictions], axis=1)
        precision_at, recall_at, f1_at = {}, {}, {}
        for k in METRICS_TOP_KS:
            top_predictions = top_k_predictions(ratings_pred, self.trainset.users, self.trainset.items, k=k)
            top_k_dest = path_join(self.predictions_dest, "top_{}".format(k))
            os.makedirs(top_k_dest, exist_ok=True)
            top_predictions.to_csv(path_join(top_k_dest, "predictions_1.tsv"), sep='\t', header=False, index=False)
            # the evaluator's switches are reachable from the experiment config (parameters.metrics_short_lists / metrics_no_relevant);
            # unset = RiVal's behaviour as restated (utilities/metrics.py), users skipped by them are logged with the metrics
            prm = self.config.parameters
            top_k_metrics(self.config.dataset.test_ratings_filepath, top_k_dest,
                          short_lists=prm.get('metrics_short_lists'), no_relevant=prm.get('metrics_no_relevant'))
            users_tsv = path_join(top_k_dest, "results_users.tsv")
            if os.path.exists(users_tsv):
                cnt = pd.read_csv(users_tsv, sep='\t').iloc[0]
                self.run_log.log_metrics({"users_skipped_short_list_at_{}".format(k): int(cnt['skipped_short_list']),
                                          "users_skipped_no_relevant_at_{}".format(k): int(cnt['skipped_no_relevant_item'])})
            results = pd.read_csv(path_join(top_k_dest, "results.tsv"), sep='\t', header=None)
            results = results.drop(0, axis=1).to_numpy().squeeze()
            precision_at[k], recall_at[k], f1_at[k] = results[0], results[1], results[2]
            self.run_log.log_metrics({"precision_at_{}".format(k): precision_at[k],
                                      "recall_at_{}".format(k): recall_at[k],
                                      "f1_at_{}".format(k): f1_at[k]})
        metrics = pd.DataFrame([precision_at, recall_at, f1_at], index=['precision_at', 'recall_at', 'f1_at'])
        self.logger.info('\n' + str(metrics))
        print('\n' + str(metrics))
        return metrics

    def run(self):
        self.train()
        metrics = self.evaluate()
        self.close()
        return metrics

    def close(self):
        for handler in list(self.logger.handlers):
            handler.close()
            self.logger.removeHandler(handler)
        self.run_log.end_run()


class MultiExperimenter:
    """Runs every experiment of an experiments file, each as overrides on the base config."""

    def __init__(self, params_path, experiments_path, run_log):
        self.run_log = run_log
        self.base_config = load_yaml(params_path)
        config = load_yaml(experiments_path) or {}
        self.experiments = dict(config.get('linear') or {})
        for grid in (config.get('grid') or {}).values():
            self.experiments.update({str(elem): elem for elem in make_grid(grid)})
        print("Retrieved experiments: {}".format(len(self.experiments)))
        for exp in self.experiments:
            print(exp)

    def run_experiment(self, exp_name):
        overrides = self.experiments[exp_name]
        config = copy.deepcopy(self.base_config)
        if overrides:                                        # None runs the base config
            config = nested_dict_update(config, overrides)
        print('-----------------------------------------------\n{}\n'.format(exp_name),
              '-----------------------------------------------\n')
        try:
            return Experimenter(config, self.run_log).run()
        except Exception as e:                               # keep going with the rest of the grid
            print(e)
            traceback.print_exc()
            self.run_log.end_run()
            return None

    def run(self):
        n_exp = len(self.experiments)
        results = {}
        for i, exp_name in enumerate(self.experiments):
            print("Experiment {}/{}".format(i + 1, n_exp))
            results[exp_name] = self.run_experiment(exp_name)
        return results


def main(argv=None):
    args = parser.parse_args(argv)
    run_log = setup_mlflow(args.exp_name, MLFLOW_PATH)
    MultiExperimenter(args.config, args.experiments, run_log).run()


if __name__ == "__main__":
    main()
