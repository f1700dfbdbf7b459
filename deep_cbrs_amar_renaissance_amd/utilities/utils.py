"""Config plumbing of the experiment driver — mirrors `/root/reference/src/utilities/utils.py:19-168`.

``nested_dict_update`` overlays an experiment's overrides on the base config; ``make_grid``
expands a nested dict of lists into the cartesian product of nested dicts.  MLflow is not
installed anywhere this runs, so run tracking is a JSON-lines file with the same keys
(``RunLog``) instead of an mlruns directory.
"""
import collections.abc
import itertools
import json
import logging
import os
import time
import uuid


def nested_dict_update(d, u):
    """Recursive dict.update: mappings merge, everything else overwrites."""
    for k, v in u.items():
        if isinstance(v, collections.abc.Mapping):
            d[k] = nested_dict_update(d.get(k, {}), v)
        else:
            d[k] = v
    return d


def linearize(dictionary):
    """Nested dict of lists -> [(key path tuple, list), ...] in insertion order."""
    flat = []
    for key, value in dictionary.items():
        if isinstance(value, collections.abc.Mapping):
            flat.extend(((key,) + path, lst) for path, lst in linearize(value))
        elif isinstance(value, list):
            flat.append(((key,), value))
        else:
            raise ValueError("Only dict or lists!!!")
    return flat


def delinearize(items):
    """[(key path tuple, value), ...] -> nested dict."""
    out = {}
    for path, value in items:
        node = out
        for key in path[:-1]:
            node = node.setdefault(key, {})
        node[path[-1]] = value
    return out


def make_grid(dict_of_list):
    """All combinations of the listed values, each as a nested dict shaped like the input."""
    flat = linearize(dict_of_list)
    paths = [p for p, _ in flat]
    return [delinearize(zip(paths, combo)) for combo in itertools.product(*[v for _, v in flat])]


def mlflow_linearize(dictionary):
    """Nested dict -> {'a.b.c': value} (keys as mlflow.log_params receives them)."""
    out = {}
    for key, value in dictionary.items():
        if isinstance(value, collections.abc.Mapping):
            out.update({'{}.{}'.format(key, k): v for k, v in mlflow_linearize(value).items()})
        else:
            out[key] = value
    return out


class RunLog:
    """JSON-lines stand-in for the MLflow tracking calls the reference makes (same keys)."""

    def __init__(self, root, exp_name):
        self.root = os.path.join(root, exp_name.replace(os.sep, '_').replace(' ', '_'))
        os.makedirs(self.root, exist_ok=True)
        self.run_id = self.run_dir = None

    def start_run(self, run_name):
        self.run_id = uuid.uuid4().hex
        self.run_dir = os.path.join(self.root, self.run_id)
        os.makedirs(os.path.join(self.run_dir, 'artifacts'), exist_ok=True)
        self._write({'event': 'start_run', 'run_name': run_name})
        return self.run_id

    def _write(self, record):
        record['time'] = time.time()
        with open(os.path.join(self.run_dir, 'run.jsonl'), 'a') as fp:
            fp.write(json.dumps(record, default=str) + '\n')

    def log_params(self, params):
        self._write({'event': 'params', 'params': params})

    def log_metrics(self, metrics):
        self._write({'event': 'metrics', 'metrics': metrics})

    def end_run(self):
        if self.run_dir is not None:
            self._write({'event': 'end_run'})
        self.run_id = self.run_dir = None


def setup_mlflow(exp_name, mlflow_path):
    """Returns the directory that holds this experiment group's runs (utils.py:102-129)."""
    return RunLog(mlflow_path, exp_name)


class FlushFileHandler(logging.FileHandler):
    def emit(self, record):
        super().emit(record)
        self.flush()


def get_experiment_logger(destination_folder):
    logger = logging.getLogger(destination_folder)
    logger.setLevel(logging.INFO)
    handler = FlushFileHandler(os.path.join(destination_folder, 'log.txt'))
    handler.setFormatter(logging.Formatter('%(asctime)s %(message)s'))
    logger.addHandler(handler)
    return logger
