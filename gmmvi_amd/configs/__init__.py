"""Configuration helpers (reference: src/gmmvi/configs/__init__.py:5-59).

The reference keeps its defaults in 28 YAML files; here the same keys and values live in ``defaults.py`` as Python
dictionaries and are merged with the reference's semantics (codename letters -> module defaults; ``update_config`` =
deep merge where the update REPLACES leaves, configs/__init__.py:57-59 ``Strategy.REPLACE``).
"""
import copy

import yaml

from .defaults import MODULE_DEFAULTS, EXPERIMENT_DEFAULTS


def load_yaml(filename):
    """configs/__init__.py:5-11."""
    with open(filename, 'r') as stream:
        try:
            return yaml.safe_load(stream)
        except yaml.YAMLError as exc:
            print(exc)


def _merge(dst, src):
    """deep merge, lists and scalars replaced (mergedeep Strategy.REPLACE)."""
    for key, val in src.items():
        if isinstance(val, dict) and isinstance(dst.get(key), dict):
            _merge(dst[key], val)
        else:
            dst[key] = copy.deepcopy(val)
    return dst


def get_default_algorithm_config(algorithm_id):
    """configs/__init__.py:13-45: one letter per design choice, merged in order."""
    print(f"Using default parameters for codename {algorithm_id}")
    merged = dict()
    for letter in algorithm_id:
        key = letter.upper()
        if key not in MODULE_DEFAULTS:
            raise KeyError(key)
        _merge(merged, MODULE_DEFAULTS[key])
    return merged


def get_default_experiment_config(experiment_id):
    """configs/__init__.py:47-50."""
    print(f"Using default parameters for experiment {experiment_id}")
    if experiment_id not in EXPERIMENT_DEFAULTS:
        raise FileNotFoundError(f"{experiment_id}.yml")
    return copy.deepcopy(EXPERIMENT_DEFAULTS[experiment_id])


def get_default_config(algorithm_id, experiment_id):
    """configs/__init__.py:52-55."""
    return {**get_default_algorithm_config(algorithm_id), **get_default_experiment_config(experiment_id)}


def update_config(default_values, updates):
    """configs/__init__.py:57-59."""
    return _merge(copy.deepcopy(dict(default_values)), updates)
