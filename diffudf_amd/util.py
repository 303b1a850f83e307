# coding: utf-8
"""Experiment plumbing with the reference's signatures — reference src/util.py:10-38."""
import json
import os
import shutil


def create_output_paths(checkpoint_path, experiment_name, overwrite=True):
    """<checkpoint_path>/<experiment_name>/{models,reconstructions}; an existing directory is reused when
    overwrite is False and wiped when it is True."""
    full_path = os.path.join(checkpoint_path, experiment_name)
    if os.path.exists(full_path) and overwrite:
        shutil.rmtree(full_path)
    for sub in ("models", "reconstructions"):
        os.makedirs(os.path.join(full_path, sub), exist_ok=True)
    return full_path


def load_experiment_parameters(parameters_path):
    try:
        with open(parameters_path, "r") as fin:
            return json.load(fin)
    except FileNotFoundError:
        print("File '{}' not found.".format(parameters_path))
        return {}


def normalize(arr):
    """Reference src/util.py:34-39: a 1-D array is divided by its norm, a 2-D (M,3) array ROW BY ROW (a zero vector
    or row gives nan/inf exactly like the reference's plain division)."""
    import numpy as np
    arr = np.asarray(arr)
    if arr.ndim == 1:
        return arr / np.linalg.norm(arr)
    return arr / np.linalg.norm(arr, axis=1, keepdims=True)
