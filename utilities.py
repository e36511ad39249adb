"""File helpers the recipes use (subset of the reference's utilities.py: load / save by extension, mkdirs; reference
utilities.py:27-58,203-226).  The offline feature extraction (get_VQT via librosa, MIDI tools) is data preparation and is
not part of this repository's scope."""
import json
import os
import pickle

import numpy as np
import yaml


def mkdirs(path):
    os.makedirs(path, exist_ok=True)


def load(path):
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        return np.load(path)
    if ext == ".json":
        with open(path) as f:
            return json.load(f)
    if ext in (".pkl", ".pickle"):
        with open(path, "rb") as f:
            return pickle.load(f)
    if ext in (".yaml", ".yml"):
        with open(path) as f:
            return yaml.safe_load(f)
    with open(path) as f:
        return f.read()


def save(obj, path):
    mkdirs(os.path.dirname(os.path.abspath(path)))
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        np.save(path, obj)
    elif ext == ".json":
        with open(path, "w") as f:
            json.dump(obj, f)
    elif ext in (".pkl", ".pickle"):
        with open(path, "wb") as f:
            pickle.dump(obj, f)
    elif ext in (".yaml", ".yml"):
        with open(path, "w") as f:
            yaml.safe_dump(obj, f)
    else:
        with open(path, "w") as f:
            f.write(str(obj))
