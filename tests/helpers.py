"""Shared test helpers (fixture loading, torch-generator draws)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_cases(fname):
    z = np.load(os.path.join(GOLDEN, fname), allow_pickle=False)
    n = int(z["n_cases"])
    cases = [dict() for _ in range(n)]
    for k in z.files:
        if k == "n_cases":
            continue
        i, name = k.split("/", 1)
        v = z[k]
        cases[int(i)][name] = v.item() if v.ndim == 0 else v
    return cases


def load_chain(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    cfg = json.loads(str(z["cfg_json"]))
    return cfg, z


def forked_generator_from_state(rng_state):
    """vidtome/utils.py:22-23: torch.Generator('cpu').set_state(torch.get_rng_state())."""
    import torch
    return torch.Generator(device="cpu").set_state(torch.from_numpy(np.asarray(rng_state)))
