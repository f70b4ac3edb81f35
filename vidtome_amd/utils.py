"""Patch helpers -- the counterpart of vidtome/utils.py (same names and behaviour).

The patched block itself does not use the closure-composition helpers (`func_warper` / `join_warper` / `split_warper`,
vidtome/utils.py:42-60): its merge / unmerge chains are composed into single row maps on the device (`vtm_compose`, see
patch.compute_merge), and joining / splitting frames are views.  They are kept for code written against the reference
that imports them."""
from __future__ import annotations

import torch


def isinstance_str(x: object, cls_name: str) -> bool:
    """vidtome/utils.py:4-16: True if any class in x's MRO is *named* cls_name (no import of the class)."""
    return any(_cls.__name__ == cls_name for _cls in x.__class__.__mro__)


# Which RNG stream the block generators fork.  "cpu" (default): the CPU state, whatever device the model is on -- the draw
# stream of the reference's CPU path, which is the parity oracle.  "device": the reference's own rule (vidtome/utils.py:18-30)
# -- the CUDA generator's state when the model is on a GPU -- so that a run of the REFERENCE ON A GPU can be reproduced draw
# for draw (one host read-back per draw).  Chosen per model with apply_patch(..., generator_device=...) or process-wide with
# VIDTOME_GENERATOR=cpu|device.
import os

GENERATOR_MODE = os.environ.get("VIDTOME_GENERATOR", "cpu")


def init_generator(device: torch.device, fallback: torch.Generator = None, mode: str = None) -> torch.Generator:
    """vidtome/utils.py:18-30: fork the current default RNG state into a private generator.

    ``mode`` "cpu" (default, see GENERATOR_MODE): always ``torch.Generator('cpu').set_state(torch.get_rng_state())`` --
    the draws (one randint per local level, one rand per global merge) are host-side control decisions and the parity
    oracle is the reference's CPU path, so the stream stays identical to it and no draw needs a device sync.
    ``mode`` "device": the reference's behaviour to the letter -- CPU tensors fork the CPU state, CUDA tensors fork
    ``torch.cuda.get_rng_state()`` into a generator on that device, any other device type keeps ``fallback`` (or forks
    the CPU state)."""
    mode = GENERATOR_MODE if mode is None else mode
    if mode not in ("cpu", "device"):
        raise ValueError(f"generator mode must be 'cpu' or 'device', got {mode!r}")
    if mode == "cpu" or device is None:
        return torch.Generator(device="cpu").set_state(torch.get_rng_state())
    device = torch.device(device)
    if device.type == "cpu":
        return torch.Generator(device="cpu").set_state(torch.get_rng_state())
    if device.type == "cuda":
        return torch.Generator(device=device).set_state(torch.cuda.get_rng_state())
    return fallback if fallback is not None else init_generator(torch.device("cpu"), mode=mode)


def join_frame(x: torch.Tensor, fsize: int) -> torch.Tensor:
    """vidtome/utils.py:32-35: '(B F) N C -> B (F N) C' -- a pure view for contiguous x."""
    BF, N, C = x.shape
    return x.reshape(BF // fsize, fsize * N, C)


def split_frame(x: torch.Tensor, fsize: int) -> torch.Tensor:
    """vidtome/utils.py:37-40: 'B (F N) C -> (B F) N C'."""
    B, FN, C = x.shape
    return x.reshape(B * fsize, FN // fsize, C)


def func_warper(funcs):
    """vidtome/utils.py:42-48: one callable applying ``funcs`` left to right, keyword arguments passed to each."""
    def fn(x, **kwarg):
        for f in funcs:          # the list itself, not a copy: the reference's closure sees later mutations too
            x = f(x, **kwarg)
        return x
    return fn


def join_warper(fsize: int):
    """vidtome/utils.py:50-54."""
    return lambda x: join_frame(x, fsize)


def split_warper(fsize: int):
    """vidtome/utils.py:56-60."""
    return lambda x: split_frame(x, fsize)
