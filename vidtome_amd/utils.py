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


def init_generator(device: torch.device, fallback: torch.Generator = None) -> torch.Generator:
    """vidtome/utils.py:18-30: fork the current default RNG state into a private generator.

    The reference forks the *device's* generator (the CUDA generator when the model is on a GPU).  The
    parity oracle is the reference's CPU path, and the draws (one randint per local level, one rand per
    global merge) are host-side control decisions, so this implementation always forks the CPU state:
    ``torch.Generator('cpu').set_state(torch.get_rng_state())``.  That keeps the draw stream identical to
    the reference CPU path and avoids a device sync per draw.
    """
    return torch.Generator(device="cpu").set_state(torch.get_rng_state())


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
