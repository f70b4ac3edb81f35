"""Plug-and-Play attention control for the patched blocks -- counterpart of the attention part of
utils/pnp_utils.py (register_time :12-37, register_attention_control :39-106).

The reference replaces ``attn1.forward`` by a closure that materialises softmax(QK^T) with einsum and, at
injection timesteps, reuses the SOURCE sample's probabilities for every batch group.  Here the same
semantics are attributes read by ``vidtome_amd.patch.self_attention``: the shared-probability mode of
``vtm_attention`` (q/k of sample ``b % (B / num_inputs)``, v per sample) -- nothing is materialised.

If a pipeline already called the reference's own ``register_attention_control``, nothing needs to be
re-registered: ``patch._pnp_num_inputs`` recognises that closure and routes it to the same kernel mode.
"""
from __future__ import annotations


def register_attention_control(model, injection_schedule, num_inputs):
    """utils/pnp_utils.py:98-105: decoder blocks 4-11 (up_blocks[1].attentions[1,2], up_blocks[2,3].*)."""
    res_dict = {1: [1, 2], 2: [0, 1, 2], 3: [0, 1, 2]}
    for res in res_dict:
        for block in res_dict[res]:
            module = model.unet.up_blocks[res].attentions[block].transformer_blocks[0].attn1
            setattr(module, "injection_schedule", injection_schedule)
            setattr(module, "vtm_num_inputs", num_inputs)
    return model


def register_time(model, t):
    """utils/pnp_utils.py:12-37: stamp the current timestep on everything the PnP hooks read it from -- the
    self- and cross-attention of every transformer block AND the resnets of the down / up blocks (the reference's
    `register_conv_control` forward, pnp_utils.py:108-172, reads `self.t` on a resnet, and `init_pnp`,
    generate.py:317-320, always installs it).  Walks whatever blocks the UNet has instead of the reference's fixed
    SD index tables; on an SD UNet the two visit the same modules."""
    unet = model.unet
    groups = list(getattr(unet, "up_blocks", [])) + list(getattr(unet, "down_blocks", []))
    for g in groups:
        for res in getattr(g, "resnets", []):
            setattr(res, "t", t)
    mid = getattr(unet, "mid_block", None)
    if mid is not None:
        groups.append(mid)              # the reference stamps the mid block's attention only (pnp_utils.py:34-37)
    for g in groups:
        for att in getattr(g, "attentions", []):
            blk = att.transformer_blocks[0]
            setattr(blk.attn1, "t", t)
            if getattr(blk, "attn2", None) is not None:
                setattr(blk.attn2, "t", t)
    return model
