"""CPU tests of the host logic and of the C-ABI library surface (no compute calls without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from vidtome_amd import build
    return build.build()


def test_library_exports_every_declared_symbol(built):
    """Every function include/vidtome_hip.h declares is exported by the .so and bound in _lib."""
    from vidtome_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "vidtome_hip.h")).read()
    declared = set(re.findall(r"\b(vtm_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"vtm_match_row_pad"}
    lib = ctypes.CDLL(built)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.exported_symbols())
    assert _lib.lib().vtm_version() == _lib.ABI_VERSION == 2
    assert _lib.lib().vtm_pad_rows(257) == 512 and _lib.lib().vtm_pad_k(320) == 320 and _lib.lib().vtm_pad_k(40) == 64


def test_shipped_library_has_no_ablation_switch(built):
    """The experiment switches of the hand-scheduled kernels (csrc/ablate.h: drop loads / barriers / stores, WRONG results)
    live in one header; the kernel sources carry no #ifdef of theirs, the library the tests load was built with none of
    them, and _lib refuses a library that was."""
    import glob
    from vidtome_amd import _lib
    assert _lib.lib().vtm_build_ablations() == 0
    assert ctypes.CDLL(built).vtm_build_ablations() == 0
    csrc = os.path.join(ROOT, "vidtome_amd", "csrc")
    for path in glob.glob(os.path.join(csrc, "*.hip")) + [os.path.join(csrc, "common.h")]:
        src = open(path).read()
        assert not re.search(r"#\s*if(n?def)?[^\n]*VTM_(EXP|LIN_NO)", src), path
    hdr = open(os.path.join(csrc, "ablate.h")).read()
    for name in re.findall(r"#ifdef (VTM_(?:EXP|LIN)_[A-Z]+)", hdr):       # every switch has a bit in the mask
        assert re.search(r"VTM_ABL_BIT_" + name.replace("VTM_EXP_", "").replace("VTM_LIN_", "LIN_"), hdr), name


def test_hot_kernels_keep_their_register_budget(built):
    """Code-object metadata of the built gfx950 kernels: the filter's hand-counted memory pipeline must not spill (a
    build whose SGPRs spilled faulted on the GPU) and must fit two waves per SIMD; the d = 40 attention kernel must keep
    four waves per SIMD (<= 128 VGPRs, no scratch); the shipped GEMM tiles fit two."""
    from vidtome_amd import build
    res = {}
    for obj in ("match_filter.o", "attention.o", "linear.o", "ff.o"):
        res.update(build.kernel_resources(os.path.join(build.LIBDIR, obj)))

    def only(*parts):
        hits = {k: v for k, v in res.items() if all(p in k for p in parts)}
        assert hits, parts
        return hits

    for k, v in only("filter_kernel").items():
        assert v["sgpr_spill_count"] == 0 and v["vgpr_spill_count"] == 0 and v["private_segment_fixed_size"] == 0, (k, v)
        assert v["vgpr_count"] <= 256, (k, v)
    for k, v in only("16attention_kernel", "Li40E").items():      # half and bf16
        assert v["vgpr_count"] <= 128 and v["vgpr_spill_count"] == 0 and v["private_segment_fixed_size"] == 0, (k, v)
    # the panel GEMMs share the filter's hand-issued memory pipeline: no spills of either kind, two workgroups per CU
    for k, v in only("panel_gemm_kernel").items():
        assert v["sgpr_spill_count"] == 0 and v["vgpr_spill_count"] == 0 and v["private_segment_fixed_size"] == 0, (k, v)
        assert v["vgpr_count"] <= 256, (k, v)
    for name in ("linear_rows_kernel", "linear_rows_ws_kernel"):
        for k, v in only(name).items():
            assert v["vgpr_count"] <= 256 and v["vgpr_spill_count"] == 0 and v["private_segment_fixed_size"] == 0, (k, v)


def test_partition_counts_match_reference_arithmetic(built, oracle):
    """vtm_partition_counts (host helper) vs the reference's own boolean-mask construction (merge.py:52-69)."""
    from vidtome_amd import _lib
    for (N, unm_pre, F, stride) in [(65536, 0, 16, 4), (40960, 24576, 4, 4), (64, 0, 4, 4), (60, 0, 6, 4), (72, 7, 5, 2),
                                    (110592 * 2, 0, 16, 4), (33, 0, 3, 4), (101, 5, 7, 3)]:
        tnum = (N - unm_pre) // F
        ts = min(stride, F)
        for randf in range(ts):
            idx = np.arange(N - unm_pre)
            sel = (idx // tnum) % ts == randf
            assert _lib.partition_counts(N, unm_pre, tnum, ts, randf) == (int((~sel).sum()), int(sel.sum()) + unm_pre)


def test_error_convention(built):
    from vidtome_amd import _lib
    L = _lib.lib()
    ns, nd = ctypes.c_int64(), ctypes.c_int64()
    rc = L.vtm_partition_counts(10, 20, 1, 1, 0, ctypes.byref(ns), ctypes.byref(nd))
    assert rc == -1 and b"vtm_partition_counts" in L.vtm_last_error()
    # null pointers are rejected before any launch
    assert L.vtm_match(None, None, 1, 1, 1, 128, 128, 32, 0, None, None) == -1
    assert L.vtm_sort_desc(None, 1, 1, None, None, 0, None) == -1
    assert L.vtm_attention(None, 8, None, 8, None, 8, None, 8, 1, 1, 1, 8, 8, 40, 1.0, 1, None, 0, None) == -1


def test_workspace_sizing_is_pure_host_arithmetic(built):
    """The *_ws_bytes exports are host-side planners (no device needed; the CU count defaults to MI355X's 256):
    the attention tail plan asks for a workspace exactly when the last round of workgroups is at most a quarter
    full behind at least one whole round (attention.hip, plan_tail)."""
    from vidtome_amd import _lib
    L = _lib.lib()
    ws = L.vtm_attention_ws_bytes
    # d = 40: 256-query workgroups, 512 resident -> 2 x 8 x 128 = 2048 = 4.00 rounds / 2176 = 4.25 / 2304 = 4.5
    assert ws(2, 8, 32768, 52224, 40) == 0
    assert ws(2, 8, 34816, 52224, 40) > 0
    assert ws(2, 8, 36864, 52224, 40) == 0
    # d = 80: 512-query workgroups, 256 resident -> cfg-2 mid blocks: 17 x 16 = 272 workgroups
    assert ws(2, 8, 8704, 13056, 80) > 0
    assert ws(2, 8, 8192, 13056, 80) == 0
    # short key axes and launches below one round are never split
    assert ws(2, 8, 34816, 77, 40) == 0 and ws(1, 1, 300, 52224, 40) == 0
    assert ws(0, 8, 34816, 52224, 40) == 0 and ws(2, 8, 34816, 52224, 41) == 0
    assert L.vtm_sort_ws_bytes(2, 49152) >= 2 * 49152 * 16 and L.vtm_sort_ws_bytes(0, 10) == 0
    assert L.vtm_match_filtered_ws_bytes(2, 320, 49152, 16384, 0) > L.vtm_match_filtered_ws_bytes(2, 320, 4096, 4096, 0) > 0


def test_cpu_tensors_fail_loudly(built):
    """There is no CPU fallback: the mirrored API raises on CPU tensors instead of computing elsewhere."""
    from vidtome_amd import merge
    x = torch.randn(2, 16, 8)
    with pytest.raises(RuntimeError, match="GPU only"):
        merge.bipartite_soft_matching_randframe(x, 4, 0.5, 0, torch.Generator().manual_seed(1))
    with pytest.raises(RuntimeError, match="GPU only"):
        merge.bipartite_soft_matching_2s(x, 8, 0.5, False)
    assert merge.do_nothing(x) is x                                   # merge.py:5-6


def test_missing_library_is_an_error(built, monkeypatch):
    from vidtome_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libvidtome_hip.so")
    with pytest.raises(RuntimeError, match="no CPU / eager fallback"):
        _lib.lib()


def test_patch_api_mechanics(built):
    """apply/remove/update/collect: class swap named ToMeBlock with _parent, shared _tome_info, hooks, kwargs."""
    import inspect

    import vidtome_amd
    from standin import Pipe, StandInUNet

    unet = StandInUNet(16, 2)
    pipe = Pipe(unet)
    sig = inspect.signature(vidtome_amd.apply_patch)
    pos = [(k, v.default) for k, v in list(sig.parameters.items())[1:] if v.kind is not inspect.Parameter.KEYWORD_ONLY]
    assert pos == [
        ("local_merge_ratio", 0.9), ("merge_global", False), ("global_merge_ratio", 0.8), ("max_downsample", 2),
        ("seed", 123), ("batch_size", 2), ("include_control", False), ("align_batch", False),
        ("target_stride", 4), ("global_rand", 0.5)]                                     # patch.py:234-245
    # the one addition is keyword-only and defaults to the reference-CPU-path behaviour
    kwonly = {k: v.default for k, v in sig.parameters.items() if v.kind is inspect.Parameter.KEYWORD_ONLY}
    assert kwonly == {"generator_device": None}
    out = vidtome_amd.apply_patch(unet, local_merge_ratio=0.5, merge_global=True)
    assert out is unet
    blocks = list(unet.blocks())
    assert all(b.__class__.__name__ == "ToMeBlock" and b._parent.__name__ == "BasicTransformerBlock" for b in blocks)
    assert all(b._tome_info is unet._tome_info for b in blocks)
    assert unet._tome_info["args"]["local_merge_ratio"] == 0.5 and unet._tome_info["args"]["merge_global"] is True
    assert len(unet._tome_info["hooks"]) == 1 + len(blocks)
    assert all(b.use_ada_layer_norm is False and b.use_ada_layer_norm_zero is False for b in blocks)
    vidtome_amd.update_patch(unet, global_tokens=None, foo=3)
    assert unet.foo == 3 and all(b.foo == 3 and b.global_tokens is None for b in blocks)   # root too (patch.py:366-369)
    got = vidtome_amd.collect_from_patch(unet, attr="foo")
    assert got[""] == 3 and len(got) == 1 + len(blocks)
    # size hook: first positional arg's (H, W)
    unet._forward_pre_hooks[next(iter(unet._forward_pre_hooks))](unet, (torch.zeros(8, 4, 6, 10),))
    assert unet._tome_info["size"] == (6, 10)
    # generator hook forks the global CPU RNG state and leaves it untouched
    torch.manual_seed(123)
    before = torch.get_rng_state().clone()
    blk = blocks[0]
    blk._forward_pre_hooks[next(iter(blk._forward_pre_hooks))](blk, (torch.zeros(1),))
    assert torch.equal(torch.get_rng_state(), before)
    draws = [int(torch.randint(0, 4, torch.Size([1]), generator=blk.generator)) for _ in range(8)]
    assert draws == [2, 1, 2, 2, 0, 2, 2, 1]                                               # SURVEY.md 8c KAT
    # re-applying re-patches cleanly; removing restores the class and the hooks
    vidtome_amd.apply_patch(unet)
    assert len(unet._tome_info["hooks"]) == 1 + len(blocks)
    ret = vidtome_amd.remove_patch(unet)
    assert ret is unet and all(b.__class__.__name__ == "BasicTransformerBlock" for b in blocks)
    assert len(unet._tome_info["hooks"]) == 0 and len(unet._forward_pre_hooks) == 0
    with pytest.raises(RuntimeError, match="Stable Diffusion / Latent Diffusion"):
        vidtome_amd.apply_patch(torch.nn.Linear(2, 2))                                      # patch.py:283-286


def test_pnp_closure_detection():
    """A module whose forward was replaced by a closure holding `num_inputs` (what the reference's
    register_attention_control does, pnp_utils.py:39-97) is recognised."""
    from vidtome_amd import patch as vpatch
    from standin import Attention

    def register(mod, num_inputs):
        def sa_forward(self):
            def forward(x, encoder_hidden_states=None, attention_mask=None, **kw):
                return x * num_inputs
            return forward
        mod.forward = sa_forward(mod)

    a = Attention(16, 2)
    assert vpatch._pnp_num_inputs(a) is None
    register(a, 3)
    assert vpatch._pnp_num_inputs(a) == 3
    a.vtm_num_inputs = 2
    assert vpatch._pnp_num_inputs(a) == 2


class _ControlNet(torch.nn.Module):                      # a second patched tree, named like Diffusers' ModelMixin models
    def __init__(self):
        super().__init__()
        from standin import BasicTransformerBlock
        self.blocks = torch.nn.ModuleList([BasicTransformerBlock(16, 2) for _ in range(2)])


class DiffusionPipeline:                                 # apply_patch tests class NAMES in the MRO (patch.py:279-280)
    def __init__(self, unet):
        self.unet = unet


class StableDiffusionControlNetPipeline(DiffusionPipeline):                       # ... and this one at patch.py:292
    def __init__(self, unet, controlnet):
        self.unet, self.controlnet = unet, controlnet


def test_apply_patch_on_pipelines_and_controlnet(built):
    """patch.py:279-295, 337-355: a pipeline is patched through `.unet`; `include_control` only acts on a class named
    StableDiffusionControlNetPipeline and then patches `.controlnet` with its own _tome_info; remove_patch looks for
    the ControlNet on the UNet (the reference's quirk), so a pipeline's ControlNet stays patched."""
    import vidtome_amd
    from standin import Pipe, StandInUNet

    # (1) a plain pipeline object (class not named like Diffusers') is rejected like any non-diffusers model
    with pytest.raises(RuntimeError, match="Stable Diffusion / Latent Diffusion"):
        vidtome_amd.apply_patch(Pipe(StandInUNet(16, 2)))
    # (2) a DiffusionPipeline: patched through .unet, the pipeline object itself is returned
    unet = StandInUNet(16, 2)
    pipe = DiffusionPipeline(unet)
    assert vidtome_amd.apply_patch(pipe, merge_global=True, batch_size=3) is pipe
    assert all(b.__class__.__name__ == "ToMeBlock" for b in unet.blocks())
    assert unet._tome_info["args"]["batch_size"] == 3 and not hasattr(pipe, "_tome_info")
    assert vidtome_amd.update_patch(pipe, global_tokens=None) is unet            # the reference returns the tree
    assert set(vidtome_amd.collect_from_patch(pipe, attr="global_tokens")) == \
        {n for n, m in unet.named_modules() if hasattr(m, "_tome_info")}
    assert vidtome_amd.remove_patch(pipe) is unet
    assert all(b.__class__.__name__ == "BasicTransformerBlock" for b in unet.blocks())

    # (3) ControlNet pipeline: untouched without include_control (what generate.py:97-98 does) ...
    unet, cn = StandInUNet(16, 2), _ControlNet()
    cpipe = StableDiffusionControlNetPipeline(unet, cn)
    vidtome_amd.apply_patch(cpipe)
    assert not hasattr(cn, "_tome_info") and all(b.__class__.__name__ == "BasicTransformerBlock" for b in cn.blocks)
    # ... patched with it, with a _tome_info of its own (sizes are recorded per model, patch.py:297-313)
    vidtome_amd.apply_patch(cpipe, include_control=True, local_merge_ratio=0.7)
    assert all(b.__class__.__name__ == "ToMeBlock" for b in cn.blocks)
    assert cn._tome_info is not unet._tome_info and cn._tome_info["args"]["local_merge_ratio"] == 0.7
    assert all(b._tome_info is cn._tome_info for b in cn.blocks)
    assert len(cn._tome_info["hooks"]) == 1 + len(cn.blocks)
    # update / collect see both trees (they look for .controlnet on the object they are given, patch.py:361,376)
    assert vidtome_amd.update_patch(cpipe, foo=7) is cn
    assert unet.foo == 7 and cn.foo == 7 and all(b.foo == 7 for b in cn.blocks)
    got = vidtome_amd.collect_from_patch(cpipe, attr="foo")
    assert "blocks.0" in got and "up_blocks.1.attentions.0.transformer_blocks.0" in got
    # remove_patch(pipe) unwraps to the UNet first and looks for .controlnet THERE: the ControlNet stays patched
    assert vidtome_amd.remove_patch(cpipe) is unet
    assert all(b.__class__.__name__ == "BasicTransformerBlock" for b in unet.blocks())
    assert all(b.__class__.__name__ == "ToMeBlock" for b in cn.blocks) and len(cn._tome_info["hooks"]) == 3
    # a UNet that carries the ControlNet as an attribute is cleaned completely (the case the quirk serves)
    unet.controlnet = cn
    assert vidtome_amd.remove_patch(unet) is cn
    assert all(b.__class__.__name__ == "BasicTransformerBlock" for b in cn.blocks) and not cn._tome_info["hooks"]
    # include_control on a pipeline with another name is ignored (patch.py:292)
    other = DiffusionPipeline(StandInUNet(16, 2))
    other.controlnet = _ControlNet()
    vidtome_amd.apply_patch(other, include_control=True)
    assert not hasattr(other.controlnet, "_tome_info")


def test_fused_attention_predicate(built):
    """The fused attn1 / attn2 path is taken only for the plain arithmetic; everything else calls the module."""
    from standin import Attention
    from vidtome_amd import patch as vpatch

    class FakeCuda(torch.Tensor):          # predicate needs x.is_cuda; emulate it without a device
        @property
        def is_cuda(self):
            return True

    x = torch.zeros(2, 8, 64).as_subclass(FakeCuda)
    ok = lambda a, **k: vpatch.fused_attention_ok(a, x, **k)
    a = Attention(64, 2)
    assert ok(a)
    assert not vpatch.fused_attention_ok(a, torch.zeros(2, 8, 64))              # CPU tensor
    b = Attention(48, 2)                                                          # head dim 24: no kernel instantiation
    assert not vpatch.fused_attention_ok(b, torch.zeros(2, 8, 48).as_subclass(FakeCuda))

    class LoRALinear(torch.nn.Module):                                            # PEFT-style wrapper, not a Linear
        def __init__(self, base):
            super().__init__()
            self.base_layer = base
            self.weight = base.weight

    a = Attention(64, 2)
    a.to_k = LoRALinear(a.to_k)
    assert not ok(a)

    class LoRACompatibleLinear(torch.nn.Linear):                                  # Diffusers' subclass
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.lora_layer = None
    a = Attention(64, 2)
    a.to_q = LoRACompatibleLinear(64, 64, bias=False)
    assert ok(a)
    a.to_q.lora_layer = torch.nn.Linear(64, 64)
    assert not ok(a)
    for attr, val in (("rescale_output_factor", 2.0), ("residual_connection", True), ("group_norm", torch.nn.Identity()),
                      ("norm_cross", torch.nn.Identity())):
        a = Attention(64, 2)
        setattr(a, attr, val)
        assert not ok(a), attr

    class CustomProc:
        pass

    class AttnProcessor2_0:
        pass
    a = Attention(64, 2)
    a.processor = CustomProc()
    assert not ok(a)
    a.processor = AttnProcessor2_0()
    assert ok(a)
    a.upcast_attention = True                         # SD-2.1: fp32 scores -- what the kernel does anyway
    assert ok(a)
    a = Attention(64, 2)
    a.forward = lambda x, **k: x                      # replaced forward that is not the PnP closure
    assert not ok(a)

    def register(mod, num_inputs):                    # the reference's PnP closure IS understood (self-attention only)
        def forward(x, encoder_hidden_states=None, attention_mask=None, **kw):
            return x * num_inputs
        mod.forward = forward
    a = Attention(64, 2)
    register(a, 3)
    assert ok(a) and not ok(a, self_attn=False)


def test_pnp_register_time_stamps_resnets(built):
    """utils/pnp_utils.py:12-37 sets `t` on attn1, attn2 AND the down / up resnets (register_conv_control reads it)."""
    from vidtome_amd import pnp

    class Blk(torch.nn.Module):
        def __init__(self, attn):
            super().__init__()
            self.resnets = torch.nn.ModuleList([torch.nn.Identity(), torch.nn.Identity()])
            if attn:
                t = torch.nn.Module()
                t.transformer_blocks = torch.nn.ModuleList([torch.nn.Module()])
                t.transformer_blocks[0].attn1 = torch.nn.Identity()
                t.transformer_blocks[0].attn2 = torch.nn.Identity()
                self.attentions = torch.nn.ModuleList([t])

    class U(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.up_blocks = torch.nn.ModuleList([Blk(False), Blk(True)])
            self.down_blocks = torch.nn.ModuleList([Blk(True), Blk(False)])
            self.mid_block = Blk(True)

    class P:
        unet = U()
    pnp.register_time(P, 481)
    u = P.unet
    for g in list(u.up_blocks) + list(u.down_blocks):
        assert all(r.t == 481 for r in g.resnets)
        for att in getattr(g, "attentions", []):
            assert att.transformer_blocks[0].attn1.t == 481 and att.transformer_blocks[0].attn2.t == 481
    assert u.mid_block.attentions[0].transformer_blocks[0].attn1.t == 481
    assert not hasattr(u.mid_block.resnets[0], "t")                     # the reference leaves the mid resnets alone


def test_generator_modes_follow_the_reference_rules():
    """utils.init_generator: "cpu" (default) forks the CPU RNG state whatever the device (the reference's CPU path, i.e.
    the oracle's stream); "device" is vidtome/utils.py:18-30 to the letter (CPU tensors -> CPU state, other device types
    -> the fallback generator or a CPU fork; the CUDA branch needs a GPU and is covered by the -m gpu tests)."""
    import torch
    from vidtome_amd import utils
    torch.manual_seed(123)
    want = [int(torch.randint(0, 4, (1,), generator=torch.Generator().manual_seed(123))) for _ in range(1)]
    g = utils.init_generator(torch.device("cpu"))
    assert g.device.type == "cpu" and int(torch.randint(0, 4, (1,), generator=g)) == want[0] == 2      # SURVEY KAT
    g2 = utils.init_generator(torch.device("cpu"), mode="device")
    assert torch.equal(g2.get_state(), torch.get_rng_state())
    fb = torch.Generator().manual_seed(7)
    assert utils.init_generator(torch.device("meta"), fallback=fb, mode="device") is fb
    assert utils.init_generator(torch.device("meta"), mode="device").device.type == "cpu"
    assert utils.init_generator(torch.device("meta"), fallback=fb).device.type == "cpu"               # default: CPU fork
    with pytest.raises(ValueError):
        utils.init_generator(torch.device("cpu"), mode="gpu")
    import vidtome_amd
    from standin import StandInUNet
    with pytest.raises(ValueError):
        vidtome_amd.apply_patch(StandInUNet(16, 2), generator_device="cuda")
    u = vidtome_amd.apply_patch(StandInUNet(16, 2), generator_device="device")
    assert u._tome_info["args"]["generator_device"] == "device"
    vidtome_amd.remove_patch(u)


def test_match_planner_state_machine_without_a_gpu():
    """merge.MatchPlanner's decisions as a pure function of the counters it is shown (the GPU tests drive it with real calls):
    plan on while < HIGH of the level lies inside the spans; the one-step scout is tried below LOW and dropped (for COOL
    calls) when ITS spans are wide -- without giving the plan up; wide spans of the deep scout -> one launch for COOL calls,
    position-ordered only at levels built with order_alone, and not even there when the first one-launch call reports that
    nothing dies in the filter either."""
    import torch
    from vidtome_amd import _lib, merge

    def planner(order_alone):
        pl = object.__new__(merge.MatchPlanner)        # (the constructor pins its buffer: needs a GPU)
        pl.mode, pl.buf = _lib.MATCH_SCOUT_RANGE, torch.zeros(8, dtype=torch.int32)
        pl.view, pl.cool, pl.switches = pl.buf.numpy(), 0, 0
        pl.probe_buf = torch.zeros(8, dtype=torch.int32)
        pl.probe_view = pl.probe_buf.numpy()
        pl.order_alone, pl.order_off, pl.order_probe = order_alone, False, False
        pl.shallow, pl.shallow_ban, pl.issued_shallow = False, 0, False
        return pl

    def show(pl, tested, alive, spans):
        pl.view[4], pl.view[5], pl.view[7] = tested, alive, spans

    def show_probe(pl, tested, alive):         # the one-launch call's own counters arrive in the OTHER buffer
        pl.probe_view[4], pl.probe_view[5] = tested, alive

    R, O = _lib.MATCH_SCOUT_RANGE, _lib.MATCH_ONE_LAUNCH
    pl = planner(True)
    assert pl.next()[0::2] == (R, True) and pl.next()[3] == 0          # nothing has arrived yet: deep scout, ordered
    show(pl, 1000, 10, 80)                                             # 0.08: inside the plan, but above LOW
    assert pl.next()[3] == 0
    show(pl, 1000, 10, 20)                                             # 0.02 < LOW: try the one-step scout
    assert pl.next()[0::3] == (R, 1)
    show(pl, 1000, 10, 25)
    assert pl.next()[0::3] == (R, 1)                                   # its spans are short too: keep it
    show(pl, 1000, 900, 700)                                           # ... now they are wide: back to the deep scout, plan kept
    assert pl.next()[0::3] == (R, 0) and pl.shallow_ban > 0
    show(pl, 1000, 10, 20)
    assert pl.next()[0::3] == (R, 0)                                   # (banned for COOL calls)
    show(pl, 1000, 300, 500)                                           # the DEEP scout's spans are wide: one launch
    mode, buf, keep, scout = pl.next()
    assert (mode, keep, scout) == (O, True, 0) and buf is not None     # ordered (order_alone), asks the filter for its counters
    show(pl, 1000, 1000, 1000)                                         # a LATE copy of an earlier scout call: must not be read as the probe
    assert pl.order_probe and pl.next()[0::2] == (O, True) and pl.order_probe
    show_probe(pl, 1000, 290)                                          # 29 % alive in the one-launch filter: the ordering stays
    mode, buf, keep, _ = pl.next()
    assert (mode, keep) == (O, True) and buf is None
    for _ in range(merge.MatchPlanner.COOL - 3):
        assert pl.next()[0] == O
    assert pl.next()[0] == R                                           # ... and tries the plan again
    # nothing dies in the filter either (uncorrelated tokens): the ordering goes too; level 2 never orders without the plan
    pl = planner(True)
    pl.next()
    show(pl, 1000, 1000, 1000)
    assert pl.next()[0::2] == (O, True)
    show_probe(pl, 1000, 1000)
    assert pl.next()[0::2] == (O, False)
    pl = planner(False)
    pl.next()
    show(pl, 1000, 300, 500)
    mode, buf, keep, _ = pl.next()
    assert (mode, keep) == (O, False) and buf is None


def test_position_order_size_helpers():
    """vtm_position_order's host-side size helpers (no GPU): the counter block holds two counters per (sample, operand,
    position + the no-position bucket), the scratch the offsets and the staging copies of both lists."""
    from vidtome_amd import _lib
    L = _lib.lib()
    B, Ns, Nd, N = 2, 34816, 34816, 4096
    assert L.vtm_position_order_counter_ints(B, N) == 2 * B * 2 * (N + 1)
    assert L.vtm_position_order_ws_bytes(B, Ns, Nd, N) == 4 * (B * 2 * (N + 2) + 2 * B * (Ns + Nd))
    assert L.vtm_position_order_counter_ints(0, N) == 0 and L.vtm_position_order_ws_bytes(B, 0, Nd, N) == 0
    assert _lib.POSITION_ORDER_MAX_N == 16360
    from vidtome_amd import merge
    assert merge.order_level(34816, 34816, 4096, False) and merge.order_level(34816, 34816, 4096, True)
    assert not merge.order_level(3072, 7168, 1024, False)            # a mid level 2: below the pair threshold
    assert not merge.order_level(34816, 34816, 16384, False)         # 1024 x 1024 images: the offsets would not fit the LDS
