"""Stand-in Diffusers classes for the tests (Diffusers itself is not installed here or on the GPU box).

apply_patch only looks at class NAMES (vidtome/utils.py:12-14, patch.py:279-280,319), so a root class named
``ModelMixin`` whose sub-modules are named ``BasicTransformerBlock`` is what the reference patches too; the
golden chain fixtures were produced by the reference on exactly this structure
(tests/golden/make_golden.py: up_blocks[1..3] x 3 blocks at downsample 4 / 2 / 1)."""
import torch


class Attention(torch.nn.Module):
    def __init__(self, C, heads):
        super().__init__()
        self.heads = heads
        self.scale = (C // heads) ** -0.5
        self.to_q = torch.nn.Linear(C, C, bias=False)
        self.to_k = torch.nn.Linear(C, C, bias=False)
        self.to_v = torch.nn.Linear(C, C, bias=False)
        self.to_out = torch.nn.ModuleList([torch.nn.Linear(C, C), torch.nn.Dropout(0.0)])

    def forward(self, x, encoder_hidden_states=None, attention_mask=None, **kw):
        raise RuntimeError("stand-in Attention.forward must not be called: attn1 runs in vidtome_amd")


class Zero(torch.nn.Module):
    def forward(self, x):
        return torch.zeros_like(x)


class BasicTransformerBlock(torch.nn.Module):
    def __init__(self, C, heads):
        super().__init__()
        self.norm1 = torch.nn.LayerNorm(C)
        self.attn1 = Attention(C, heads)
        self.attn2 = None
        self.norm2 = None
        self.norm3 = torch.nn.Identity()
        self.ff = Zero()
        self.only_cross_attention = False


class _Attn2D(torch.nn.Module):
    def __init__(self, C, heads):
        super().__init__()
        self.transformer_blocks = torch.nn.ModuleList([BasicTransformerBlock(C, heads)])


class _UpBlock(torch.nn.Module):
    def __init__(self, C, heads, n, ds):
        super().__init__()
        self.ds = ds
        if n:
            self.attentions = torch.nn.ModuleList([_Attn2D(C, heads) for _ in range(n)])


class ModelMixin(torch.nn.Module):
    pass


class StandInUNet(ModelMixin):
    """forward(latent, hiddens): feeds hiddens[i] to block i; the latent only provides (H, W) to the size hook."""

    def __init__(self, C, heads):
        super().__init__()
        self.up_blocks = torch.nn.ModuleList([_UpBlock(C, heads, 0, 8), _UpBlock(C, heads, 3, 4),
                                              _UpBlock(C, heads, 3, 2), _UpBlock(C, heads, 3, 1)])

    def blocks(self):
        for ub in self.up_blocks:
            if hasattr(ub, "attentions"):
                for a in ub.attentions:
                    yield a.transformer_blocks[0]

    def forward(self, latent, hiddens):
        return [blk(h) for blk, h in zip(self.blocks(), hiddens)]


class Pipe:
    def __init__(self, unet):
        self.unet = unet


def load_block_weights(unet, z, device, dtype):
    """Copy the fixture's weights (saved from the reference run's stand-in) into the blocks."""
    sd = {}
    for k in z.files:
        if k.startswith("w/up_blocks"):
            sd[k[2:]] = torch.from_numpy(z[k])
    own = unet.state_dict()
    for k in own:
        own[k] = sd[k]
    unet.load_state_dict(own)
    return unet.to(device=device, dtype=dtype)
