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


class CrossAttention(Attention):
    """attn2 of an SD block: queries from the tokens, keys / values from the conditioning (fullblock16_*.npz)."""

    def __init__(self, C, heads, cond_dim):
        super().__init__(C, heads)
        self.to_k = torch.nn.Linear(cond_dim, C, bias=False)
        self.to_v = torch.nn.Linear(cond_dim, C, bias=False)


class GEGLU(torch.nn.Module):
    def __init__(self, C, D):
        super().__init__()
        self.proj = torch.nn.Linear(C, 2 * D)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * torch.nn.functional.gelu(gate)


class FeedForward(torch.nn.Module):
    def __init__(self, C):
        super().__init__()
        self.net = torch.nn.ModuleList([GEGLU(C, 4 * C), torch.nn.Dropout(0.0), torch.nn.Linear(4 * C, C)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(torch.nn.Module):
    def __init__(self, C, heads, full=False, cond_dim=0):
        super().__init__()
        self.norm1 = torch.nn.LayerNorm(C)
        self.attn1 = Attention(C, heads)
        self.only_cross_attention = False
        if full:        # the whole SD block (vidtome/patch.py:171-199), as in tests/golden/make_golden_fullblock.py
            self.norm2 = torch.nn.LayerNorm(C)
            self.attn2 = CrossAttention(C, heads, cond_dim)
            self.norm3 = torch.nn.LayerNorm(C)
            self.ff = FeedForward(C)
        else:
            self.attn2 = None
            self.norm2 = None
            self.norm3 = torch.nn.Identity()
            self.ff = Zero()


class _Attn2D(torch.nn.Module):
    """Stand-in Transformer2DModel: calls its block with the FULL keyword set of the patched forward (patch.py:128-137)."""

    def __init__(self, C, heads, full=False, cond_dim=0):
        super().__init__()
        self.transformer_blocks = torch.nn.ModuleList([BasicTransformerBlock(C, heads, full, cond_dim)])

    def forward(self, hidden_states, encoder_hidden_states=None, timestep=None, class_labels=None,
                cross_attention_kwargs=None, attention_mask=None, encoder_attention_mask=None):
        for block in self.transformer_blocks:
            hidden_states = block(hidden_states, attention_mask=attention_mask, encoder_hidden_states=encoder_hidden_states,
                                  encoder_attention_mask=encoder_attention_mask, timestep=timestep,
                                  cross_attention_kwargs=cross_attention_kwargs, class_labels=class_labels)
        return hidden_states


class _UpBlock(torch.nn.Module):
    def __init__(self, C, heads, n, ds, full=False, cond_dim=0):
        super().__init__()
        self.ds = ds
        if n:
            self.attentions = torch.nn.ModuleList([_Attn2D(C, heads, full, cond_dim) for _ in range(n)])


class ModelMixin(torch.nn.Module):
    pass


class StandInUNet(ModelMixin):
    """forward(latent, hiddens): feeds hiddens[i] to block i; the latent only provides (H, W) to the size hook."""

    def __init__(self, C, heads, full=False, cond_dim=0):
        super().__init__()
        self.full = full
        self.up_blocks = torch.nn.ModuleList([_UpBlock(C, heads, 0, 8), _UpBlock(C, heads, 3, 4, full, cond_dim),
                                              _UpBlock(C, heads, 3, 2, full, cond_dim), _UpBlock(C, heads, 3, 1, full, cond_dim)])

    def blocks(self):
        for ub in self.up_blocks:
            if hasattr(ub, "attentions"):
                for a in ub.attentions:
                    yield a.transformer_blocks[0]

    def transformers(self):
        for ub in self.up_blocks:
            if hasattr(ub, "attentions"):
                for a in ub.attentions:
                    yield a

    def forward(self, latent, hiddens, encoder_hidden_states=None, timestep=None):
        """hiddens[i] feeds block i (None = skip that block).  The full stand-in goes through its Transformer2DModel
        stand-ins, i.e. the blocks are called with every keyword of patch.py:128-137."""
        if self.full:
            return [None if h is None else t2d(h, encoder_hidden_states=encoder_hidden_states, timestep=timestep)
                    for t2d, h in zip(self.transformers(), hiddens)]
        return [blk(h) for blk, h in zip(self.blocks(), hiddens)]


class Pipe:
    def __init__(self, unet):
        self.unet = unet


def load_block_weights(unet, z, device, dtype, portable=None):
    """Copy the fixture's weights (saved from the reference run's stand-in) into the blocks.  ``portable`` (the full-block
    fixtures): a function (name, shape) -> matrix for the 2-D parameters the fixture does not store
    (tests/golden/inputs.portable_weight), and blocks the fixture keeps no parameters of stay as constructed."""
    sd = {}
    for k in z.files:
        if k.startswith("w/up_blocks"):
            sd[k[2:]] = torch.from_numpy(z[k])
    own = unet.state_dict()
    for k in own:
        if k in sd:
            own[k] = sd[k]
        elif portable is not None:
            prefix = k.rsplit(".transformer_blocks.0.", 1)[0] + ".transformer_blocks.0."
            if own[k].ndim == 2 and any(n.startswith(prefix) for n in sd):
                own[k] = torch.from_numpy(portable(k, tuple(own[k].shape)))
        else:
            own[k] = sd[k]               # KeyError: the fixture must hold every parameter
    unet.load_state_dict(own)
    return unet.to(device=device, dtype=dtype)
