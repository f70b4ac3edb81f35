"""Reference-free, deterministic input builders shared by make_golden.py and the tests."""
import numpy as np


def planted_inputs(Ns, Nd, C, seed, B=1):
    """Each src row = c_i * dst_pi(i) + s_i * e_i_perp with strictly spaced c_i:
    argmax and rank order are separated by far more than fp32 rounding.  Deterministic numpy code shared by
    make_golden.py and the tests."""
    rng = np.random.default_rng(seed)
    out_a, out_b = [], []
    for b in range(B):
        dst = rng.standard_normal((Nd, C))
        dst /= np.linalg.norm(dst, axis=1, keepdims=True)
        pi = rng.integers(0, Nd, size=Ns)
        c = np.linspace(0.99, 0.55, Ns)
        rng.shuffle(c)
        e = rng.standard_normal((Ns, C))
        d = dst[pi]
        e -= (e * d).sum(1, keepdims=True) * d
        e /= np.linalg.norm(e, axis=1, keepdims=True)
        src = c[:, None] * d + np.sqrt(1 - c * c)[:, None] * e
        src *= rng.uniform(0.5, 2.0, size=(Ns, 1))
        dscale = rng.uniform(0.5, 2.0, size=(Nd, 1))
        out_a.append(src.astype(np.float32))
        out_b.append((dst * dscale).astype(np.float32))
    return np.stack(out_a), np.stack(out_b)


def portable_weight(name, shape):
    """A weight matrix as a pure function of its parameter NAME and shape, in integer arithmetic only (splitmix64 over the
    element index, top 12 bits -> k / 2048 in [-1, 1), times a power of two near sqrt(3 / fan_in)): the same bits on any
    machine and library version, exactly representable in fp16, variance ~1 / fan_in.  The full-block fixtures
    (make_golden_fullblock.py) draw every 2-D block weight from here instead of storing 20+ MB of them."""
    import zlib
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) + np.uint64(zlib.crc32(name.encode())) * np.uint64(0x9E3779B97F4A7C15)
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    k = (z >> np.uint64(52)).astype(np.int64) - 2048                        # 12 bits -> [-2048, 2047]
    scale = 2.0 ** np.round(np.log2(np.sqrt(3.0 / shape[-1])))
    return (k.astype(np.float32) / 2048.0 * np.float32(scale)).reshape(shape)


def fuzz_inputs(cfg):
    """Chunk inputs of a fuzz_compute_merge.npz configuration: (B*F, N, C) fp32 per chunk, nothing but torch.randn on a
    seeded CPU generator (shared by make_golden_fuzz.py and the tests)."""
    import torch
    g = torch.Generator().manual_seed(int(cfg["data_seed"]))
    N = (int(cfg["H"]) // int(cfg["ds"])) * (int(cfg["W"]) // int(cfg["ds"]))
    return [torch.randn(int(cfg["B"]) * int(F), N, int(cfg["C"]), generator=g) for F in cfg["frames"]]


def fuzz_hash(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(np.asarray(a).astype(np.float32)).tobytes()).hexdigest()


def load_fuzz_configs(path):
    z = np.load(path)
    keys = ["B", "H", "W", "ds", "C", "align", "merge_global", "global_ratio", "local_ratio", "global_rand", "gen_seed",
            "data_seed"]
    cfgs = []
    for i in range(len(z["B"])):
        c = {k: z[k][i].item() for k in keys}
        c["frames"] = [int(f) for f in z["frames"][i]]
        c["hashes"] = z["hashes"][i]
        cfgs.append(c)
    return cfgs



def planted_batch(Ns, Nd, C, seed, B=2, lo=0.55, hi=0.99):
    """`planted_inputs` for a batch whose samples may be matched TOGETHER (align_batch concatenates the samples' scores
    along the dst axis, merge.py:93-97): sample b draws its planted cosines from the grid
    linspace(hi, lo, Ns) - b * step / B, so that all B * Ns cosines -- hence the per-row maxima over the samples and the
    sorted order of those maxima -- are strictly spaced (by step / B) whichever sample wins a row.  Returns
    src (B, Ns, C), dst (B, Nd, C) fp32 with arbitrary row norms."""
    rng = np.random.default_rng(seed)
    step = (hi - lo) / max(Ns - 1, 1)
    out_a, out_b = [], []
    for b in range(B):
        dst = rng.standard_normal((Nd, C))
        dst /= np.linalg.norm(dst, axis=1, keepdims=True)
        pi = rng.integers(0, Nd, size=Ns)
        c = np.linspace(hi, lo, Ns) - b * step / B
        rng.shuffle(c)
        e = rng.standard_normal((Ns, C))
        d = dst[pi]
        e -= (e * d).sum(1, keepdims=True) * d
        e /= np.linalg.norm(e, axis=1, keepdims=True)
        src = c[:, None] * d + np.sqrt(1 - c * c)[:, None] * e
        src *= rng.uniform(0.5, 2.0, size=(Ns, 1))
        out_a.append(src.astype(np.float32))
        out_b.append((dst * rng.uniform(0.5, 2.0, size=(Nd, 1))).astype(np.float32))
    return np.stack(out_a), np.stack(out_b)


def planted_local_chunk(B, F, tnum, unm_pre, C, randf, seed, target_stride=4):
    """A joined chunk (B, unm_pre + F * tnum, C) for bipartite_soft_matching_randframe whose src / dst partition under
    `randf` (merge.py:56-69: dst = frames with f % ts == randf, plus the first unm_pre tokens) is a planted matching."""
    ts = min(target_stride, F)
    frames = np.arange(F * tnum) // tnum
    is_dst = np.concatenate([np.ones(unm_pre, bool), frames % ts == randf])
    Nd, Ns = int(is_dst.sum()), int((~is_dst).sum())
    a, b = planted_batch(Ns, Nd, C, seed, B)
    x = np.empty((B, unm_pre + F * tnum, C), np.float32)
    x[:, is_dst] = b
    x[:, ~is_dst] = a
    return x


def idx_sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(np.asarray(a).astype(np.int32)).tobytes()).hexdigest()
