"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
(1) the CPU oracle on seeded inputs -- BITWISE for the matching / index path, 1e-3 for attention;
(2) the golden vectors generated from the reference itself (tests/golden);
(3) size-independent properties at BASELINE.json's full sizes."""
import hashlib

import numpy as np
import pytest
import torch

from helpers import forked_generator_from_state, load_cases, load_chain
from inputs import planted_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda"
IDX = ("a_idx", "b_idx", "unm_idx", "src_idx", "dst_idx")
# whole-block outputs of an fp16 model against the reference's fp32 run, relative to the output scale (the self-attention
# segment alone: north_star's 1e-3); see test_full_block_vs_reference_chain
FULL_BLOCK_TOL = 2e-3


@pytest.fixture(scope="module")
def L():
    from vidtome_amd import _lib
    _lib.lib()
    return _lib


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def _deinterleave(op, n, C):
    """(B, G, 2, n_pad, 4) k-panel operand [b][g][kh][row][e] (channel 8g + 2e + kh) -> (B, n, C) plain."""
    B, G, _, n_pad, _ = op.shape
    g = op.permute(0, 3, 1, 4, 2).reshape(B, n_pad, G * 8)
    return g[:, :n, :C]


def _bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


# ---------------------------------------------------------------------------------------------------
# primitives vs oracle
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_normalize_gather_bitwise(L, oracle, dtype):
    g = torch.Generator().manual_seed(0)
    B, P0, P1, C, n = 2, 300, 77, 40, 211
    x0 = torch.randn(B, P0, C, generator=g).to(dtype)
    x1 = torch.randn(B, P1, C, generator=g).to(dtype)
    x0[1, 5] = 0          # zero row -> NaN
    rows = torch.randint(0, P0 + P1, (B, n), generator=g, dtype=torch.int32)
    rows[1, 3] = 5
    op, norms = L.normalize_gather(x0.to(DEV), x1.to(DEV), rows.to(DEV))
    pool = torch.cat([x0, x1], 1).float().numpy()
    ref = oracle.normalize_gather(pool, rows.numpy())
    got = _deinterleave(op, n, C).cpu().numpy()
    assert np.array_equal(_bits(got), _bits(ref))
    assert torch.count_nonzero(op[:, :, :, n:]) == 0 and torch.count_nonzero(op[:, C // 8:]) == 0


@pytest.mark.parametrize("shape", [(2, 37, 29, 24), (3, 300, 513, 40), (1, 129, 128, 320), (2, 1000, 700, 64)])
@pytest.mark.parametrize("align", [False, True])
def test_match_bitwise(L, oracle, shape, align):
    B, Ns, Nd, C = shape
    rng = np.random.default_rng(Ns * 7 + Nd)
    x = rng.standard_normal((B, Ns + Nd, C)).astype(np.float32)
    x[:, Ns + Nd // 2] = x[:, Ns + 1]                 # duplicate dst rows -> exact ties, first index wins
    ra = np.broadcast_to(np.arange(Ns, dtype=np.int32), (B, Ns)).copy()
    rb = np.broadcast_to(np.arange(Ns, Ns + Nd, dtype=np.int32), (B, Nd)).copy()
    a_op, _ = L.normalize_gather(_t(x), None, _t(ra))
    b_op, _ = L.normalize_gather(_t(x), None, _t(rb))
    best = L.match(a_op, b_op, Ns, Nd, align)
    nm, ni = L.decode_best(best)
    a = oracle.normalize_gather(x, ra)
    b = oracle.normalize_gather(x, rb)
    rm, ri = oracle.match(a, b, align)
    rm = rm + np.float32(0.0)
    assert np.array_equal(ni.cpu().numpy().reshape(ri.shape), ri)
    assert np.array_equal(_bits(nm.cpu().numpy().reshape(rm.shape)), _bits(rm))


def test_match_nan_semantics(L, oracle):
    rng = np.random.default_rng(3)
    B, Ns, Nd, C = 2, 70, 150, 16
    x = rng.standard_normal((B, Ns + Nd, C)).astype(np.float32)
    x[0, 5] = 0            # NaN src row -> (NaN, 0)
    x[1, Ns + 40] = 0      # NaN dst row -> every src row of sample 1 -> (NaN, 40)
    x[1, Ns + 90] = 0
    ra = np.broadcast_to(np.arange(Ns, dtype=np.int32), (B, Ns)).copy()
    rb = np.broadcast_to(np.arange(Ns, Ns + Nd, dtype=np.int32), (B, Nd)).copy()
    a_op, _ = L.normalize_gather(_t(x), None, _t(ra))
    b_op, _ = L.normalize_gather(_t(x), None, _t(rb))
    for align in (False, True):
        nm, ni = L.decode_best(L.match(a_op, b_op, Ns, Nd, align))
        rm, ri = oracle.match(oracle.normalize_gather(x, ra), oracle.normalize_gather(x, rb), align)
        assert np.array_equal(ni.cpu().numpy().reshape(ri.shape), ri)
        assert np.array_equal(np.isnan(nm.cpu().numpy().reshape(rm.shape)), np.isnan(rm))
        perm = L.sort_desc(L.match(a_op, b_op, Ns, Nd, align)).cpu().numpy()
        assert np.array_equal(perm.reshape(rm.shape), oracle.sort_desc(rm))


def _filtered_vs_exact(L, x, Ns, Nd, align, expect_flag=None):
    """vtm_match_filtered (fp16 filter + fp32 refine) must equal vtm_match (plain fp32 MFMA) bit for bit."""
    B = x.shape[0]
    ra = torch.arange(Ns, dtype=torch.int32, device=DEV).expand(B, Ns).contiguous()
    rb = torch.arange(Ns, Ns + Nd, dtype=torch.int32, device=DEV).expand(B, Nd).contiguous()
    a_op, _ = L.normalize_gather(x, None, ra)
    b_op, _ = L.normalize_gather(x, None, rb)
    exact = L.match(a_op, b_op, Ns, Nd, align)
    got, flag = L.match_filtered(x, None, ra, rb, align, want_flag=True)
    assert torch.equal(got, exact)
    if expect_flag is not None:
        assert int(flag[0].item()) == expect_flag, flag.tolist()
    return int(flag[0].item())


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32, torch.bfloat16])
@pytest.mark.parametrize("align", [False, True])
def test_match_filtered_equals_exact(L, dtype, align):
    g = torch.Generator().manual_seed(11)
    for (B, Ns, Nd, C) in [(2, 37, 29, 24), (3, 300, 513, 40), (2, 1000, 700, 64), (2, 2048, 1500, 320), (1, 700, 3000, 640)]:
        x = torch.randn(B, Ns + Nd, C, generator=g).to(dtype).to(DEV)
        _filtered_vs_exact(L, x, Ns, Nd, align, expect_flag=0)


def test_match_filtered_run_to_run(L):
    """The dst splits of a row share their running maximum WHILE they run (atomicMax / re-read every step), so the candidate
    sets depend on timing; the result must not.  Same call 20 times, thousands of near-ties inside the window, vs the exact matcher
    (tools/stress_match.py runs the cfg-2 shapes 100 times each)."""
    g = torch.Generator(device=DEV).manual_seed(3)
    for (B, Ns, Nd, C, noise) in [(2, 12288, 4096, 320, 1e-3), (2, 6000, 9000, 64, 0.1)]:
        base = torch.randn(B, Nd, C, generator=g, device=DEV)
        idx = torch.arange(Ns + Nd, device=DEV) % Nd
        x = (base[:, idx] + noise * torch.randn(B, Ns + Nd, C, generator=g, device=DEV)).half()
        ra = torch.arange(Ns, dtype=torch.int32, device=DEV).expand(B, Ns).contiguous()
        rb = torch.arange(Ns, Ns + Nd, dtype=torch.int32, device=DEV).expand(B, Nd).contiguous()
        a_op, _ = L.normalize_gather(x, None, ra)
        b_op, _ = L.normalize_gather(x, None, rb)
        exact = L.match(a_op, b_op, Ns, Nd, False)
        for _ in range(20):
            assert torch.equal(L.match_filtered(x, None, ra, rb, False), exact)


def test_match_filtered_fuzz(L):
    """Random shapes / dtypes / data regimes (tools/fuzz_match.py; 1 800 cases were run while developing)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "fuzz_match", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_match.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(80, seed=3, verbose=False) == 0


def test_match_filtered_hard_cases(L):
    g = torch.Generator().manual_seed(12)
    B, Ns, Nd, C = 2, 600, 900, 320
    # (1) frame-correlated tokens: many near-maximal scores per row
    base = torch.randn(B, 1, C, generator=g)
    x = (base + 0.05 * torch.randn(B, Ns + Nd, C, generator=g)).half().to(DEV)
    _filtered_vs_exact(L, x, Ns, Nd, False)
    _filtered_vs_exact(L, x, Ns, Nd, True)
    # (2) exact duplicates among dst rows (anchors hold copies, patch.py:80): first index must win
    x = torch.randn(B, Ns + Nd, C, generator=g).half()
    x[:, Ns + 100:Ns + 120] = x[:, Ns + 7:Ns + 8]
    x[:, 5] = x[:, Ns + 7]                     # a src row identical to the duplicated dst row -> score ~1, 21-way tie
    _filtered_vs_exact(L, x.to(DEV), Ns, Nd, False, expect_flag=0)
    # (3) near ties far below fp16 resolution: dst rows differing by one fp32 ulp-scale perturbation
    xf = torch.randn(B, Ns + Nd, C, generator=g)
    xf[:, Ns + 1] = xf[:, Ns] * (1 + 1e-7) + 1e-7 * torch.randn(B, C, generator=g)
    xf[:, Ns + 2] = xf[:, Ns] + 3e-7 * torch.randn(B, C, generator=g)
    xf[:, 0] = xf[:, Ns] + 1e-3 * torch.randn(B, C, generator=g)
    _filtered_vs_exact(L, xf.to(DEV), Ns, Nd, False, expect_flag=0)
    # (4) zero token among the DST rows -> NaN column for every src row -> device flag -> every row recomputed by
    #     exact_rows_kernel; a zero SRC token alone only sends its own row there (no whole-call escape)
    x = torch.randn(B, Ns + Nd, C, generator=g).half()
    x[0, 3] = 0
    x[1, Ns + 50] = 0
    _filtered_vs_exact(L, x.to(DEV), Ns, Nd, False, expect_flag=1)
    _filtered_vs_exact(L, x.to(DEV), Ns, Nd, True, expect_flag=1)
    x = torch.randn(B, Ns + Nd, C, generator=g).half()
    x[0, 3] = 0
    x[1, 77] = 0
    for align in (False, True):
        ra_ = torch.arange(Ns, dtype=torch.int32, device=DEV).expand(B, Ns).contiguous()
        rb_ = torch.arange(Ns, Ns + Nd, dtype=torch.int32, device=DEV).expand(B, Nd).contiguous()
        _filtered_vs_exact(L, x.to(DEV), Ns, Nd, align, expect_flag=0)
        _, fl = L.match_filtered(x.to(DEV), None, ra_, rb_, align, want_flag=True)
        assert fl.tolist()[:3] == [0, 1, 2], fl.tolist()          # not whole-call, "special rows seen", two rows listed
    # (4b) fp32 tokens with norms far outside the range where the filter's reciprocal-multiply operands are trustworthy
    #      (1e-38 .. 1e-33 and 1e+33): same flag, same exact result; moderate scales (1e-20, 1e+15) stay on the fast path
    xf = torch.randn(B, Ns + Nd, C, generator=g)
    xs = xf.clone()
    xs[0, 7] *= 1e-20
    xs[1, Ns + 3] *= 1e15
    _filtered_vs_exact(L, xs.to(DEV), Ns, Nd, False, expect_flag=0)
    xs = xf.clone()
    xs[0, 7] *= 1e-36                                     # a SRC row: only that row takes the escape
    _filtered_vs_exact(L, xs.to(DEV), Ns, Nd, False, expect_flag=0)
    xs = xf.clone()
    xs[0, Ns + 7] *= 1e-36                                # a DST row: the whole call
    _filtered_vs_exact(L, xs.to(DEV), Ns, Nd, False, expect_flag=1)
    xs = xf.clone()
    xs[1, Ns + 3] *= 3e31
    _filtered_vs_exact(L, xs.to(DEV), Ns, Nd, True, expect_flag=1)
    # (4c) fp32 / bf16 rows with components 25-40 orders of magnitude below the rest: the refine pass divides by a per-row
    #      reciprocal only where that is bit-identical to x / norm (|x| >= 2^-102 and x / norm >= 2^-124) and falls back to the
    #      IEEE division element group by element group elsewhere
    for dt in (torch.float32, torch.bfloat16):
        xs = torch.randn(B, Ns + Nd, C, generator=g)
        tiny = torch.tensor([1e-25, 1e-30, 1e-36, 3e-38, 1e-40, 0.0])
        for r in range(0, Ns + Nd, 7):
            xs[:, r, (r * 5) % C] = tiny[r % 6]
            xs[:, r, (r * 11 + 3) % C] = -tiny[(r + 2) % 6]
        xs[0, 11] *= 1e-12
        xs[1, Ns + 9] *= 1e10
        _filtered_vs_exact(L, xs.to(dt).to(DEV), Ns, Nd, False, expect_flag=0)
        _filtered_vs_exact(L, xs.to(dt).to(DEV), Ns, Nd, True, expect_flag=0)
    # (5) candidate overflow: every dst row identical -> more than 64 candidates per row -> those rows are
    #     recomputed by exact_rows_kernel (no whole-call fallback)
    x = torch.randn(B, Ns + Nd, C, generator=g).half()
    x[:, Ns:] = x[:, Ns:Ns + 1]
    for align in (False, True):
        ra = torch.arange(Ns, dtype=torch.int32, device=DEV).expand(B, Ns).contiguous()
        rb = torch.arange(Ns, Ns + Nd, dtype=torch.int32, device=DEV).expand(B, Nd).contiguous()
        _filtered_vs_exact(L, x.to(DEV), Ns, Nd, align, expect_flag=0)
        _, flag = L.match_filtered(x.to(DEV), None, ra, rb, align, want_flag=True)
        assert int(flag[2].item()) == (Ns if align else B * Ns)


def _regime_tokens(regime, B, Ns, Nd, C, N, seed=0):
    """(B, Ns + Nd, C) fp16 tokens of a data regime (sites.DATA_REGIMES) after a LayerNorm -- what the matcher sees."""
    from vidtome_amd import sites
    g = torch.Generator().manual_seed(seed)
    x = sites.regime_tokens(regime, B, (Ns + Nd) // N, N, C, g)
    return torch.nn.functional.layer_norm(x, (C,)).reshape(B, Ns + Nd, C).half()


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32, torch.bfloat16])
@pytest.mark.parametrize("align", [False, True])
def test_match_filtered_escape_rows_equal_exact(L, dtype, align):
    """The escape (exact_rows_kernel: fp32-MFMA tiles over the rows whose candidate list overflowed, operands normalised on
    the fly) against the exact matcher, bit for bit: flat regions (a quarter of the positions of every frame hold one
    content vector + 2 % noise -> hundreds of dst rows inside each such row's window), mixed with ordinary rows in the same
    call, ragged sizes (rows past the lists / past Nd inside the tiles), batch 3 (three lists; aligned: one list, every
    sample's dst set), C not a multiple of the 32-channel step, fp32 / bf16 tokens with tiny components (IEEE-division
    fallback inside the per-row reciprocal form)."""
    from vidtome_amd import sites
    g = torch.Generator().manual_seed(21)
    for (B, F, N, C, fs) in [(2, 8, 256, 320, 6), (3, 5, 200, 40, 3), (2, 6, 333, 72, 4), (1, 4, 640, 640, 3)]:
        x = sites.regime_tokens("flat25", B, F, N, C, g)
        x = torch.nn.functional.layer_norm(x, (C,)).reshape(B, F * N, C)
        if dtype != torch.float16:
            x[:, ::7, 3] = 1e-30
            x[:, 5::11, C - 1] = 0.0
        x = x.to(dtype).to(DEV)
        Ns, Nd = fs * N, (F - fs) * N
        ra = torch.arange(Ns, dtype=torch.int32, device=DEV).expand(B, Ns).contiguous()
        rb = torch.arange(Ns, Ns + Nd, dtype=torch.int32, device=DEV).expand(B, Nd).contiguous()
        _filtered_vs_exact(L, x, Ns, Nd, align, expect_flag=0)
        _, flag = L.match_filtered(x, None, ra, rb, align, want_flag=True)
        rows = Ns if align else B * Ns
        assert 0 < int(flag[2].item()) < rows, flag.tolist()       # some rows escaped, not all
        # a zero dst token on top: the whole call goes through the escape (every row, every list), same bits
        x[B - 1, Ns + 7] = 0
        _filtered_vs_exact(L, x, Ns, Nd, align, expect_flag=1)


def test_match_filtered_seeds_never_change_the_result(L):
    """vtm_match_filtered_seeded starts every src row from the score of ONE guessed pair (the dst row at the same token
    position).  Any pair's score is a valid running maximum, so the packed result must equal the exact matcher's bit for
    bit WHATEVER the guesses are: the right ones (identity table on frame-ordered rows), a random table, positions for the x1
    rows, out-of-range and missing entries, aligned batches -- and with the right ones a flat-region input sends only its flat
    rows to the exact escape (the rows that merely MEET the flat region first no longer overflow)."""
    from vidtome_amd import sites
    g = torch.Generator().manual_seed(33)
    B, F, N, C, fs = 2, 8, 256, 320, 6
    x = sites.regime_tokens("flat25", B, F, N, C, g)
    x = torch.nn.functional.layer_norm(x, (C,)).reshape(B, F * N, C).half().to(DEV)
    Ns, Nd = fs * N, (F - fs) * N
    ra = torch.arange(Ns, dtype=torch.int32, device=DEV).expand(B, Ns).contiguous()
    rb = torch.arange(Ns, Ns + Nd, dtype=torch.int32, device=DEV).expand(B, Nd).contiguous()
    a_op, _ = L.normalize_gather(x, None, ra)
    b_op, _ = L.normalize_gather(x, None, rb)
    for align in (False, True):
        exact = L.match(a_op, b_op, Ns, Nd, align)
        plain, f0 = L.match_filtered(x, None, ra, rb, align, want_flag=True)
        assert torch.equal(plain, exact)
        # the right guesses: src row i is token (i // N, i % N), dst index p is the token at position p of the first dst frame
        seeded, f1 = L.match_filtered(x, None, ra, rb, align, want_flag=True, seed=(N, F * N, None, None))
        assert torch.equal(seeded, exact)
        if not align:
            assert f1[2].item() <= f0[2].item() and f1[2].item() <= 0.3 * B * Ns, (f0.tolist(), f1.tolist())
        # garbage guesses
        table = torch.randint(-3, Nd + 50, (B, N), generator=g, dtype=torch.int32).to(DEV)
        assert torch.equal(L.match_filtered(x, None, ra, rb, align, seed=(N, F * N, None, table)), exact)
        assert torch.equal(L.match_filtered(x, None, ra, rb, align, seed=(N, 100, None, table)), exact)       # most rows: no position
    # two-part pool: the dst rows live in x1 and carry positions
    x0, x1 = x[:, :Ns].contiguous(), x[:, Ns:].contiguous()
    rb1 = torch.arange(Ns, Ns + Nd, dtype=torch.int32, device=DEV).expand(B, Nd).contiguous()
    pos1 = (torch.arange(Nd, dtype=torch.int32, device=DEV) % N).expand(B, Nd).contiguous()
    table = torch.arange(N, dtype=torch.int32, device=DEV).expand(B, N).contiguous()
    exact = L.match(a_op, b_op, Ns, Nd, False)
    assert torch.equal(L.match_filtered(x0, x1, ra, rb1, False, seed=(N, Ns, pos1, table)), exact)
    # ... and the src rows in x1 (anchors as the src side): positions come from pos1
    ra1 = rb1
    rb0 = torch.arange(0, Ns, dtype=torch.int32, device=DEV).expand(B, Ns).contiguous()
    e2 = L.match(b_op, a_op, Nd, Ns, False)
    t2 = torch.arange(N, dtype=torch.int32, device=DEV).expand(B, N).contiguous()
    assert torch.equal(L.match_filtered(x0, x1, ra1, rb0, False, seed=(N, Ns, pos1, t2)), e2)


@pytest.mark.parametrize("align", [False, True])
def test_match_scout_range_plan_equals_exact(L, align):
    """The scout + range launch plan (round 5: a scout launch marks the 256 x 128 tile pairs whose partial sums can still
    reach a row's window, the filter launch proper shrinks every workgroup's dst range -- one dst frame per split -- to the span
    of the marked tiles; vtm_match_filtered_plan, VTM_MATCH_SCOUT_RANGE) must give the bits of the exact fp32 matcher in every
    data regime -- it is a launch plan, never a different result: frames of one clip at three noise levels, a drifting smooth
    field, exact duplicates, a flat region (lists overflow -> exact escape), UNCORRELATED tokens (every tile stays marked),
    garbage seeds, no seeds (one launch), ragged sizes (rows past Ns / Nd inside the tiles, a last partial dst frame), batch 3,
    C = 256 / 320 / 640, frames that are not whole tiles (one launch); a zero dst token sends the whole call to the escape."""
    from vidtome_amd import sites
    g = torch.Generator().manual_seed(77)
    cases = [("corr01", 2, 8, 256, 320, 6), ("corr002", 2, 8, 256, 320, 6), ("corr05", 2, 8, 256, 320, 6),
             ("smooth", 2, 8, 256, 320, 6), ("dup", 2, 8, 256, 320, 6), ("flat25", 2, 8, 256, 320, 6),
             ("n01", 2, 8, 256, 320, 6), ("corr01", 3, 5, 384, 256, 3), ("corr05", 2, 6, 512, 640, 4),
             ("corr01", 1, 7, 441, 320, 5), ("corr01", 2, 12, 1024, 320, 9)]
    for (regime, B, F, N, C, fs) in cases:
        x = sites.regime_tokens(regime, B, F, N, C, g)
        x = torch.nn.functional.layer_norm(x, (C,)).reshape(B, F * N, C).half().to(DEV)
        Ns, Nd = fs * N, (F - fs) * N
        ra = torch.arange(Ns, dtype=torch.int32, device=DEV).expand(B, Ns).contiguous()
        rb = torch.arange(Ns, Ns + Nd, dtype=torch.int32, device=DEV).expand(B, Nd).contiguous()
        a_op, _ = L.normalize_gather(x, None, ra)
        b_op, _ = L.normalize_gather(x, None, rb)
        exact = L.match(a_op, b_op, Ns, Nd, align)
        seed = (N, F * N, None, None)
        got, flag = L.match_filtered(x, None, ra, rb, align, want_flag=True, seed=seed, mode=L.MATCH_SCOUT_RANGE)
        assert torch.equal(got, exact), (regime, B, F, N, C)
        f = flag.tolist()
        if N % 128 == 0:
            assert f[4] > 0 and 0 < f[7] <= f[4] and f[5] <= f[4], (regime, f)           # the scout ran, the spans are not empty
            if regime in ("corr01", "corr002") and N >= 1024:
                assert f[7] < 0.3 * f[4], (regime, f)                                     # ... and short on a low-noise clip
            if regime == "n01":
                assert f[7] > 0.9 * f[4], (regime, f)                                     # ... the whole level on uncorrelated tokens
        else:
            assert f[7] == 0, (regime, f)                                                  # frames are not whole tiles: one launch
        # garbage seeds, and no seeds at all (then the plan is the one-launch one): exact whatever the starting maxima are
        table = torch.randint(-3, Nd + 50, (B, N), generator=g, dtype=torch.int32).to(DEV)
        assert torch.equal(L.match_filtered(x, None, ra, rb, align, seed=(N, F * N, None, table), mode=L.MATCH_SCOUT_RANGE), exact)
        assert torch.equal(L.match_filtered(x, None, ra, rb, align, mode=L.MATCH_SCOUT_RANGE), exact)
        assert torch.equal(L.match_filtered(x, None, ra, rb, align, seed=seed), exact)
        # a shallower scout (VTM_MATCH_SCOUT_STEPS: its own rest norms) marks more tiles, never fewer than needed
        for k in (1, 3, 200):
            got1, f1 = L.match_filtered(x, None, ra, rb, align, want_flag=True, seed=seed, mode=L.MATCH_SCOUT_RANGE, scout_steps=k)
            assert torch.equal(got1, exact), (regime, k)
            if k == 1 and N % 128 == 0:
                assert f1[7] >= f[7], (regime, f, f1.tolist())
        if regime == "corr05":
            x[B - 1, Ns + 7] = 0
            a_op, _ = L.normalize_gather(x, None, ra)
            b_op, _ = L.normalize_gather(x, None, rb)
            got, flag = L.match_filtered(x, None, ra, rb, align, want_flag=True, seed=seed, mode=L.MATCH_SCOUT_RANGE)
            assert torch.equal(got, L.match(a_op, b_op, Ns, Nd, align)) and int(flag[0]) == 1


def test_position_order_is_a_stable_sort_by_position(L):
    """vtm_position_order: both lists sorted by token position (chunk rows: row % N; x1 rows: pos1; rows without a position
    last), ties in original order, `order` = the original index of every sorted entry, `table` = position -> first sorted dst
    entry holding it, and the counter block all zero again afterwards."""
    g = torch.Generator().manual_seed(3)
    for (B, Ns, Nd, Lrows, N, P1) in [(2, 1000, 1500, 2048, 256, 700), (1, 37, 5, 64, 16, 0), (3, 4096, 4096, 8192, 1024, 3000),
                                       (2, 300, 200, 0, 7, 600)]:
        pool = Lrows + P1
        ra = torch.randint(0, pool, (B, Ns), generator=g, dtype=torch.int32).to(DEV)
        rb = torch.randint(0, pool, (B, Nd), generator=g, dtype=torch.int32).to(DEV)
        pos1 = torch.randint(-2, N + 3, (B, P1), generator=g, dtype=torch.int32).to(DEV) if P1 else None
        a_s, a_o, b_s, b_o, table = L.position_order(ra, rb, Lrows, N, pos1, Lrows)
        torch.cuda.synchronize()

        def pos_of(rows):
            r = rows.long()
            p = torch.where(r < Lrows, r % N, torch.full_like(r, N))
            if pos1 is not None:
                idx = (r - Lrows).clamp(0, P1 - 1)
                pp = torch.gather(pos1.long(), 1, idx)
                pp = torch.where((pp >= 0) & (pp < N), pp, torch.full_like(pp, N))
                p = torch.where(r >= Lrows, pp, p)
            return p
        for rows, srt, order in ((ra, a_s, a_o), (rb, b_s, b_o)):
            assert torch.equal(torch.gather(rows, 1, order.long()), srt)
            key = pos_of(rows) * rows.shape[1] + torch.arange(rows.shape[1], device=DEV)       # (position, original index)
            want = torch.argsort(key, dim=1).to(torch.int32)
            assert torch.equal(order, want)
        pb = pos_of(b_s)
        for b in range(B):
            t = table[b].tolist()
            pbl = pb[b].tolist()
            for p_ in range(N):
                first = pbl.index(p_) if p_ in pbl else -1
                assert t[p_] == first, (b, p_)
        key = next(k for k in L._ZEROED)
        assert int(L._ZEROED[key].abs().sum()) == 0


def test_match_position_ordered_equals_exact(L):
    """vtm_match_filtered_ordered: the matcher handed both row lists sorted by token position must report, in the ORIGINAL
    indexing, the bits of the exact matcher on the original lists -- rows in random order (what levels 2 / global see), rows
    listed twice (exact ties between dst rows: the lowest ORIGINAL index wins, in refine and in the escape), every data regime
    incl. the flat region (escape) and duplicates, a two-part pool with positions, rows without a position, a zero dst token
    (whole-call escape, NaN rule by original index), both launch plans."""
    from vidtome_amd import sites
    g = torch.Generator().manual_seed(91)
    cases = [("corr01", 2, 8, 256, 320, 6), ("corr002", 2, 8, 256, 320, 6), ("corr05", 2, 8, 256, 320, 6),
             ("smooth", 2, 8, 256, 320, 6), ("dup", 2, 8, 256, 320, 6), ("flat25", 2, 8, 256, 320, 6),
             ("n01", 2, 8, 256, 320, 6), ("corr01", 3, 5, 384, 256, 3), ("corr05", 2, 6, 512, 640, 4),
             ("corr01", 1, 7, 441, 320, 5), ("corr01", 2, 12, 1024, 320, 9), ("flat25", 1, 6, 1024, 320, 4)]
    for ci, (regime, B, F, N, C, fs) in enumerate(cases):
        x = sites.regime_tokens(regime, B, F, N, C, g)
        x = torch.nn.functional.layer_norm(x, (C,)).reshape(B, F * N, C).half().to(DEV)
        Ns0, Nd0 = fs * N, (F - fs) * N
        # random order, a tenth of the rows listed twice, ragged lengths
        def listing(lo, n):
            extra = torch.randint(lo, lo + n, (B, n // 10 + 3), generator=g)
            rows = torch.cat([torch.arange(lo, lo + n).expand(B, n), extra], 1)
            perm = torch.stack([torch.randperm(rows.shape[1], generator=g) for _ in range(B)])
            return torch.gather(rows, 1, perm).to(torch.int32).to(DEV).contiguous()
        ra, rb = listing(0, Ns0), listing(Ns0, Nd0)
        Ns, Nd = ra.shape[1], rb.shape[1]
        a_op, _ = L.normalize_gather(x, None, ra)
        b_op, _ = L.normalize_gather(x, None, rb)
        exact = L.match(a_op, b_op, Ns, Nd, False)
        a_s, a_o, b_s, b_o, table = L.position_order(ra, rb, F * N, N, None, F * N)
        seed = (N, F * N, None, table)
        for mode in (L.MATCH_ONE_LAUNCH, L.MATCH_SCOUT_RANGE):
            got, flag = L.match_filtered(x, None, a_s, b_s, False, want_flag=True, seed=seed, mode=mode, order=(a_o, b_o))
            assert torch.equal(got, exact), (regime, B, F, N, C, mode)
            f = flag.tolist()
            if mode == L.MATCH_SCOUT_RANGE and Nd > 128:
                assert f[4] > 0 and 0 < f[7] <= f[4], (regime, f)
                if regime in ("corr01", "corr002") and N >= 1024:
                    assert f[7] < 0.3 * f[4], (regime, f)
            if regime == "flat25":
                assert f[2] > 0, (regime, f)                       # the escape ran (and broke its ties by original index)
        assert torch.equal(L.match_filtered(x, None, a_s, b_s, False, seed=seed, mode=L.MATCH_SCOUT_RANGE, order=(a_o, b_o),
                                            scout_steps=1), exact)
        # aligned matching (one result row per src index over all samples' dst rows): ONE order for every sample, sample 0's --
        # the samples' lists here are different rows in different orders, the shared order only has to keep entry i the same
        # original index everywhere
        if B > 1 and ci in (0, 3, 4, 5, 7, 10):
            exact_al = L.match(a_op, b_op, Ns, Nd, True)
            sa, oa, sb, ob, tb_al = L.position_order(ra, rb, F * N, N, None, F * N, shared=True)
            assert all(torch.equal(oa[0], oa[b_]) and torch.equal(ob[0], ob[b_]) for b_ in range(B))
            assert torch.equal(torch.gather(ra, 1, oa.long()), sa) and torch.equal(torch.gather(rb, 1, ob.long()), sb)
            for mode in (L.MATCH_ONE_LAUNCH, L.MATCH_SCOUT_RANGE):
                got = L.match_filtered(x, None, sa, sb, True, seed=(N, F * N, None, tb_al), mode=mode, order=(oa, ob),
                                       scout_steps=1 if ci == 0 else 0)
                assert torch.equal(got, exact_al), (regime, "aligned", mode)
        # no seeds / garbage seeds
        assert torch.equal(L.match_filtered(x, None, a_s, b_s, False, order=(a_o, b_o)), exact)
        junk = torch.randint(-3, Nd + 50, (B, N), generator=g, dtype=torch.int32).to(DEV)
        assert torch.equal(L.match_filtered(x, None, a_s, b_s, False, seed=(N, F * N, None, junk), mode=L.MATCH_SCOUT_RANGE,
                                            order=(a_o, b_o)), exact)
        if ci in (2, 5):
            # a zero dst token: the whole call is recomputed by the escape; NaN maxima report the lowest ORIGINAL NaN index
            x2 = x.clone()
            x2[B - 1, Ns0 + 7] = 0
            a2, _ = L.normalize_gather(x2, None, ra)
            b2, _ = L.normalize_gather(x2, None, rb)
            got, flag = L.match_filtered(x2, None, a_s, b_s, False, want_flag=True, seed=seed, mode=L.MATCH_SCOUT_RANGE,
                                         order=(a_o, b_o))
            assert torch.equal(got, L.match(a2, b2, Ns, Nd, False)) and int(flag[0]) == 1
        if ci in (0, 4, 8):
            # two-part pool: the dst rows (and then the src rows) are x1 rows with positions, some of them unknown
            x0, x1 = x[:, :Ns0].contiguous(), x[:, Ns0:].contiguous()
            pos1 = (torch.arange(Nd0, dtype=torch.int32) % N).expand(B, Nd0).clone()
            pos1[:, ::17] = -1
            pos1 = pos1.to(DEV).contiguous()
            for (la, lb, ea, eb) in ((ra, rb, a_op, b_op), (rb, ra, b_op, a_op)):
                ex = L.match(ea, eb, la.shape[1], lb.shape[1], False)
                s_a, o_a, s_b, o_b, tb = L.position_order(la, lb, Ns0, N, pos1, Ns0)
                for mode in (L.MATCH_ONE_LAUNCH, L.MATCH_SCOUT_RANGE):
                    got = L.match_filtered(x0, x1, s_a, s_b, False, seed=(N, Ns0, pos1, tb), mode=mode, order=(o_a, o_b))
                    assert torch.equal(got, ex), (regime, mode)


@pytest.mark.parametrize("scale", [1e-25, 2e-21, 3e-20, 1e15])
def test_match_filtered_seeds_of_tiny_tokens_with_all_negative_scores(L, scale):
    """the seed of a row is (a . b) / |a| / |b| in fp32.  With fp32 tokens around 2e-21 the norms (~2^-65) are
    still "usable" (>= 2^-100) but the dot product of two rows (~1e-39) is a denormal with no relative accuracy, or 0 -- and 0 would be published as a CERTIFIED running maximum above a row whose real scores are all negative, after
    which the filter's window and the escape's tile pruning discard the true argmax.  Round 6 publishes a seed only when the dot
    product itself is a normal number (or both norms are >= 2^-40).  Here every cosine is about -0.98 (all tokens = +-(one common
    vector + noise)), the same-position seeds included: filtered (seeded, both launch plans) == exact, bit for bit; the other
    scales check that ordinary small / large magnitudes keep their seeds working."""
    g = torch.Generator().manual_seed(44)
    B, F, N, C, fs = 2, 4, 512, 320, 3
    common = torch.randn(B, 1, 1, C, generator=g)
    x = common + 0.1 * torch.randn(B, F, N, C, generator=g)
    x[:, fs:] = -x[:, fs:]                                   # the dst frame: every src . dst is negative
    x = (x * scale).reshape(B, F * N, C).float().to(DEV)
    Ns, Nd = fs * N, (F - fs) * N
    ra = torch.arange(Ns, dtype=torch.int32, device=DEV).expand(B, Ns).contiguous()
    rb = torch.arange(Ns, Ns + Nd, dtype=torch.int32, device=DEV).expand(B, Nd).contiguous()
    a_op, _ = L.normalize_gather(x, None, ra)
    b_op, _ = L.normalize_gather(x, None, rb)
    exact = L.match(a_op, b_op, Ns, Nd, False)
    nm, _ = L.decode_best(exact)
    if scale > 1e-24:        # (1e-25: every x^2 underflows, every norm is 0 and every score NaN -- in the reference too; 2e-21: the
        # squares and the dot products are denormals, norms ~2^-65: the case the guard is for)
        assert float(nm.max()) < -0.5, float(nm.max())      # the premise: all maxima are negative
    for mode in (L.MATCH_ONE_LAUNCH, L.MATCH_SCOUT_RANGE):
        got = L.match_filtered(x, None, ra, rb, False, seed=(N, F * N, None, None), mode=mode)
        assert torch.equal(got, exact), (scale, mode)
    assert torch.equal(L.match_filtered(x, None, ra, rb, False), exact)


def test_match_planner_steers_by_the_previous_calls_counters(L):
    """merge.MatchPlanner: a scout + range call copies its counters into the planner's pinned buffer (no synchronisation in
    the product path; the test synchronises to look); uncorrelated tokens (every wave tile alive) send the block's first level
    back to the one-launch plan for COOL calls, frames of a low-noise clip keep the scout + range plan.  Behind the first level
    the planner also says whether the rows are position-ordered while the plan is off: the global level keeps the ordering
    (`order_alone`) unless nothing at all died at the scout's test, level 2 drops it."""
    from vidtome_amd import merge, sites
    g = torch.Generator().manual_seed(5)
    B, F, N, C, fs = 2, 8, 4096, 320, 6          # 4 096 tokens per frame: a src tile's matches sit in 2 of a frame's 32 dst tiles
    Ns, Nd = fs * N, (F - fs) * N
    ra = torch.arange(Ns, dtype=torch.int32, device=DEV).expand(B, Ns).contiguous()
    rb = torch.arange(Ns, Ns + Nd, dtype=torch.int32, device=DEV).expand(B, Nd).contiguous()
    seed = (N, F * N, None, None)
    for regime, expect in (("corr01", L.MATCH_SCOUT_RANGE), ("corr05", L.MATCH_ONE_LAUNCH), ("n01", L.MATCH_ONE_LAUNCH)):
        x = sites.regime_tokens(regime, B, F, N, C, g)
        x = torch.nn.functional.layer_norm(x, (C,)).reshape(B, F * N, C).half().to(DEV)
        pl = merge.MatchPlanner()
        pl.LOW = pl.HIGH                              # (the product keeps a margin between the two)
        mode, buf, keep, scout = pl.next()
        assert mode == L.MATCH_SCOUT_RANGE and buf.is_pinned() and keep and scout == 0
        L.match_filtered(x, None, ra, rb, False, seed=seed, mode=mode, stats_host=buf)
        torch.cuda.synchronize()
        assert int(buf[4]) > 0, buf.tolist()
        mode, buf, keep, scout = pl.next()
        assert mode == expect, (regime, pl.view.tolist())
        if expect == L.MATCH_SCOUT_RANGE:
            # short spans (below LOW; this planner: LOW = HIGH): the next scout tests after ONE channel step, and stays there
            # while its own spans are short
            assert scout == 1
            L.match_filtered(x, None, ra, rb, False, seed=seed, mode=mode, stats_host=buf, scout_steps=scout)
            torch.cuda.synchronize()
            assert pl.next()[3] == 1, pl.view.tolist()
        if expect == L.MATCH_ONE_LAUNCH:
            for _ in range(merge.MatchPlanner.COOL - 1):
                assert pl.next()[0] == L.MATCH_ONE_LAUNCH
            assert pl.next()[0] == L.MATCH_SCOUT_RANGE          # ... and tries again
    # a position-ordered level (rows in random order, sorted for the matcher): low-noise clip -> ordered + range; a noisy one
    # -> one launch, ordered at the global level (order_alone) and not at level 2; uncorrelated tokens -> the first one-launch
    # call reports that nothing dies in the filter either and the ordering is dropped
    perm_a = torch.stack([torch.randperm(Ns, generator=g) for _ in range(B)]).to(torch.int32).to(DEV)
    perm_b = torch.stack([Ns + torch.randperm(Nd, generator=g) for _ in range(B)]).to(torch.int32).to(DEV)
    for regime, order_alone, expect, settled in (("corr01", True, (L.MATCH_SCOUT_RANGE, True), None),
                                                 ("corr05", True, (L.MATCH_ONE_LAUNCH, True), True),
                                                 ("corr05", False, (L.MATCH_ONE_LAUNCH, False), None),
                                                 ("n01", True, (L.MATCH_ONE_LAUNCH, True), False)):
        x = sites.regime_tokens(regime, B, F, N, C, g)
        x = torch.nn.functional.layer_norm(x, (C,)).reshape(B, F * N, C).half().to(DEV)
        pl = merge.MatchPlanner(order_alone)
        mode, buf, keep, scout = pl.next()
        assert mode == L.MATCH_SCOUT_RANGE and keep
        a_s, a_o, b_s, b_o, tb = L.position_order(perm_a, perm_b, F * N, N, None, F * N)
        L.match_filtered(x, None, a_s, b_s, False, seed=(N, F * N, None, tb), mode=mode, stats_host=buf, order=(a_o, b_o))
        torch.cuda.synchronize()
        mode, buf, keep, scout = pl.next()
        assert (mode, keep) == expect, (regime, order_alone, pl.view.tolist())
        if settled is not None:
            assert buf is not None                    # the one-launch call is asked for its counters once ...
            L.match_filtered(x, None, a_s, b_s, False, seed=(N, F * N, None, tb), mode=mode, stats_host=buf, order=(a_o, b_o))
            torch.cuda.synchronize()
            mode, buf, keep, scout = pl.next()
            assert mode == L.MATCH_ONE_LAUNCH and buf is None and keep == settled, (regime, pl.view.tolist())


def test_planner_cost_is_bounded_on_alternating_regimes(L):
    """what merge.MatchPlanner's hysteresis costs when a clip keeps changing character.  One top
    level-1 geometry (cfg-2: 49 152 x 16 384 x 320), the data alternating between a low-noise clip (corr01: the scout + range
    plan wins) and uncorrelated tokens (n01: it loses) every 2, 8 and 300 calls.  Each (regime, plan) pair is timed once
    (median of 9); the planner is then driven with REAL calls (its decisions are functions of the counters those calls copy
    back) and the sequence is priced with the medians -- independent of timer noise.  Bounds: never more than 3 % above
    ALWAYS the one-launch plan (rounds 1-4's: the planner cannot cost more than its exploring calls), and within 5 % of a
    per-call oracle once a regime lasts 300 calls.  (Against faster alternation nothing that steers by the previous call can
    follow; measured in round 6: at period 2 always scouting after one step is 12 % cheaper than what the planner does --
    the scout saves 0.42 ms on the clip and costs 0.19 ms on the noise -- at the price of +17 % on data it never pays for.
    Consecutive calls of one level are consecutive chunks of one video at one noise level: the planner is tuned for that.)"""
    from vidtome_amd import merge, sites
    g = torch.Generator().manual_seed(11)
    B, F, N, C, fs = 2, 16, 4096, 320, 12
    Ns, Nd = fs * N, (F - fs) * N
    ra = torch.arange(Ns, dtype=torch.int32, device=DEV).expand(B, Ns).contiguous()
    rb = torch.arange(Ns, Ns + Nd, dtype=torch.int32, device=DEV).expand(B, Nd).contiguous()
    seed = (N, F * N, None, None)
    xs = {}
    for regime in ("corr01", "n01"):
        x = sites.regime_tokens(regime, B, F, N, C, g)
        xs[regime] = torch.nn.functional.layer_norm(x, (C,)).reshape(B, F * N, C).half().to(DEV)
    plans = {"one": (L.MATCH_ONE_LAUNCH, 0), "range": (L.MATCH_SCOUT_RANGE, 0), "range1": (L.MATCH_SCOUT_RANGE, 1)}
    t = {}
    for regime, x in xs.items():
        ref = None
        for pname, (mode, scout) in plans.items():
            call = lambda: L.match_filtered(x, None, ra, rb, False, seed=seed, mode=mode, scout_steps=scout)
            best = call()
            ref = best if ref is None else ref
            assert torch.equal(best, ref), (regime, pname)
            ts = []
            for _ in range(9):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                call()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            t[regime, pname] = sorted(ts)[4]
    assert t["corr01", "range"] < t["corr01", "one"] and t["n01", "range"] > t["n01", "one"], t   # the premise
    report = {}
    for period, ncalls in ((2, 600), (8, 600), (300, 1200)):
        pl = merge.MatchPlanner()
        total = oracle_total = 0.0
        fixed = {p: 0.0 for p in plans}
        issued = {p: 0 for p in plans}
        for i in range(ncalls):
            regime = ("corr01", "n01")[(i // period) % 2]
            mode, buf, _, scout = pl.next()
            L.match_filtered(xs[regime], None, ra, rb, False, seed=seed, mode=mode, stats_host=buf, scout_steps=scout)
            torch.cuda.synchronize()                    # (the product never waits; the test makes the reading deterministic)
            pname = "one" if mode == L.MATCH_ONE_LAUNCH else ("range1" if scout else "range")
            issued[pname] += 1
            total += t[regime, pname]
            oracle_total += min(t[regime, p] for p in plans)
            for p in plans:
                fixed[p] += t[regime, p]
        report[period] = (total / fixed["one"], total / min(fixed.values()), total / oracle_total, issued)
        assert total <= 1.03 * fixed["one"], (period, report[period], t)
    assert report[300][2] <= 1.05, (report, t)
    print("planner cost (vs the one-launch plan, vs the best fixed plan, vs a per-call oracle, calls per plan):", report,
          "ms per call:", t)


def test_refine_many_candidates_per_row(L):
    """Rows with MANY candidates inside the fp16 filter's window (what frames of one clip at low noise produce: 4-17 per
    row): every src row has 6-12 dst rows whose scores differ by ~1e-7 ... 1e-3, exact duplicates among them, C up to 1280,
    fp32 / bf16 tokens -- refine_kernel's 64-row workgroups (round 5) reserve, write and work off pair slices of up to
    64 x CAP entries: filtered == exact bit for bit."""
    g = torch.Generator(device=DEV).manual_seed(9)
    for (B, Ns, Nd, C, reps, noise, dtype) in [(2, 3000, 2400, 320, 12, 1e-4, torch.float16), (1, 1500, 1200, 1280, 12, 3e-5, torch.float32),
                                               (2, 2000, 2400, 640, 8, 1e-3, torch.bfloat16), (2, 1024, 1536, 320, 6, 0.0, torch.float16)]:
        nb = Nd // reps
        base = torch.randn(B, nb, C, generator=g, device=DEV)
        dst = base.repeat(1, reps, 1) + noise * torch.randn(B, nb * reps, C, generator=g, device=DEV)
        src = base[:, torch.randint(0, nb, (Ns,), generator=g, device=DEV)] + 0.05 * torch.randn(B, Ns, C, generator=g, device=DEV)
        x = torch.cat([src, dst], dim=1).to(dtype).contiguous()
        Nd_ = nb * reps
        ra = torch.arange(Ns, dtype=torch.int32, device=DEV).expand(B, Ns).contiguous()
        rb = torch.arange(Ns, Ns + Nd_, dtype=torch.int32, device=DEV).expand(B, Nd_).contiguous()
        a_op, _ = L.normalize_gather(x, None, ra)
        b_op, _ = L.normalize_gather(x, None, rb)
        for align in (False, True):
            exact = L.match(a_op, b_op, Ns, Nd_, align)
            got, flag = L.match_filtered(x, None, ra, rb, align, want_flag=True)
            assert torch.equal(got, exact), (C, dtype, align)
            rows = Ns if align else B * Ns
            assert flag[3].item() >= 3 * rows or flag[2].item() > 0, flag.tolist()     # several candidates per row reached refine


def test_match_filtered_worst_cases_are_bounded(L):
    """TIME, not only bits: the escapes of the filtered matcher must cost what the exact fp32-MFMA matcher
    costs, not the ~1000x of the scalar row pass rounds 1-3 fell back to.  cfg-2 top-block level 1 (2 x 49 152 x 16 384 x
    320): (a) a quarter of every frame is a flat region -> 25 % of the src rows overflow their candidate lists; (b) ONE zero
    token among the dst rows -> every row of the call is recomputed.  Both within 2x the exact kernel on the same input."""
    B, Ns, Nd, C, N = 2, 49152, 16384, 320, 4096
    ra = torch.arange(Ns, dtype=torch.int32, device=DEV).expand(B, Ns).contiguous()
    rb = torch.arange(Ns, Ns + Nd, dtype=torch.int32, device=DEV).expand(B, Nd).contiguous()

    def ms(fn, n=3):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return sorted(ts)[len(ts) // 2]

    report = {}
    for name in ("corr05", "flat25", "zero"):
        x = _regime_tokens("corr05" if name == "zero" else name, B, Ns, Nd, C, N)
        if name == "zero":
            x[0, Ns + 5] = 0
        x = x.to(DEV)
        a_op, _ = L.normalize_gather(x, None, ra)
        b_op, _ = L.normalize_gather(x, None, rb)
        exact = L.match(a_op, b_op, Ns, Nd, False)
        got, flag = L.match_filtered(x, None, ra, rb, False, want_flag=True)
        assert torch.equal(got, exact), name
        t_exact = ms(lambda: L.match(a_op, b_op, Ns, Nd, False))
        t_filt = ms(lambda: L.match_filtered(x, None, ra, rb, False))
        report[name] = (round(t_filt, 2), round(t_exact, 2), flag.tolist())
        del a_op, b_op
    print("filtered ms / exact ms / flags:", report)
    assert report["corr05"][2][0] == 0 and report["corr05"][2][2] == 0
    assert report["flat25"][2][0] == 0 and report["flat25"][2][2] >= B * Ns // 5      # ~25 % of the rows escaped
    assert report["zero"][2][0] == 1
    for name in ("flat25", "zero"):
        assert report[name][0] <= 2.0 * report[name][1], report


@pytest.mark.parametrize("C", [64, 256, 1024])
def test_match_filtered_adversarial_window(L, C):
    """The filter drops the fp16 residuals (`lo`) of both operands; its window budget (match_filter.hip, EPS) assumes
    the worst case sum |a_k b_k| = 1 with every dropped half-ulp pulling the same way.  Build exactly that: rows whose
    normalised components all sit a hair under the rounding midpoint above a power of two (1024 * xhat = 2^e * (1 +
    0.96 * 2^-11): hi = 2^e, lo = +0.96 half-ulp -- the largest relative residual fp16 allows), all products positive,
    cosine ~1 between all of them, and true scores that differ only at the 1e-6 level while the filter errs by up to
    ~9.6e-4 in one direction for some pairs and not for others.  filtered must still equal exact, bit for bit."""
    g = torch.Generator().manual_seed(C)
    B, Ns, Nd = 2, 300, 500
    h = 1024.0 / C ** 0.5                           # 1024 * xhat of a flat unit vector: a power of two for these C
    assert 2 ** round(np.log2(h)) == h
    up = h * (1 + 0.96 * 2.0 ** -11) / 1024.0       # rounds DOWN to h: lo = +0.96 half-ulp (relative 4.7e-4)
    dn = h * (1 - 0.96 * 2.0 ** -12) / 1024.0       # rounds UP to h:   lo = -0.96 half-ulp of the binade below
    flat = h / 1024.0

    def rows(n, kinds):
        r = torch.empty(B, n, C, dtype=torch.float64)
        for i in range(n):
            v = kinds[i % len(kinds)]
            r[:, i] = v
            r[:, i, (7 * i) % C] = (1.0 - (C - 1) * v * v) ** 0.5     # one component absorbs the rest of the unit norm
        r *= 1 + 2e-6 * torch.randn(B, n, C, generator=g, dtype=torch.float64)      # decides the TRUE order
        return r
    # src rows of both signs of residual; dst rows of both signs plus exactly representable ones
    x = torch.cat([rows(Ns, [up, dn]), rows(Nd, [up, dn, flat, up])], dim=1).float()
    flag = _filtered_vs_exact(L, x.to(DEV), Ns, Nd, False)
    _filtered_vs_exact(L, x.to(DEV), Ns, Nd, True)
    assert flag == 0
    # the construction does what it says: the hi-only products of an (up, up) pair underestimate by ~2u
    xh = x[0, :1].double() / x[0, :1].double().norm()
    hi = (xh * 1024).half().double()
    assert 8e-4 < float(1.0 - (hi * hi).sum() / 1024 ** 2) < 9.9e-4


def test_match_filtered_full_size(L):
    """cfg-2 top-block level-1 size on frame-correlated fp16 tokens: filtered == exact, no fallback."""
    g = torch.Generator().manual_seed(13)
    B, Ns, Nd, C = 2, 49152, 16384, 320
    base = torch.randn(B, 4096, C, generator=g)
    x = (base.repeat(1, 16, 1) + 0.5 * torch.randn(B, Ns + Nd, C, generator=g)).half().to(DEV)
    _filtered_vs_exact(L, x, Ns, Nd, False, expect_flag=0)


def test_match_filtered_mid_global_size(L):
    """cfg-2 mid-block global level (8 704 x 8 704, C = 640): the shape whose dst-split count is lowered to 6 so that
    an XCD's share of workgroups fits one round (match_filter.hip launcher) -- filtered == exact, no fallback."""
    g = torch.Generator().manual_seed(14)
    B, Ns, Nd, C = 2, 8704, 8704, 640
    base = torch.randn(B, Nd, C, generator=g)
    x = (base.repeat(1, 2, 1) + 0.3 * torch.randn(B, Ns + Nd, C, generator=g)).half().to(DEV)
    _filtered_vs_exact(L, x, Ns, Nd, False, expect_flag=0)


@pytest.mark.parametrize("n", [1, 2, 17, 1000, 1024, 3072, 5000, 8704, 12288, 16384, 16385, 49152, 110592])
def test_sort_desc(L, oracle, n):
    rng = np.random.default_rng(n)
    k = rng.standard_normal((2, n)).astype(np.float32)
    if n > 20:
        k[:, ::7] = k[:, 1::7][:, :k[:, ::7].shape[1]]     # exact ties
        k[0, 3] = np.nan
        k[1, 11] = -0.0
        k[1, 12] = 0.0
    # build packed keys the way vtm_match does
    f = k + np.float32(0.0)
    u = f.view(np.uint32).astype(np.uint64)
    o = np.where(u & 0x80000000, (~u) & 0xFFFFFFFF, u | 0x80000000)
    o = np.where(np.isnan(k), np.uint64(0xFFFFFFFF), o)
    packed = (o << np.uint64(32)) | np.uint64(0x12345)
    perm = L.sort_desc(_t(packed.view(np.int64))).cpu().numpy()
    assert np.array_equal(perm, oracle.sort_desc(k))


def test_sort_desc_fuzz_vs_torch_stable_sort(L):
    """Random lengths (1 ... 120 000 keys, 1-3 rows) with heavy ties in every digit position, against torch's stable
    descending sort of the same 32-bit keys (ties keep ascending index order)."""
    g = torch.Generator().manual_seed(9)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    for case in range(24):
        rows = ri(1, 3)
        n = [1, 2, 255, 256, 257, 1000, 4097, 12288, 49152, 110592][case % 10] if case < 10 else ri(1, 120000)
        distinct = [2, 17, 300, 70000, 1 << 31][case % 5]
        hi = torch.randint(0, distinct, (rows, n), generator=g, dtype=torch.int64)
        if case % 4 == 1:
            hi = hi << ri(0, 24) & 0xFFFFFFFF          # ties concentrated in the low / high digits
        packed = (hi << 32) | torch.randint(0, 1 << 31, (rows, n), generator=g, dtype=torch.int64)
        perm = L.sort_desc(packed.to(DEV)).cpu().to(torch.int64)
        ref = torch.sort(hi, dim=1, descending=True, stable=True).indices
        assert torch.equal(perm, ref), (case, rows, n, distinct)


def test_gather_and_unmerge_add(L, oracle):
    g = torch.Generator().manual_seed(5)
    for dtype in (torch.float32, torch.float16, torch.bfloat16):
        B, P0, P1, C, M, Ln = 2, 100, 30, 48, 77, 150
        x0 = torch.randn(B, P0, C, generator=g).to(dtype)
        x1 = torch.randn(B, P1, C, generator=g).to(dtype)
        idx = torch.randint(0, P0 + P1, (B, M), generator=g, dtype=torch.int32)
        out = L.gather_rows(x0.to(DEV), x1.to(DEV), idx.to(DEV), pad_to=8)
        ref = torch.gather(torch.cat([x0, x1], 1), 1, idx.long()[..., None].expand(-1, -1, C))
        assert out.shape[1] == 80 and torch.equal(out[:, :M].cpu(), ref) and torch.count_nonzero(out[:, M:]) == 0
        inv = torch.randint(0, M, (B, Ln), generator=g, dtype=torch.int32)
        resid = torch.randn(B, Ln, C, generator=g).to(dtype)
        got = L.unmerge_add(out, inv.to(DEV), resid.to(DEV)).cpu()
        exp = torch.gather(ref, 1, inv.long()[..., None].expand(-1, -1, C)) + resid
        assert torch.equal(got, exp)


# ---------------------------------------------------------------------------------------------------
# matchers through the mirrored Python API vs golden vectors from the reference
# ---------------------------------------------------------------------------------------------------
def _cmp_info(info, c):
    for n in IDX:
        assert np.array_equal(info[n].cpu().numpy().astype(np.int64), c[n]), n


def test_randframe_golden_gpu(L):
    from vidtome_amd import merge
    for c in load_cases("randframe.npz"):
        torch.manual_seed(int(c["seed"]))
        gen = torch.Generator(device="cpu").set_state(torch.get_rng_state())
        x = _t(c["x"])
        m, u, info = merge.bipartite_soft_matching_randframe(x, int(c["F"]), float(c["ratio"]), int(c["unm_pre"]),
                                                             gen, int(c["stride"]), bool(c["align"]))
        assert info["unm_num"] == c["unm_num"]
        if c["noop"]:
            assert m is merge.do_nothing and u is merge.do_nothing
            continue
        _cmp_info(info, c)
        assert np.array_equal(m(x).cpu().numpy(), c["merged"])
        assert np.array_equal(u(_t(c["y"])).cpu().numpy(), c["unmerged"])


def test_2s_golden_gpu(L):
    from vidtome_amd import merge
    for c in load_cases("twos.npz"):
        x = _t(c["x"])
        m, u, info = merge.bipartite_soft_matching_2s(x, int(c["src_len"]), float(c["ratio"]), bool(c["align"]),
                                                      unmerge_chunk=int(c["unmerge_chunk"]))
        _cmp_info(info, c)
        assert np.array_equal(m(x).cpu().numpy(), c["merged"])
        assert np.array_equal(u(_t(c["y"])).cpu().numpy(), c["unmerged"])
    assert len(merge.bipartite_soft_matching_2s(_t(load_cases("twos.npz")[0]["x"]), 40, 0.0, False)) == 2  # merge.py:364-365


def test_nan_and_kat_golden_gpu(L):
    from vidtome_amd import merge
    nan_c, kat = load_cases("misc.npz")
    torch.manual_seed(int(nan_c["seed"]))
    gen = torch.Generator(device="cpu").set_state(torch.get_rng_state())
    _, _, info = merge.bipartite_soft_matching_randframe(_t(nan_c["x"]), int(nan_c["F"]), float(nan_c["ratio"]), 0, gen)
    _cmp_info(info, nan_c)
    for align, sfx in ((False, ""), (True, "_al")):
        gen = torch.Generator().manual_seed(123)
        _, _, info = merge.bipartite_soft_matching_randframe(_t(kat["x"]), 4, 0.5, 0, gen, 4, align)
        for n in ("unm_idx", "src_idx", "dst_idx"):
            assert np.array_equal(info[n].cpu().numpy(), kat[n + sfx]), n


def _planted_gpu(c):
    from vidtome_amd import merge
    a, b = planted_inputs(int(c["Ns"]), int(c["Nd"]), int(c["C"]), seed=int(c["seed"]))
    F, randf = int(c["F"]), int(c["randf"])
    Ltot = a.shape[1] + b.shape[1]
    tnum = Ltot // F
    x = np.empty((1, Ltot, a.shape[2]), np.float32)
    is_dst = (np.arange(Ltot) // tnum) % 4 == randf
    x[0, is_dst] = b[0]
    x[0, ~is_dst] = a[0]
    lv = merge.local_level(_t(x), None, Ltot, F, float(c["ratio"]), 0, randf, 4, False, want_indices=True)
    for n in ("unm_idx", "src_idx", "dst_idx"):
        got = getattr(lv, n)[0].cpu().numpy().astype(np.int32)
        assert np.array_equal(got[:16], c[n + "_head"]), n
        assert hashlib.sha256(got.tobytes()).hexdigest() == str(c[n + "_sha256"]), n


def test_planted_golden_gpu(L):
    """cfg-2 top-block level-1 size (49152 x 16384 x 320): the REFERENCE's index arrays (sha256) reproduced
    bit-exactly by the HIP path."""
    for c in load_cases("planted.npz"):
        _planted_gpu(c)


@pytest.mark.parametrize("fname", ["planted_mid.npz", "planted_cfg14.npz"])
@pytest.mark.parametrize("mode", ["filtered", "exact"])
def test_planted_mid_and_full_size_golden_gpu(L, mode, fname, monkeypatch):
    """tests/golden/planted_mid.npz (REFERENCE runs, make_golden_mid.py): local matcher at 256 / 1 024 tokens per frame,
    C = 320 / 640, level-2 shape, aligned batches; GLOBAL matcher (`bipartite_soft_matching_2s`) up to the full cfg-2 sizes
    8 704^2 x 640 and 34 816^2 x 320, both unmerge_chunk values, rectangular, aligned; and the LARGEST levels of any
    BASELINE configuration, cfg-5's level 1 (110 592 x 36 864 x 320) and its ragged global level (64 513^2 x 320).  The HIP
    path -- the default filtered matcher AND the exact fp32 kernel -- reproduces the reference's index arrays bit for bit
    (sha256).  planted_cfg14.npz (round 4): the exact level shapes of cfg-1 (3 072 x 1 024 x 320; 768 x 256 x 640) and of a
    cfg-4 chunk (24 576 x 8 192, 4 096 x 16 384 with carried-over unmerged tokens, global 18 432^2; mid: 6 144 x 2 048 x 640,
    4 608^2)."""
    from inputs import idx_sha, planted_batch, planted_local_chunk
    from vidtome_amd import merge
    monkeypatch.setattr(merge, "MATCH_MODE", mode)
    for c in load_cases(fname):
        name = str(c["name"])
        if mode == "exact" and ("cfg5" in name or "cfg3" in name):
            continue                                    # 1.5-2.6 TFLOP each on the fp32 MFMA: covered by the filtered run
        if str(c["kind"]) == "local":
            x = planted_local_chunk(int(c["B"]), int(c["F"]), int(c["tnum"]), int(c["unm_pre"]), int(c["C"]),
                                    int(c["randf"]), int(c["seed"]))
            torch.manual_seed(123)
            gen = torch.Generator(device="cpu").set_state(torch.get_rng_state())
            m, u, info = merge.bipartite_soft_matching_randframe(_t(x), int(c["F"]), float(c["ratio"]), int(c["unm_pre"]),
                                                                 gen, 4, bool(c["align"]))
        else:
            a, b = planted_batch(int(c["src_len"]), int(c["dst_len"]), int(c["C"]), int(c["seed"]), int(c["B"]))
            x = np.concatenate([a, b], axis=1)
            m, u, info = merge.bipartite_soft_matching_2s(_t(x), int(c["src_len"]), float(c["ratio"]), bool(c["align"]),
                                                          unmerge_chunk=int(c["unmerge_chunk"]))
            y = torch.zeros((x.shape[0], info["unm_num"] + int(c["dst_len"]), 8), device=DEV)
            assert u(y).shape[1] == int(c["unmerged_len"]), name
        assert info["unm_num"] == int(c["unm_num"]), name
        for n in ("unm_idx", "src_idx", "dst_idx"):
            got = info[n].cpu().numpy().astype(np.int32)
            assert tuple(got.shape) == tuple(c[n + "_shape"]), (name, n, got.shape)
            assert np.array_equal(got[..., :16], c[n + "_head"]), (name, n)
            assert idx_sha(got) == str(c[n + "_sha256"]), (name, n)


class _FixedPlan:
    """Stands in for merge.MatchPlanner in the pinning tests below: ALWAYS the same launch plan (mode, scout depth, position
    order or not), counters copied into a pinned buffer the test reads behind a synchronise."""

    def __init__(self, mode, scout, order):
        self.mode, self.scout, self.order = mode, scout, order
        self.buf = torch.zeros(8, dtype=torch.int32).pin_memory()

    def next(self):
        self.buf.zero_()
        return self.mode, self.buf, self.order, self.scout


@pytest.mark.parametrize("fname", ["planted.npz", "planted_mid.npz", "planted_cfg14.npz"])
def test_launch_plans_vs_reference_sha_at_full_size(L, fname, monkeypatch):
    """VERDICT r05 missing 3: the launch plans round 5 added -- scout + range, the one-step scout, position-ordered levels
    with the shared (aligned) order -- held to the REFERENCE's own index arrays (sha256, tests/golden/make_golden*.py) at the
    level shapes of every BASELINE configuration, not only to the one-launch HIP plan at <= 12 288 rows.  Every case runs
    under each plan {one launch, scout + range} x scout depth {the filter's own, ONE channel step}; the global
    (`bipartite_soft_matching_2s`) cases additionally with the rows handed over in position order (vtm_position_order;
    merge.POSITION_ORDER_MIN_PAIRS lifted so that the small ones order too), the anchors' positions chosen so that a src
    row's planted partner sits at ITS position wherever that is free (the seeds then start every row at its true maximum);
    `flags_out[7]` (blocks inside the spans of the second launch) says the plan really ran.  (Planted cosines go down to 0.55,
    below the rest bound at the scout's depth, so most tiles stay marked here; the sparse-span regime at full size is the
    low-noise clip of test_compute_merge_with_live_planners_equals_exact_at_cfg2.)  merge.py:87-117, 392-421."""
    from inputs import idx_sha, planted_batch, planted_local_chunk
    from vidtome_amd import merge
    monkeypatch.setattr(merge, "MATCH_MODE", "filtered")
    monkeypatch.setattr(merge, "MATCH_PLAN", "auto")
    monkeypatch.setattr(merge, "SHALLOW_SCOUT", True)
    monkeypatch.setattr(merge, "POSITION_ORDER_MIN_PAIRS", 1)
    R, O = L.MATCH_SCOUT_RANGE, L.MATCH_ONE_LAUNCH
    ran_range = ran_ordered = 0

    def check(c, lv, what):
        name = str(c["name"])
        assert lv.unm_num == int(c["unm_num"]) if "unm_num" in c else True, (name, what)
        for n in ("unm_idx", "src_idx", "dst_idx"):
            got = getattr(lv, n).cpu().numpy().astype(np.int32)
            if str(c["kind"]) == "planted":
                got = got[0]
            assert np.array_equal(got[..., :16], c[n + "_head"]), (name, what, n)
            sha = hashlib.sha256(got.tobytes()).hexdigest() if str(c["kind"]) == "planted" else idx_sha(got)
            assert sha == str(c[n + "_sha256"]), (name, what, n)

    for c in load_cases(fname):
        kind, name = str(c["kind"]), str(c["name"])
        if kind in ("planted", "local"):
            if kind == "planted":
                a, b = planted_inputs(int(c["Ns"]), int(c["Nd"]), int(c["C"]), seed=int(c["seed"]))
                F, randf, unm_pre, align = int(c["F"]), int(c["randf"]), 0, False
                Ltot = a.shape[1] + b.shape[1]
                tnum = Ltot // F
                x = np.empty((1, Ltot, a.shape[2]), np.float32)
                is_dst = (np.arange(Ltot) // tnum) % 4 == randf
                x[0, is_dst], x[0, ~is_dst] = b[0], a[0]
            else:
                F, randf, unm_pre, align, tnum = int(c["F"]), int(c["randf"]), int(c["unm_pre"]), bool(c["align"]), int(c["tnum"])
                x = planted_local_chunk(int(c["B"]), F, tnum, unm_pre, int(c["C"]), randf, int(c["seed"]))
            if unm_pre:
                continue                 # (a level-2 SHAPE without the sequence behind it has no positions: compute_merge test below)
            xt = _t(x)
            can_range = tnum >= 256 and tnum % 128 == 0 and x.shape[2] >= 256     # include/vidtome_hip.h: the plan's conditions
            for mode, scout in ((O, 0), (R, 0), (R, 1)):
                fp = _FixedPlan(mode, scout, True)
                lv = merge.local_level(xt, None, x.shape[1], F, float(c["ratio"]), 0, randf, 4, align, True, tokens=tnum,
                                       planner=fp)
                torch.cuda.synchronize()
                if mode == R and can_range:
                    assert int(fp.buf[7]) > 0 and int(fp.buf[4]) > 0, (name, scout, fp.buf.tolist())
                    ran_range += 1
                check(c, lv, (mode, scout))
        else:
            src_len, dst_len, B, C = int(c["src_len"]), int(c["dst_len"]), int(c["B"]), int(c["C"])
            a, b = planted_batch(src_len, dst_len, C, int(c["seed"]), B)
            xs, xd = _t(a), _t(b)
            tokens = 4096 if C == 320 else 1024
            align = bool(c["align"])
            # positions: local (src) row i sits at i % tokens (rows of the joined chunk); dst row j gets the position of the src
            # row the REFERENCE-equal unseeded run matched to it (last writer wins), else j % tokens
            lv0 = merge.global_level(xs, xd, None, src_len, True, float(c["ratio"]), align, True)
            check(c, lv0, "unseeded")
            pos = (torch.arange(dst_len, device=DEV) % tokens).to(torch.int32).expand(B, dst_len).contiguous()
            _, node_idx = L.decode_best(lv0.best)                   # (B, Ns), aligned: (1, Ns) over the B * Nd concatenation
            node_idx = node_idx.long().reshape(-1, src_len)
            src_pos = (torch.arange(src_len, device=DEV) % tokens).to(torch.int32)
            for bi in range(B):
                if node_idx.shape[0] == B:
                    pos[bi, node_idx[bi]] = src_pos
                else:                                               # the sample whose dst row won the aligned maximum
                    mine = (node_idx[0] // dst_len) == bi
                    pos[bi, (node_idx[0] % dst_len)[mine]] = src_pos[mine]
            for mode, scout, order in ((O, 0, False), (O, 0, True), (R, 0, True), (R, 1, True)):
                fp = _FixedPlan(mode, scout, order)
                lv = merge.global_level(xs, xd, None, src_len, True, float(c["ratio"]), align, True, tokens=tokens,
                                        anchor_positions=pos, planner=fp)
                torch.cuda.synchronize()
                if mode == R and C >= 256 and dst_len >= 256:
                    assert int(fp.buf[7]) > 0 and int(fp.buf[4]) > 0, (name, scout, fp.buf.tolist())
                    ran_range += 1
                ran_ordered += order
                check(c, lv, (mode, scout, order))
    assert ran_range > 0 and (fname == "planted.npz" or ran_ordered > 0)


@pytest.mark.parametrize("regime", ["corr01", "n01"])
def test_compute_merge_with_live_planners_equals_exact_at_cfg2(L, regime, monkeypatch):
    """VERDICT r05 missing 3, second half: the code the bench times.  cfg-2's top and mid merging sites (16 frames of 64 x 64
    / 32 x 32 tokens, batch 2, local 0.5 + global 0.5), five chunks of one clip, `compute_merge` with the block's
    MatchPlanners LIVE (they switch plans between the chunks: scout + range and the one-step scout on the low-noise clip,
    back to one launch and out of the position order on uncorrelated tokens) -- every level's unm / src / dst indices and
    the anchors it leaves behind equal the run with the exact fp32 matcher (VIDTOME_MATCH=exact, itself pinned to the
    reference's sha256 at these shapes: test_planted_mid_and_full_size_golden_gpu).  patch.py:44-85."""
    import vidtome_amd
    from vidtome_amd import merge, sites
    from vidtome_amd import patch as vpatch
    B, F, latent = 2, 16, (64, 64)
    sl = [sites.Site("up3.0", 1, 320, 8), sites.Site("up2.0", 2, 640, 8)]
    issued = []
    orig = L.match_filtered

    def logged(*a, **k):
        issued.append((k.get("mode", 0), k.get("scout_steps", 0), k.get("order") is not None))
        return orig(*a, **k)

    runs = {}
    for mode in ("exact", "filtered"):
        monkeypatch.setattr(merge, "MATCH_MODE", mode)
        monkeypatch.setattr(L, "match_filtered", logged)
        unet = sites.SiteUNet(sl, seed=4).to(device=DEV, dtype=torch.float16)
        vidtome_amd.apply_patch(unet, local_merge_ratio=0.5, merge_global=True, global_merge_ratio=0.5, batch_size=B)
        unet.set_size(latent)
        torch.manual_seed(123)
        rec = []
        with torch.no_grad():
            for ck in range(5):
                for i, (site, blk) in enumerate(zip(sl, unet.blocks)):
                    if not hasattr(blk, "generator"):
                        blk.generator = vpatch.init_generator(torch.device(DEV))
                    h = sites.synthetic_hidden(site, B, F, latent, torch.float16, DEV, seed=700 + 13 * (ck % 3) + i,
                                               clip_seed=77 + i, regime=regime, frame0=(ck % 3) * F)
                    m, _, _ = vpatch.compute_merge(blk, vpatch.layer_norm(blk.norm1, h), blk._tome_info, want_indices=True,
                                                   materialize=False)
                    plan = m.plan
                    lvs = list(plan.levels) + ([plan.global_level] if plan.global_level is not None else [])
                    rec.append([(lv.unm_idx.cpu(), lv.src_idx.cpu(), lv.dst_idx.cpu()) for lv in lvs]
                               + [blk.global_tokens.float().cpu()])
        runs[mode] = rec
        vidtome_amd.remove_patch(unet)
    assert len(runs["exact"]) == len(runs["filtered"]) == 10
    for ra, rb in zip(runs["exact"], runs["filtered"]):
        assert len(ra) == len(rb)
        for la, lb in zip(ra[:-1], rb[:-1]):
            for ta, tb in zip(la, lb):
                assert torch.equal(ta, tb)
        assert torch.equal(ra[-1], rb[-1])
    modes = {m for m, _, _ in issued}
    assert len(issued) >= 24, len(issued)
    if regime == "corr01":
        assert (L.MATCH_SCOUT_RANGE, 1, True) in issued and (L.MATCH_SCOUT_RANGE, 1, False) in issued, sorted(set(issued))
    else:
        assert modes == {L.MATCH_SCOUT_RANGE, L.MATCH_ONE_LAUNCH}, sorted(set(issued))
        assert (L.MATCH_ONE_LAUNCH, 0, False) in issued[-6:], issued[-6:]      # out of the position order at the end


@pytest.mark.parametrize("shape", ["top_l2", "top_g", "mid_g"])
def test_match_row_slices_vs_oracle_at_full_shapes(L, oracle, shape):
    """The exact and the filtered HIP matcher against the CPU oracle on ROW SLICES at the full cfg-2 shapes the bench
    times (top level 2: 12 288 x 28 672 x 320, top global: 34 816^2 x 320, mid global: 8 704^2 x 640): frame-correlated
    fp16 tokens, slices at the start, across tile boundaries and at the end of the src range; node_max / node_idx
    bitwise for the slice rows, filtered == exact bitwise for ALL rows."""
    B, Ns, Nd, C = {"top_l2": (2, 12288, 28672, 320), "top_g": (2, 34816, 34816, 320), "mid_g": (2, 8704, 8704, 640)}[shape]
    g = torch.Generator().manual_seed(Ns + C)
    npos = 1024
    base = torch.randn(B, npos, C, generator=g)
    x = (base[:, torch.arange(Ns + Nd) % npos] + 0.5 * torch.randn(B, Ns + Nd, C, generator=g)).half()
    xd = x.to(DEV)
    ra = torch.arange(Ns, dtype=torch.int32, device=DEV).expand(B, Ns).contiguous()
    rb = torch.arange(Ns, Ns + Nd, dtype=torch.int32, device=DEV).expand(B, Nd).contiguous()
    a_op, _ = L.normalize_gather(xd, None, ra)
    b_op, _ = L.normalize_gather(xd, None, rb)
    best_e = L.match(a_op, b_op, Ns, Nd, False)
    best_f = L.match_filtered(xd, None, ra, rb, False)
    assert torch.equal(best_e, best_f)
    nm, ni = L.decode_best(best_f)
    nm, ni = nm.cpu().numpy(), ni.cpu().numpy()
    xf = x.float().numpy()
    an = oracle.normalize_gather(xf, np.arange(Ns)[None])
    bn = oracle.normalize_gather(xf, np.arange(Ns, Ns + Nd)[None])
    for r0, r1 in ((0, 96), (Ns // 2 - 40, Ns // 2 + 56), (Ns - 96, Ns)):
        onm, oni = oracle.match(an, bn, rows=(r0, r1))
        assert np.array_equal(_bits(onm[:, r0:r1]), _bits(nm[:, r0:r1])), (shape, r0)
        assert np.array_equal(oni[:, r0:r1], ni[:, r0:r1]), (shape, r0)


# ---------------------------------------------------------------------------------------------------
# apply_patch on the stand-in UNet vs the reference's recorded run
# ---------------------------------------------------------------------------------------------------
def _tie_aware_equal(level, ref_unm, ref_src, ref_dst, align):
    """exact, except inside groups of exactly equal node_max (see tests/test_oracle_golden.py)."""
    got = [getattr(level, n).cpu().numpy() for n in ("unm_idx", "src_idx", "dst_idx")]
    if all(np.array_equal(g, r) for g, r in zip(got, (ref_unm, ref_src, ref_dst))):
        return
    from vidtome_amd import _lib
    nm, ni = _lib.decode_best(level.best)
    nm, ni = nm.cpu().numpy(), ni.cpu().numpy()
    for b in range(ref_src.shape[0]):
        nmb, nib = (nm[0], ni[0]) if align else (nm[b], ni[b])
        assert np.array_equal(_bits(nmb[ref_src[b]]), _bits(nmb[got[1][b]]))
        assert np.array_equal(_bits(nmb[ref_unm[b]]), _bits(nmb[got[0][b]]))
        exp_dst = nib[ref_src[b]] % level.Nd if align else nib[ref_src[b]]
        assert np.array_equal(exp_dst, ref_dst[b])


@pytest.fixture(params=["by_geometry", "every_level"])
def order_rule(request, monkeypatch):
    """Which levels behind the first meet their rows in position order (merge.order_level): the product's rule (levels of
    >= 2^26 pairs -- none of the small fixtures), or every one (the fixtures then run through vtm_position_order +
    vtm_match_filtered_ordered end to end)."""
    from vidtome_amd import merge
    if request.param == "every_level":
        monkeypatch.setattr(merge, "POSITION_ORDER_MIN_PAIRS", 0)
    return request.param


@pytest.mark.parametrize("name", ["chain_cfg_f4", "chain_cfg_f8", "chain_pnp_f4", "chain_local_f4"])
def test_chain_golden_gpu(L, name, order_rule):
    import vidtome_amd
    from vidtome_amd import patch as vpatch
    from vidtome_amd import pnp
    from standin import Pipe, StandInUNet, load_block_weights

    cfg, z = load_chain(name)
    unet = load_block_weights(StandInUNet(cfg["C"], cfg["heads"]), z, DEV, torch.float32)
    pipe = Pipe(unet)
    if cfg["injection"] is not None:
        pnp.register_attention_control(pipe, cfg["injection"], cfg["B"])
        pnp.register_time(pipe, cfg["t"])
    vidtome_amd.apply_patch(unet, local_merge_ratio=cfg["local_ratio"], merge_global=cfg["merge_global"],
                            global_merge_ratio=cfg["global_ratio"], batch_size=cfg["B"], align_batch=cfg["align"],
                            target_stride=4, global_rand=0.5)
    blocks = list(unet.blocks())
    assert all(b.__class__.__name__ == "ToMeBlock" for b in blocks)
    torch.set_rng_state(torch.from_numpy(z["rng_state"]))       # what the block generators fork (patch.py:219)

    # capture every plan compute_merge builds
    plans = []
    orig = vpatch.compute_merge

    def rec(module, x, info, **kw):
        res = orig(module, x, info, want_indices=True)
        plans.append(getattr(res[0], "plan", None))
        return res

    vpatch.compute_merge = rec
    try:
        names = [str(s) for s in z["block_names"]]
        for ck, F in enumerate(cfg["chunk_frames"]):
            if ck in cfg.get("reset_before", []):
                vidtome_amd.update_patch(unet, global_tokens=None)             # generate.py:233-236
            plans.clear()
            hiddens = [_t(z[f"c{ck}/b{bi}/hidden"]) for bi in range(9)]
            with torch.no_grad():
                outs = unet(_t(z[f"c{ck}/latent"]), hiddens)
            ti = 0
            for bi in range(9):
                plan = plans[bi]
                ref_merged = z[f"c{ck}/b{bi}/merged"]
                if plan is not None:
                    for lv in plan.levels:
                        for n in ("unm_idx", "src_idx", "dst_idx"):
                            assert np.array_equal(getattr(lv, n).cpu().numpy(), z[f"c{ck}/t{ti}/{n}"]), (ck, bi, n)
                        ti += 1
                    if plan.global_level is not None:
                        _tie_aware_equal(plan.global_level, z[f"c{ck}/t{ti}/unm_idx"], z[f"c{ck}/t{ti}/src_idx"],
                                         z[f"c{ck}/t{ti}/dst_idx"], cfg["align"])
                        ti += 1
                    # merged tokens are row copies of the (torch) LayerNorm output: tight tolerance
                    np.testing.assert_allclose(plan.merged[:, :plan.M].cpu().numpy(), ref_merged, rtol=1e-5, atol=1e-6)
                # fp32 fixture: projections run in fp32, the attention core on the fp16 MFMA (fp32 accumulate):
                # the block output is held to north_star's 1e-3 (of the output scale)
                ref_out = z[f"c{ck}/b{bi}/out"]
                err = np.abs(outs[bi].cpu().numpy() - ref_out).max()
                assert err < 1e-3 * max(1.0, np.abs(ref_out).max()), (ck, bi, err)
            assert f"c{ck}/t{ti}/kind" not in z.files
            gts = vidtome_amd.collect_from_patch(unet, attr="global_tokens")
            for bi, nme in enumerate(names):
                key = f"c{ck}/gt/{nme}"
                if key in z.files:
                    np.testing.assert_allclose(gts[nme].cpu().numpy(), z[key], rtol=1e-5, atol=1e-6)
    finally:
        vpatch.compute_merge = orig
    vidtome_amd.remove_patch(unet)
    assert all(b.__class__.__name__ == "BasicTransformerBlock" for b in blocks)


@pytest.mark.parametrize("name,proj", [("chain16_cfg_f4_d40", "auto"), ("chain16_pnp_f4_d64", "auto"),
                                       ("chain16_pnp_f4_d64", "panels"), ("chain16_cfg_f8_d80", "auto"),
                                       ("chain16_cfg_f8_d80", "panels")])
def test_default_fp16_path_vs_reference_chain(L, name, proj, monkeypatch, order_rule):
    """The DEFAULT path of an fp16 model -- gather-fed projection GEMMs (C <= 320) / panel GEMMs (C = 640), live and
    compacted queries, the fp16 attention core -- against BLOCK OUTPUTS RECORDED FROM THE REFERENCE
    (tests/golden/make_golden_chain16.py: the reference's apply_patch + ToMeBlock.forward + sa_forward on its CPU fp32 path,
    weights and hidden states on the fp16 grid so that the fp16 model holds them exactly; cases screened so that the merge
    decisions do not hinge on what an fp16 rounding of norm1's output can move).  Multi-chunk chains with global merging,
    both coin outcomes, an anchor reset, a single-frame chunk, PnP batch 3 with aligned matching and shared probabilities,
    two local levels.  Block outputs and anchors within 1e-3 of the output scale (north_star's fp16 tolerance)."""
    import vidtome_amd
    from vidtome_amd import patch as vpatch
    from vidtome_amd import pnp
    from standin import Pipe, StandInUNet, load_block_weights

    monkeypatch.setattr(vpatch, "PROJ_MODE", proj)      # auto: gather-fed GEMMs (C <= 320); panels: the C >= 640 sites' path
    cfg, z = load_chain(name)
    unet = load_block_weights(StandInUNet(cfg["C"], cfg["heads"]), z, DEV, torch.float16)
    for k in z.files:                      # the weights ARE fp16 values: the fp16 model is the reference's model
        if k.startswith("w/up_blocks"):
            assert z[k].dtype == np.float16, k
    pipe = Pipe(unet)
    if cfg["injection"] is not None:
        pnp.register_attention_control(pipe, cfg["injection"], cfg["B"])
        pnp.register_time(pipe, cfg["t"])
    vidtome_amd.apply_patch(unet, local_merge_ratio=cfg["local_ratio"], merge_global=cfg["merge_global"],
                            global_merge_ratio=cfg["global_ratio"], batch_size=cfg["B"], align_batch=cfg["align"],
                            target_stride=4, global_rand=0.5)
    torch.set_rng_state(torch.from_numpy(z["rng_state"]))
    # the path under test really is the in-house one: count its projection launches
    calls = {"linear_rows": 0, "linear_panels": 0}
    orig = {n: getattr(L, n) for n in calls}

    def counted(n):
        def f(*a, **k):
            calls[n] += 1
            return orig[n](*a, **k)
        return f
    for n in calls:
        setattr(L, n, counted(n))
    try:
        names = [str(s) for s in z["block_names"]]
        worst = 0.0
        for ck, F in enumerate(cfg["chunk_frames"]):
            if ck in cfg.get("reset_before", []):
                vidtome_amd.update_patch(unet, global_tokens=None)             # generate.py:233-236
            hiddens = [_t(z[f"c{ck}/b{bi}/hidden"]) for bi in range(9)]       # stored as fp16
            assert all(h.dtype == torch.float16 for h in hiddens)
            latent = torch.zeros(tuple(int(v) for v in z[f"c{ck}/latent_shape"]), device=DEV, dtype=torch.float16)
            with torch.no_grad():
                outs = unet(latent, hiddens)
            for bi in range(9):
                ref_out = z[f"c{ck}/b{bi}/out"]
                err = np.abs(outs[bi].float().cpu().numpy() - ref_out).max() / max(1.0, np.abs(ref_out).max())
                worst = max(worst, err)
                assert err < 1e-3, (name, ck, bi, err)
            gts = vidtome_amd.collect_from_patch(unet, attr="global_tokens")
            for nme in names:
                key = f"c{ck}/gt/{nme}"
                if key in z.files:      # anchors are row copies of norm1's output (ours: rounded to fp16 once)
                    err = np.abs(gts[nme].float().cpu().numpy() - z[key]).max() / max(1.0, np.abs(z[key]).max())
                    assert err < 1e-3, (name, ck, nme, err)
        print(name, "worst block-output error / scale:", worst, calls)
    finally:
        for n in calls:
            setattr(L, n, orig[n])
    assert calls["linear_rows" if proj == "auto" else "linear_panels"] > 0, calls
    vidtome_amd.remove_patch(unet)


@pytest.mark.parametrize("name,mode", [("fullblock16_cfg_f4_d40", "panels"), ("fullblock16_cfg_f4_d40", "blas"),
                                       ("fullblock16_pnp_f4_d64", "panels"), ("fullblock16_pnp_f4_d64", "blas")])
def test_full_block_vs_reference_chain(L, name, mode, monkeypatch):
    """The WHOLE patched block against the reference's recorded `ToMeBlock.forward` (tests/golden/make_golden_fullblock.py):
    a full SD block -- norm1 / attn1 on merged tokens, norm2 / attn2 over 77 conditioning tokens (the reference's own
    `sa_forward` cross branch), norm3 / GEGLU feed-forward -- called by a stand-in Transformer2DModel with every keyword of
    patch.py:128-137, CPU fp32 reference on an fp16-grid model, multi-chunk chains with global merging (both coins, a
    single-frame chunk; PnP batch 3 with aligned matching and shared probabilities).  The fp16 model's patched forward runs
    the panel-GEMM path (default: vtm_layernorm_panels / vtm_linear_panels / vtm_ff_geglu / vtm_attention_kv) and, forced, the
    library-GEMM path.  Tolerance FULL_BLOCK_TOL of the output scale: the segment alone is held to north_star's 1e-3
    (test_default_fp16_path_vs_reference_chain); the full block adds two more fp16 residual roundings, a rounded
    LayerNorm -> GEMM chain twice over and the gated activation's three roundings on top of it."""
    import vidtome_amd
    from vidtome_amd import patch as vpatch
    from vidtome_amd import pnp
    from inputs import portable_weight
    from standin import Pipe, StandInUNet, load_block_weights

    monkeypatch.setattr(vpatch, "FF_MODE", mode)
    monkeypatch.setattr(vpatch, "PROJ_MODE", "auto" if mode == "panels" else "blas")
    monkeypatch.setattr(vpatch, "FUSED_PROJ", mode == "panels")
    cfg, z = load_chain(name)
    keep = cfg["keep_blocks"]
    unet = load_block_weights(StandInUNet(cfg["C"], cfg["heads"], True, cfg["cond_dim"]), z, DEV, torch.float16,
                              portable=portable_weight)
    pipe = Pipe(unet)
    if cfg["injection"] is not None:
        pnp.register_attention_control(pipe, cfg["injection"], cfg["B"])
        pnp.register_time(pipe, cfg["t"])
    vidtome_amd.apply_patch(unet, local_merge_ratio=cfg["local_ratio"], merge_global=cfg["merge_global"],
                            global_merge_ratio=cfg["global_ratio"], batch_size=cfg["B"], align_batch=cfg["align"],
                            target_stride=4, global_rand=0.5)
    torch.set_rng_state(torch.from_numpy(z["rng_state"]))
    calls = {"linear_panels": 0, "ff_geglu": 0, "layernorm_panels": 0, "attention_kv": 0}
    orig = {n: getattr(L, n) for n in calls}

    def counted(n):
        def f(*a, **k):
            calls[n] += 1
            return orig[n](*a, **k)
        return f
    for n in calls:
        setattr(L, n, counted(n))
    try:
        names = [str(s) for s in z["block_names"]]
        worst = 0.0
        for ck, F in enumerate(cfg["chunk_frames"]):
            if ck in cfg.get("reset_before", []):
                vidtome_amd.update_patch(unet, global_tokens=None)
            hiddens = [(_t(z[f"c{ck}/b{bi}/hidden"]) if bi in keep else None) for bi in range(9)]
            cond = _t(z[f"c{ck}/cond"])                                   # (B, 77, cond_dim) fp16, repeated over the frames
            cond = cond[:, None].expand(-1, F, -1, -1).reshape(cfg["B"] * F, cond.shape[1], cond.shape[2]).contiguous()
            latent = torch.zeros(tuple(int(v) for v in z[f"c{ck}/latent_shape"]), device=DEV, dtype=torch.float16)
            with torch.no_grad():
                outs = unet(latent, hiddens, encoder_hidden_states=cond, timestep=cfg["t"])
            for bi in keep:
                ref_out = z[f"c{ck}/b{bi}/out"]
                err = np.abs(outs[bi].float().cpu().numpy() - ref_out).max() / max(1.0, np.abs(ref_out).max())
                worst = max(worst, err)
                assert err < FULL_BLOCK_TOL, (name, mode, ck, bi, err)
            gts = vidtome_amd.collect_from_patch(unet, attr="global_tokens")
            for bi in keep:
                key = f"c{ck}/gt/{names[bi]}"
                if key in z.files:
                    ref = z[key].astype(np.float32)
                    err = np.abs(gts[names[bi]].float().cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max())
                    assert err < 1e-3, (name, ck, bi, err)
        print(name, mode, "worst full-block output error / scale:", worst, calls)
    finally:
        for n in calls:
            setattr(L, n, orig[n])
    if mode == "panels":        # the path under test really is the in-house one
        assert calls["ff_geglu"] > 0 and calls["layernorm_panels"] > 0 and calls["linear_panels"] > 0, calls
    assert calls["attention_kv"] > 0, calls
    vidtome_amd.remove_patch(unet)


def test_merge_modes_golden_gpu(L):
    """The closures' non-"replace" merge modes on the HIP path (vtm_merge_reduce: torch's CPU scatter_reduce arithmetic --
    sequential fp32 accumulation in index order, one rounding for 16-bit tokens, "mean" = rounded sum / (1 + sources),
    rounded again; NaN-propagating amax / amin) against what the REFERENCE's closures returned (tests/golden/modes.npz):
    bit for bit, fp32 / fp16 / bf16 tokens, local and global matcher, aligned batches."""
    from vidtome_amd import merge
    for c in load_cases("modes.npz"):
        x = _t(c["x"])
        if str(c["kind"]) == "randframe":
            torch.manual_seed(123)
            gen = torch.Generator(device="cpu").set_state(torch.get_rng_state())
            m, u, info = merge.bipartite_soft_matching_randframe(x, int(c["F"]), float(c["ratio"]), int(c["unm_pre"]), gen, 4,
                                                                 bool(c["align"]))
        else:
            m, u, info = merge.bipartite_soft_matching_2s(x, int(c["F"]), float(c["ratio"]), bool(c["align"]))
        for n in ("unm_idx", "src_idx", "dst_idx"):
            assert np.array_equal(info[n].cpu().numpy(), c[n]), n
        assert torch.equal(m(x, mode="replace"), m(x))
        for mode in ("sum", "prod", "mean", "amax", "amin"):
            got = m(x, mode=mode).cpu().numpy()
            want32 = c[f"f32/{mode}"]
            assert bool(((got.view(np.uint32) == want32.view(np.uint32)) | (np.isnan(got) & np.isnan(want32))).all()), \
                (str(c["kind"]), mode)
            for name, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
                goth = m(x.to(dt), mode=mode).cpu()
                want = torch.from_numpy(c[f"{name}/{mode}"]).view(dt)
                # bit for bit, except that a NaN is a NaN whatever its payload (one case carries a NaN token)
                same = (goth.view(torch.int16) == want.view(torch.int16)) | (goth.isnan() & want.isnan())
                assert bool(same.all()), (str(c["kind"]), name, mode, int((~same).sum()))
        with pytest.raises(ValueError):
            m(x, mode="median")


def test_device_generator_opt_in(L):
    """apply_patch(generator_device="device"): the reference's own generator rule on a GPU (vidtome/utils.py:24-25) -- the
    block generator is a CUDA generator forked from torch.cuda.get_rng_state(), and every draw of compute_merge (one
    randint per local level, merge.py:57-58; one rand per global level, patch.py:62) comes from it in the reference's
    order.  The default ("cpu") forks the CPU state on the same model."""
    import vidtome_amd
    from vidtome_amd import merge as vmerge
    from vidtome_amd import patch as vpatch
    from vidtome_amd import sites
    B, F, latent = 2, 8, (16, 16)
    site = sites.Site("s", 1, 320, 8)
    drawn = []
    orig_r, orig_c = vmerge.draw_randf, vpatch._draw_coin

    def rec_r(gen, ts):
        v = orig_r(gen, ts)
        drawn.append(("randf", gen.device.type, ts, v))
        return v

    def rec_c(gen):
        v = orig_c(gen)
        drawn.append(("coin", gen.device.type, None, v))
        return v
    vmerge.draw_randf, vpatch._draw_coin = rec_r, rec_c
    try:
        for mode in ("device", None):
            unet = sites.SiteUNet([site], seed=1).to(device=DEV, dtype=torch.float16)
            vidtome_amd.apply_patch(unet, local_merge_ratio=0.5, merge_global=True, global_merge_ratio=0.5, batch_size=B,
                                    generator_device=mode)
            unet.set_size(latent)
            torch.manual_seed(77)               # seeds the CPU and the CUDA default generators
            want_dev = torch.Generator(device=DEV).set_state(torch.cuda.get_rng_state())
            want_cpu = torch.Generator().set_state(torch.get_rng_state())
            drawn.clear()
            for ck in range(2):
                h = sites.synthetic_hidden(site, B, F, latent, torch.float16, DEV, seed=9 + ck, clip_seed=3)
                with torch.no_grad():
                    sites.run_segment_pass(unet, [h])
            g = want_dev if mode == "device" else want_cpu
            assert unet.blocks[0].generator.device.type == ("cuda" if mode == "device" else "cpu")
            assert [d[0] for d in drawn] == ["randf", "randf", "randf", "randf", "coin"]
            for kind, dev_type, ts, v in drawn:
                assert dev_type == g.device.type
                if kind == "randf":
                    assert v == int(torch.randint(0, ts, torch.Size([1]), generator=g, device=g.device))
                else:
                    assert v == float(torch.rand(1, generator=g, device=g.device))
            vidtome_amd.remove_patch(unet)
    finally:
        vmerge.draw_randf, vpatch._draw_coin = orig_r, orig_c


# ---------------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------------
def test_attention_golden_gpu(L):
    """softmax(QK^T)V through vtm_attention vs the reference's sa_forward outputs (fp16, 1e-3)."""
    from vidtome_amd import patch as vpatch
    from standin import Attention
    for c in load_cases("attention.npz"):
        B, h, M, d = int(c["B"]), int(c["heads"]), int(c["M"]), int(c["d"])
        attn = Attention(h * d, h)
        with torch.no_grad():
            attn.to_q.weight.copy_(torch.from_numpy(c["wq"].astype(np.float32)))
            attn.to_k.weight.copy_(torch.from_numpy(c["wk"].astype(np.float32)))
            attn.to_v.weight.copy_(torch.from_numpy(c["wv"].astype(np.float32)))
            attn.to_out[0].weight.copy_(torch.from_numpy(c["wo"].astype(np.float32)))
            attn.to_out[0].bias.copy_(torch.from_numpy(c["bo"].astype(np.float32)))
        attn = attn.to(DEV).half()
        if c["inject"]:
            attn.injection_schedule, attn.t, attn.vtm_num_inputs = [500], 500, B
        x = _t(c["x"]).half()
        with torch.no_grad():
            y = vpatch.self_attention(attn, x, M)[:, :M].float().cpu().numpy()
        err = np.abs(y[:, c["rows"], :] - c["y_rows"]).max()
        assert err < 1e-3 * max(1.0, np.abs(c["y_rows"]).max()), (d, err)


def _golden_attn(c):
    from standin import Attention
    h, d = int(c["heads"]), int(c["d"])
    attn = Attention(h * d, h)
    with torch.no_grad():
        attn.to_q.weight.copy_(torch.from_numpy(c["wq"].astype(np.float32)))
        attn.to_k.weight.copy_(torch.from_numpy(c["wk"].astype(np.float32)))
        attn.to_v.weight.copy_(torch.from_numpy(c["wv"].astype(np.float32)))
        attn.to_out[0].weight.copy_(torch.from_numpy(c["wo"].astype(np.float32)))
        attn.to_out[0].bias.copy_(torch.from_numpy(c["bo"].astype(np.float32)))
    return attn.to(DEV).half()


def test_attention_golden_gpu_gather_fed_path(L):
    """The DEFAULT fp16 path of the merged sites (`self_attention_rows`: q / k / v^T / out projections by vtm_linear_rows
    through a row map, attention core, nothing materialised) against the REFERENCE's recorded sa_forward outputs
    (attention.npz), 1e-3 of the output scale like the library-GEMM path above.  The tokens are presented the way the
    patched block presents them: scattered over a pool [x0 | x1] and addressed through a gather map; the non-injected
    cases also run with a live-query subset (`q_rows`), whose rows must equal the full result's."""
    from vidtome_amd import patch as vpatch
    for c in load_cases("attention.npz"):
        B, M = int(c["B"]), int(c["M"])
        attn = _golden_attn(c)
        if c["inject"]:
            attn.injection_schedule, attn.t, attn.vtm_num_inputs = [500], 500, B
        x = _t(c["x"]).half()
        C = x.shape[2]
        g = torch.Generator().manual_seed(M)
        perm = torch.stack([torch.randperm(M, generator=g) for _ in range(B)]).to(DEV)        # token i lives at pool row perm[i]
        P0 = M // 3
        pool = torch.empty_like(x)
        pool.scatter_(1, perm[:, :, None].expand(B, M, C), x)
        x0, x1 = pool[:, :P0].contiguous(), pool[:, P0:].contiguous()
        rows = perm.to(torch.int32).contiguous()
        with torch.no_grad():
            y = vpatch.self_attention_rows(attn, x0, x1, rows)[:, :M].float().cpu().numpy()
            y_id = vpatch.self_attention_rows(attn, x.contiguous(), None, None)[:, :M].float().cpu().numpy()
        scale = max(1.0, np.abs(c["y_rows"]).max())
        for got in (y, y_id):
            err = np.abs(got[:, c["rows"], :] - c["y_rows"]).max()
            assert err < 1e-3 * scale, (int(c["d"]), err)
        if not c["inject"]:
            q_rows = _t(np.ascontiguousarray(np.broadcast_to(c["rows"].astype(np.int32), (B, len(c["rows"])))))
            with torch.no_grad():
                yq = vpatch.self_attention_rows(attn, x0, x1, rows, q_rows)[:, :q_rows.shape[1]].float().cpu().numpy()
            assert np.abs(yq - c["y_rows"]).max() < 1e-3 * scale
            assert np.abs(yq - y[:, c["rows"], :]).max() < 2e-3 * scale     # same operands, another query tiling


E2E_CASES = {
    # name: F, N_side, C, heads, coins, proj, dtype, B, aligned + shared probabilities, ratios, tolerance
    "c320_d40": (8, 16, 320, 8, (0.0, 1.0), "auto", torch.float16, 2, False, (0.5, 0.5), 1e-3),
    "c320_d40_f16_n1024": (16, 32, 320, 8, (1.0, 0.0), "auto", torch.float16, 2, False, (0.5, 0.5), 1e-3),
    "c640_d80_panels": (8, 16, 640, 8, (0.0, 1.0), "auto", torch.float16, 2, False, (0.5, 0.5), 1e-3),
    "c640_d80_rows": (8, 16, 640, 8, (1.0, 0.0), "rows", torch.float16, 2, False, (0.5, 0.5), 1e-3),
    "c320_d40_panels": (8, 16, 320, 8, (0.0, 1.0), "panels", torch.float16, 2, False, (0.5, 0.5), 1e-3),
    "sd21_d64_ratio06": (8, 16, 320, 5, (0.0, 1.0), "auto", torch.float16, 2, False, (0.6, 0.6), 1e-3),        # cfg-5's head dim / ratios
    "pnp_b3_aligned_shared": (4, 16, 320, 8, (0.0, 1.0), "auto", torch.float16, 3, True, (0.5, 0.5), 1e-3),  # cfg-3
    "single_frame_chunks": (1, 16, 320, 8, (0.0, 1.0), "auto", torch.float16, 2, False, (0.5, 0.5), 1e-3),   # no local level
    "bf16": (8, 16, 320, 8, (1.0, 0.0), "auto", torch.bfloat16, 2, False, (0.5, 0.5), 8e-3),                 # bf16: 8 mantissa bits
    # duplicate-key folding: anchors with copies on the dst side (src, src), then on the src side (dst)
    "c320_d40_fold": (8, 16, 320, 8, (0.0, 0.0, 1.0), "auto", torch.float16, 2, False, (0.5, 0.5), 1e-3),
    "bf16_fold": (4, 16, 320, 8, (0.0, 0.0, 1.0), "auto", torch.bfloat16, 2, False, (0.5, 0.5), 8e-3),
}


@pytest.mark.parametrize("case", sorted(E2E_CASES))
def test_default_fp16_block_vs_oracle_end_to_end(L, oracle, case, monkeypatch):
    """What bench.py times, end to end, against the pinned oracle: an fp16 site with the bench's channel counts and head
    dims (C = 320 / d = 40: gather-fed projections + live / compacted queries; C = 640 / d = 80: panel-GEMM projections; and
    each of them forced onto the other path), three
    chunks of one clip (first chunk stores its tokens; then one chunk with the local chunk as src, one as dst -- the
    coin thresholds are switched between chunks).  The oracle starts from OUR norm1 output (vtm_layernorm has its own
    test against fp32 PyTorch; the matcher works on the fp16-rounded norm output, which is what the fp16 reference
    model's matcher sees too): merge indices of every level BIT-EQUAL, anchors equal, block output within 1e-3 of the
    output scale (north_star's fp16 tolerance)."""
    import vidtome_amd
    from vidtome_amd import patch as vpatch
    from vidtome_amd import sites
    F, N_side, C, heads, coins, proj, dtype, B, pnp, ratios, tol = E2E_CASES[case]
    monkeypatch.setattr(vpatch, "PROJ_MODE", proj)            # auto: gather-fed GEMMs at C = 320, panel GEMMs at C = 640
    latent = (N_side, N_side)
    site = sites.Site("s", 1, C, heads)
    unet = sites.SiteUNet([site], seed=3).to(device=DEV, dtype=dtype)
    if pnp:      # what pnp.register_attention_control + register_time set: shared probabilities at this timestep
        a_ = unet.blocks[0].attn1
        a_.injection_schedule, a_.t, a_.vtm_num_inputs = [500], 500, B
    with torch.no_grad():
        unet.blocks[0].norm1.weight.copy_(1.0 + 0.1 * torch.randn(C))
        unet.blocks[0].norm1.bias.copy_(0.1 * torch.randn(C))
        unet.blocks[0].attn1.to_out[0].bias.copy_(0.1 * torch.randn(C))
    vidtome_amd.apply_patch(unet, local_merge_ratio=ratios[0], merge_global=True, global_merge_ratio=ratios[1], batch_size=B,
                            global_rand=0.5, align_batch=pnp)
    unet.set_size(latent)
    blk = unet.blocks[0]
    torch.manual_seed(123)
    rng_state = torch.get_rng_state()
    blk.generator = torch.Generator().set_state(rng_state)
    draws = oracle.RandomDraws.from_torch_generator(torch.Generator().set_state(rng_state))
    f32 = lambda t: t.detach().float().cpu().numpy()
    a1 = blk.attn1
    w = {"wq": f32(a1.to_q.weight), "wk": f32(a1.to_k.weight), "wv": f32(a1.to_v.weight), "wo": f32(a1.to_out[0].weight),
         "bo": f32(a1.to_out[0].bias)}
    plans, norms = [], []
    orig_cm, orig_ln = vpatch.compute_merge, vpatch.layer_norm

    def rec_cm(module, x, info, **kw):
        kw["want_indices"] = True
        res = orig_cm(module, x, info, **kw)
        plans.append(res[0].plan)
        return res

    def rec_ln(norm, x):
        y = orig_ln(norm, x)
        norms.append(y)
        return y
    monkeypatch.setattr(vpatch, "compute_merge", rec_cm)
    monkeypatch.setattr(vpatch, "layer_norm", rec_ln)
    state = {"global_tokens": None}
    args = dict(unet._tome_info["args"])
    seen = set()
    folded = 0
    for ck in range(1 + len(coins)):
        if ck > 0:
            unet._tome_info["args"]["global_rand"] = args["global_rand"] = coins[ck - 1]
        hidden = sites.synthetic_hidden(site, B, F, latent, dtype, DEV, seed=50 + ck, clip_seed=7)
        with torch.no_grad():
            out = sites.run_segment_pass(unet, [hidden])[0]
        plan, nh = plans[-1], f32(norms[-1])
        m_o, u_o, merged_o, trace = oracle.compute_merge(nh, latent, args, draws, state)
        assert len(plan.levels) == len(trace["levels"]) == (0 if F == 1 else 2 if F > 4 else 1)
        for lv, rl in zip(plan.levels, trace["levels"]):
            for n in ("unm_idx", "src_idx", "dst_idx"):
                assert np.array_equal(getattr(lv, n).cpu().numpy(), rl[n]), (ck, n)
        assert (plan.global_level is None) == (trace["global"] is None) == (ck == 0)
        if trace["global"] is not None:
            seen.add(trace["global"]["local_chunk"])
            assert plan.local_chunk == trace["global"]["local_chunk"]
            _tie_aware_equal(plan.global_level, trace["global"]["unm_idx"], trace["global"]["src_idx"],
                             trace["global"]["dst_idx"], pnp)
        assert np.array_equal(f32(blk.global_tokens), state["global_tokens"]), ck      # row copies of the 16-bit pool
        attn_o = oracle.self_attention(merged_o, w["wq"], w["wk"], w["wv"], w["wo"], w["bo"], heads,
                                       share_groups=B if pnp else 1)
        ref = u_o(attn_o) + f32(hidden)
        err = np.abs(f32(out) - ref).max()
        assert err < tol * max(1.0, np.abs(ref).max()), (ck, err)
        if plan.fold_args is not None and proj == "auto" and C == 320 and heads == 8 and not pnp:
            # the block behind a local-is-src block: its anchors carried content ids, attn1 ran over the folded key list
            assert plan._key_fold is not None
            key_sel, k_bias, k_count = plan._key_fold
            cur = plan.gather_map.cpu().numpy()
            kc = k_count.cpu().numpy()
            for b in range(B):      # every dropped key is an exact copy of a kept one (same pool row content)
                pool = np.concatenate([f32(plan.x_joined[b]), f32(plan.anchors_in[b])])
                rows_all = pool[cur[b]]
                kept = key_sel[b, :kc[b]].cpu().numpy()
                assert np.all(np.diff(kept) > 0)
                uniq = {r.tobytes() for r in rows_all[kept]}
                assert all(r.tobytes() in uniq for r in rows_all)
            folded += int((kc < plan.M).any())
    assert seen == {0, 1}
    if case.endswith("_fold"):
        assert folded >= 2          # copies existed on both kinds of pass (otherwise the case tests nothing)
    vidtome_amd.remove_patch(unet)


def test_block_vs_oracle_from_the_oracles_own_layernorm(L, oracle, monkeypatch):
    """norm1 -> matcher -> attention -> unmerge end to end with NOTHING shared between the two sides but the fp16 hidden
    states and weights: the oracle applies its own LayerNorm (fp32, rounded once to fp16 -- what an fp16 reference model's
    norm1 returns), the HIP path runs vtm_layernorm (test_default_fp16_block_vs_oracle_end_to_end hands OUR norm1 output to
    the oracle).  Two correct fp16 LayerNorms may round a few elements differently and a merge decision can hinge on one, so
    the clip seed is screened offline (tests/golden/screen_ln_seed.py: the oracle's outputs and anchors stay within 3e-4
    when a random 2e-4 of its LayerNorm outputs move by one ulp) and the case is small -- at the bench's sizes every one of
    400 seeds has SOME rank-r decision that flips, which replaces whole tokens.  Four chunks of 4 frames x 6 x 6 tokens,
    C = 320, 8 heads, three global levels (src, dst, src): block outputs and anchors within 1e-3 of their scale."""
    import screen_ln_seed as case
    import vidtome_amd
    from vidtome_amd import sites
    assert (case.B, case.C, case.HEADS) == (2, 320, 8)
    seed = case.KEPT_SEED
    ref = case.outputs(seed)                                   # [block out, anchors] per chunk, CPU oracle
    w, b, wts, bo = case.case_weights(seed)
    site = sites.Site("s", 1, case.C, case.HEADS)
    unet = sites.SiteUNet([site], seed=3).to(device=DEV, dtype=torch.float16)
    blk = unet.blocks[0]
    with torch.no_grad():
        blk.norm1.weight.copy_(_t(w))
        blk.norm1.bias.copy_(_t(b))
        for name, lin in (("wq", blk.attn1.to_q), ("wk", blk.attn1.to_k), ("wv", blk.attn1.to_v), ("wo", blk.attn1.to_out[0])):
            lin.weight.copy_(_t(wts[name]))
        blk.attn1.to_out[0].bias.copy_(_t(bo))
    vidtome_amd.apply_patch(unet, local_merge_ratio=0.5, merge_global=True, global_merge_ratio=0.5, batch_size=case.B,
                            global_rand=0.5)
    unet.set_size(case.LATENT)
    torch.manual_seed(123)
    blk.generator = torch.Generator().set_state(torch.get_rng_state())
    # how far apart the two LayerNorms are: a handful of elements, one ulp each
    hidden0 = sites.synthetic_hidden(site, case.B, case.F, case.LATENT, torch.float16, "cpu", seed=50, clip_seed=seed)
    ours = L.layernorm(hidden0.to(DEV), blk.norm1.weight, blk.norm1.bias, blk.norm1.eps).cpu()
    theirs = case.ln_fp16(hidden0, w, b)
    differ = (ours.view(torch.int16) != theirs.view(torch.int16))
    assert differ.float().mean().item() < 2e-3
    assert ((ours.float() - theirs.float()).abs() <= theirs.float().abs() * 2.0 ** -10 + 1e-6).all()
    worst = 0.0
    for ck in range(1 + len(case.COINS)):
        if ck > 0:
            unet._tome_info["args"]["global_rand"] = case.COINS[ck - 1]
        hidden = sites.synthetic_hidden(site, case.B, case.F, case.LATENT, torch.float16, DEV, seed=50 + ck, clip_seed=seed)
        with torch.no_grad():
            out = sites.run_segment_pass(unet, [hidden])[0]
        for got, want in ((out, ref[2 * ck]), (blk.global_tokens, ref[2 * ck + 1])):
            err = np.abs(got.float().cpu().numpy().reshape(want.shape) - want).max() / max(1.0, np.abs(want).max())
            worst = max(worst, err)
            assert err < 1e-3, (ck, err)
    print("own-LayerNorm case: worst error / scale", worst, "LayerNorm elements that differ:", int(differ.sum()), "of", differ.numel())
    vidtome_amd.remove_patch(unet)


@pytest.mark.parametrize("shape", [(2, 3, 200, 40), (1, 2, 333, 80), (2, 2, 129, 64), (1, 1, 70, 160), (3, 2, 64, 8)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_attention_core_vs_oracle(L, oracle, shape, dtype):
    B, h, M, d = shape
    C = h * d
    g = torch.Generator().manual_seed(M)
    q = torch.randn(B, M, C, generator=g).to(dtype)
    k = torch.randn(B, M, C, generator=g).to(dtype)
    v = torch.randn(B, M, C, generator=g).to(dtype)
    Mp = (M + 7) // 8 * 8
    pad = lambda t: torch.nn.functional.pad(t, (0, 0, 0, Mp - M))
    qd, kd = pad(q).to(DEV), pad(k).to(DEV)
    vt = pad(v).to(DEV).transpose(1, 2).contiguous()
    for share in (1, B) if B > 1 else (1,):
        o = L.attention(qd, kd, vt, h, M, d ** -0.5, share)[:, :M].float().cpu().numpy()
        ref = oracle.attention(q.float().numpy(), k.float().numpy(), v.float().numpy(), h, share_groups=share)
        tol = 1e-3 if dtype == torch.float16 else 8e-3
        assert np.abs(o - ref).max() < tol * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("shape", [(32, 256, 3840, 2560, 1280), (3, 70, 192, 64, 64), (2, 1000, 960, 320, 640)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_transpose_cols_vs_torch(L, shape, dtype):
    """vtm_transpose_cols: the V columns of a fused q | k | v projection, channel-major, token padding zero-filled."""
    BF, N, ld, c0, C = shape
    x = torch.randn(BF, N, ld, generator=torch.Generator().manual_seed(N)).to(dtype).to(DEV)
    out = L.transpose_cols(x, c0, C)
    Np = (N + 7) // 8 * 8
    assert out.shape == (BF, C, Np)
    assert torch.equal(out[:, :, :N], x[:, :, c0:c0 + C].transpose(1, 2))
    assert not out[:, :, N:].any()


def _fold_reference(cur, L, cid):
    """numpy restatement of vtm_fold_keys: first copy of every content id survives, with the number of copies present."""
    B, M = cur.shape
    sel, cnt = [], []
    for b in range(B):
        first, n = {}, {}
        for m in range(M):
            if cur[b, m] >= L:
                c = int(cid[b, cur[b, m] - L])
                first.setdefault(c, m)
                n[c] = n.get(c, 0) + 1
        keep = [m for m in range(M) if cur[b, m] < L or first[int(cid[b, cur[b, m] - L])] == m]
        sel.append(keep)
        cnt.append([1 if cur[b, m] < L else n[int(cid[b, cur[b, m] - L])] for m in keep])
    return sel, cnt


@pytest.mark.parametrize("shape", [(2, 500, 300, 260), (1, 70, 10, 64), (3, 4099, 1000, 3000)])
def test_fold_keys_vs_numpy(L, shape):
    """vtm_fold_keys (the anchors' exact copies -> one key each): surviving positions, counts and the log2 bias pair."""
    B, M, Lrows, Ma = shape
    rng = np.random.default_rng(M)
    n_ids = Ma
    cid = rng.integers(0, max(2, Ma // 3), size=(B, Ma)).astype(np.int32)
    cur = np.stack([rng.permutation(Lrows + Ma)[:M] for _ in range(B)]).astype(np.int32)
    for dtype in (torch.float16, torch.bfloat16):
        key_sel, k_bias, k_count = L.fold_keys(_t(cur), Lrows, _t(cid), n_ids, dtype)
        sel, cnt = _fold_reference(cur, Lrows, cid)
        kb = k_bias.cpu().numpy().view(np.uint32)
        for b in range(B):
            n = int(k_count[b])
            assert n == len(sel[b])
            assert np.array_equal(key_sel[b, :n].cpu().numpy(), np.array(sel[b], dtype=np.int32))
            assert not key_sel[b, n:].any()
            words = torch.from_numpy(kb[b, :n].astype(np.int64))
            hi = (words & 0xffff).to(torch.int16).view(dtype).float().numpy()
            lo = (words >> 16).to(torch.int16).view(dtype).float().numpy()
            want = np.log2(np.array(cnt[b], dtype=np.float64))
            assert np.abs(hi + lo - want).max() < (2e-6 if dtype == torch.float16 else 4e-5)
            assert np.all((hi + lo)[np.array(cnt[b]) == 1] == 0.0)


@pytest.mark.parametrize("dtype,d", [(torch.float16, 40), (torch.bfloat16, 40), (torch.float16, 8)])
def test_attention_folded_keys_equal_the_duplicated_ones(L, oracle, dtype, d, monkeypatch):
    """vtm_attention_kv_folded: m identical keys weigh like one key with + log2(m) on its score -- the folded launch (device-side
    key count, bias pair in the spare k-slots) against the oracle's attention over the sequence WITH the copies, and against
    the unfolded kernel on the same sequence; with and without a device-side query count, ragged last tile, split tails."""
    B, h, Mq, Mu = 2, 8, 333, 1500
    C = h * d
    g = torch.Generator().manual_seed(11)
    q = torch.randn(B, Mq, C, generator=g).to(dtype)
    ku = torch.randn(B, Mu, C, generator=g).to(dtype)
    vu = torch.randn(B, Mu, C, generator=g).to(dtype)
    rng = np.random.default_rng(5)
    mult = rng.choice([1, 1, 1, 2, 3, 5, 17], size=(B, Mu))
    mult[1, Mu - 200:] = 0                        # sample 1 has fewer distinct keys than sample 0: per-sample device counts
    Mk = int(mult.sum(1).max())
    Mkp, Mup, Mqp = (Mk + 7) // 8 * 8, (Mu + 7) // 8 * 8, (Mq + 7) // 8 * 8
    kd = torch.zeros(B, Mkp, C, dtype=dtype)
    vd = torch.zeros(B, Mkp, C, dtype=dtype)
    counts_full = []
    for b in range(B):
        idx = np.repeat(np.arange(Mu), mult[b])
        idx = idx[rng.permutation(len(idx))]
        counts_full.append(len(idx))
        kd[b, :len(idx)] = ku[b, idx]
        vd[b, :len(idx)] = vu[b, idx]
    pad = lambda t, n: torch.nn.functional.pad(t, (0, 0, 0, n - t.shape[1]))
    qd = pad(q, Mqp).to(DEV)
    scale = d ** -0.5
    # folded operands: the distinct keys (those with mult > 0) in front, bias words, counts
    kf = torch.zeros(B, Mup, C, dtype=dtype)
    vf = torch.zeros(B, Mup, C, dtype=dtype)
    bias = torch.zeros(B, Mup, dtype=torch.int32)
    kc = []
    for b in range(B):
        live = np.nonzero(mult[b])[0]
        kc.append(len(live))
        kf[b, :len(live)] = ku[b, live]
        vf[b, :len(live)] = vu[b, live]
        kf[b, len(live):] = 7.0                   # rows past the count: finite garbage nobody may read as a key
        lg = torch.log2(torch.from_numpy(mult[b, live].astype(np.float32)))
        hi = lg.to(dtype)
        lo = (lg - hi.float()).to(dtype)
        bias[b, :len(live)] = (hi.view(torch.int16).int() & 0xffff) | (lo.view(torch.int16).int() << 16)
    k_count = torch.tensor(kc, dtype=torch.int32, device=DEV)
    tol = (1e-3 if dtype == torch.float16 else 8e-3)
    for b in range(B):      # the oracle on each sample's own duplicated sequence
        n = counts_full[b]
        ref = oracle.attention_qkv(q[b:b + 1].float().numpy(), kd[b:b + 1, :n].float().numpy(), vd[b:b + 1, :n].float().numpy(), h)
        of = L.attention_kv(qd, kf.to(DEV), vf.to(DEV).transpose(1, 2).contiguous(), h, Mq, Mu, scale,
                            k_fold=(k_count, bias.to(DEV)))[b, :Mq].float().cpu().numpy()
        assert np.abs(of - ref[0]).max() < tol * max(1.0, np.abs(ref).max()), b
        ou = L.attention_kv(qd[b:b + 1], kd[b:b + 1].to(DEV), vd[b:b + 1].to(DEV).transpose(1, 2).contiguous(), h, Mq, n,
                            scale)[0, :Mq].float().cpu().numpy()
        assert np.abs(of - ou).max() < tol * max(1.0, np.abs(ref).max()), b
    # + a device-side query count (the split-all launch plan reads the device-side key count in every split)
    q_count = torch.tensor([Mq, Mq - 77], dtype=torch.int32, device=DEV)
    oq = L.attention_kv(qd, kf.to(DEV), vf.to(DEV).transpose(1, 2).contiguous(), h, Mq, Mu, scale, q_count=q_count,
                        k_fold=(k_count, bias.to(DEV)))
    of = L.attention_kv(qd, kf.to(DEV), vf.to(DEV).transpose(1, 2).contiguous(), h, Mq, Mu, scale, k_fold=(k_count, bias.to(DEV)))
    # (round 6: a query-bounded d = 40 launch is planned on the device -- this small one is key-split to fill the chip, so
    # the two launches sum the keys in different orders; with the bounded workspace withheld both take the same plan)
    sc = max(1.0, float(of.float().abs().max()))
    for b in range(B):
        n = int(q_count[b])
        assert (oq[b, :n].float() - of[b, :n].float()).abs().max() < tol * sc, b
    monkeypatch.setattr(L, "SPLIT_ALL_BOUNDED", False)
    oq0 = L.attention_kv(qd, kf.to(DEV), vf.to(DEV).transpose(1, 2).contiguous(), h, Mq, Mu, scale, q_count=q_count,
                         k_fold=(k_count, bias.to(DEV)))
    for b in range(B):
        n = int(q_count[b])
        assert torch.equal(oq0[b, :n], of[b, :n])


def test_attention_folded_keys_under_the_split_plans(L):
    """The launch plans that split work items along the key axis (every item in two for query-bounded launches, the last
    round otherwise) cut the DEVICE-side key count, not the host bound: a folded launch big enough to take them against
    the unfolded kernel over the sequence with the copies (kernel against kernel; the oracle anchors the small case)."""
    B, h, d, Mq, Mu = 2, 8, 40, 16640, 4800
    C = h * d
    dtype = torch.float16
    g = torch.Generator(device=DEV).manual_seed(3)
    q = torch.randn(B, Mq, C, generator=g, device=DEV).to(dtype)
    ku = torch.randn(B, Mu, C, generator=g, device=DEV).to(dtype)
    vu = torch.randn(B, Mu, C, generator=g, device=DEV).to(dtype)
    rng = np.random.default_rng(9)
    mult = rng.choice([1, 1, 2, 4], size=(B, Mu))
    mult[1, Mu - 333:] = 0
    Mk = int(mult.sum(1).max())
    Mkp = (Mk + 7) // 8 * 8
    scale = d ** -0.5
    bias = torch.zeros(B, Mu, dtype=torch.int32)
    kf, vf = torch.zeros_like(ku), torch.zeros_like(vu)
    kc, outs_u = [], []
    for b in range(B):
        live = np.nonzero(mult[b])[0]
        kc.append(len(live))
        kf[b, :len(live)] = ku[b, live]
        vf[b, :len(live)] = vu[b, live]
        lg = torch.log2(torch.from_numpy(mult[b, live].astype(np.float32)))
        hi = lg.to(dtype)
        lo = (lg - hi.float()).to(dtype)
        bias[b, :len(live)] = (hi.view(torch.int16).int() & 0xffff) | (lo.view(torch.int16).int() << 16)
        idx = torch.from_numpy(np.repeat(np.arange(Mu), mult[b])).to(DEV)
        n = len(idx)
        kd = torch.zeros(1, Mkp, C, dtype=dtype, device=DEV)
        vd = torch.zeros(1, Mkp, C, dtype=dtype, device=DEV)
        kd[0, :n], vd[0, :n] = ku[b, idx], vu[b, idx]
        outs_u.append(L.attention_kv(q[b:b + 1], kd, vd.transpose(1, 2).contiguous(), h, Mq, n, scale)[0].float())
    k_count = torch.tensor(kc, dtype=torch.int32, device=DEV)
    vft = vf.transpose(1, 2).contiguous()
    q_count = torch.tensor([Mq, Mq - 1000], dtype=torch.int32, device=DEV)
    o_plain = L.attention_kv(q, kf, vft, h, Mq, Mu, scale, k_fold=(k_count, bias.to(DEV))).float()
    o_bound = L.attention_kv(q, kf, vft, h, Mq, Mu, scale, q_count=q_count, k_fold=(k_count, bias.to(DEV))).float()
    for b in range(B):
        n = int(q_count[b])
        sc = max(1.0, float(outs_u[b].abs().max()))
        assert float((o_plain[b] - outs_u[b]).abs().max()) < 1e-3 * sc
        assert float((o_bound[b, :n] - outs_u[b][:n]).abs().max()) < 1e-3 * sc


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows,C,affine", [(37, 320, True), (1000, 640, True), (5, 1280, True), (3, 8, False), (64, 2048, True)])
def test_layernorm_vs_torch_fp32(L, rows, C, affine, dtype):
    """norm1 (patch.py:146): vtm_layernorm against a plain PyTorch fp32 LayerNorm of the same (rounded) input."""
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * 3.0 + 0.7).to(dtype)
    w = (1.0 + 0.2 * torch.randn(C, generator=g)).to(dtype) if affine else None
    b = (0.1 * torch.randn(C, generator=g)).to(dtype) if affine else None
    ref = torch.nn.functional.layer_norm(x.float(), (C,), w.float() if affine else None, b.float() if affine else None, 1e-5)
    y = L.layernorm(x.to(DEV), w.to(DEV) if affine else None, b.to(DEV) if affine else None, 1e-5).float().cpu()
    tol = {torch.float32: 2e-6, torch.float16: 1e-3, torch.bfloat16: 8e-3}[dtype]
    assert (y - ref).abs().max() <= tol * max(1.0, float(ref.abs().max()))
    if dtype != torch.float32:   # at most one rounding step away from the correctly rounded fp32 result
        assert ((y - ref.to(dtype).float()).abs() <= (ref.abs() * 2.0 ** (-10 if dtype == torch.float16 else -7) + 1e-6)).all()


# ---------------------------------------------------------------------------------------------------
# the rest of the patched block (patch.py:171-199): cross-attention and the GEGLU feed-forward
# ---------------------------------------------------------------------------------------------------
class GEGLU(torch.nn.Module):                       # named like the Diffusers class the host code recognises
    def __init__(self, C, inner):
        super().__init__()
        self.proj = torch.nn.Linear(C, 2 * inner)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * torch.nn.functional.gelu(gate)


class _FeedForward(torch.nn.Module):
    def __init__(self, C):
        super().__init__()
        self.net = torch.nn.ModuleList([GEGLU(C, 4 * C), torch.nn.Dropout(0.0), torch.nn.Linear(4 * C, C)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_geglu_vs_torch_fp32(L, dtype):
    g = torch.Generator().manual_seed(5)
    x = (2.0 * torch.randn(53, 2 * 320, generator=g)).to(dtype)
    h, gate = x.float().chunk(2, dim=-1)
    ref = h * torch.nn.functional.gelu(gate)
    y = L.geglu(x.to(DEV)).float().cpu()
    tol = {torch.float32: 2e-6, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
    assert (y - ref).abs().max() <= tol * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,P0,P1,K,N,n", [(2, 300, 100, 64, 96, 257), (1, 128, 0, 32, 128, 128), (3, 50, 77, 320, 320, 91),
                                            (2, 4096, 1000, 640, 640, 3000),
                                            # K = 320: the weight-stationary kernel -- several blocks per wave, ragged last block,
                                            # more than one 160-channel half per sample
                                            (2, 60000, 3000, 320, 320, 40013), (1, 2000, 0, 320, 640, 1500)])
def test_linear_rows_vs_torch_fp32(L, dtype, B, P0, P1, K, N, n):
    """vtm_linear_rows = Linear(gather(pool, rows)): both output layouts, one- and two-level maps, bias, ragged tiles."""
    g = torch.Generator().manual_seed(K + n)
    x0 = torch.randn(B, P0, K, generator=g).to(dtype).to(DEV)
    x1 = torch.randn(B, P1, K, generator=g).to(dtype).to(DEV) if P1 else None
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype).to(DEV)
    bias = torch.randn(N, generator=g).to(dtype).to(DEV)
    pool = x0 if x1 is None else torch.cat([x0, x1], dim=1)
    Mfull = max(n, (P0 + P1) // 2)
    rows = torch.randint(0, P0 + P1, (B, Mfull), generator=g).to(torch.int32).to(DEV)
    rows2 = torch.randint(0, Mfull, (B, n), generator=g).to(torch.int32).to(DEV)
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2

    def ref(idx, b_):
        a = torch.gather(pool.float(), 1, idx.long().unsqueeze(-1).expand(-1, -1, K))
        y = a @ w.float().t()
        return y + b_.float() if b_ is not None else y

    def check(got, want):
        err = (got.float() - want).abs().max().item()
        assert err < tol * max(1.0, want.abs().max().item()), err

    n_pad = (n + 7) // 8 * 8
    # one-level map, token-major, with and without bias
    for b_ in (None, bias):
        out = L.linear_rows(x0, x1, rows[:, :n].contiguous(), None, n, w, b_)
        assert out.shape == (B, n_pad, N)
        check(out[:, :n], ref(rows[:, :n], b_))
    # two-level map (live-query rows), channel-major (V^T layout)
    comp = torch.gather(rows.long(), 1, rows2.long())
    out_t = L.linear_rows(x0, x1, rows, rows2, n, w, bias, transposed=True)
    assert out_t.shape == (B, N, n_pad)
    check(out_t[:, :, :n].transpose(1, 2), ref(comp, bias))
    # identity rows over x0 (the to_out projection of the attention output), written into a strided view
    m = min(n, P0)
    buf = torch.zeros(B, (m + 7) // 8 * 8, 2 * N, dtype=dtype, device=DEV)
    L.linear_rows(x0, None, None, None, m, w, None, out=buf[:, :, N:])
    check(buf[:, :m, N:], x0[:, :m].float() @ w.float().t())
    assert (buf[:, :, :N] == 0).all()
    # pool ids through rows2 alone
    out2 = L.linear_rows(x0, x1, None, comp.to(torch.int32).contiguous(), n, w, None)
    check(out2[:, :n], ref(comp, None))


@pytest.mark.parametrize("B,N,Mk,C,heads", [(4, 256, 77, 320, 8), (2, 100, 77, 640, 8), (3, 64, 10, 160, 2)])
def test_cross_attention_and_ff_vs_torch_fp32(L, B, N, Mk, C, heads):
    """attn2 / ff of the patched block (patch.py:171-199) against plain PyTorch fp32 on the same rounded inputs."""
    from standin import Attention
    from vidtome_amd import patch as vpatch
    torch.manual_seed(N + C)
    attn = Attention(C, heads).eval()
    ff = _FeedForward(C).eval()
    x = torch.randn(B, N, C).half()
    enc = torch.randn(B, Mk, C).half()
    with torch.no_grad():
        # fp32 reference with the weights rounded to fp16 like the device copy
        attn_h, ff_h = attn.half(), ff.half()
        w = lambda m: m.weight.float()
        q, k, v = x.float() @ w(attn_h.to_q).t(), enc.float() @ w(attn_h.to_k).t(), enc.float() @ w(attn_h.to_v).t()
        q, k, v = (t.half().float().reshape(B, -1, heads, C // heads).transpose(1, 2) for t in (q, k, v))
        p = torch.softmax(q @ k.transpose(-1, -2) * attn_h.scale, dim=-1)
        o = (p @ v).transpose(1, 2).reshape(B, N, C)
        ref_attn = o @ w(attn_h.to_out[0]).t() + attn_h.to_out[0].bias.float()
        y = vpatch.cross_attention(attn_h.to(DEV), x.to(DEV), enc.to(DEV)).float().cpu()
        assert (y - ref_attn).abs().max() < 2e-3 * max(1.0, float(ref_attn.abs().max()))
        ref_ff = _FeedForward(C).eval()
        ref_ff.load_state_dict({k_: v_.float() for k_, v_ in ff_h.state_dict().items()})
        r = ref_ff(x.float())
        z = vpatch.feed_forward(ff_h.to(DEV), x.to(DEV)).float().cpu()
        assert (z - r).abs().max() < 4e-3 * max(1.0, float(r.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows,C", [(300, 320), (1000, 640), (70, 1280), (512, 64)])
def test_layernorm_panels_and_to_panels(L, rows, C, dtype):
    """vtm_layernorm_panels == vtm_layernorm bit for bit, laid out as k-panels [C / 8][rows_pad][8]; vtm_to_panels with and
    without a row order."""
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g).to(dtype).to(DEV)
    w, b = (1 + 0.1 * torch.randn(C, generator=g)).to(dtype).to(DEV), (0.1 * torch.randn(C, generator=g)).to(dtype).to(DEV)
    y = L.layernorm(x, w, b, 1e-5)
    yp = L.layernorm_panels(x, w, b, 1e-5)
    assert yp.shape == (C // 8, L.panel_rows(rows), 8)
    assert torch.equal(yp[:, :rows].permute(1, 0, 2).reshape(rows, C), y)
    xp = L.to_panels(x)
    assert torch.equal(xp[:, :rows].permute(1, 0, 2).reshape(rows, C), x) and not xp[:, rows:].any()
    order = torch.randperm(rows, generator=g)[: rows // 2].to(torch.int32)
    order[3] = -1
    op = L.to_panels(x, order.to(DEV))
    want = x[order.long().clamp(min=0)]
    want[3] = 0
    assert torch.equal(op[:, :rows // 2].permute(1, 0, 2).reshape(rows // 2, C), want)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows,C,bias", [(300, 320, True), (1000, 640, True), (70, 1280, False), (4096, 320, True), (513, 64, True)])
def test_ff_panels_vs_torch_fp32(L, rows, C, bias, dtype, monkeypatch):
    """`ff(norm3(h)) + h` (patch.py:187-199) as LayerNorm -> panels, GEGLU projection with the gated activation in the
    GEMM's epilogue, output Linear + bias + residual -- against plain PyTorch fp32 on the same rounded inputs / weights, and
    against the unfused library path of round 2 (same roundings of the projection, gelu and product)."""
    from vidtome_amd import patch as vpatch
    torch.manual_seed(rows + C)
    ff = _FeedForward(C).eval()
    norm = torch.nn.LayerNorm(C)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.1 * torch.randn(C))
        norm.bias.copy_(0.1 * torch.randn(C))
        if not bias:
            ff.net[0].proj.bias = None
            ff.net[2].bias = None
    ff, norm = ff.to(dtype), norm.to(dtype)
    h = (1.5 * torch.randn(2, rows // 2 if rows % 2 == 0 else rows, C)).to(dtype)
    with torch.no_grad():
        ref_ff, ref_norm = _FeedForward(C).eval(), torch.nn.LayerNorm(C)
        if not bias:
            ref_ff.net[0].proj.bias = None
            ref_ff.net[2].bias = None
        ref_ff.load_state_dict({k_: v_.float() for k_, v_ in ff.state_dict().items()})
        ref_norm.load_state_dict({k_: v_.float() for k_, v_ in norm.state_dict().items()})
        ref = ref_ff(ref_norm(h.float())) + h.float()
        ffd, nd, hd = ff.to(DEV), norm.to(DEV), h.to(DEV)
        assert vpatch.fused_ff_ok(nd, ffd, hd)
        y = vpatch.norm_feed_forward_residual(nd, ffd, hd)
        z = vpatch.feed_forward(ffd, vpatch.layer_norm(nd, hd)) + hd                # library GEMMs + vtm_geglu
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    scale = max(1.0, float(ref.abs().max()))
    assert y.shape == h.shape
    assert (y.float().cpu() - ref).abs().max() < tol * scale
    assert (y.float() - z.float()).abs().max().item() < (2e-3 if dtype == torch.float16 else 2e-2) * scale


@pytest.mark.parametrize("B,N,Mk,C,heads", [(4, 256, 77, 320, 8), (2, 104, 77, 640, 8), (16, 64, 77, 1280, 8)])
def test_cross_attention_panels_vs_torch_fp32(L, B, N, Mk, C, heads):
    """`attn2(norm2(h), enc) + h` (patch.py:171-185) with the query and output projections as panel GEMMs."""
    from standin import Attention
    from vidtome_amd import patch as vpatch
    torch.manual_seed(N + C)
    attn = Attention(C, heads).eval().half()
    norm = torch.nn.LayerNorm(C).half()
    h = torch.randn(B, N, C).half()
    enc = torch.randn(B, Mk, C).half()
    with torch.no_grad():
        w = lambda m: m.weight.float()
        nh = torch.nn.functional.layer_norm(h.float(), (C,), norm.weight.float(), norm.bias.float(), norm.eps).half().float()
        q, k, v = nh @ w(attn.to_q).t(), enc.float() @ w(attn.to_k).t(), enc.float() @ w(attn.to_v).t()
        q, k, v = (t.half().float().reshape(B, -1, heads, C // heads).transpose(1, 2) for t in (q, k, v))
        p = torch.softmax(q @ k.transpose(-1, -2) * attn.scale, dim=-1)
        o = (p @ v).transpose(1, 2).reshape(B, N, C)
        ref = o @ w(attn.to_out[0]).t() + attn.to_out[0].bias.float() + h.float()
        ad, nd, hd, ed = attn.to(DEV), norm.to(DEV), h.to(DEV), enc.to(DEV)
        assert vpatch.fused_cross_ok(nd, ad, hd, ed, None, {})
        y = vpatch.norm_cross_attention_residual(nd, ad, hd, ed).float().cpu()
    assert (y - ref).abs().max() < 3e-3 * max(1.0, float(ref.abs().max()))


def test_panel_gemm_fuzz_vs_torch_fp32(L):
    """vtm_linear_panels / vtm_ff_geglu on random shapes (token counts around the 256-row tile edges, K = 64 ... 1280, output
    widths with ragged last 128-row weight tiles, with / without bias and residual, row-range operands) against fp32 PyTorch
    on the same fp16 operands."""
    g = torch.Generator().manual_seed(11)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    for trial in range(24):
        n = [1, 255, 256, 257, 511, 1000, 4097][ri(0, 6)] if trial % 2 else ri(1, 3000)
        K = 64 * ri(1, 20)
        N = 8 * ri(1, 80)
        x = torch.randn(n, K, generator=g).half()
        w = (torch.randn(N, K, generator=g) * K ** -0.5).half()
        bias = (0.1 * torch.randn(N, generator=g)).float() if ri(0, 1) else None
        resid = torch.randn(n, N, generator=g).half() if ri(0, 1) else None
        ref = x.float() @ w.float().t() + (bias if bias is not None else 0.0)
        ref = ref.half().float() + (resid.float() if resid is not None else 0.0)       # torch rounds the Linear output first
        xp, wp = L.to_panels(x.to(DEV)), L.to_panels(w.to(DEV))
        y = L.linear_panels(xp, n, wp, N, None if bias is None else bias.to(DEV), None if resid is None else resid.to(DEV))
        assert y.shape == (n, N)
        assert (y.float().cpu() - ref).abs().max() < 3e-3 * max(1.0, float(ref.abs().max())), (trial, n, K, N)
        # the same product with the operand roles swapped (how V^T = W_v X^T is computed): (N, n_pad8) = w x^T
        n8 = (n + 7) // 8 * 8
        yt = torch.zeros((N, L.panel_rows(n)), dtype=torch.float16, device=DEV)
        L.linear_panels(wp, N, xp, n8, None, out=yt)
        assert (yt[:, :n].float().cpu() - (x.float() @ w.float().t()).t()).abs().max() < 3e-3 * max(1.0, float(ref.abs().max()))
    for trial in range(8):                                   # GEGLU epilogue: D % 64 == 0
        n, K, D = ri(1, 2000), 64 * ri(1, 20), 64 * ri(1, 12)
        x = torch.randn(n, K, generator=g).half()
        w = (torch.randn(2 * D, K, generator=g) * K ** -0.5).half()
        b = (0.1 * torch.randn(2 * D, generator=g)).float()
        t = torch.arange(D // 64)[:, None] * 64 + torch.arange(64)[None, :]
        order = torch.cat([t, t + D], dim=1).reshape(-1).to(torch.int32)
        wp = L.to_panels(w.to(DEV), order.to(DEV))
        hp = L.ff_geglu(L.to_panels(x.to(DEV)), n, wp, D, b[order.long()].to(DEV))
        got = hp[:, :n].permute(1, 0, 2).reshape(n, D).float().cpu()
        proj = (x.float() @ w.float().t() + b).half().float()
        ref = proj[:, :D] * torch.nn.functional.gelu(proj[:, D:]).half().float()
        assert (got - ref).abs().max() < 3e-3 * max(1.0, float(ref.abs().max())), (trial, n, K, D)


def test_full_block_forward_panels_equals_library_path(L, monkeypatch):
    """The WHOLE patched block (ToMeBlock.forward: segment, norm2 / attn2 over 77 conditioning tokens, norm3 / GEGLU
    feed-forward; patch.py:128-201) at a merged top site, a merged mid site and an un-merged site: the panel-GEMM path
    (default) against the library-GEMM path of rounds 1-2 (VIDTOME_FF=blas, VIDTOME_PROJ=blas) over two chunks -- same merge
    plan, outputs within the fp16 tolerance of the two paths' own tests."""
    import vidtome_amd
    from vidtome_amd import patch as vpatch
    from vidtome_amd import sites
    sl = [sites.Site("top", 1, 320, 8), sites.Site("mid", 2, 640, 8), sites.Site("low", 4, 1280, 8)]
    B, F, latent = 2, 4, (16, 16)
    cond = torch.randn(B * F, 77, 768, generator=torch.Generator().manual_seed(3)).half().to(DEV)
    outs = {}
    for mode in ("panels", "blas"):
        monkeypatch.setattr(vpatch, "FF_MODE", mode)
        monkeypatch.setattr(vpatch, "PROJ_MODE", "auto" if mode == "panels" else "blas")
        monkeypatch.setattr(vpatch, "FUSED_PROJ", mode == "panels")
        unet = sites.SiteUNet(sl, seed=0, full=True).to(device=DEV, dtype=torch.float16)
        vidtome_amd.apply_patch(unet, local_merge_ratio=0.5, merge_global=True, global_merge_ratio=0.5, batch_size=B)
        unet.set_size(latent)
        torch.manual_seed(123)
        res = []
        with torch.no_grad():
            for ck in range(2):
                hs = [sites.synthetic_hidden(s_, B, F, latent, torch.float16, DEV, seed=20 * ck + i) for i, s_ in enumerate(sl)]
                res.append([o.float() for o in sites.run_block_pass(unet, hs, cond)])
        outs[mode] = res
        vidtome_amd.remove_patch(unet)
    for a, b in zip(outs["panels"], outs["blas"]):
        for x, y in zip(a, b):
            assert x.shape == y.shape and torch.isfinite(x).all()
            assert (x - y).abs().max().item() < 6e-3 * max(1.0, y.abs().max().item())


def test_attention_fuzz_vs_torch_fp32(L):
    """Random (B, h, d, Mq, Mk) incl. every supported head dim, ragged lengths and Mq != Mk, against a plain PyTorch
    fp32 softmax attention of the same fp16 inputs."""
    g = torch.Generator().manual_seed(21)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    for case in range(48):
        d = [8, 16, 32, 40, 64, 80, 96, 128, 160][case % 9]
        B, h = ri(1, 3), ri(1, 4)
        Mk = ri(1, 700)
        Mq = Mk if case % 3 == 0 else ri(1, 700)
        C = h * d
        Mqp, Mkp = (Mq + 7) // 8 * 8, (Mk + 7) // 8 * 8
        q = torch.zeros(B, Mqp, C)
        k = torch.zeros(B, Mkp, C)
        v = torch.zeros(B, Mkp, C)
        q[:, :Mq] = torch.randn(B, Mq, C, generator=g)
        k[:, :Mk] = torch.randn(B, Mk, C, generator=g)
        v[:, :Mk] = torch.randn(B, Mk, C, generator=g)
        q, k, v = q.half().to(DEV), k.half().to(DEV), v.half().to(DEV)
        o = L.attention_kv(q, k, v.transpose(1, 2).contiguous(), h, Mq, Mk, d ** -0.5)[:, :Mq].float()
        qh, kh, vh = (t.float().reshape(B, -1, h, d).transpose(1, 2) for t in (q[:, :Mq], k[:, :Mk], v[:, :Mk]))
        ref = (torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, dim=-1) @ vh).transpose(1, 2).reshape(B, Mq, C)
        err = float((o - ref).abs().max())
        assert err <= 1e-3 * max(1.0, float(ref.abs().max())), (case, B, h, d, Mq, Mk, err)


def test_attention_d40_wide_scores_fuzz(L):
    """d = 40 carries the softmax shift inside the contraction and only checks the maximum after the exps (on the
    packed P): inputs scaled so that the scores span tens of log2 units make that check fire in many tiles, in
    both directions (a late large score, long runs of very negative ones)."""
    g = torch.Generator().manual_seed(29)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    d, worst = 40, 0.0
    for case in range(24):
        B, h = ri(1, 2), ri(1, 3)
        Mk, Mq = ri(65, 2500), ri(1, 900)
        sq, sk = [1.0, 2.0, 4.0][case % 3], [1.0, 3.0][case % 2]
        C = h * d
        Mqp, Mkp = (Mq + 7) // 8 * 8, (Mk + 7) // 8 * 8
        q, k, v = torch.zeros(B, Mqp, C), torch.zeros(B, Mkp, C), torch.zeros(B, Mkp, C)
        q[:, :Mq] = sq * torch.randn(B, Mq, C, generator=g)
        k[:, :Mk] = sk * torch.randn(B, Mk, C, generator=g)
        v[:, :Mk] = torch.randn(B, Mk, C, generator=g)
        if case % 4 == 1:   # ascending key norms: the running maximum keeps growing along the key axis
            k[:, :Mk] *= torch.linspace(0.2, 1.0, Mk)[None, :, None]
        q, k, v = q.half().to(DEV), k.half().to(DEV), v.half().to(DEV)
        o = L.attention_kv(q, k, v.transpose(1, 2).contiguous(), h, Mq, Mk, d ** -0.5)[:, :Mq].float()
        assert torch.isfinite(o).all(), (case, B, h, Mq, Mk)
        qh, kh, vh = (t.float().reshape(B, -1, h, d).transpose(1, 2) for t in (q[:, :Mq], k[:, :Mk], v[:, :Mk]))
        ref = (torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, dim=-1) @ vh).transpose(1, 2).reshape(B, Mq, C)
        err = float((o - ref).abs().max())
        worst = max(worst, err)
        # a stress test of the shift logic, not of the accuracy contract (pinned at 1e-3 on realistic score ranges by
        # the tests above): with scores of up to +-60 log2 units the one fp16 rounding of the pre-scaled query
        # (relative 2^-12 per product) is worth ~1e-3 in the softmax weights of the few keys that compete
        assert err <= 2e-3 * max(1.0, float(ref.abs().max())), (case, B, h, Mq, Mk, sq, sk, err)


@pytest.mark.parametrize("d,Mq", [(80, 8704), (40, 8448), (64, 8704), (160, 8704)])
def test_attention_split_last_round_equals_single_launch(L, d, Mq):
    """A launch whose last round of workgroups is nearly empty runs those query blocks as key-split workgroups plus
    a combine kernel (attention.hip, plan_tail).  d = 80: shape of the cfg-2 mid blocks, 17 x 16 = 272 workgroups on
    a 256-CU chip; d = 40 (16-row O^T blocks, their own record layout): 33 x 16 = 528 workgroups on 512 slots."""
    B, h, Mk = 2, 8, 4160                           # 65 key tiles, the last one full
    C = h * d
    assert L.lib().vtm_attention_ws_bytes(B, h, Mq, Mk, d) > 0, "this shape is expected to take the split path"
    g = torch.Generator(device=DEV).manual_seed(3)
    q = torch.randn(B, Mq, C, generator=g, device=DEV, dtype=torch.float16)
    k = torch.randn(B, Mk, C, generator=g, device=DEV, dtype=torch.float16)
    vt = torch.randn(B, C, Mk, generator=g, device=DEV, dtype=torch.float16)
    a = L.attention_kv(q, k, vt, h, Mq, Mk, d ** -0.5)
    b = L.attention_kv(q, k, vt, h, Mq, Mk, d ** -0.5, use_workspace=False)
    assert torch.isfinite(a).all()
    assert (a.float() - b.float()).abs().max() <= 1e-3
    # ragged key count (last tile partly masked) through the split path
    Mk2, Mkp2 = Mk - 37, (Mk - 37 + 7) // 8 * 8
    k2 = torch.zeros(B, Mkp2, C, device=DEV, dtype=torch.float16)
    k2[:, :Mk2] = k[:, :Mk2]
    vt2 = torch.zeros(B, C, Mkp2, device=DEV, dtype=torch.float16)
    vt2[:, :, :Mk2] = vt[:, :, :Mk2]
    a2 = L.attention_kv(q, k2, vt2, h, Mq, Mk2, d ** -0.5)
    b2 = L.attention_kv(q, k2, vt2, h, Mq, Mk2, d ** -0.5, use_workspace=False)
    assert (a2.float() - b2.float()).abs().max() <= 1e-3


@pytest.mark.parametrize("name,B,h,d,Mq,Mk", [
    ("cfg-2 top block (live queries x all keys)", 2, 8, 40, 34816, 52224),
    ("cfg-2 top block, every row a query (VIDTOME_LIVE_QUERIES=0)", 2, 8, 40, 52224, 52224),
    ("cfg-2 mid block", 2, 8, 80, 8704, 13056),
    ("cfg-4 top block", 2, 8, 40, 18432, 27648),
    ("cfg-5 top block (SD-2.1-768)", 2, 5, 64, 64513, 90319),
])
def test_attention_full_size_vs_oracle(L, oracle, name, B, h, d, Mq, Mk):
    """The launches bench.py times (and cfg-4 / cfg-5's largest), at FULL size, against the oracle on sampled query
    rows: 1e-3 of the output scale (north_star).  The sample covers the first and last query blocks, the key-split
    tail workgroups of the last round (workspace path on) and rows spread over the rest; ragged Mq / Mk at cfg-5."""
    C = h * d
    Mqp, Mkp = (Mq + 7) // 8 * 8, (Mk + 7) // 8 * 8
    g = torch.Generator(device=DEV).manual_seed(Mq + Mk)
    # frame-correlated keys (merged video tokens are) and moderately peaked softmax rows
    base = torch.randn(B, 1, C, generator=g, device=DEV)
    q = torch.zeros(B, Mqp, C, device=DEV, dtype=torch.float16)
    k = torch.zeros(B, Mkp, C, device=DEV, dtype=torch.float16)
    q[:, :Mq] = (1.5 * torch.randn(B, Mq, C, generator=g, device=DEV)).half()
    k[:, :Mk] = (0.3 * base + torch.randn(B, Mk, C, generator=g, device=DEV)).half()
    vt = torch.zeros(B, C, Mkp, device=DEV, dtype=torch.float16)
    vt[:, :, :Mk] = torch.randn(B, C, Mk, generator=g, device=DEV).half()
    # which of these launches end in key-split tail workgroups (attention.hip, plan_tail, on a 256-CU chip)
    assert (L.lib().vtm_attention_ws_bytes(B, h, Mq, Mk, d) > 0) == name.startswith(("cfg-2 top block (live", "cfg-2 mid",
                                                                                     "cfg-4 top"))
    out = L.attention_kv(q, k, vt, h, Mq, Mk, d ** -0.5)
    rs = np.random.default_rng(Mq)
    rows = np.unique(np.concatenate([np.arange(0, 64), np.arange(Mq - 96, Mq), rs.choice(Mq, 384, replace=False)]))
    assert len(rows) >= 512
    ridx = torch.from_numpy(rows).to(DEV)
    got = out[:, ridx].float().cpu().numpy()
    ref = oracle.attention_qkv(q[:, ridx].float().cpu().numpy(), k[:, :Mk].float().cpu().numpy(),
                               vt[:, :, :Mk].transpose(1, 2).float().cpu().numpy(), h, d ** -0.5)
    scale = float(np.abs(ref).max())
    err = float(np.abs(got - ref).max())
    assert scale > 0.05, "the comparison should not be about zeros"
    assert err < 1e-3 * max(1.0, scale), (name, err, scale)      # north_star's bound
    assert err < 5e-3 * scale, (name, err, scale)                # and well inside it relative to the outputs themselves
    assert torch.isfinite(out[:, :Mq]).all()
    if Mqp != Mq:
        assert (out[:, Mq:] == 0).all()


@pytest.mark.parametrize("d", [40, 64])
@pytest.mark.parametrize("case", ["spike_late", "spike_every_tile", "all_very_negative", "wide_range"])
def test_attention_rescale_paths(L, oracle, d, case):
    """The online-softmax corner cases: the running maximum jumps late (deferred-rescale branch, and for d = 40 the
    shift that rides in the spare k-slot has to be re-installed), grows in every tile, or sits far below zero."""
    B, h, M = 1, 2, 700          # 11 key tiles of 64
    C = h * d
    g = torch.Generator().manual_seed(77 + d)
    q = torch.randn(B, M, C, generator=g)
    k = torch.randn(B, M, C, generator=g)
    v = torch.randn(B, M, C, generator=g)
    if case == "spike_late":
        k[:, 600] = 6.0 * q[:, 17]                     # one key dominates query 17 from tile 9 on
        k[:, 333, :d] = 5.0 * q[:, 400, :d]
    elif case == "spike_every_tile":
        for t in range(11):
            k[:, min(64 * t + 5, M - 1)] = (0.5 + 0.45 * t) * q[:, 3]
    elif case == "all_very_negative":
        k = -3.0 * q[:, :1].expand(B, M, C).clone() + 0.05 * k   # every score of query 0 ~ -3 |q|^2 / sqrt(d)
        q[:, 1:] = 3.0 * q[:, :1] + 0.05 * q[:, 1:]
    else:
        q, k = 3.0 * q, 3.0 * k
    q, k, v = q.half(), k.half(), v.half()
    Mp = (M + 7) // 8 * 8
    pad = lambda t: torch.nn.functional.pad(t, (0, 0, 0, Mp - M))
    o = L.attention(pad(q).to(DEV), pad(k).to(DEV), pad(v).to(DEV).transpose(1, 2).contiguous(), h, M, d ** -0.5, 1)
    o = o[:, :M].float().cpu().numpy()
    ref = oracle.attention(q.float().numpy(), k.float().numpy(), v.float().numpy(), h, share_groups=1)
    assert np.isfinite(o).all()
    # 1e-3 as everywhere else, except the deliberately extreme "wide_range" logits (|s * scale| up to ~40, softmax
    # decided by differences of a few 1e-3) on the heads whose query is pre-scaled in fp16 (d % 16 != 0, see
    # attention.hip): the extra rounding of q costs |logit| * 2^-12 there.  An fp16 reference that rounds its scores
    # to fp16 (the Diffusers attention the reference calls) is off by |logit| * 2^-11 on the same input.
    tol = 2e-3 if (case == "wide_range" and d % 16) else 1e-3
    assert np.abs(o - ref).max() < tol * max(1.0, np.abs(ref).max()), np.abs(o - ref).max()


# ---------------------------------------------------------------------------------------------------
# the other BASELINE.json configs at full size, through apply_patch on SD-shaped block sites
# ---------------------------------------------------------------------------------------------------
def _site_pass_checks(unet, sites_list, hiddens, B, F, expected_M):
    """One hot-path pass; checks sizes, finiteness and that surviving tokens attend exactly like a dense
    recomputation of a few of their rows (attention over the merged set, oracle-free)."""
    from vidtome_amd import patch as vpatch
    from vidtome_amd import sites as S
    seen = {}
    orig = vpatch.compute_merge

    def rec(module, x, info, **kw):
        res = orig(module, x, info, **kw)
        seen[id(module)] = getattr(res[0], "plan", None)
        return res

    vpatch.compute_merge = rec
    try:
        with torch.no_grad():
            outs = S.run_segment_pass(unet, hiddens)
    finally:
        vpatch.compute_merge = orig
    plans = [seen.get(id(blk)) for blk in unet.blocks]      # (un-merged sites may never call compute_merge)
    for site, h, o, plan in zip(sites_list, hiddens, outs, plans):
        assert o.shape == h.shape and bool(torch.isfinite(o).all()), site.name
        if site.downsample <= 2:
            assert plan is not None and plan.M == expected_M[site.downsample], (site.name, plan.M)
        else:
            assert plan is None
    return outs, plans


def _block_rows_vs_oracle(oracle, blk, plan, hidden, out, fsize, share=1, n_rows=192, seed=0, out_ulp=False):
    """Block outputs at FULL size against the oracle on SAMPLED token positions: for position i of the joined chunk the
    reference is  to_out(softmax(q K^T) V)[inv[i]] + hidden[i]  with q / K / V projected (fp32) from the merged tokens the
    plan selects, the attention rows by oracle.attention_qkv (double accumulation); `share` > 1 = PnP shared
    probabilities (q / k of the source sample for every sample, pnp_utils.py:57-67).  1e-3 of the output scale.  With
    `out_ulp` (cfg-4's 18 432-key chunks, where a few rows have a sharply peaked softmax): all but 1e-4 of the sampled
    elements within 1e-3 and every one within 2e-3 -- tools/diag/cfg4_err.py shows 2 of 164 000 elements at 1.3e-3 of the
    scale (p99.9 = 3.7e-4), identically for an fp16 and an fp32 model: the fp16 rounding of q / k of a row whose largest
    logit is tens of log2 units, not the projections (profiles/HISTORY.md section 10.6)."""
    from vidtome_amd.utils import join_frame
    a = blk.attn1
    f32 = lambda t: t.detach().float().cpu().numpy()
    wq, wk, wv, wo, bo = f32(a.to_q.weight), f32(a.to_k.weight), f32(a.to_v.weight), f32(a.to_out[0].weight), f32(a.to_out[0].bias)
    merged = f32(plan.merged[:, :plan.M])                        # (B, M, C): row copies of OUR norm1 output
    Bn, L = plan.inv.shape
    g = np.random.default_rng(seed)
    idx = np.unique(np.concatenate([np.arange(8), np.arange(L - 8, L), g.integers(0, L, n_rows)]))
    inv = plan.inv.cpu().numpy()
    m = inv[:, idx]                                              # (B, S) merged row of every sampled position
    src = lambda t: np.broadcast_to(t[:1], t.shape) if share > 1 else t
    k, v = merged @ wk.T, merged @ wv.T
    q = np.stack([src(merged)[b, m[b]] for b in range(Bn)]) @ wq.T
    o = oracle.attention_qkv(np.ascontiguousarray(q), np.ascontiguousarray(src(k)), v, a.heads)
    hj, oj = f32(join_frame(hidden, fsize)), f32(join_frame(out, fsize))
    ref = o @ wo.T + bo + hj[:, idx]
    diff = np.abs(oj[:, idx] - ref)
    if out_ulp:
        scale = max(1.0, np.abs(ref).max())
        assert (diff > 1e-3 * scale).mean() < 1e-4 and diff.max() < 2e-3 * scale, (diff.max(), scale)
        return
    err = diff.max()
    assert err < 1e-3 * max(1.0, np.abs(ref).max()), err


@pytest.mark.parametrize("B,Ml,U,Nd", [(2, 1000, 400, 700), (1, 17, 0, 5), (3, 34816, 17408, 34816), (2, 300, 300, 64)])
def test_compact_queries_vs_torch(L, B, Ml, U, Nd):
    """vtm_compact_queries against a torch restatement: distinct merged positions ([0, U) then the matched anchor rows
    ascending), the row each local token reads, the per-sample count; entries past the count are valid indices."""
    g = torch.Generator().manual_seed(Ml + Nd)
    loc = torch.empty(B, Ml, dtype=torch.int64)
    for b in range(B):
        perm = torch.randperm(Ml, generator=g)
        pos = torch.empty(Ml, dtype=torch.int64)
        pos[perm[:U]] = torch.arange(U)                                   # unmerged tokens: one row each
        hot = torch.randint(0, max(1, Nd // 3), (Ml - U,), generator=g)   # merged ones: many share an anchor row
        pos[perm[U:]] = U + hot
        loc[b] = pos
    qc, tmap, count = L.compact_queries(loc.to(torch.int32).to(DEV), U, Nd)
    qc, tmap, count = qc.cpu().long(), tmap.cpu().long(), count.cpu().long()
    for b in range(B):
        want = torch.cat([torch.arange(U), torch.unique(loc[b][loc[b] >= U])])   # unique() sorts ascending
        n = int(count[b])
        assert n == want.numel()
        assert torch.equal(qc[b, :n], want)
        assert int(qc[b, n:].min() if n < Ml else 0) >= 0 and int(qc[b].max()) < U + Nd
        assert torch.equal(qc[b][tmap[b]], loc[b])                        # every token finds its own position
        assert int(tmap[b].max()) < n


def test_attention_bounded_equals_unbounded_prefix(L, monkeypatch):
    """vtm_attention_kv_bounded: rows below the per-sample count equal the plain launch's rows -- bit for bit when both take
    the same launch plan (the bounded workspace withheld), within the tolerance of another summation order when the
    bounded launch is planned on the device from the counts (round 6: a geometric key-split tail); the launch covers counts
    inside the first block, mid-sequence and the full length."""
    B, h, d, Mq, Mk = 3, 8, 40, 8448, 9000
    C = h * d
    g = torch.Generator(device=DEV).manual_seed(5)
    q = torch.randn(B, Mq, C, generator=g, device=DEV, dtype=torch.float16)
    k = torch.randn(B, (Mk + 7) // 8 * 8, C, generator=g, device=DEV, dtype=torch.float16)
    vt = torch.randn(B, C, (Mk + 7) // 8 * 8, generator=g, device=DEV, dtype=torch.float16)
    full = L.attention_kv(q, k, vt, h, Mq, Mk, d ** -0.5)
    count = torch.tensor([100, 5000, Mq], dtype=torch.int32, device=DEV)
    got = L.attention_kv(q, k, vt, h, Mq, Mk, d ** -0.5, q_count=count)
    scale = float(full.float().abs().max())
    for b, n in enumerate(count.tolist()):
        assert (got[b, :n].float() - full[b, :n].float()).abs().max() < 2e-3 * scale, b
    monkeypatch.setattr(L, "SPLIT_ALL_BOUNDED", False)
    got = L.attention_kv(q, k, vt, h, Mq, Mk, d ** -0.5, q_count=count)
    for b, n in enumerate(count.tolist()):
        assert torch.equal(got[b, :n], full[b, :n]), b


@pytest.mark.parametrize("d,h", [(40, 8), (64, 5), (80, 8)])
def test_attention_device_planned_tail_every_live_fraction(L, d, h):
    """Round 6: a query-bounded launch of two or more rounds of workgroups is planned on the device from the live counts
    (attention16_plan_kernel: whole items, then tiers split 2 / 4 / 8 / 16 ways along the key axis; csrc/attention16_parts.h) --
    by attention16s_kernel at d = 40 and by attention_kernel at every other head dim.  Whatever the counts make of the plan
    (no tail, one tier, several, a launch that no longer fills the chip; different counts per sample), every row below its
    sample's count must equal the plain launch's row within the tolerance of another summation order.  vidtome/patch.py:157-162
    on the rows merge.py:439-460 reads."""
    B, Mk = 2, 2100                                     # 33 key tiles: pieces of >= 8 tiles, up to 4-way splits
    Mq = 32768 if d != 64 else 53248                    # x B x h >= 2 x 256 slots of 512-query items in every instantiation
    C = h * d
    g = torch.Generator(device=DEV).manual_seed(13)
    q = torch.randn(B, Mq, C, generator=g, device=DEV, dtype=torch.float16)
    k = torch.randn(B, (Mk + 7) // 8 * 8, C, generator=g, device=DEV, dtype=torch.float16)
    vt = torch.randn(B, C, (Mk + 7) // 8 * 8, generator=g, device=DEV, dtype=torch.float16)
    full = L.attention_kv(q, k, vt, h, Mq, Mk, d ** -0.5)
    scale = float(full.float().abs().max())
    for f0, f1 in ((1.0, 1.0), (0.97, 0.9), (0.83, 0.83), (0.56, 0.7), (0.5, 0.25), (0.26, 0.3), (0.05, 0.02), (0.001, 0.6)):
        count = torch.tensor([max(1, int(Mq * f0)), max(1, int(Mq * f1))], dtype=torch.int32, device=DEV)
        got = L.attention_kv(q, k, vt, h, Mq, Mk, d ** -0.5, q_count=count)
        for b, n in enumerate(count.tolist()):
            assert torch.isfinite(got[b, :n]).all(), (d, f0, f1, b)
            assert (got[b, :n].float() - full[b, :n].float()).abs().max() < 2e-3 * scale, (d, f0, f1, b)
        # the plan the launch ran on sits in the first 256 bytes of the call's workspace (csrc/attention16_parts.h: DevPlan =
        # {nqb, ntiers, split_items, pad, 8 x {wg0, item0, items, nsplit, rec0}}): its invariants, whatever the counts
        ws = L._workspace("attention", 256, q.device)
        plan = ws[:176].view(torch.int32).cpu().tolist()
        nqb, ntiers, split_items = plan[0], plan[1], plan[2]
        tiers = [plan[4 + 5 * i: 9 + 5 * i] for i in range(8)]
        QB, slots = 512, 256
        assert nqb == max(-(-int(c) // QB) for c in count.tolist()), (d, f0, f1, plan)
        items_total = nqb * h * B
        assert 1 <= ntiers <= 8 and sum(t[2] for t in tiers[:ntiers]) == items_total, (d, f0, f1, plan)
        assert tiers[0][:2] == [0, 0] and tiers[0][3] == 1 and split_items == items_total - tiers[0][2], (d, f0, f1, plan)
        wg, item, rec = tiers[0][2], tiers[0][2], 0
        for t in tiers[1:ntiers]:
            assert t[0] == wg and t[1] == item and t[4] == rec and 2 <= t[3] <= 4 and t[2] > 0, (d, f0, f1, plan)   # 33 key tiles: <= 4 pieces
            wg, item, rec = wg + t[2] * t[3], item + t[2], rec + t[2] * t[3]
        assert rec <= 7 * slots and split_items <= slots, (d, f0, f1, plan)


@pytest.mark.parametrize("d,h,groups", [(40, 8, 3), (40, 8, 2), (80, 8, 3), (64, 5, 3)])
def test_attention_shared_probabilities_with_a_query_bound(L, d, h, groups):
    """vtm_attention_kv_shared_bounded (round 6): the PnP shared-probability launch (pnp_utils.py:57-67, 75-90: q / k of the
    source sample, v per sample) on compacted live queries -- under align_batch every sample of a group has the same live
    rows, so one count bounds the group (merge.py:439-460 reads nothing else).  Rows below the count equal the plain shared
    launch (attention16g at d = 40: the probabilities once per group; attention_kernel elsewhere: the device-planned tail)
    and, directly, softmax(q_src k_src^T) v_b in fp32; every count from a full launch down to one block."""
    B, Mq, Mk = groups, 19456, 2100
    C = h * d
    g = torch.Generator(device=DEV).manual_seed(17)
    q = torch.randn(B, Mq, C, generator=g, device=DEV, dtype=torch.float16)
    k = torch.randn(B, (Mk + 7) // 8 * 8, C, generator=g, device=DEV, dtype=torch.float16)
    vt = torch.randn(B, C, (Mk + 7) // 8 * 8, generator=g, device=DEV, dtype=torch.float16)
    full = L.attention_kv(q, k, vt, h, Mq, Mk, d ** -0.5, share_groups=groups)
    scale = float(full.float().abs().max())
    # fp32 statement of a few rows: probabilities of the SOURCE sample (sample 0), values of sample b
    rows = torch.tensor([0, 1, 255, 256, 4097, 12345, Mq - 1], device=DEV)
    for hd in (0, h - 1):
        qs = q[0, rows, hd * d:(hd + 1) * d].float()
        ks = k[0, :Mk, hd * d:(hd + 1) * d].float()
        p = torch.softmax(qs @ ks.t() * d ** -0.5, dim=-1)
        for b in range(B):
            ref = p @ vt[b, hd * d:(hd + 1) * d, :Mk].float().t()
            assert (full[b, rows, hd * d:(hd + 1) * d].float() - ref).abs().max() < 2e-3 * scale, (d, b, hd)
    for frac in (1.0, 0.93, 0.78, 0.5, 0.27, 0.01):
        n = max(1, int(Mq * frac))
        count = torch.full((B,), n, dtype=torch.int32, device=DEV)
        got = L.attention_kv(q, k, vt, h, Mq, Mk, d ** -0.5, q_count=count, share_groups=groups)
        assert torch.isfinite(got[:, :n]).all(), (d, frac)
        assert (got[:, :n].float() - full[:, :n].float()).abs().max() < 2e-3 * scale, (d, frac)
    with pytest.raises(RuntimeError):          # folded keys do not go with shared probabilities
        L.attention_kv(q, k, vt, h, Mq, Mk, d ** -0.5, share_groups=groups,
                       k_fold=(torch.full((B,), Mk, dtype=torch.int32, device=DEV), torch.zeros(B, Mk, dtype=torch.int32, device=DEV)))


def test_attention_bounded_split_all_vs_plain(L, monkeypatch):
    """A query-bounded launch that fills at least two rounds of the chip splits EVERY work item in two along the key axis
    (split-major order, partial records merged by attention_combine_kernel): its rows below the per-sample count must
    match the plain launch within the tolerance of another summation order, rows of samples with a small count included;
    with the switch off the two are bit-identical."""
    B, h, d, Mq, Mk = 2, 8, 40, 17408, 9000            # 68 query blocks x 16 pairs = 1 088 workgroups >= 2 x 512
    C = h * d
    g = torch.Generator(device=DEV).manual_seed(7)
    q = torch.randn(B, Mq, C, generator=g, device=DEV, dtype=torch.float16)
    k = torch.randn(B, (Mk + 7) // 8 * 8, C, generator=g, device=DEV, dtype=torch.float16)
    vt = torch.randn(B, C, (Mk + 7) // 8 * 8, generator=g, device=DEV, dtype=torch.float16)
    assert L.lib().vtm_attention_kv_bounded_ws_bytes(B, h, Mq, Mk, d) > L.lib().vtm_attention_ws_bytes(B, h, Mq, Mk, d)
    full = L.attention_kv(q, k, vt, h, Mq, Mk, d ** -0.5)
    count = torch.tensor([300, Mq - 5], dtype=torch.int32, device=DEV)
    got = L.attention_kv(q, k, vt, h, Mq, Mk, d ** -0.5, q_count=count)
    scale = float(full.float().abs().max())
    for b, n in enumerate(count.tolist()):
        assert (got[b, :n].float() - full[b, :n].float()).abs().max() < 2e-3 * scale, b
    monkeypatch.setattr(L, "SPLIT_ALL_BOUNDED", False)
    got0 = L.attention_kv(q, k, vt, h, Mq, Mk, d ** -0.5, q_count=count)
    for b, n in enumerate(count.tolist()):
        assert torch.equal(got0[b, :n], full[b, :n]), b


@pytest.mark.parametrize("F,global_rand", [(4, 0.0), (4, 1.0), (1, 0.5), (8, 0.5)])
def test_live_queries_equal_full_attention(L, F, global_rand):
    """With a global level the block computes attention only for the merged rows whose output unmerge() reads
    (MergePlan.q_rows).  The block output must equal the one obtained by attending from every merged row, as the
    reference does (patch.py:157-169) -- for both coin outcomes (local chunk = src / dst) and single-frame chunks."""
    import vidtome_amd
    from vidtome_amd import patch as vpatch
    from vidtome_amd import sites as S
    B, latent = 2, (16, 16)
    sl = [s for s in S.sd15_sites() if s.name in ("up3.0", "up2.0")]
    outs = {}
    for live in (True, False):
        unet = S.SiteUNet(sl, seed=3).to(device=DEV, dtype=torch.float16)
        vidtome_amd.apply_patch(unet, local_merge_ratio=0.5, merge_global=True, global_merge_ratio=0.5, batch_size=B,
                                global_rand=global_rand)
        unet.set_size(latent)
        torch.manual_seed(123)
        old = vpatch.LIVE_QUERIES
        vpatch.LIVE_QUERIES = live
        try:
            res = []
            for chunk in range(3):       # chunk 0 stores anchors, chunks 1 and 2 merge against them
                hiddens = [S.synthetic_hidden(s, B, F, latent, torch.float16, DEV, seed=90 + 7 * chunk + i)
                           for i, s in enumerate(sl)]
                with torch.no_grad():
                    res.append([o.float().cpu() for o in S.run_segment_pass(unet, hiddens)])
            outs[live] = res
        finally:
            vpatch.LIVE_QUERIES = old
            vidtome_amd.remove_patch(unet)
    for a_chunk, b_chunk in zip(outs[True], outs[False]):
        for a, b in zip(a_chunk, b_chunk):
            assert a.shape == b.shape and torch.isfinite(a).all()
            assert (a - b).abs().max() <= 2e-3 * max(1.0, float(b.abs().max()))


def test_two_chunks_in_flight_equal_one_stream(L):
    """Round 6: consecutive chunks may run on different HIP streams (bench.py `two_in_flight`; sites.ClipStream(inflight=2)).  The
    anchors chunk k + 1 takes from chunk k (patch.py:60-82) are handed over by a device-side event the producer records behind
    them (patch.mark_anchors_ready / await_anchors); everything else a chunk touches is its own (workspaces are keyed by
    stream).  Six steady-state chunks of a top, a mid and an un-merged site through one stream and through two: block outputs
    and the anchors left behind bit-identical, generators in the same state."""
    import vidtome_amd
    from vidtome_amd import sites as S
    B, F, latent = 2, 8, (32, 32)
    sl = [S.Site("up3.0", 1, 320, 8), S.Site("up2.0", 2, 640, 8), S.Site("up1.0", 4, 1280, 8)]
    res = {}
    for inflight in (1, 2):
        unet = S.SiteUNet(sl, seed=5).to(device=DEV, dtype=torch.float16)
        vidtome_amd.apply_patch(unet, local_merge_ratio=0.5, merge_global=True, global_merge_ratio=0.5, batch_size=B)
        unet.set_size(latent)
        torch.manual_seed(123)
        stream = S.ClipStream(unet, sl, B, F, latent, torch.float16, torch.device(DEV, 0), n_sets=3, chunks_per_step=4,
                              regime="corr01", inflight=inflight)
        stream.populate()
        torch.cuda.synchronize()
        outs = [stream.step(c) for c in range(6)]                 # (no synchronisation between the chunks)
        torch.cuda.synchronize()
        res[inflight] = ([[o.float().cpu() for o in oc] for oc in outs],
                         [b.global_tokens.float().cpu() for b in unet.blocks if getattr(b, "global_tokens", None) is not None],
                         [b.generator.get_state() for b in unet.blocks if hasattr(b, "generator")])
        vidtome_amd.remove_patch(unet)
    for oa, ob in zip(res[1][0], res[2][0]):
        for a, b in zip(oa, ob):
            assert torch.equal(a, b)
    assert len(res[1][1]) == len(res[2][1]) > 0
    for a, b in zip(res[1][1], res[2][1]):
        assert torch.equal(a, b)
    for a, b in zip(res[1][2], res[2][2]):
        assert torch.equal(a, b)


def test_run_step_with_streams_equals_the_sequential_loop(L):
    """scheduler.run_step(..., streams=[s0, s1]) (round 6): the chunks of a denoising step issued on alternating HIP streams, in
    the reference's order (generate.py:215-219), COLD caches included (the first chunks build the packed weights and the
    workspaces the next chunk's stream reads).  Chunk lengths as the reference draws them (a random first chunk, 1 .. 4
    frames: single-frame chunks have no local level).  Outputs of every chunk, the anchors and the generator states equal the
    sequential loop's bit for bit; the anchors are reset after the step either way (generate.py:233-236)."""
    import vidtome_amd
    from vidtome_amd import scheduler as sch
    from vidtome_amd import sites as S
    B, latent, n_frames = 2, (32, 32), 14
    sl = [S.Site("up3.0", 1, 320, 8), S.Site("up2.0", 2, 640, 8)]
    res = {}
    for use_streams in (False, True):
        unet = S.SiteUNet(sl, seed=6).to(device=DEV, dtype=torch.float16)
        vidtome_amd.apply_patch(unet, local_merge_ratio=0.5, merge_global=True, global_merge_ratio=0.5, batch_size=B)
        unet.set_size(latent)
        np.random.seed(3)
        torch.manual_seed(3)
        sc = sch.ChunkScheduler(chunk_size=4, merge_global=True, chunk_ord="seq")
        outs, anchors = {}, {}

        def process(chunk):
            F, f0 = len(chunk), int(chunk[0])
            hs = [S.synthetic_hidden(s_, B, F, latent, torch.float16, DEV, seed=500 + f0 + 31 * i, clip_seed=9 + i, regime="corr01")
                  for i, s_ in enumerate(sl)]
            with torch.no_grad():
                outs[f0] = S.run_segment_pass(unet, hs)
            anchors[f0] = [b.global_tokens for b in unet.blocks]

        streams = [torch.cuda.Stream(), torch.cuda.Stream()] if use_streams else None
        for _ in range(2):                                       # two denoising steps: the second meets warm caches
            chunks = sch.run_step(unet, sc, n_frames, process, streams=streams)
            assert all(getattr(b, "global_tokens", None) is None for b in unet.blocks)
        torch.cuda.synchronize()
        res[use_streams] = ({k: [o.float().cpu() for o in v] for k, v in outs.items()},
                            {k: [a.float().cpu() for a in v] for k, v in anchors.items()},
                            [b.generator.get_state() for b in unet.blocks], [len(c) for c in chunks])
        vidtome_amd.remove_patch(unet)
    assert res[False][3] == res[True][3] and res[False][0].keys() == res[True][0].keys()
    for k in res[False][0]:
        for a, b in zip(res[False][0][k], res[True][0][k]):
            assert torch.equal(a, b), k
        for a, b in zip(res[False][1][k], res[True][1][k]):
            assert torch.equal(a, b), k
    for a, b in zip(res[False][2], res[True][2]):
        assert torch.equal(a, b)


def test_projection_paths_agree(L):
    """The patched segment with its projections fed through the composed merge map (vtm_linear_rows), as panel GEMMs
    (vtm_gather_panels / vtm_layernorm_panels + vtm_linear_panels) and in the default mix of the two ("auto": rows at
    C <= 320, panels above -- no library GEMM anywhere) equals the same segment over materialised merged tokens with library
    GEMMs (VIDTOME_PROJ=blas) -- three chunks, local + global levels, both coin outcomes with live / compacted queries,
    merged and un-merged sites."""
    import vidtome_amd
    from vidtome_amd import patch as vpatch
    from vidtome_amd import sites
    sl = [sites.Site("top", 1, 320, 8), sites.Site("mid", 2, 640, 8), sites.Site("low", 4, 1280, 8)]
    B, F, latent = 2, 4, (16, 16)
    outs = {}
    saved = vpatch.PROJ_MODE
    for mode in ("blas", "rows", "panels", "auto"):
        vpatch.FUSED_PROJ, vpatch.PROJ_MODE = mode != "blas", mode
        try:
            unet = sites.SiteUNet(sl, seed=0).to(device=DEV, dtype=torch.float16)
            vidtome_amd.apply_patch(unet, local_merge_ratio=0.5, merge_global=True, global_merge_ratio=0.5, batch_size=B)
            unet.set_size(latent)
            torch.manual_seed(123)
            res = []
            with torch.no_grad():
                for ck in range(3):
                    unet._tome_info["args"]["global_rand"] = [0.5, 0.0, 1.0][ck]      # chunk 1: local = src, chunk 2: dst
                    hs = [sites.synthetic_hidden(s_, B, F, latent, torch.float16, DEV, seed=10 * ck + i)
                          for i, s_ in enumerate(sl)]
                    res.append([o.float() for o in sites.run_segment_pass(unet, hs)])
            outs[mode] = res
        finally:
            vpatch.FUSED_PROJ, vpatch.PROJ_MODE = True, saved
    for mode in ("rows", "panels", "auto"):
        for a, b in zip(outs[mode], outs["blas"]):
            for x, y in zip(a, b):
                assert torch.isfinite(x).all()
                assert (x - y).abs().max().item() < 4e-3 * max(1.0, y.abs().max().item()), mode


def test_cfg5_sd21_768_full_size(L, oracle):
    """cfg-5: SD-2.1-768, 16 frames 768x768 (latent 96x96), ratio 0.6, fp16: N = 9216 / 2304 tokens per frame,
    head dim 64, ragged merged lengths (64 513 / 16 129 local, 90 319 / 22 581 with global merging)."""
    import vidtome_amd
    from vidtome_amd import sites as S
    B, F, latent = 2, 16, (96, 96)
    sl = [s for s in S.sd21_sites() if s.name in ("down0.0", "down1.0", "down2.0", "mid")]
    unet = S.SiteUNet(sl, seed=1).to(device=DEV, dtype=torch.float16)
    vidtome_amd.apply_patch(unet, local_merge_ratio=0.6, merge_global=True, global_merge_ratio=0.6, batch_size=B)
    unet.set_size(latent)
    torch.manual_seed(123)
    hiddens = [S.synthetic_hidden(s, B, F, latent, torch.float16, DEV, seed=50 + i) for i, s in enumerate(sl)]
    outs, plans = _site_pass_checks(unet, sl, hiddens, B, F, {1: 64513, 2: 16129})           # SURVEY.md 8d sizes
    _block_rows_vs_oracle(oracle, unet.blocks[1], plans[1], hiddens[1], outs[1], F)          # mid site, local levels only
    hiddens2 = [S.synthetic_hidden(s, B, F, latent, torch.float16, DEV, seed=150 + i) for i, s in enumerate(sl)]
    outs, plans = _site_pass_checks(unet, sl, hiddens2, B, F, {1: 90319, 2: 22581})          # second chunk: + global merge
    for i in (0, 1):                                                                         # top (d = 64, ragged 90 319) and mid
        _block_rows_vs_oracle(oracle, unet.blocks[i], plans[i], hiddens2[i], outs[i], F, seed=i)
    vidtome_amd.remove_patch(unet)


def test_cfg3_pnp_batch3_aligned_full_size(L, oracle):
    """cfg-3 shape: batch 3 (source | uncond | cond), align_batch=True (one matching shared by the batch,
    merge.py:93-108) and PnP shared-probability attention (pnp_utils.py:57-67,86-90) at 16 x 512x512."""
    import vidtome_amd
    from vidtome_amd import patch as vpatch
    from vidtome_amd import sites as S
    B, F, latent = 3, 16, (64, 64)
    sl = [s for s in S.sd15_sites() if s.name in ("up3.0", "up2.0", "up1.0")]
    unet = S.SiteUNet(sl, seed=2).to(device=DEV, dtype=torch.float16)
    for blk in unet.blocks:      # what pnp.register_attention_control sets on the decoder blocks
        blk.attn1.injection_schedule, blk.attn1.t, blk.attn1.vtm_num_inputs = [981], 981, B
    vidtome_amd.apply_patch(unet, local_merge_ratio=0.5, merge_global=True, global_merge_ratio=0.5, batch_size=B,
                            align_batch=True)
    unet.set_size(latent)
    torch.manual_seed(123)
    hiddens = [S.synthetic_hidden(s, B, F, latent, torch.float16, DEV, seed=70 + i) for i, s in enumerate(sl)]
    for expected in ({1: 34816, 2: 8704}, {1: 52224, 2: 13056}):
        outs, plans = _site_pass_checks(unet, sl, hiddens, B, F, expected)
        for plan in [p for p in plans if p is not None]:
            # aligned matching: every sample of the batch shares ONE index set
            gm = plan.gather_map
            assert torch.equal(gm[0], gm[1]) and torch.equal(gm[0], gm[2])
        # block outputs at full size vs the oracle (shared probabilities: q / k of the source sample), sampled positions
        for i in (1, 2):            # up2.0 (C = 640, d = 80) and up3.0 (C = 320, d = 40); up1.0 does not merge
            _block_rows_vs_oracle(oracle, unet.blocks[i], plans[i], hiddens[i], outs[i], F, share=B, seed=i)
    # shared probabilities: with identical V across the batch groups the three groups' attention outputs
    # coincide although their own q/k differ (q/k of the source group are used for all)
    blk = unet.blocks[2]         # up3.0: C = 320
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 1024, 320, generator=g).half().to(DEV)
    with torch.no_grad():
        y = vpatch.self_attention(blk.attn1, x)
        x2 = x.clone()
        x2[1:] = x[:1]          # same tokens in all groups -> same v; q/k come from group 0 anyway
        y2 = vpatch.self_attention(blk.attn1, x2)
    assert torch.allclose(y2[0], y2[1]) and torch.allclose(y2[0], y2[2]) and torch.allclose(y[0], y2[0])
    vidtome_amd.remove_patch(unet)


def test_pnp_shared_probabilities_take_the_live_query_rows(L, oracle):
    """Round 6: with align_batch the levels' indices -- hence the live query rows of a global level whose local chunk is
    the src side (merge.py:439-460: only the local tokens' outputs are ever read) -- are the same rows in every sample, so
    the PnP shared-probability attention (pnp_utils.py:57-67, 75-90: q / k of the source sample, v per sample) computes
    the live rows only, like the un-shared path (cfg-3 attention 46.5 -> 35.8 ms).  `global_rand=0` puts the local chunk
    on the src side of every global level.  Checked: the shared launch really ran on a compacted query list; block outputs
    equal the full-layout run (VIDTOME_LIVE_QUERIES=0) to 1e-3 of the output scale and the oracle's shared-probability
    rows at sampled positions; without align_batch the full layout is kept."""
    import vidtome_amd
    from vidtome_amd import _lib
    from vidtome_amd import patch as vpatch
    from vidtome_amd import sites as S
    B, F, latent = 3, 8, (32, 32)
    sl = [s for s in S.sd15_sites() if s.name in ("up3.0", "up2.0")]
    res = {}
    keep_live, orig_kv = vpatch.LIVE_QUERIES, _lib.attention_kv
    try:
        for aligned, live in ((True, True), (True, False), (False, True)):
            vpatch.LIVE_QUERIES = live
            unet = S.SiteUNet(sl, seed=4).to(device=DEV, dtype=torch.float16)
            for blk in unet.blocks:      # what pnp.register_attention_control sets on the decoder blocks
                blk.attn1.injection_schedule, blk.attn1.t, blk.attn1.vtm_num_inputs = [981], 981, B
            vidtome_amd.apply_patch(unet, local_merge_ratio=0.5, merge_global=True, global_merge_ratio=0.5, batch_size=B,
                                    align_batch=aligned, global_rand=0.0)
            unet.set_size(latent)
            torch.manual_seed(7)
            shared_kv = []

            def spy(q, k, vt, heads, Mq, Mk, scale, *a, **kw):
                if kw.get("share_groups", 1) != 1:
                    shared_kv.append((Mq, Mk))
                return orig_kv(q, k, vt, heads, Mq, Mk, scale, *a, **kw)

            _lib.attention_kv = spy
            passes = []
            for c in range(2):           # chunk 0 leaves the anchors, chunk 1 merges with them
                hiddens = [S.synthetic_hidden(s_, B, F, latent, torch.float16, DEV, seed=90 + 10 * c + i, clip_seed=3 + i, regime="corr01")
                           for i, s_ in enumerate(sl)]
                seen, orig_cm = {}, vpatch.compute_merge

                def rec(module, x, info, **kw):
                    r = orig_cm(module, x, info, **kw)
                    seen[id(module)] = r[0].plan
                    return r

                vpatch.compute_merge = rec
                try:
                    with torch.no_grad():
                        outs = S.run_segment_pass(unet, hiddens)
                finally:
                    vpatch.compute_merge = orig_cm
                plans = [seen[id(blk)] for blk in unet.blocks]
                assert all(bool(torch.isfinite(o).all()) for o in outs)
                assert all((p.global_level is not None) == (c == 1) for p in plans)
                passes.append((hiddens, outs, plans))
            _lib.attention_kv = orig_kv
            torch.cuda.synchronize()
            res[(aligned, live)] = (passes, list(shared_kv))
            if aligned and live:
                hiddens, outs, plans = passes[1]
                for i in range(len(sl)):
                    assert plans[i].q_rows is not None and plans[i].aligned
                    assert torch.equal(plans[i].q_rows[0], plans[i].q_rows[1]) and torch.equal(plans[i].q_rows[0], plans[i].q_rows[2])
                    _block_rows_vs_oracle(oracle, unet.blocks[i], plans[i], hiddens[i], outs[i], F, share=B, seed=i)
            vidtome_amd.remove_patch(unet)
    finally:
        vpatch.LIVE_QUERIES, _lib.attention_kv = keep_live, orig_kv
    # the shared launches of chunk 1 ran on the local rows only (Mq = the local merged tokens < Mk), one per site
    assert len(res[(True, True)][1]) == len(sl) and all(mq < mk for mq, mk in res[(True, True)][1])
    assert res[(True, False)][1] == [] and res[(False, True)][1] == []
    for (_, oa, _), (_, ob, _) in zip(res[(True, True)][0], res[(True, False)][0]):
        for a, b in zip(oa, ob):
            scale = float(b.float().abs().max())
            assert float((a.float() - b.float()).abs().max()) <= 1e-3 * scale


# ---------------------------------------------------------------------------------------------------
# full-size, size-independent properties (cfg-2 top block: B=2, F=16, N=4096, C=320)
# ---------------------------------------------------------------------------------------------------
def test_full_size_properties(L):
    import vidtome_amd
    from vidtome_amd import patch as vpatch

    class Blk(torch.nn.Module):
        pass

    B, F, N, C = 2, 16, 4096, 320
    g = torch.Generator().manual_seed(1234)
    base = torch.randn(B, 1, N, C, generator=g)
    x = (base + 0.5 * torch.randn(B, F, N, C, generator=g)).reshape(B * F, N, C).half().to(DEV)
    blk = Blk()
    blk.generator = torch.Generator().manual_seed(123)
    info = {"size": (64, 64), "args": dict(max_downsample=2, generator=None, seed=123, batch_size=B, align_batch=False,
                                           merge_global=True, global_merge_ratio=0.5, local_merge_ratio=0.5,
                                           global_rand=0.5, target_stride=4)}
    for step in range(3):            # chunk 0 stores anchors, chunks 1-2 merge against them
        m, u, merged = vpatch.compute_merge(blk, x, info, want_indices=True)
        plan = m.plan
        Lj = F * N
        assert plan.levels[0].Ns == 49152 and plan.levels[0].Nd == 16384 and plan.levels[0].r == 24576
        assert plan.levels[1].Ns == 12288 and plan.levels[1].Nd == 28672 and plan.levels[1].r == 6144
        M = plan.M
        assert M == (34816 if step == 0 else 52224)
        gm, inv = plan.gather_map, plan.inv
        # every merged row is a pool row; the maps are consistent: pool[gather_map[inv[i]]] is the row that
        # replaces token i, and tokens that survive (unmerged / dst) map back to THEMSELVES
        assert int(gm.min()) >= 0 and int(inv.min()) >= 0 and int(inv.max()) < M
        back = torch.gather(gm.long(), 1, inv.long())                    # (B, L) pool id restored at each position
        ar = torch.arange(Lj, device=DEV)[None].expand(B, -1)
        kept = back == ar
        n_kept_expected = 34816 if step == 0 else None
        if step == 0:
            assert int(kept.sum(1)[0]) == 34816                         # exactly the merged rows survive
            # idempotence: unmerge(merge(x)) == x on surviving rows, and merge(unmerge(y)) == y
            xj = x.reshape(B, Lj, C)
            rt = u(merged).reshape(B, Lj, C)
            assert torch.equal(rt[kept], xj[kept])
            y = torch.randn(B, merged.shape[1], C, generator=g).half().to(DEV)
            assert torch.equal(m(u(y))[:, :M], y[:, :M])
        # each src row is restored from a dst row of a DIFFERENT position; sorted-ness of level-1 ranking
        lv = plan.levels[0]
        from vidtome_amd import _lib
        nm, _ = _lib.decode_best(lv.best)
        ranked = torch.gather(nm, 1, torch.cat([lv.src_idx, lv.unm_idx], 1).long())
        assert bool((ranked[:, 1:] <= ranked[:, :-1]).all())            # descending similarity order
        assert torch.equal(torch.sort(torch.cat([lv.src_idx, lv.unm_idx], 1), 1).values,
                           torch.arange(lv.Ns, device=DEV, dtype=torch.int32)[None].expand(B, -1))


def test_compute_merge_reference_fuzz_gpu(L, order_rule):
    """The same 115 reference-generated configurations as tests/test_oracle_golden.py::test_oracle_vs_reference_fuzz,
    through the HIP planner: merged tokens, stored anchor tokens and u(merged) hash-equal to the reference's."""
    import os
    from inputs import fuzz_hash, fuzz_inputs, load_fuzz_configs
    from vidtome_amd import patch as vpatch

    class Blk(torch.nn.Module):
        pass

    cfgs = load_fuzz_configs(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_compute_merge.npz"))
    for cfg in cfgs:
        blk = Blk()
        blk.generator = torch.Generator().manual_seed(int(cfg["gen_seed"]))
        info = {"size": (cfg["H"], cfg["W"]),
                "args": dict(max_downsample=2, generator=None, seed=123, batch_size=cfg["B"], align_batch=bool(cfg["align"]),
                             merge_global=bool(cfg["merge_global"]), global_merge_ratio=cfg["global_ratio"],
                             local_merge_ratio=cfg["local_ratio"], global_rand=cfg["global_rand"], target_stride=4)}
        for ck, x in enumerate(fuzz_inputs(cfg)):
            m, u, merged = vpatch.compute_merge(blk, x.to(DEV), info)
            plan = getattr(m, "plan", None)
            M = plan.M if plan is not None else merged.shape[1]
            want = cfg["hashes"][ck]
            assert fuzz_hash(merged[:, :M].cpu().numpy()) == want[0], (cfg, ck)
            if want[1]:
                gt = blk.global_tokens
                assert fuzz_hash(gt.cpu().numpy()) == want[1], (cfg, ck)
            assert fuzz_hash(u(merged).cpu().numpy()) == want[2], (cfg, ck)


def test_compute_merge_fuzz_vs_oracle(L, oracle, order_rule):
    """End-to-end planner fuzz: random (B, F per chunk, token grid, C, ratios, align_batch, global merging, coin
    threshold) over three consecutive chunks of a block; merged tokens, anchor tokens and the unmerge of a random
    tensor must equal the CPU oracle's (patch.py:14-91 restated) exactly -- they are row copies, so any index
    difference shows, and both sides draw from identically seeded generators."""
    from vidtome_amd import patch as vpatch

    class Blk(torch.nn.Module):
        pass

    g = torch.Generator().manual_seed(31)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    for case in range(40):
        B = ri(1, 3)
        H, W = [(8, 8), (6, 10), (12, 4)][ri(0, 2)]
        ds = ri(1, 2)
        N = (H // ds) * (W // ds)
        C = [8, 16, 40][ri(0, 2)]
        args = dict(max_downsample=2, generator=None, seed=123, batch_size=B, align_batch=bool(ri(0, 1)),
                    merge_global=bool(ri(0, 3)), global_merge_ratio=[0.3, 0.5, 0.8, 1.0][ri(0, 3)],
                    local_merge_ratio=[0.3, 0.5, 0.9, 1.0][ri(0, 3)], global_rand=[0.0, 0.5, 1.0][ri(0, 2)],
                    target_stride=4)
        seed = ri(0, 10 ** 6)
        blk = Blk()
        blk.generator = torch.Generator().manual_seed(seed)
        draws = oracle.RandomDraws.from_torch_generator(torch.Generator().manual_seed(seed))
        state = {}
        info = {"size": (H, W), "args": dict(args)}
        for chunk in range(3):
            F = ri(1, 9)
            x = torch.randn(B * F, N, C, generator=g)
            m_o, u_o, merged_o, _ = oracle.compute_merge(x.numpy(), (H, W), args, draws, state)
            m, u, merged = vpatch.compute_merge(blk, x.to(DEV), info)
            M = merged_o.shape[1]
            tag = (case, chunk, B, F, N, C, args)
            assert np.array_equal(merged[:, :M].cpu().numpy(), merged_o), tag
            if args["merge_global"]:
                assert np.array_equal(blk.global_tokens[:, :state["global_tokens"].shape[1]].cpu().numpy(),
                                      state["global_tokens"]), tag
            y = torch.randn(merged_o.shape, generator=g)
            yp = torch.zeros(merged.shape)
            yp[:, :M] = y
            assert np.array_equal(u(yp.to(DEV)).cpu().numpy(), u_o(y.numpy())), tag


# ---------------------------------------------------------------------------------------------------
# caller-side tail of a step: CFG combine + DDIM update vs the reference's pred_next_x
# ---------------------------------------------------------------------------------------------------
def test_cfg_ddim_golden_gpu(L):
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ddim.npz"))
    for k in range(int(z["n"])):
        _, _, inversion, guidance, mu, sigma, mu_p, sigma_p = z[f"{k}/meta"].tolist()   # generate.py:299-302
        coef = (mu_p, sigma_p, mu, sigma) if inversion else (mu, sigma, mu_p, sigma_p)   # generate.py:304-309
        x, eu, ec = (_t(z[f"{k}/{n}"]) for n in ("x", "eu", "ec"))
        xn, eps = L.cfg_ddim(x, eu, ec, guidance, *coef, want_eps=True)
        assert np.array_equal(eps.cpu().numpy(), z[f"{k}/eps"]), k
        assert np.array_equal(xn.cpu().numpy(), z[f"{k}/xn"]), k
