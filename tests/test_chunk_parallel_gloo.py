"""World-size-2 gloo tests (CPU) of the chunk-parallel anchor exchange (vidtome_amd/chunk_parallel.py): a step of
FIVE chunks of unequal length (more chunks than ranks, a single-frame chunk, a 6-frame chunk whose level sizes depend
on the draw) dealt round-robin to two ranks must reproduce the sequential run -- indices, merged tokens, anchors and
the generator state at the end of the step -- in the exact ring mode, and the documented "parallel anchors" semantics
in the neighbour / all-gather modes.  The merge itself is computed by the CPU oracle here (tests may use it) through
the same protocol calls `patch.compute_merge` makes; tests/test_gpu_parity.py drives the HIP path through them."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ARGS = {"batch_size": 2, "max_downsample": 2, "target_stride": 4, "local_merge_ratio": 0.5,
        "merge_global": True, "global_merge_ratio": 0.5, "align_batch": False, "global_rand": 0.5}
B, C, HW = 2, 16, (4, 4)
NBLK = 2
STEPS = {"ring": [[3, 6, 4, 1, 4], [2, 4, 4]], "neighbour": [[3, 6, 4, 1, 4], [2, 4, 4]], "allgather": [[3, 6, 4, 1], [4, 2]]}


def _hidden(step, chunk, blk, frames):
    g = torch.Generator().manual_seed(1000 * step + 100 * chunk + blk)
    return torch.randn(B * frames, HW[0] * HW[1], C, generator=g).numpy()


def _fork(seed=123):
    torch.manual_seed(seed)
    return torch.Generator(device="cpu").set_state(torch.get_rng_state())


def _local_tokens(oracle, h, gen):
    """The chunk's tokens after its local levels, computed with a COPY of the generator (same draws)."""
    g = torch.Generator(device="cpu").set_state(gen.get_state())
    return oracle.compute_merge(h, HW, dict(ARGS, merge_global=False), oracle.RandomDraws.from_torch_generator(g), {})[2]


def _reference(oracle, mode, steps):
    """Sequential run (generate.py:215-219 order).  ring: the reference's chained anchors; otherwise every chunk
    merges against the LOCAL tokens of the chunk before it."""
    res, end_state = {}, {}
    for blk in range(NBLK):
        gen = _fork()
        draws = oracle.RandomDraws.from_torch_generator(gen)
        for s, frames in enumerate(steps):
            state, prev_local = {"global_tokens": None}, None
            for ck, F in enumerate(frames):
                h = _hidden(s, ck, blk, F)
                local = _local_tokens(oracle, h, gen)
                if mode != "ring":
                    state = {"global_tokens": prev_local}
                m, u, merged, trace = oracle.compute_merge(h, HW, ARGS, draws, state)
                res[(s, ck, blk)] = (merged.copy(), state["global_tokens"].copy(), trace)
                prev_local = local
        end_state[blk] = gen.get_state()
    return res, end_state


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _map_of(like, local):
    """(B, M_local) int32 rows of ``like`` (B, L, C) that ``local`` (B, M_local, C) -- row copies of them -- came from:
    what patch.compute_merge hands the exchange as the composed local merge map."""
    like = like.numpy() if hasattr(like, "numpy") else like
    if local.shape[1] == like.shape[1]:
        return None                                    # no local level (single-frame chunk): the tokens ARE the chunk
    out = np.empty(local.shape[:2], np.int32)
    for b in range(like.shape[0]):
        index = {row.tobytes(): j for j, row in enumerate(like[b])}
        out[b] = [index[row.tobytes()] for row in local[b]]
    return torch.from_numpy(out)


class _Module:                     # what the exchange needs from a patched block: its generator
    def __init__(self, gen):
        self.generator = gen


def _worker(rank, world, port, mode, q, early=True):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import oracle
        from vidtome_amd import chunk_parallel as cp
        steps = STEPS[mode]
        ref, ref_end = _reference(oracle, mode, steps)
        tpf = HW[0] * HW[1]
        ex = cp.AnchorExchange(mode)
        ex.early = early
        mods = [_Module(_fork()) for _ in range(NBLK)]
        ok, why = True, []

        def check(cond, what):
            nonlocal ok
            if not cond:
                ok = False
                why.append(what)

        for s, frames in enumerate(steps):
            ex.begin_step(frames)
            mine = ex.my_chunks()
            check(mine == list(range(rank, len(frames), world)), "chunk assignment")
            for i in mine:
                ex.begin_chunk(i)
                for blk in range(NBLK):
                    key, mod = f"b{blk}", mods[blk]
                    h = _hidden(s, i, blk, frames[i])
                    like = torch.from_numpy(h).reshape(B, frames[i] * tpf, C)
                    ex.begin_block(mod, key, frames[i], tpf, ARGS, like)          # what patch.compute_merge does
                    local = _local_tokens(oracle, h, mod.generator)

                    class State(dict):
                        def get(self, k, d=None):
                            if k != "global_tokens":
                                return dict.get(self, k, d)
                            got = ex.anchors_for(key, lambda: torch.from_numpy(local), like, _map_of(like, local))
                            return None if got is None else got.numpy()
                    state = State()
                    m, u, merged, trace = oracle.compute_merge(h, HW, ARGS, oracle.RandomDraws.from_torch_generator(mod.generator), state)
                    ex.publish(key, torch.from_numpy(np.ascontiguousarray(state["global_tokens"])))
                    r_merged, r_anchors, r_trace = ref[(s, i, blk)]
                    check(np.array_equal(merged, r_merged), f"merged s{s} c{i} b{blk}")
                    check(np.array_equal(state["global_tokens"], r_anchors), f"anchors s{s} c{i} b{blk}")
                    for lv, rl in zip(trace["levels"], r_trace["levels"]):
                        check(all(np.array_equal(lv[n], rl[n]) for n in ("unm_idx", "src_idx", "dst_idx")), "local idx")
                    check((trace["global"] is None) == (r_trace["global"] is None) == (i == 0), "global presence")
                    if trace["global"] is not None and r_trace["global"] is not None:
                        check(all(np.array_equal(trace["global"][n], r_trace["global"][n])
                                  for n in ("unm_idx", "src_idx", "dst_idx")), "global idx")
                        check(trace["global"]["coin"] == r_trace["global"]["coin"], "coin")
            ex.end_step()
        for blk in range(NBLK):      # every rank's generators end where the sequential run's do
            check(torch.equal(mods[blk].generator.get_state(), ref_end[blk]), f"generator state b{blk}")
        check(ex.bytes_received > 0, "nothing received")
        if mode != "allgather":
            check(ex.bytes_sent > 0, "nothing sent")
        q.put((rank, bool(ok), "; ".join(why)))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


STEPS4 = {"ring": [[3, 6, 4, 1, 4], [2, 4, 4]], "neighbour": [[3, 6, 4, 1, 4, 2], [2, 4, 4]], "allgather": [[3, 6, 4, 1], [4, 2, 2, 5]]}


def _worker4(rank, world, port, mode, q):
    STEPS[mode] = STEPS4[mode]          # (module global of this spawned process only)
    _worker(rank, world, port, mode, q, True)


@pytest.mark.parametrize("mode", ["ring", "neighbour", "neighbour-single-message", "allgather"])
def test_two_rank_exchange_gloo(oracle, mode):
    """neighbour: the early hand-over (joined chunk when the block starts, composed map after the local levels, the
    receiver gathers) and the single-message form (merged tokens after the local levels) give the same anchors."""
    early = mode != "neighbour-single-message"
    mode = mode.split("-")[0]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q, early)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, err in results:
        assert ok, f"rank {rank} failed: {err}"


@pytest.mark.parametrize("mode", ["ring", "neighbour", "allgather"])
def test_four_rank_exchange_gloo(oracle, mode):
    """World size 4 (the per-pair communicators of a real ring: no wrap-around group as with two ranks; idle ranks in the
    3-chunk step; the all-gather pads to the round's longest chunk)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker4, args=(r, 4, port, mode, q)) for r in range(4)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, err in results:
        assert ok, f"rank {rank} failed: {err}"


def _worker8(rank, world, port, mode, q):
    STEPS[mode] = [[3, 6, 4, 1, 4, 2, 5, 4, 3, 2], [2, 4, 4]]
    _worker(rank, world, port, mode, q, True)


def test_eight_rank_neighbour_gloo(oracle):
    """The world size of the driver's scaling run (8 ranks, bench.py's default exchange with the early hand-over): a
    10-chunk step (two rounds, the second partial) and a 3-chunk step (five ranks idle)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, "neighbour", q)) for r in range(8)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, err in results:
        assert ok, f"rank {rank} failed: {err}"


@pytest.mark.parametrize("W", [1, 3])
def test_fake_ranks_in_process(oracle, W):
    """the single-process N-fake-rank replay -- W exchange endpoints on an in-memory transport,
    the chunks executed in schedule order, equal the sequential run (ring) bit for bit; W = 1 is the degenerate
    hand-over in place."""
    from vidtome_amd import chunk_parallel as cp
    steps = STEPS["ring"]
    for mode in ("ring", "neighbour"):
        ref, ref_end = _reference(oracle, mode, steps)
        tpf = HW[0] * HW[1]
        fabric = cp.LocalTransport.fabric(W)
        exs = [cp.AnchorExchange(mode, transport=t) for t in fabric]
        mods = [[_Module(_fork()) for _ in range(NBLK)] for _ in range(W)]
        for s, frames in enumerate(steps):
            for ex in exs:
                ex.begin_step(frames)
            for i, F in enumerate(frames):
                ex, r = exs[i % W], i % W
                ex.begin_chunk(i)
                for blk in range(NBLK):
                    key, mod = f"b{blk}", mods[r][blk]
                    h = _hidden(s, i, blk, F)
                    like = torch.from_numpy(h).reshape(B, F * tpf, C)
                    ex.begin_block(mod, key, F, tpf, ARGS, like)
                    local = _local_tokens(oracle, h, mod.generator)

                    class State(dict):
                        def get(self, k, d=None):
                            got = ex.anchors_for(key, lambda: torch.from_numpy(local), like, _map_of(like, local))
                            return None if got is None else got.numpy()
                    state = State()
                    merged = oracle.compute_merge(h, HW, ARGS, oracle.RandomDraws.from_torch_generator(mod.generator), state)[2]
                    ex.publish(key, torch.from_numpy(np.ascontiguousarray(state["global_tokens"])))
                    assert np.array_equal(merged, ref[(s, i, blk)][0]), (mode, s, i, blk)
                    assert np.array_equal(state["global_tokens"], ref[(s, i, blk)][1]), (mode, s, i, blk)
            for ex in exs:
                ex.end_step()
        for r in range(W):
            for blk in range(NBLK):
                assert torch.equal(mods[r][blk].generator.get_state(), ref_end[blk])


def test_simulated_draws_match_compute_merge(oracle):
    """simulate_block_draws consumes exactly the draws (and predicts the sizes) of a real compute_merge."""
    from vidtome_amd import chunk_parallel as cp
    tpf = HW[0] * HW[1]
    for frames in (1, 2, 3, 4, 6, 8, 16):
        for has_anchors in (False, True):
            g1, g2 = _fork(7), _fork(7)
            state = {"global_tokens": np.random.default_rng(0).standard_normal((B, 24, C)).astype(np.float32)
                     if has_anchors else None}
            gg = torch.Generator().manual_seed(frames)
            x = torch.randn(B * frames, tpf, C, generator=gg).numpy()
            m, u, merged, trace = oracle.compute_merge(x, HW, ARGS, oracle.RandomDraws.from_torch_generator(g1), state)
            sim = cp.simulate_block_draws(g2, frames, tpf, ARGS, has_anchors)
            assert torch.equal(g1.get_state(), g2.get_state())
            assert sim["randf"] == [lv["randf"] for lv in trace["levels"]]
            m_local = trace["global"]["src_len"] if (trace["global"] and trace["global"]["local_chunk"] == 0) else None
            if trace["global"] is None:
                assert sim["M_local"] == merged.shape[1]
            elif m_local is not None:
                assert sim["M_local"] == m_local


def test_long_stream_releases_finished_sends(oracle):
    """bench.py runs its whole N > 1 measurement as ONE step of many chunks: the exchange must not keep every sent tensor
    alive until the end of it (finished transfers are dropped once more than 64 are on the books)."""
    from vidtome_amd import chunk_parallel as cp
    frames = [2] * 80
    tpf = HW[0] * HW[1]
    fabric = cp.LocalTransport.fabric(2)
    exs = [cp.AnchorExchange("neighbour", transport=t) for t in fabric]
    mods = [[_Module(_fork()) for _ in range(NBLK)] for _ in range(2)]
    for ex in exs:
        ex.begin_step(frames)
    for i, F in enumerate(frames):
        ex, r = exs[i % 2], i % 2
        ex.begin_chunk(i)
        for blk in range(NBLK):
            key, mod = f"b{blk}", mods[r][blk]
            h = _hidden(0, i % 5, blk, F)
            like = torch.from_numpy(h).reshape(B, F * tpf, C)
            ex.begin_block(mod, key, F, tpf, ARGS, like)
            local = _local_tokens(oracle, h, mod.generator)

            class State(dict):
                def get(self, k, d=None):
                    got = ex.anchors_for(key, lambda: torch.from_numpy(local), like, _map_of(like, local))
                    return None if got is None else got.numpy()
            state = State()
            oracle.compute_merge(h, HW, ARGS, oracle.RandomDraws.from_torch_generator(mod.generator), state)
            ex.publish(key, torch.from_numpy(np.ascontiguousarray(state["global_tokens"])))
        assert len(ex._inflight) <= 65
    for ex in exs:
        ex.end_step()
        assert not ex._inflight
    assert torch.equal(mods[0][0].generator.get_state(), mods[1][0].generator.get_state())


class _LateModule:                 # a patched block before its first forward: no generator yet
    pass


@pytest.mark.parametrize("mode", ["ring", "neighbour"])
def test_idle_ranks_keep_the_sequential_draw_stream(oracle, mode):
    """A step may have fewer chunks than ranks (the first chunk of a step has a random length, generate.py:176-178), so a
    rank can sit out a whole step -- here rank 2 of 3 in the first AND the third step, rank 1 in the third -- before it
    runs a block for the first time.  Its generators are forked at the first begin_step (registered blocks, same point
    of the global stream on every rank) and the block replays the steps it missed when it first appears, so merged
    tokens, anchors and the final generator states still equal the sequential run's.  The 6- and 5-frame chunks make the
    merged length depend on the draws: a rank on a wrong stream would also post receives of the wrong shape."""
    from vidtome_amd import chunk_parallel as cp
    steps = [[3, 6], [2, 4, 4, 5, 6], [6], [5, 3, 6]]
    W = 3
    ref, ref_end = _reference(oracle, mode, steps)
    tpf = HW[0] * HW[1]
    fabric = cp.LocalTransport.fabric(W)
    exs = [cp.AnchorExchange(mode, transport=t) for t in fabric]
    mods = [[_LateModule() for _ in range(NBLK)] for _ in range(W)]
    for r in range(W):
        for blk in range(NBLK):
            exs[r].register(f"b{blk}", mods[r][blk])
    torch.manual_seed(123)                                   # what _fork() seeds in _reference
    for s, frames in enumerate(steps):
        for ex in exs:
            ex.begin_step(frames)
        if s == 0:
            torch.randperm(7)                                # later draws from the global stream must not matter
        for i, F in enumerate(frames):
            ex, r = exs[i % W], i % W
            ex.begin_chunk(i)
            for blk in range(NBLK):
                key, mod = f"b{blk}", mods[r][blk]
                h = _hidden(s, i, blk, F)
                like = torch.from_numpy(h).reshape(B, F * tpf, C)
                ex.begin_block(mod, key, F, tpf, ARGS, like)
                local = _local_tokens(oracle, h, mod.generator)

                class State(dict):
                    def get(self, k, d=None):
                        got = ex.anchors_for(key, lambda: torch.from_numpy(local), like, _map_of(like, local))
                        return None if got is None else got.numpy()
                state = State()
                merged = oracle.compute_merge(h, HW, ARGS, oracle.RandomDraws.from_torch_generator(mod.generator), state)[2]
                ex.publish(key, torch.from_numpy(np.ascontiguousarray(state["global_tokens"])))
                assert np.array_equal(merged, ref[(s, i, blk)][0]), (mode, s, i, blk)
                assert np.array_equal(state["global_tokens"], ref[(s, i, blk)][1]), (mode, s, i, blk)
        for ex in exs:
            ex.end_step()
    # rank 2 never ran a chunk in steps 0 and 2; every block it DID run ends on the sequential stream
    for r in range(W):
        for blk in range(NBLK):
            st = exs[r]._blocks.get(f"b{blk}")
            assert st is not None and st.steps_done == len(steps)
            assert torch.equal(mods[r][blk].generator.get_state(), ref_end[blk]), (r, blk)
