"""World-size-2 gloo tests (CPU) of the chunk-parallel anchor exchange: the protocol objects in
vidtome_amd/chunk_parallel.py move the right tensors and -- with the draw replay -- a ring run over 2 ranks
reproduces the sequential run (reference order) exactly.  The merge itself is computed by the CPU oracle here
(tests may use it); on the GPU box the same exchange objects are driven by the HIP path."""
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
CHUNKS = [6, 4]          # frames per chunk: F=6 makes the level sizes depend on the randf draw
NBLK = 2


def _hidden(chunk, blk):
    g = torch.Generator().manual_seed(100 * chunk + blk)
    return torch.randn(B * CHUNKS[chunk], HW[0] * HW[1], C, generator=g).numpy()


def _fork(seed=123):
    torch.manual_seed(seed)
    return torch.Generator(device="cpu").set_state(torch.get_rng_state())


def _sequential(oracle):
    res = {}
    for blk in range(NBLK):
        draws = oracle.RandomDraws.from_torch_generator(_fork())
        state = {"global_tokens": None}
        for ck in range(len(CHUNKS)):
            m, u, merged, trace = oracle.compute_merge(_hidden(ck, blk), HW, ARGS, draws, state)
            res[(ck, blk)] = (merged.copy(), state["global_tokens"].copy(), trace)
    return res


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import oracle
        from vidtome_amd import chunk_parallel as cp
        seq = _sequential(oracle)
        tpf = HW[0] * HW[1]
        ok = True
        if mode == "ring":
            ex = cp.RingExchange()
            for blk in range(NBLK):
                gen = _fork()
                cp.replay_draws(gen, CHUNKS[:rank], tpf, ARGS)          # draws of the chunks before mine
                draws = oracle.RandomDraws.from_torch_generator(gen)
                like = torch.zeros(1)
                got = ex.anchors_for(f"b{blk}", lambda: None, like)
                state = {"global_tokens": None if got is None else got.numpy()}
                m, u, merged, trace = oracle.compute_merge(_hidden(rank, blk), HW, ARGS, draws, state)
                ex.publish(f"b{blk}", torch.from_numpy(state["global_tokens"]))
                ref_merged, ref_anchors, ref_trace = seq[(rank, blk)]
                ok &= np.array_equal(merged, ref_merged) and np.array_equal(state["global_tokens"], ref_anchors)
                for lv, rl in zip(trace["levels"], ref_trace["levels"]):
                    ok &= all(np.array_equal(lv[n], rl[n]) for n in ("unm_idx", "src_idx", "dst_idx"))
                ok &= (trace["global"] is None) == (ref_trace["global"] is None)
                if trace["global"] is not None:
                    ok &= all(np.array_equal(trace["global"][n], ref_trace["global"][n])
                              for n in ("unm_idx", "src_idx", "dst_idx"))
                    ok &= trace["global"]["coin"] == ref_trace["global"]["coin"]
            ok &= (ex.bytes_sent > 0) == (rank == 0)
        else:
            ex = cp.AllGatherExchange()
            local_args = dict(ARGS, merge_global=False)
            for blk in range(NBLK):
                # every rank runs the SAME chunk length here so the gathered tensors have one shape
                def local_of(r):
                    draws = oracle.RandomDraws.from_torch_generator(_fork())
                    h = _hidden(1, blk) + np.float32(r)
                    return oracle.compute_merge(h, HW, local_args, draws, {})[2]
                mine = torch.from_numpy(local_of(rank))
                got = ex.anchors_for(f"b{blk}", lambda: mine, mine)
                ok &= np.array_equal(got.numpy(), local_of((rank - 1) % world))
            ok &= ex.bytes_gathered > 0
        q.put((rank, bool(ok), ""))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["ring", "allgather"])
def test_two_rank_exchange_gloo(oracle, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, err in results:
        assert ok, f"rank {rank} failed: {err}"


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
