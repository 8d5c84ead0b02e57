"""CPU, world_size 2 over gloo: the host logic of the multi-GPU path (frame sharding, flat gradient all-reduce with
grad=None handling, parameter broadcast).  The GPU build uses the same code with backend nccl (= RCCL)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "4d-facial-avatars_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerf import distributed as D
        import nerf
        torch.manual_seed(10 + rank)                      # different init per rank on purpose
        m = nerf.models.ConditionalBlendshapePaperNeRFModel(num_encoding_fn_xyz=10, num_encoding_fn_dir=4,
                                                            include_input_xyz=True, include_input_dir=False)
        table = torch.nn.Parameter(torch.randn(6, 32))
        params = list(m.parameters()) + [table]
        D.broadcast_parameters(params, src=0)
        chk = torch.stack([p.detach().double().sum() for p in params])
        gathered = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(gathered, chk)
        same_init = all(torch.equal(gathered[0], g) for g in gathered)
        # rank-dependent synthetic gradients; layers_dir.3 has none anywhere; rank r touches latent row r only
        for i, p in enumerate(params[:-1]):
            name = list(dict(m.named_parameters()))[i]
            p.grad = None if name.startswith("layers_dir.3") else torch.full_like(p, float(rank + 1) * (i + 1))
        table.grad = torch.zeros_like(table)
        table.grad[rank] = float(rank + 1)
        red = D.GradientAllReducer(params)
        red.enable_timing()
        red.reduce()
        red.reduce()                                      # (averaging the averaged gradients again leaves them unchanged)
        st = red.stats()
        live = sum(p.numel() for n_, p in m.named_parameters() if not n_.startswith("layers_dir.3")) + table.numel()
        ok = st["ranks_seen"] == world and st["bytes_allreduced"] == 4 * live and st["calls"] == 2 and st["allreduce_us"] > 0
        ok &= st["backend"] == "gloo"
        for i, p in enumerate(params[:-1]):
            name = list(dict(m.named_parameters()))[i]
            if name.startswith("layers_dir.3"):
                ok &= p.grad is None
            else:
                ok &= bool(torch.allclose(p.grad, torch.full_like(p, (i + 1) * sum(r + 1 for r in range(world)) / world)))
        want = torch.zeros(6, 32)
        for r in range(world):
            want[r] = (r + 1) / world
        ok &= bool(torch.allclose(table.grad, want))
        frames = D.shard_frames(11)
        q.put((rank, same_init, ok, frames, D.rank_seed(42)))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_gradient_allreduce_and_frame_shards():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    assert all(r[1] for r in res), "broadcast_parameters did not equalise the ranks"
    assert all(r[2] for r in res), "averaged gradients are wrong"
    assert res[0][3] == [0, 2, 4, 6, 8, 10] and res[1][3] == [1, 3, 5, 7, 9]
    assert sorted(res[0][3] + res[1][3]) == list(range(11))
    assert res[0][4] == 42 and res[1][4] == 43


def test_single_process_is_a_noop():
    sys.path.insert(0, os.path.join(ROOT, "4d-facial-avatars_amd"))
    from nerf import distributed as D
    assert D.shard_frames(5) == [0, 1, 2, 3, 4]
    assert D.gather_frame_order(5, 2) == [(0, 0), (1, 0), (0, 1), (1, 1), (0, 2)]
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.ones(3)
    D.GradientAllReducer([p]).reduce()
    assert torch.equal(p.grad, torch.ones(3))
