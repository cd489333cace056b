"""N>1 path on CPU: world_size-2 gloo processes exercise utterance sharding and the flat
gradient all-reduce / NaN-flag agreement / global-norm clip used for data-parallel training."""
import os
import pytest
import socket
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sincformer_metacog_speech_enhancement_amd import dp
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 4))
    sync = dp.FlatGradSynchronizer(model.parameters(), bucket_bytes=256)      # several buckets
    assert len(sync.buckets) > 1
    # rank-dependent gradients written through the views
    sync.zero()
    for i, p in enumerate(model.parameters()):
        p.grad.add_(float(rank + 1) * (i + 1))
    r = sync.sync(loss=torch.tensor(1.0), max_norm=None)
    got = [float(p.grad.flatten()[0]) for p in model.parameters()]
    exp = [(1 + 2) / 2.0 * (i + 1) for i in range(4)]
    ok = np.allclose(got, exp) and not r["skip"]
    # clip on the reduced gradient: same coefficient on both ranks
    sync.zero()
    for p in model.parameters():
        p.grad.add_(float(rank + 1))
    r2 = sync.sync(loss=torch.tensor(0.5), max_norm=5.0)
    n = sum(p.numel() for p in model.parameters())
    ok = ok and abs(r2["grad_norm"] - 1.5 * np.sqrt(n)) < 1e-4 and abs(float(torch.linalg.vector_norm(sync.flat)) - 5.0) < 1e-3
    # a NaN loss on ONE rank makes BOTH ranks skip the step
    sync.zero()
    r3 = sync.sync(loss=torch.tensor(float("nan")) if rank == 1 else torch.tensor(1.0))
    ok = ok and r3["skip"]
    s, e = dp.shard_range(7, rank, world)
    q.put((rank, ok, (s, e)))
    dist.destroy_process_group()


def test_flat_grad_allreduce_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    assert res[0][2] == (0, 4) and res[1][2] == (4, 7)


def test_shard_range_covers_everything():
    from sincformer_metacog_speech_enhancement_amd import dp
    for n in (1, 7, 64, 255):
        for w in (1, 2, 4, 8):
            spans = [dp.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= 1


def _overlap_worker(rank, world, port, q, steal=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sincformer_metacog_speech_enhancement_amd import dp
    torch.manual_seed(0)                                     # same weights on both ranks
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(),
                                torch.nn.Linear(16, 4))
    sync = dp.FlatGradSynchronizer(model.parameters(), bucket_bytes=400, overlap=True, steal=steal)
    nb = len(sync.buckets)
    torch.manual_seed(100 + rank)                            # different data per rank (the utterance shard)
    x, y = torch.randn(5, 8), torch.randn(5, 4)
    ok = nb >= 3
    for it in range(2):                                      # twice: zero() must re-arm the hooks' counters
        sync.zero()
        loss = (model(x) - y).pow(2).mean()
        loss.backward()
        launched_in_backward = sum(sync._launched)
        ok = ok and launched_in_backward == nb - 1           # every bucket but the flag-carrying one went out from a hook
        sync.finish(loss)
        got = sync.flat.clone() / world
        # reference: gather both ranks' local gradients computed without the synchronizer
        ref_model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(),
                                        torch.nn.Linear(16, 4))
        ref_model.load_state_dict(model.state_dict())
        (ref_model(x) - y).pow(2).mean().backward()
        local = torch.cat([p.grad.reshape(-1) for p in ref_model.parameters()])
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        ok = ok and torch.allclose(got, sum(gathered) / world, atol=1e-6)
        ok = ok and float(sync.flag) == 0.0
        # after finish() every p.grad is the flat view again (steal mode rebinds it), holding the sum over ranks
        ok = ok and all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(sync.params, sync._views))
    # flag: NaN loss on one rank -> flag > 0 everywhere after finish()
    sync.zero()
    sync.finish(torch.tensor(float("inf")) if rank == 0 else torch.tensor(1.0))
    ok = ok and float(sync.flag) > 0
    q.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("steal", [False, True])
def test_bucket_overlap_hooks_two_ranks(steal):
    """steal=True: zero() unbinds p.grad, autograd keeps the backward nodes' tensors, each bucket is gathered (torch.cat) into the
    flat buffer right before it goes out - same reduced gradients, same launch order"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q, steal)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)


def _touched_worker(rank, world, port, q, steal=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sincformer_metacog_speech_enhancement_amd import dp
    torch.manual_seed(0)
    trunk, head_a, head_b = torch.nn.Linear(8, 8), torch.nn.Linear(8, 2), torch.nn.Linear(8, 2)
    params = list(trunk.parameters()) + list(head_a.parameters()) + list(head_b.parameters())
    sync = dp.FlatGradSynchronizer(params, bucket_bytes=128, overlap=True, steal=steal)
    x = torch.randn(4, 8)
    sync.zero()
    # data-dependent routing: rank 0's graph reaches head_a only, rank 1's head_b only; an extra head is reached by nobody
    h = trunk(x)
    loss = (head_a(h) if rank == 0 else head_b(h)).pow(2).mean()
    loss.backward()
    local_idle = len(sync.untouched())
    sync.finish(loss)
    mask = sync.touched_dev.clone()
    # trunk: both ranks (2), each head: one rank (1) -> every replica steps all six tensors; locally two were idle
    ok = local_idle == 2 and mask.tolist() == [2.0, 2.0, 1.0, 1.0, 1.0, 1.0]
    gathered = [torch.zeros_like(mask) for _ in range(world)]
    dist.all_gather(gathered, mask)
    ok = ok and all(torch.equal(g_, mask) for g_ in gathered)
    q.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("steal", [False, True])
def test_touched_mask_is_reduced_with_the_gradients(steal):
    """replicas whose backward passes reach different parameters must still step the same set: the per-rank 0 / 1 mask rides
    in the header of bucket 0 and comes back summed (optim.FlatAdamW steps a parameter when the sum is > 0)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_touched_worker, args=(r, 2, port, q, steal)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
