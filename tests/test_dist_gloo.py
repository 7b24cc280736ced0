"""world_size-2 CPU (gloo) coverage of the multi-GPU form of the path: ray sharding, the gradient all-reduce (big tensors
+ the flat small-gradient buffer), world-size-invariant masked means and the sphere-tracing trip-count reduction."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ls2fm import dist as ldist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert ldist.is_distributed()
        # ---- sharding: by view when the views divide evenly, otherwise contiguous ray ranges
        g = torch.Generator().manual_seed(0)
        center = torch.randn(4, 6, 3, generator=g)
        ray = torch.randn(4, 6, 3, generator=g)
        c, r = ldist.shard_rays(center, ray)
        assert c.shape == (2, 6, 3) and torch.equal(c, center[rank * 2:(rank + 1) * 2])
        c3, r3 = ldist.shard_rays(center[:3], ray[:3])            # 3 views over 2 ranks -> flattened ranges
        assert c3.shape[0] == 1 and c3.shape[1] == 9
        assert torch.equal(c3[0], center[:3].reshape(-1, 3)[rank * 9:(rank + 1) * 9])
        # every ray is owned by exactly one rank
        owned = torch.zeros(18)
        owned[rank * 9:(rank + 1) * 9] = 1
        dist.all_reduce(owned)
        assert torch.equal(owned, torch.ones(18))

        # ---- gradient all-reduce: one big tensor + several small ones
        big = torch.nn.Parameter(torch.zeros(1 << 20))
        smalls = [torch.nn.Parameter(torch.zeros(64, 35)), torch.nn.Parameter(torch.zeros(64, 1)),
                  torch.nn.Parameter(torch.zeros(1))]
        big.grad = torch.full_like(big, float(rank + 1))
        for k, p in enumerate(smalls):
            p.grad = torch.full_like(p, float((rank + 1) * (k + 2)))
        red = ldist.GradAllReducer([big, *smalls])
        red.all_reduce()
        assert torch.equal(big.grad, torch.full_like(big, 3.0))
        for k, p in enumerate(smalls):
            assert torch.equal(p.grad, torch.full_like(p, 3.0 * (k + 2)))
        # the fused backward's flat small-gradient buffer is reduced in place, without packing
        from ls2fm import fused
        a, b = torch.nn.Parameter(torch.zeros(2, 3)), torch.nn.Parameter(torch.zeros(5))
        flat, (ga, gb) = fused.flat_gradient_views([a, b])            # what the fused backward returns its gradients in
        assert flat.numel() == 8 + 8 and gb.data_ptr() - flat.data_ptr() == 32          # 16-byte segments
        flat.copy_(torch.arange(16, dtype=torch.float32) * (rank + 1))
        a.grad, b.grad = ga, gb
        red2 = ldist.GradAllReducer([a, b])
        assert red2._all_in_flat() is flat
        red2.all_reduce()
        assert torch.equal(flat, torch.arange(16, dtype=torch.float32) * 3)
        assert a.grad.data_ptr() == flat.data_ptr()

        # ---- masked mean is world-size invariant (sum / count all-reduced separately)
        vals = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])
        mask = torch.tensor([1, 1, 1, 1, 0, 1], dtype=torch.bool)           # unbalanced between the two halves
        lo, hi = rank * 3, rank * 3 + 3
        m = ldist.global_mean(vals[lo:hi][mask[lo:hi]].sum(), mask[lo:hi].sum())
        assert abs(m.item() - vals[mask].mean().item()) < 1e-6
        assert ldist.global_max_int(3 + 4 * rank, "cpu") == 7
        assert ldist.global_any(torch.tensor(rank == 1)) is True and ldist.global_any(False) is False
        # (sum, count) pairs of the loss head -> global counts: the two documented modes
        sums = torch.tensor([1.0, 3, 2, 10, 0.5, float(rank), 4, 3], dtype=torch.float64) * (rank + 1)
        a = sums.clone(); ldist.globalize_loss_sums(a, "allreduce")
        assert torch.equal(a, torch.tensor([3.0, 9, 6, 30, 1.5, 2, 12, 9], dtype=torch.float64))
        b = sums.clone(); ldist.globalize_loss_sums(b, "uniform")
        assert torch.equal(b[1::2], sums[1::2] * 2) and torch.equal(b[0::2], sums[0::2])
        # a level group's two table slices go out as one coalesced launch where the backend can (else one each)
        buf = torch.arange(24, dtype=torch.float32) * (rank + 1)
        works = ldist._all_reduce_together([buf[2:7], buf[13:20]])
        for wk in works:
            wk.wait()
        want = torch.arange(24, dtype=torch.float32) * (rank + 1)
        want[2:7] = torch.arange(2, 7, dtype=torch.float32) * 3
        want[13:20] = torch.arange(13, 20, dtype=torch.float32) * 3
        assert torch.equal(buf, want), (buf, want)
        with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def _sharded_worker(rank, world, port, out_dir):
    """reduce-scatter -> Adam on this rank's shard -> all-gather == one process running Adam on the summed gradients"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ls2fm import fused
        g = torch.Generator().manual_seed(0)
        shapes = [(1000,), (64, 35), (64, 1), (64,), (17, 64), (1,), (1001,), (3, 64)]     # odd sizes: segments get padded
        init = [torch.randn(sh, generator=g) for sh in shapes]
        lrs = [1e-2 if k < 6 else 3e-3 for k in range(len(shapes))]                        # two rates, as lr_sdf / lr_color
        params = [torch.nn.Parameter(t.clone()) for t in init]
        ref = [torch.nn.Parameter(t.clone()) for t in init]
        ref_opt = torch.optim.Adam([dict(params=[ref[k] for k in range(6)], lr=1e-2), dict(params=[ref[k] for k in range(6, 8)], lr=3e-3)])
        opt = ldist.ShardedAdam(params, group_lrs=lrs, update=lambda groups: torch.optim.Adam(groups))
        assert opt.total % (4 * world) == 0 and opt.shard * world == opt.total
        for p, t in zip(params, init):                      # same values, storage now inside the flat buffer
            assert torch.equal(p.detach(), t) and p.data_ptr() >= opt.flat.data_ptr()
        for it in range(4):
            per_rank = [[torch.randn(sh, generator=g) for sh in shapes] for _ in range(world)]
            if it % 2 == 0:      # the fused backward's way: one flat buffer at this optimizer's offsets (padded to its length)
                flat, views = fused.flat_gradient_views(params)
                assert flat.numel() == opt.total
                for v, t in zip(views, per_rank[rank]):
                    v.copy_(t)
                for p, v in zip(params, views):
                    p.grad = v
                assert opt._flat_gradient() is flat
            else:                # loose gradients (composed form): packed
                for p, t in zip(params, per_rank[rank]):
                    p.grad = t.clone()
                    p._ls2fm_grad_flat = None
            opt.step()
            for p, *gs in zip(ref, *per_rank):
                p.grad = sum(gs)
            ref_opt.step()
            for k, (p, q) in enumerate(zip(params, ref)):
                assert torch.allclose(p.detach(), q.detach(), rtol=0, atol=1e-6), (it, k, float((p - q).abs().max()))
        # every rank holds only its shard of the optimizer state
        n_state = sum(st["exp_avg"].numel() for st in opt.inner.state.values())
        assert n_state <= opt.shard
        with open(os.path.join(out_dir, f"shard_ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def _pipelined_worker(rank, world, port, out_dir):
    """ShardedAdam(n_groups >= 2): per level group reduce-scatter -> Adam on this rank's slice -> all-gather, small tensors and
    slice remainders replicated == one process running Adam on the summed gradients; state_dict round trip; close()"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ls2fm import fused
        g = torch.Generator().manual_seed(0)
        # two "hash tables" of 5 levels (entries: 8, 24, 40, 64, 64 -> 2 floats each) between small tensors of odd sizes
        level_offsets = [0, 8, 32, 72, 136, 200]
        shapes = [(400,), (64, 35), (64, 1), (17,), (1,), (400,), (3, 64), (5,)]
        tables_at = (0, 5)
        init = [torch.randn(sh, generator=g) for sh in shapes]
        lrs = [1e-2 if k < 5 else 3e-3 for k in range(len(shapes))]
        for n_groups in (2, 3):
            params = [torch.nn.Parameter(t.clone()) for t in init]
            ref = [torch.nn.Parameter(t.clone()) for t in init]
            ref_opt = torch.optim.Adam([dict(params=ref[:5], lr=1e-2), dict(params=ref[5:], lr=3e-3)])
            opt = ldist.ShardedAdam(params, group_lrs=lrs, update=lambda groups: torch.optim.Adam(groups), n_groups=n_groups,
                                    tables=[params[k] for k in tables_at], level_offsets=level_offsets)
            assert opt.n_groups == n_groups and opt._pipe is not None
            sharded = sum(pc["n"] for grp in opt._pipe["pieces"] for pc in grp)
            assert sharded >= 2 * 400 - 2 * n_groups * 4 * world            # all of both tables but the slices' remainders
            assert getattr(params[0], "_ls2fm_overlap_groups") == n_groups and opt.in_backward
            for it in range(4):
                per_rank = [[torch.randn(sh, generator=g) for sh in shapes] for _ in range(world)]
                if it % 2 == 0:
                    flat, views = fused.flat_gradient_views(params)
                    for v, t in zip(views, per_rank[rank]):
                        v.copy_(t)
                    for p_, v in zip(params, views):
                        p_.grad = v
                else:
                    for p_, t in zip(params, per_rank[rank]):
                        p_.grad = t.clone()
                        p_._ls2fm_grad_flat = None
                opt.step()
                for p_, *gs in zip(ref, *per_rank):
                    p_.grad = sum(gs)
                ref_opt.step()
                for k, (p_, q) in enumerate(zip(params, ref)):
                    assert torch.allclose(p_.detach(), q.detach(), rtol=0, atol=1e-6), (n_groups, it, k, float((p_ - q).abs().max()))
            # per-rank state: a fresh optimizer over the same values continues identically after load_state_dict
            import copy
            state = copy.deepcopy(opt.state_dict())                  # (as torch.save / torch.load would hand it over)
            params2 = [torch.nn.Parameter(p_.detach().clone()) for p_ in params]
            opt2 = ldist.ShardedAdam(params2, group_lrs=lrs, update=lambda groups: torch.optim.Adam(groups), n_groups=n_groups,
                                     tables=[params2[k] for k in tables_at], level_offsets=level_offsets)
            opt2.load_state_dict(state)
            per_rank = [[torch.randn(sh, generator=g) for sh in shapes] for _ in range(world)]
            for ps_, o in ((params, opt), (params2, opt2)):
                for p_, t in zip(ps_, per_rank[rank]):
                    p_.grad = t.clone()
                    p_._ls2fm_grad_flat = None
                o.step()
            for k, (p_, q) in enumerate(zip(params, params2)):
                assert torch.equal(p_.detach(), q.detach()), (n_groups, k)
            opt.close(); opt2.close()
            assert not hasattr(params[0], "_ls2fm_group_exchange") and not hasattr(params[0], "_ls2fm_flat_total")
        with open(os.path.join(out_dir, f"pipe_ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def test_pipelined_sharded_adam_two_rank_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_pipelined_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "pipe_ok0").exists() and (tmp_path / "pipe_ok1").exists()


def test_sharded_adam_two_rank_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_sharded_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "shard_ok0").exists() and (tmp_path / "shard_ok1").exists()


def _pending_worker(rank, world, port, out_dir):
    """group reductions launched from inside a backward are always waited for, and refused when the step had a second producer"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ls2fm import fused
        a, b = torch.nn.Parameter(torch.zeros(8)), torch.nn.Parameter(torch.zeros(4))
        flat, (ga, gb) = fused.flat_gradient_views([a, b])
        flat.fill_(float(rank + 1))
        h = dist.all_reduce(flat, async_op=True)              # what launch_group_reductions leaves behind
        flat._ls2fm_pending = [h]
        ldist._PENDING[id(flat)] = (flat, [h], [a, b])
        a.grad, b.grad = ga, gb
        ldist.GradAllReducer([a, b]).all_reduce()             # sole producer: waits, nothing reduced twice
        assert torch.equal(flat, torch.full_like(flat, 3.0)) and not ldist._PENDING
        # a second producer replaced a gradient: the launched reduction is still waited for, then the step is refused
        flat2, (ga2, gb2) = fused.flat_gradient_views([a, b])
        flat2.fill_(1.0)
        h2 = dist.all_reduce(flat2, async_op=True)
        ldist._PENDING[id(flat2)] = (flat2, [h2], [a, b])
        a.grad, b.grad = ga2 + 1.0, gb2                       # autograd summed another node's gradient into a NEW tensor
        try:
            ldist.GradAllReducer([a, b]).all_reduce()
            raised = False
        except RuntimeError as e:
            raised = "only gradient producer" in str(e)
        assert raised and not ldist._PENDING and h2.is_completed()
        # two field pairs with a reducer each (round-3 advisor): a reducer takes only the reductions of ITS parameters
        c, d = torch.nn.Parameter(torch.zeros(8)), torch.nn.Parameter(torch.zeros(4))
        flat_a, (ga3, gb3) = fused.flat_gradient_views([a, b])
        flat_c, (gc3, gd3) = fused.flat_gradient_views([c, d])
        flat_a.fill_(float(rank + 1)); flat_c.fill_(10.0 * (rank + 1))
        a.grad, b.grad, c.grad, d.grad = ga3, gb3, gc3, gd3
        ha, hc = dist.all_reduce(flat_a, async_op=True), dist.all_reduce(flat_c, async_op=True)
        ldist._PENDING[id(flat_a)] = (flat_a, [ha], [a, b])
        ldist._PENDING[id(flat_c)] = (flat_c, [hc], [c, d])
        ldist.GradAllReducer([a, b]).all_reduce()             # must not drain (or trip over) the other pair's entry
        assert list(ldist._PENDING) == [id(flat_c)] and torch.equal(flat_a, torch.full_like(flat_a, 3.0))
        ldist.GradAllReducer([c, d]).all_reduce()
        assert not ldist._PENDING and torch.equal(flat_c, torch.full_like(flat_c, 30.0))
        # a backward never followed by all_reduce(): its entry is retired when the same parameters launch again
        flat_o, _ = fused.flat_gradient_views([a, b])
        flat_o.fill_(1.0)
        ho = dist.all_reduce(flat_o, async_op=True)
        ldist._PENDING[id(flat_o)] = (flat_o, [ho], [a, b])
        ldist._retire_superseded([a, b])
        assert not ldist._PENDING and ho.is_completed()
        with open(os.path.join(out_dir, f"pend_ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def test_pending_group_reductions_are_always_waited(tmp_path):
    port = _free_port()
    mp.spawn(_pending_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "pend_ok0").exists() and (tmp_path / "pend_ok1").exists()


def test_two_rank_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_single_process_paths_are_noops():
    c, r = torch.zeros(2, 4, 3), torch.ones(2, 4, 3)
    assert ldist.shard_rays(c, r, rank=0, world=1)[0] is c
    assert not ldist.is_distributed()
    assert ldist.global_max_int(5, "cpu") == 5
    assert ldist.global_mean(torch.tensor(6.0), torch.tensor(3)).item() == 2.0
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.ones(3)
    ldist.GradAllReducer([p]).all_reduce()
    assert torch.equal(p.grad, torch.ones(3))


def _picks_worker(rank, world, port, out_dir):
    """under distributed=True every rank must evaluate the replicated terms on the same pixels and view;
    the default picks came from per-rank generators.  Shared picks: rank 0's seed, broadcast once; the view from the same stream."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ls2fm.stage import _HostPicks
        torch.manual_seed(1000 + 17 * rank)                      # the ranks' own seeds differ
        import random
        random.seed(5 + rank)
        shared = _HostPicks(4096, 16, "cpu", tail=1, shared=True)
        local = _HostPicks(4096, 16, "cpu", tail=1, shared=False)
        again = _HostPicks(4096, 16, "cpu", tail=1, shared=True)
        rows = []
        for _ in range(3):
            shared.draw(shared.view(7))
            rows.append(shared.dev.clone())
        local.draw(local.view(7))
        again.draw(again.view(7))
        mine = torch.stack(rows)
        both = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        assert torch.equal(both[0], both[1])                     # same pixels, same view, every iteration
        assert len(set(mine[0, :16].tolist())) == 16 and int(mine[:, 16].max()) < 7
        assert not torch.equal(mine[0], mine[1])                 # a stream, not a constant
        assert not torch.equal(again.dev, mine[0])               # a second loop instance does not replay the first one's sequence
        loc = [torch.zeros_like(local.dev) for _ in range(world)]
        dist.all_gather(loc, local.dev)
        assert not torch.equal(loc[0][:16], loc[1][:16])         # (unshared picks stay per-rank)
        with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def test_shared_default_picks_two_rank_gloo(tmp_path):
    mp.spawn(_picks_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
