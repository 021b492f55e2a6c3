"""CPU tests of the multi-GPU layer with the gloo backend, world_size 2 (one process per rank, rendezvous on 127.0.0.1)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _fake_render(H, W, focal, mc, mf, batch, options, scene_id, mode="validation", scene_config=None):
    """stand-in for run_one_iter_of_nerf (no GPU here): a per-ray function, so sharding must not change any pixel"""
    ro, rd = batch[0], batch[1]
    c = torch.sin(ro * 3.0 + rd)
    f = torch.cos(rd * 2.0) * ro.sum(-1, keepdim=True)
    return c, None, None, f, None, None, None, None, None


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import nvsr_amd
    D = nvsr_amd.distributed
    torch.manual_seed(0)                       # same data on every rank (scene and view are replicated)
    H, W = 7, 5                                # 35 rays: not divisible by 2
    ro, rd = torch.randn(H, W, 3), torch.randn(H, W, 3)
    full = _fake_render(H, W, 1.0, None, None, torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0), None, None)
    c, f = D.render_image_sharded(H, W, 1.0, None, None, ro, rd, None, "s", None, render_fn=_fake_render)
    assert torch.equal(c.reshape(-1, 3), full[0]) and torch.equal(f.reshape(-1, 3), full[3])
    lc, lf, (lo, hi) = D.render_image_sharded(H, W, 1.0, None, None, ro, rd, None, "s", None, render_fn=_fake_render, gather=False)
    assert (lo, hi) == D.shard_bounds(H * W, rank, world) and torch.equal(lc, full[0][lo:hi])
    # several views, each sharded by ROW blocks, rendered in one launch per rank and assembled with one all_gather (the bench's default
    # partition); H = 7 rows over 2 ranks is uneven
    poses = [torch.randn(3, 3) for _ in range(3)]
    ray_fn = lambda h, w, focal, pose: (ro @ pose, rd @ pose.T)
    vc, vf = D.render_views_sharded(H, W, 1.0, None, None, poses, None, "s", None, render_fn=_fake_render, ray_fn=ray_fn)
    assert vc.shape == (3, H, W, 3)
    for v, pose in enumerate(poses):
        o, d = ray_fn(H, W, 1.0, pose)
        ref = _fake_render(H, W, 1.0, None, None, torch.stack([o.reshape(-1, 3), d.reshape(-1, 3)], 0), None, None)
        assert torch.equal(vc[v].reshape(-1, 3), ref[0]) and torch.equal(vf[v].reshape(-1, 3), ref[3])
    bc, bf, (rlo, rhi) = D.render_views_sharded(H, W, 1.0, None, None, poses, None, "s", None, render_fn=_fake_render, ray_fn=ray_fn, gather=False)
    assert (rlo, rhi) == D.shard_bounds(H, rank, world) and torch.equal(bc, vc[:, rlo:rhi])
    # gradient all-reduce: rank r holds r+1 everywhere -> average 1.5
    grads = [torch.full((1000,), float(rank + 1)), torch.full((3, 5), float(rank + 1)), torch.full((70000,), float(rank + 1))]
    D.allreduce_gradients(grads, bucket_bytes=1 << 16)
    assert all(torch.allclose(g, torch.full_like(g, 1.5)) for g in grads)
    # the RCCL branch of allreduce_gradients (allreduce_in_place: one in-place asynchronous all-reduce per dense tensor, queued, then waited
    # for, then ONE fused scale) on this backend: row-major and channels_last tensors -- the plane gradients are channels_last -- with values
    # that tell a transposed reduction from a correct one; a non-dense tensor is refused
    base = torch.arange(2 * 6 * 4 * 5, dtype=torch.float32).reshape(2, 6, 4, 5)
    cl = (base * (rank + 1)).contiguous(memory_format=torch.channels_last)
    rm = torch.arange(37, dtype=torch.float32) * (rank + 1)
    plane = (base[:1] * (rank + 2)).contiguous(memory_format=torch.channels_last)
    assert not cl.is_contiguous() and cl.is_contiguous(memory_format=torch.channels_last)
    ptrs = [t.data_ptr() for t in (cl, rm, plane)]
    D.allreduce_in_place([cl, rm, plane], scale=0.5)
    assert torch.equal(cl, base * 1.5) and torch.equal(rm, torch.arange(37, dtype=torch.float32) * 1.5) and torch.equal(plane, base[:1] * 2.5)
    assert [t.data_ptr() for t in (cl, rm, plane)] == ptrs and cl.is_contiguous(memory_format=torch.channels_last)     # in place, layout kept
    D.allreduce_in_place([])
    try:
        D.allreduce_in_place([base[:, ::2]])
        raise AssertionError("a strided tensor was accepted")
    except ValueError:
        pass
    # the bucketed RCCL branch (allreduce_coalesced: an SR network's 69 small weight gradients share flat buckets) on this backend: several
    # buckets, a channels_last member, values that tell a mis-sliced bucket from a correct one; layouts and storage kept
    many = [torch.arange(n, dtype=torch.float32) * (rank + 1) + k for k, n in enumerate([300, 7, 1024, 513, 2, 4096, 33])]
    many.insert(3, (base * (rank + 1)).contiguous(memory_format=torch.channels_last))
    ptrs = [t.data_ptr() for t in many]
    D.allreduce_coalesced(many, scale=0.5, bucket_bytes=4096)
    for k, n in enumerate([300, 7, 1024, 513, 2, 4096, 33]):
        assert torch.equal(many[k if k < 3 else k + 1], torch.arange(n, dtype=torch.float32) * 1.5 + k)
    assert torch.equal(many[3], base * 1.5) and many[3].is_contiguous(memory_format=torch.channels_last)
    assert [t.data_ptr() for t in many] == ptrs
    D.allreduce_coalesced([])
    # gather_row_blocks into a caller's buffer: ragged split (3 + 4 rows) and the one-rank case
    mine = torch.full((3 + rank, 2), float(rank + 1))
    out = torch.empty(7, 2)
    got = D.gather_row_blocks(mine, [3, 4], out=out)
    assert got is out and torch.equal(out[:3], torch.full((3, 2), 1.0)) and torch.equal(out[3:], torch.full((4, 2), 2.0))
    # band-sharded SR stage: a stand-in SR function whose value depends on the HR position only (like the real net, whose output
    # does not depend on the ROI), NaN outside the ROI it was asked for; R0 = 7 rows is not divisible by 2
    sf, R0, R1 = 4, 7, 6
    truth = torch.arange(3 * R0 * sf * R1 * sf, dtype=torch.float32).reshape(1, 3, R0 * sf, R1 * sf)

    class FakeSR:
        scale_factor = sf
        LR_planes = {"a": torch.zeros(1, 3, R0, R1), "b": torch.zeros(1, 3, R0, R1)}
        SR_planes = {}

    def fake_sr(name, roi):
        import math
        lo = max(0, math.floor(R0 * (1 + float(roi[0, 0])) / 2) - 1)
        hi = min(R0, math.ceil(R0 * (1 + float(roi[1, 0])) / 2) + 1)
        out = torch.full_like(truth, float("nan"))
        out[:, :, lo * sf: hi * sf] = truth[:, :, lo * sf: hi * sf] + (1000.0 if name == "b" else 0.0)
        return out

    got = D.super_resolve_planes_sharded(FakeSR, ["a", "b"], sr_fn=fake_sr)
    assert torch.equal(got[0], truth) and torch.equal(got[1], truth + 1000.0) and sorted(FakeSR.SR_planes) == ["a", "b"]
    lo, hi = D.shard_bounds(R0, rank, world)
    roi = D.band_roi(lo, hi, R0)
    import math
    assert math.floor(R0 * (1 + float(roi[0, 0])) / 2) == lo and math.ceil(R0 * (1 + float(roi[1, 0])) / 2) == hi
    # the tail of a data-parallel training iteration (training.TrainStep.apply_gradients: backward -> grad_sync -> optimizer steps) with the
    # three optimizers of a joint SR-refinement step: every rank has its OWN rays (here: its own loss), the gradients are averaged by
    # distributed.allreduce_gradients, and both ranks must end every iteration with bit-identical parameters that equal the one-process step on
    # the mean of the two losses
    def make_params():
        gen = torch.Generator().manual_seed(7)
        return [torch.nn.Parameter(torch.randn(sh, generator=gen)) for sh in ((1, 4, 6, 6), (16, 8), (8, 4, 3, 3))]

    def loss_of(params, r, it):
        gen = torch.Generator().manual_seed(100 * it + r)
        return sum(((p * torch.randn(p.shape, generator=gen)).sum() ** 2 + (p ** 2).sum() * 0.1) for p in params)

    def make_step(params, sync):
        planes, decp, srp = params
        return nvsr_amd.training.TrainStep(None, None, None, {"LR_planes", "decoder", "SR"}, optimizer=torch.optim.Adam([decp], lr=1e-2),
                                           SR_optimizer=torch.optim.Adam([srp], lr=1e-2), planes_optimizer=torch.optim.Adam([planes], lr=1e-2),
                                           grad_sync=sync)
    mine = make_params()
    step = make_step(mine, lambda: D.allreduce_gradients([p.grad for p in mine]))
    alone = make_params()
    ref_step = make_step(alone, None)
    for it in range(3):
        for p in mine + alone:
            p.grad = None
        step.apply_gradients(loss_of(mine, rank, it), sr_iter=True)
        ref_step.apply_gradients(sum(loss_of(alone, r, it) for r in range(world)) / world, sr_iter=True)
        for p, q in zip(mine, alone):
            both = [torch.empty_like(p.data) for _ in range(world)]
            dist.all_gather(both, p.data.contiguous())
            assert torch.equal(both[0], both[1]), "the ranks' parameters drifted apart in iteration %d" % it
            assert torch.allclose(p.data, q.data, rtol=1e-5, atol=1e-6), float((p.data - q.data).abs().max())
    # the SR network's gradient all-reduced bucket by bucket INSIDE the backward (distributed.OverlappedSRGradSync: what
    # ops.PlanesSRBatchFn.backward does with the blob nvsr_planes_sr_backward_batch_marks fills) against the post-backward path above: a stand-in
    # autograd node with the real node's protocol (one weight-gradient blob in state-dict order, plan() -> buckets that are suffixes of the blob,
    # reduce_marked() before the blob goes back to autograd).  Parameters bit-identical to the post-backward path on both ranks, every iteration.
    class FakeSR(torch.nn.Module):
        def __init__(self):
            super().__init__()
            gen = torch.Generator().manual_seed(11)
            self.ws = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(sh, generator=gen) * 0.1)
                                              for sh in ((8, 4, 3, 3), (8, 8, 3, 3), (8, 8, 3, 3), (8, 8, 3, 3), (32, 8, 3, 3), (4, 8, 3, 3))])

    class FakeBatchFn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, sync, sizes, coef, blob):
            ctx.sync, ctx.sizes = sync, sizes
            ctx.save_for_backward(coef, blob)
            return ((blob * coef) ** 2).sum()

        @staticmethod
        def backward(ctx, d):
            coef, blob = ctx.saved_tensors
            g = 2.0 * blob * coef * coef * d
            if ctx.sync is not None and ctx.sync.active():
                marks = ctx.sync.plan(ctx.sizes, g)
                assert marks[0][2] == g.numel() and marks[-1][1] == 0 and all(a[1] == b[2] for a, b in zip(marks, marks[1:]))      # suffixes, last first
                assert [m[0] for m in marks] == sorted((m[0] for m in marks), reverse=True)
                ctx.sync.reduce_marked(g, marks)
            return None, None, None, g

    def sr_loss(net, sync, r, it):
        gen = torch.Generator().manual_seed(1000 * it + r)
        blob = torch.cat([w.reshape(-1) for w in net.ws])
        return FakeBatchFn.apply(sync, [w.numel() for w in net.ws], torch.randn(blob.shape, generator=gen), blob)

    def run(overlapped):
        torch.manual_seed(5)
        net, others = FakeSR(), make_params()[:2]
        if overlapped:
            sync = D.OverlappedSRGradSync(net, other_parameters=others, bucket_bytes=4096)
            assert net.__dict__["grad_bucket_sync"] is sync
        else:
            sync = lambda: D.allreduce_gradients([p.grad for p in others + list(net.parameters()) if p.grad is not None])
        st = nvsr_amd.training.TrainStep(None, None, None, {"LR_planes", "decoder", "SR"}, optimizer=torch.optim.Adam([others[1]], lr=1e-2),
                                         SR_optimizer=torch.optim.Adam(net.parameters(), lr=1e-2), planes_optimizer=torch.optim.Adam([others[0]], lr=1e-2),
                                         SR_model=None, grad_sync=sync)
        hist = []
        for it in range(3):
            for p in others + list(net.parameters()):
                p.grad = None
            loss = sr_loss(net, sync if overlapped else None, rank, it) + loss_of(others, rank, it)
            st.apply_gradients(loss, sr_iter=True)
            if overlapped:
                assert sync.stats["buckets"] >= 3 and not sync.reduced_in_backward          # several buckets; the flag was consumed by sync()
            hist.append([p.detach().clone() for p in others + list(net.parameters())])
        return hist
    a, b = run(True), run(False)
    for it in range(3):
        for p, q in zip(a[it], b[it]):
            assert torch.equal(p, q), "overlapped and post-backward gradient averaging differ in iteration %d" % it
            both = [torch.empty_like(p) for _ in range(world)]
            dist.all_gather(both, p.contiguous())
            assert torch.equal(both[0], both[1])
    # an iteration whose SR backward did NOT hand its blob over (one plane / exact f32: PlanesSR.forward): sync() averages the SR gradients itself
    net = FakeSR()
    sync = D.OverlappedSRGradSync(net, other_parameters=[])
    for w in net.ws:
        w.grad = torch.full_like(w, float(rank + 1))
    sync()
    assert all(torch.equal(w.grad, torch.full_like(w, 1.5)) for w in net.ws)
    sync.detach()
    assert "grad_bucket_sync" not in net.__dict__
    # buckets of allreduce_coalesced: one dtype each, never beyond the bound
    mixed = [torch.full((300,), float(rank + 1)), torch.full((300,), float(rank + 1), dtype=torch.float64), torch.full((700,), float(rank + 1))]
    D.allreduce_coalesced(mixed, scale=0.5, bucket_bytes=2048)
    assert all(torch.equal(t, torch.full_like(t, 1.5)) for t in mixed) and mixed[1].dtype == torch.float64
    try:
        nvsr_amd.training.GraphedTrainStep(step, torch.zeros(4, 4, 3), torch.eye(4), 4, 4, 1.0, 1, "s", None, 8)
        raise AssertionError("a data-parallel step was accepted for graph capture")
    except ValueError as e:
        assert "grad_sync" in str(e)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, "ok%d" % rank), "w").write("ok")


def test_gather_row_blocks_fills_the_callers_buffer_on_one_rank():
    sys.path.insert(0, ROOT)
    import nvsr_amd
    local, out = torch.arange(6.0).reshape(3, 2), torch.zeros(3, 2)
    got = nvsr_amd.distributed.gather_row_blocks(local, [3], out=out)
    assert got is out and torch.equal(out, local)


def test_shard_bounds_cover_everything():
    sys.path.insert(0, ROOT)
    import nvsr_amd
    sb = nvsr_amd.distributed.shard_bounds
    for n in (0, 1, 7, 640000, 640001):
        for world in (1, 2, 3, 8):
            spans = [sb(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_sharded_render_and_grad_allreduce(tmp_path):
    world, port = 2, 29000 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / ("ok%d" % r)).exists() for r in range(world))
