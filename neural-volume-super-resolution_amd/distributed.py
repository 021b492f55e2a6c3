"""Multi-GPU layer of the path: one process per GPU, rays sharded, scene replicated.

The reference is single-process.  Rays are independent units, so inference shards contiguous blocks of rays (image rows) over
the ranks with NO data-path collective; the only exchange is an optional all_gather of the finished pixels (7.7 MB for an
800x800 frame).  Training adds one all-reduce (RCCL over xGMI; backend "nccl" on ROCm) of the plane / decoder gradients, bucketed
so that a ring step moves a few MB per link."""
import torch
import torch.distributed as dist


def world_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_bounds(n, rank, world):
    """[lo, hi) of rank's contiguous block when n units are split as evenly as possible (first n % world ranks get one more)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_rays(batch_rays, rank=None, world=None, group=None):
    """batch_rays [2,N,3] (run_one_iter_of_nerf's input) -> this rank's [2,n_local,3] block and its (lo, hi)."""
    if rank is None:
        rank, world = world_info(group)
    lo, hi = shard_bounds(batch_rays.shape[1], rank, world)
    return batch_rays[:, lo:hi], (lo, hi)


def _collective_device(t, group=None):
    """the device a collective of this backend runs on: RCCL ("nccl") moves device buffers over xGMI; gloo (CPU rehearsals of the N > 1
    path, also with the ranks' tensors on a GPU) implements all_gather for host buffers only -> stage through the host"""
    return t.device if dist.get_backend(group) == "nccl" else torch.device("cpu")


def gather_row_blocks(local, counts, group=None, out=None):
    """all_gather of per-rank row blocks (counts[r] rows on rank r) -> [sum(counts), ...] on every rank.
    Even split: ONE all_gather_into_tensor straight into the assembled frame (`out`, preallocated by the caller or here) -- no padding, no
    list of parts, no cat.  Ragged split: blocks padded to the longest one, gathered the same way, the padding rows dropped by one
    index_select.  A backend that only moves host buffers (gloo rehearsals) stages through the host."""
    rank, world = world_info(group)
    if world == 1:
        if out is not None:
            out.copy_(local)
            return out
        return local
    assert local.shape[0] == counts[rank]
    cdev = _collective_device(local, group)
    tail = tuple(local.shape[1:])
    even = all(c == counts[0] for c in counts)
    n_max = max(counts)
    src = local.contiguous()
    if not even:
        src = torch.zeros((n_max,) + tail, dtype=local.dtype, device=local.device)
        src[: local.shape[0]] = local
    if even and out is not None and out.device == cdev and out.is_contiguous():
        buf = out
    else:
        buf = torch.empty((world * n_max,) + tail, dtype=local.dtype, device=cdev)
    dist.all_gather_into_tensor(buf, src.to(cdev), group=group)
    if not even:
        keep = torch.cat([torch.arange(r * n_max, r * n_max + counts[r]) for r in range(world)]).to(buf.device)
        buf = buf.index_select(0, keep)
    if out is not None and buf is not out:
        out.copy_(buf)
        return out
    return buf.to(local.device)


def gather_rows(local, n_total, group=None):
    """gather_row_blocks for the even split of shard_bounds(n_total)"""
    rank, world = world_info(group)
    counts = [shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0] for r in range(world)]
    return gather_row_blocks(local, counts, group)


def render_image_sharded(height, width, focal, model_coarse, model_fine, ray_origins, ray_directions, options, scene_id, scene_config,
                         group=None, gather=True, render_fn=None):
    """eval_nerf with the rays of one view sharded over the ranks (contiguous row blocks).  Returns (rgb_coarse, rgb_fine) as
    [H,W,3] on every rank when gather=True, else this rank's [n_local,3] blocks and its (lo, hi)."""
    grid = {}
    if render_fn is None:
        from .train_utils import run_one_iter_of_nerf as render_fn
        grid = dict(ray_grid_width=width)      # (used when the rank's block is whole rows: the renderer then works in 8 x 4 pixel patches)
    rank, world = world_info(group)
    batch = torch.stack([ray_origins.reshape(-1, 3), ray_directions.reshape(-1, 3)], 0)
    local, (lo, hi) = shard_rays(batch, rank, world)
    out = render_fn(height, width, focal, model_coarse, model_fine, local, options, scene_id, mode="validation", scene_config=scene_config, **grid)
    rgb_c, rgb_f = out[0], out[3]
    if not gather:
        return rgb_c, rgb_f, (lo, hi)
    n = height * width
    rgb_c = gather_rows(rgb_c, n, group).reshape(height, width, 3)
    rgb_f = None if rgb_f is None else gather_rows(rgb_f, n, group).reshape(height, width, 3)
    return rgb_c, rgb_f


def render_views_sharded(height, width, focal, model_coarse, model_fine, poses, options, scene_id, scene_config, group=None, gather=True,
                         render_fn=None, ray_fn=None):
    """Several views of one scene, EVERY view sharded over the ranks by contiguous blocks of image rows -- rank r renders
    rows shard_bounds(height, r, world) of each view, all of them in ONE launch of the fused passes (rays are independent units, so the
    rank's blocks of the V views are simply concatenated: with V = world views per step a rank renders one frame's worth of rays per
    step, whatever the world size).  One all_gather assembles the frames.  Returns (rgb_coarse, rgb_fine) as [V,H,W,3] on every rank
    (gather=True) or this rank's [V, rows, W, 3] blocks and its (lo, hi)."""
    grid = {}
    if render_fn is None:
        from .train_utils import run_one_iter_of_nerf as render_fn
        grid = dict(ray_grid_width=width)
    if ray_fn is None:
        from .nerf_helpers import get_ray_bundle as ray_fn
    rank, world = world_info(group)
    lo, hi = shard_bounds(height, rank, world)
    V, rows = len(poses), hi - lo
    blocks = []
    for pose in poses:
        ro, rd = ray_fn(height, width, focal, pose)
        blocks.append(torch.stack([ro[lo:hi].reshape(-1, 3), rd[lo:hi].reshape(-1, 3)], 0))
    out = render_fn(height, width, focal, model_coarse, model_fine, torch.cat(blocks, 1), options, scene_id, mode="validation",
                    scene_config=scene_config, **grid)
    local = [None if t is None else t.reshape(V, rows, width, 3) for t in (out[0], out[3])]
    if not gather:
        return local[0], local[1], (lo, hi)
    counts = [shard_bounds(height, r, world)[1] - shard_bounds(height, r, world)[0] for r in range(world)]
    full = []
    for t in local:                    # rows first for gather_row_blocks, then back to [V,H,W,3]
        full.append(None if t is None else gather_row_blocks(t.permute(1, 0, 2, 3).contiguous(), counts, group).permute(1, 0, 2, 3).contiguous())
    return full[0], full[1]


def band_roi(lo, hi, rows):
    """Region of interest (PlanesSR's [[ymin, xmin], [ymax, xmax]] in [-1, 1]) whose pixel arithmetic (floor / ceil, then one pixel
    of margin, models.py:902-905) covers LR rows [lo, hi) and all columns."""
    return torch.tensor([[2.0 * (lo + 0.5) / rows - 1.0, -1.0], [2.0 * (hi - 0.5) / rows - 1.0, 1.0]], dtype=torch.float32)


def super_resolve_planes_sharded(sr_model, plane_names, group=None, sr_fn=None):
    """SR stage of a scene over the ranks: every rank super-resolves one horizontal band of every plane -- PlanesSR's own
    ROI path supplies the 68-pixel LR halo the network's receptive field needs -- and ONE all_gather per plane assembles the HR planes
    on all ranks (123 MB per 800^2 plane).  The result lands in `sr_model.SR_planes`, where the renderer finds it.
    sr_fn(plane_name, roi) -> [1,C,sf*R0,sf*R1] (NaN outside the ROI); default: the model itself."""
    rank, world = world_info(group)
    if sr_fn is None:
        def sr_fn(name, roi):
            with torch.no_grad():
                return sr_model((name, roi))
    sf = sr_model.scale_factor
    for name in plane_names:
        if name in sr_model.SR_planes:
            continue
        R0 = sr_model.LR_planes[name].shape[-2]
        if world == 1 or R0 < world:
            with torch.no_grad():
                sr_model(name)
            continue
        lo, hi = shard_bounds(R0, rank, world)
        hr = sr_fn(name, band_roi(lo, hi, R0))
        band = hr[0, :, lo * sf: hi * sf, :].permute(1, 0, 2).contiguous()        # [rows, C, W]: rows first for gather_rows
        assert not torch.isnan(band).any(), "the ROI must cover the band"
        counts = [sf * (shard_bounds(R0, r, world)[1] - shard_bounds(R0, r, world)[0]) for r in range(world)]
        full = gather_row_blocks(band, counts, group)
        sr_model.SR_planes[name] = full.permute(1, 0, 2).unsqueeze(0).contiguous()
    return [sr_model.SR_planes[n] for n in plane_names]


def _dense(t):
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


def allreduce_in_place(tensors, group=None, scale=None):
    """Sum DENSE tensors (row-major, or channels_last like the plane gradients) over the ranks IN PLACE: one asynchronous all-reduce per
    tensor, all queued before the first wait, no flattening copy in and none out; then one fused multiply by `scale`.  A channels_last
    [N,C,H,W] tensor is handed to the collective as its row-major [N,H,W,C] view (same storage): every backend reduces a plain contiguous
    buffer.  This is the RCCL path of allreduce_gradients; tests/test_distributed.py runs it over gloo on host tensors."""
    if not tensors:
        return
    for t in tensors:
        if not _dense(t):
            raise ValueError("allreduce_in_place: tensor with strides %s is not dense" % (tuple(t.stride()),))
    views = [t if t.is_contiguous() else t.permute(0, 2, 3, 1) for t in tensors]
    works = [dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group, async_op=True) for v in views]
    for wk in works:
        wk.wait()
    if scale is not None:
        torch._foreach_mul_(list(tensors), scale)


def allreduce_coalesced(tensors, group=None, scale=None, bucket_bytes=64 << 20):
    """Sum MANY SMALL dense tensors over the ranks: flat buckets of ~bucket_bytes (one concatenation in, one all-reduce per bucket, all queued
    before the first wait, one multiply by `scale`, one fused copy back).  The EDSR gradient of a refine iteration is 69 tensors of 2.4-9.4 MB
    (173 MB): one collective each would pay RCCL's per-call latency 69 times for messages far below the size at which the xGMI
    links run at their rate; three buckets do not, and the two copies move 2 x 173 MB through HBM (~0.1 ms of an ~88 ms iteration)."""
    tensors = list(tensors)
    if not tensors:
        return
    # a bucket holds ONE dtype (torch.cat would silently promote a mixed bucket: a bf16 gradient reduced in f32 and cast back) and is closed
    # BEFORE the tensor that would take it past bucket_bytes (a bucket of small tensors never exceeds the bound; a larger tensor is its own)
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    buckets = []
    for group_ in by_dtype.values():
        cur, size = [], 0
        for t in group_:
            nb = t.numel() * t.element_size()
            if cur and size + nb > bucket_bytes:
                buckets.append(cur)
                cur, size = [], 0
            cur.append(t)
            size += nb
        if cur:
            buckets.append(cur)
    flats = [torch.cat([t.reshape(-1) for t in b]) for b in buckets]                  # (logical order: any dense layout)
    works = [dist.all_reduce(f, op=dist.ReduceOp.SUM, group=group, async_op=True) for f in flats]
    for wk in works:
        wk.wait()
    if scale is not None:
        torch._foreach_mul_(flats, scale)
    dst, src = [], []
    for b, f in zip(buckets, flats):
        off = 0
        for t in b:
            dst.append(t)
            src.append(f[off: off + t.numel()].view(t.shape))
            off += t.numel()
    torch._foreach_copy_(dst, src)


# RCCL path of allreduce_gradients: with more than DIRECT_MAX_TENSORS dense tensors, those below COALESCE_BELOW bytes share buckets.  The bound is
# per TENSOR and sits below the plane gradients (7.7 MB each at 200^2) and the decoder blobs' own size class: those stay in place like round 3
# measured them; what shares buckets is an SR network's many 2-4 MB weight gradients (at 16 MB the planes were flattened again).
DIRECT_MAX_TENSORS = 8
COALESCE_BELOW = 4 << 20


class OverlappedSRGradSync:
    """The EDSR weight gradient of a data-parallel SR-refinement iteration (173 MB) all-reduced WHILE the SR backward is still running,
    and the `grad_sync` callable of training.TrainStep for everything else.

    The reference has no distributed code; its autograd produces the 69 weight gradients layer by layer, the last layer
    first (models.py:789-822).  The library's batched backward (ops.PlanesSRBatchFn -> nvsr_planes_sr_backward_batch_marks) fills ONE blob in
    state-dict order and records an event each time a suffix of it is final; `reduce_marked` queues, per bucket of ~bucket_bytes, an asynchronous
    all-reduce BEHIND that event on a collective stream -- the links move bucket k while the matrix pipe computes the layers of bucket k + 1 --
    and finally makes the iteration's stream wait for the collectives and multiplies the blob by 1 / world.  What autograd then hands to the
    parameters is already the averaged gradient; only the LAST bucket's transfer (the first layers of the network) is exposed.

        sync = OverlappedSRGradSync(sr_model, other_parameters=planes + decoder_params)
        step = TrainStep(..., SR_model=sr_model, grad_sync=sync)          # TrainStep calls sync() between backward and the optimizers

    sync() averages the OTHER parameters' gradients (allreduce_gradients) -- and the SR network's too in an iteration whose SR backward did not go
    through the batched path (one plane, exact f32 per plane: PlanesSR.forward), so the parameters of all ranks stay identical either way.
    RCCL: collectives on the process group's own stream, ordered behind the bucket's event through `stream` (a stream that only ever carries
    event waits; pass TrainStep's idle prologue stream to spend no additional hardware queue: training.TrainStep does).  gloo with host tensors
    (tests/test_distributed.py): asynchronous all-reduces of the bucket views.  gloo with device tensors (one-GPU rehearsals): staged through the
    host at the end of the backward -- no overlap, same values.  Buckets are reduced in place: no flattening copy."""

    def __init__(self, sr_model, other_parameters=(), group=None, bucket_bytes=48 << 20, stream=None, single_rank_ok=False):
        self.sr_model, self.other, self.group, self.bucket_bytes, self.stream = sr_model, list(other_parameters), group, int(bucket_bytes), stream
        self.single_rank_ok = bool(single_rank_ok)          # run the marked path on a one-rank group too (the RCCL test of a one-GPU box)
        self.reduced_in_backward = False          # did this iteration's SR backward hand its blob over?
        self.stats = {"buckets": 0, "bytes": 0}   # of the last overlapped backward
        self._events = []
        self.attach(sr_model)

    def attach(self, sr_model):
        sr_model.__dict__["grad_bucket_sync"] = self          # PlanesSR.forward_many puts it into the cfg of ops.PlanesSRBatchFn

    def detach(self):
        self.sr_model.__dict__.pop("grad_bucket_sync", None)

    def active(self):
        return world_info(self.group)[1] > 1 or (self.single_rank_ok and dist.is_available() and dist.is_initialized())

    def plan(self, layer_floats, blob):
        """-> [(first layer of the bucket, lo, hi, event)]: suffixes of the blob in the order they become final.  A bucket is closed once it holds
        bucket_bytes; the events are created (recorded once) here and re-recorded by the library where the bucket is final."""
        offs = [0]
        for n in layer_floats:
            offs.append(offs[-1] + int(n))
        assert offs[-1] == blob.numel()
        marks, hi = [], offs[-1]
        for l in range(len(layer_floats) - 1, -1, -1):
            if (hi - offs[l]) * blob.element_size() >= self.bucket_bytes or l == 0:
                marks.append([l, offs[l], hi, None])
                hi = offs[l]
        if blob.is_cuda:
            while len(self._events) < len(marks):
                self._events.append(torch.cuda.Event())
            cur = torch.cuda.current_stream(blob.device)
            for m, ev in zip(marks, self._events):
                ev.record(cur)                    # (creates the handle; the library records it again behind the bucket's last layer)
                m[3] = ev
        return [tuple(m) for m in marks]

    def reduce_marked(self, blob, marks):
        """all-reduce the buckets [lo, hi) of `blob`, each behind its event, then average; returns with the CURRENT stream ordered behind all of it"""
        rank, world = world_info(self.group)
        self.reduced_in_backward = True
        self.stats = {"buckets": len(marks), "bytes": blob.numel() * blob.element_size()}
        if world == 1 and not self.single_rank_ok:
            return
        views = [blob[lo:hi] for _, lo, hi, _ in marks]
        backend = dist.get_backend(self.group)
        if blob.is_cuda and backend != "nccl":                 # rehearsal on one GPU: host-staged, after the fact
            for v in views:
                host = v.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                v.copy_(host)
        elif blob.is_cuda:
            cur = torch.cuda.current_stream(blob.device)
            side = self.stream
            if side is None:
                side = self.stream = torch.cuda.Stream(device=blob.device)
            works = []
            with torch.cuda.stream(side):
                for v, (_, _, _, ev) in zip(views, marks):
                    side.wait_event(ev)                        # the bucket is final on the iteration's stream
                    works.append(dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group, async_op=True))    # (RCCL's stream waits for `side`)
            with torch.cuda.stream(cur):
                for wk in works:
                    wk.wait()                                  # the iteration's stream behind the collectives (no host wait)
            blob.record_stream(side)
        else:
            works = [dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for v in views]
            for wk in works:
                wk.wait()
        blob.mul_(1.0 / world)

    def __call__(self):
        """TrainStep.grad_sync: between the backward and the optimizer steps"""
        grads = [p.grad for p in self.other if p.grad is not None]
        if not self.reduced_in_backward:                       # the SR backward took another path this iteration (or the SR network did not train)
            grads += [p.grad for p in self.sr_model.parameters() if p.grad is not None]
        self.reduced_in_backward = False
        allreduce_gradients(grads, self.group)


def allreduce_gradients(tensors, group=None, bucket_bytes=32 << 20, average=True):
    """Sum (or average) a list of gradient tensors over the ranks.  The loss is a mean over rays (train_nerf.py:884-891), so with rays
    sharded evenly the data-parallel gradient is the average of the per-rank gradients.
    RCCL: every dense tensor (row-major or channels_last -- the plane gradients are channels_last views) is reduced IN PLACE by its own
    asynchronous all-reduce, all queued before the first wait: no flattening copy in, none out (round 2 concatenated the 23 MB of plane
    gradients into a bucket and copied them back every step) -- up to DIRECT_MAX_TENSORS tensors; beyond that (an SR network in `what`) the
    tensors below COALESCE_BELOW bytes go through allreduce_coalesced's buckets.  Other backends (gloo rehearsals) and non-dense tensors go through flat buckets of
    ~bucket_bytes, staged through the host when the backend needs it."""
    rank, world = world_info(group)
    if world == 1:
        return
    tensors = [t for t in tensors if t is not None]
    scale = 1.0 / world if average else None

    # RCCL: one asynchronous collective per tensor, queued back to back on its stream (4 planes + 2 decoder blobs per step), then one wait
    direct = [t for t in tensors if t.is_cuda and _dense(t)] if dist.get_backend(group) == "nccl" else []
    ids = {id(t) for t in direct}
    if len(direct) > DIRECT_MAX_TENSORS:       # many tensors (an SR network's 69 weight gradients): the small ones in buckets, the large ones in place
        small = [t for t in direct if t.numel() * t.element_size() < COALESCE_BELOW]
        small_ids = {id(t) for t in small}
        direct = [t for t in direct if id(t) not in small_ids]
        allreduce_coalesced(small, group, scale)
    allreduce_in_place(direct, group, scale)
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([t.reshape(-1) for t in bucket])
        cdev = _collective_device(flat, group)
        if cdev != flat.device:           # (gloo rehearsal with device tensors)
            host = flat.to(cdev)
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if scale is not None:
            flat *= scale
        off = 0
        for t in bucket:
            t.copy_(flat[off: off + t.numel()].view(t.shape))          # (logical order both ways: any strides of t)
            off += t.numel()
        bucket, size = [], 0

    for t in tensors:
        if id(t) in ids:
            continue
        bucket.append(t)
        size += t.numel() * t.element_size()
        if size >= bucket_bytes:
            flush()
    flush()
