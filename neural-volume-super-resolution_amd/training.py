"""Train / evaluate step glue -- the computational core of the reference's `train()` and `evaluate()` closures
(train_nerf.py:790-923 and :625-788), without its logging, dataset and checkpoint plumbing.

What changes against the reference:
  * rays of the selected pixels only are generated (`nvsr_get_ray_bundle_at`) -- the reference builds all H*W rays per iteration
    and gathers `num_random_rays` of them (:814,:842-844); results are bit-identical;
  * everything else is the reference's arithmetic in the reference's order: pixel selection with numpy's global RNG (:816-846),
    coarse / fine MSE gating by `what2train` and `super_resolution.training.loss` (:884-891), virtual batches and per-module
    optimizer gating (:848-853, :905-914), the SR-vs-no-SR double render and PSNR bookkeeping of `evaluate()` (:655-713);
  * optional: `DevicePixelSampler` draws the pixels on the device (`nvsr_sample_pixels`: one kernel instead of a host permutation of H*W
    indices), and the coarse + fine MSE of one step come from one launch (`mse_loss_pair`)."""
import os
from collections.abc import Mapping

import numpy as np
import torch

from . import capi
from .nerf_helpers import get_focal, img2mse, mse2psnr
from .train_utils import _cfg, eval_nerf, pack_rays, run_one_iter_of_nerf


def downsampling_offset(ds_factor):
    """train_nerf.py:610: sub-pixel offset of rays rendered for images that were down-sampled by ds_factor"""
    return (ds_factor - 1) / (2 * ds_factor)


def get_ray_bundle_at(height, width, focal_length, tform_cam2world, rows_cols, downsampling_offset=0.0):
    """get_ray_bundle(height, width, focal, c2w, downsampling_offset=...)[rows, cols] -> (ray_origins [N,3], ray_directions [N,3])"""
    c2w = capi.f32c(tform_cam2world)
    rc = rows_cols.to(device=c2w.device, dtype=torch.int32).contiguous()
    N = rc.shape[0]
    ro = torch.empty((N, 3), dtype=torch.float32, device=c2w.device)
    rd = torch.empty_like(ro)
    if N == 0:
        return ro, rd
    capi.call("nvsr_get_ray_bundle_at", height, width, float(get_focal(focal_length, "H")), float(get_focal(focal_length, "W")),
              capi.ptr(c2w), float(downsampling_offset), N, capi.ptr(rc), capi.ptr(ro), capi.ptr(rd), capi.stream())
    return ro, rd


def mse_loss(x, y, weights=None):
    """train_nerf.py:618-623"""
    if weights is None:
        return torch.nn.functional.mse_loss(x, y)
    return (torch.nn.functional.mse_loss(x, y, reduction="none").mean(1) * weights).mean()


def avg_downsampling(pixels, ds_factor):
    """train_nerf.py:614-616: average every rendered (ds x ds) patch into one pixel"""
    return torch.mean(pixels.reshape(-1, ds_factor, ds_factor, 3), dim=(1, 2))


def downsample_plane(plane, ds_factor, plane_interp, align_corners, antialias=False):
    """nerf_helpers.py:498-499"""
    return torch.nn.functional.interpolate(plane, scale_factor=1 / ds_factor, mode=plane_interp, align_corners=align_corners, antialias=antialias)


def calc_im_inconsistency_loss(sr, ds_factor, plane_interp, align_corners=True, gt_lr=None, gt_hr=None):
    """nerf_helpers.py:501-505"""
    assert (gt_lr is None) ^ (gt_hr is None)
    ref = gt_lr if gt_hr is None else downsample_plane(gt_hr, ds_factor, plane_interp, align_corners, antialias=True)
    return torch.nn.functional.l1_loss(ref, downsample_plane(sr, ds_factor, plane_interp, align_corners, antialias=True))


def select_training_pixels(img_target, num_random_rays, consistency_ds=None):
    """train_nerf.py:816-846.  -> (rows_cols [N,2] int64 on img_target's device, target_s [n_targets, C]).
    consistency_ds: None for an ordinary iteration; the coupler's ds_factor for an image-consistency iteration, where
    num_random_rays // ds^2 LR pixels are drawn and each is expanded to its (ds x ds) patch of HR pixel coordinates."""
    dev = img_target.device
    h, w = img_target.shape[0], img_target.shape[1]
    # `coords = stack(meshgrid_xy(arange(H), arange(W)), -1).reshape(-1, 2)` (:818-828) enumerates pixels COLUMN by column:
    # coords[k] = (row k % H, col k // H); keeping that order keeps a seeded run on the reference's pixels
    if consistency_ds is None:
        n = min(h * w, num_random_rays)
        flat = np.random.choice(h * w, size=(n), replace=False)
        sel = torch.from_numpy(np.stack([flat % h, flat // h], -1)).to(dev)
        return sel, img_target[sel[:, 0], sel[:, 1], :]
    ds = int(consistency_ds)
    n = min(h * w, num_random_rays // (ds ** 2))
    flat = np.random.choice(h * w, size=(n), replace=False)
    corners = torch.from_numpy(np.stack([flat % h, flat // h], -1)).to(dev)
    target_s = img_target[corners[:, 0], corners[:, 1], :]
    return _expand_consistency_patches(corners, ds), target_s


def _expand_consistency_patches(corners, ds):
    """train_nerf.py:829-835: every drawn LR pixel becomes its (ds x ds) patch of HR pixel coordinates"""
    corners = ds * corners[:, None, None, :]
    ar = torch.arange(ds, device=corners.device, dtype=corners.dtype)
    rows = corners[..., :1] + ar.reshape(1, -1, 1, 1).repeat(1, 1, ds, 1)
    cols = corners[..., 1:] + ar.reshape(1, 1, -1, 1).repeat(1, ds, 1, 1)
    return torch.cat([rows, cols], -1).reshape(-1, 2)


class DevicePixelSampler:
    """`pixel_sampler` of TrainStep drawing on the device: n distinct pixels of the target image, uniform, per call -- the reference's
    `np.random.choice(H * W, n, replace=False)` (train_nerf.py:836-838) with `nvsr_sample_pixels` as the generator: ONE kernel writes the
    (row, col) pairs in the reference's column-by-column enumeration and gathers `target_s`, where the reference permutes all H*W indices
    on the host (640 000 for an 800 x 800 view: several ms, more than the whole step here).  Not the reference's random stream: a seeded run
    trains on different (equally distributed) pixels than a seeded reference run; `select_training_pixels` keeps the reference's draws.

    seed     the draws of call k are entries of the permutation keyed by (seed, k)
    n_draw, lo   data-parallel training on ONE global batch: every rank constructs the sampler with the same seed, n_draw = the global
             number of rays and lo = its first ray; a call then returns rays [lo, lo + num_random_rays) of the global draw."""

    def __init__(self, seed=0, n_draw=None, lo=0):
        self.seed, self.n_draw, self.lo, self.calls = int(seed), n_draw, int(lo), 0
        self.state = None        # device words {seed, calls, 0, 0} once use_device_state() was called (HIP-graph replays)

    def key(self):
        """key of the next draw: nvsr_sample_key(seed, calls) = splitmix64(splitmix64(seed) ^ calls) (include/nvsr.h)"""
        return int(capi.lib().nvsr_sample_key(self.seed & 0xFFFFFFFFFFFFFFFF, self.calls & 0xFFFFFFFFFFFFFFFF))

    def use_device_state(self, device):
        """keep (seed, calls) in device memory from now on: every draw is an `nvsr_sample_pixels_seq` launch that derives the key on the
        device and advances `calls` itself, so a launch captured into a HIP graph draws a new batch at every replay -- the batch the eager
        sampler with the same seed draws at that call."""
        def s64(v):
            v &= 0xFFFFFFFFFFFFFFFF
            return v - (1 << 64) if v >= 1 << 63 else v
        self.state = torch.tensor([s64(self.seed), s64(self.calls), 0, 0], dtype=torch.int64, device=device)
        return self

    def __call__(self, img_target, num_random_rays, consistency_ds=None):
        img = capi.f32c(img_target)
        if not img.is_cuda:
            raise RuntimeError("DevicePixelSampler draws on the GPU: the target image must be a CUDA tensor (select_training_pixels draws on the host)")
        h, w, ch = img.shape
        ds = 1 if consistency_ds is None else int(consistency_ds)
        n = min(h * w, num_random_rays // (ds ** 2))
        # a rank's window [lo, lo + n) of one global draw; on an image-consistency iteration the draw is of LR pixels, ds^2 rays each
        first = (self.lo // (ds ** 2)) if self.n_draw is not None else 0
        if self.n_draw is not None and (self.lo % (ds ** 2) or first + n > self.n_draw // (ds ** 2)):
            raise ValueError("DevicePixelSampler: rays [%d, %d) do not tile a global draw of %d rays in %d x %d patches"
                             % (self.lo, self.lo + num_random_rays, self.n_draw, ds, ds))
        if first + n > h * w:
            raise ValueError("DevicePixelSampler: rays [%d, %d) of a draw from %d pixels" % (first, first + n, h * w))
        rc = torch.empty((n, 2), dtype=torch.int32, device=img.device)
        target_s = torch.empty((n, ch), dtype=torch.float32, device=img.device)
        if self.state is not None:
            capi.call("nvsr_sample_pixels_seq", h * w, h, w, capi.ptr(self.state), first, n, capi.ptr(img), ch, capi.ptr(rc), capi.ptr(target_s),
                      capi.stream())
        else:
            capi.call("nvsr_sample_pixels", h * w, h, w, self.key(), first, n, capi.ptr(img), ch, capi.ptr(rc), capi.ptr(target_s), capi.stream())
        self.calls += 1
        return (rc if consistency_ds is None else _expand_consistency_patches(rc, ds)), target_s


class _MsePair(torch.autograd.Function):
    """(F.mse_loss(a, t), F.mse_loss(b, t)) from one launch; the gradients 2 (x - t) / n are written by the same kernel"""

    @staticmethod
    def forward(ctx, a, b, t):
        a, b, t = capi.f32c(a), capi.f32c(b), capi.f32c(t)
        losses = torch.empty(2, dtype=torch.float32, device=a.device)
        need_a, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        ga = torch.empty_like(a) if need_a else None
        gb = torch.empty_like(b) if need_b else None
        capi.call("nvsr_mse_pair", a.numel(), capi.ptr(a), capi.ptr(b), capi.ptr(t), capi.ptr(losses), capi.ptr(ga), capi.ptr(gb), capi.stream())
        ctx.save_for_backward(ga, gb)
        ctx.set_materialize_grads(False)
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g0, g1):
        ga, gb = ctx.saved_tensors
        return (None if ga is None or g0 is None else ga * g0), (None if gb is None or g1 is None else gb * g1), None


class _MsePairSum(torch.autograd.Function):
    """(F.mse_loss(a, t), F.mse_loss(b, t), their sum) from one launch (nvsr_mse_pair_sum); the backward is ONE launch that writes both gradients
    with the incoming gradients folded in (nvsr_mse_pair_backward) -- autograd's own chain for `coarse_loss + fine_loss` is an addition in the
    forward and two multiplies behind the stored gradients in the backward: three few-microsecond kernels of a 1.6 ms iteration.  The three
    scalars are views of one [3] tensor (`packed`), which the iteration's metrics copy to the host as they are."""

    @staticmethod
    def forward(ctx, a, b, t):
        a, b, t = capi.f32c(a), capi.f32c(b), capi.f32c(t)
        losses = torch.empty(3, dtype=torch.float32, device=a.device)
        capi.call("nvsr_mse_pair_sum", a.numel(), capi.ptr(a), capi.ptr(b), capi.ptr(t), capi.ptr(losses), capi.stream())
        ctx.save_for_backward(a, b, t)
        ctx.set_materialize_grads(False)
        return losses[0], losses[1], losses[2], losses

    @staticmethod
    def backward(ctx, g0, g1, g2, _gp):
        a, b, t = ctx.saved_tensors
        need_a, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        def scale(g, gs):          # the incoming gradient of one loss: its own + the sum's (either may be absent)
            if g is None:
                return gs
            return g if gs is None else g + gs
        sa, sb = scale(g0, g2), scale(g1, g2)
        ga = torch.empty_like(a) if need_a and sa is not None else None
        gb = torch.empty_like(b) if need_b and sb is not None else None
        if ga is not None or gb is not None:
            sa_, sb_ = (None if sa is None else capi.f32c(sa)), (None if sb is None else capi.f32c(sb))
            capi.call("nvsr_mse_pair_backward", a.numel(), capi.ptr(a), capi.ptr(b), capi.ptr(t), capi.ptr(sa_), capi.ptr(sb_), capi.ptr(ga), capi.ptr(gb),
                      capi.stream())
        return ga, gb, None


def _pair_on_device(a, b, target):
    return (a.is_cuda and a.shape == b.shape == target.shape and a.dtype == b.dtype == target.dtype == torch.float32
            and 0 < a.numel() <= capi.MSE_PAIR_MAX_ELEMS and not target.requires_grad)


def mse_loss_pair_sum(a, b, target):
    """(mse_loss(a, target), mse_loss(b, target), their sum, packed) -- train_nerf.py:893-905's coarse loss, fine loss and `coarse_loss +
    fine_loss`; `packed` = the [3] device tensor holding the three (None off the device path: torch's own operators)."""
    if _pair_on_device(a, b, target):
        lc, lf, both, packed = _MsePairSum.apply(a, b, target)
        return lc, lf, both, packed.detach()
    lc, lf = mse_loss(a, target), mse_loss(b, target)
    return lc, lf, lc + lf, None


def mse_loss_pair(a, b, target):
    """(mse_loss(a, target), mse_loss(b, target)) -- train_nerf.py:893-905's coarse and fine losses.  One `nvsr_mse_pair` launch for float32
    CUDA images of one shape (a training batch); anything else goes through torch like mse_loss()."""
    if (a.is_cuda and a.shape == b.shape == target.shape and a.dtype == b.dtype == target.dtype == torch.float32
            and 0 < a.numel() <= capi.MSE_PAIR_MAX_ELEMS and not target.requires_grad):
        return _MsePair.apply(a, b, target)
    return mse_loss(a, target), mse_loss(b, target)


class StepMetrics(Mapping):
    """loss / psnr / coarse_loss / fine_loss of one iteration, with the reference's keys and python-float values.

    The reference reads them with `.item()` in the middle of the iteration (train_nerf.py:893-921): the host waits for the forward pass
    before it may enqueue the backward pass, and again for the optimizer before the next iteration -- on this GPU that leaves the queue
    empty for ~15 % of a 10 ms step.  Here the scalars are gathered on the device after the optimizer steps are enqueued and copied to
    pinned host memory on the step's stream; the first read of a value waits for that copy.  A loop that logs every k-th iteration
    never waits on the others."""

    KEYS = ("loss", "psnr", "coarse_loss", "fine_loss")
    RANGE_ERROR = ("NVSR_ARITH_F16X2 range exceeded in training iteration %s (a weight >= 255, a feature or activation >= 4094, or a non-finite "
                   "parameter): the kernels wrote NaN; set model.arithmetic = 'bf16x3' (capi.set_decoder_arithmetic('bf16x3')) for this model")

    def __init__(self, loss, rendering_loss, coarse_loss, fine_loss, with_psnr, range_word=None, it=None, packed=None):
        self._present = dict(loss=True, psnr=with_psnr and isinstance(rendering_loss, torch.Tensor), coarse_loss=coarse_loss is not None,
                             fine_loss=fine_loss is not None)
        self._it = it
        self._vals = None
        self._packed_host = None
        dev = loss.device
        if packed is not None and dev.type == "cuda":
            # packed = [coarse_loss, fine_loss, loss = rendering_loss] as the loss kernel wrote them (mse_loss_pair_sum): two asynchronous copies
            # (the three scalars; the range flag word) and no launch, where gathering five scalars is a conversion, a stack and a copy
            self._packed_host = torch.empty(3, dtype=torch.float32, pin_memory=True)
            self._packed_host.copy_(packed, non_blocking=True)
            self._flag_host = None
            if range_word is not None:
                self._flag_host = torch.empty(range_word.shape, dtype=range_word.dtype, pin_memory=True)
                self._flag_host.copy_(range_word, non_blocking=True)
            self._host = None
            self._event = torch.cuda.Event()
            self._event.record()
            return
        src = [loss, rendering_loss, coarse_loss, fine_loss]
        # (5th value: the library's range flag as it stands at the end of this iteration -- capi.RangeFlag; read with the others)
        src.append(range_word.reshape(()) if range_word is not None else 0.0)
        vals = torch.stack([(v.detach().to(torch.float32).reshape(()) if isinstance(v, torch.Tensor)
                             else torch.full((), float("nan") if v is None else float(v), device=dev)) for v in src])
        if dev.type == "cuda":
            self._host = torch.empty(5, dtype=torch.float32, pin_memory=True)
            self._host.copy_(vals, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()
        else:
            self._host, self._event = vals, None

    def _read(self):
        if self._vals is None:
            if self._event is not None:
                self._event.synchronize()
            if self._packed_host is not None:
                c, f, s = self._packed_host.tolist()
                flag = 0.0 if self._flag_host is None else float(self._flag_host.reshape(-1)[0].item())
                self._vals = [s, s, c, f, flag]
            else:
                self._vals = self._host.tolist()
        if self._vals[4] != 0.0:
            raise capi.NvsrError((self.RANGE_ERROR % ("?" if self._it is None else self._it)) + " [range flag bits %d: 1 = decoder kernels, 2 = SR network]" % int(self._vals[4]))
        return self._vals

    def poll(self):
        """raise if this iteration has finished with the range flag up; never waits"""
        if self._vals is None and self._event is not None and not self._event.query():
            return
        self._read()

    def __getitem__(self, k):
        if k not in self._present:
            raise KeyError(k)
        if not self._present[k]:
            return None
        v = self._read()
        return mse2psnr(v[1]) if k == "psnr" else v[self.KEYS.index(k)]

    def __iter__(self):
        return iter(self.KEYS)

    def __len__(self):
        return len(self.KEYS)


class TrainStep:
    """One optimisation iteration of train() for the planes model.

    what2train  subset of {'LR_planes', 'decoder', 'SR'} (cfg.nerf.train.what, train_nerf.py:75-77)
    optimizer   decoder optimizer or None;  SR_optimizer or None;  planes_optimizer: any object with zero_grad() / step() (the
                reference's PlanesOptimizer, or a torch optimizer over the plane parameters) or None
    sr_loss     cfg.super_resolution.training.loss in {'both', 'fine', 'coarse'} (:885,:889)"""

    def __init__(self, model_coarse, model_fine, options, what2train, optimizer=None, SR_optimizer=None, planes_optimizer=None, SR_model=None,
                 virtual_batch_size=1, rendering_loss_w=1.0, im_inconsistency_loss_w=None, sr_loss="both", ds_factor=1,
                 separate_decoder_sr=False, grad_sync=None, pixel_sampler=None):
        self.mc, self.mf, self.options = model_coarse, model_fine, options
        self.what = set(what2train)
        self.optimizer, self.SR_optimizer, self.planes_optimizer, self.SR_model = optimizer, SR_optimizer, planes_optimizer, SR_model
        self.vbs = max(1, int(virtual_batch_size))
        self.rendering_loss_w, self.im_inconsistency_loss_w = rendering_loss_w, im_inconsistency_loss_w
        self.sr_loss, self.ds_factor, self.separate_decoder_sr = sr_loss, int(ds_factor), separate_decoder_sr
        self.grad_sync = grad_sync          # callable() run between backward and the optimizer steps (data-parallel all-reduce)
        self.pixel_sampler = pixel_sampler or select_training_pixels   # (img_target, num_random_rays, consistency_ds) -> (rows_cols, target_s)
        # SR training: pixels, rays and regions of interest on a side stream ahead of the iteration (_draw_rays); NVSR_PROLOGUE_AHEAD=0: in-stream
        self.prologue_ahead = os.environ.get("NVSR_PROLOGUE_AHEAD", "1") != "0"
        import collections
        self._pending = collections.deque()

    def __call__(self, it, img_target, pose_target, H, W, focal, cur_ds_factor, scene_id, scene_config, num_random_rays, sr_iter=False,
                 im_consistency_iter=False, confinements=(), randoms=None):
        # an iteration whose operands left NVSR_ARITH_F16X2's range wrote NaN into the loss and the parameters: found without a host wait,
        # from the metrics of the iterations that have finished by now (each carries the range flag of its end), and raised
        while self._pending and (self._pending[0]._vals is not None or self._pending[0]._event is None or self._pending[0]._event.query()):
            self._pending.popleft().poll()
        word = self._range_word(img_target.device)
        out = self.run(it, img_target, pose_target, H, W, focal, cur_ds_factor, scene_id, scene_config, num_random_rays, sr_iter,
                       im_consistency_iter, confinements, randoms)
        m = StepMetrics(*out, range_word=word, it=it, packed=self.__dict__.pop("_packed_metrics", None))
        if m._event is not None:
            self._pending.append(m)
            if len(self._pending) > 64:          # (a consumer that never lets the queue drain: wait for the oldest)
                self._pending.popleft()._read()
        return m

    def _range_word(self, device):
        if device.type != "cuda":
            return None
        flag = capi.range_flag(device)
        # zeroed on the stream at the start of EVERY iteration: the word an iteration's metrics carry is what its own launches raised -- not an
        # evaluation frame, a direct model call or a stand-alone planes_sr in F16X2 that ran in between (those read and heal the flag themselves,
        # or do not look at it at all)
        flag.reset()
        return flag.word

    def run(self, it, img_target, pose_target, H, W, focal, cur_ds_factor, scene_id, scene_config, num_random_rays, sr_iter=False,
            im_consistency_iter=False, confinements=(), randoms=None):
        """the iteration itself -> (loss, rendering_loss, coarse_loss, fine_loss, with_psnr), device scalars (what __call__ wraps in StepMetrics);
        nothing here waits for the device or allocates host memory: GraphedTrainStep captures it into a HIP graph"""
        first_v, last_v = it % self.vbs == 0, it % self.vbs == self.vbs - 1
        if "SR" in self.what and self.SR_model is not None:
            self.SR_model.train()
        if "decoder" in self.what:
            self.mc.train()
            if self.mf is not None:
                self.mf.train()
        if im_consistency_iter:      # render in HR although the target image is LR (:806-811)
            H, W, focal, cur_ds_factor = H * self.ds_factor, W * self.ds_factor, focal * self.ds_factor, cur_ds_factor // self.ds_factor
        ro, rd, target_s = self._draw_rays(img_target, pose_target, H, W, focal, cur_ds_factor, scene_id, scene_config, num_random_rays,
                                           self.ds_factor if im_consistency_iter else None)
        batch_rays = (ro, rd)               # (run_one_iter_of_nerf indexes [0] / [1]: the reference's stacked tensor costs a copy kernel)
        if first_v:
            for o in (self.optimizer, self.SR_optimizer):
                if o is not None:
                    o.zero_grad()
        if self.planes_optimizer is not None:
            self.planes_optimizer.zero_grad()
        if hasattr(self.mf, "SR_model") and sr_iter and "LR_planes" in self.what:
            self.mf.SR_model.clear_SR_planes(all_planes=True)
            self.mf.assign_LR_planes(scene=scene_id)
        try:
            out = run_one_iter_of_nerf(H, W, focal, self.mc, self.mf, batch_rays, self.options, scene_id, mode="train", scene_config=scene_config,
                                       randoms=randoms)
        finally:
            if self.mf is not None:          # (the regions of interest drawn ahead for THIS batch never outlive it, consumed or not)
                self.mf.__dict__.pop("_roi_hint", None)
        rgb_coarse, rgb_fine = out[0], out[3]
        target = target_s[..., :3]
        if im_consistency_iter:
            rgb_coarse = avg_downsampling(rgb_coarse, self.ds_factor)
            rgb_fine = None if rgb_fine is None else avg_downsampling(rgb_fine, self.ds_factor)
        coarse_loss = fine_loss = both = packed = None
        trains_scene = bool(self.what & {"decoder", "LR_planes"})
        if self.rendering_loss_w is not None:
            want_c = trains_scene or self.sr_loss != "fine"
            want_f = rgb_fine is not None and (trains_scene or self.sr_loss != "coarse")
            if want_c and want_f:
                coarse_loss, fine_loss, both, packed = mse_loss_pair_sum(rgb_coarse, rgb_fine, target)
            elif want_c:
                coarse_loss = mse_loss(rgb_coarse, target)
            elif want_f:
                fine_loss = mse_loss(rgb_fine, target)
        if both is not None:
            rendering_loss = both           # (coarse_loss + fine_loss, added by the loss kernel)
        else:
            rendering_loss = (coarse_loss if coarse_loss is not None else 0.0) + (fine_loss if fine_loss is not None else 0.0)
        loss_w = self.im_inconsistency_loss_w if im_consistency_iter else self.rendering_loss_w
        loss = rendering_loss if loss_w == 1.0 else loss_w * rendering_loss        # (x * 1.0 is x: one kernel and its backward less)
        # the iteration's scalars as ONE device tensor where they are one (coarse, fine, sum = rendering loss = loss): StepMetrics copies it as it is
        self._packed_metrics = packed if (packed is not None and loss is both) else None
        self.apply_gradients(loss, last_v, sr_iter, confinements)
        return loss, rendering_loss, coarse_loss, fine_loss, not im_consistency_iter

    def _draw_rays(self, img_target, pose_target, H, W, focal, cur_ds_factor, scene_id, scene_config, num_random_rays, consistency_ds):
        """-> (ray origins, ray directions, target pixels) of this iteration's batch (train_nerf.py:814-846).
        An iteration that trains THROUGH the SR network needs the regions of interest of the batch on the HOST before it can size a single launch
        (models.py:270-284; the reference reads them back too).  Read on the iteration's own stream, that copy waits for everything queued before
        it -- the whole previous iteration: the queue drains, and the GPU then idles while the host prepares the SR forward (measured on the refine
        workload: 20 ms of a 110 ms iteration).  Pixels, rays and regions depend on nothing the previous iteration computes, so with a device-side
        sampler they are produced on a SIDE stream and only that stream is waited for: the host has the regions ~0.2 ms after it asks, while the
        GPU is still busy with the previous iteration's backward, and runs ahead of the queue from there.  The regions are left with the fine
        model (`_roi_hint`, TwoDimPlanesModel.training_planes)."""
        def draw():
            sel, target_s = self.pixel_sampler(img_target, num_random_rays, consistency_ds)
            ro, rd = get_ray_bundle_at(H, W, focal, pose_target, sel, downsampling_offset=downsampling_offset(cur_ds_factor))
            return ro, rd, target_s

        mf = self.mf
        sr = getattr(mf, "SR_model", None) if mf is not None else None
        ahead = (self.prologue_ahead and sr is not None and not getattr(mf, "skip_SR_", False) and sr.training and isinstance(self.pixel_sampler, DevicePixelSampler)
                 and self.pixel_sampler.state is None and img_target.is_cuda and torch.is_tensor(pose_target) and pose_target.is_cuda
                 and mf.is_native_geometry() and not getattr(mf, "point_coords_noise", 0))
        if not ahead:
            return draw()
        dev = img_target.device
        cur = torch.cuda.current_stream(dev)
        side = self.__dict__.get("_prologue_stream")
        if side is None:
            side = self.__dict__["_prologue_stream"] = torch.cuda.Stream(device=dev, priority=-1)
        if not self._prologue_known(img_target, pose_target):
            side.wait_stream(cur)
        with torch.cuda.stream(side):
            ro, rd, target_s = draw()
            mf.set_cur_scene_id(scene_id)
            # (the rays as run_one_iter_of_nerf will pack them: NDC scenes -- LLFF, `no_ndc: False` -- project them first, train_utils.py:215-218)
            dims, rois = mf.training_rois(pack_rays(ro, rd, _cfg(scene_config, "near"), _cfg(scene_config, "far"), H, W, focal,
                                                    no_ndc=_cfg(scene_config, "no_ndc")))
            host = None
            if dims:
                host = torch.empty((len(dims), 4), dtype=torch.float32, pin_memory=True)
                host.copy_(rois.reshape(len(dims), 4), non_blocking=True)
            done = torch.cuda.Event()
            done.record(side)
        done.synchronize()                  # the side stream only
        cur.wait_stream(side)
        for x in (ro, rd, target_s):
            x.record_stream(cur)            # (allocated on the side stream, read by the iteration's stream)
        if host is not None:
            mf._roi_hint = (ro.shape[0], host.tolist())
        return ro, rd, target_s

    # bound of the side stream's input cache: entries, and bytes of device memory its retained storages may keep alive (resident datasets are alive
    # anyway; a caller that produces a new target every iteration has at most this much kept behind it)
    PROLOGUE_CACHE_ENTRIES, PROLOGUE_CACHE_BYTES = 1024, 4 << 30

    def _prologue_known(self, img_target, pose_target):
        """May the side stream read these two tensors WITHOUT waiting for the iteration's stream?  Only if it has met this very (storage, version)
        pair before -- whatever wrote them was waited for then.  The cache entry RETAINS both storages: a key on (data_ptr, _version)
        alone is met again by a freshly produced tensor -- version 0, on the block the allocator just recycled -- while its producer is still queued
        behind the previous iteration; with the storage kept alive the allocator cannot hand that address to anything else, so an equal key IS the
        same memory in the same state.  Least-recently-used entries are dropped beyond PROLOGUE_CACHE_ENTRIES / PROLOGUE_CACHE_BYTES (a dropped
        input costs one drained queue at its next use, never a stale read).  False -> the caller makes the side stream wait, and the pair is known
        from now on."""
        import collections
        cache = self.__dict__.get("_prologue_inputs")
        if cache is None:
            cache = self.__dict__["_prologue_inputs"] = collections.OrderedDict()
            self.__dict__["_prologue_bytes"] = 0
        key = (img_target.data_ptr(), img_target._version, pose_target.data_ptr(), pose_target._version)
        if key in cache:
            cache.move_to_end(key)
            return True
        keep = (img_target.untyped_storage(), pose_target.untyped_storage())
        nbytes = sum(int(st.nbytes()) for st in keep)
        cache[key] = (keep, nbytes)
        self.__dict__["_prologue_bytes"] += nbytes
        while len(cache) > 1 and (len(cache) > self.PROLOGUE_CACHE_ENTRIES or self.__dict__["_prologue_bytes"] > self.PROLOGUE_CACHE_BYTES):
            _, (_, nb) = cache.popitem(last=False)
            self.__dict__["_prologue_bytes"] -= nb
        return False

    def apply_gradients(self, loss, last_v=True, sr_iter=False, confinements=()):
        """the tail of an iteration (train_nerf.py:903-914): backward, [data-parallel: grad_sync() averages the gradients over the ranks],
        then the optimizer steps the iteration is entitled to -- every rank steps from the same averaged gradients, so the ranks' parameters
        stay identical (tests/test_distributed.py runs this tail on two gloo ranks)"""
        # (the seed of the backward pass from a cached scalar: `loss.backward()` fills a new ones_like(loss) -- one more launch -- every iteration.
        #  READ-ONLY by contract: _MsePairSum hands it to its backward kernel as an input, nothing in this package writes through it.  A caller who
        #  registers tensor hooks that modify gradients IN PLACE on the loss itself must set `step.clone_seed = True`: the hook then gets a copy)
        seed = self.__dict__.get("_seed")
        if seed is None or seed.device != loss.device or seed.dtype != loss.dtype:
            seed = self.__dict__["_seed"] = torch.ones((), dtype=loss.dtype, device=loss.device)
        if getattr(self, "clone_seed", False):
            seed = seed.clone()
        loss.backward(seed if loss.dim() == 0 else None)
        if self.grad_sync is not None:
            self.grad_sync()
        if self.planes_optimizer is not None:
            self._step(self.planes_optimizer)
        if last_v:
            if self.optimizer is not None:
                decoder_step = "decoder" not in confinements
                if "SR" in self.what and self.separate_decoder_sr:
                    decoder_step &= not sr_iter
                if decoder_step:
                    self._step(self.optimizer)
            if self.SR_optimizer is not None and sr_iter and "SR" not in confinements:
                self._step(self.SR_optimizer)

    def _step(self, opt):
        """opt.step(), and the version counters of what it updated.  Every derived copy of the parameters -- packed decoder blobs, EDSR fragment
        blobs, channel-last copies of NCHW planes, the f16 range cache -- is keyed on (data_ptr, tensor._version).  torch's FUSED optimizers
        (`Adam(fused=True)`: one multi-tensor kernel writing the parameters) do not bump the counters: the next forward would be served the blobs
        packed from the parameters as they were before the step, and training would run on frozen weights (found in round 5: the refine bench packed
        its EDSR weights once per process).  Parameters whose counter stood still across the step are bumped here; an optimizer object that cannot
        list its parameters (PlanesOptimizer wrappers) makes the models forget their copies instead."""
        if not isinstance(opt, torch.optim.Optimizer):
            opt.step()
            for m in (self.mc, self.mf, self.SR_model):
                if m is not None and hasattr(m, "invalidate"):
                    m.invalidate()
            return
        params = [p for grp in opt.param_groups for p in grp["params"] if p.grad is not None]
        before = [p._version for p in params]
        opt.step()
        stale = [p for p, v in zip(params, before) if p._version == v]
        if stale:
            mark_updated(stale)


def mark_updated(params):
    """Tell the models that `params` were written in place by something that does not bump tensor version counters (a fused optimizer stepped
    outside TrainStep, a custom kernel): bumps the counters, so every derived copy keyed on them is rebuilt at its next use.  (Writes through
    `.data` replace storage or bypass autograd altogether: call the models' invalidate() for those.)"""
    params = [p for p in params if p is not None]
    if not params:
        return
    try:
        torch._C._increment_version(params)
    except (AttributeError, TypeError):          # (a torch without the list form)
        with torch.no_grad():
            for p in params:
                p.add_(0)


class GraphedTrainStep:
    """One TrainStep iteration captured into a HIP graph and replayed: the ~30 kernel launches, the autograd graph and the Python of a step
    (0.9-1.1 ms of host time against 1.7 ms of kernels for Feature_Planes_Only.yml) become ONE hipGraphLaunch.  train_nerf.py:790-923 is the
    iteration being replayed; what a replay cannot do is change its mind: the arguments below are fixed at capture.

    step         a TrainStep with virtual_batch_size 1 whose torch optimizers were built with capturable=True and whose pixel_sampler is a
                 DevicePixelSampler (its key moves to device memory: every replay draws a new batch, the batch the eager step would draw)
    img_target, pose_target   CUDA tensors; the graph reads them at every replay (copy a new view / pose INTO them to change the view)
    randoms_fn   callable() -> the `randoms` dict of TrainStep, drawn with torch's device generators (torch advances a captured generator's
                 offset per replay; pass other generators than the default one in `generators`), or a dict of static tensors the caller refills
    The captured iteration is `step.run(0, ...)`; metrics() reads loss / psnr / coarse_loss / fine_loss of the last replay."""

    def __init__(self, step, img_target, pose_target, H, W, focal, cur_ds_factor, scene_id, scene_config, num_random_rays, randoms_fn=None,
                 generators=(), warmup=3, **step_kwargs):
        if step.grad_sync is not None:
            raise ValueError("GraphedTrainStep: a data-parallel step (grad_sync) would capture its all-reduces -- RCCL collectives, or the host-staged "
                             "gloo path, inside a HIP graph have never been run on this pool; launch the multi-rank iteration with TrainStep")
        if step.vbs != 1:
            raise ValueError("GraphedTrainStep: virtual_batch_size > 1 alternates between two different iterations; capture needs one")
        for o in (step.optimizer, step.SR_optimizer, step.planes_optimizer):
            if isinstance(o, torch.optim.Optimizer) and not all(g.get("capturable", True) for g in o.param_groups):
                raise ValueError("GraphedTrainStep: build %s with capturable=True (its step counter must live on the device)" % type(o).__name__)
        if not isinstance(step.pixel_sampler, DevicePixelSampler):
            raise ValueError("GraphedTrainStep: the pixel sampler must be a DevicePixelSampler (a host draw cannot be replayed)")
        for m in (step.mc, step.mf):
            if m is not None and getattr(m, "point_coords_noise", 0):
                raise ValueError("GraphedTrainStep: point_coords_noise is drawn by torch.normal on the CPU generator for every model call "
                                 "(models.py:291-293): a host draw cannot be replayed; use TrainStep")
        for m in (step.mc, step.mf):
            sr = getattr(m, "SR_model", None) if m is not None else None
            if sr is not None and not getattr(m, "skip_SR_", False) and "SR" in step.what:
                raise ValueError("GraphedTrainStep: an iteration that trains through the SR network sizes every launch from the regions of interest of "
                                 "its batch (models.py:270-284: read on the host per iteration) -- its launch sequence changes with every batch and "
                                 "cannot be replayed; use TrainStep (its regions are drawn ahead on a side stream)")
        if not (img_target.is_cuda and torch.as_tensor(pose_target).is_cuda):
            raise ValueError("GraphedTrainStep: img_target and pose_target must be CUDA tensors (the graph reads them at every replay)")
        self.step, self.sampler = step, step.pixel_sampler
        dev = img_target.device
        if self.sampler.state is None:
            self.sampler.use_device_state(dev)
        args = (img_target, pose_target, H, W, focal, cur_ds_factor, scene_id, scene_config, num_random_rays)
        draw = randoms_fn if callable(randoms_fn) else (lambda: randoms_fn)

        flag = capi.range_flag(dev)

        def iteration():
            flag.reset()          # (a memset node at the head of the graph: every replay reports the flag of its own launches, like TrainStep)
            out = step.run(0, *args, randoms=draw(), **step_kwargs)
            vals = [(v.detach().to(torch.float32).reshape(()) if isinstance(v, torch.Tensor)
                     else torch.full((), float("nan") if v is None else float(v), device=dev)) for v in out[:4]]
            return torch.stack(vals + [flag.word.to(torch.float32).reshape(())]), out      # (5th: the range flag, as in StepMetrics)

        # decoder training re-packs the weights every iteration, NCHW planes are converted: those kernels must be IN the graph, not skipped
        # by the models' version-keyed caches at capture time
        def forget_caches():
            from . import models
            for m in (step.mc, step.mf):
                if m is None:
                    continue
                m._packed_cache = None
                m._packed_bwd_cache = None
                for k in (getattr(m, "planes_", None) or {}):
                    models._PLANE_CACHE.pop(k, None)
                    models._PLANE_CACHE.pop(k + "/SR", None)
                # (the host copy of the box / projection constants stays: reading it back is a device-to-host copy, illegal in a capture)

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, int(warmup))):          # allocator, caches and optimizer state settle before the capture
                iteration()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        forget_caches()
        self.graph = torch.cuda.CUDAGraph()
        for g in generators:
            self.graph.register_generator_state(g)
        flag.reset()
        with torch.cuda.graph(self.graph, stream=side):
            self._vals, out = iteration()
        forget_caches()
        self.sampler.calls -= 1          # (the capture ran the sampler's Python, not its kernel: the device counter did not move)
        self._present = dict(loss=True, psnr=bool(out[4]) and isinstance(out[1], torch.Tensor), coarse_loss=out[2] is not None,
                             fine_loss=out[3] is not None)
        self.replays = 0
        # what the captured optimizers update in place.  A replay runs no Python: the tensors' version counters -- the key of every derived copy
        # (packed decoder blobs, channel-last copies of NCHW planes, EDSR fragment blobs, the f16 range cache) -- would stand still and an eager
        # render / TrainStep between replays would be served the copies of the parameters as they were after the capture
        self._updated = [p for o in (step.optimizer, step.SR_optimizer, step.planes_optimizer) if isinstance(o, torch.optim.Optimizer)
                         for grp in o.param_groups for p in grp["params"]]
        self._opaque_optimizer = any(o is not None and not isinstance(o, torch.optim.Optimizer)
                                     for o in (step.optimizer, step.SR_optimizer, step.planes_optimizer))

    def __call__(self):
        """replay the iteration once (asynchronous, on the current stream)"""
        self.graph.replay()
        self.sampler.calls += 1          # host mirror of the device counter
        self.replays += 1
        bumped = False
        if self._updated:
            try:
                torch._C._increment_version(self._updated)     # the replay wrote them in place
                bumped = True
            except (AttributeError, TypeError):                # (a torch without this hook / without its list form: drop the derived copies instead)
                pass
        if self._opaque_optimizer or not bumped:           # (an optimizer object whose parameters cannot be listed: drop the derived copies)
            for m in (self.step.mc, self.step.mf, self.step.SR_model):
                if m is not None and hasattr(m, "invalidate"):
                    m.invalidate()

    def metrics(self):
        """loss / psnr / coarse_loss / fine_loss of the last replay, python floats (waits for the device)"""
        v = self._vals.tolist()
        if v[4] != 0.0:
            raise capi.NvsrError(StepMetrics.RANGE_ERROR % ("(replay %d)" % self.replays))
        d = dict(zip(StepMetrics.KEYS, v))
        d["psnr"] = mse2psnr(v[1]) if self._present["psnr"] else None
        for k in ("coarse_loss", "fine_loss"):
            if not self._present[k]:
                d[k] = None
        return d


def evaluate_view(model_coarse, model_fine, options, scene_id, scene_config, img_target, pose_target, H, W, focal, cur_ds_factor=1,
                  SR_model=None, sr_scene=False, ds_factor=1, im_inconsistency=False):
    """One view of evaluate() (train_nerf.py:655-713): render; on an SR scene render again with the SR model bypassed as the
    reference image, and report the SR PSNR gain.  -> dict of images and metrics (python floats)."""
    for m in (model_coarse, model_fine, SR_model):
        if m is not None:
            m.eval()
    with torch.no_grad():
        from .nerf_helpers import get_ray_bundle
        ro, rd = get_ray_bundle(H, W, focal, pose_target, downsampling_offset=downsampling_offset(cur_ds_factor))

        def render_view():
            rgb_c, _, _, rgb_f, _, _, rgb_sr, _, _ = eval_nerf(H, W, focal, model_coarse, model_fine, ro, rd, options, mode="validation",
                                                             scene_id=scene_id, scene_config=scene_config)
            return rgb_c, rgb_f, rgb_sr

        rgb_coarse, rgb_fine, rgb_SR = render_view()
        tgt = img_target[..., :3]
        res = dict(loss=img2mse(rgb_fine[..., :3], tgt).item())
        res["psnr"] = mse2psnr(res["loss"])
        if sr_scene:
            if im_inconsistency:
                res["im_inconsistency"] = calc_im_inconsistency_loss(gt_hr=tgt.permute(2, 0, 1)[None], sr=rgb_fine[..., :3].permute(2, 0, 1)[None],
                                                                     ds_factor=ds_factor, plane_interp="bilinear").item()
            if SR_model is not None:
                rgb_SR = 1 * rgb_fine
                model_coarse.skip_SR(True)
                model_fine.skip_SR(True)
                rgb_coarse, rgb_fine, _ = render_view()          # the same view from the LR planes, as reference
                model_coarse.skip_SR(False)
                model_fine.skip_SR(False)
            res["fine_loss"] = img2mse(rgb_fine[..., :3], tgt).item()
            if SR_model is not None:
                res["SR_psnr_gain"] = res["psnr"] - mse2psnr(res["fine_loss"])
        else:
            res["coarse_loss"] = img2mse(rgb_coarse[..., :3], tgt).item()
            res["fine_loss"] = img2mse(rgb_fine[..., :3], tgt).item() if rgb_fine is not None else 0.0
        if SR_model is not None:
            SR_model.clear_SR_planes()       # evaluate() drops the cached super-resolved planes when it is done with a scene (:714-718)
        res.update(rgb_coarse=rgb_coarse, rgb_fine=rgb_fine, rgb_SR=rgb_SR)
        return res
