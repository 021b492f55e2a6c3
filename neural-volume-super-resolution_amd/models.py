"""Tri-plane decoder model -- mirror of the reference's models.py hot subset (a6; 8b surface).

`TwoDimPlanesModel` keeps the reference's constructor signature, attribute protocol (`planes_`, `box_coords`, `cur_id`, ...)
and state-dict keys, so checkpoints and the `PlanesOptimizer`-style wiring of the reference carry over; `forward` runs the
fused gather + MFMA decoder kernel (csrc/render.hip) on channel-last copies of the planes and a fragment-packed copy of the
weights, both cached and refreshed when the source tensors change."""
import ctypes as C
import os
import weakref
from re import search

import numpy as np
import torch
import torch.nn as nn

from . import capi
from . import ops  # noqa: F401  (registers torch.ops.nvsr.*)


def get_plane_name(scene_id, dimension):
    """models.py:110-113"""
    if scene_id is None:
        return "_D%d" % (dimension)
    return "sc%s_D%d" % (scene_id, dimension)


def plane_name2scene(plane_name):
    """models.py:115-116"""
    return search("(?<=sc).*(?=_D)", plane_name).group(0)


def get_scene_id(basedir, ds_factor, plane_res):
    """models.py:928-929"""
    return "%s_DS%d%s" % (basedir, ds_factor, "" if plane_res[0] is None else "_PlRes%d_%d" % (plane_res))


def create_plane(resolution, num_plane_channels, init_STD, channels_last=False):
    """models.py:436-439 -- planes are [1,C,H,W] tensors at the interface, like the reference.  channels_last=True (extension) gives the
    parameter torch's channels_last memory format: same shape, same indexing, same state-dict entry, but its memory IS the kernels'
    native [H][W][C] layout -- no re-layout kernel per training step and no second copy of the planes (is_native_layout below)."""
    if not isinstance(resolution, list):
        resolution = [resolution, resolution]
    p = init_STD * torch.randn(size=[1, num_plane_channels, resolution[0], resolution[1]])
    return nn.Parameter(p.contiguous(memory_format=torch.channels_last) if channels_last else p)


def is_native_layout(plane):
    """a [1,C,H,W] float32 plane whose memory is already [H][W][C] (torch.channels_last)"""
    return (plane.dim() == 4 and plane.shape[0] == 1 and plane.shape[1] > 1 and plane.dtype == torch.float32
            and not plane.is_contiguous() and plane.is_contiguous(memory_format=torch.channels_last))


class CoordProjector(nn.Module):
    """models.py:471-497: one orthonormal frame per position plane (column 0 = the plane's normal, columns 1:3 = the axes a point is
    projected on).  Up to three planes: the standard basis and two permutations of it.  More: random frames, drawn from NumPy's global
    generator with the reference's calls in the reference's order -- a seeded construction yields the reference's frames bit for bit
    (tests: g22 `planes5.rot*`)."""

    SPREAD_TRIALS = 10000

    def __init__(self, N: int = None, rot_mats=None) -> None:
        super().__init__()
        if rot_mats is not None:
            assert len(rot_mats) == N
            self.rot_mats_NON_LEARNED = rot_mats
            return
        if N <= 3:
            eye = torch.eye(3)
            frames = [eye, eye[:, [1, 0, 2]], eye[:, [2, 0, 1]]][:N]
        else:
            frames = [torch.from_numpy(f) for f in self.spread_frames(N)]
        self.rot_mats_NON_LEARNED = nn.ParameterList([nn.Parameter(f) for f in frames])

    @classmethod
    def spread_frames(cls, N):
        """N plane normals that keep away from each other and from each other's mirror images: of SPREAD_TRIALS random sets of N unit
        vectors, the set whose 2N directions (every normal and its negative) have the largest sum of squared distances to their nearest
        neighbour.  Each normal is then completed to an orthonormal frame by the QR factorisation of [normal | random 3x2]."""
        sets = np.random.uniform(low=-1, high=1, size=[cls.SPREAD_TRIALS, N, 3])
        sets /= np.sqrt(np.sum(sets ** 2, 2, keepdims=True))
        both = np.concatenate((sets, -1 * sets), 1)                                        # [trial, 2N, 3]
        d2 = np.sum((both[..., None, :] - np.expand_dims(both, 1)) ** 2, -1)               # [trial, 2N, 2N] squared distances
        nearest = np.sort(d2, 1)[:, 1, ...]                                                # (row 0 of the sort is the distance to itself)
        normals = both[np.argmax(np.sum(nearest, -1))][:N]
        frames = []
        for nrm in normals:
            while True:
                m = np.concatenate([nrm[:, None], np.random.uniform(size=[3, 2])], 1)
                if np.linalg.matrix_rank(m) == 3:
                    break
            frames.append(np.linalg.qr(m)[0])
        return frames

    def forward(self, points_dim):
        with torch.no_grad():
            return torch.matmul(points_dim[0], self.rot_mats_NON_LEARNED[points_dim[1]][:, 1:].type(points_dim[0].type()))


def to_channel_last(plane_nchw):
    """[1,C,H,W] or [C,H,W] -> [H,W,C] through the re-layout kernel; a channels_last plane is returned as a view of its own memory"""
    if is_native_layout(plane_nchw):
        return plane_nchw.permute(0, 2, 3, 1)[0]
    capi.require_cuda(plane_nchw)
    return torch.ops.nvsr.plane_to_channel_last(plane_nchw)


def from_channel_last(plane_hwc, like=None):
    """[H,W,C] -> [1,C,H,W]; for a channels_last `like` (the plane the gradient belongs to) a view with that memory format, no kernel"""
    if like is not None and is_native_layout(like) and plane_hwc.is_contiguous():
        return plane_hwc.unsqueeze(0).permute(0, 3, 1, 2)
    capi.require_cuda(plane_hwc)
    return torch.ops.nvsr.plane_from_channel_last(plane_hwc)


# plane name -> (weak reference to the source tensor, its version, channel-last copy or view).  Shared by the coarse and the fine model,
# which sample the same planes (the reference assigns one ParameterDict to both, models.py:601,707); an entry is refreshed when the source
# tensor changes and EVICTED when the source tensor dies (weakref callback), when its SR model clears its planes, and when a model moves to
# another scene -- like the reference, only the current scene's planes stay resident (multi-scene runs would otherwise grow by 368 MB
# per 800^2 scene and 5.9 GB per super-resolved scene).
_PLANE_CACHE = {}


def clear_plane_cache(prefix=None, keep_scene=None):
    """drop every cached channel-last plane; with `prefix` only the entries whose name starts with it; with `keep_scene` every entry that
    does not belong to that scene id"""
    if prefix is None and keep_scene is None:
        _PLANE_CACHE.clear()
        return
    for name in list(_PLANE_CACHE):
        if (prefix is not None and name.startswith(prefix)) or (keep_scene is not None and plane_name2scene(name.split("/")[0]) != keep_scene):
            _PLANE_CACHE.pop(name, None)


def _cache_plane(name, src):
    def _evict(ref, name=name):
        hit = _PLANE_CACHE.get(name)
        if hit is not None and hit[0] is ref:
            _PLANE_CACHE.pop(name, None)

    hit = (weakref.ref(src, _evict), src._version, to_channel_last(src.detach()))
    _PLANE_CACHE[name] = hit
    return hit


DECODER_KEYS = (
    [("density_dec.0.%d.weight" % i, "density_dec.0.%d.bias" % i) for i in range(4)]
    + [("fc_alpha.0.weight", "fc_alpha.0.bias")]
    + [("rgb_dec.0.%d.weight" % i, "rgb_dec.0.%d.bias" % i) for i in range(4)]
    + [("fc_rgb.0.weight", "fc_rgb.0.bias")]
)


class TwoDimPlanesModel(nn.Module):
    """models.py:118-434"""

    def __init__(self, use_viewdirs, dec_density_layers=4, dec_rgb_layers=4, dec_channels=128, skip_connect_every=None,
                 num_plane_channels=48, num_viewdir_plane_channels=None, rgb_dec_input="projections", proj_combination="sum",
                 plane_interp="bilinear", align_corners=True, viewdir_proj_combination=None, num_planes_or_rot_mats=3,
                 plane_stats=False, detach_LR_planes=False, scene_coupler=None, point_coords_noise=0, ensemble_size=1):
        self.num_density_planes = num_planes_or_rot_mats if isinstance(num_planes_or_rot_mats, int) else len(num_planes_or_rot_mats)
        super().__init__()
        self.box_coords = None
        self.use_viewdirs = use_viewdirs
        assert use_viewdirs or (viewdir_proj_combination is None and num_viewdir_plane_channels is None)
        # (models.py:291-293: a training-mode forward jitters the normalised sample positions; such a model trains through the generic kernels,
        #  which take the jitter as an input -- jitter_std / _generic_forward)
        self.point_coords_noise = point_coords_noise
        self.num_plane_channels = num_plane_channels
        if num_viewdir_plane_channels is None:
            num_viewdir_plane_channels = num_plane_channels if use_viewdirs else 0
        self.num_viewdir_plane_channels = num_viewdir_plane_channels
        self.plane_stats = plane_stats
        self.detach_LR_planes = detach_LR_planes
        self.align_corners = align_corners
        assert rgb_dec_input in ["projections", "features", "projections_features"]
        self.rgb_dec_input = rgb_dec_input
        assert proj_combination in ["sum", "concat", "avg"]
        if viewdir_proj_combination is None:
            viewdir_proj_combination = proj_combination
        assert viewdir_proj_combination in ["sum", "concat", "avg", "mult", "concat_pos"]
        if num_viewdir_plane_channels != num_plane_channels:
            assert "concat" in viewdir_proj_combination
        self.proj_combination = proj_combination
        self.viewdir_proj_combination = viewdir_proj_combination
        self.plane_interp = plane_interp
        self.skip_connect_every = skip_connect_every
        self.dec_channels = dec_channels
        self.dec_density_layers, self.dec_rgb_layers, self.ensemble_size = dec_density_layers, dec_rgb_layers, ensemble_size
        self.coord_projector = CoordProjector(self.num_density_planes,
                                              rot_mats=None if isinstance(num_planes_or_rot_mats, int) else num_planes_or_rot_mats)
        self.scene_coupler = scene_coupler

        # same module tree / state-dict keys as the reference (models.py:169-195)
        self.density_dec = nn.ModuleDict([(str(i), nn.ModuleList()) for i in range(ensemble_size)])
        in_channels = num_plane_channels * (self.num_density_planes if proj_combination == "concat" else 1)
        for i in range(ensemble_size):
            self.density_dec[str(i)].append(nn.Linear(in_channels, dec_channels))
            for layer_num in range(dec_density_layers - 1):
                if self.is_skip_layer(layer_num=layer_num):
                    self.density_dec[str(i)].append(nn.Linear(in_channels + dec_channels, dec_channels))
                else:
                    self.density_dec[str(i)].append(nn.Linear(dec_channels, dec_channels))
        self.fc_alpha = nn.ModuleDict([(str(i), nn.Linear(dec_channels, 1)) for i in range(ensemble_size)])
        if "features" in self.rgb_dec_input:
            self.fc_feat = nn.ModuleDict([(str(i), nn.Linear(dec_channels, num_plane_channels)) for i in range(ensemble_size)])
        self.rgb_dec = nn.ModuleDict([(str(i), nn.ModuleList()) for i in range(ensemble_size)])
        plane_C_mult = 0
        if proj_combination == "concat" or viewdir_proj_combination == "concat_pos":
            plane_C_mult += self.num_density_planes
        rgb_in = num_viewdir_plane_channels + num_plane_channels * plane_C_mult
        for i in range(ensemble_size):
            self.rgb_dec[str(i)].append(nn.Linear(rgb_in, dec_channels))
            for layer_num in range(dec_rgb_layers - 1):
                if self.is_skip_layer(layer_num=layer_num):
                    self.rgb_dec[str(i)].append(nn.Linear(rgb_in + dec_channels, dec_channels))
                else:
                    self.rgb_dec[str(i)].append(nn.Linear(dec_channels, dec_channels))
        self.fc_rgb = nn.ModuleDict([(str(i), nn.Linear(dec_channels, 3)) for i in range(ensemble_size)])

        self.skip_SR_ = False
        self._packed_cache = None  # (source key, packed blob)
        # arithmetic of the decoder GEMMs for the calls made through this model: 'f32' | 'bf16x3' | 'f16x2' | None = the process default
        # (capi.set_decoder_arithmetic / NVSR_DECODER_ARITHMETIC).  Passed to every kernel launch explicitly; a training forward stores
        # the mode it ran in and its backward uses that one.
        self.arithmetic = None

    # ---- reference protocol ------------------------------------------------------------------------------------------
    def is_skip_layer(self, layer_num):
        """models.py:203-207"""
        if self.skip_connect_every is None:
            return False
        return layer_num % self.skip_connect_every == 0 and layer_num > 0

    def rot_mats(self):
        return self.coord_projector.rot_mats_NON_LEARNED

    def rot_mat_backward_support(self, loaded_dict):
        """models.py:246-249"""
        if not any(["rot_mats" in k for k in loaded_dict]):
            loaded_dict.update(dict([(k, v) for k, v in self.state_dict().items() if "rot_mats" in k]))
        return loaded_dict

    def assign_SR_model(self, SR_model, SR_viewdir):
        """models.py:251-256"""
        self.SR_model = SR_model
        self.SR_model.align_corners = self.align_corners
        self.SR_model.SR_viewdir = SR_viewdir
        self.skip_SR_ = False
        assert not SR_viewdir, "Ceased supporting this option"

    def set_cur_scene_id(self, scene_id):
        if getattr(self, "cur_id", None) is not None and scene_id != self.cur_id:
            # only the current scene stays resident in the channel-last cache
            keep = scene_id if self.scene_coupler is None else None
            if keep is not None:
                clear_plane_cache(keep_scene=keep)
        self.cur_id = scene_id

    def invalidate(self):
        """Forget every derived copy of the parameters (packed decoder blobs, channel-last planes of the current scene, host copies of the box).
        The caches are keyed on (data_ptr, tensor._version); writes through `.data` (`p.data.copy_()`, `p.data = ...`) do NOT bump the
        version, so code that updates parameters that way -- checkpoint loaders, the reference's planes2cpu -- must call this."""
        self._packed_cache = None
        self._packed_bwd_cache = None
        self.__dict__.pop("_scene_consts", None)
        if getattr(self, "planes_", None) is not None:
            for k in self.planes_:
                _PLANE_CACHE.pop(k, None)
                _PLANE_CACHE.pop(k + "/SR", None)

    def train(self, mode=True):
        """nn.Module.train, and on a CHANGE of mode the derived copies are dropped: a training loop may have updated the parameters through
        something that does not bump version counters (a fused optimizer outside TrainStep); the first forward of the other mode re-derives"""
        if bool(mode) != self.training:
            self.invalidate()
        return super().train(mode)

    def skip_SR(self, skip):
        self.skip_SR_ = skip

    def planes2cpu(self):
        for p in self.planes_.values():
            p.data = p.data.to("cpu")
        self.invalidate()

    def assign_LR_planes(self, scene=None):
        """models.py:426-434 (no plane down-sampling: `should_downsample` is always False in the supported configs)"""
        for k in self.planes_:
            if scene is not None and self.scene_coupler is not None and self.scene_coupler.scene2saved[scene] not in k:
                continue
            if self.scene_coupler is not None:
                self._refuse_plane_downsampling(k, for_LR_loading=True)
            if not self.SR_model.SR_viewdir and get_plane_name(None, self.num_density_planes) in k:
                continue
            plane = self.planes_[k]
            self.SR_model.set_LR_plane(plane.detach() if self.detach_LR_planes else plane, id=k, save_interpolated=False)

    # ---- native path ---------------------------------------------------------------------------------------------------
    def is_native_geometry(self):
        """the configuration the fused MFMA kernels are compiled for: 3 + 1 planes x 48 channels, 'avg' / 'concat_pos', 4 + 4 layers x 128,
        no skip layer (every shipped YAML).  Any other geometry the reference's layer sizes admit renders through the generic kernels
        and trains through the generic kernels (csrc/generic.hip: plain layer-by-layer kernels, activations through HBM)."""
        return (self.use_viewdirs and self.num_density_planes == 3 and self.num_plane_channels == capi.PLANE_CHANNELS
                and self.num_viewdir_plane_channels == capi.PLANE_CHANNELS and self.dec_channels == capi.DEC_CHANNELS
                and self.dec_density_layers == 4 and self.dec_rgb_layers == 4 and self.ensemble_size == 1
                and self.proj_combination == "avg" and self.viewdir_proj_combination == "concat_pos"
                and self.rgb_dec_input == "projections" and self.plane_interp == "bilinear" and self.align_corners
                and not any(self.is_skip_layer(l) for l in range(3)) and not (self.point_coords_noise and self.training))

    def jitter_std(self):
        """std of the jitter a forward adds to the normalised sample positions right now: point_coords_noise * 2 / (1 + plane resolution) in
        training mode (models.py:291-293; the resolution is read off the scene id like there), 0 otherwise"""
        if not (self.point_coords_noise and self.training):
            return 0.0
        import re
        return float(self.point_coords_noise) * 2 / (1 + int(re.search(r"(?<=PlRes)(\d)+(?=_)", self.cur_id).group(0)))

    def _check_native_geometry(self):
        if not self.is_native_geometry():
            raise NotImplementedError(
                "this entry point (training / fused passes) runs on the kernels compiled for the shipped configuration (3+1 planes x 48 "
                "channels, 'avg' / 'concat_pos', 4+4 layers x 128, no skip layer); other TwoDimPlanesModel geometries go through the "
                "generic kernels (model.forward / run_one_iter_of_nerf pass by pass)")

    def generic_geometry(self):
        """[plane_channels, viewdir_channels, hidden, density_layers, rgb_layers, skip_connect_every, proj, view] for
        torch.ops.nvsr.triplane_decode_generic (struct nvsr_decoder_geometry)"""
        if not (self.use_viewdirs and 1 <= self.num_density_planes <= capi.MAX_POSITION_PLANES and self.ensemble_size == 1
                and self.rgb_dec_input == "projections" and self.plane_interp in ("bilinear", "bicubic")):
            raise NotImplementedError("the generic decoder kernels cover use_viewdirs=True, 1 .. %d position planes, ensemble_size 1, "
                                      "rgb_dec_input='projections' and bilinear / bicubic planes" % capi.MAX_POSITION_PLANES)
        return [self.num_plane_channels, self.num_viewdir_plane_channels, self.dec_channels, self.dec_density_layers, self.dec_rgb_layers,
                int(self.skip_connect_every or 0), {"sum": 0, "avg": 1, "concat": 2}[self.proj_combination],
                {"sum": 0, "avg": 1, "mult": 2, "concat": 3, "concat_pos": 4}[self.viewdir_proj_combination]]

    def decoder_parameters(self):
        """the decoder's weights and biases in state-dict order (planes and the fixed projection matrices excluded)"""
        sd = dict(self.named_parameters())
        keys = [("density_dec.0.%d.weight" % i, "density_dec.0.%d.bias" % i) for i in range(self.dec_density_layers)] + \
               [("fc_alpha.0.weight", "fc_alpha.0.bias")] + \
               [("rgb_dec.0.%d.weight" % i, "rgb_dec.0.%d.bias" % i) for i in range(self.dec_rgb_layers)] + [("fc_rgb.0.weight", "fc_rgb.0.bias")]
        return [sd[k] for pair in keys for k in pair]

    def natural_blob(self, differentiable=False):
        """Decoder parameters flattened in state-dict order (the layout nvsr_pack_decoder consumes).  differentiable=True keeps
        the autograd graph, so a gradient with respect to the blob reaches the individual parameters."""
        ps = self.decoder_parameters()
        return torch.cat([(p if differentiable else p.detach()).reshape(-1).float() for p in ps])

    def packed_decoder(self):
        self._check_native_geometry()
        params = self.decoder_parameters()
        key = tuple((p.data_ptr(), p._version) for p in params)
        if self._packed_cache is None or self._packed_cache[0] != key:
            nat = self.natural_blob()
            capi.require_cuda(nat)
            self._packed_cache = (key, torch.ops.nvsr.pack_decoder(nat))
        return self._packed_cache[1]

    def packed_decoder_bwd(self):
        """transposed layers for the backward kernels (nvsr_pack_decoder_bwd), cached like packed_decoder()"""
        self._check_native_geometry()
        params = self.decoder_parameters()
        key = tuple((p.data_ptr(), p._version) for p in params)
        cache = getattr(self, "_packed_bwd_cache", None)
        if cache is None or cache[0] != key:
            nat = self.natural_blob()
            capi.require_cuda(nat)
            cache = (key, torch.ops.nvsr.pack_decoder_bwd(nat))
            self._packed_bwd_cache = cache
        return cache[1]

    def _should_SR(self, plane_name):
        if not hasattr(self, "SR_model") or self.skip_SR_:
            return False
        if self.scene_coupler is None:
            return True
        return self.scene_coupler.should_SR(plane_name, plane_not_scene=True)

    def _refuse_plane_downsampling(self, plane_name, **kw):
        """models.py:231-238,273: with 'HR_planes' in nerf.train.what (SceneCoupler(planes_res='HR')) an LR scene samples its HR couple's planes
        DOWN-sampled.  Not mirrored (no shipped config trains HR planes): loud instead of sampling the HR plane as it is."""
        rank = getattr(self, "plane_rank", None)
        if rank:
            # models.py:223-230 (PlanesOptimizer(planes_rank_ratio=...), which train_nerf.py never passes): planes stored as two low-rank factors
            raise NotImplementedError("low-rank planes (plane_rank / planes_rank_ratio, models.py:223-230) are not implemented")
        sd = getattr(self.scene_coupler, "should_downsample", None)
        if sd is not None and sd(plane_name, **kw):
            raise NotImplementedError("plane down-sampling ('HR_planes' in nerf.train.what: models.py:231-238) is not implemented; "
                                      "the shipped configs store LR planes (what: ['LR_planes', ...])")

    def _plane_source(self, dim_num):
        """models.py:270-284 `planes()`: the NCHW tensor a projection samples from (raw or super-resolved)."""
        plane_name = get_plane_name(self.cur_id, dim_num)
        super_resolve = dim_num < self.num_density_planes and self._should_SR(plane_name)
        self._refuse_plane_downsampling(plane_name)
        if self.scene_coupler is not None:
            plane_name = self.scene_coupler.scene_with_saved_plane(plane_name, plane_not_scene=True)
        if super_resolve:
            return plane_name + "/SR", self.SR_model(plane_name)
        return plane_name, self.planes_[plane_name]

    def channel_last_plane(self, dim_num):
        name, src = self._plane_source(dim_num)
        # keyed on the source tensor OBJECT (weakly) + its version counter: a data pointer is not an identity -- the allocator hands
        # the address of a dropped super-resolved plane to the next one
        hit = _PLANE_CACHE.get(name)
        if hit is None or hit[0]() is not src or hit[1] != src._version:
            hit = _cache_plane(name, src)
        return hit[2]

    def training_rois(self, rays=None, normalized_points=None):
        """The regions of interest of a training batch (models.py:270-284: the ROI of every point chunk; the bounding box of the ray segments
        [near, far] contains all of them, and the super-resolved values do not depend on the ROI): for every position plane that is
        super-resolved in training mode, [[ymin, xmin], [ymax, xmax]] in normalised plane coordinates.  -> (plane indices, device tensor
        [n, 2, 2]) -- the arithmetic stays on the device; the caller decides when the host reads it (one copy for all planes).
        rays: packed [N,11]; or normalized_points [P,3]: the normalised (and jittered) positions of a model call."""
        names = [get_plane_name(self.cur_id, d) for d in range(self.num_density_planes)]
        dims = [d for d, name in enumerate(names) if self._should_SR(name)] if (hasattr(self, "SR_model") and self.SR_model.training) else []
        if not dims:
            return [], None
        n_ends = normalized_points
        if n_ends is None:
            # (the box and the projection matrices on the device, cached per version: a host tensor's .to(device) is a blocking copy per call)
            box_t = self.box_coords[self.cur_id + ""]
            key = (self.cur_id, str(rays.device), box_t.data_ptr(), box_t._version)
            cache = self.__dict__.get("_roi_consts")
            if cache is None or cache[0] != key:
                box = box_t.to(rays.device)
                cache = (key, box[0, :3].float(), (box[1, :3] - box[0, :3]).float())
                self.__dict__["_roi_consts"] = cache
            lo, rng = cache[1], cache[2]
            ends = torch.cat([rays[:, 0:3] + rays[:, 3:6] * rays[:, 6:7], rays[:, 0:3] + rays[:, 3:6] * rays[:, 7:8]], 0)
            n_ends = 2 * (ends - lo) / rng - 1
        rois = []
        for d in dims:
            m = self.coord_projector.rot_mats_NON_LEARNED[d].detach().float().to(n_ends.device)[:, 1:]
            grid = n_ends @ m                                         # [2N, (x, y)]
            gmin, gmax = grid.min(0)[0], grid.max(0)[0]
            rois.append(torch.stack([torch.stack([gmin[1], gmin[0]]), torch.stack([gmax[1], gmax[0]])], 0))   # rows (min, max), cols (y, x)
        return dims, torch.stack(rois, 0)

    def training_planes(self, rays, normalized_points=None):
        """The NCHW tensors a training step samples, as autograd sees them: the raw plane parameters, or -- where a plane is
        super-resolved -- the output of PlanesSR on the region of interest the batch covers (training_rois).  All super-resolved planes of the
        scene go through the SR network together (PlanesSR.forward_many: one launch per layer for all regions of interest).
        A caller that has the regions on the host already (training.TrainStep computes them ahead of the iteration on a side stream, so that
        the host never waits for the queue to drain) leaves them in `self._roi_hint = (number of rays, [[4 floats] per plane])`."""
        names = [get_plane_name(self.cur_id, d) for d in range(self.num_density_planes + 1)]
        saved = [self.scene_coupler.scene_with_saved_plane(n, plane_not_scene=True) if self.scene_coupler is not None else n for n in names]
        # models.py:273: every plane name goes through the coupler -- an HR scene coupled to an LR scene samples the LR scene's saved planes
        out = [None] * len(names)
        sr_dims = []
        for d, name in enumerate(names):
            if not (d < self.num_density_planes and self._should_SR(name)):
                out[d] = self.planes_[saved[d]]
            elif not self.SR_model.training:
                out[d] = self.SR_model(saved[d])                       # full plane, cached (models.py:277: ROI only in training)
            else:
                sr_dims.append(d)
        if sr_dims:
            hint = self.__dict__.pop("_roi_hint", None)
            if hint is not None and normalized_points is None and rays is not None and hint[0] == rays.shape[0] and len(hint[1]) == len(sr_dims):
                rois = hint[1]
            else:
                dims, dev_rois = self.training_rois(rays, normalized_points)
                assert dims == sr_dims
                rois = dev_rois.detach().reshape(len(dims), 4).cpu().tolist()          # ONE host read for all planes
            for d, plane in zip(sr_dims, self.SR_model.forward_many([(saved[d], rois[k]) for k, d in enumerate(sr_dims)])):
                out[d] = plane
        return out

    def _generic_forward(self, x, coord_noise=None):
        """forward through the generic kernels (any geometry); in training mode with gradients for the planes and the decoder.
        coord_noise [P,3]: the jitter of the normalised sample positions (models.py:291-293) when the caller has drawn it (run_one_iter_of_nerf
        draws in the reference's order of chunks and network batches); None = drawn here like the reference's forward does, one
        torch.normal call of the input's size on the CPU generator."""
        std = self.jitter_std()
        if std and coord_noise is None:
            coord_noise = torch.normal(mean=0, std=std, size=[x.shape[0], 3])
        if coord_noise is not None:
            coord_noise = capi.f32c(coord_noise.to(x.device))
        if torch.is_grad_enabled():                                    # (like any torch module: a graph whenever gradients are on)
            names = [get_plane_name(self.cur_id, d) for d in range(self.num_density_planes + 1)]
            dec = any(p.requires_grad for p in self.decoder_parameters())
            sr_on = hasattr(self, "SR_model") and not self.skip_SR_
            sr_grad = sr_on and self.SR_model.training and self.SR_model.inner_model.wants_grad(*self.SR_model.LR_planes.values())
            if dec or sr_grad or any(n in self.planes_ and self.planes_[n].requires_grad for n in names):
                if self.training:
                    np.random.randint(self.ensemble_size)             # models.py:393 (see forward())
                if sr_on:
                    # training THROUGH super-resolved planes (models.py:270-284): PlanesSR on the region the call's points cover, as part of the graph
                    box = self.box_coords[self.cur_id + ""].to(x.device)
                    n = 2 * (x[:, :3] - box[0, :3].float()) / (box[1, :3] - box[0, :3]).float() - 1
                    planes = self.training_planes(None, normalized_points=n if coord_noise is None else n + coord_noise)
                else:
                    planes = [self.planes_[n] for n in names]
                return _GenericDecodeFn.apply(self, x, self.natural_blob(differentiable=True) if dec else None, coord_noise, *planes)
        if hasattr(self, "SR_model") and not self.skip_SR_:
            names = [get_plane_name(self.cur_id, d) for d in range(self.num_density_planes)]
            names = [n for n in names if self._should_SR(n)]
            if self.scene_coupler is not None:
                names = [self.scene_coupler.scene_with_saved_plane(n, plane_not_scene=True) for n in names]
            self.SR_model.super_resolve_many(names)
        planes = [self.channel_last_plane(d) for d in range(self.num_density_planes + 1)]
        planes, consts = self.scene_args(planes=planes, check_native=False)
        nat = self.natural_blob()
        capi.require_cuda(nat)
        return torch.ops.nvsr.triplane_decode_generic(planes, consts, nat, self.generic_geometry(), x, bool(self.align_corners), coord_noise,
                                                      self.plane_interp == "bicubic")

    def scene_args(self, planes=None, check_native=True):
        """(planes, consts) of the current scene id as the torch.ops.nvsr operators take them: the channel-last planes (position planes, then
        the view-direction plane) + the host floats of struct nvsr_scene / nvsr_scene_ext (lo[5], range[5], a 3x2 projection per position
        plane: 28 for the three planes of the MFMA kernels).  planes: optional explicit channel-last planes (the training path samples
        tensors that are part of the autograd graph)."""
        if check_native:
            self._check_native_geometry()
        if planes is None:
            if hasattr(self, "SR_model") and not self.skip_SR_ and not (self.SR_model.training and torch.is_grad_enabled()):
                # evaluation: super-resolve every plane that needs it in one batched pass; _plane_source then hits the cache
                names = [get_plane_name(self.cur_id, d) for d in range(self.num_density_planes)]
                names = [n for n in names if self._should_SR(n)]
                if self.scene_coupler is not None:
                    names = [self.scene_coupler.scene_with_saved_plane(n, plane_not_scene=True) for n in names]
                self.SR_model.super_resolve_many(names)
            planes = [self.channel_last_plane(d) for d in range(self.num_density_planes + 1)]
        # box and projection matrices as host floats: read back once per version (a device-to-host copy drains the queue, and this runs
        # for every pass of every training iteration)
        box_t = self.box_coords[self.cur_id + ""]
        rots = [self.coord_projector.rot_mats_NON_LEARNED[d] for d in range(self.num_density_planes)]
        key = (self.cur_id, box_t.data_ptr(), box_t._version) + tuple((r.data_ptr(), r._version) for r in rots)
        cache = self.__dict__.get("_scene_consts")
        if cache is None or cache[0] != key:
            box = box_t.detach().double().cpu().numpy()
            lo = [np.float32(box[0, i]) for i in range(5)]
            rng = [np.float32(box[1, i] - box[0, i]) for i in range(5)]   # subtraction in double, then cast (models.py:264-265)
            proj = []
            for r in rots:
                m = r.detach().float().cpu().numpy()[:, 1:]
                proj.append([float(m[k, c]) for k in range(3) for c in range(2)])
            consts = [float(v) for v in lo] + [float(v) for v in rng] + [v for row in proj for v in row]
            cache = (key, lo, rng, proj, consts)
            self.__dict__["_scene_consts"] = cache
        return list(planes), cache[4]

    def native_scene(self, planes=None):
        """struct nvsr_scene for the current scene id (+ the tensors that must outlive the launch) for direct C-ABI calls"""
        planes, consts = self.scene_args(planes)
        return ops._scene(planes, consts), planes

    def arith(self):
        """NVSR_ARITH_* code of this model's calls (-1 = the process default)"""
        return capi.arith_code(self.arithmetic)

    # |value| limits of NVSR_ARITH_F16X2's static scales (include/nvsr.h): weights are packed as W 2^8, features held as x 2^4
    F16_WEIGHT_LIMIT, F16_FEATURE_LIMIT = 255.0, 4094.0

    def f16_operands_in_range(self, planes):
        """True if the decoder weights and the given channel-last planes fit NVSR_ARITH_F16X2's ranges.  One reduction + one host read per
        tensor VERSION (cached): evaluation renders call it, training steps (whose planes change every iteration) do not -- there an
        out-of-range operand shows as a NaN loss, never as a wrong number (csrc/render3.hip)."""
        cache = self.__dict__.setdefault("_f16_range_cache", {})
        ok = True
        # the matrices the matrix pipe multiplies: the hidden / feature layers' weights (heads and biases stay f32 vector arithmetic)
        mats = [p for n, p in self.named_parameters() if n.endswith(".weight") and (n.startswith("density_dec.") or n.startswith("rgb_dec."))]
        for t, limit in [(w, self.F16_WEIGHT_LIMIT) for w in mats] + [(p, self.F16_FEATURE_LIMIT) for p in planes]:
            key = (t.data_ptr(), t._version, tuple(t.shape))
            hit = cache.get(key)
            if hit is None:
                if len(cache) > 64:
                    cache.clear()
                hit = cache[key] = bool(torch.isfinite(t).all()) and float(t.detach().abs().max()) < limit if t.numel() else True
            ok = ok and hit
        return ok

    def render_arithmetic(self, planes, training):
        """the NVSR_ARITH_* code a render of `planes` through this model runs in: the model's / the process's choice, except that an
        EVALUATION render whose operands do not fit the f16 limbs falls back to the 3-bf16-limb arithmetic (warned once) instead of
        rendering NaN pixels.  Hidden activations beyond the range cannot be known beforehand: those still render NaN."""
        code = capi.resolve_decoder_arithmetic(self.arithmetic)
        if code == capi.ARITHMETIC["f16x2"] and not training and not self.f16_operands_in_range(planes):
            if not self.__dict__.get("_f16_warned"):
                import warnings
                warnings.warn("decoder weights or plane values beyond NVSR_ARITH_F16X2's range (|W| < 255, |feature| < 4094): rendering this "
                              "model in 'bf16x3' (set model.arithmetic to silence)")
                self.__dict__["_f16_warned"] = True
            return capi.ARITHMETIC["bf16x3"]
        return code

    def forward(self, x, coord_noise=None):
        """models.py:381-421: x [P,6] = [xyz, viewdir] -> [P,4] = [rgb, sigma] (pre-activation).  coord_noise (not in the reference's
        signature): the point_coords_noise jitter of this call, [P,3], when the caller draws it (_generic_forward)."""
        x = capi.f32c(x)
        assert x.shape[-1] == 6, "TwoDimPlanesModel expects [xyz, viewdir] rows"
        P = x.numel() // 6
        if P == 0:
            return torch.empty(list(x.shape[:-1]) + [4], dtype=torch.float32, device=x.device)
        if not self.is_native_geometry():
            return self._generic_forward(x.reshape(P, 6), coord_noise).reshape(list(x.shape[:-1]) + [4])
        assert coord_noise is None, "a jitter for a forward that does not jitter (point_coords_noise == 0 or evaluation mode)"
        if self.training and torch.is_grad_enabled() and not (hasattr(self, "SR_model") and not self.skip_SR_):
            names = [get_plane_name(self.cur_id, d) for d in range(self.num_density_planes + 1)]
            planes = [self.planes_[n] for n in names]
            dec = any(p.requires_grad for p in self.decoder_parameters())
            if dec or any(p.requires_grad for p in planes):
                # run_network's differentiable model call (train_utils.py:15-64): gradients for the planes and the decoder
                # models.py:393 `np.random.randint(len(self.density_dec))`: the reference picks an ensemble member on every training-mode
                # forward.  With ensemble_size == 1 (the only supported size) the call returns 0 WITHOUT drawing from NumPy's stream (a
                # zero-width range consumes no state: tests/test_host.py::test_single_member_ensemble_draw_leaves_numpy_stream_alone), so
                # the pixel selection of a seeded run (train_nerf.py:839) is unaffected; the call is kept for literal parity.
                np.random.randint(self.ensemble_size)
                out = _DecodePointsFn.apply(self, x.reshape(P, 6), *planes, self.natural_blob(differentiable=True) if dec else None)
                return out.reshape(list(x.shape[:-1]) + [4])
        planes, consts = self.scene_args()
        out = torch.ops.nvsr.triplane_decode(planes, consts, self.packed_decoder(), x.reshape(P, 6), self.arith())
        return out.reshape(list(x.shape[:-1]) + [4])


class _DecodePointsFn(torch.autograd.Function):
    """TwoDimPlanesModel.forward on a list of points with gradients for the planes and the decoder.  The points are handed to the
    ray-tiled training kernels as one-sample rays (origin = point, direction = 0, depth = 0): decode with published ReLU gates (+ the
    layer-input record when the decoder trains), gate-driven backward, weight-gradient contraction."""

    @staticmethod
    def forward(ctx, model, x, p0, p1, p2, pv, nat):
        P, dev = x.shape[0], x.device
        rays = torch.zeros((P, 11), dtype=torch.float32, device=dev)
        rays[:, 0:3] = x[:, 0:3]
        rays[:, 8:11] = x[:, 3:6]
        z = torch.zeros((P, 1), dtype=torch.float32, device=dev)
        planes_cl = [to_channel_last(p.detach()) for p in (p0, p1, p2, pv)]
        ctx.plane_srcs = (p0, p1, p2, pv)
        planes_cl, consts = model.scene_args(planes=planes_cl)
        arith = capi.resolve_decoder_arithmetic(model.arithmetic)      # the backward runs in the mode the gates / record were made in
        raw, gates, rec = torch.ops.nvsr.decode_rays(planes_cl, consts, model.packed_decoder(), rays, z, True, nat is not None, arith)
        ctx.model, ctx.state = model, (planes_cl, consts, rays, z, gates, rec if nat is not None else None, arith)
        return raw.reshape(P, 4)

    @staticmethod
    def backward(ctx, g_out):
        model = ctx.model
        planes_cl, consts, rays, z, gates, rec, arith = ctx.state
        P = rays.shape[0]
        need = ctx.needs_input_grad
        need_planes = [bool(n) for n in need[2:6]]
        if not any(need_planes) and rec is None:
            return (None,) * 7
        g_raw = capi.f32c(g_out).reshape(P, 1, 4)
        gplanes = torch.ops.nvsr.decode_rays_backward(planes_cl, consts, model.packed_decoder(), model.packed_decoder_bwd(), rays, z, g_raw, gates,
                                                      rec, need_planes, arith)
        gnat = torch.ops.nvsr.decoder_weight_grad(rec, P, 1, arith) if (rec is not None and need[6]) else None
        return (None, None) + tuple(from_channel_last(g, like=p_) if n else None for g, p_, n in zip(gplanes, ctx.plane_srcs, need_planes)) + (gnat,)


class _GenericDecodeFn(torch.autograd.Function):
    """TwoDimPlanesModel.forward for a decoder geometry / option set other than the shipped one, with gradients for the planes and the
    decoder (torch.ops.nvsr.triplane_decode_generic / _backward: csrc/generic.hip recomputes the forward in the backward).
    inputs: model, x, natural blob (None = the decoder does not train), jitter (None = none), then the planes of the scene."""

    @staticmethod
    def forward(ctx, model, x, nat, coord_noise, *plane_srcs):
        planes_cl = [to_channel_last(p.detach()) for p in plane_srcs]
        planes_cl, consts = model.scene_args(planes=planes_cl, check_native=False)
        natural = model.natural_blob() if nat is None else nat.detach()
        capi.require_cuda(natural)
        geometry = model.generic_geometry()
        ctx.plane_srcs = plane_srcs
        bicubic = model.plane_interp == "bicubic"
        ctx.state = (planes_cl, consts, natural, geometry, x, bool(model.align_corners), coord_noise, bicubic)
        return torch.ops.nvsr.triplane_decode_generic(planes_cl, consts, natural, geometry, x, bool(model.align_corners), coord_noise, bicubic)

    @staticmethod
    def backward(ctx, g_out):
        planes_cl, consts, natural, geometry, x, align, coord_noise, bicubic = ctx.state
        need = ctx.needs_input_grad
        need_planes = [bool(n) for n in need[4:]]
        if not any(need_planes) and not need[2]:
            return (None,) * len(need)
        g = torch.ops.nvsr.triplane_decode_generic_backward(planes_cl, consts, natural, geometry, x, capi.f32c(g_out), bool(need[2]), need_planes,
                                                            align, coord_noise, bicubic)
        return (None, None, g[0] if need[2] else None, None) + \
               tuple(from_channel_last(gp, like=p_) if n else None for gp, p_, n in zip(g[1:], ctx.plane_srcs, need_planes))


# =======================================================================================================================
# Feature-plane super-resolution: EDSR wrapped by PlanesSR (models.py:768-926)
# =======================================================================================================================
def _cfg_get(node, name, default=None):
    if isinstance(node, dict):
        return node.get(name, default)
    return getattr(node, name, default)


class _Residual_Block(nn.Module):
    """models.py:769-786 (parameters only; the arithmetic is the fused conv -> ReLU -> conv -> x0.1 + cropped identity kernels)"""

    def __init__(self, hidden_size, padding, kernel_size):
        super().__init__()
        self.margins = None if (padding or kernel_size == 1) else 2 * (kernel_size // 2)
        self.conv1 = nn.Conv2d(hidden_size, hidden_size, kernel_size, stride=1, padding=padding, bias=False)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(hidden_size, hidden_size, kernel_size, stride=1, padding=padding, bias=False)


class EDSR(nn.Module):
    """models.py:789-822.  Same constructor, `required_padding` arithmetic and state-dict keys; forward runs nvsr_edsr_forward
    (MFMA implicit-GEMM convolutions with fused ReLU / residual / PixelShuffle epilogues).

    receptive_field_bound (models.py:793-798): a layer whose 3 x 3 kernel would push the receptive field beyond the bound is a 1 x 1
    convolution.  The parameters keep the reference's shapes; the kernels run every layer as a 3 x 3 convolution, a 1 x 1 weight sitting in
    the centre of a zero kernel.  Such a layer returns the true layer's output minus a one-texel border, so the network is fed `_extra_pad`
    more texels of context per side (they meet zero weights only) and its output is cropped accordingly: same values, nine times the
    arithmetic in those layers (no shipped config sets the option)."""

    def __init__(self, in_channels, out_channels, hidden_size, n_blocks, scale_factor, padding, receptive_field_bound=np.iinfo(np.int32).max,
                 **kwargs):
        super().__init__()
        import math
        KERNEL_SIZE = 3
        self.required_padding, rf_factor = 0, 1
        self._all3_padding = 0           # required_padding of the network the kernels run (every layer 3 x 3)
        if padding != 0:
            raise NotImplementedError("the SR kernels implement the un-padded convolutions PlanesSR uses (PADDING = 0, models.py:836)")

        def kernel_size(num_layers=1):
            grow = rf_factor * num_layers * (KERNEL_SIZE // 2)
            self._all3_padding += grow
            if (1 + 2 * (self.required_padding + rf_factor * num_layers * ((KERNEL_SIZE - 1) // 2))) <= receptive_field_bound:
                self.required_padding += grow
                return KERNEL_SIZE
            return 1

        self.conv_input = nn.Conv2d(in_channels, hidden_size, kernel_size(), stride=1, padding=padding, bias=False)
        self.residual = nn.Sequential()
        for _ in range(n_blocks):
            self.residual.append(_Residual_Block(hidden_size=hidden_size, padding=padding, kernel_size=kernel_size(2)))
        self.conv_mid = nn.Conv2d(hidden_size, hidden_size, kernel_size(), stride=1, padding=padding, bias=False)
        assert math.log2(scale_factor) == int(math.log2(scale_factor)), "Supperting only scale factors that are an integer power of 2."
        upscaling_layers = []
        for _ in range(int(math.log2(scale_factor))):
            upscaling_layers += [nn.Conv2d(hidden_size, hidden_size * 4, kernel_size(), stride=1, padding=padding, bias=False), nn.PixelShuffle(2)]
            rf_factor /= 2
        self.upscale = nn.Sequential(*upscaling_layers)
        self.conv_output = nn.Conv2d(hidden_size, out_channels, kernel_size(), stride=1, padding=padding, bias=False)
        self.geometry = (in_channels, out_channels, hidden_size, n_blocks, int(math.log2(scale_factor)))
        # stand-alone forward: context the 3 x 3 stand-ins of the 1 x 1 layers consume, in input texels (rounded up; the rest is cropped from the
        # output, a whole number of output texels because every increment is a multiple of 1 / scale_factor)
        extra = self._all3_padding - self.required_padding
        self._extra_pad = int(math.ceil(extra))
        self._extra_crop = int(round((self._extra_pad - extra) * scale_factor))
        self._packed_cache = None
        self.arithmetic = None     # 'f32' | 'bf16x3' | None = the process default (capi.set_conv_arithmetic / NVSR_CONV_ARITHMETIC)

    def invalidate(self):
        """forget the packed weight blobs (needed after writes through `.data`, which do not bump a tensor's version counter)"""
        self._packed_cache = None
        self._packed_dgrad_cache = None

    def train(self, mode=True):
        """nn.Module.train; a CHANGE of mode drops the packed blobs (see TwoDimPlanesModel.train)"""
        if bool(mode) != self.training:
            self.invalidate()
        return super().train(mode)

    def arith(self):
        return capi.arith_code(self.arithmetic)

    def conv_parameters(self):
        """the convolution weights in state-dict order, as the reference shapes them (3 x 3 or 1 x 1)"""
        ws = [self.conv_input.weight]
        for blk in self.residual:
            ws += [blk.conv1.weight, blk.conv2.weight]
        ws.append(self.conv_mid.weight)
        ws += [m.weight for m in self.upscale if isinstance(m, nn.Conv2d)]
        ws.append(self.conv_output.weight)
        return ws

    def conv_weights(self):
        """the 3 x 3 kernels the library runs: a 1 x 1 weight (receptive_field_bound) in the centre of a zero kernel -- differentiable, so a
        gradient of the 3 x 3 blob reaches the 1 x 1 parameter through its centre tap"""
        return [w if w.shape[-1] == 3 else torch.nn.functional.pad(w, (1, 1, 1, 1)) for w in self.conv_parameters()]

    def packed_weights(self, arithmetic=None):
        """the fragment blob of the forward convolutions, cached on the parameters' (data_ptr, version).  arithmetic: None = every fragment region
        (a blob any arithmetic can run: evaluation, whose range fallback re-renders in bf16x3); an NVSR_ARITH_* code = only the regions that
        arithmetic reads (training: the weights change every iteration and both blobs are re-packed -- a fifth of the bytes); a cached blob with
        every region serves any request"""
        key = tuple((w.data_ptr(), w._version) for w in self.conv_parameters())
        kinds = capi.PACK_ALL_ARITHMETICS if (arithmetic is None or os.environ.get("NVSR_PACK_ALL") == "1") else int(arithmetic)
        c = self._packed_cache
        if c is None or c[0] != key or c[2] not in (kinds, capi.PACK_ALL_ARITHMETICS):
            ws = self.conv_weights()
            nat = torch.cat([w.detach().reshape(-1).float() for w in ws])
            capi.require_cuda(nat)
            c = self._packed_cache = (key, torch.ops.nvsr.pack_edsr(nat, list(self.geometry), False, kinds), kinds)
        return c[1]

    def natural_blob(self, differentiable=False):
        """conv weights flattened in state-dict order (the layout nvsr_pack_edsr consumes and nvsr_edsr_backward fills)"""
        return torch.cat([(w if differentiable else w.detach()).reshape(-1).float() for w in self.conv_weights()])

    def packed_dgrad_weights(self, arithmetic=None):
        """fragments of every layer's data gradient (flipped, transposed kernels), cached like packed_weights()"""
        key = tuple((w.data_ptr(), w._version) for w in self.conv_parameters())
        kinds = capi.PACK_ALL_ARITHMETICS if (arithmetic is None or os.environ.get("NVSR_PACK_ALL") == "1") else int(arithmetic)
        cache = getattr(self, "_packed_dgrad_cache", None)
        if cache is None or cache[0] != key or cache[2] not in (kinds, capi.PACK_ALL_ARITHMETICS):
            nat = self.natural_blob()
            capi.require_cuda(nat)
            cache = (key, torch.ops.nvsr.pack_edsr(nat, list(self.geometry), True, kinds), kinds)
            self._packed_dgrad_cache = cache
        return cache[1]

    def wants_grad(self, *inputs):
        return torch.is_grad_enabled() and (any(w.requires_grad for w in self.conv_parameters()) or
                                            any(t is not None and t.requires_grad for t in inputs))

    def forward(self, x):
        x = capi.f32c(x)
        lead = x.shape[:-3]
        assert int(np.prod(lead)) == 1, "the SR network runs one plane at a time"
        x4 = x.reshape((1,) + tuple(x.shape[-3:]))
        if self._extra_pad:                                # (1 x 1 layers run as 3 x 3: see the class comment; the padding meets zero weights only)
            x4 = torch.nn.functional.pad(x4, (self._extra_pad,) * 4)
        if self.wants_grad(x):
            # gradients for the conv weights (through the flat state-dict-order blob) and the input: torch.ops.nvsr.edsr_train
            arith = capi.resolve_conv_arithmetic(self.arithmetic)
            out, _ = torch.ops.nvsr.edsr_train(x4, self.natural_blob(differentiable=True), self.packed_weights(arith), self.packed_dgrad_weights(arith),
                                               list(self.geometry), arith)
        else:
            out = torch.ops.nvsr.edsr(x4, self.packed_weights(), list(self.geometry), self.arith())
        if self._extra_crop:
            out = out[..., self._extra_crop:-self._extra_crop, self._extra_crop:-self._extra_crop]
        return out.reshape(tuple(lead) + tuple(out.shape[-3:]))


class PlanesSR(nn.Module):
    """models.py:824-926.  `forward(plane_name | (plane_name, roi))` -> super-resolved plane [1,C,sf*R,sf*R]; NaN outside the ROI."""

    def __init__(self, model_arch, scale_factor, in_channels, out_channels, sr_config, plane_interp):
        super().__init__()
        import math
        model_cfg = _cfg_get(sr_config, "model")
        hidden_size, n_blocks = _cfg_get(model_cfg, "hidden_size"), _cfg_get(model_cfg, "n_blocks")
        input_normalization = _cfg_get(sr_config, "input_normalization", False)
        self.scale_factor = scale_factor
        self.plane_interp = plane_interp
        self.input_noise = _cfg_get(sr_config, "sr_input_noise", 0)
        self.output_noise = _cfg_get(sr_config, "sr_output_noise", 0)
        if plane_interp not in ("bilinear", "bicubic"):
            raise NotImplementedError("PlanesSR kernels implement the bilinear and bicubic residual up-sampling (config/TrainModels.yml:72,172)")
        self.inner_model = model_arch(in_channels=in_channels, out_channels=out_channels, hidden_size=hidden_size, n_blocks=n_blocks,
                                      scale_factor=scale_factor, padding=0,
                                      receptive_field_bound=_cfg_get(model_cfg, "receptive_field_bound", np.iinfo(np.int32).max))
        # models.py:840-842
        self.HR_overpadding = int(self.inner_model.required_padding * self.scale_factor)
        self.inner_model.required_padding = int(np.ceil(self.inner_model.required_padding))
        self.HR_overpadding = self.inner_model.required_padding * self.scale_factor - self.HR_overpadding
        # the same two numbers for the network the kernels run (all 3 x 3: EDSR's class comment); equal to the above without receptive_field_bound
        all3 = getattr(self.inner_model, "_all3_padding", self.inner_model.required_padding)
        self._kernel_pad = int(np.ceil(all3))
        self._kernel_over = self._kernel_pad * self.scale_factor - int(all3 * self.scale_factor)
        for m in self.modules():       # models.py:843-848
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n) / 10)
        self.clear_SR_planes(all_planes=True)
        self.align_corners = True
        self.SR_viewdir = False
        if input_normalization:
            self.normalization_params({"mean": float("nan") * torch.ones([in_channels]), "std": float("nan") * torch.ones([in_channels])})

    def normalization_params(self, norm_dict):
        self.planes_mean_NON_LEARNED = nn.Parameter(norm_dict["mean"].reshape([1, -1, 1, 1]))
        self.planes_std_NON_LEARNED = nn.Parameter(norm_dict["std"].reshape([1, -1, 1, 1]))

    def interpolate_LR(self, id):
        return torch.nn.functional.interpolate(self.LR_planes[id], scale_factor=self.scale_factor, mode=self.plane_interp,
                                               align_corners=self.align_corners)

    def set_LR_plane(self, plane, id: str, save_interpolated: bool):
        assert id not in self.LR_planes, "Plane ID already exists."
        self.LR_planes[id] = plane
        # `save_interpolated` cached the bilinear up-sampling on the CPU in the reference (models.py:874-875); the finish kernel
        # recomputes it on the fly, so there is nothing to store.

    def invalidate(self):
        """forget everything derived from the parameters / LR planes (see EDSR.invalidate): packed weights and super-resolved planes"""
        self.inner_model.invalidate()
        self.clear_SR_planes()

    def clear_SR_planes(self, all_planes=False):
        for name in list(getattr(self, "SR_planes", {})):          # ... and their channel-last copies in the renderer's plane cache
            _PLANE_CACHE.pop(name + "/SR", None)
        self.__dict__.pop("_planes_arith", None)                     # (train_utils._sr_fallback: the arithmetic the cached planes were made in)
        planes_2_clear = ["SR_planes"]
        if all_planes:
            planes_2_clear += ["LR_planes", "residual_planes"]
        for attr in planes_2_clear:
            setattr(self, attr, {})

    def super_resolve_many(self, plane_names):
        """Full-plane, no-grad super-resolution of several equally sized planes in ONE batched pass (fills the cache `forward`
        serves from).  The reference super-resolves the planes one by one on first use (models.py:270-284); the values are the
        same, the batch keeps all 256 CUs busy."""
        todo = [n for n in plane_names if n not in self.SR_planes]
        if len(todo) < 2 or (self.training and (self.input_noise > 0 or self.output_noise > 0)):
            return
        lrs = [capi.f32c(self.LR_planes[n].detach()) for n in todo]
        if len({tuple(t.shape[-3:]) for t in lrs}) != 1:
            return
        Cc, R0, R1 = lrs[0].shape[-3:]
        cin, cout, hid, nb, n_up = self.inner_model.geometry
        mean = std = None
        if hasattr(self, "planes_mean_NON_LEARNED"):
            mean, std = capi.f32c(self.planes_mean_NON_LEARNED.detach().reshape(-1)), capi.f32c(self.planes_std_NON_LEARNED.detach().reshape(-1))
        pad, over = self._kernel_pad, self._kernel_over
        if capi.lib().nvsr_planes_sr_workspace_floats(Cc, R0, R1, hid, nb, n_up, pad, None) < 0:
            return
        outs = torch.ops.nvsr.planes_sr([t.reshape(Cc, R0, R1) for t in lrs], self.inner_model.packed_weights(), list(self.inner_model.geometry),
                                        pad, over, None, mean, std, self.inner_model.arith(), bool(self.align_corners),
                                        self.plane_interp == "bicubic")
        for n, o in zip(todo, outs):
            self.SR_planes[n] = o

    def _apply_training_noise(self, out, noise_in, lr_clean):
        """The training-mode noises of models.py:896-897,920-921 on the operator's output `out` = EDSR(prep(LR + n_in)) + up(LR + n_in) (NaN
        outside the region of interest): remove up(n_in) -- the reference's residual is up(LR) -- then add
        N(0, (sr_output_noise * difference.std())^2) over the region, difference = the network's (cropped) output.  Training only, not a
        hot path: plain torch on the device, draws from the CPU generator in the reference's order (input noise first)."""
        if noise_in is not None:
            out = out - torch.nn.functional.interpolate(noise_in, scale_factor=self.scale_factor, mode=self.plane_interp, align_corners=self.align_corners)
        if self.output_noise > 0:
            inside = ~torch.isnan(out.detach()[0, 0])
            rows, cols = inside.any(1).nonzero().flatten(), inside.any(0).nonzero().flatten()
            r0, r1, c0, c1 = int(rows[0]), int(rows[-1]) + 1, int(cols[0]), int(cols[-1]) + 1
            # difference = out - up(LR) on the region (the noisy LR plane's residual was removed above)
            resid = torch.nn.functional.interpolate(lr_clean.detach().float(), scale_factor=self.scale_factor, mode=self.plane_interp,
                                                    align_corners=self.align_corners)[..., r0:r1, c0:c1]
            diff = out.detach()[..., r0:r1, c0:c1] - resid
            sd_out = float(self.output_noise * diff.std())
            n_out = torch.normal(mean=0.0, std=sd_out, size=tuple(diff.shape)).to(device=out.device, dtype=torch.float32)
            pad_n = torch.zeros_like(out.detach())
            pad_n[..., r0:r1, c0:c1] = n_out
            out = out + pad_n
        return out

    def forward_many(self, requests):
        """[(plane_name, roi | None), ...] -> the super-resolved planes, like one forward() call per request (models.py:884-926).  In training,
        when gradients are wanted and the planes are equally sized, all requests go through the network TOGETHER (ops.PlanesSRBatchFn: one
        launch per layer convolves every plane's region of interest, one weight-gradient pass per layer in the backward); anything else --
        evaluation, the training noises of models.py:896-897,920-921, more than four planes -- is one forward() per request."""
        reqs = [(r, None) if isinstance(r, str) else (r[0], r[1]) for r in requests]
        lrs = [self.LR_planes[n] for n, _ in reqs]
        net = self.inner_model
        noisy = self.training and (self.input_noise > 0 or self.output_noise > 0)
        batched = (len(reqs) >= 2 and len(reqs) <= capi.SR_BATCH_MAX and self.training and net.wants_grad(*lrs) and not noisy
                   and len({tuple(t.shape[-3:]) for t in lrs}) == 1 and all(t.dtype == torch.float32 for t in lrs)
                   and all((roi is None) == (reqs[0][1] is None) for _, roi in reqs))
        if not batched:
            return [self.forward(n if roi is None else (n, roi)) for n, roi in reqs]
        rois = None
        if reqs[0][1] is not None:
            rois = [float(v) for _, roi in reqs for v in (roi if isinstance(roi, (list, tuple)) else torch.as_tensor(roi).detach().cpu().reshape(-1).tolist())]
        mean = std = None
        if hasattr(self, "planes_mean_NON_LEARNED"):
            mean, std = capi.f32c(self.planes_mean_NON_LEARNED.detach().reshape(-1)), capi.f32c(self.planes_std_NON_LEARNED.detach().reshape(-1))
        arith = capi.resolve_conv_arithmetic(net.arithmetic)
        cfg = dict(packed=net.packed_weights(arith), packed_dgrad=net.packed_dgrad_weights(arith), geometry=list(net.geometry), pad=self._kernel_pad,
                   over=self._kernel_over, rois=rois, mean=mean, std=std, arithmetic=arith,
                   align_corners=bool(self.align_corners), bicubic=self.plane_interp == "bicubic",
                   # data-parallel training: the weight-gradient blob all-reduced bucket by bucket inside the backward (distributed.OverlappedSRGradSync.attach)
                   bucket_sync=self.__dict__.get("grad_bucket_sync"))
        return list(ops.PlanesSRBatchFn.apply(cfg, net.natural_blob(differentiable=True), *lrs))

    def forward(self, plane_name):
        if isinstance(plane_name, tuple):
            full_plane, plane_roi, plane_name = False, plane_name[1], plane_name[0]
        else:
            full_plane, plane_roi = True, None
        lr_src = lr_clean = self.LR_planes[plane_name]
        differentiable = self.training and self.inner_model.wants_grad(lr_src)
        if plane_name in self.SR_planes and not differentiable:      # (a cached plane carries no graph: never serve it to a training step)
            return self.SR_planes[plane_name]
        noisy = self.training and (self.input_noise > 0 or self.output_noise > 0)
        noise_in = None
        if noisy and self.input_noise > 0:
            # models.py:896-897: LR_plane + N(0, (sr_input_noise * LR_plane.std())^2), drawn by torch.normal from the CPU generator (the
            # reference's call has no device: a seeded run draws the same numbers here).  Only the NETWORK sees the noisy plane -- the
            # bilinear residual is taken from self.LR_planes (models.py:858-868) -- so the fused operator runs on LR + n and the
            # up-sampled noise is taken out of its (linear) residual term below.
            sd_in = float(self.input_noise * lr_src.detach().float().std())
            noise_in = torch.normal(mean=0.0, std=sd_in, size=tuple(lr_src.shape)).to(device=lr_src.device, dtype=torch.float32)
            lr_src = lr_src + noise_in
        lr = capi.f32c(lr_src.detach())
        Cc, R0, R1 = lr.shape[-3:]
        cin, cout, hid, nb, n_up = self.inner_model.geometry
        assert Cc == cin == cout
        roi = None
        if plane_roi is not None:
            roi = [float(v) for v in torch.as_tensor(plane_roi).detach().cpu().reshape(-1)]
        mean = std = None
        if hasattr(self, "planes_mean_NON_LEARNED"):
            mean, std = capi.f32c(self.planes_mean_NON_LEARNED.detach().reshape(-1)), capi.f32c(self.planes_std_NON_LEARNED.detach().reshape(-1))
        pad, over = self._kernel_pad, self._kernel_over
        geometry = list(self.inner_model.geometry)
        if differentiable:
            # training: gradients for the EDSR weights and the (non-detached) LR plane; the result is never cached
            net = self.inner_model
            arith = capi.resolve_conv_arithmetic(net.arithmetic)
            out, _ = torch.ops.nvsr.planes_sr_train(lr_src if lr_src.dtype == torch.float32 else lr_src.float(), net.natural_blob(differentiable=True),
                                                    net.packed_weights(arith), net.packed_dgrad_weights(arith), geometry, pad, over, roi, mean, std,
                                                    arith, bool(self.align_corners), self.plane_interp == "bicubic")
            return self._apply_training_noise(out, noise_in, lr_clean) if noisy else out
        out = torch.ops.nvsr.planes_sr([lr.reshape(Cc, R0, R1)], self.inner_model.packed_weights(), geometry, pad, over, roi, mean, std,
                                       self.inner_model.arith(), bool(self.align_corners), self.plane_interp == "bicubic")[0]
        if noisy:
            return self._apply_training_noise(out, noise_in, lr_clean)          # (like the reference, a noisy plane is never cached: training only)
        if full_plane:
            self.SR_planes[plane_name] = out      # kept on the GPU (the reference parks it on the CPU and re-uploads per call, :893,:925)
        return out


# =======================================================================================================================
# Positional-encoding baseline (models.py:14-108) -- interface completeness, not the tri-plane hot path
# =======================================================================================================================
class FlexibleNeRFModel(nn.Module):
    """models.py:14-108 with use_viewdirs=True, num_layers_dir=1, scalar hidden_size (what train_nerf.py:342-348 builds)."""

    def __init__(self, num_layers=4, num_layers_dir=1, dirs_hidden_width_ratio=2, hidden_size=128, skip_connect_every=4,
                 num_encoding_fn_xyz=6, num_encoding_fn_dir=4, include_input_xyz=True, include_input_dir=True, use_viewdirs=True,
                 input_dim=None, xyz_input_2_dir=False):
        super().__init__()
        if isinstance(hidden_size, list) or not use_viewdirs or num_layers_dir != 1 or dirs_hidden_width_ratio != 2 or xyz_input_2_dir \
                or input_dim is not None:
            raise NotImplementedError("only the default FlexibleNeRFModel head layout is implemented natively")
        self.skip_connect_every = skip_connect_every
        self.dim_xyz = (3 if include_input_xyz else 0) + 2 * 3 * num_encoding_fn_xyz
        self.dim_dir = (3 if include_input_dir else 0) + 2 * 3 * num_encoding_fn_dir
        self.hidden_size, self.num_layers = hidden_size, num_layers
        self.layer1 = nn.Linear(self.dim_xyz, hidden_size)
        self.layers_xyz = nn.ModuleList()
        for i in range(num_layers - 1):
            if i % self.skip_connect_every == 0 and i > 0 and i != num_layers - 1:
                self.layers_xyz.append(nn.Linear(self.dim_xyz + hidden_size, hidden_size))
            else:
                self.layers_xyz.append(nn.Linear(hidden_size, hidden_size))
        self.use_viewdirs = True
        self.layers_dir = nn.ModuleList([nn.Linear(self.dim_dir + hidden_size, hidden_size // 2)])
        self.fc_alpha = nn.Linear(hidden_size, 1)
        self.fc_rgb = nn.Linear(hidden_size // 2, 3)
        self.fc_feat = nn.Linear(hidden_size, hidden_size)
        for i, l in enumerate(self.layers_xyz):      # the reference's forward concatenates whenever i % skip == 0 and i > 0 (:90-95)
            expect = hidden_size + (self.dim_xyz if (i % skip_connect_every == 0 and i > 0) else 0)
            if l.in_features != expect:
                raise NotImplementedError("this num_layers / skip_connect_every combination is inconsistent in the reference as well")

    def forward(self, x):
        x = capi.f32c(x)
        assert x.shape[-1] == self.dim_xyz + self.dim_dir
        P = x.numel() // x.shape[-1]
        mods = [self.layer1] + list(self.layers_xyz) + [self.layers_dir[0], self.fc_alpha, self.fc_rgb, self.fc_feat]
        blob = torch.cat([t.detach().reshape(-1).float() for m in mods for t in (m.weight, m.bias)])
        capi.require_cuda(blob)
        out = torch.empty((P, 4), dtype=torch.float32, device=x.device)
        capi.call("nvsr_flexible_nerf_forward", P, capi.ptr(x), self.dim_xyz, self.dim_dir, self.hidden_size, self.num_layers,
                  self.skip_connect_every, capi.ptr(blob), capi.ptr(out), capi.stream())
        return out.reshape(list(x.shape[:-1]) + [4])
