"""numpy (float64) restatement of TwoDimPlanesModel.forward for ANY decoder geometry.  TEST INFRASTRUCTURE ONLY (see oracle.py).

The C oracle (nvsr_oracle.c) covers the shipped geometry; this restates the general forward of the reference -- models.py:381-421 with
normalize_coords :261-268, cart2az_el nerf_helpers.py:492-496, CoordProjector :495-497, project_xyz / project_viewdir :289-326
(grid_sample bilinear, padding_mode='border', either align_corners), the training-mode jitter of the normalised positions :291-293, any
number of position planes (CoordProjector's frames come with the state dict), combine_pos_planes :355-361, combine_all_planes :363-379,
the layer lists of the constructor :169-195 and is_skip_layer :203-207 -- for the checks of the generic HIP path (csrc/generic.hip).
Pinned by tests/golden/g18_decoder_variants.npz and g22_model_options.npz (tests/test_oracle.py)."""
import numpy as np


def is_skip_layer(layer_num, skip_connect_every):
    """models.py:203-207"""
    return skip_connect_every is not None and layer_num % skip_connect_every == 0 and layer_num > 0


def _bilinear(plane, gx, gy, align_corners=True):
    """F.grid_sample(plane[1,C,H,W], grid (x, y) in [-1,1], bilinear, padding_mode='border') -> [P,C]
    (ATen GridSamplerKernel: unnormalise -- -1 / +1 are the corner texels' centres with align_corners, their outer edges without --,
    clip to [0, size-1], floor, the four taps with clamped indices)"""
    C, H, W = plane.shape[-3:]
    p = plane.reshape(C, H, W).astype(np.float64)
    if align_corners:
        x, y = (gx + 1.0) * 0.5 * (W - 1), (gy + 1.0) * 0.5 * (H - 1)
    else:
        x, y = ((gx + 1.0) * W - 1.0) * 0.5, ((gy + 1.0) * H - 1.0) * 0.5
    x = np.clip(x, 0.0, W - 1.0)
    y = np.clip(y, 0.0, H - 1.0)
    x0, y0 = np.floor(x), np.floor(y)
    wx, wy = x - x0, y - y0
    x0, y0 = x0.astype(np.int64), y0.astype(np.int64)
    x1, y1 = np.minimum(x0 + 1, W - 1), np.minimum(y0 + 1, H - 1)
    out = (p[:, y0, x0] * ((1 - wx) * (1 - wy)) + p[:, y0, x1] * (wx * (1 - wy)) + p[:, y1, x0] * ((1 - wx) * wy) + p[:, y1, x1] * (wx * wy))
    return out.T                                                   # [P,C]


def _cubic_coefficients(t):
    """UpSample.h get_cubic_upsample_coefficients, A = -0.75"""
    A = -0.75
    x0, x1, x2, x3 = t + 1.0, t, 1.0 - t, 2.0 - t
    return [((A * x0 - 5 * A) * x0 + 8 * A) * x0 - 4 * A, ((A + 2) * x1 - (A + 3)) * x1 * x1 + 1, ((A + 2) * x2 - (A + 3)) * x2 * x2 + 1,
            ((A * x3 - 5 * A) * x3 + 8 * A) * x3 - 4 * A]


def _bicubic(plane, gx, gy, align_corners=True):
    """F.grid_sample(mode='bicubic', padding_mode='border'): the coordinate is unnormalised and NOT clipped, the 4 x 4 taps start at floor - 1
    and each tap's index is clipped to the plane (ATen GridSampler.h get_value_bounded)"""
    C, H, W = plane.shape[-3:]
    p = plane.reshape(C, H, W).astype(np.float64)
    if align_corners:
        x, y = (gx + 1.0) * 0.5 * (W - 1), (gy + 1.0) * 0.5 * (H - 1)
    else:
        x, y = ((gx + 1.0) * W - 1.0) * 0.5, ((gy + 1.0) * H - 1.0) * 0.5
    fx, fy = np.floor(x), np.floor(y)
    cx, cy = _cubic_coefficients(x - fx), _cubic_coefficients(y - fy)
    out = 0.0
    for i in range(4):
        yi = np.clip(fy - 1 + i, 0, H - 1).astype(np.int64)
        row = 0.0
        for j in range(4):
            xi = np.clip(fx - 1 + j, 0, W - 1).astype(np.int64)
            row = row + p[:, yi, xi] * cx[j]
        out = out + row * cy[i]
    return out.T


def decode(sd, planes, box, x, use_viewdirs=True, dec_density_layers=4, dec_rgb_layers=4, skip_connect_every=None, proj_combination="sum",
           viewdir_proj_combination=None, prefix="", align_corners=True, coord_noise=None, plane_interp="bilinear", **_ignored):
    """sd: state dict (numpy arrays, reference key names); planes: the position planes then the view-direction plane, [1,C,R,R] each;
    box [2,5]; x [P,6] = [xyz, viewdir] -> [P,4].  coord_noise [P,3]: the jitter a training-mode call adds to the normalised positions."""
    assert use_viewdirs
    if viewdir_proj_combination is None:
        viewdir_proj_combination = proj_combination
    x = np.asarray(x, np.float32)
    xyz, d = x[:, :3].astype(np.float32), x[:, 3:].astype(np.float32)
    az = np.arctan2(d[:, 1], d[:, 0])
    el = np.arctan2(d[:, 2], np.sqrt(d[:, 0] ** 2 + d[:, 1] ** 2))
    x5 = np.concatenate([xyz, az[:, None], el[:, None]], 1).astype(np.float32)
    box = np.asarray(box, np.float64)
    lo, rng = box[0].astype(np.float32), (box[1] - box[0]).astype(np.float32)
    n5 = (2 * (x5 - lo) / rng - 1)
    if coord_noise is not None:
        n5[:, :3] = n5[:, :3] + np.asarray(coord_noise, np.float32)
    n5 = n5.astype(np.float64)
    sample = _bicubic if plane_interp == "bicubic" else _bilinear
    pos, n_pos = [], len(planes) - 1
    for dnum in range(n_pos):
        rot = np.asarray(sd[prefix + "coord_projector.rot_mats_NON_LEARNED.%d" % dnum]).astype(np.float32).astype(np.float64)   # (.type(float32) in forward)
        g = n5[:, :3] @ rot[:, 1:]
        pos.append(sample(np.asarray(planes[dnum]), g[:, 0], g[:, 1], align_corners))
    view = sample(np.asarray(planes[n_pos]), n5[:, 3], n5[:, 4], align_corners)

    def combine_pos(ts):
        if proj_combination == "sum":
            return np.stack(ts, 0).sum(0)
        if proj_combination == "avg":
            return np.stack(ts, 0).mean(0)
        return np.concatenate(ts, 1)

    def mlp(inp, name, nlayers, head):
        h = inp
        for l in range(nlayers):
            if is_skip_layer(l - 1, skip_connect_every):
                h = np.concatenate([h, inp], 1)
            W_, b_ = np.asarray(sd[prefix + "%s.0.%d.weight" % (name, l)], np.float64), np.asarray(sd[prefix + "%s.0.%d.bias" % (name, l)], np.float64)
            h = np.maximum(h @ W_.T + b_, 0.0)
        return h @ np.asarray(sd[prefix + head + ".0.weight"], np.float64).T + np.asarray(sd[prefix + head + ".0.bias"], np.float64)

    dens_in = combine_pos(pos)
    alpha = mlp(dens_in, "density_dec", dec_density_layers, "fc_alpha")
    v = viewdir_proj_combination
    if v == "concat_pos":
        rgb_in = np.concatenate(pos + [view], 1)
    else:
        pp = combine_pos(pos)
        shape = pp.shape
        if v != "concat" and shape[1] > view.shape[1]:
            pp = pp.reshape(shape[0], view.shape[1], -1)
            vv = view[:, :, None]
        else:
            vv = view
        if v == "sum":
            rgb_in = (pp + vv).reshape(shape)
        elif v == "avg":
            rgb_in = ((pp + vv) / 2).reshape(shape)
        elif v == "mult":
            rgb_in = (pp * (1 + vv)).reshape(shape)
        elif v == "concat":
            rgb_in = np.concatenate([pp, view], 1)
        else:
            raise ValueError(v)
    rgb = mlp(rgb_in, "rgb_dec", dec_rgb_layers, "fc_rgb")
    return np.concatenate([rgb, alpha], 1)
