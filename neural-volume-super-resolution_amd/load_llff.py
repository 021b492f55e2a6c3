"""Forward-facing / 360-degree real scenes in the LLFF layout: poses_bounds.npy + images[_<f>]/ -> images, recentred camera-to-world
matrices with [H, W, focal] in their fifth column, depth bounds, a render path and the hold-out view (reference: load_llff.py:70-141
and :143-360).  Host I/O and pose algebra only, numpy / PIL; no imageio, no ImageMagick `_minify`."""
import os

import numpy as np
import torch

from .nerf_helpers import calc_resize_crop_margins, im_resize

_EXT = ("JPG", "jpg", "png")


def _image_files(d):
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(_EXT)]


def _read_rgb(path):
    from PIL import Image

    with Image.open(path) as im:
        return np.asarray(im.convert("RGB") if im.mode != "RGB" else im)[..., :3] / 255.0


def _load_data(basedir, factor=None, base_factor=1, max_factor=1, width=None, height=None, load_imgs=True, min_eval_frames=None):
    """-> (poses [3,5,n], bds [2,n], imgs [H,W,3,n] | file list, (base_factor, crop margins))      load_llff.py:70-141"""
    arr = np.load(os.path.join(basedir, "poses_bounds.npy"))
    repeat = None
    if min_eval_frames is not None:          # densify the camera path by linear interpolation between the captured frames
        from scipy.interpolate import interp1d

        n = len(arr)
        min_eval_frames = int(np.ceil(min_eval_frames / (n - 1)) * (n - 1) + 1)
        repeat = (min_eval_frames - 1) // (n - 1)
        dense = interp1d(np.arange(n), arr, axis=0)(np.linspace(start=0, stop=n - 1, num=min_eval_frames))
        dense[::repeat, :] = arr
        arr = dense
    poses = arr[:, :-2].reshape([-1, 3, 5]).transpose([1, 2, 0])
    bds = arr[:, -2:].transpose([1, 0])

    def subdir(f):
        return "images" + ("_%d" % f if f > 1 else "")

    while not os.path.isdir(os.path.join(basedir, subdir(base_factor))):      # the largest pre-scaled copy not finer than asked for
        assert base_factor >= 1
        base_factor //= 2
    assert factor % base_factor == 0
    files = _image_files(os.path.join(basedir, subdir(base_factor)))
    if repeat is not None:
        files = [g for f in files for g in [f] + (repeat - 1) * [None]][:-repeat + 1]
    if poses.shape[-1] != len(files):
        print("Mismatch between imgs {} and poses {} !!!!".format(len(files), poses.shape[-1]))
        return
    rel = factor // base_factor
    sh = np.array(_read_rgb(files[0]).shape)
    marg = calc_resize_crop_margins(sh, max_factor // base_factor)
    if marg is not None:
        sh[:2] -= 2 * marg
    poses[:2, 4, :] = np.array([sh[0] // rel, sh[1] // rel]).reshape([2, 1])
    poses[2, 4, :] = poses[2, 4, :] * 1.0 / factor
    if not load_imgs:
        return poses, bds, files, (base_factor, marg)
    imgs = [_read_rgb(f) for f in files]
    if marg is not None:
        imgs = [im[marg[0]: im.shape[0] - marg[0], marg[1]: im.shape[1] - marg[1], :] for im in imgs]
    if rel != 1:
        imgs = [im_resize(im, scale_factor=rel) for im in imgs]
    return poses, bds, np.stack(imgs, -1), (base_factor, marg)


def normalize(x):
    return x / np.linalg.norm(x)


def viewmatrix(z, up, pos):
    """[3,4] camera frame with its third axis along z and its second axis as close to `up` as orthogonality allows"""
    az = normalize(z)
    ax = normalize(np.cross(up, az))
    return np.stack([ax, normalize(np.cross(az, ax)), az, pos], 1)


def ptstocam(pts, c2w):
    return np.matmul(c2w[:3, :3].T, (pts - c2w[:3, 3])[..., np.newaxis])[..., 0]


def poses_avg(poses):
    """mean camera: centre = mean position, axes from the summed view and up directions; fifth column = the first pose's [H,W,f]"""
    frame = viewmatrix(normalize(poses[:, :3, 2].sum(0)), poses[:, :3, 1].sum(0), poses[:, :3, 3].mean(0))
    return np.concatenate([frame, poses[0, :3, -1:]], 1)


def render_path_spiral(c2w, up, rads, focal, zdelta, zrate, rots, N):
    rads = np.array(list(rads) + [1.0])
    look_at = np.dot(c2w[:3, :4], np.array([0, 0, -focal, 1.0]))
    out = []
    for theta in np.linspace(0.0, 2.0 * np.pi * rots, N + 1)[:-1]:
        c = np.dot(c2w[:3, :4], np.array([np.cos(theta), -np.sin(theta), -np.sin(theta * zrate), 1.0]) * rads)
        out.append(np.concatenate([viewmatrix(normalize(c - look_at), up, c), c2w[:, 4:5]], 1))
    return out


def _to44(p):
    return np.concatenate([p, np.broadcast_to(np.array([0, 0, 0, 1.0]), (p.shape[0], 1, 4))], 1)


def recenter_poses(poses):
    """express every pose in the frame of the mean camera (load_llff.py:189-201)"""
    out = poses + 0
    mean44 = _to44(poses_avg(poses)[None, :3, :4])[0]
    out[:, :3, :4] = (np.linalg.inv(mean44) @ _to44(poses[:, :3, :4]))[:, :3, :4]
    return out


def spherify_poses(poses, bds):
    """360-degree captures (load_llff.py:204-279): centre = the point closest to all optical axes, up = mean camera offset, scale to
    unit mean radius; render path = 120 views on the circle of that sphere at the cameras' mean height."""
    d, o = poses[:, :3, 2:3], poses[:, :3, 3:4]
    A = np.eye(3) - d * np.transpose(d, [0, 2, 1])
    center = np.squeeze(-np.linalg.inv((np.transpose(A, [0, 2, 1]) @ A).mean(0)) @ (-A @ o).mean(0))
    v0 = normalize((poses[:, :3, 3] - center).mean(0))
    v1 = normalize(np.cross([0.1, 0.2, 0.3], v0))
    frame = np.stack([v1, normalize(np.cross(v0, v1)), v0, center], 1)
    reset = np.linalg.inv(_to44(frame[None])) @ _to44(poses[:, :3, :4])
    rad = np.sqrt(np.mean(np.sum(np.square(reset[:, :3, 3]), -1)))
    sc = 1.0 / rad
    reset[:, :3, 3] *= sc
    bds *= sc
    rad *= sc
    zh = np.mean(reset[:, :3, 3], 0)[2]
    rc = np.sqrt(rad ** 2 - zh ** 2)
    ring = []
    for th in np.linspace(0.0, 2.0 * np.pi, 120):
        origin = np.array([rc * np.cos(th), rc * np.sin(th), zh])
        az = normalize(origin)
        ax = normalize(np.cross(az, np.array([0, 0, -1.0])))
        ring.append(np.stack([ax, normalize(np.cross(az, ax)), az, origin], 1))
    ring = np.stack(ring, 0)
    hwf = poses[0, :3, -1:]
    ring = np.concatenate([ring, np.broadcast_to(hwf, ring[:, :3, -1:].shape)], -1)
    reset = np.concatenate([reset[:, :3, :4], np.broadcast_to(hwf, reset[:, :3, -1:].shape)], -1)
    return reset, ring, bds


def _views_first(poses_35n, bds_2n, imgs):
    """LLFF stores poses [3,5,n] with rotation columns [down, right, back]; the renderer wants [right, up, back] and the view index first
    (load_llff.py:291-297).  -> poses [n,3,5], bds [n,2], imgs [n,H,W,3] (or the untouched file list), all float32."""
    rot_fixed = np.concatenate([poses_35n[:, 1:2, :], -poses_35n[:, 0:1, :], poses_35n[:, 2:, :]], 1)
    front = lambda a: np.moveaxis(a, -1, 0).astype(np.float32)
    return front(rot_fixed), front(bds_2n), (front(imgs) if isinstance(imgs, np.ndarray) else imgs)


def _normalise_depth_scale(poses, bds, bd_factor):
    """scale the scene so that the nearest depth bound sits at 1 / bd_factor (load_llff.py:299-302); in place"""
    if bd_factor is not None:
        s = 1.0 / (bds.min() * bd_factor)
        poses[:, :3, 3] *= s
        bds *= s
    return poses, bds


def _forward_facing_path(poses, bds, flat):
    """the spiral fly-through of a forward-facing capture (load_llff.py:310-341): around the mean camera, radii = 90th percentile of the
    camera offsets, looking at a depth between the bounds (harmonic mix, 3/4 towards the far bound); `flat` (path_zflat): half as many
    views on one planar turn, pushed slightly towards the scene"""
    centre = poses_avg(poses)
    up = normalize(poses[:, :3, 1].sum(0))
    near, far = bds.min() * 0.9, bds.max() * 5.0
    w_far = 0.75
    focus_depth = 1.0 / ((1.0 - w_far) / near + w_far / far)
    radii = np.percentile(np.abs(poses[:, :3, 3]), 90, 0)
    views, turns = 120, 2
    if flat:
        centre[:3, 3] = centre[:3, 3] + (-near * 0.1) * centre[:3, 2]
        radii[2] = 0.0
        views, turns = views // 2, 1
    return render_path_spiral(centre, up, radii, focus_depth, near * 0.2, zrate=0.5, rots=turns, N=views)


def _most_central_view(poses):
    """index of the camera closest to the mean camera position: the reference's hold-out view (load_llff.py:349-352)"""
    offsets = poses_avg(poses)[:3, 3] - poses[:, :3, 3]
    return np.argmin((offsets * offsets).sum(-1))


def load_llff_data(basedir, factor=8, base_factor=1, max_factor=1, recenter=True, bd_factor=0.75, spherify=False, path_zflat=False,
                   load_imgs=True, min_eval_frames=None):
    """load_llff.py:282-359 -> (images [n,H,W,3] float32 tensor | file list, poses [n,3,5] tensor, bds [n,2], render_poses [m,3,5], i_test,
    (base_factor, margins))"""
    raw_poses, raw_bds, raw_imgs, load_params = _load_data(basedir, factor=factor, base_factor=base_factor, max_factor=max_factor,
                                                           load_imgs=load_imgs, min_eval_frames=min_eval_frames)
    poses, bds, imgs = _views_first(raw_poses, raw_bds, raw_imgs if load_imgs else None)
    poses, bds = _normalise_depth_scale(poses, bds, bd_factor)
    if recenter:
        poses = recenter_poses(poses)
    if spherify:
        poses, path, bds = spherify_poses(poses, bds)
    else:
        path = _forward_facing_path(poses, bds, flat=path_zflat)
    held_out = _most_central_view(poses)
    images = torch.from_numpy(imgs) if load_imgs else raw_imgs
    return images, torch.from_numpy(poses.astype(np.float32)), bds, np.array(path).astype(np.float32), held_out, load_params
