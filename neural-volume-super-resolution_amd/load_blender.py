"""Blender-format ("NeRF synthetic") scenes: transforms_{train,val,test}.json + PNG frames -> images, camera-to-world matrices,
intrinsics and the 40-view orbit used for rendering (reference: load_blender.py:15-40 and :232-332).

Pure host I/O on the input side of the rendering path -- numpy / PIL, no cv2, imageio or libmagic.  The reference's multi-scene
`BlenderDataset` bookkeeping (scene groups, sampling probabilities, on-the-fly loading) is control plane and is not mirrored."""
import json
import os

import numpy as np
import torch

from .nerf_helpers import im_resize, imread

_SPLITS = ("train", "val", "test")


def translate_by_t_along_z(t):
    m = np.eye(4, dtype=np.float32)
    m[2, 3] = t
    return m


def rotate_by_phi_along_x(phi):
    c, s = np.cos(phi), np.sin(phi)
    m = np.eye(4, dtype=np.float32)
    m[1:3, 1:3] = [[c, -s], [s, c]]
    return m


def rotate_by_theta_along_y(theta):
    c, s = np.cos(theta), np.sin(theta)
    m = np.eye(4, dtype=np.float32)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, -s, s, c
    return m


_BLENDER_AXES = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]])


def pose_spherical(theta, phi, radius):
    """camera on a sphere of `radius` looking at the origin: azimuth theta, elevation phi (degrees); load_blender.py:34-39"""
    c2w = rotate_by_phi_along_x(phi / 180.0 * np.pi) @ translate_by_t_along_z(radius)
    c2w = rotate_by_theta_along_y(theta / 180 * np.pi) @ c2w
    return _BLENDER_AXES @ c2w


def _png_size(path):
    from PIL import Image

    with Image.open(path) as im:
        return im.size[1], im.size[0]


def load_blender_data(basedir, half_res=False, testskip=1, debug=False, downsampling_factor=1, val_downsampling_factor=None, cfg=None,
                      splits2use=("train", "val"), load_imgs=True, degradation=None):
    """-> (imgs, poses [n,4,4], render_poses [40,4,4], [H, W, focal, ds_factor] (per-image lists), i_split)

    imgs: list of float32 [H,W,3] tensors (file paths with load_imgs=False); images of a split are down-scaled by that split's factor
    (block mean), H / W / focal are the values AFTER down-scaling; only every `testskip`-th validation frame is kept."""
    assert not half_res and not debug and cfg is None, "Depricated"
    if val_downsampling_factor is None:
        val_downsampling_factor = downsampling_factor
    assert all(s in _SPLITS for s in splits2use)
    imgs, poses, H, W, focal, ds_factor, bounds = [], [], [], [], [], [], [0]
    scene = os.path.basename(os.path.normpath(basedir))
    for split in _SPLITS:
        if split in splits2use:
            with open(os.path.join(basedir, "transforms_%s.json" % split)) as fp:
                meta = json.load(fp)
            f_over_w = 0.5 / np.tan(0.5 * float(meta["camera_angle_x"]))
            factor = val_downsampling_factor if split == "val" else downsampling_factor
            for frame in meta["frames"][::testskip if split == "val" else 1]:
                path = os.path.join(basedir, frame["file_path"] + ".png")
                if load_imgs:
                    im = imread(path)
                    h, w = im.shape[:2]
                    imgs.append(torch.from_numpy(im_resize(im, scale_factor=factor, degradation=degradation,
                                                           fname="%s_%s" % (scene, os.path.basename(frame["file_path"])))))
                else:
                    h, w = _png_size(path)
                    imgs.append(path)
                H.append(h // factor)
                W.append(w // factor)
                focal.append(f_over_w * W[-1])
                ds_factor.append(factor)
                poses.append(np.asarray(frame["transform_matrix"], np.float32).reshape(4, 4))
        bounds.append(len(imgs))
    i_split = [np.arange(bounds[k], bounds[k + 1]) for k in range(len(_SPLITS))]
    poses = torch.from_numpy(np.stack(poses, 0) if poses else np.zeros((0, 4, 4), np.float32))
    render_poses = torch.stack([torch.from_numpy(pose_spherical(a, -30.0, 4.0)) for a in np.linspace(-180, 180, 40 + 1)[:-1]], 0)
    return imgs, poses, render_poses, [H, W, focal, ds_factor], i_split
