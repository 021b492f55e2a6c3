"""Plane store: reader / writer of the reference's per-scene plane files.

The reference keeps every scene's planes in `<run>/planes/coarse_<scene_id>.par`, written by `PlanesOptimizer.save_params`
(models.py:640-670) through `safe_saving` (nerf_helpers.py:19-48) and read back by `load_scene` (models.py:589-610) through
`safe_loading` (nerf_helpers.py:50-67):
    torch.save({'params': nn.ParameterDict{'sc<scene>_D<d>': [1,48,R,R]}, 'opt_states': [Adam state | None]*4,
                'coords_normalization': [2,5] float64 box})
with an atomic temp-file + rename on write and a `.par` -> `.par_temp` -> `.par_bckp` fallback chain on read.
This module reads and writes that format and wires a scene into a pair of TwoDimPlanesModel's the way `load_scene` does;
the conversion to the channel-last layout the kernels sample happens lazily in `TwoDimPlanesModel.native_scene()`."""
import os

import torch
import torch.nn as nn

from .models import get_plane_name

SUFFIX = "par"


def plane_file(planes_dir, scene_id, model_name="coarse", best=False):
    """models.py `param_path`: <planes_dir>/<model_name>_<scene_id>.par[_best]"""
    return os.path.join(planes_dir, "%s_%s.%s%s" % (model_name, scene_id, SUFFIX, "_best" if best else ""))


def load_plane_file(path, map_location="cpu"):
    """safe_loading (nerf_helpers.py:50-67): try the file, then its _temp and _bckp siblings."""
    last = None
    for version in ("", "_temp", "_bckp"):
        try:
            content = torch.load(path + version if version else path, map_location=map_location, weights_only=False)
            break
        except Exception as e:   # corrupted / missing: fall through to the next copy like the reference
            last = e
            if version == "_bckp":
                raise last
    for key in ("params", "coords_normalization"):
        if key not in content:
            raise KeyError("%s: not a plane file (missing '%s')" % (path, key))
    return content


def save_plane_file(path, planes, coords_normalization, opt_states=None):
    """safe_saving (nerf_helpers.py:35-48): write <path>_temp, rotate the old file to _bckp, rename, drop the backup."""
    params = planes if isinstance(planes, nn.ParameterDict) else nn.ParameterDict({k: nn.Parameter(v.detach().cpu()) for k, v in planes.items()})
    content = {"params": params, "opt_states": opt_states if opt_states is not None else [None for _ in params],
               "coords_normalization": coords_normalization}
    tmp, bck = path + "_temp", path + "_bckp"
    torch.save(content, tmp)
    had_old = os.path.isfile(path)
    if had_old:
        os.rename(path, bck)
    os.rename(tmp, path)
    if had_old:
        os.remove(bck)


def load_scene(models, planes_dir, scene_id, device="cuda", best=False, model_name="coarse"):
    """What PlanesOptimizer.load_scene does for the hot path (models.py:589-610): read the scene's planes, put them on the
    device, assign the SAME ParameterDict and box to every model, reset the SR caches.  Returns the loaded dict."""
    content = load_plane_file(plane_file(planes_dir, scene_id, model_name, best))
    planes = nn.ParameterDict({k: nn.Parameter(v.detach().to(device)) for k, v in content["params"].items()})
    expect = [get_plane_name(scene_id, d) for d in range(4)]
    missing = [n for n in expect if n not in planes]
    if missing:
        raise KeyError("plane file of scene %s lacks %s" % (scene_id, missing))
    box = torch.as_tensor(content["coords_normalization"], dtype=torch.float64)
    for m in models:
        m.planes_ = planes
        m.box_coords = {scene_id: box}
        m.set_cur_scene_id(scene_id)
        if hasattr(m, "SR_model"):
            m.SR_model.clear_SR_planes(all_planes=True)
            m.assign_LR_planes()
    return content
