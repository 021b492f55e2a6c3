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
import pickle
import re

import torch
import torch.nn as nn

from .models import get_plane_name

SUFFIX = "par"


# ---- the reference's file protocol (nerf_helpers.py:19-67) -------------------------------------------------------------------------
def _versioned(file_name, suffix, tag):
    """<name>.<suffix> -> <name>.<suffix><tag>  (tag = '_temp' | '_bckp' | '_best')"""
    return file_name.replace(".%s" % suffix, ".%s%s" % (suffix, tag))


def safe_saving(file_name, content, suffix, best=False, run_time_signature=0):
    """nerf_helpers.py:19-48.  Write <file>_temp, rotate an existing file to <file>_bckp, rename the new one into place, drop the backup:
    at every instant one complete copy exists under one of the three names safe_loading tries.  `run_time_signature`: the reference
    keeps the start time of the newest run in <run folder>/time_sig.txt and an older run that finds a newer signature there exits;
    the same rule applies here (raises SystemExit like the reference's sys.exit)."""
    if run_time_signature:
        sig_file = os.path.join(os.path.dirname(file_name.replace("/planes/", "/")), "time_sig.txt")
        saved = None
        if os.path.exists(sig_file):
            with open(sig_file) as f:
                saved = float(f.read())
        if saved is not None and saved > run_time_signature:
            raise SystemExit("Exiting run %f since a newer run %f has started." % (run_time_signature, saved))
        if saved is None or saved < run_time_signature:
            with open(sig_file, "w") as f:
                f.write(str(run_time_signature))
    if best:
        file_name = _versioned(file_name, suffix, "_best")
    tmp, bck = _versioned(file_name, suffix, "_temp"), _versioned(file_name, suffix, "_bckp")
    if suffix == "pkl":
        with open(tmp, "wb") as f:
            pickle.dump(content, f)
    else:
        torch.save(content, tmp)
    had_old = os.path.isfile(file_name)
    if had_old:
        os.rename(file_name, bck)
    os.rename(tmp, file_name)
    if had_old:
        os.remove(bck)


def safe_loading(file_name, suffix, best=False, map_location="cpu"):
    """nerf_helpers.py:50-67: the file, else its _temp copy, else its _bckp copy (whatever an interrupted safe_saving left behind).
    The files hold pickled containers (an nn.ParameterDict in plane files), hence weights_only=False -- load only files you trust,
    exactly as with the reference's torch.load."""
    if best:
        file_name = _versioned(file_name, suffix, "_best")
    last = None
    for tag in ("", "_temp", "_bckp"):
        path = _versioned(file_name, suffix, tag) if tag else file_name
        try:
            if suffix == "pkl":
                with open(path, "rb") as f:
                    return pickle.load(f)
            return torch.load(path, map_location=map_location, weights_only=False)
        except Exception as e:      # corrupted / missing: fall through to the next copy like the reference
            last = e
    raise last


def find_latest_checkpoint(ckpt_path, sr, find_best=False):
    """train_utils.py:333-345: in a run folder, the decoder (`checkpoint<iter>.ckpt`) or SR (`SR_checkpoint<iter>.ckpt`) checkpoint with
    the highest iteration number, or -- find_best -- the `..._best` copy; None when ckpt_path is not a folder."""
    if not os.path.isdir(ckpt_path):
        return None
    stem = "SR_checkpoint" if sr else "checkpoint"
    names = os.listdir(ckpt_path)
    if find_best:
        hits = [f for f in names if re.search(r"^%s(\d)*\.ckpt_best" % stem, f)]
        return os.path.join(ckpt_path, hits[0])            # (IndexError when there is none, like the reference)
    numbered = [(int(m.group(1)), f) for f in names for m in [re.search(r"^%s(\d+)\.ckpt$" % stem, f)] if m]
    return os.path.join(ckpt_path, max(numbered)[1])


def load_decoder_checkpoint(path, model_coarse, model_fine=None, map_location="cpu"):
    """What train_nerf.py:534-548 does with a decoder checkpoint: {model_coarse_state_dict, model_fine_state_dict[, optimizer]} saved
    without the planes / SR model (and, for the fine model, without the shared projection matrices, train_nerf.py:1001-1005) ->
    load_state_dict(strict=False), the fine model's rot_mats filled in from its own (rot_mat_backward_support).  Returns the dict."""
    ck = safe_loading(path, suffix="ckpt", map_location=map_location)
    model_coarse.load_state_dict(ck["model_coarse_state_dict"], strict=False)
    if model_fine is not None and "model_fine_state_dict" in ck:
        model_fine.load_state_dict(model_fine.rot_mat_backward_support(dict(ck["model_fine_state_dict"])), strict=False)
    for m in (model_coarse, model_fine):
        if m is not None and hasattr(m, "invalidate"):
            m.invalidate()
    return ck


def load_sr_checkpoint(path, sr_model, map_location="cpu"):
    """train_nerf.py:503: SR_checkpoint*.ckpt = {SR_model, SR_optimizer} -> PlanesSR.load_state_dict"""
    ck = safe_loading(path, suffix="ckpt", map_location=map_location)
    sr_model.load_state_dict(ck["SR_model"])
    sr_model.invalidate()
    return ck


def plane_file(planes_dir, scene_id, model_name="coarse", best=False):
    """models.py `param_path`: <planes_dir>/<model_name>_<scene_id>.par[_best]"""
    return os.path.join(planes_dir, "%s_%s.%s%s" % (model_name, scene_id, SUFFIX, "_best" if best else ""))


def load_plane_file(path, map_location="cpu"):
    """a plane file through safe_loading (the file, then its _temp and _bckp siblings); checks that it is one"""
    content = safe_loading(path, SUFFIX, map_location=map_location)
    for key in ("params", "coords_normalization"):
        if key not in content:
            raise KeyError("%s: not a plane file (missing '%s')" % (path, key))
    return content


def save_plane_file(path, planes, coords_normalization, opt_states=None):
    """safe_saving (nerf_helpers.py:35-48): write <path>_temp, rotate the old file to _bckp, rename, drop the backup."""
    params = planes if isinstance(planes, nn.ParameterDict) else nn.ParameterDict({k: nn.Parameter(v.detach().cpu()) for k, v in planes.items()})
    content = {"params": params, "opt_states": opt_states if opt_states is not None else [None for _ in params],
               "coords_normalization": coords_normalization}
    safe_saving(path, content, SUFFIX)


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
        m.invalidate()                   # new parameter objects: drop the channel-last copies / host constants of the old ones
        if hasattr(m, "SR_model"):
            m.SR_model.clear_SR_planes(all_planes=True)
            m.assign_LR_planes()
    return content
