"""Ray generation and sampling helpers -- mirror of the reference's nerf_helpers.py hot subset (a1, a2, a8, a12).
Same names, argument order and return shapes; the arithmetic runs in csrc/aux.hip."""
import math
from typing import Optional

import torch

from . import capi


def get_focal(data, dim: str):
    """nerf_helpers.py:432-437"""
    assert dim in ["H", "W"]
    if isinstance(data, list):
        return data[1] if dim == "H" else data[0]
    return data


def get_ray_bundle(height: int, width: int, focal_length, tform_cam2world: torch.Tensor, padding_size: int = 0,
                   downsampling_offset: float = 0):
    """nerf_helpers.py:507-549 -> (ray_origins, ray_directions), each [H+2p, W+2p, 3]; directions are not normalised."""
    c2w = capi.f32c(tform_cam2world)
    Hp, Wp = height + 2 * padding_size, width + 2 * padding_size
    ro = torch.empty((Hp, Wp, 3), dtype=torch.float32, device=c2w.device)
    rd = torch.empty_like(ro)
    capi.call("nvsr_get_ray_bundle", height, width, float(get_focal(focal_length, "H")), float(get_focal(focal_length, "W")),
              capi.ptr(c2w), padding_size, float(downsampling_offset), capi.ptr(ro), capi.ptr(rd), capi.stream())
    return ro, rd


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """nerf_helpers.py:578-605"""
    ro, rd = capi.f32c(rays_o), capi.f32c(rays_d)
    o, d = torch.empty_like(ro), torch.empty_like(rd)
    if ro.numel() == 0:
        return o, d
    capi.call("nvsr_ndc_rays", H, W, float(focal), float(near), ro.numel() // 3, capi.ptr(ro), capi.ptr(rd), capi.ptr(o),
              capi.ptr(d), capi.stream())
    return o, d


def sample_pdf_2(bins, weights, num_samples, det=False, u=None):
    """nerf_helpers.py:668-702.  `u` (extension) supplies the uniform draws explicitly; otherwise det=False draws them on the
    CPU generator like the reference (:683) so that seeded runs consume the same random stream."""
    bins, weights = capi.f32c(bins), capi.f32c(weights)
    lead = bins.shape[:-1]
    nb = bins.shape[-1]
    assert weights.shape[-1] == nb - 1, "weights must have one entry fewer than bins"
    N = bins.numel() // nb
    if u is None and not det:
        u = torch.rand(list(lead) + [num_samples]).to(weights)
    if u is not None:
        u = capi.f32c(u)
    out = torch.empty(list(lead) + [num_samples], dtype=torch.float32, device=bins.device)
    capi.call("nvsr_sample_pdf", N, nb, num_samples, capi.ptr(bins), capi.ptr(weights), capi.ptr(u), capi.ptr(out), capi.stream())
    return out


sample_pdf = sample_pdf_2  # the renderer imports sample_pdf_2 under this name (train_utils.py:4)


def sort_depths(z):
    """values of torch.sort(z, dim=-1) (train_utils.py:155), rows of up to 512 entries"""
    z = capi.f32c(z)
    out = torch.empty_like(z)
    capi.call("nvsr_sort_rows", z.numel() // z.shape[-1], z.shape[-1], capi.ptr(z), capi.ptr(out), capi.stream())
    return out


def positional_encoding(tensor, num_encoding_functions=6, include_input=True) -> torch.Tensor:
    """nerf_helpers.py:552-575: [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)] along the last dim"""
    x = capi.f32c(tensor)
    D = x.shape[-1]
    P = x.numel() // D
    width = (D if include_input else 0) + 2 * D * num_encoding_functions
    out = torch.empty(list(x.shape[:-1]) + [width], dtype=torch.float32, device=x.device)
    capi.call("nvsr_positional_encoding", P, D, capi.ptr(x), num_encoding_functions, int(bool(include_input)), capi.ptr(out), capi.stream())
    return out


def cumprod_exclusive(tensor: torch.Tensor) -> torch.Tensor:
    """nerf_helpers.py:409-430: exclusive cumulative product along the last dim (for callers outside the fused path; the render kernels carry
    the running product in a register).  A registered operator (torch.ops.nvsr.cumprod_exclusive) with autograd, like the reference's
    differentiable torch helper: the gradient comes from nvsr_cumprod_exclusive_backward."""
    from . import ops  # noqa: F401  (registers torch.ops.nvsr.*)

    return torch.ops.nvsr.cumprod_exclusive(capi.f32c(tensor))


def get_minibatches(inputs: torch.Tensor, chunksize: Optional[int] = 1024 * 8):
    """nerf_helpers.py:277-287 (spatial_margin variant is unused by the hot path)"""
    return [inputs[i:i + chunksize] for i in range(0, inputs.shape[0], chunksize)]


def meshgrid_xy(tensor1: torch.Tensor, tensor2: torch.Tensor):
    """nerf_helpers.py:396-406"""
    ii, jj = torch.meshgrid(tensor1, tensor2, indexing="ij")
    return ii.transpose(-1, -2), jj.transpose(-1, -2)


def mse2psnr(mse):
    """nerf_helpers.py:265-269"""
    if mse == 0:
        mse = 1e-5
    return -10.0 * math.log10(mse)


def img2mse(img_src, img_tgt):
    return torch.nn.functional.mse_loss(img_src, img_tgt)


class null_with:
    """nerf_helpers.py (context manager used as model.optional_no_grad)"""

    def __enter__(self):
        pass

    def __exit__(self, a, b, c):
        pass


# ---- image files (the data format on the input side of the path) ---------------------------------------------
def imread(path, with_alpha=False, **_ignored):
    """nerf_helpers.py:256-260: 8-bit image file -> float32 [H,W,C] in [0,1]; RGBA is flattened to RGB with pixels of zero alpha
    blacked out (the reference reads with imageio; PIL decodes the same 8-bit samples, gamma chunks ignored)."""
    import numpy as np
    from PIL import Image

    with Image.open(path) as im:
        if im.mode not in ("RGB", "RGBA"):
            im = im.convert("RGBA" if "A" in im.getbands() else "RGB")
        a = np.asarray(im)
    if not with_alpha and a.shape[2] > 3:
        a = a[..., :3] * (a[..., 3:] > 0)
    return (a / 255.0).astype(np.float32)


def calc_resize_crop_margins(im_shape, ds_factor):
    """nerf_helpers.py:312-322: margins (rows, cols) to crop on every side so that both sides become multiples of ds_factor.  None only
    when EVERY entry of im_shape divides -- the reference tests the channel count too, so for an [H,W,3] shape and an even factor the
    result is [0, 0] rather than None; callers see that value (load_llff_data returns it), so it is kept."""
    import numpy as np

    if not any(v % ds_factor for v in im_shape):
        return None
    marg = np.zeros([2], np.int32)
    for d in (0, 1):
        while (im_shape[d] - 2 * marg[d]) % ds_factor:
            marg[d] += 1
    return marg


def im_resize(image, scale_factor, degradation=None, fname=None):
    """nerf_helpers.py:294-310 without the optional degradations: down-scaling by an integer factor with cv2.INTER_AREA, which for an
    integer factor is the mean of every (factor x factor) block."""
    if degradation is not None:
        raise NotImplementedError("blur / noise degradations of the LR images are dataset preparation, outside the rendering path")
    f = int(scale_factor)
    assert all(v % f == 0 for v in image.shape[:2]), "Currently not supporting downscaling to an ambiguous size."
    if f == 1:
        return image
    h, w = image.shape[0] // f, image.shape[1] // f
    return image.reshape(h, f, w, f, -1).mean(axis=(1, 3)).astype(image.dtype).reshape((h, w) + image.shape[2:])


def safe_saving(file_name, content, suffix, best=False, run_time_signature=0):
    """nerf_helpers.py:19-48 (implemented in plane_store.py)"""
    from .plane_store import safe_saving as impl
    return impl(file_name, content, suffix, best, run_time_signature)


def safe_loading(file_name, suffix, best=False):
    """nerf_helpers.py:50-67 (implemented in plane_store.py)"""
    from .plane_store import safe_loading as impl
    return impl(file_name, suffix, best)
