"""Alpha compositing -- mirror of the reference's volume_rendering_utils.py (a7)."""
import torch

from . import capi
from . import ops  # noqa: F401  (registers torch.ops.nvsr.*)


def volume_render_radiance_field(radiance_field, depth_values, ray_directions, radiance_field_noise_std=0.0,
                                 white_background=False, mip_nerf=False, noise=None):
    """volume_rendering_utils.py:6-51 -> (rgb_map, disp_map, acc_map, weights, depth_map).

    `noise` (extension): explicit density noise [..., S], already multiplied by the std.  Without it, a positive
    radiance_field_noise_std draws torch.randn on the CPU generator exactly like the reference (:32-33)."""
    mip = bool(mip_nerf)     # the Mip-NeRF baseline's intervals: depth_values [..., S+1] edges, radiance_field [..., S, 4] (:19-26,:41-42)
    raw, z, rd = capi.f32c(radiance_field), capi.f32c(depth_values), capi.f32c(ray_directions)
    lead, S = z.shape[:-1], z.shape[-1] - (1 if mip else 0)
    assert raw.shape[-2] == S, "radiance_field must hold one sample per depth (mip_nerf: per interval between the S+1 depths)"
    N = z.numel() // z.shape[-1]
    if noise is None and radiance_field_noise_std > 0.0:
        noise = (torch.randn(raw[..., 3].shape) * radiance_field_noise_std).to(raw)
    if noise is not None:
        noise = capi.f32c(noise)
    # torch.ops.nvsr.composite: differentiable in the radiance field through rgb_map, disp_map, acc_map and depth_map like the reference's
    # (the per-sample weights are returned without a gradient path)
    rgb, disp, acc, weights, depth = torch.ops.nvsr.composite(raw.reshape(N, S, 4), z.reshape(N, z.shape[-1]), rd.reshape(N, 3),
                                                              None if noise is None else noise.reshape(N, S), bool(white_background), mip)
    lead = list(lead)
    return rgb.reshape(lead + [3]), disp.reshape(lead), acc.reshape(lead), weights.reshape(lead + [S]), depth.reshape(lead)
