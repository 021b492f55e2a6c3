"""Alpha compositing -- mirror of the reference's volume_rendering_utils.py (a7)."""
import torch

from . import capi


class _CompositeFn(torch.autograd.Function):
    """volume_render_radiance_field with a gradient for the radiance field (rgb_map and acc_map are differentiable; disp, weights and
    depth are returned without a graph -- the reference's losses use rgb only, train_nerf.py:884-891)."""

    @staticmethod
    def forward(ctx, raw, z, rd, noise, white, mip=False):
        lead, S = z.shape[:-1], z.shape[-1] - (1 if mip else 0)          # mip: z holds the S + 1 interval edges
        N = z.numel() // z.shape[-1]
        dev = raw.device
        rgb = torch.empty(list(lead) + [3], dtype=torch.float32, device=dev)
        disp, acc, depth = (torch.empty(list(lead), dtype=torch.float32, device=dev) for _ in range(3))
        weights = torch.empty(list(lead) + [S], dtype=torch.float32, device=dev)
        if N:
            capi.call("nvsr_composite_mip" if mip else "nvsr_composite", N, S, capi.ptr(raw), capi.ptr(z), capi.ptr(rd), capi.ptr(noise), white,
                      capi.ptr(rgb), capi.ptr(disp), capi.ptr(acc), capi.ptr(weights), capi.ptr(depth), capi.stream())
        ctx.save_for_backward(raw, z, rd)
        ctx.noise, ctx.white, ctx.mip = noise, white, mip
        ctx.mark_non_differentiable(disp, weights, depth)
        return rgb, disp, acc, weights, depth

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_w, g_depth):
        raw, z, rd = ctx.saved_tensors
        S = z.shape[-1] - (1 if ctx.mip else 0)
        N = z.numel() // z.shape[-1]
        g_raw = torch.zeros_like(raw)
        if N and S <= 512:
            g_rgb = torch.zeros(list(z.shape[:-1]) + [3], dtype=torch.float32, device=raw.device) if g_rgb is None else capi.f32c(g_rgb)
            g_acc = None if g_acc is None else capi.f32c(g_acc)
            capi.call("nvsr_composite_backward_mip" if ctx.mip else "nvsr_composite_backward", N, S, capi.ptr(raw), capi.ptr(z), capi.ptr(rd), capi.ptr(ctx.noise), ctx.white, capi.ptr(g_rgb),
                      capi.ptr(g_acc), capi.ptr(g_raw), capi.stream())
        elif N:
            raise NotImplementedError("nvsr_composite_backward handles up to 512 samples per ray")
        return g_raw, None, None, None, None, None


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
    if torch.is_grad_enabled() and radiance_field.requires_grad:
        return _CompositeFn.apply(raw, z, rd, noise, int(bool(white_background)), mip)
    dev = raw.device
    rgb = torch.empty(list(lead) + [3], dtype=torch.float32, device=dev)
    disp, acc, depth = (torch.empty(list(lead), dtype=torch.float32, device=dev) for _ in range(3))
    weights = torch.empty(list(lead) + [S], dtype=torch.float32, device=dev)
    if N:
        capi.call("nvsr_composite_mip" if mip else "nvsr_composite", N, S, capi.ptr(raw), capi.ptr(z), capi.ptr(rd), capi.ptr(noise),
                  int(bool(white_background)), capi.ptr(rgb), capi.ptr(disp), capi.ptr(acc), capi.ptr(weights), capi.ptr(depth), capi.stream())
    return rgb, disp, acc, weights, depth
