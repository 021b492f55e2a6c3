"""ctypes front-end of the CPU checker (oracle/nvsr_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (neural-volume-super-resolution_amd/) never does.  numpy in, numpy out.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = C.POINTER(C.c_float)


def build(force=False):
    """Compile liborc.so / liborc_f32.so with gcc (oracle/Makefile)."""
    libs = [os.path.join(HERE, n) for n in ("liborc.so", "liborc_f32.so")]
    src = os.path.join(HERE, "nvsr_oracle.c")
    if force or not all(os.path.exists(l) and os.path.getmtime(l) >= os.path.getmtime(src) for l in libs):
        subprocess.check_call(["make", "-s", "-C", HERE] + (["-B"] if force else []))


def _p(a):
    return None if a is None else a.ctypes.data_as(_f32p)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class _Decoder(C.Structure):
    _fields_ = [("C", C.c_int), ("hidden", C.c_int), ("nd", C.c_int), ("nr", C.c_int), ("blob", _f32p)]


class _Cfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("num_coarse", "num_fine", "lindisp", "perturb", "white_background")]


DECODER_KEYS = (
    [("density_dec.0.%d.weight" % i, "density_dec.0.%d.bias" % i) for i in range(4)]
    + [("fc_alpha.0.weight", "fc_alpha.0.bias")]
    + [("rgb_dec.0.%d.weight" % i, "rgb_dec.0.%d.bias" % i) for i in range(4)]
    + [("fc_rgb.0.weight", "fc_rgb.0.bias")]
)


def decoder_blob(sd, prefix=""):
    """Flatten a TwoDimPlanesModel state dict (models.py:169-195 key names) into the natural blob."""
    parts = []
    for w, b in DECODER_KEYS:
        parts += [np.asarray(sd[prefix + w], np.float32).ravel(), np.asarray(sd[prefix + b], np.float32).ravel()]
    return np.concatenate(parts)


class Oracle:
    def __init__(self, f32=False):
        build()
        # NVSR_ORACLE_SANITIZE=1 (tests/test_oracle.py, with libasan preloaded): both flavours run the -fsanitize=address,undefined build
        name = "liborc_san.so" if os.environ.get("NVSR_ORACLE_SANITIZE") == "1" else ("liborc_f32.so" if f32 else "liborc.so")
        if name == "liborc_san.so" and not os.path.exists(os.path.join(HERE, name)):
            subprocess.check_call(["make", "-s", "-C", HERE, "san"])
        self.lib = C.CDLL(os.path.join(HERE, name))
        self.lib.orc_decoder_blob_floats.restype = C.c_long
        self.lib.orc_edsr_blob_floats.restype = C.c_long
        self._keep = []

    # -- rays ------------------------------------------------------------------------------
    def get_ray_bundle(self, H, W, focal, c2w, padding_size=0, downsampling_offset=0.0):
        fx, fy = (focal[1], focal[0]) if isinstance(focal, (list, tuple)) else (focal, focal)
        c2w = _f(c2w)
        Hp, Wp = H + 2 * padding_size, W + 2 * padding_size
        ro = np.empty((Hp, Wp, 3), np.float32)
        rd = np.empty((Hp, Wp, 3), np.float32)
        self.lib.orc_get_ray_bundle(H, W, C.c_double(fx), C.c_double(fy), _p(c2w), padding_size,
                                    C.c_double(downsampling_offset), _p(ro), _p(rd))
        return ro, rd

    def ndc_rays(self, H, W, focal, near, ro, rd):
        ro, rd = _f(ro).reshape(-1, 3), _f(rd).reshape(-1, 3)
        o, d = np.empty_like(ro), np.empty_like(rd)
        self.lib.orc_ndc_rays(H, W, C.c_double(focal), C.c_double(near), ro.shape[0], _p(ro), _p(rd), _p(o), _p(d))
        return o, d

    def coarse_z(self, near, far, Nc, lindisp=False, perturb=False, t_rand=None):
        near, far = _f(near).ravel(), _f(far).ravel()
        z = np.empty((near.size, Nc), np.float32)
        t = _f(t_rand) if (perturb and t_rand is not None) else None
        self.lib.orc_coarse_z(near.size, Nc, _p(near), _p(far), int(lindisp), int(perturb), _p(t), _p(z))
        return z

    # -- scene / decoder handles ----------------------------------------------------------------
    def scene(self, planes, box, rot=None):
        """planes: 4 NCHW arrays ([1,C,H,W] or [C,H,W]); box [2,5] float64; rot: optional [3,3,3]."""
        pl = [_f(np.asarray(p).reshape(p.shape[-3:])) for p in planes]
        hw = np.array([v for p in pl for v in p.shape[1:]], np.int32)
        boxd = np.ascontiguousarray(box, np.float64)
        rotf = None if rot is None else _f(rot)
        buf = C.create_string_buffer(self.lib.orc_scene_sizeof())
        self.lib.orc_scene_init(buf, _p(pl[0]), _p(pl[1]), _p(pl[2]), _p(pl[3]), hw.ctypes.data_as(C.POINTER(C.c_int)),
                                boxd.ctypes.data_as(C.POINTER(C.c_double)), _p(rotf))
        self._keep.append((pl, hw, boxd, rotf))
        return buf

    def decoder(self, blob, Cch=48, hidden=128, nd=4, nr=4):
        blob = _f(blob)
        assert blob.size == self.lib.orc_decoder_blob_floats(Cch, hidden, nd, nr), "decoder blob size mismatch"
        self._keep.append(blob)
        return _Decoder(Cch, hidden, nd, nr, _p(blob))

    def triplane_decode(self, scene, dec, x, want_feats=False):
        x = _f(x).reshape(-1, 6)
        P = x.shape[0]
        out = np.empty((P, 4), np.float32)
        feats = np.empty((P, 5 * dec.C), np.float32) if want_feats else None
        n5 = np.empty((P, 5), np.float32) if want_feats else None
        self.lib.orc_triplane_decode(scene, C.byref(dec), C.c_long(P), _p(x), _p(out), _p(feats), _p(n5))
        return (out, feats, n5) if want_feats else out

    # -- compositing / sampling --------------------------------------------------------------
    def composite(self, raw, z, rd, noise=None, white_background=False, mip_nerf=False):
        raw, z, rd = _f(raw), _f(z), _f(rd)
        N, S = z.shape[0], z.shape[1] - (1 if mip_nerf else 0)
        noise = None if noise is None else _f(noise)
        rgb = np.empty((N, 3), np.float32)
        disp, acc, depth = (np.empty(N, np.float32) for _ in range(3))
        w = np.empty((N, S), np.float32)
        fn = self.lib.orc_composite_mip if mip_nerf else self.lib.orc_composite
        fn(C.c_long(N), S, _p(raw), _p(z), _p(rd), _p(noise), int(white_background), _p(rgb), _p(disp), _p(acc), _p(w), _p(depth))
        return rgb, disp, acc, w, depth

    def cumprod_exclusive(self, t):
        t = _f(t)
        out = np.empty_like(t)
        self.lib.orc_cumprod_exclusive(C.c_long(int(np.prod(t.shape[:-1]))), t.shape[-1], _p(t), _p(out))
        return out

    def sample_pdf(self, bins, weights, u):
        bins, weights, u = _f(bins), _f(weights), _f(u)
        N, nb = bins.shape
        out = np.empty_like(u)
        self.lib.orc_sample_pdf(C.c_long(N), nb, u.shape[1], _p(bins), _p(weights), _p(u), _p(out))
        return out

    def sort_rows(self, a):
        a = _f(a).copy()
        self.lib.orc_sort_rows(C.c_long(a.shape[0]), a.shape[1], _p(a))
        return a

    def pack_rays(self, ro, rd, near, far, dirs_for_view=None):
        ro, rd = _f(ro).reshape(-1, 3), _f(rd).reshape(-1, 3)
        v = rd if dirs_for_view is None else _f(dirs_for_view).reshape(-1, 3)
        rays = np.empty((ro.shape[0], 11), np.float32)
        self.lib.orc_pack_rays(C.c_long(ro.shape[0]), _p(ro), _p(rd), _p(v), C.c_double(near), C.c_double(far), _p(rays))
        return rays

    def render_rays(self, scene, dec_c, dec_f, rays, num_coarse, num_fine, lindisp=False, perturb=False,
                    white_background=False, t_rand=None, u=None, noise_coarse=None, noise_fine=None, want_aux=False):
        rays = _f(rays)
        N = rays.shape[0]
        cfg = _Cfg(num_coarse, num_fine, int(lindisp), int(perturb), int(white_background))
        o = {k: np.empty((N, 3), np.float32) for k in ("rgb_coarse", "rgb_fine")}
        o.update({k: np.empty(N, np.float32) for k in ("disp_coarse", "acc_coarse", "disp_fine", "acc_fine")})
        zf = np.empty((N, num_coarse + num_fine), np.float32) if want_aux else None
        wc = np.empty((N, num_coarse), np.float32) if want_aux else None
        t_rand, u, nc, nf = (None if a is None else _f(a) for a in (t_rand, u, noise_coarse, noise_fine))
        self.lib.orc_render_rays(scene, C.byref(dec_c), C.byref(dec_f), C.byref(cfg), C.c_long(N), _p(rays), _p(t_rand),
                                 _p(u), _p(nc), _p(nf), _p(o["rgb_coarse"]), _p(o["disp_coarse"]), _p(o["acc_coarse"]),
                                 _p(o["rgb_fine"]), _p(o["disp_fine"]), _p(o["acc_fine"]), _p(zf), _p(wc))
        if num_fine <= 0:
            for k in ("rgb_fine", "disp_fine", "acc_fine"):
                o[k] = None
        if want_aux:
            o["z_fine"], o["weights_coarse"] = zf, wc
        return o

    def render_given_z(self, scene, dec, rays, z, noise=None, white_background=False, want_raw=False):
        rays, z = _f(rays), _f(z)
        N, S = z.shape
        noise = None if noise is None else _f(noise)
        o = dict(rgb=np.empty((N, 3), np.float32), disp=np.empty(N, np.float32), acc=np.empty(N, np.float32),
                 weights=np.empty((N, S), np.float32), depth=np.empty(N, np.float32))
        raw = np.empty((N, S, 4), np.float32) if want_raw else None
        self.lib.orc_render_given_z(scene, C.byref(dec), C.c_long(N), S, _p(rays), _p(z), _p(noise), int(white_background),
                                    _p(o["rgb"]), _p(o["disp"]), _p(o["acc"]), _p(o["weights"]), _p(o["depth"]), _p(raw))
        if want_raw:
            o["raw"] = raw
        return o

    def render_backward(self, scene, plane_shapes, dec_c, dec_f, rays, num_coarse, num_fine, g_rgb_coarse, g_rgb_fine, lindisp=False,
                        perturb=False, white_background=False, t_rand=None, u=None, noise_coarse=None, noise_fine=None, z_fine=None):
        """gradient of sum(g_rgb_coarse * rgb_coarse) + sum(g_rgb_fine * rgb_fine) wrt the 4 planes ([C,H,W] each)"""
        rays = _f(rays)
        N = rays.shape[0]
        cfg = _Cfg(num_coarse, num_fine, int(lindisp), int(perturb), int(white_background))
        t_rand, u, nc, nf, gc, gf, zf = (None if a is None else _f(a) for a in (t_rand, u, noise_coarse, noise_fine, g_rgb_coarse, g_rgb_fine, z_fine))
        grads = [np.zeros(tuple(sh[-3:]), np.float32) for sh in plane_shapes]
        self.lib.orc_render_backward(scene, C.byref(dec_c), C.byref(dec_f), C.byref(cfg), C.c_long(N), _p(rays), _p(t_rand), _p(u), _p(nc),
                                     _p(nf), _p(gc), _p(gf), _p(zf), *[_p(g) for g in grads])
        return grads

    def render_backward_decoder(self, scene, dec_c, dec_f, rays, num_coarse, num_fine, g_rgb_coarse, g_rgb_fine, lindisp=False,
                                perturb=False, white_background=False, t_rand=None, u=None, noise_coarse=None, noise_fine=None, z_fine=None):
        """same step as render_backward: gradients wrt the decoder parameters of (coarse, fine), state-dict order"""
        rays = _f(rays)
        N = rays.shape[0]
        cfg = _Cfg(num_coarse, num_fine, int(lindisp), int(perturb), int(white_background))
        t_rand, u, nc, nf, gc, gf, zf = (None if a is None else _f(a) for a in (t_rand, u, noise_coarse, noise_fine, g_rgb_coarse, g_rgb_fine, z_fine))
        n = self.lib.orc_decoder_blob_floats(dec_c.C, dec_c.hidden, dec_c.nd, dec_c.nr)
        gc_, gf_ = np.zeros(n, np.float32), np.zeros(n, np.float32)
        self.lib.orc_render_backward_dec(scene, C.byref(dec_c), C.byref(dec_f), C.byref(cfg), C.c_long(N), _p(rays), _p(t_rand), _p(u), _p(nc),
                                         _p(nf), _p(gc), _p(gf), _p(zf), None, None, None, None, _p(gc_), _p(gf_))
        return gc_, gf_

    # -- feature-plane super-resolution -----------------------------------------------------------
    @staticmethod
    def edsr_keys(sd, prefix="inner_model.", n_up=2):
        keys = [prefix + "conv_input.weight"]
        b = 0
        while prefix + "residual.%d.conv1.weight" % b in sd:
            keys += [prefix + "residual.%d.conv1.weight" % b, prefix + "residual.%d.conv2.weight" % b]
            b += 1
        keys += [prefix + "conv_mid.weight"] + [prefix + "upscale.%d.weight" % (2 * i) for i in range(n_up)]
        keys += [prefix + "conv_output.weight"]
        return keys

    @staticmethod
    def edsr_blob(sd, prefix="inner_model.", nblocks=None, n_up=2):
        keys = Oracle.edsr_keys(sd, prefix, n_up)
        return np.concatenate([np.asarray(sd[k], np.float32).ravel() for k in keys]), (len(keys) - 2 - n_up) // 2

    def conv3x3(self, x, w, relu=False):
        x, w = _f(x), _f(w)
        Ci, H, W = x.shape[-3:]
        Co = w.shape[0]
        out = np.empty((Co, H - 2, W - 2), np.float32)
        self.lib.orc_conv3x3_valid(_p(x), Ci, H, W, _p(w), Co, int(relu), _p(out))
        return out

    def conv3x3_backward(self, x, w, dy):
        """-> (dx [Ci,H,W], dw [Co,Ci,3,3]) of the valid 3x3 conv"""
        x, w, dy = _f(x), _f(w), _f(dy)
        Ci, H, W = x.shape[-3:]
        Co = w.shape[0]
        dx = np.empty((Ci, H, W), np.float32)
        dw = np.zeros((Co, Ci, 3, 3), np.float64)
        self.lib.orc_conv3x3_valid_backward(_p(x), Ci, H, W, _p(w), Co, _p(dy), _p(dx), dw.ctypes.data_as(C.POINTER(C.c_double)))
        return dx, dw

    def edsr_forward(self, x, blob, Cout, hid, nblocks, n_up):
        x, blob = _f(x), _f(blob)
        Cin, H, W = x.shape[-3:]
        assert blob.size == self.lib.orc_edsr_blob_floats(Cin, Cout, hid, nblocks, n_up)
        Ho, Wo = C.c_int(), C.c_int()
        self.lib.orc_edsr_out_size(H, W, nblocks, n_up, C.byref(Ho), C.byref(Wo))
        out = np.empty((Cout, Ho.value, Wo.value), np.float32)
        self.lib.orc_edsr_forward(_p(x), Cin, H, W, _p(blob), Cout, hid, nblocks, n_up, _p(out))
        return out

    def upsample_bilinear(self, x, sf):
        x = _f(x)
        Cc, H, W = x.shape[-3:]
        out = np.empty((Cc, H * sf, W * sf), np.float32)
        self.lib.orc_upsample_bilinear_ac(_p(x), Cc, H, W, sf, _p(out))
        return out

    def planes_sr(self, lr, blob, hid, nblocks, n_up, pad, over, roi=None, mean=None, std=None):
        lr, blob = _f(lr), _f(blob)
        Cc, R0, R1 = lr.shape[-3:]
        sf = 1 << n_up
        out = np.empty((Cc, R0 * sf, R1 * sf), np.float32)
        roi = None if roi is None else _f(roi)
        mean, std = (None if a is None else _f(a) for a in (mean, std))
        self.lib.orc_planes_sr(_p(lr), Cc, R0, R1, _p(blob), hid, nblocks, n_up, pad, over, _p(roi), _p(mean), _p(std), _p(out))
        return out

    def edsr_backward(self, x, blob, Cout, hid, nblocks, n_up, d_out, want_dx=True):
        """-> (d_blob [same order as blob], dx or None)"""
        x, blob, d_out = _f(x), _f(blob), _f(d_out)
        Cin, H, W = x.shape[-3:]
        d_blob = np.empty_like(blob)
        dx = np.empty((Cin, H, W), np.float32) if want_dx else None
        self.lib.orc_edsr_backward(_p(x), Cin, H, W, _p(blob), Cout, hid, nblocks, n_up, _p(d_out), _p(d_blob), _p(dx))
        return d_blob, dx

    def planes_sr_backward(self, lr, blob, hid, nblocks, n_up, pad, over, d_out, roi=None, mean=None, std=None, want_dlr=True):
        """-> (d_blob, d_lr or None); entries of d_out outside the ROI are ignored"""
        lr, blob, d_out = _f(lr), _f(blob), _f(d_out)
        Cc, R0, R1 = lr.shape[-3:]
        roi = None if roi is None else _f(roi)
        mean, std = (None if a is None else _f(a) for a in (mean, std))
        d_blob = np.empty_like(blob)
        d_lr = np.empty((Cc, R0, R1), np.float32) if want_dlr else None
        self.lib.orc_planes_sr_backward(_p(lr), Cc, R0, R1, _p(blob), hid, nblocks, n_up, pad, over, _p(roi), _p(mean), _p(std), _p(d_out),
                                        _p(d_blob), _p(d_lr))
        return d_blob, d_lr

    # -- positional-encoding baseline ------------------------------------------------------------
    def positional_encoding(self, x, L=6, include_input=True):
        x = _f(x)
        P, D = x.shape
        out = np.empty((P, (D if include_input else 0) + 2 * D * L), np.float32)
        self.lib.orc_positional_encoding(C.c_long(P), D, _p(x), L, int(include_input), _p(out))
        return out

    def flexible_nerf(self, x, blob, dim_xyz, dim_dir, hidden, num_layers, skip_every):
        x, blob = _f(x), _f(blob)
        out = np.empty((x.shape[0], 4), np.float32)
        self.lib.orc_flexible_nerf(C.c_long(x.shape[0]), _p(x), dim_xyz, dim_dir, hidden, num_layers, skip_every, _p(blob), _p(out))
        return out


# ---- training inputs: the device pixel sampler (include/nvsr.h: nvsr_sample_pixels) and the paired MSE ---------------------------
# Not a restatement of reference code (the reference draws with numpy's host generator, train_nerf.py:836-838): the restatement of the
# permutation nvsr.h specifies, in numpy integer arithmetic, so that the HIP kernel is checked bit for bit.
_M64 = (1 << 64) - 1


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def pixel_permutation(total, key, first, n):
    """entries [first, first + n) of the keyed permutation of range(total): 8-round balanced Feistel network + cycle walking -> int64 [n]"""
    hb = 1
    while hb < 31 and (1 << (2 * hb)) < total:
        hb += 1
    mask = np.uint64((1 << hb) - 1)
    rk = [np.uint64(_splitmix64((key + r) & _M64) >> 32) for r in range(8)]
    m32 = np.uint64(0xFFFFFFFF)

    def feistel(x):
        L, R = (x >> np.uint64(hb)) & mask, x & mask
        for r in range(8):
            h = (R * np.uint64(0x9E3779B1) + rk[r]) & m32
            h ^= h >> np.uint64(15); h = (h * np.uint64(0x85EBCA77)) & m32
            h ^= h >> np.uint64(13); h = (h * np.uint64(0xC2B2AE3D)) & m32
            h ^= h >> np.uint64(16)
            L, R = R, L ^ (h & mask)
        return (L << np.uint64(hb)) | R

    x = feistel(np.arange(first, first + n, dtype=np.uint64))
    while True:
        out = x >= np.uint64(total)
        if not out.any():
            return x.astype(np.int64)
        x[out] = feistel(x[out])


def sample_pixels(total, H, key, first, n):
    """-> (row, col) [n,2] of nvsr_sample_pixels: index k of the permutation is pixel (k % H, k // H) (train_nerf.py:818-828)"""
    k = pixel_permutation(total, key, first, n)
    return np.stack([k % H, k // H], -1)
