"""Synthetic NRD inputs: an analytic scene rendered into the exact input encodings the sample's path tracer
writes (the reference's recorded G-buffers do not exist).

Encodings follow the producer shader of the reference (relative to /root/reference):
  IN_MV .................. Shaders/TraceOpaque.cs.hlsl:609-614 + Shaders/Shared.hlsli:318-335  (xy = pixel motion, z = viewZ delta)
  IN_VIEWZ ............... Shaders/TraceOpaque.cs.hlsl:616-620  (sky = +-INF, INF = 1e5 Shaders/Shared.hlsli:141)
  IN_NORMAL_ROUGHNESS .... Shaders/TraceOpaque.cs.hlsl:657      (R10G10B10A2: oct normal, linear roughness, materialID/3)
  IN_DIFF/SPEC_RADIANCE .. Shaders/TraceOpaque.cs.hlsl:418-421, 756-757 (YCoCg radiance, hit distance normalised for REBLUR only)
  IN_PENUMBRA/TRANSLUCENCY Shaders/TraceOpaque.cs.hlsl:779-804
Scene: ground plane + three spheres + back wall, 90 deg horizontal FOV pinhole
(Settings.camFov, Source/NRDSample.cpp:237), lateral camera dolly, sun from azimuth -147 / elevation 45
(Source/NRDSample.cpp:238-239), sun angular diameter 0.533 deg (:240).
"""
import math

import numpy as np

INF = 1e5
FP16_MAX = 65504.0


def _normalize(v):
    return v / np.maximum(np.sqrt((v * v).sum(-1, keepdims=True)), 1e-20)


def oct_encode(n):
    n = n / np.abs(n).sum(-1, keepdims=True)
    x, y, z = n[..., 0], n[..., 1], n[..., 2]
    sx = np.where(x >= 0, 1.0, -1.0)
    sy = np.where(y >= 0, 1.0, -1.0)
    ox = np.where(z < 0, (1 - np.abs(y)) * sx, x)
    oy = np.where(z < 0, (1 - np.abs(x)) * sy, y)
    return ox * 0.5 + 0.5, oy * 0.5 + 0.5


def pack_normal_roughness(n, roughness, material_id):
    """NRD_FrontEnd_PackNormalAndRoughness for NRD_NORMAL_ENCODING 2 / NRD_ROUGHNESS_ENCODING 1."""
    ox, oy = oct_encode(n.astype(np.float64))
    x = np.floor(np.clip(ox, 0, 1) * 1023 + 0.5).astype(np.uint32)
    y = np.floor(np.clip(oy, 0, 1) * 1023 + 0.5).astype(np.uint32)
    z = np.floor(np.clip(roughness, 0, 1) * 1023 + 0.5).astype(np.uint32)
    return x | (y << 10) | (z << 20) | ((material_id.astype(np.uint32) & 3) << 30)


def linear_to_ycocg(c):
    r, g, b = c[..., 0], c[..., 1], c[..., 2]
    return np.stack([0.25 * r + 0.5 * g + 0.25 * b, 0.5 * r - 0.5 * b, -0.25 * r + 0.5 * g - 0.25 * b], -1)


def ycocg_to_linear(c):
    y, co, cg = c[..., 0], c[..., 1], c[..., 2]
    t = y - cg
    return np.maximum(np.stack([t + co, y + cg, t - co], -1), 0.0)


def reblur_hitdist_norm(view_z, roughness, params=(3.0, 0.1, 20.0, -25.0)):
    a, b, c, d = params
    return (a + np.abs(view_z) * b) * (1.0 + (c - 1.0) * np.exp2(d * roughness * roughness))


def perspective(hfov_deg, aspect):
    """Left-handed, D3D-style column-major projection (clip.w = +z)."""
    m0 = 1.0 / math.tan(math.radians(hfov_deg) * 0.5)
    m5 = m0 * aspect
    near, far = 0.05, 1000.0
    m = np.zeros(16, dtype=np.float32)
    m[0] = m0
    m[5] = m5
    m[10] = far / (far - near)
    m[11] = 1.0
    m[14] = -near * far / (far - near)
    return m


def orthographic(half_width, aspect):
    """Left-handed column-major orthographic projection (clip.w = 1) - the sample's "Ortho" camera (Source/NRDSample.cpp:1214, :1971)."""
    near, far = 0.05, 1000.0
    m = np.zeros(16, dtype=np.float32)
    m[0] = 1.0 / half_width
    m[5] = m[0] * aspect
    m[10] = 1.0 / (far - near)
    m[14] = -near / (far - near)
    m[15] = 1.0
    return m


def look_along(pos, forward):
    """world-to-view matrix of a camera at `pos` looking along `forward` (y-up, no roll, left-handed: +z = forward)"""
    f = np.asarray(forward, dtype=np.float64)
    f = f / np.linalg.norm(f)
    right = np.cross(np.array([0.0, 1.0, 0.0]), f)
    right = right / max(np.linalg.norm(right), 1e-9)
    up = np.cross(f, right)
    r = np.stack([right, up, f])
    m = np.zeros((4, 4))
    m[:3, :3] = r
    m[:3, 3] = -r @ np.asarray(pos, dtype=np.float64)
    m[3, 3] = 1
    return m.T.reshape(16).astype(np.float32)  # column-major


def world_to_view(pos, yaw=0.0, pitch=0.0, roll=0.0):
    cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
    ry = np.array([[cy, 0, -sy], [0, 1, 0], [sy, 0, cy]])
    rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    r = rx @ ry
    if roll != 0.0:  # camera rolled about its view axis (roll = pi / 2 puts the sky at one side of the frame)
        cr, sr = math.cos(roll), math.sin(roll)
        r = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]]) @ r
    t = -r @ np.asarray(pos, dtype=np.float64)
    m = np.zeros((4, 4))
    m[:3, :3] = r
    m[:3, 3] = t
    m[3, 3] = 1
    return m.T.reshape(16).astype(np.float32)  # column-major


class Scene:
    """Analytic scene rendered with torch (CPU or the HIP device); ``frame()`` returns numpy arrays on CPU and device tensors
    on a GPU so that large (4K/8K) frames are produced where the denoiser consumes them."""

    def __init__(self, width, height, seed=0x9E3779B9, hfov=90.0, dolly=0.01, denoiser="REBLUR", rough_bands=True,
                 translucent_sphere=True, device="cpu", frame_height=None, row0=0, ortho=False, roll_deg=0.0, forward=None,
                 sun_deg=(-147.0, 45.0, 0.533), hit_dist_scale=3.0, wall_height=5.0):
        self.w, self.h, self.seed = width, height, seed
        self.ortho = ortho
        self.roll = math.radians(roll_deg)
        # view direction override (y-up world, +z = the default heading): a recorded preset's camera orientation (sample_tests.py)
        self.forward = None if forward is None else np.asarray(forward, dtype=np.float64) / np.linalg.norm(forward)
        self.hit_dist_scale = float(hit_dist_scale)  # ReblurHitDistanceParameters::A (the sample's "HitT scale", NRDSample.cpp:3675)
        self.hfov, self.dolly = hfov, dolly
        self.relax = denoiser == "RELAX"
        self.rough_bands = rough_bands
        self.translucent_sphere = translucent_sphere
        self.device = device
        # row band of a taller frame (row tiling): this scene renders rows [row0, row0 + height) of a frame_height-row image
        self.frame_h = frame_height or height
        self.row0 = row0
        self.proj = orthographic(4.0, width / self.frame_h) if ortho else perspective(hfov, width / self.frame_h)
        # spheres: centre, radius, roughness, materialID
        self.spheres = [((-1.6, 0.7, 4.0), 0.7, 0.05, 1), ((0.3, 1.0, 5.5), 1.0, 0.3, 0), ((2.2, 0.6, 3.5), 0.6, 0.7, 1)]
        self.wall_z = 9.0
        self.wall_height = float(wall_height)  # sky shows above the back wall; float("inf") closes the view (no sky pixel: bench.py's full-coverage leg)
        az, el = math.radians(sun_deg[0]), math.radians(sun_deg[1])
        # the sample's sun direction is z-up (Source/NRDSample.cpp:587-594); this scene is y-up: swap
        sun = np.array([math.cos(az) * math.cos(el), math.sin(el), math.sin(az) * math.cos(el)])
        sun = -sun if sun[2] > 0 else sun  # keep the sun behind the camera so shadows fall into view
        sun[1] = abs(sun[1])
        self.sun = sun / np.linalg.norm(sun)
        self.tan_sun = math.tan(math.radians(sun_deg[2]) * 0.5)
        self.scene_radius = 12.0

    # ---- camera ---------------------------------------------------------------------------------
    def cam_pos(self, frame):
        return np.array([-0.3 + self.dolly * frame, 1.4, -0.5])

    def matrices(self, frame, prev_frame=None):
        prev_frame = max(frame - 1, 0) if prev_frame is None else prev_frame
        if self.forward is not None:
            return look_along(self.cam_pos(frame), self.forward), look_along(self.cam_pos(prev_frame), self.forward)
        return world_to_view(self.cam_pos(frame), 0.0, 0.12, self.roll), world_to_view(self.cam_pos(prev_frame), 0.0, 0.12, self.roll)

    @staticmethod
    def _rot_pos(m):
        m4 = m.reshape(4, 4).T.astype(np.float64)
        return m4[:3, :3], -m4[:3, :3].T @ m4[:3, 3]

    # ---- ray casting (torch) ----------------------------------------------------------------------
    def _t(self, a):
        import torch
        return torch.as_tensor(np.asarray(a, dtype=np.float64), device=self.device)

    def _intersect(self, o, d):
        """nearest hit of rays o + t d: t (inf on miss), object id (-1 miss, 0 ground, 1 wall, 2.. spheres)"""
        import torch
        inf = float("inf")
        t = torch.full(d.shape[:-1], inf, dtype=torch.float64, device=d.device)
        obj = torch.full(d.shape[:-1], -1, dtype=torch.int32, device=d.device)
        tg = -o[..., 1] / d[..., 1]
        ok = (tg > 1e-4) & torch.isfinite(tg)
        t = torch.where(ok, tg, t)
        obj = torch.where(ok, torch.zeros_like(obj), obj)
        tw = (self.wall_z - o[..., 2]) / d[..., 2]
        ok = (tw > 1e-4) & torch.isfinite(tw) & (tw < t) & ((o[..., 1] + tw * d[..., 1]) < self.wall_height)
        t = torch.where(ok, tw, t)
        obj = torch.where(ok, torch.ones_like(obj), obj)
        for i, (c, r, _, _) in enumerate(self.spheres):
            oc = o - self._t(c)
            b = (oc * d).sum(-1)
            cc = (oc * oc).sum(-1) - r * r
            disc = b * b - cc
            sq = torch.sqrt(torch.clamp(disc, min=0))
            ts = -b - sq
            ts = torch.where(ts > 1e-4, ts, -b + sq)
            ok = (disc > 0) & (ts > 1e-4) & (ts < t)
            t = torch.where(ok, ts, t)
            obj = torch.where(ok, torch.full_like(obj, 2 + i), obj)
        return t, obj

    def _normal(self, p, obj):
        import torch
        n = torch.zeros_like(p)
        n = torch.where((obj == 0)[..., None], self._t((0, 1, 0)), n)
        n = torch.where((obj == 1)[..., None], self._t((0, 0, -1)), n)
        for i, (c, r, _, _) in enumerate(self.spheres):
            n = torch.where((obj == 2 + i)[..., None], (p - self._t(c)) / r, n)
        return n

    def _env(self, d):
        import torch
        up = torch.clamp(d[..., 1] * 0.5 + 0.5, 0, 1)
        sky = torch.stack([0.35 + 0.25 * up, 0.45 + 0.3 * up, 0.6 + 0.4 * up], -1)
        s = torch.clamp((d * self._t(self.sun)).sum(-1), 0, 1) ** 64
        return sky + s[..., None] * self._t((6.0, 5.0, 4.0))

    # ---- one frame ------------------------------------------------------------------------------
    def frame(self, index, noise=True, prev_index=None):
        """inputs of frame `index`; motion vectors / previous matrices refer to camera position `prev_index` (default index - 1)"""
        import torch
        w, h = self.w, self.h
        dev = self.device
        gen = torch.Generator(device=dev)
        gen.manual_seed(int((self.seed ^ (index * 0x85EBCA6B) ^ (self.row0 * 0x27D4EB2F)) & 0x7FFFFFFF))

        def randn(*shape):
            return torch.randn(*shape, dtype=torch.float64, device=dev, generator=gen)

        w2v, w2v_prev = self.matrices(index, prev_index)
        rot, pos = self._rot_pos(w2v)
        rot_p, pos_p = self._rot_pos(w2v_prev)
        rot_t, rot_pt, pos_t, pos_pt = self._t(rot), self._t(rot_p), self._t(pos), self._t(pos_p)
        m0, m5 = float(self.proj[0]), float(self.proj[5])
        u = (torch.arange(w, dtype=torch.float64, device=dev) + 0.5) / w
        v = (torch.arange(h, dtype=torch.float64, device=dev) + self.row0 + 0.5) / self.frame_h
        vv, uu = torch.meshgrid(v, u, indexing="ij")
        if self.ortho:  # parallel rays along the view axis, origins spread over the view plane
            ov = torch.stack([(2 * uu - 1) / m0, (1 - 2 * vv) / m5, torch.zeros_like(uu)], -1)
            dv = torch.stack([torch.zeros_like(uu), torch.zeros_like(uu), torch.ones_like(uu)], -1)
        else:
            ov = None
            dv = torch.stack([(2 * uu - 1) / m0, (1 - 2 * vv) / m5, torch.ones_like(uu)], -1)  # view-space ray, z = 1
        dw = dv @ rot_t  # R^T applied to row vectors
        dlen = torch.sqrt((dw * dw).sum(-1, keepdim=True))
        dn = dw / dlen
        o = pos_t.expand_as(dn) if ov is None else pos_t + ov @ rot_t
        t, obj = self._intersect(o, dn)
        hit = obj >= 0
        tt = torch.where(hit, t, torch.zeros_like(t))
        p = o + dn * tt[..., None]
        view_z = torch.where(hit, tt / dlen[..., 0], torch.full_like(tt, INF))
        n = self._normal(p, obj)
        n = torch.where(hit[..., None], n, self._t((0, 0, -1)))

        # roughness / material
        rough = torch.full((h, w), 0.5, dtype=torch.float64, device=dev)
        mat = torch.zeros((h, w), dtype=torch.int64, device=dev)
        if self.rough_bands:
            band = torch.floor(p[..., 0] * 0.5 + 100).to(torch.int64) % 3
            rb = torch.where(band == 0, 0.05, torch.where(band == 1, 0.3, 0.7)).to(torch.float64)
            rough = torch.where(obj == 0, rb, rough)
        rough = torch.where(obj == 1, torch.full_like(rough, 0.6), rough)
        for i, (_, _, r, m) in enumerate(self.spheres):
            rough = torch.where(obj == 2 + i, torch.full_like(rough, r), rough)
            mat = torch.where(obj == 2 + i, torch.full_like(mat, m), mat)

        # motion: previous-frame projection of the same (static) world point
        pv_prev = (p - pos_pt) @ rot_pt.T
        zp = torch.where(hit, pv_prev[..., 2], torch.ones_like(tt))
        wp = torch.ones_like(zp) if self.ortho else zp  # clip.w
        up = 0.5 + 0.5 * (m0 * pv_prev[..., 0] / wp)
        vp = 0.5 - 0.5 * (m5 * pv_prev[..., 1] / wp)
        zero = torch.zeros_like(tt)
        mv = torch.stack([torch.where(hit, (up - uu) * w, zero), torch.where(hit, (vp - vv) * self.frame_h, zero),
                          torch.where(hit, zp - view_z, zero), zero], -1)

        # lighting ("ground truth" demodulated signals)
        sun_t = self._t(self.sun)
        shadow_o = p + n * 1e-3
        sun_dirs = sun_t.expand_as(p)
        if noise:
            sun_dirs = sun_dirs + randn(h, w, 3) * (self.tan_sun * 0.7)
            sun_dirs = sun_dirs / torch.sqrt((sun_dirs * sun_dirs).sum(-1, keepdim=True))
        ts, sobj = self._intersect(shadow_o, sun_dirs)
        ndl = torch.clamp((n * sun_t).sum(-1), 0, 1)
        lit = (sobj < 0) & (ndl > 0)
        glass = (sobj == 3) if self.translucent_sphere else torch.zeros_like(lit)
        dmin = torch.full((h, w), 10.0, dtype=torch.float64, device=dev)
        for c, r, _, _ in self.spheres:
            dmin = torch.minimum(dmin, torch.abs(torch.sqrt(((p - self._t(c)) ** 2).sum(-1)) - r))
        ao = torch.clamp(0.35 + 0.65 * dmin / 1.5, 0, 1)
        sky_irr = self._t((0.45, 0.55, 0.75)) * ao[..., None] * (0.6 + 0.4 * torch.clamp(n[..., 1:2], 0, 1))
        diff = sky_irr  # the sun goes through SIGMA, not through the diffuse signal
        refl = dn - 2 * (dn * n).sum(-1, keepdim=True) * n
        tr, robj = self._intersect(shadow_o, refl)
        spec = self._env(refl) * torch.where(robj[..., None] >= 0, 0.35, 1.0)
        diff_hit = torch.clamp(dmin * 1.5 + 0.2, 0.05, 8.0)
        spec_hit = torch.where(robj >= 0, torch.clamp(tr, max=50.0), torch.full_like(tr, 50.0))
        if noise:
            sigma = 1.0
            diff = diff * torch.exp(sigma * randn(h, w, 1) - 0.5 * sigma * sigma)
            sig_s = 0.3 + 0.9 * rough[..., None]
            spec = spec * torch.exp(sig_s * randn(h, w, 1) - 0.5 * sig_s * sig_s)
            diff_hit = diff_hit * torch.exp(0.5 * randn(h, w))
            spec_hit = spec_hit * torch.exp(0.3 * rough * randn(h, w))

        def ycocg(c):
            r_, g_, b_ = c[..., 0], c[..., 1], c[..., 2]
            return torch.stack([0.25 * r_ + 0.5 * g_ + 0.25 * b_, 0.5 * r_ - 0.5 * b_, -0.25 * r_ + 0.5 * g_ - 0.25 * b_], -1)

        def hitnorm(z, r_):
            return (self.hit_dist_scale + torch.abs(z) * 0.1) * (1.0 + 19.0 * torch.exp2(-25.0 * r_ * r_))

        out = {}
        out["viewz"] = view_z.to(torch.float32)
        out["mv"] = mv.to(torch.float16)
        # R10G10B10A2 pack (int64 arithmetic, stored as the int32 bit pattern)
        na = n / torch.abs(n).sum(-1, keepdim=True)
        nx, ny, nz = na[..., 0], na[..., 1], na[..., 2]
        sx = torch.where(nx >= 0, 1.0, -1.0)
        sy = torch.where(ny >= 0, 1.0, -1.0)
        ox = torch.where(nz < 0, (1 - torch.abs(ny)) * sx, nx) * 0.5 + 0.5
        oy = torch.where(nz < 0, (1 - torch.abs(nx)) * sy, ny) * 0.5 + 0.5
        q = lambda a: torch.floor(torch.clamp(a, 0, 1) * 1023 + 0.5).to(torch.int64)
        packed = q(ox) | (q(oy) << 10) | (q(rough) << 20) | ((mat & 3) << 30)
        packed = torch.where(packed >= 2 ** 31, packed - 2 ** 32, packed).to(torch.int32)
        out["normal_roughness"] = packed
        if self.relax:
            d4 = torch.cat([torch.clamp(diff, max=FP16_MAX), diff_hit[..., None]], -1)
            s4 = torch.cat([torch.clamp(spec, max=FP16_MAX), spec_hit[..., None]], -1)
        else:
            nd = torch.clamp(diff_hit / hitnorm(view_z, torch.ones_like(rough)), 0, 1)
            ns = torch.clamp(spec_hit / hitnorm(view_z, rough), 0, 1)
            d4 = torch.cat([ycocg(diff), nd[..., None]], -1)
            s4 = torch.cat([ycocg(spec), ns[..., None]], -1)
        d4 = torch.where(hit[..., None], d4, torch.zeros_like(d4))
        s4 = torch.where(hit[..., None], s4, torch.zeros_like(s4))
        out["diff"] = d4.to(torch.float16)
        out["spec"] = s4.to(torch.float16)
        # SH variants (REBLUR_/RELAX_FrontEnd_PackSh, Shaders/TraceOpaque.cs.hlsl:738-752): SH1 = incoming direction weighted by
        # the luminance of the sample (diffuse: around the normal, specular: around the reflection vector)
        def sh1(direction, rad):
            lum = 0.2126 * rad[..., 0:1] + 0.7152 * rad[..., 1:2] + 0.0722 * rad[..., 2:3]
            v = torch.cat([direction * lum, torch.zeros_like(lum)], -1)
            return torch.where(hit[..., None], v, torch.zeros_like(v)).to(torch.float16)
        out["diff_sh1"] = sh1(n, diff)
        out["spec_sh1"] = sh1(refl, spec)
        if not self.relax:
            # DIRECTIONAL_OCCLUSION (REBLUR_FrontEnd_PackDirectionalOcclusion, Shaders/TraceOpaque.cs.hlsl:753-754):
            # one RGBA16F texel {ray direction * normHitDist, normHitDist}; the noisy direction scatters around the normal
            jit = torch.stack([nd, 1.0 - nd, nd * 0.5], -1) - 0.5  # deterministic per-pixel perturbation from the noisy hit distance
            dirn = n + 0.35 * jit
            dirn = dirn / torch.sqrt((dirn * dirn).sum(-1, keepdim=True))
            do4 = torch.cat([dirn * nd[..., None], nd[..., None]], -1)
            out["diff_dirocc"] = torch.where(hit[..., None], do4, torch.zeros_like(do4)).to(torch.float16)
        if not self.relax:  # OCCLUSION variants take the normalised hit distance alone, R16_UNORM
            for key, src in (("diff_hitdist", d4), ("spec_hitdist", s4)):
                q = torch.floor(torch.clamp(src[..., 3], 0, 1) * 65535 + 0.5).to(torch.int32)  # 0..65535 (sky = 0)
                # stored as a 16-bit pattern: int16 reinterpretation keeps the bits on both backends
                out[key] = torch.where(q >= 32768, q - 65536, q).to(torch.int16)
        pen = torch.where(lit, torch.full_like(ts, FP16_MAX), torch.clamp(torch.where(torch.isfinite(ts), ts, torch.zeros_like(ts)) * self.tan_sun, max=1000.0))
        pen = torch.where(ndl > 0, pen, torch.zeros_like(pen))  # back-facing: fully shadowed at distance 0
        pen = torch.where(hit, pen, torch.full_like(pen, FP16_MAX))
        out["penumbra"] = pen.to(torch.float16)
        tl = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)
        tl[..., 0] = torch.where(lit, 255, 0).to(torch.uint8)
        tint = torch.tensor([230, 150, 80], dtype=torch.uint8, device=dev)
        tl[..., 1:] = torch.where((glass & ~lit)[..., None], tint, torch.zeros_like(tint))
        out["translucency"] = tl
        cw, ch = (w + 4) // 5, (h + 4) // 5
        conf = torch.zeros((ch, cw, 4), dtype=torch.float16, device=dev)
        conf[..., 0] = 1.0
        out["confidence"] = conf
        comp = d4[..., :3] * 0.8 + s4[..., :3] * 0.2
        out["signal"] = torch.cat([comp, torch.ones((h, w, 1), dtype=torch.float64, device=dev)], -1).to(torch.float16)
        out["clean_diff"] = sky_irr.to(torch.float32)
        if dev == "cpu":
            out = {k: (x.numpy().view(np.uint32) if k == "normal_roughness" else x.numpy()) for k, x in out.items()}
            for k in ("diff_hitdist", "spec_hitdist"):
                if k in out:
                    out[k] = out[k].view(np.uint16)
        out["world_to_view"], out["world_to_view_prev"] = w2v, w2v_prev
        out["view_to_clip"] = self.proj
        return out

    def common_settings(self, api, frame_data, index, reset=False):
        """nrd::CommonSettings exactly as Sample::RenderFrame fills it (Source/NRDSample.cpp:3835-3876)."""
        cs = api.CommonSettings()
        for i in range(16):
            cs.viewToClipMatrix[i] = cs.viewToClipMatrixPrev[i] = float(frame_data["view_to_clip"][i])
            cs.worldToViewMatrix[i] = float(frame_data["world_to_view"][i])
            cs.worldToViewMatrixPrev[i] = float(frame_data["world_to_view_prev"][i])
        cs.motionVectorScale[0] = 1.0 / self.w
        cs.motionVectorScale[1] = 1.0 / self.frame_h
        cs.motionVectorScale[2] = 1.0
        for k in ("resourceSize", "resourceSizePrev", "rectSize", "rectSizePrev"):
            getattr(cs, k)[0] = self.w
            getattr(cs, k)[1] = self.frame_h
        cs.viewZScale = 1.0
        cs.denoisingRange = 4.0 * self.scene_radius
        cs.disocclusionThreshold = 0.01
        cs.disocclusionThresholdAlternate = 0.1
        cs.splitScreen = 0.0
        cs.frameIndex = index
        cs.accumulationMode = int(api.AccumulationMode.CLEAR_AND_RESTART if reset else api.AccumulationMode.CONTINUE)
        cs.isHistoryConfidenceAvailable = True
        return cs
