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


def world_to_view(pos, yaw=0.0, pitch=0.0):
    cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
    ry = np.array([[cy, 0, -sy], [0, 1, 0], [sy, 0, cy]])
    rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    r = rx @ ry
    t = -r @ np.asarray(pos, dtype=np.float64)
    m = np.zeros((4, 4))
    m[:3, :3] = r
    m[:3, 3] = t
    m[3, 3] = 1
    return m.T.reshape(16).astype(np.float32)  # column-major


class Scene:
    def __init__(self, width, height, seed=0x9E3779B9, hfov=90.0, dolly=0.01, denoiser="REBLUR", rough_bands=True,
                 translucent_sphere=True):
        self.w, self.h, self.seed = width, height, seed
        self.hfov, self.dolly = hfov, dolly
        self.relax = denoiser == "RELAX"
        self.rough_bands = rough_bands
        self.translucent_sphere = translucent_sphere
        self.proj = perspective(hfov, width / height)
        # spheres: centre, radius, roughness, materialID
        self.spheres = [((-1.6, 0.7, 4.0), 0.7, 0.05, 1), ((0.3, 1.0, 5.5), 1.0, 0.3, 0), ((2.2, 0.6, 3.5), 0.6, 0.7, 1)]
        self.wall_z = 9.0
        az, el = math.radians(-147.0), math.radians(45.0)
        # sample's sun direction uses z-up; this scene is y-up: swap
        self.sun = np.array([math.cos(az) * math.cos(el), math.sin(el), math.sin(az) * math.cos(el)])
        self.sun = -self.sun if self.sun[2] > 0 else self.sun  # keep the sun behind the camera so shadows fall into view
        self.sun = self.sun / np.linalg.norm(self.sun)
        self.tan_sun = math.tan(math.radians(0.533) * 0.5)
        self.scene_radius = 12.0

    # ---- camera ---------------------------------------------------------------------------------
    def cam_pos(self, frame):
        return np.array([-0.3 + self.dolly * frame, 1.4, -0.5])

    def matrices(self, frame):
        return world_to_view(self.cam_pos(frame), 0.0, 0.12), world_to_view(self.cam_pos(max(frame - 1, 0)), 0.0, 0.12)

    def _rot_pos(self, m):
        m4 = m.reshape(4, 4).T.astype(np.float64)
        return m4[:3, :3], -m4[:3, :3].T @ m4[:3, 3]

    # ---- ray casting ----------------------------------------------------------------------------
    def _intersect(self, o, d, skip_translucent=False):
        """nearest hit of rays o + t d. Returns t (inf on miss), object id (-1 miss, 0 ground, 1 wall, 2.. spheres)."""
        t = np.full(d.shape[:-1], np.inf)
        obj = np.full(d.shape[:-1], -1, dtype=np.int32)
        with np.errstate(divide="ignore", invalid="ignore"):
            tg = -o[..., 1] / d[..., 1]
        ok = (tg > 1e-4) & np.isfinite(tg)
        t = np.where(ok & (tg < t), tg, t)
        obj = np.where(ok & (tg == t), 0, obj)
        with np.errstate(divide="ignore", invalid="ignore"):
            tw = (self.wall_z - o[..., 2]) / d[..., 2]
        ok = (tw > 1e-4) & np.isfinite(tw) & (tw < t) & ((o[..., 1] + tw * d[..., 1]) < 5.0)
        t = np.where(ok, tw, t)
        obj = np.where(ok, 1, obj)
        for i, (c, r, _, _) in enumerate(self.spheres):
            if skip_translucent and self.translucent_sphere and i == 1:
                continue
            oc = o - np.asarray(c)
            b = (oc * d).sum(-1)
            cc = (oc * oc).sum(-1) - r * r
            disc = b * b - cc
            sq = np.sqrt(np.maximum(disc, 0))
            ts = -b - sq
            ts = np.where(ts > 1e-4, ts, -b + sq)
            ok = (disc > 0) & (ts > 1e-4) & (ts < t)
            t = np.where(ok, ts, t)
            obj = np.where(ok, 2 + i, obj)
        return t, obj

    def _normal(self, p, obj):
        n = np.zeros(p.shape)
        n[obj == 0] = (0, 1, 0)
        n[obj == 1] = (0, 0, -1)
        for i, (c, r, _, _) in enumerate(self.spheres):
            m = obj == 2 + i
            n[m] = (p[m] - np.asarray(c)) / r
        return n

    def _env(self, d):
        """smooth environment radiance along direction d"""
        up = np.clip(d[..., 1] * 0.5 + 0.5, 0, 1)
        sky = np.stack([0.35 + 0.25 * up, 0.45 + 0.3 * up, 0.6 + 0.4 * up], -1)
        s = np.clip((d * self.sun).sum(-1), 0, 1) ** 64
        return sky + s[..., None] * np.array([6.0, 5.0, 4.0])

    # ---- one frame ------------------------------------------------------------------------------
    def frame(self, index, noise=True):
        w, h = self.w, self.h
        rng = np.random.default_rng((self.seed ^ (index * 0x85EBCA6B)) & 0xFFFFFFFF)
        w2v, w2v_prev = self.matrices(index)
        rot, pos = self._rot_pos(w2v)
        rot_p, pos_p = self._rot_pos(w2v_prev)
        m0, m5 = float(self.proj[0]), float(self.proj[5])
        u = (np.arange(w) + 0.5) / w
        v = (np.arange(h) + 0.5) / h
        uu, vv = np.meshgrid(u, v)
        dv = np.stack([(2 * uu - 1) / m0, (1 - 2 * vv) / m5, np.ones_like(uu)], -1)  # view-space ray, z = 1
        dw = dv @ rot  # R^T applied to row vectors
        dlen = np.sqrt((dw * dw).sum(-1, keepdims=True))
        dn = dw / dlen
        o = np.broadcast_to(pos, dn.shape)
        t, obj = self._intersect(o, dn)
        hit = obj >= 0
        tt = np.where(hit, t, 0.0)
        p = o + dn * tt[..., None]
        view_z = np.where(hit, tt / dlen[..., 0], INF).astype(np.float32)
        n = self._normal(p, obj)
        n[~hit] = (0, 0, -1)

        # roughness / material
        rough = np.full((h, w), 0.5)
        mat = np.zeros((h, w), dtype=np.uint32)
        if self.rough_bands:
            band = np.floor(p[..., 0] * 0.5 + 100).astype(np.int64) % 3
            rough = np.where(obj == 0, np.choose(band, [0.05, 0.3, 0.7]), rough)
        rough = np.where(obj == 1, 0.6, rough)
        for i, (_, _, r, m) in enumerate(self.spheres):
            rough = np.where(obj == 2 + i, r, rough)
            mat = np.where(obj == 2 + i, m, mat)

        # motion: previous-frame projection of the same (static) world point
        pv_prev = (p - pos_p) @ rot_p.T
        zp = np.where(hit, pv_prev[..., 2], 1.0)
        up = 0.5 + 0.5 * (m0 * pv_prev[..., 0] / zp)
        vp = 0.5 - 0.5 * (m5 * pv_prev[..., 1] / zp)
        mv = np.zeros((h, w, 4), dtype=np.float32)
        mv[..., 0] = np.where(hit, (up - uu) * w, 0)
        mv[..., 1] = np.where(hit, (vp - vv) * h, 0)
        mv[..., 2] = np.where(hit, zp - view_z, 0)

        # lighting ("ground truth" demodulated signals)
        shadow_o = p + n * 1e-3
        sun_dirs = np.broadcast_to(self.sun, p.shape)
        if noise:
            j = rng.standard_normal((h, w, 3)) * self.tan_sun * 0.7
            sun_dirs = _normalize(sun_dirs + j)
        ts, sobj = self._intersect(shadow_o, sun_dirs)
        ndl = np.clip((n * self.sun).sum(-1), 0, 1)
        lit = (sobj < 0) & (ndl > 0)
        glass = self.translucent_sphere & (sobj == 3)
        # ambient occlusion-like term from the distance to the nearest sphere
        dmin = np.full((h, w), 10.0)
        for c, r, _, _ in self.spheres:
            dmin = np.minimum(dmin, np.abs(np.sqrt(((p - np.asarray(c)) ** 2).sum(-1)) - r))
        ao = np.clip(0.35 + 0.65 * dmin / 1.5, 0, 1)
        sky_irr = np.stack([0.45, 0.55, 0.75]) * ao[..., None] * (0.6 + 0.4 * np.clip(n[..., 1:2], 0, 1))
        diff = sky_irr  # sun goes through SIGMA, not the diffuse signal
        refl = dn - 2 * (dn * n).sum(-1, keepdims=True) * n
        tr, robj = self._intersect(shadow_o, refl)
        spec = self._env(refl) * np.where(robj[..., None] >= 0, 0.35, 1.0)
        diff_hit = np.clip(dmin * 1.5 + 0.2, 0.05, 8.0)
        spec_hit = np.where(robj >= 0, np.minimum(tr, 50.0), 50.0)
        if noise:
            sigma = 1.0
            diff = diff * np.exp(sigma * rng.standard_normal((h, w, 1)) - 0.5 * sigma * sigma)
            sig_s = 0.3 + 0.9 * rough[..., None]
            spec = spec * np.exp(sig_s * rng.standard_normal((h, w, 1)) - 0.5 * sig_s * sig_s)
            diff_hit = diff_hit * np.exp(0.5 * rng.standard_normal((h, w)))
            spec_hit = spec_hit * np.exp(0.3 * rough * rng.standard_normal((h, w)))

        out = {}
        out["viewz"] = view_z
        out["mv"] = mv.astype(np.float16)
        out["normal_roughness"] = pack_normal_roughness(n, rough, mat)
        if self.relax:
            d4 = np.concatenate([np.minimum(diff, FP16_MAX), diff_hit[..., None]], -1)
            s4 = np.concatenate([np.minimum(spec, FP16_MAX), spec_hit[..., None]], -1)
        else:
            nd = np.clip(diff_hit / reblur_hitdist_norm(view_z, 1.0), 0, 1)
            ns = np.clip(spec_hit / reblur_hitdist_norm(view_z, rough), 0, 1)
            d4 = np.concatenate([linear_to_ycocg(diff), nd[..., None]], -1)
            s4 = np.concatenate([linear_to_ycocg(spec), ns[..., None]], -1)
        d4[~hit] = 0
        s4[~hit] = 0
        out["diff"] = d4.astype(np.float16)
        out["spec"] = s4.astype(np.float16)
        pen = np.where(lit, FP16_MAX, np.minimum(np.where(np.isfinite(ts), ts, 0.0) * self.tan_sun, 1000.0))
        pen = np.where(ndl > 0, pen, 0.0)  # back-facing: fully shadowed at distance 0
        pen[~hit] = FP16_MAX
        out["penumbra"] = pen.astype(np.float16)
        tl = np.zeros((h, w, 4), dtype=np.uint8)
        tl[..., 0] = np.where(lit, 255, 0)
        tint = np.array([230, 150, 80], dtype=np.uint8)
        tl[..., 1:] = np.where((glass & ~lit)[..., None], tint, 0)
        out["translucency"] = tl
        cw, ch = (w + 4) // 5, (h + 4) // 5
        conf = np.zeros((ch, cw, 4), dtype=np.float16)
        conf[..., 0] = 1.0
        out["confidence"] = conf
        comp = ycocg_to_linear(d4[..., :3]) * 0.8 + ycocg_to_linear(s4[..., :3]) * 0.2 if not self.relax else d4[..., :3] * 0.8 + s4[..., :3] * 0.2
        out["signal"] = np.concatenate([comp, np.ones((h, w, 1))], -1).astype(np.float16)
        out["world_to_view"], out["world_to_view_prev"] = w2v, w2v_prev
        out["view_to_clip"] = self.proj
        out["clean_diff"] = sky_irr.astype(np.float32)
        return out

    def common_settings(self, api, frame_data, index, reset=False):
        """nrd::CommonSettings exactly as Sample::RenderFrame fills it (Source/NRDSample.cpp:3835-3876)."""
        cs = api.CommonSettings()
        for i in range(16):
            cs.viewToClipMatrix[i] = cs.viewToClipMatrixPrev[i] = float(frame_data["view_to_clip"][i])
            cs.worldToViewMatrix[i] = float(frame_data["world_to_view"][i])
            cs.worldToViewMatrixPrev[i] = float(frame_data["world_to_view_prev"][i])
        cs.motionVectorScale[0] = 1.0 / self.w
        cs.motionVectorScale[1] = 1.0 / self.h
        cs.motionVectorScale[2] = 1.0
        for k in ("resourceSize", "resourceSizePrev", "rectSize", "rectSizePrev"):
            getattr(cs, k)[0] = self.w
            getattr(cs, k)[1] = self.h
        cs.viewZScale = 1.0
        cs.denoisingRange = 4.0 * self.scene_radius
        cs.disocclusionThreshold = 0.01
        cs.disocclusionThresholdAlternate = 0.1
        cs.splitScreen = 0.0
        cs.frameIndex = index
        cs.accumulationMode = int(api.AccumulationMode.CLEAR_AND_RESTART if reset else api.AccumulationMode.CONTINUE)
        cs.isHistoryConfidenceAvailable = True
        return cs
