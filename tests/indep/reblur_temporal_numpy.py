"""TEST INFRASTRUCTURE - INDEPENDENT restatement (numpy, float64, vectorised; shares no code with oracle/orc_math.h or the kernels) of
the temporal half of REBLUR_DIFFUSE_SPECULAR: TemporalAccumulation (surface-motion and virtual-motion footprints with their
occlusion tests, accumulation speeds, confidence), HistoryFix (sparse 5x5 reconstruction of short histories, fast-history clamp, the
tap texels it hands to the Blur) and TemporalStabilization (5x5 luma moments, stabilized-luma history through the footprints
TemporalAccumulation recorded, antilag). Exact 1/x, sqrt, exp2, atan and pow - none of the oracle's software reciprocals or
polynomials. tests/test_independent.py feeds every pass the planes the oracle fed its own pass and holds the oracle's outputs to
<= 1 fp16 ULP on >= 99 % of the values (the rest: thresholds decided the other way in float64 than in float32).
What it is written from: DESIGN.md section 4, SURVEY.md 8a-5 and the data contract at the sample's call sites
(Source/NRDSample.cpp:3835-3876 CommonSettings, Shaders/Shared.hlsli:318-335 2.5-D motion vectors)."""
import numpy as np

import reblur_numpy as sp  # the spatial half: guide decode, smoothstep, curves

MAX_ACCUM = 63.0
PREV_NORMAL_COS = 0.7
NORMAL_ANGLE_MIN = 0.02


def f16(a):
    return np.clip(a, -65504.0, 65504.0).astype(np.float16)


def mat4(a):
    return np.asarray(a, np.float64).reshape(4, 4).T  # column-major storage -> the matrix


class Consts:
    """per-frame constants of a perspective frame without jitter, rect = resource"""

    def __init__(self, fr, W, H, denoising_range, disocclusion_threshold):
        M = np.asarray(fr["view_to_clip"], np.float64)
        s = 1.0 if M[11] > 0 else -1.0
        self.W, self.H = W, H
        self.pj = np.array([M[0], M[5], M[8], M[9], s])
        self.fr = np.array([(-s - M[8]) / M[0], (s - M[9]) / M[5], 2.0 * s / M[0], -2.0 * s / M[5]])
        self.pv = np.array([self.fr[0] + 0.5 * self.fr[2] / W, self.fr[1] + 0.5 * self.fr[3] / H, self.fr[2] / W, self.fr[3] / H])
        cur, prev = mat4(fr["world_to_view"]), mat4(fr["world_to_view_prev"])
        self.w2v, self.w2v_prev = cur[:3, :3], prev[:3, :3]
        pos, pos_prev = -self.w2v.T @ cur[:3, 3], -self.w2v_prev.T @ prev[:3, 3]
        self.cam_delta = pos_prev - pos
        self.unproject = 1.0 / (0.5 * H * abs(self.pj[1]))
        self.min_dim_unproject = min(W, H) * self.unproject
        self.range = denoising_range
        self.disocclusion = disocclusion_threshold
        self.mv_scale = np.array([1.0 / W, 1.0 / H, 1.0])

    def reconstruct_px(self, x, y, z):
        return np.stack([z * (self.pv[2] * x + self.pv[0]), z * (self.pv[3] * y + self.pv[1]), z], -1)

    def reconstruct_uv(self, u, v, z):
        return np.stack([z * (u * self.fr[2] + self.fr[0]), z * (v * self.fr[3] + self.fr[1]), z], -1)

    def project(self, X):
        cw = self.pj[4] * X[..., 2]
        ok = cw > 1e-6
        cws = np.where(ok, cw, 1.0)
        u = 0.5 + 0.5 * ((self.pj[0] * X[..., 0] + self.pj[2] * X[..., 2]) / cws)
        v = 0.5 - 0.5 * ((self.pj[1] * X[..., 1] + self.pj[3] * X[..., 2]) / cws)
        return ok, u, v


def reproject(c, Xv, u, v, mv):
    """2.5-D motion vectors: uv offset in mv.xy, view-z offset in mv.z; returns previous uv / depth / view and world positions"""
    m = mv.astype(np.float64)[..., :3] * c.mv_scale
    su, sv = u + m[..., 0], v + m[..., 1]
    z_prev = Xv[..., 2] + m[..., 2]
    Xv_prev = c.reconstruct_uv(su, sv, z_prev)
    Xw = Xv @ c.w2v  # == v2w . Xv (v2w = w2v^T)
    Xw_prev = Xv_prev @ c.w2v_prev + c.cam_delta
    return dict(su=su, sv=sv, z_prev=z_prev, Xv_prev=Xv_prev, Xw=Xw, Xw_prev=Xw_prev)


def footprint(c, pu, pv, Nv_prev, Xv_prev, N, mat, min_mat, threshold, gprev):
    """bilinear footprint at previous uv (pu, pv) with the per-texel occlusion test against the plane through Xv_prev"""
    zp, np_, _, matp = gprev
    px, py = pu * c.W - 0.5, pv * c.H - 0.5
    fx0, fy0 = np.floor(px), np.floor(py)
    fx, fy = px - fx0, py - fy0
    sane = (fx0 >= -2.0) & (fx0 <= c.W + 1.0) & (fy0 >= -2.0) & (fy0 <= c.H + 1.0)
    ix, iy = np.where(sane, fx0, -4).astype(np.int64), np.where(sane, fy0, -4).astype(np.int64)
    bw = [(1 - fx) * (1 - fy), fx * (1 - fy), (1 - fx) * fy, fx * fy]
    plane_ref = (Nv_prev * Xv_prev).sum(-1)
    g0 = Nv_prev[..., 0] * c.pv[0] + Nv_prev[..., 1] * c.pv[1] + Nv_prev[..., 2]
    gx, gy = Nv_prev[..., 0] * c.pv[2], Nv_prev[..., 1] * c.pv[3]
    w, bits, wsum = [], np.zeros(pu.shape, np.int64), np.zeros(pu.shape)
    for i in range(4):
        tx, ty = ix + (i & 1), iy + (i >> 1)
        inside = sane & (tx >= 0) & (tx < c.W) & (ty >= 0) & (ty < c.H)
        cx, cy = np.clip(tx, 0, c.W - 1), np.clip(ty, 0, c.H - 1)
        z_t = zp[cy, cx]
        plane = z_t * (gx * tx + gy * ty + g0)
        ok = inside & (np.abs(z_t) <= c.range) & (np.abs(plane - plane_ref) <= threshold) & ((N * np_[cy, cx]).sum(-1) > PREV_NORMAL_COS)
        mp = matp[cy, cx]
        ok &= ~((mat != mp) & (np.maximum(mat, mp) >= min_mat))
        wi = np.where(ok, bw[i], 0.0)
        w.append(wi)
        wsum = wsum + wi
        bits |= np.where(ok, 1 << i, 0)
    return dict(ix=ix, iy=iy, w=w, wsum=wsum, bits=bits, bw=bw, sane=sane)


def fetch(c, plane, f):
    """footprint-weighted mean of plane [H, W, ...] (float64)"""
    acc = 0.0
    for i in range(4):
        cx, cy = np.clip(f["ix"] + (i & 1), 0, c.W - 1), np.clip(f["iy"] + (i >> 1), 0, c.H - 1)
        wi = f["w"][i]
        acc = acc + plane[cy, cx] * (wi[..., None] if plane.ndim == 3 else wi)
    ws = np.where(f["wsum"] > 0, f["wsum"], 1.0)
    return acc / (ws[..., None] if plane.ndim == 3 else ws)


def sample_confidence(conf, u, v):
    ch, cw = conf.shape[:2]
    P = conf[..., 0].astype(np.float64)
    px, py = u * cw - 0.5, v * ch - 0.5
    x0, y0 = np.floor(px), np.floor(py)
    fx, fy = px - x0, py - y0
    x0, y0 = x0.astype(np.int64), y0.astype(np.int64)

    def at(x, y):
        return P[np.clip(y, 0, ch - 1), np.clip(x, 0, cw - 1)]
    a = at(x0, y0) + (at(x0 + 1, y0) - at(x0, y0)) * fx
    b = at(x0, y0 + 1) + (at(x0 + 1, y0 + 1) - at(x0, y0 + 1)) * fx
    return np.clip(a + (b - a) * fy, 0, 1)


def virtual_uv(c, r, hit_dist, rough):
    to_cam = sp.normalize(r["Xw"])
    Xvirt = r["Xw"] + to_cam * (hit_dist * sp.spec_dominant_factor(rough))[..., None]
    Xvirt_prev = Xvirt + (r["Xw_prev"] - r["Xw"])
    Xp = (Xvirt_prev - c.cam_delta) @ c.w2v_prev.T
    return c.project(Xp)


def spec_accum_limit(rough, NoV, parallax):
    a = np.sqrt(np.clip(1.0 - NoV * 0.99999, 0, 1))
    b = rough * rough + 1.1
    power = 1.0 + 2.0 * ((b + a) / (b - a)) * parallax
    r01 = np.clip(rough, 0, 1)
    f = (1.0 - np.exp2(-200.0 * rough * rough)) * np.where(r01 > 0, np.power(np.maximum(r01, 1e-300), 0.5 * power), 0.0)
    return MAX_ACCUM * f


def unpack_speeds(d1):
    d1 = d1.astype(np.int64)
    return (d1 & 255) * 0.25, (d1 >> 8) * 0.25


def pack_speeds(a_diff, a_spec):
    q = lambda a: np.floor(np.clip(a, 0, MAX_ACCUM) * 4.0 + 0.5).astype(np.int64)
    return (q(a_diff) | (q(a_spec) << 8)).astype(np.uint16)


def luma_of(v, rec709):
    """luminance of a radiance texel [..., 4]: channel 0 (YCoCg planes) or Rec.709 of linear RGB (RELAX in the default build flavour)"""
    return 0.2126 * v[..., 0] + 0.7152 * v[..., 1] + 0.0722 * v[..., 2] if rec709 else v[..., 0]


def temporal_accumulation(c, s, gcur, gprev, mv, tmp1, hist, fast_prev, speeds_prev, hit_track, conf, history_ok, relax=None):
    """tmp1 / hist: [H, W, 2, 4] fp16 (diffuse, specular); fast_prev [H, W, 2] fp16; speeds_prev [H, W] uint16; hit_track [H, W] fp16.
    Returns tmp2 [H, W, 2, 4] fp16, fast [H, W, 2] fp16, speeds uint16, data2 uint32.
    relax = dict(moments_prev [H, W, 2] fp16, max_a_spec, max_fast_spec, rec709): RELAX's TemporalAccumulation (round 6) - the same two
    footprints; per-signal history caps; the fast history and a second-moment history (returned in info["moments"]) of the LUMINANCE of the
    linear-RGB texel; bits 16..23 of data2 carry the reprojection quality of the specular history for the A-trous relaxation.
    relax["sh1_in"] / relax["sh1_hist"] [H, W, 2, 4] fp16 (the SH denoisers): the second texel of a signal follows the first - same footprints,
    same blend factors; returned in info["sh1"]"""
    H, W = c.H, c.W
    rec709 = bool(relax and relax["rec709"])
    z, n, rough, mat = gcur
    sky = ~(np.abs(z) <= c.range)
    yy, xx = np.mgrid[0:H, 0:W]
    u, v = (xx + 0.5) / W, (yy + 0.5) / H
    Xv = c.reconstruct_px(xx, yy, z)
    Nv = n @ c.w2v.T
    NoV = np.abs((Nv * -sp.normalize(Xv)).sum(-1))
    r = reproject(c, Xv, u, v, mv)
    Nv_prev = n @ c.w2v_prev.T
    threshold = c.disocclusion * c.min_dim_unproject * np.abs(r["z_prev"])
    max_a, max_fast = float(min(s["maxAccumulatedFrameNum"], 63)), float(min(s["maxFastAccumulatedFrameNum"], 63))
    max_a_s = float(min(relax["max_a_spec"], 63)) if relax else max_a
    max_fast_s = float(min(relax["max_fast_spec"], 63)) if relax else max_fast
    min_any = min(s["minMaterialForDiffuse"], s["minMaterialForSpecular"])
    smb = footprint(c, r["su"], r["sv"], Nv_prev, r["Xv_prev"], n, mat, min_any, threshold, gprev)
    smb_ok = history_ok & (smb["wsum"] > 0)
    pd, ps = unpack_speeds(speeds_prev)
    prev_d = np.where(smb_ok, np.minimum(fetch(c, pd, smb) + 1.0, max_a), 0.0)
    prev_s = np.where(smb_ok, np.minimum(fetch(c, ps, smb) + 1.0, max_a_s), 0.0)
    quality = np.where(smb_ok, smb["wsum"], 0.0)
    data2 = np.where(smb_ok, smb["bits"], 0).astype(np.int64)
    t1, hs, fp = tmp1.astype(np.float64), hist.astype(np.float64), fast_prev.astype(np.float64)
    out = np.zeros((H, W, 2, 4))
    fast = np.zeros((H, W, 2))
    moments = np.zeros((H, W, 2))
    mp = relax["moments_prev"].astype(np.float64) if relax else None
    cD = sample_confidence(conf, u, v)
    # ---- diffuse
    cin = t1[:, :, 0]
    cY = luma_of(cin, rec709)
    A = prev_d * cD
    A = A * (quality + (1.0 - quality) / (1.0 + A))
    non_lin = 1.0 / (1.0 + A)
    h = np.where(smb_ok[..., None], fetch(c, hs[:, :, 0], smb), cin)
    fh = np.where(smb_ok, fetch(c, fp[..., 0], smb), cY)
    out[:, :, 0] = h + (cin - h) * non_lin[..., None]
    sh = relax is not None and relax.get("sh1_in") is not None
    out1 = np.zeros((H, W, 2, 4))
    if sh:
        i1, h1p = relax["sh1_in"].astype(np.float64), relax["sh1_hist"].astype(np.float64)
        h1 = np.where(smb_ok[..., None], fetch(c, h1p[:, :, 0], smb), i1[:, :, 0])
        out1[:, :, 0] = h1 + (i1[:, :, 0] - h1) * non_lin[..., None]
    fast[..., 0] = fh + (cY - fh) / (1.0 + np.minimum(A, max_fast))
    if relax:
        m2 = cY * cY
        m2p = np.where(smb_ok, fetch(c, mp[..., 0], smb), m2)
        moments[..., 0] = m2p + (m2 - m2p) * non_lin
    out_d = A
    # ---- specular
    cin = t1[:, :, 1]
    cY = luma_of(cin, rec709)
    hd = hit_track.astype(np.float64)
    Xpar = (r["Xw_prev"] - c.cam_delta) @ c.w2v.T
    okp, pu, pv_ = c.project(Xpar)
    parallax = np.where(okp, np.sqrt(((pu - r["su"]) * W) ** 2 + ((pv_ - r["sv"]) * H) ** 2), 0.0)
    A_smb = np.minimum(prev_s, spec_accum_limit(rough, NoV, parallax))
    okv, vu, vv = virtual_uv(c, r, hd, rough)
    okv = okv & history_ok
    vmb = footprint(c, np.where(okv, vu, -10.0), np.where(okv, vv, -10.0), Nv_prev, r["Xv_prev"], n, mat, s["minMaterialForSpecular"], threshold, gprev)
    vmb_ok = okv & (vmb["wsum"] > 0)
    prev_rough = fetch(c, gprev[2], vmb)
    roughA = 1.0 / (0.01 + 0.99 * np.clip(rough * s["roughnessFraction"], 0, 1))
    rconf = sp.smoothstep01(1.0 - np.abs((prev_rough - rough) * roughA))
    amount = np.where(vmb_ok, sp.spec_dominant_factor(rough) * vmb["wsum"] * rconf, 0.0)
    A_vmb = np.where(vmb_ok, np.minimum(fetch(c, ps, vmb) + 1.0, max_a_s), 0.0)
    vh = np.where(vmb_ok[..., None], fetch(c, hs[:, :, 1], vmb), cin)
    vf = np.where(vmb_ok, fetch(c, fp[..., 1], vmb), cY)
    sh_ = np.where(smb_ok[..., None], fetch(c, hs[:, :, 1], smb), cin)
    sf = np.where(smb_ok, fetch(c, fp[..., 1], smb), cY)
    A_smb = np.where(smb_ok, A_smb, 0.0)
    A = A_smb + (A_vmb - A_smb) * amount
    A = A * cD  # one texture is bound to both confidence slots (Source/NRDSample.cpp:457, :462)
    q = quality + (1.0 - quality) * amount
    A = A * (q + (1.0 - q) / (1.0 + A))
    if s["responsiveRoughnessThreshold"] > 0.0:
        t = sp.smoothstep01(rough / s["responsiveRoughnessThreshold"])
        A = np.minimum(A, s["responsiveMinAccum"] + (max_a_s - s["responsiveMinAccum"]) * t)
    non_lin = 1.0 / (1.0 + A)
    h = sh_ + (vh - sh_) * amount[..., None]
    fh = sf + (vf - sf) * amount
    out[:, :, 1] = h + (cin - h) * non_lin[..., None]
    if sh:
        s1 = np.where(smb_ok[..., None], fetch(c, h1p[:, :, 1], smb), i1[:, :, 1])
        v1 = np.where(vmb_ok[..., None], fetch(c, h1p[:, :, 1], vmb), i1[:, :, 1])
        hh = s1 + (v1 - s1) * amount[..., None]
        out1[:, :, 1] = hh + (i1[:, :, 1] - hh) * non_lin[..., None]
    fast[..., 1] = fh + (cY - fh) / (1.0 + np.minimum(A, max_fast_s))
    data2 = data2 | (np.where(vmb_ok, vmb["bits"], 0) << 4) | (np.floor(np.clip(amount, 0, 1) * 255.0 + 0.5).astype(np.int64) << 8)
    if relax:
        m2 = cY * cY
        m2s = np.where(smb_ok, fetch(c, mp[..., 1], smb), m2)
        m2v = np.where(vmb_ok, fetch(c, mp[..., 1], vmb), m2)
        m2h = m2s + (m2v - m2s) * amount
        moments[..., 1] = m2h + (m2 - m2h) * non_lin
        data2 = data2 | (np.floor(np.clip(q, 0, 1) * 255.0 + 0.5).astype(np.int64) << 16)
    speeds = pack_speeds(out_d, A)
    out[sky], fast[sky], moments[sky], out1[sky] = 0.0, 0.0, 0.0, 0.0
    speeds = np.where(sky, 0, speeds).astype(np.uint16)
    data2 = np.where(sky, 0, data2).astype(np.uint32)
    return f16(out), f16(fast), speeds, data2, dict(amount=amount, smb_ok=smb_ok, vmb_ok=vmb_ok, moments=f16(moments), sh1=f16(out1))


def pixel_geo(c, z, n, sens):
    yy, xx = np.mgrid[0:c.H, 0:c.W]
    Xv = c.reconstruct_px(xx, yy, z)
    Nv = n @ c.w2v.T
    geoA = 1.0 / (sens * c.min_dim_unproject * np.abs(z))
    return dict(gax=Nv[..., 0] * c.pv[2] * geoA, gay=Nv[..., 1] * c.pv[3] * geoA, ga0=(Nv[..., 0] * c.pv[0] + Nv[..., 1] * c.pv[1] + Nv[..., 2]) * geoA,
                geoB=-(Nv * Xv).sum(-1) * geoA)


def moments5x5(c, plane, centre, z, skip_centre=False):
    """5x5 mean / second moment of plane around every pixel; texels outside the frame or beyond the denoising range count as `centre`"""
    H, W = c.H, c.W
    yy, xx = np.mgrid[0:H, 0:W]
    m1, m2 = np.zeros((H, W)), np.zeros((H, W))
    for j in range(-2, 3):
        for i in range(-2, 3):
            if skip_centre and i == 0 and j == 0:
                continue
            px, py = xx + i, yy + j
            inside = (px >= 0) & (px < W) & (py >= 0) & (py < H)
            cx, cy = np.clip(px, 0, W - 1), np.clip(py, 0, H - 1)
            f = np.where(inside & (np.abs(z[cy, cx]) <= c.range), plane[cy, cx], centre)
            m1 += f
            m2 += f * f
    k = 24.0 if skip_centre else 25.0
    return m1 / k, m2 / k


def pack_tap_guide(viewz, packed_nr, denoising_range=None):
    """guide part of a tap texel = the pixel's guide texel: viewZ rounded to 22 bits | 10-bit roughness code ; 3 x 10-bit normal | material"""
    return sp.guide_words(viewz, packed_nr, denoising_range=denoising_range)


def history_fix(c, s, gcur, tmp2, speeds_tmp, fast, viewz, packed_nr, upstream=False, relax=None):
    """returns signal [H, W, 2, 4] fp16 (what goes into the tap texels), speeds uint16, tap guide words (w0, w1).
    relax = dict(moments [H, W, 2] fp16, normal_power, accel, spatial, temporal, reset, max_fast_spec, rec709): RELAX's HistoryFix (round 6) -
    the reconstruction's normal weight is pow(N.Ns, historyFixEdgeStoppingNormalPower), the clamp works on the luminance of the texel
    (scaling r, g, b together in linear RGB), a clamped pixel accelerates its history by accelerationAmount, and a history farther from the
    fast 5x5 mean than spatialSigmaScale x spatial sigma + temporalSigmaScale x temporal sigma is reset by up to resetAmount (antilag).
    relax["sh1"] [H, W, 2, 4] fp16 (the SH denoisers): the second texel is reconstructed with the first one's weights and its xyz scaled by the
    clamp's luminance ratio; a fourth return value holds it"""
    H, W = c.H, c.W
    rec709 = bool(relax and relax["rec709"])
    z, n, rough_g, mat = gcur
    sky = ~(np.abs(z) <= c.range)
    yy, xx = np.mgrid[0:H, 0:W]
    pg = pixel_geo(c, z, n, s["planeDistanceSensitivity"])
    t2 = tmp2.astype(np.float64)
    Ad, As = unpack_speeds(speeds_tmp)
    fs = fast.astype(np.float64)
    out = np.zeros((H, W, 2, 4))
    sh1 = relax["sh1"].astype(np.float64) if (relax and relax.get("sh1") is not None) else None
    out1 = np.zeros((H, W, 2, 4))
    outA = [Ad.copy(), As.copy()]
    nfix, base = float(s["historyFixFrameNum"]), float(s["historyFixBasePixelStride"])
    max_fast = float(min(s["maxFastAccumulatedFrameNum"], 63))
    max_fast_s = float(min(relax["max_fast_spec"], 63)) if relax else max_fast
    for sig, is_spec in ((0, False), (1, True)):
        A = As if is_spec else Ad
        rough = rough_g if is_spec else np.ones_like(rough_g)
        min_mat = s["minMaterialForSpecular"] if is_spec else s["minMaterialForDiffuse"]
        val = t2[:, :, sig].copy()
        stride = np.floor(base * (1.0 - np.clip(A / max(nfix, 1e-9), 0, 1)) + 0.5).astype(np.int64)
        fix = (A < nfix) & (nfix > 0) & (stride > 0) & ~sky
        angle = np.arctan(3.0 * np.clip(rough, 0, 1) ** 2) * (s["lobeAngleFraction"] + (1.0 - s["lobeAngleFraction"]) / (1.0 + A))
        normal_w = 1.0 / np.maximum(angle, NORMAL_ANGLE_MIN)
        roughA = 1.0 / (0.01 + 0.99 * np.clip(rough * s["roughnessFraction"], 0, 1))
        acc, wsum = val * (1.0 + A)[..., None], 1.0 + A
        val1 = sh1[:, :, sig].copy() if sh1 is not None else None
        acc1 = val1 * (1.0 + A)[..., None] if sh1 is not None else None
        for j in range(-2, 3):
            for i in range(-2, 3):
                if (i == 0 and j == 0) or (abs(i) == 2 and abs(j) == 2):
                    continue
                px, py = xx + i * stride, yy + j * stride
                inside = (px >= 0) & (px < W) & (py >= 0) & (py < H)
                cx, cy = np.clip(px, 0, W - 1), np.clip(py, 0, H - 1)
                zs, ms = z[cy, cx], mat[cy, cx]
                ok = fix & inside & (np.abs(zs) <= c.range) & ~((mat != ms) & (np.maximum(mat, ms) >= min_mat))
                w = 1.0 / (1.0 + i * i + j * j)
                w = w * sp.smoothstep01(1.0 - np.abs(zs * (pg["gax"] * px + pg["gay"] * py + pg["ga0"]) + pg["geoB"]))
                if relax:
                    w = w * np.power(np.clip(sp.normal_cos(n, n[cy, cx]), 0.0, 1.0), relax["normal_power"])
                else:
                    w = w * sp.normal_weight(sp.normal_cos(n, n[cy, cx]), normal_w, upstream)  # (upstream: the default build flavour's form)
                if is_spec:
                    w = w * sp.smoothstep01(1.0 - np.abs(rough_g[cy, cx] * roughA - rough * roughA))
                tA = (As if is_spec else Ad)[cy, cx]
                w = np.where(ok, w * (1.0 + tA), 0.0)
                acc = acc + t2[cy, cx, sig] * w[..., None]
                if sh1 is not None:
                    acc1 = acc1 + sh1[cy, cx, sig] * w[..., None]
                wsum = wsum + w
        val = np.where(fix[..., None], acc / wsum[..., None], val)
        if sh1 is not None:
            val1 = np.where(fix[..., None], acc1 / wsum[..., None], val1)
        if s["maxFastAccumulatedFrameNum"] < s["maxAccumulatedFrameNum"]:
            fc = fs[..., sig]
            m1, m2 = moments5x5(c, fc, fc, z)
            sigma = np.sqrt(np.maximum(m2 - m1 * m1, 0.0)) * s["fastHistoryClampingSigmaScale"]
            Y = luma_of(val, rec709)
            Yc = np.clip(Y, m1 - sigma, m1 + sigma)
            scale = (Yc + 1e-6) / (Y + 1e-6)
            val = np.stack([val[..., 0] * scale if rec709 else Yc, val[..., 1] * scale, val[..., 2] * scale, val[..., 3]], -1)
            if sh1 is not None:
                val1 = np.concatenate([val1[..., :3] * scale[..., None], val1[..., 3:]], -1)
            f = np.clip(np.abs(Yc - Y) / np.maximum(np.maximum(Y, Yc), 1e-6), 0, 1)
            if relax:
                f = f * np.clip(relax["accel"], 0, 1)
            a_new = A + (np.minimum(A, max_fast_s if is_spec else max_fast) - A) * f
            if relax:
                sig_s = np.sqrt(np.maximum(m2 - m1 * m1, 0.0))
                sig_t = np.sqrt(np.maximum(relax["moments"].astype(np.float64)[..., sig] - Y * Y, 0.0))
                thr = relax["spatial"] * sig_s + relax["temporal"] * sig_t
                over = np.clip(np.abs(Y - m1) / np.maximum(thr, 1e-6) - 1.0, 0, 1)
                a_new = a_new * (1.0 - np.clip(relax["reset"], 0, 1) * over)
            outA[1 if is_spec else 0] = a_new
        out[:, :, sig] = val
        if sh1 is not None:
            out1[:, :, sig] = val1
    out[sky], out1[sky] = 0.0, 0.0
    speeds = np.where(sky, 0, pack_speeds(outA[0], outA[1])).astype(np.uint16)
    if sh1 is not None:
        return f16(out), speeds, pack_tap_guide(viewz, packed_nr, c.range), f16(out1)
    return f16(out), speeds, pack_tap_guide(viewz, packed_nr, c.range)


def temporal_stabilization(c, s, gcur, mv, hist, speeds, data2, stab_prev, hit_track, history_ok):
    """hist: PostBlur output [H, W, 2, 4] fp16; stab_prev [H, W, 2] fp16. Returns OUT_DIFF / OUT_SPEC [H, W, 2, 4] fp16, stab [H, W, 2] fp16"""
    H, W = c.H, c.W
    z, n, rough, mat = gcur
    sky = ~(np.abs(z) <= c.range)
    yy, xx = np.mgrid[0:H, 0:W]
    u, v = (xx + 0.5) / W, (yy + 0.5) / H
    Xv = c.reconstruct_px(xx, yy, z)
    r = reproject(c, Xv, u, v, mv)
    Ad, As = unpack_speeds(speeds)
    hs, spv = hist.astype(np.float64), stab_prev.astype(np.float64)
    d2 = data2.astype(np.int64)
    max_stab = float(min(s["maxStabilizedFrameNum"], 63))

    def fetch_stab(pu, pv, bits, sig):
        px, py = pu * W - 0.5, pv * H - 0.5
        fx0, fy0 = np.floor(px), np.floor(py)
        fx, fy = px - fx0, py - fy0
        sane = (fx0 >= -2.0) & (fx0 <= W + 1.0) & (fy0 >= -2.0) & (fy0 <= H + 1.0)
        ix, iy = np.where(sane, fx0, 0).astype(np.int64), np.where(sane, fy0, 0).astype(np.int64)
        bw = [(1 - fx) * (1 - fy), fx * (1 - fy), (1 - fx) * fy, fx * fy]
        acc, wsum = np.zeros((H, W)), np.zeros((H, W))
        for i in range(4):
            on = sane & ((bits >> i) & 1).astype(bool)
            cx, cy = np.clip(ix + (i & 1), 0, W - 1), np.clip(iy + (i >> 1), 0, H - 1)
            acc = acc + np.where(on, spv[cy, cx, sig] * bw[i], 0.0)
            wsum = wsum + np.where(on, bw[i], 0.0)
        ok = sane & (wsum > 0)
        return ok, acc / np.where(wsum > 0, wsum, 1.0)

    out, stab = np.zeros((H, W, 2, 4)), np.zeros((H, W, 2))
    for sig, is_spec in ((0, False), (1, True)):
        cur = hs[:, :, sig]
        m1, m2 = moments5x5(c, cur[..., 0], cur[..., 0], z)
        sigma = np.sqrt(np.maximum(m2 - m1 * m1, 0.0))
        smb_ok, smb_y = fetch_stab(r["su"], r["sv"], d2 & 15, sig)
        smb_ok = smb_ok & history_ok
        y_hist, have = np.where(smb_ok, smb_y, cur[..., 0]), smb_ok
        if is_spec:
            amount = ((d2 >> 8) & 255) / 255.0
            okv, vu, vv = virtual_uv(c, r, hit_track.astype(np.float64), rough)
            vok, vmb_y = fetch_stab(np.where(okv, vu, -10.0), np.where(okv, vv, -10.0), (d2 >> 4) & 15, sig)
            vok = vok & okv & (amount > 0) & history_ok
            y_hist = np.where(smb_ok & vok, smb_y + (vmb_y - smb_y) * amount, np.where(smb_ok, smb_y, np.where(vok, vmb_y, cur[..., 0])))
            have = smb_ok | vok
        A = As if is_spec else Ad
        Y = cur[..., 0]
        band = sigma * s["antilagSigmaScale"]
        dlt = np.maximum(np.abs(y_hist - m1) - band, 0.0) / (np.maximum(y_hist, m1) + 1e-6)
        antilag = 1.0 / (1.0 + dlt * s["antilagSensitivity"] * A)
        y_cl = np.clip(y_hist, m1 - band, m1 + band)
        frames = np.where(have, np.minimum(A, max_stab) * antilag, 0.0)
        w_hist = frames / (1.0 + frames)
        y_out = Y + (y_cl - Y) * w_hist
        scale = (y_out + 1e-6) / (Y + 1e-6)
        out[:, :, sig] = np.stack([y_out, cur[..., 1] * scale, cur[..., 2] * scale, cur[..., 3]], -1)
        stab[..., sig] = y_out
    out[sky], stab[sky] = 0.0, 0.0
    return f16(out), f16(stab)


# =====================================================================================================================
# RELAX: one variance-guided A-trous iteration (3x3 taps at stride 2^it), iteration 0 and the middle iterations
# =====================================================================================================================
def atrous_iteration(c, s, gcur, plane_in, it, speeds=None, moments=None, data2=None, upstream=False, sh1=None):
    """plane_in [H, W, 2, 4] fp16: iteration 0 = History {Y, Co, Cg, hitT}, later = {Y, Co, Cg, variance}; moments [H, W, 2] fp16
    (second luma moment, iteration 0 only); data2 uint32 (reprojection confidence of the specular history in bits 16..23).
    Returns the ping-pong texel {Y, Co, Cg, variance} as fp16.
    upstream = True: the DEFAULT build flavour - the texels are linear {r, g, b, .} and a luminance is Rec.709 of them, the luminance weight
    is exp(-3 x), the normal weight sits on the chord of the two normals.
    sh1 [H, W, 2, 4] fp16 (the SH denoisers): filtered with the weights of the first texel; returned as a second value"""
    H, W = c.H, c.W
    p1 = sh1.astype(np.float64) if sh1 is not None else None
    out1 = np.zeros((H, W, 2, 4))
    luma = (lambda v: 0.2126 * v[..., 0] + 0.7152 * v[..., 1] + 0.0722 * v[..., 2]) if upstream else (lambda v: v[..., 0])
    z, n, rough_g, mat = gcur
    sky = ~(np.abs(z) <= c.range)
    yy, xx = np.mgrid[0:H, 0:W]
    stride = 1 << it
    relax_edges = stride <= 4
    pg = pixel_geo(c, z, n, max(s["depthThreshold"], 0.001) * 4.0)
    pin = plane_in.astype(np.float64)
    Ad, As = unpack_speeds(speeds) if speeds is not None else (None, None)
    out = np.zeros((H, W, 2, 4))

    def variance0(sig):
        Y = luma(pin[:, :, sig])
        var = np.maximum(moments.astype(np.float64)[..., sig] - Y * Y, 0.0)
        A = As if sig == 1 else Ad
        sy, sy2, cnt = np.zeros((H, W)), np.zeros((H, W)), np.zeros((H, W))
        for j in (-1, 0, 1):
            for i in (-1, 0, 1):
                px, py = xx + i, yy + j
                ok = (px >= 0) & (px < W) & (py >= 0) & (py < H)
                cx, cy = np.clip(px, 0, W - 1), np.clip(py, 0, H - 1)
                ok &= np.abs(z[cy, cx]) <= c.range
                Yt = np.where(ok, Y[cy, cx], 0.0)
                sy, sy2, cnt = sy + Yt, sy2 + Yt * Yt, cnt + ok
        cnt = np.maximum(cnt, 1.0)
        spatial = np.maximum(sy2 / cnt - (sy / cnt) ** 2, 0.0)
        var = np.where(A < s["spatialVarianceEstimationHistoryThreshold"], np.maximum(var, spatial), var)
        return var * (1.0 + s["specularVarianceBoost"]) if sig == 1 else var

    for sig, is_spec in ((0, False), (1, True)):
        rough = rough_g if is_spec else np.ones_like(rough_g)
        min_mat = s["minMaterialForSpecular"] if is_spec else s["minMaterialForDiffuse"]
        c0 = pin[:, :, sig]
        var = variance0(sig) if it == 0 else c0[..., 3]
        # what a TAP contributes as its variance: its texel's variance channel; in iteration 0 the tap's temporal variance alone (the
        # spatial estimate and the specular boost belong to the centre pixel)
        var_all = np.maximum(moments.astype(np.float64)[..., sig] - luma(c0) ** 2, 0.0) if it == 0 else var
        sigma = np.sqrt(var)
        phi = s["specularPhiLuminance"] if is_spec else s["diffusePhiLuminance"]
        min_lw = s["specularMinLuminanceWeight"] if is_spec else s["diffuseMinLuminanceWeight"]
        inv_l = 0.3333 / (phi * sigma + 1e-4)
        angle = np.arctan(3.0 * np.clip(rough, 0, 1) ** 2) * s["lobeAngleFraction"]
        if is_spec:
            angle = angle + s["specularLobeAngleSlack"] * 0.017453292
        normal_w = 1.0 / np.maximum(angle, NORMAL_ANGLE_MIN)
        rough_relax = 1.0
        if is_spec and relax_edges:
            conf = ((data2.astype(np.int64) >> 16) & 255) / 255.0
            inv_l = inv_l * (1.0 + (conf - 1.0) * np.clip(s["luminanceEdgeStoppingRelaxation"], 0, 1))
            normal_w = normal_w * (1.0 + (conf - 1.0) * np.clip(s["normalEdgeStoppingRelaxation"], 0, 1))
            rough_relax = 1.0 + (conf - 1.0) * np.clip(s["roughnessEdgeStoppingRelaxation"], 0, 1)
        roughA = 1.0 / (0.01 + 0.99 * np.clip(rough * s["roughnessFraction"], 0, 1))
        acc, acc_var, wsum = c0[..., :3].copy(), var.copy(), np.ones((H, W))
        acc1 = p1[:, :, sig].copy() if p1 is not None else None
        for j in (-1, 0, 1):
            for i in (-1, 0, 1):
                if i == 0 and j == 0:
                    continue
                px, py = xx + i * stride, yy + j * stride
                inside = (px >= 0) & (px < W) & (py >= 0) & (py < H)
                cx, cy = np.clip(px, 0, W - 1), np.clip(py, 0, H - 1)
                zs, ms = z[cy, cx], mat[cy, cx]
                ok = inside & (np.abs(zs) <= c.range) & ~((mat != ms) & (np.maximum(mat, ms) >= min_mat))
                sv = pin[cy, cx, sig]
                w = 0.5 if (i == 0 or j == 0) else 0.25
                w = w * sp.smoothstep01(1.0 - np.abs(zs * (pg["gax"] * px + pg["gay"] * py + pg["ga0"]) + pg["geoB"]))
                w = w * sp.normal_weight(sp.normal_cos(n, n[cy, cx]), normal_w, upstream)
                if is_spec and s["enableRoughnessEdgeStopping"]:
                    rw = sp.smoothstep01(1.0 - np.abs(rough_g[cy, cx] * roughA - rough * roughA))
                    w = w * ((1.0 + (rw - 1.0) * rough_relax) if relax_edges else rw)
                dl = np.abs(luma(sv) - luma(c0)) * inv_l
                w = w * np.maximum(np.exp(-3.0 * dl) if upstream else np.clip(1.0 - dl, 0, 1) ** 2, min_lw)
                w = np.where(ok, w, 0.0)
                acc = acc + sv[..., :3] * w[..., None]
                if p1 is not None:
                    acc1 = acc1 + p1[cy, cx, sig] * w[..., None]
                acc_var = acc_var + var_all[cy, cx] * w * w
                wsum = wsum + w
        out[:, :, sig, :3] = acc / wsum[..., None]
        out[:, :, sig, 3] = acc_var / (wsum * wsum)
        if p1 is not None:
            out1[:, :, sig] = acc1 / wsum[..., None]
    out[sky], out1[sky] = 0.0, 0.0
    return (f16(out), f16(out1)) if p1 is not None else f16(out)


# =====================================================================================================================
# SIGMA_SHADOW_TRANSLUCENCY: Blur / PostBlur and TemporalStabilization
# =====================================================================================================================
SIGMA_MAX_PIXEL_RADIUS, SIGMA_BLUR_REACH, SIGMA_STAB_SIGMA_SCALE = 48.0, 56, 2.0


def sigma_classify_tiles(c, z, pen, tile=16):
    """SIGMA ClassifyTiles (round 6): per 16 x 16 tile - bit 0: some pixel with geometry lies in penumbra (a finite IN_PENUMBRA), bit 1: some
    pixel with geometry is lit (IN_PENUMBRA = fp16 max, the sample's "no occluder" code, Shaders/TraceOpaque.cs.hlsl:800-801), bits 8..15:
    the largest penumbra radius of the tile in pixels (world size / pixel size at the depth), rounded up, capped at 255.
    z [H, W] view depth, pen [H, W] fp16"""
    H, W = c.H, c.W
    ty, tx = (H + tile - 1) // tile, (W + tile - 1) // tile
    geo = np.abs(z) <= c.range
    p = pen.astype(np.float64)
    lit = p >= 65504.0
    r_px = np.minimum(p / (c.unproject * np.maximum(np.abs(z), 1e-300)), 255.0)
    out = np.zeros((ty, tx), np.uint16)
    for j in range(ty):
        for i in range(tx):
            sl = (slice(j * tile, min((j + 1) * tile, H)), slice(i * tile, min((i + 1) * tile, W)))
            g, l = geo[sl], lit[sl]
            flags = (1 if (g & ~l).any() else 0) | (2 if (g & l).any() else 0)
            m = float(r_px[sl][g & ~l].max()) if (g & ~l).any() else 0.0
            out[j, i] = flags | (min(int(np.floor(m + 0.999)), 255) << 8)
    return out


def sigma_smooth_tiles(tiles):
    """SIGMA SmoothTiles: a tile is filtered (bit 0) when its 3 x 3 tile neighbourhood holds BOTH penumbra and lit pixels - a shadow edge
    passes through or next to it; its radius is the largest of the neighbourhood"""
    ty, tx = tiles.shape
    t = tiles.astype(np.int64)
    out = np.zeros_like(tiles)
    for j in range(ty):
        for i in range(tx):
            nb = t[max(j - 1, 0):j + 2, max(i - 1, 0):i + 2]
            flags = int(np.bitwise_or.reduce((nb & 3).ravel()))
            out[j, i] = (1 if flags == 3 else 0) | (int((nb >> 8).max()) << 8)
    return out


def sigma_input_visibility(pen, transl):
    """raw input texel -> visibility: lit -> 1, shadowed -> (0, translucency.yzw)"""
    lit = pen >= 65504.0
    t = transl.astype(np.float64) / 255.0
    v = np.stack([np.zeros_like(pen), t[..., 1], t[..., 2], t[..., 3]], -1)
    return np.where(lit[..., None], 1.0, v)


def sigma_blur(c, s, z, n, tiles_smooth, pen_in, vis_in, frame_index, pass_index):
    """pass 0 (Blur): pen_in = IN_PENUMBRA, vis_in = input visibility; pass 1 (PostBlur): pen_in = Penumbra1, vis_in = Shadow1.
    Returns (shadow [H, W, 4] fp16, penumbra [H, W] fp16 - meaningful for pass 0)"""
    H, W = c.H, c.W
    yy, xx = np.mgrid[0:H, 0:W]
    sky = ~(np.abs(z) <= c.range)
    absz = np.abs(z)
    pen = pen_in.astype(np.float64)
    lit = (pen >= 65504.0) if pass_index == 0 else ~(pen > 0.0)
    centre = vis_in.astype(np.float64)
    tile = tiles_smooth.astype(np.int64)[yy // 16, xx // 16]
    mixed = (tile & 1) != 0
    pixel_world = c.unproject * absz
    radius = np.minimum(np.where(lit, (tile >> 8).astype(np.float64), pen / pixel_world), SIGMA_MAX_PIXEL_RADIUS)
    world_radius = radius * pixel_world
    Xv = c.reconstruct_px(xx, yy, z)
    Nv = n @ c.w2v.T
    geoA = 1.0 / (s["planeDistanceSensitivity"] * c.min_dim_unproject * absz)
    gax, gay = Nv[..., 0] * c.pv[2] * geoA, Nv[..., 1] * c.pv[3] * geoA
    ga0 = (Nv[..., 0] * c.pv[0] + Nv[..., 1] * c.pv[1] + Nv[..., 2]) * geoA
    geoB = -(Nv * Xv).sum(-1) * geoA
    T, B = sp.basis(Nv)
    T, B = T * world_radius[..., None], B * world_radius[..., None]
    inv = 1.0 / (c.pj[4] * z)
    nu = (c.pj[0] * Xv[..., 0] + c.pj[2] * z) * inv
    nv_ = (c.pj[1] * Xv[..., 1] + c.pj[3] * z) * inv
    kuz, kvz = c.pj[2] - nu * c.pj[4], c.pj[3] - nv_ * c.pj[4]
    ju, jv = 0.5 * W * inv, -0.5 * H * inv
    jtx, jty = ju * (c.pj[0] * T[..., 0] + kuz * T[..., 2]), jv * (c.pj[1] * T[..., 1] + kvz * T[..., 2])
    jbx, jby = ju * (c.pj[0] * B[..., 0] + kuz * B[..., 2]), jv * (c.pj[1] * B[..., 1] + kvz * B[..., 2])
    if pass_index == 0:
        k = sp.hash_px(xx.astype(np.int64), yy.astype(np.int64), frame_index, 17) & 63
    else:
        k = np.full((H, W), sp.hash_px(0, 0, frame_index, 18) & 63, np.int64)
    ang = 2.0 * np.pi * k / 64.0
    rc, rs = np.cos(ang).astype(np.float32).astype(np.float64), np.sin(ang).astype(np.float32).astype(np.float64)
    jtx, jbx = rc * jtx + rs * jbx, rc * jbx - rs * jtx
    jty, jby = rc * jty + rs * jby, rc * jby - rs * jty
    acc, wsum = centre.copy(), np.ones((H, W))
    pen_sum, pen_w = np.where(lit, 0.0, pen), np.where(lit, 0.0, 1.0)
    for t in range(8):
        ox, oy, pw = sp.POISSON8[t]
        fpx, fpy = np.floor(ox * jtx + oy * jbx + xx + 0.5), np.floor(ox * jty + oy * jby + yy + 0.5)
        ok = (radius > 0) & (fpx >= 0) & (fpx < W) & (fpy >= 0) & (fpy < H) & (np.abs(fpx - xx) <= SIGMA_BLUR_REACH) & (np.abs(fpy - yy) <= SIGMA_BLUR_REACH)
        px, py = np.clip(fpx, 0, W - 1).astype(np.int64), np.clip(fpy, 0, H - 1).astype(np.int64)
        zs = z[py, px]
        ok &= np.abs(zs) <= c.range
        w = pw * sp.smoothstep01(1.0 - np.abs(zs * (gax * fpx + gay * fpy + ga0) + geoB))
        w = np.where(ok, w, 0.0)
        ps = pen[py, px]
        lits = (ps >= 65504.0) if pass_index == 0 else ~(ps > 0.0)
        acc = acc + centre[py, px] * w[..., None]
        wsum = wsum + w
        pen_sum = pen_sum + np.where(lits, 0.0, ps * w)
        pen_w = pen_w + np.where(lits, 0.0, w)
    shadow = np.where(mixed[..., None], acc / wsum[..., None], centre)
    pen_out = np.where(mixed, np.where(pen_w > 0, pen_sum / np.where(pen_w > 0, pen_w, 1.0), 0.0), np.where(lit, 0.0, pen))
    shadow[sky], pen_out = 0.0, np.where(sky, 0.0, pen_out)
    return f16(shadow), f16(pen_out)


def sigma_encode(v):
    q = np.floor(np.sqrt(np.clip(v, 0, 1)) * 255.0 + 0.5).astype(np.uint32)
    return q[..., 0] | (q[..., 1] << 8) | (q[..., 2] << 16) | (q[..., 3] << 24)


def sigma_decode(p):
    b = np.stack([(p >> (8 * i)) & 255 for i in range(4)], -1).astype(np.float64) / 255.0
    return b * b


def sigma_temporal_stabilization(c, s, gcur, gprev, mv, shadow2, tiles_smooth, hist_prev, history_ok):
    """returns the packed RGBA8 (sqrt-encoded) history / output [H, W] uint32"""
    H, W = c.H, c.W
    z, n, _, _ = gcur
    zp, np_, _, _ = gprev
    yy, xx = np.mgrid[0:H, 0:W]
    u, v = (xx + 0.5) / W, (yy + 0.5) / H
    sky = ~(np.abs(z) <= c.range)
    cur = shadow2.astype(np.float64)
    mixed = (tiles_smooth.astype(np.int64)[yy // 16, xx // 16] & 1) != 0
    m1, m2 = np.zeros((H, W, 4)), np.zeros((H, W, 4))
    for j in range(-2, 3):
        for i in range(-2, 3):
            px, py = xx + i, yy + j
            inside = (px >= 0) & (px < W) & (py >= 0) & (py < H)
            cx, cy = np.clip(px, 0, W - 1), np.clip(py, 0, H - 1)
            f = np.where((inside & (np.abs(z[cy, cx]) <= c.range))[..., None], cur[cy, cx], cur)
            m1, m2 = m1 + f, m2 + f * f
    m1, m2 = m1 / 25.0, m2 / 25.0
    Xv = c.reconstruct_px(xx, yy, z)
    m = mv.astype(np.float64)[..., :3] * c.mv_scale
    su, sv = u + m[..., 0], v + m[..., 1]
    Xv_prev = c.reconstruct_uv(su, sv, z + m[..., 2])
    Nv_prev = n @ c.w2v_prev.T
    threshold = c.disocclusion * c.min_dim_unproject * np.abs(Xv_prev[..., 2])
    px, py = su * W - 0.5, sv * H - 0.5
    fx0, fy0 = np.floor(px), np.floor(py)
    fx, fy = px - fx0, py - fy0
    sane = (fx0 >= -2.0) & (fx0 <= W + 1.0) & (fy0 >= -2.0) & (fy0 <= H + 1.0)
    ix, iy = np.where(sane, fx0, 0).astype(np.int64), np.where(sane, fy0, 0).astype(np.int64)
    bw = [(1 - fx) * (1 - fy), fx * (1 - fy), (1 - fx) * fy, fx * fy]
    plane_ref = (Nv_prev * Xv_prev).sum(-1)
    g0 = Nv_prev[..., 0] * c.pv[0] + Nv_prev[..., 1] * c.pv[1] + Nv_prev[..., 2]
    gx, gy = Nv_prev[..., 0] * c.pv[2], Nv_prev[..., 1] * c.pv[3]
    hp = sigma_decode(hist_prev.astype(np.uint32))
    acc, wsum = np.zeros((H, W, 4)), np.zeros((H, W))
    for i in range(4):
        tx, ty = ix + (i & 1), iy + (i >> 1)
        ok = sane & (tx >= 0) & (tx < W) & (ty >= 0) & (ty < H)
        cx, cy = np.clip(tx, 0, W - 1), np.clip(ty, 0, H - 1)
        zt = zp[cy, cx]
        ok &= (np.abs(zt) <= c.range) & (np.abs(zt * (gx * tx + gy * ty + g0) - plane_ref) <= threshold) & ((n * np_[cy, cx]).sum(-1) > PREV_NORMAL_COS)
        acc = acc + np.where(ok[..., None], hp[cy, cx] * bw[i][..., None], 0.0)
        wsum = wsum + np.where(ok, bw[i], 0.0)
    have = history_ok & (wsum > 0)
    hist = np.where(have[..., None], acc / np.where(wsum > 0, wsum, 1.0)[..., None], cur)
    max_stab = float(min(s["maxStabilizedFrameNum"], 7))
    w = np.where(have, max_stab / (1.0 + max_stab), 0.0)
    sigma = np.sqrt(np.maximum(m2 - m1 * m1, 0.0)) * SIGMA_STAB_SIGMA_SCALE
    o = cur + (np.clip(hist, m1 - sigma, m1 + sigma) - cur) * w[..., None]
    packed = np.where(mixed, sigma_encode(o), sigma_encode(cur))
    return np.where(sky, 0, packed).astype(np.uint32)
