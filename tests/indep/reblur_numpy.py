"""TEST INFRASTRUCTURE - an INDEPENDENT restatement (numpy, float64, vectorised; shares no code with oracle/orc_math.h or the
kernels) of three pieces of the frozen algorithm: REFERENCE accumulation, the guide decode of ClassifyTiles and the REBLUR PrePass
(8-tap Poisson bilateral filter of both signals). tests/test_independent.py holds the oracle to it at <= 1 fp16 ULP, and uses its
switches to MEASURE the known deviations from the recalled upstream formulas listed in oracle/README.md ("deviation ledger").

Switches (all False = the frozen algorithm):
  exp_hit_weight ...... hit-distance weight exp(-3 |x|) instead of the compact-support stand-in (1 - |x|)^2
  angle_normal_weight . normal weight on the angle itself, smoothstep(1 - AcosApprox(cos) / angleMax) with upstream's AcosApprox =
                        sqrt(2 (1 - cos)) (the chord), instead of the squared-angle form
  no_reach ............ no hard tap reach (the reach exists for row tiling: it bounds what a pass may read beyond a band)
  f32_guide ........... guide kept at decode precision (fp32 depth, normalised fp64 normal) instead of the 8-byte guide texel's 22-bit depth
                        and 3 x 10-bit normal codes
Measurement-only switches on top of the DEFAULT flavour (exp_hit_weight = angle_normal_weight = True), ledger rows 16-18 - what the default
build's cheap evaluations move against the exact functions this restatement otherwise takes:
  one_step_sqrt ....... the chord's square root as the build evaluates it: magic seed + ONE tuned Newton step (relative error <= 6.5e-4)
  poly3_exp2 .......... the hit-distance weight's exponential as the build evaluates it: round-to-nearest split + degree-3 polynomial (8.0e-5)
  arc_normal_weight ... the true angle acos(cos) instead of the chord sqrt(2 (1 - cos)) (upstream's AcosApprox IS the chord: this row measures
                        what the substitution the recalled shader makes is worth, not a deviation of this build)
"""
import numpy as np

POISSON8 = np.array([[-0.4706069, -0.4427112, 0.7592], [-0.9057375, 0.3003471, 0.5483], [-0.3487388, 0.4037880, 0.8287], [0.1023042, 0.6439373, 0.7554],
                     [0.5699277, 0.3513750, 0.7439], [0.2939128, -0.1131226, 0.9366], [0.7836658, -0.4208784, 0.5932], [0.1564120, -0.8198990, 0.6314]], np.float32).astype(np.float64)
NORMAL_ANGLE_MIN = 0.02


def f16(a):
    return np.clip(a, -65504.0, 65504.0).astype(np.float16)


def hash_px(x, y, frame, salt):
    m = 0xFFFFFFFF
    h = ((x * 73856093) & m) ^ ((y * 19349663) & m) ^ ((frame * 83492791) & m) ^ ((salt * 2654435761) & m)
    h ^= h >> 13
    h = (h * 0x5BD1E995) & m
    h ^= h >> 15
    return h


def smoothstep01(x):
    x = np.clip(x, 0.0, 1.0)
    return x * x * (3.0 - 2.0 * x)


GUIDE_SKY_DEPTH = 0x7F7FFC00  # canonical stored depth of a pixel without geometry (csrc/nrd_device.h encode_guide)


def decode_guide(viewz, packed, viewz_scale=1.0, f32_guide=False, denoising_range=None):
    """ClassifyTiles: viewZ * scale; octahedral normal (10 + 10 bits), linear roughness (10 bits), materialID (2 bits); the guide
    plane stores {depth 22 bit | roughness code, normal 3 x 10 bit | material} and every consumer reads THAT"""
    ox, oy = (packed & 1023) / 1023.0, ((packed >> 10) & 1023) / 1023.0
    fx, fy = ox * 2.0 - 1.0, oy * 2.0 - 1.0
    nz = 1.0 - np.abs(fx) - np.abs(fy)
    t = np.clip(-nz, 0.0, 1.0)
    n = np.stack([fx + np.where(fx >= 0, -t, t), fy + np.where(fy >= 0, -t, t), nz], -1)
    n /= np.sqrt((n * n).sum(-1, keepdims=True))
    rough = ((packed >> 20) & 1023) / 1023.0
    z = viewz.astype(np.float64) * viewz_scale
    if not f32_guide:
        # the 8-byte guide texel: normal as 3 x 10-bit codes (code * 2/1023 - 1, not re-normalised), roughness = the input's own 10-bit
        # code, depth rounded to 22 bits with the roughness code in the 10 low mantissa bits - and read back as one float
        codes = np.floor(np.clip(n * 511.5 + 512.0, 0.0, 1023.0))
        n = codes * (2.0 / 1023.0) - 1.0
        zb = z.astype(np.float32).view(np.uint32).astype(np.uint64)
        zb = (((zb + 0x200) & 0xFFFFFC00) | ((packed.astype(np.uint64) >> 20) & 1023)).astype(np.uint32)
        if denoising_range is not None:  # beyond the range (before or after the rounding), Inf, NaN: the canonical sky depth
            geo = (np.abs(z) <= denoising_range) & (np.abs(zb.view(np.float32).astype(np.float64)) <= denoising_range)
            zb = np.where(geo, zb, (GUIDE_SKY_DEPTH | ((packed.astype(np.uint64) >> 20) & 1023)).astype(np.uint32))
        z = zb.view(np.float32).astype(np.float64)
    return z, n, rough, (packed >> 30).astype(np.int64)


def guide_words(viewz, packed, viewz_scale=1.0, denoising_range=None):
    """the two 32-bit words of the guide texel as ClassifyTiles stores them (for a bit-exact comparison with the guide plane)"""
    z, n, rough, mat = decode_guide(viewz, packed, viewz_scale, denoising_range=denoising_range)
    w0 = z.astype(np.float32).view(np.uint32)
    c = np.round((n + 1.0) * (1023.0 / 2.0)).astype(np.uint32)
    return w0, c[..., 0] | (c[..., 1] << 10) | (c[..., 2] << 20) | (mat.astype(np.uint32) << 30)


def normal_cos(n, ns, unit_vectors=False):
    """cosine between two guide normals: 1 - |n - ns|^2 / 2 (the guide's normals are 10-bit codes that are not re-normalised: their dot
    product cannot resolve 1 - cos at the 1e-4 level the narrow specular lobes need, the squared difference can); for true unit vectors
    (the f32_guide switch) the plain dot product"""
    if unit_vectors:
        return (n * ns).sum(-1)
    d = n - ns
    return 1.0 - 0.5 * (d * d).sum(-1)


def reference_accumulate(history, signal, frames_since_reset, max_accum, restart):
    """REFERENCE: fp32 running mean; returns (new history fp32, output fp16)"""
    x = signal.astype(np.float32)
    if restart:
        h = x
    else:
        w = np.float32(1.0) / np.float32(1.0 + min(frames_since_reset, max_accum))
        h = (history.astype(np.float64) + (x.astype(np.float64) - history.astype(np.float64)) * np.float64(w)).astype(np.float32)
    return h, f16(h.astype(np.float64))


def basis(n):
    sz = np.where(n[..., 2] >= 0, 1.0, -1.0)
    a = -1.0 / (sz + n[..., 2])
    bb = n[..., 0] * n[..., 1] * a
    t = np.stack([1.0 + sz * n[..., 0] * n[..., 0] * a, sz * bb, -sz * n[..., 0]], -1)
    b = np.stack([bb, sz + n[..., 1] * n[..., 1] * a, -n[..., 1]], -1)
    return t, b


def normalize(v):
    return v / np.sqrt(np.maximum((v * v).sum(-1, keepdims=True), 1e-30))


def spec_magic_curve(r):
    return (1.0 - np.exp2(-200.0 * r * r)) * np.sqrt(np.clip(r, 0, 1))


def hitdist_norm(absz, hp, r):
    return (hp[0] + absz * hp[1]) * (1.0 + (hp[2] - 1.0) * np.exp2(hp[3] * r * r))


def spec_dominant_factor(r):
    s = np.clip(1.0 - r, 0, 1)
    return s * (np.sqrt(s) + r)


def plane_terms(viewz, packed_nr, view_to_clip, world_to_view, plane_distance_sensitivity):
    """(gax, gay, ga0, geoB) of the plane-distance weight |zs (gax px + gay py + ga0) + geoB| of every centre pixel (perspective camera)"""
    H, W = viewz.shape
    M = np.asarray(view_to_clip, np.float64)
    sgn = 1.0 if M[11] > 0 else -1.0
    fr = np.array([(-sgn - M[8]) / M[0], (sgn - M[9]) / M[5], 2.0 * sgn / M[0], -2.0 * sgn / M[5]])
    pv = np.array([fr[0] + 0.5 * fr[2] / W, fr[1] + 0.5 * fr[3] / H, fr[2] / W, fr[3] / H])
    w2v = np.asarray(world_to_view, np.float64).reshape(4, 4).T[:3, :3]
    z, n, _, _ = decode_guide(viewz, packed_nr)
    yy, xx = np.mgrid[0:H, 0:W]
    Xv = np.stack([z * (pv[2] * xx + pv[0]), z * (pv[3] * yy + pv[1]), z], -1)
    Nv = n @ w2v.T
    geoA = 1.0 / (plane_distance_sensitivity * min(W, H) / (0.5 * H * abs(M[5])) * np.abs(z))
    return Nv[..., 0] * pv[2] * geoA, Nv[..., 1] * pv[3] * geoA, (Nv[..., 0] * pv[0] + Nv[..., 1] * pv[1] + Nv[..., 2]) * geoA, -(Nv * Xv).sum(-1) * geoA


def sqrt_one_step(x):
    """sqrt(x) the way csrc/nrd_device.h sqrt1_unscaled_ evaluates it (float32): seed 0x5F1FFFF9 - (bits >> 1), one Newton step with the
    tuned constants 0.703952253 / 2.38924456 - restated here to MEASURE it (ledger row 16), not to share it"""
    x = np.asarray(x, np.float32)
    y = (np.uint32(0x5F1FFFF9) - (x.view(np.uint32) >> np.uint32(1))).view(np.float32)
    u = x * y
    r = np.float32(0.703952253) * (u * (np.float32(2.38924456) - u * y))
    return np.where(x > 0, r, 0.0).astype(np.float64)


def exp2_poly3(x):
    """2^x, x <= 0, the way csrc/nrd_device.h exp2_poly_neg evaluates it: round-to-nearest-even split, degree-3 polynomial (ledger row 17)"""
    x = np.maximum(np.asarray(x, np.float64), -126.0)
    fi = np.rint(x)
    f = x - fi
    p = ((5.519811809062958e-2 * f + 2.4267692863941193e-1) * f + 6.932618021965027e-1) * f + 9.999227523803711e-1
    return np.ldexp(p, fi.astype(np.int64))


def prepass(viewz, packed_nr, diff, spec, view_to_clip, world_to_view, frame_index, denoising_range, s, exp_hit_weight=False,
            angle_normal_weight=False, no_reach=False, f32_guide=False, one_step_sqrt=False, poly3_exp2=False, arc_normal_weight=False,
            relax_in=False, to_ycocg=False, sh1=None):
    """REBLUR_DIFFUSE_SPECULAR PrePass of one frame (perspective, no jitter, radiance mode, full frame). `s`: dict of the
    ReblurSettings fields used. Returns (Tmp1 [H, W, 2, 4] fp16: filtered diffuse / specular texel, hitTrack [H, W] fp16).
    relax_in (round 6): RELAX's PrePass - the inputs carry WORLD-space hit distances (hitDistanceParameters {1, 0, 1, 0}: no normalisation),
    so the hit-distance weight compares them relative to the centre's (scale 1 / max(hitT, 1e-3)); to_ycocg: the frozen build flavour
    converts the linear-RGB texels to YCoCg on the way in (the default flavour keeps RELAX in linear RGB).
    sh1 = (diffuse SH1 plane, specular SH1 plane) [H, W, 4] fp16 (round 6, the SH denoisers): the second texel of a signal rides along with
    EXACTLY the weights of the first; a third return value holds its filtered texels [H, W, 2, 4]"""
    def conv(t):
        if not to_ycocg:
            return t
        r, g_, b = t[..., 0], t[..., 1], t[..., 2]
        return np.stack([0.25 * r + 0.5 * g_ + 0.25 * b, 0.5 * r - 0.5 * b, 0.5 * g_ - 0.25 * r - 0.25 * b, t[..., 3]], -1)
    H, W = viewz.shape
    M = np.asarray(view_to_clip, np.float64)
    sgn = 1.0 if M[11] > 0 else -1.0
    pj = np.array([M[0], M[5], M[8], M[9], sgn])
    fr = np.array([(-sgn - M[8]) / M[0], (sgn - M[9]) / M[5], 2.0 * sgn / M[0], -2.0 * sgn / M[5]])
    pv = np.array([fr[0] + 0.5 * fr[2] / W, fr[1] + 0.5 * fr[3] / H, fr[2] / W, fr[3] / H])
    w2v = np.asarray(world_to_view, np.float64).reshape(4, 4).T[:3, :3]  # column-major -> rotation rows
    unproject = 1.0 / (0.5 * H * abs(pj[1]))
    min_dim_unproject = min(W, H) * unproject
    z, n, rough_g, mat = decode_guide(viewz, packed_nr, f32_guide=f32_guide)
    sky = ~(np.abs(z) <= denoising_range)
    yy, xx = np.mgrid[0:H, 0:W]
    Xv = np.stack([z * (pv[2] * xx + pv[0]), z * (pv[3] * yy + pv[1]), z], -1)
    Nv = n @ w2v.T
    absz = np.abs(z)
    frustum = min_dim_unproject * absz
    geoA = 1.0 / (s["planeDistanceSensitivity"] * frustum)
    gax, gay = Nv[..., 0] * pv[2] * geoA, Nv[..., 1] * pv[3] * geoA
    ga0 = (Nv[..., 0] * pv[0] + Nv[..., 1] * pv[1] + Nv[..., 2]) * geoA
    geoB = -(Nv * Xv).sum(-1) * geoA
    V = -normalize(Xv)
    inv = 1.0 / (pj[4] * z)
    nu = (pj[0] * Xv[..., 0] + pj[2] * z) * inv
    nv_ = (pj[1] * Xv[..., 1] + pj[3] * z) * inv
    kuz, kvz = pj[2] - nu * pj[4], pj[3] - nv_ * pj[4]
    ju, jv = 0.5 * W * inv, -0.5 * H * inv
    # per-frame rotation of the Poisson disk (PrePass: salt 1)
    k = hash_px(0, 0, frame_index, 1) & 63
    ang = 2.0 * np.pi * k / 64.0
    rc, rs = np.float64(np.float32(np.cos(ang))), np.float64(np.float32(np.sin(ang)))
    taps = np.stack([POISSON8[:, 0] * rc - POISSON8[:, 1] * rs, POISSON8[:, 0] * rs + POISSON8[:, 1] * rc], -1).astype(np.float32).astype(np.float64)
    reach = 10 ** 9 if no_reach else int(max(s["diffusePrepassBlurRadius"], s["specularPrepassBlurRadius"]) * 1.1) + 3
    out = np.zeros((H, W, 2, 4), np.float64)
    out1 = np.zeros((H, W, 2, 4), np.float64)
    track = np.zeros((H, W), np.float64)
    hp = s["hitDistanceParameters"]
    for sig, (plane, is_spec) in enumerate(((diff, False), (spec, True))):
        center = conv(plane.astype(np.float64))
        rough = rough_g if is_spec else np.ones_like(rough_g)
        min_mat = s["minMaterialForSpecular"] if is_spec else s["minMaterialForDiffuse"]
        hn = hitdist_norm(absz, hp, rough)
        hit = center[..., 3] * hn
        hdf = np.clip(hit / frustum, 0, 1)
        smc = spec_magic_curve(rough) if is_spec else np.ones_like(rough)
        radius = (s["specularPrepassBlurRadius"] if is_spec else s["diffusePrepassBlurRadius"]) * hdf * smc
        active = radius > 0
        world_radius = radius * unproject * absz
        T, B = basis(Nv)
        if is_spec:
            NoV = (Nv * V).sum(-1, keepdims=True)
            R = Nv * 2.0 * NoV - V
            D = normalize(Nv + (R - Nv) * spec_dominant_factor(rough)[..., None])
            NoD = (Nv * D).sum(-1, keepdims=True)
            skewed = (NoD[..., 0] < 0.999) & (rough < 0.95)
            Dr = Nv * 2.0 * NoD - D
            T2 = normalize(np.cross(Nv, Dr))
            B2 = np.cross(Dr, T2)
            T2 = T2 * ((0.5 + 0.5 * rough)[..., None] + (1.0 - (0.5 + 0.5 * rough)[..., None]) * NoD)
            T, B = np.where(skewed[..., None], T2, T), np.where(skewed[..., None], B2, B)
        T, B = T * world_radius[..., None], B * world_radius[..., None]
        jtx, jty = ju * (pj[0] * T[..., 0] + kuz * T[..., 2]), jv * (pj[1] * T[..., 1] + kvz * T[..., 2])
        jbx, jby = ju * (pj[0] * B[..., 0] + kuz * B[..., 2]), jv * (pj[1] * B[..., 1] + kvz * B[..., 2])
        angle = np.arctan(3.0 * np.clip(rough, 0, 1) ** 2)  # lerp(lobeAngleFraction, 1, nonLinearAccumSpeed = 1) = 1 in the PrePass
        normal_w = 1.0 / np.maximum(angle, NORMAL_ANGLE_MIN)
        hitA = 1.0 / (1e-6 + (1.0 - 1e-6) * np.minimum(1.0, smc))
        if relax_in:
            hitA = hitA / np.maximum(center[..., 3], 1e-3)
        hitB = -center[..., 3] * hitA
        roughA = 1.0 / (0.01 + 0.99 * np.clip(rough * s["roughnessFraction"], 0, 1))
        roughB = -rough * roughA
        acc, wsum, min_hit = center.copy(), np.ones((H, W)), hit.copy()
        acc1 = sh1[sig].astype(np.float64) if sh1 is not None else None
        for t in range(8):
            fpx = np.floor(taps[t, 0] * jtx + taps[t, 1] * jbx + xx + 0.5)
            fpy = np.floor(taps[t, 0] * jty + taps[t, 1] * jby + yy + 0.5)
            in_win = (fpx >= np.maximum(xx - reach, 0)) & (fpx <= np.minimum(xx + reach, W - 1)) & (fpy >= np.maximum(yy - reach, 0)) & (fpy <= np.minimum(yy + reach, H - 1))
            px, py = np.clip(fpx, 0, W - 1).astype(np.int64), np.clip(fpy, 0, H - 1).astype(np.int64)
            zs, ns, rs_, ms = z[py, px], n[py, px], rough_g[py, px], mat[py, px]
            sv = conv(plane[py, px].astype(np.float64))
            valid = in_win & active & ~sky[py, px] & ~((mat != ms) & (np.maximum(mat, ms) >= min_mat))
            w = POISSON8[t, 2] * smoothstep01(1.0 - np.abs(zs * (gax * fpx + gay * fpy + ga0) + geoB))
            cosa = normal_cos(n, ns, f32_guide)
            if arc_normal_weight:
                w = w * smoothstep01(1.0 - np.arccos(np.clip(cosa, -1, 1)) * normal_w)  # the true angle (measurement only)
            elif angle_normal_weight and one_step_sqrt:
                w = w * smoothstep01(1.0 - sqrt_one_step(2.0 * np.clip(1.0 - cosa, 0, 1)) * normal_w)
            elif angle_normal_weight:
                w = w * smoothstep01(1.0 - np.sqrt(2.0 * np.clip(1.0 - cosa, 0, 1)) * normal_w)  # Math::AcosApprox: the chord
            else:
                w = w * smoothstep01(1.0 - 2.0 * np.clip(1.0 - cosa, 0, 1) * normal_w * normal_w)
            if is_spec:
                w = w * smoothstep01(1.0 - np.abs(rs_ * roughA + roughB))
            ax = np.abs(sv[..., 3] * hitA + hitB)
            e = (exp2_poly3(-4.32808512 * ax) if poly3_exp2 else np.exp(-3.0 * ax)) if exp_hit_weight else np.clip(1.0 - ax, 0, 1) ** 2
            w = w * (s["minHitDistanceWeight"] + (1.0 - s["minHitDistanceWeight"]) * e)
            w = np.where(valid, w, 0.0)
            acc = acc + np.where(valid[..., None], sv, 0.0) * w[..., None]
            if sh1 is not None:
                acc1 = acc1 + np.where(valid[..., None], sh1[sig][py, px].astype(np.float64), 0.0) * w[..., None]
            wsum = wsum + w
            min_hit = np.where(valid & (w > 0), np.minimum(min_hit, sv[..., 3] * hn), min_hit)
        res = acc / wsum[..., None]
        out[:, :, sig] = np.where(sky[..., None], 0.0, res)
        if sh1 is not None:
            out1[:, :, sig] = np.where(sky[..., None], 0.0, acc1 / wsum[..., None])
        if is_spec:
            track = np.where(sky, 0.0, min_hit)
    return (f16(out), f16(track), f16(out1)) if sh1 is not None else (f16(out), f16(track))


def hash_px_arr(x, y, frame, salt):
    """hash_px over integer arrays (uint32 wrap-around arithmetic)"""
    m = np.uint64(0xFFFFFFFF)
    x, y = x.astype(np.uint64), y.astype(np.uint64)
    h = ((x * np.uint64(73856093)) & m) ^ ((y * np.uint64(19349663)) & m) ^ np.uint64((frame * 83492791) & 0xFFFFFFFF) ^ np.uint64((salt * 2654435761) & 0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0x5BD1E995)) & m
    h ^= h >> np.uint64(15)
    return h


MIN_CONVERGED_RADIUS_SCALE, POST_BLUR_RADIUS_SCALE = 0.25, 2.0


def normal_weight(cosa, normal_w, upstream):
    """frozen flavour: on the squared angle; default (upstream) flavour: on the angle as Math::AcosApprox takes it, sqrt(2 (1 - cos))"""
    if upstream:
        return smoothstep01(1.0 - np.sqrt(2.0 * np.clip(1.0 - cosa, 0, 1)) * normal_w)
    return smoothstep01(1.0 - 2.0 * np.clip(1.0 - cosa, 0, 1) * normal_w * normal_w)


def blur_pass(post, viewz, packed_nr, sig_in, speeds, view_to_clip, world_to_view, frame_index, denoising_range, s, upstream=False):
    """REBLUR_DIFFUSE_SPECULAR Blur (post = False) / PostBlur (post = True) of one frame on tap texels: `sig_in` [H, W, 2, 4] fp16 = the
    signal halves of the tap texels the pass gathers (HistoryFix's for Blur, Blur's for PostBlur), `speeds` [H, W] uint16 = Data1 (diffuse
    | specular accumulation speed in quarter frames). Differences from the PrePass: the radius comes from the accumulation speed
    (converged pixels blur less, PostBlur twice as far), the normal-weight lobe narrows with it, Blur rotates its disk per 2x2 pixel
    quad (PostBlur per frame), a rejected tap enters with weight 0, no hit-distance tracking. Returns [H, W, 2, 4] fp16.
    upstream = True: the DEFAULT build flavour - Blur rotates per PIXEL, hit-distance weight exp(-3 |x|), normal weight on the chord."""
    H, W = viewz.shape
    M = np.asarray(view_to_clip, np.float64)
    sgn = 1.0 if M[11] > 0 else -1.0
    pj = np.array([M[0], M[5], M[8], M[9], sgn])
    fr = np.array([(-sgn - M[8]) / M[0], (sgn - M[9]) / M[5], 2.0 * sgn / M[0], -2.0 * sgn / M[5]])
    pv = np.array([fr[0] + 0.5 * fr[2] / W, fr[1] + 0.5 * fr[3] / H, fr[2] / W, fr[3] / H])
    w2v = np.asarray(world_to_view, np.float64).reshape(4, 4).T[:3, :3]
    unproject = 1.0 / (0.5 * H * abs(pj[1]))
    min_dim_unproject = min(W, H) * unproject
    z, n, rough_g, mat = decode_guide(viewz, packed_nr)
    sky = ~(np.abs(z) <= denoising_range)
    yy, xx = np.mgrid[0:H, 0:W]
    Xv = np.stack([z * (pv[2] * xx + pv[0]), z * (pv[3] * yy + pv[1]), z], -1)
    Nv = n @ w2v.T
    absz = np.abs(z)
    frustum = min_dim_unproject * absz
    geoA = 1.0 / (s["planeDistanceSensitivity"] * frustum)
    gax, gay = Nv[..., 0] * pv[2] * geoA, Nv[..., 1] * pv[3] * geoA
    ga0 = (Nv[..., 0] * pv[0] + Nv[..., 1] * pv[1] + Nv[..., 2]) * geoA
    geoB = -(Nv * Xv).sum(-1) * geoA
    V = -normalize(Xv)
    inv = 1.0 / (pj[4] * z)
    nu = (pj[0] * Xv[..., 0] + pj[2] * z) * inv
    nv_ = (pj[1] * Xv[..., 1] + pj[3] * z) * inv
    kuz, kvz = pj[2] - nu * pj[4], pj[3] - nv_ * pj[4]
    ju, jv = 0.5 * W * inv, -0.5 * H * inv
    if post:  # one rotation per frame (salt 3)
        k = np.full((H, W), hash_px(0, 0, frame_index, 3) & 63, np.int64)
    else:     # one rotation per 2x2 pixel quad (salt 2)
        sh = 0 if upstream else 1
        k = (hash_px_arr(xx >> sh, yy >> sh, frame_index, 2) & np.uint64(63)).astype(np.int64)
    ang = 2.0 * np.pi * np.arange(64) / 64.0
    rot_c, rot_s = np.cos(ang).astype(np.float32).astype(np.float64)[k], np.sin(ang).astype(np.float32).astype(np.float64)[k]
    reach = int((s["maxBlurRadius"] + s["minBlurRadius"]) * (2.2 if post else 1.1)) + 3
    out = np.zeros((H, W, 2, 4), np.float64)
    hp = s["hitDistanceParameters"]
    A_all = [(speeds & 255).astype(np.float64) * 0.25, (speeds >> 8).astype(np.float64) * 0.25]
    for sig, is_spec in enumerate((False, True)):
        plane = sig_in[:, :, sig]
        center = plane.astype(np.float64)
        rough = rough_g if is_spec else np.ones_like(rough_g)
        min_mat = s["minMaterialForSpecular"] if is_spec else s["minMaterialForDiffuse"]
        hn = hitdist_norm(absz, hp, rough)
        hdf = np.clip(center[..., 3] * hn / frustum, 0, 1)
        non_lin = 1.0 / (1.0 + A_all[sig])
        smc = spec_magic_curve(rough) if is_spec else np.ones_like(rough)
        r = s["maxBlurRadius"] * (MIN_CONVERGED_RADIUS_SCALE + (1.0 - MIN_CONVERGED_RADIUS_SCALE) * non_lin) * (hdf + (1.0 - hdf) * non_lin) + s["minBlurRadius"]
        r = r * (POST_BLUR_RADIUS_SCALE if post else 1.0) * smc
        radius = r if s["maxBlurRadius"] != 0.0 else np.zeros_like(r)
        active = radius > 0
        world_radius = radius * unproject * absz
        T, B = basis(Nv)
        if is_spec:
            NoV = (Nv * V).sum(-1, keepdims=True)
            R = Nv * 2.0 * NoV - V
            D = normalize(Nv + (R - Nv) * spec_dominant_factor(rough)[..., None])
            NoD = (Nv * D).sum(-1, keepdims=True)
            skewed = (NoD[..., 0] < 0.999) & (rough < 0.95)
            Dr = Nv * 2.0 * NoD - D
            T2 = normalize(np.cross(Nv, Dr))
            B2 = np.cross(Dr, T2)
            T2 = T2 * ((0.5 + 0.5 * rough)[..., None] + (1.0 - (0.5 + 0.5 * rough)[..., None]) * NoD)
            T, B = np.where(skewed[..., None], T2, T), np.where(skewed[..., None], B2, B)
        T, B = T * world_radius[..., None], B * world_radius[..., None]
        jtx, jty = ju * (pj[0] * T[..., 0] + kuz * T[..., 2]), jv * (pj[1] * T[..., 1] + kvz * T[..., 2])
        jbx, jby = ju * (pj[0] * B[..., 0] + kuz * B[..., 2]), jv * (pj[1] * B[..., 1] + kvz * B[..., 2])
        angle = np.arctan(3.0 * np.clip(rough, 0, 1) ** 2) * (s["lobeAngleFraction"] + (1.0 - s["lobeAngleFraction"]) * non_lin)
        normal_w = 1.0 / np.maximum(angle, NORMAL_ANGLE_MIN)
        hitA = 1.0 / (1e-6 + (1.0 - 1e-6) * np.minimum(non_lin, smc))
        hitB = -center[..., 3] * hitA
        roughA = 1.0 / (0.01 + 0.99 * np.clip(rough * s["roughnessFraction"], 0, 1))
        roughB = -rough * roughA
        acc, wsum = center.copy(), np.ones((H, W))
        for t in range(8):
            ox = POISSON8[t, 0] * rot_c - POISSON8[t, 1] * rot_s
            oy = POISSON8[t, 0] * rot_s + POISSON8[t, 1] * rot_c
            fpx = np.floor(ox * jtx + oy * jbx + xx + 0.5)
            fpy = np.floor(ox * jty + oy * jby + yy + 0.5)
            in_win = (fpx >= np.maximum(xx - reach, 0)) & (fpx <= np.minimum(xx + reach, W - 1)) & (fpy >= np.maximum(yy - reach, 0)) & (fpy <= np.minimum(yy + reach, H - 1))
            px, py = np.clip(fpx, 0, W - 1).astype(np.int64), np.clip(fpy, 0, H - 1).astype(np.int64)
            zs, ns, rs_, ms = z[py, px], n[py, px], rough_g[py, px], mat[py, px]
            sv = plane[py, px].astype(np.float64)
            valid = in_win & active & ~sky[py, px] & ~((mat != ms) & (np.maximum(mat, ms) >= min_mat))
            w = POISSON8[t, 2] * smoothstep01(1.0 - np.abs(zs * (gax * fpx + gay * fpy + ga0) + geoB))
            w = w * normal_weight(normal_cos(n, ns), normal_w, upstream)
            if is_spec:
                w = w * smoothstep01(1.0 - np.abs(rs_ * roughA + roughB))
            ax = np.abs(sv[..., 3] * hitA + hitB)
            w = w * (s["minHitDistanceWeight"] + (1.0 - s["minHitDistanceWeight"]) * (np.exp(-3.0 * ax) if upstream else np.clip(1.0 - ax, 0, 1) ** 2))
            w = np.where(valid, w, 0.0)
            acc = acc + np.where(valid[..., None], sv, 0.0) * w[..., None]
            wsum = wsum + w
        out[:, :, sig] = np.where(sky[..., None], 0.0, acc / wsum[..., None])
    return f16(out)
