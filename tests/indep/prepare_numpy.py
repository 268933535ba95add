"""TEST INFRASTRUCTURE - an INDEPENDENT restatement (numpy, float64; shares no code with oracle/ or the kernels) of the PrepareInputs pass
of REBLUR_DIFFUSE_SPECULAR (the sample's default operating point, Source/NRDSample.cpp:267, :545-548): the checkerboard resolve of
half-width noisy inputs and the AREA_3X3 / AREA_5X5 hit-distance reconstruction. tests/test_independent.py holds the oracle to it.

Contract restated (Shaders/TraceOpaque.cs.hlsl:482-508 for the layout): pixel (x, y) of a checkerboarded signal lives at texel (x >> 1, y) of
its half-width input when ((x ^ y ^ frameIndex) & 1) equals the signal's phase (WHITE: diffuse 1, specular 0); a pixel of the other colour takes
the mean of its left / right neighbours weighted by smoothstep(1 - |dz| / (0.03 |z|)), the plain mean of whichever neighbour has geometry when
both weights vanish; a texel whose hit distance is 0 takes the bilateral mean (plane distance x normal [x roughness]) of the positive hit
distances among the texels of the window that carry the signal."""
import numpy as np

from reblur_numpy import decode_guide, normal_cos, normal_weight, smoothstep01, f16

NORMAL_ANGLE_MIN = 0.02


def prepare_inputs(viewz, packed_nr, half_in, phase, frame_index, denoising_range, is_spec, radius, geo, s, upstream=True):
    """half_in [H, ceil(W / 2), 4] fp16 -> the dense [H, W, 4] fp16 signal. `geo` = the per-pixel plane terms (gax, gay, ga0, geoB) of the
    spatial passes (reblur_numpy.prepass computes the same ones), only used when radius > 0."""
    H, W = viewz.shape
    z, n, rough_g, mat = decode_guide(viewz, packed_nr)
    sky = ~(np.abs(z) <= denoising_range)
    yy, xx = np.mgrid[0:H, 0:W]
    src = half_in.astype(np.float64)
    has = (((xx ^ yy) ^ frame_index) & 1) == phase

    def fetch(px, py):
        return src[py, px >> 1]

    own = fetch(xx, yy)
    inv_dz = 1.0 / (0.03 * np.maximum(np.abs(z), 1e-6))
    acc, wsum, ok_any, plain, plain_n = np.zeros((H, W, 4)), np.zeros((H, W)), np.zeros((H, W), bool), np.zeros((H, W, 4)), np.zeros((H, W))
    for d in (-1, 1):
        px = xx + d
        inside = (px >= 0) & (px < W)
        cpx = np.clip(px, 0, W - 1)
        ok = inside & ~sky[yy, cpx]
        w = np.where(ok, smoothstep01(1.0 - np.abs(z[yy, cpx] - z) * inv_dz), 0.0)
        v = fetch(cpx, yy)
        acc += v * w[..., None]
        wsum += w
        plain += np.where(ok[..., None], v, 0.0)
        plain_n += ok
        ok_any |= ok
    edge = ~(wsum > 0)
    resolved = np.where(edge[..., None], plain / np.maximum(plain_n, 1)[..., None], acc / np.where(wsum > 0, wsum, 1.0)[..., None])
    resolved = np.where((edge & (plain_n == 0))[..., None], 0.0, resolved)
    v = np.where(has[..., None], own, resolved)
    if radius > 0:
        gax, gay, ga0, geoB = geo
        rough = rough_g if is_spec else np.ones_like(rough_g)
        min_mat = s["minMaterialForSpecular"] if is_spec else s["minMaterialForDiffuse"]
        angle = np.arctan(3.0 * np.clip(rough, 0, 1) ** 2) * s["lobeAngleFraction"]
        normal_w = 1.0 / np.maximum(angle, NORMAL_ANGLE_MIN)
        roughA = 1.0 / (0.01 + 0.99 * np.clip(rough * s["roughnessFraction"], 0, 1))
        hsum, hw = np.zeros((H, W)), np.zeros((H, W))
        for j in range(-radius, radius + 1):
            for i in range(-radius, radius + 1):
                if i == 0 and j == 0:
                    continue
                px, py = xx + i, yy + j
                inside = (px >= 0) & (px < W) & (py >= 0) & (py < H)
                cx, cy = np.clip(px, 0, W - 1), np.clip(py, 0, H - 1)
                ms = mat[cy, cx]
                ok = inside & has[cy, cx] & ~sky[cy, cx] & ~((mat != ms) & (np.maximum(mat, ms) >= min_mat))
                h = fetch(cx, cy)[..., 3]
                ok &= h > 0
                w = smoothstep01(1.0 - np.abs(z[cy, cx] * (gax * px + gay * py + ga0) + geoB))
                w = w * normal_weight(normal_cos(n, n[cy, cx]), normal_w, upstream)
                if is_spec:
                    w = w * smoothstep01(1.0 - np.abs(rough_g[cy, cx] * roughA - rough * roughA))
                w = np.where(ok, w, 0.0)
                hsum, hw = hsum + h * w, hw + w
        fill = (v[..., 3] == 0) & (hw > 0)
        v[..., 3] = np.where(fill, hsum / np.where(hw > 0, hw, 1.0), v[..., 3])
    v = np.where(sky[..., None], 0.0, v)
    return f16(v)
