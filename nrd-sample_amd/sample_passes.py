"""Host-side twins of the sample's passes either side of the denoiser over the C-ABI:

  confidence_blur  - "History confidence - Blur" loop, Source/NRDSample.cpp:3999-4026 (5 x Shaders/ConfidenceBlur.cs.hlsl)
  backend_unpack   - the NRD-facing part of Shaders/Composition.cs.hlsl:57-64, 74-175

Planes are [H, row bytes] uint8 arrays: numpy for the CPU oracle backend (tests only), torch CUDA tensors for the HIP backend.
"""
import ctypes as C

import numpy as np

from . import api

SHARC_DOWNSCALE = 5  # Shaders/Shared.hlsli (render resolution / 5, Source/NRDSample.cpp:596-598)


def sharc_dims(render_w, render_h):
    """Sample::GetSharcDims(): 16 * ceil(ceil(render / 5) / 16)"""
    f = lambda v: 16 * ((v // SHARC_DOWNSCALE + 15) // 16)
    return f(render_w), f(render_h)


def _stream(backend):
    if backend.is_device:
        import torch
        return torch.cuda.current_stream().cuda_stream
    return None


def _pp(buf):
    if buf is None:
        return None, 0
    ptr, pitch, _, _ = api._ptr_pitch(buf)
    return ptr, pitch


def confidence_blur(backend, ping, pong, width, height, camera_frustum, rect_width, unproject, frame_index, max_accumulated_frame_num,
                    relax=False, ortho_mode=0.0, first_pass=0, passes_num=5):
    d = api.ConfidenceBlurDesc()
    d.ping, pitch = _pp(ping)
    d.pong, pitch2 = _pp(pong)
    assert pitch == pitch2
    d.pitch_bytes, d.width, d.height = pitch, width, height
    d.camera_frustum[:] = [float(v) for v in camera_frustum]
    d.inv_size[:] = [float(np.float32(1.0) / np.float32(width)), float(np.float32(1.0) / np.float32(height))]
    d.rect_width, d.unproject, d.ortho_mode = float(rect_width), float(unproject), float(ortho_mode)
    d.frame_index, d.max_accumulated_frame_num, d.relax = int(frame_index), int(max_accumulated_frame_num), 1 if relax else 0
    d.first_pass, d.passes_num = first_pass, passes_num
    r = backend.confidence_blur(C.byref(d), _stream(backend))
    if r != 0:
        raise api.NrdError("confidence_blur", int(r))


def backend_unpack(backend, width, height, mode=api.UNPACK_NORMAL, relax=False, resolve=False, diff=None, spec=None, diff_sh1=None,
                   spec_sh1=None, normal_roughness=None, shadow=None, shadow_bytes_per_texel=4, out_diff=None, out_spec=None,
                   out_shadow=None, view_to_world=None, camera_frustum=(0, 0, 0, 0)):
    d = api.UnpackDesc()
    d.width, d.height, d.mode, d.relax, d.resolve = width, height, int(mode), 1 if relax else 0, 1 if resolve else 0
    d.diff, d.diff_pitch = _pp(diff)
    d.spec, d.spec_pitch = _pp(spec)
    d.diff_sh1, d.diff_sh1_pitch = _pp(diff_sh1)
    d.spec_sh1, d.spec_sh1_pitch = _pp(spec_sh1)
    d.normal_roughness, d.normal_roughness_pitch = _pp(normal_roughness)
    d.shadow, d.shadow_pitch = _pp(shadow)
    d.shadow_bytes_per_texel = shadow_bytes_per_texel
    d.out_diff, d.out_diff_pitch = _pp(out_diff)
    d.out_spec, d.out_spec_pitch = _pp(out_spec)
    d.out_shadow, d.out_shadow_pitch = _pp(out_shadow)
    v2w = np.eye(3, dtype=np.float32) if view_to_world is None else np.asarray(view_to_world, dtype=np.float32).reshape(3, 3)
    d.view_to_world[:] = [float(v) for v in v2w.reshape(-1)]
    d.camera_frustum[:] = [float(v) for v in camera_frustum]
    d.inv_rect_size[:] = [float(np.float32(1.0) / np.float32(width)), float(np.float32(1.0) / np.float32(height))]
    r = backend.backend_unpack(C.byref(d), _stream(backend))
    if r != 0:
        raise api.NrdError("backend_unpack", int(r))


def frontend_pack(backend, width, height, mode=api.UNPACK_NORMAL, relax=False, sanitize=True, hit_distance_parameters=(3.0, 0.1, 20.0, -25.0),
                  tan_of_light_angular_radius=0.0, **planes):
    """nrdhip_frontend_pack: raw fp32 planes -> NRD input planes. `planes`: normal, material_id, viewz, diff, spec, diff_direction,
    spec_direction, shadow (inputs) and out_normal_roughness, out_diff, out_spec, out_diff_sh1, out_spec_sh1, out_penumbra,
    out_translucency (outputs); byte views [rows, pitch] (numpy or torch), absent = skipped"""
    d = api.FrontendPackDesc()
    d.width, d.height, d.mode, d.relax, d.sanitize = width, height, int(mode), 1 if relax else 0, 1 if sanitize else 0
    d.hit_distance_parameters[:] = [float(v) for v in hit_distance_parameters]
    d.tan_of_light_angular_radius = float(tan_of_light_angular_radius)
    for name, buf in planes.items():
        ptr, pitch = _pp(buf)
        setattr(d, name, ptr)
        setattr(d, name + "_pitch", pitch)
    r = backend.frontend_pack(C.byref(d), _stream(backend))
    if r != 0:
        raise api.NrdError("frontend_pack", int(r))


def compose(backend, width, height, sh=False, relax=False, hair_material_id=0xFFFFFFFF, view_to_world=None, camera_frustum=(0, 0, 0, 0), **planes):
    """nrdhip_compose: re-jitter (SH mode) + material re-modulation of the planes nrdhip_backend_unpack wrote. `planes`: diff, spec,
    diff_sh0, diff_sh1, spec_sh0, spec_sh1, normal_roughness, viewz, base_color_metalness, out_diff, out_spec"""
    d = api.ComposeDesc()
    d.width, d.height, d.sh, d.relax, d.hair_material_id = width, height, 1 if sh else 0, 1 if relax else 0, int(hair_material_id)
    for name, buf in planes.items():
        ptr, pitch = _pp(buf)
        setattr(d, name, ptr)
        setattr(d, name + "_pitch", pitch)
    v2w = np.eye(3, dtype=np.float32) if view_to_world is None else np.asarray(view_to_world, dtype=np.float32).reshape(3, 3)
    d.view_to_world[:] = [float(v) for v in v2w.reshape(-1)]
    d.camera_frustum[:] = [float(v) for v in camera_frustum]
    d.inv_rect_size[:] = [float(np.float32(1.0) / np.float32(width)), float(np.float32(1.0) / np.float32(height))]
    r = backend.compose(C.byref(d), _stream(backend))
    if r != 0:
        raise api.NrdError("compose", int(r))


def taa(backend, mv, composed, history, result, rect_width, rect_height, render_width=None, render_height=None, rect_width_prev=0,
        rect_height_prev=0, tonemap=True, hdr_scale=1.0, taa_min_mix=0.1):
    """Shaders/Taa.cs.hlsl: one dispatch; `history` is last frame's `result` (ping-pong by the caller, Source/NRDSample.cpp TAA pass)"""
    d = api.TaaDesc()
    d.mv, d.mv_pitch = _pp(mv)
    d.composed, d.composed_pitch = _pp(composed)
    d.history, d.history_pitch = _pp(history)
    d.result, d.result_pitch = _pp(result)
    d.rect_width, d.rect_height = rect_width, rect_height
    d.rect_width_prev, d.rect_height_prev = rect_width_prev, rect_height_prev
    d.render_width, d.render_height = render_width or rect_width, render_height or rect_height
    d.tonemap, d.hdr_scale, d.taa = 1 if tonemap else 0, float(hdr_scale), float(taa_min_mix)
    r = backend.taa(C.byref(d), _stream(backend))
    if r != 0:
        raise api.NrdError("taa", int(r))


def synth_gradient(width, height, seed=0, sky_fraction=0.05):
    """Synthetic Gradient_Ping plane [H, W, 4] float16: {gradient, octahedral view normal xy, viewZ * 0.125}
    (layout of Shaders/SharcUpdate.cs.hlsl:249) - a slanted floor, a far wall and some sky texels."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:height, 0:width].astype(np.float32)
    floor = y > height * 0.5
    z = np.where(floor, 4.0 + 40.0 * (height - y) / height, 30.0).astype(np.float32)
    n = np.where(floor[..., None], np.array([0.0, 0.8, -0.6], np.float32), np.array([0.0, 0.0, -1.0], np.float32))
    n = n + 0.05 * rng.standard_normal(n.shape).astype(np.float32)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    # octahedral encode to [0, 1]
    s = np.abs(n).sum(-1, keepdims=True)
    o = n[..., :2] / s
    neg = n[..., 2:3] < 0
    o = np.where(neg, (1.0 - np.abs(o[..., ::-1])) * np.where(o >= 0, 1.0, -1.0), o)
    o = o * 0.5 + 0.5
    g = (rng.random((height, width)).astype(np.float32) ** 4) * 4.0
    sky = rng.random((height, width)) < sky_fraction
    z = np.where(sky, 2.0e5, z)
    out = np.zeros((height, width, 4), np.float16)
    out[..., 0] = g
    out[..., 1:3] = o
    out[..., 3] = z * 0.125
    return out
