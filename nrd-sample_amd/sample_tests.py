"""The sample's recorded test presets (`Tests/<Scene>.bin` of the reference tree) as operating points for this backend.

A preset file is an array of 648-byte records `{Settings m_Settings (168 B); Camera state (480 B)}` - written by the "Add" button
(Source/NRDSample.cpp:1911-1923), loaded by index with fseek + fread (:1868-1901, record size :1787-1788), after which the sample
forces debug = 0, denoiser = REBLUR, TAA / jitter on and a history reset (:1886-1896). `struct Settings` is Source/NRDSample.cpp:233-297
(one double, 20 floats, 12 int32, 27 bools, padded to 8). The camera state belongs to the absent NRIFramework submodule; the two fields
used here were located in the data itself and are checked by tests/test_sample_presets.py: the global position (three doubles at byte
0, z-up) and the camera rotation (4x4 float matrix at byte 160, followed by its transpose at byte 224; its third column is the world's
up axis in view space, x component 0: the camera never rolls).

What a preset drives here (the G-buffer itself stays the procedural scene of synth.py - the reference holds no recorded inputs,
SURVEY.md 8c): the NRD operating point exactly as Sample::PrepareFrame derives it from m_Settings (:2157-2189: accumulation lengths
with the reset factor, :3675 hit-distance scale, :587-594 + :4072-4076 sun direction -> SIGMA light direction), the field of view, the
sun, and the camera's view direction.

The preset files committed under tests/golden/sample_tests/ are data fixtures the reference's own test procedure holds."""
import math
import struct
from collections import namedtuple

import numpy as np

RECORD_SIZE = 648
SETTINGS_SIZE = 168
SETTINGS_FMT = "<d20f12i27B5x"
SETTINGS_FIELDS = (
    "motionStartTime maxFps camFov sunAzimuth sunElevation sunAngularDiameter exposure roughnessOverride metalnessOverride "
    "emissionIntensityLights emissionIntensityCubes debug meterToUnitsMultiplier emulateMotionSpeed animatedObjectScale separator "
    "animationProgress animationSpeed hitDistScale resolutionScale sharpness "
    "maxAccumulatedFrameNum maxFastAccumulatedFrameNum onScreen forcedMaterial animatedObjectNum activeAnimation motionMode denoiser rpp "
    "bounceNum tracingMode mvType "
    "cameraJitter limitFps SHARC PSR indirectDiffuse indirectSpecular normalMap TAA animatedObjects animateScene animateSun nineBrothers "
    "blink pauseAnimation emission linearMotion emissiveObjects importanceSampling specularLobeTrimming ortho adaptiveAccumulation "
    "usePrevFrame windowAlignment boost SR RR confidence").split()
assert struct.calcsize(SETTINGS_FMT) == SETTINGS_SIZE and len(SETTINGS_FIELDS) == 60
CAMERA_POSITION_OFFSET = 0      # double3, z-up world
CAMERA_ROTATION_OFFSET = 160    # float4x4: world -> view rotation stored column-major (= view -> world row-major)
MAX_HISTORY_FRAME_NUM = 60      # Source/NRDSample.cpp:40 (min(60, REBLUR / RELAX maxima))
ACCUMULATION_TIME = 0.5        # seconds, Source/NRDSample.cpp:27
RESOLUTION_HALF = 2             # tracingMode of the sample's default operating point (Source/NRDSample.cpp:267)

Preset = namedtuple("Preset", "index settings position rotation")


def load_presets(path):
    """-> [Preset]: settings = dict of `struct Settings`, position = np.float64[3] (the sample's z-up world), rotation = 3x3 world ->
    view (rows: the camera's right, up, forward axes in world coordinates)."""
    data = open(path, "rb").read()
    if len(data) % RECORD_SIZE:
        raise ValueError("%s: %d bytes is not a whole number of %d-byte records" % (path, len(data), RECORD_SIZE))
    out = []
    for i in range(len(data) // RECORD_SIZE):
        rec = data[i * RECORD_SIZE:(i + 1) * RECORD_SIZE]
        vals = struct.unpack_from(SETTINGS_FMT, rec, 0)
        st = dict(zip(SETTINGS_FIELDS, vals))
        for k in SETTINGS_FIELDS[33:]:
            st[k] = bool(st[k])
        cam = rec[SETTINGS_SIZE:]
        pos = np.frombuffer(cam, dtype=np.float64, count=3, offset=CAMERA_POSITION_OFFSET).copy()
        m = np.frombuffer(cam, dtype=np.float32, count=16, offset=CAMERA_ROTATION_OFFSET).reshape(4, 4).astype(np.float64)
        out.append(Preset(i, st, pos, m.T[:3, :3].copy()))  # memory is column-major: element [c][r]
    return out


def forced_after_load(settings):
    """what the sample overrides right after reading a record (Source/NRDSample.cpp:1886-1894)"""
    s = dict(settings)
    s.update(debug=0.0, denoiser=0, RR=False, TAA=True, cameraJitter=True)
    return s


def forward_y_up(preset):
    """view direction in synth.py's y-up world: the sample's world is z-up (x, y, z) -> (x, z, y)"""
    f = preset.rotation[2]
    return np.array([f[0], f[2], f[1]])


def sun_direction(settings):
    """Sample::GetSunDirection (Source/NRDSample.cpp:587-594), z-up"""
    az, el = math.radians(settings["sunAzimuth"]), math.radians(settings["sunElevation"])
    return np.array([math.cos(az) * math.cos(el), math.sin(az) * math.cos(el), math.sin(el)])


def accumulation(settings, reset_history_factor, fps=60.0, accumulation_time=ACCUMULATION_TIME, get_max_accumulated_frame_num=None):
    """(maxAccumulatedFrameNum, maxFastAccumulatedFrameNum, maxStabilizedFrameNum) the way Sample::PrepareFrame sets them
    (Source/NRDSample.cpp:2161-2184). With adaptive accumulation the lengths follow the frame rate: GetMaxAccumulatedFrameNum(time,
    fps) = time x fps rounded (include/NRDSettings.h), capped at MAX_HISTORY_FRAME_NUM, fast = accumulated / 5; otherwise the recorded
    sliders. The reset factor (1 / (1 + 0.2 d), 0 on a forced reset, :2155-2158) scales both, rounded half up."""
    acc, fast = int(settings["maxAccumulatedFrameNum"]), int(settings["maxFastAccumulatedFrameNum"])
    stab = acc
    if settings["adaptiveAccumulation"] and get_max_accumulated_frame_num is not None:
        t = accumulation_time * (0.667 if (settings["boost"] and settings["SHARC"]) else 1.0)
        acc = min(max(int(get_max_accumulated_frame_num(t, min(fps, 121.0))), 1), MAX_HISTORY_FRAME_NUM)
        fast = acc // 5
        stab = acc
    return int(acc * reset_history_factor + 0.5), int(fast * reset_history_factor + 0.5), stab


def scene_kwargs(preset):
    """keyword arguments for synth.Scene that follow the preset: field of view, sun, hit-distance scale, view direction"""
    s = preset.settings
    return dict(hfov=float(s["camFov"]), forward=forward_y_up(preset), ortho=bool(s["ortho"]),
                sun_deg=(float(s["sunAzimuth"]), float(s["sunElevation"]), float(s["sunAngularDiameter"])),
                hit_dist_scale=float(s["hitDistScale"]) * float(s["meterToUnitsMultiplier"]))


def denoiser_settings(api, preset, scene, denoisers, first_frame, fps=60.0):
    """{denoiser: settings struct} for one frame of a preset: the sample's fixed REBLUR / RELAX / SIGMA defaults (:563-585) with the
    per-frame fields of PrepareFrame; `first_frame` = the frame right after the load (m_ForceHistoryReset: reset factor 0)."""
    s = forced_after_load(preset.settings)
    factor = 0.0 if first_frame else 1.0
    acc, fast, stab = accumulation(s, factor, fps=fps, accumulation_time=ACCUMULATION_TIME,
                                   get_max_accumulated_frame_num=api.get_max_accumulated_frame_num)
    out = {}
    for d in denoisers:
        if d.name.startswith("REBLUR"):
            st = api.ReblurSettings(minMaterialForDiffuse=0, minMaterialForSpecular=1, fastHistoryClampingSigmaScale=1.5,
                                    maxAccumulatedFrameNum=acc, maxFastAccumulatedFrameNum=fast, maxStabilizedFrameNum=stab)
            st.hitDistanceParameters.A = float(s["hitDistScale"]) * float(s["meterToUnitsMultiplier"])  # :3675
        elif d.name.startswith("RELAX"):
            st = api.RelaxSettings(minMaterialForDiffuse=0, minMaterialForSpecular=1, fastHistoryClampingSigmaScale=1.5,
                                   diffuseMaxAccumulatedFrameNum=acc, diffuseMaxFastAccumulatedFrameNum=fast,
                                   specularMaxAccumulatedFrameNum=acc, specularMaxFastAccumulatedFrameNum=fast)
        elif d.name.startswith("SIGMA"):
            st = api.SigmaSettings(lightDirection=list(scene.sun))  # :4072-4076 (the scene keeps its sun above the horizon, y-up)
        else:
            st = api.ReferenceSettings()
        out[d] = st
    return out


def distinct_operating_points(presets):
    """indices of presets whose NRD-relevant fields differ from every earlier one"""
    seen, out = set(), []
    for p in presets:
        s = p.settings
        key = (s["maxAccumulatedFrameNum"], s["maxFastAccumulatedFrameNum"], round(s["hitDistScale"], 4), round(s["camFov"], 3),
               round(s["sunAzimuth"], 2), round(s["sunElevation"], 2), s["adaptiveAccumulation"], s["boost"], s["ortho"], s["tracingMode"])
        if key not in seen:
            seen.add(key)
            out.append(p.index)
    return out
