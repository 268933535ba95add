"""Host-side mirror of the nrd:: API over the C-ABI (ctypes).

This is the Python twin of include/NRDIntegration.h: same entry points, argument meaning and error
behaviour as the reference's ``nrd::Integration`` as used by the sample
(Source/NRDSample.cpp:440-531 ``Sample::Denoise``, :924-984 creation, :3878-3879 per-frame settings,
:4068-4154 call order).  Structures mirror include/NRDSettings.h byte for byte; ``Backend.check_abi``
compares ``ctypes.sizeof`` against the library's own ``sizeof``.

The same driver runs against two libraries exposing the same entry points under different prefixes:
``libnrdhip.so`` (prefix ``nrdhip_``, the product: HIP kernels on MI355X) and, in tests only,
``oracle/_build/liboracle.so`` (prefix ``orc_``, the CPU oracle).
"""
import ctypes as C
import enum
import os

# --------------------------------------------------------------------------------------------------
# enums (include/NRDDescs.h, include/NRDSettings.h)
# --------------------------------------------------------------------------------------------------


class Result(enum.IntEnum):
    SUCCESS = 0
    FAILURE = 1
    INVALID_ARGUMENT = 2
    UNSUPPORTED = 3
    NON_UNIQUE_IDENTIFIER = 4


class Denoiser(enum.IntEnum):
    REBLUR_DIFFUSE = 0
    REBLUR_DIFFUSE_OCCLUSION = 1
    REBLUR_DIFFUSE_SH = 2
    REBLUR_SPECULAR = 3
    REBLUR_SPECULAR_OCCLUSION = 4
    REBLUR_SPECULAR_SH = 5
    REBLUR_DIFFUSE_SPECULAR = 6
    REBLUR_DIFFUSE_SPECULAR_OCCLUSION = 7
    REBLUR_DIFFUSE_SPECULAR_SH = 8
    REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION = 9
    RELAX_DIFFUSE = 10
    RELAX_DIFFUSE_SH = 11
    RELAX_SPECULAR = 12
    RELAX_SPECULAR_SH = 13
    RELAX_DIFFUSE_SPECULAR = 14
    RELAX_DIFFUSE_SPECULAR_SH = 15
    SIGMA_SHADOW = 16
    SIGMA_SHADOW_TRANSLUCENCY = 17
    REFERENCE = 18


class ResourceType(enum.IntEnum):
    IN_MV = 0
    IN_NORMAL_ROUGHNESS = 1
    IN_VIEWZ = 2
    IN_BASECOLOR_METALNESS = 3
    IN_DIFF_CONFIDENCE = 4
    IN_SPEC_CONFIDENCE = 5
    IN_DISOCCLUSION_THRESHOLD_MIX = 6
    IN_DIFF_RADIANCE_HITDIST = 7
    IN_SPEC_RADIANCE_HITDIST = 8
    IN_DIFF_HITDIST = 9
    IN_SPEC_HITDIST = 10
    IN_DIFF_DIRECTION_HITDIST = 11
    IN_DIFF_SH0 = 12
    IN_DIFF_SH1 = 13
    IN_SPEC_SH0 = 14
    IN_SPEC_SH1 = 15
    IN_PENUMBRA = 16
    IN_TRANSLUCENCY = 17
    IN_SIGNAL = 18
    OUT_DIFF_RADIANCE_HITDIST = 19
    OUT_SPEC_RADIANCE_HITDIST = 20
    OUT_DIFF_SH0 = 21
    OUT_DIFF_SH1 = 22
    OUT_SPEC_SH0 = 23
    OUT_SPEC_SH1 = 24
    OUT_DIFF_HITDIST = 25
    OUT_SPEC_HITDIST = 26
    OUT_DIFF_DIRECTION_HITDIST = 27
    OUT_SHADOW_TRANSLUCENCY = 28
    OUT_SIGNAL = 29
    OUT_VALIDATION = 30
    TRANSIENT_POOL = 31
    PERMANENT_POOL = 32


class Format(enum.IntEnum):
    R8_UNORM = 0
    R8_UINT = 1
    RGBA8_UNORM = 2
    R16_UINT = 3
    R16_SFLOAT = 4
    RG16_SFLOAT = 5
    RGBA16_SFLOAT = 6
    R32_UINT = 7
    R32_SFLOAT = 8
    RG32_UINT = 9
    RGBA32_SFLOAT = 10
    R10_G10_B10_A2_UNORM = 11
    RGBA32_UINT = 12
    R16_UNORM = 13
    RGBA16_SNORM = 14


FORMAT_BYTES = {
    Format.R8_UNORM: 1, Format.R8_UINT: 1, Format.RGBA8_UNORM: 4, Format.R16_UINT: 2, Format.R16_SFLOAT: 2,
    Format.RG16_SFLOAT: 4, Format.RGBA16_SFLOAT: 8, Format.R32_UINT: 4, Format.R32_SFLOAT: 4, Format.RG32_UINT: 8,
    Format.RGBA32_SFLOAT: 16, Format.R10_G10_B10_A2_UNORM: 4, Format.RGBA32_UINT: 16, Format.R16_UNORM: 2,
    Format.RGBA16_SNORM: 8,
}


class CheckerboardMode(enum.IntEnum):
    OFF = 0
    BLACK = 1
    WHITE = 2


class AccumulationMode(enum.IntEnum):
    CONTINUE = 0
    RESTART = 1
    CLEAR_AND_RESTART = 2


class HitDistanceReconstructionMode(enum.IntEnum):
    OFF = 0
    AREA_3X3 = 1
    AREA_5X5 = 2


REBLUR_MAX_HISTORY_FRAME_NUM = 63
RELAX_MAX_HISTORY_FRAME_NUM = 255
SIGMA_MAX_HISTORY_FRAME_NUM = 7
REFERENCE_MAX_HISTORY_FRAME_NUM = 4095
SIGMA_DEFAULT_ACCUMULATION_TIME = 0.084


def get_max_accumulated_frame_num(accumulation_time, fps):
    """nrd::GetMaxAccumulatedFrameNum (Source/NRDSample.cpp:2167)."""
    return int(accumulation_time * fps + 0.5)


# --------------------------------------------------------------------------------------------------
# settings structs (include/NRDSettings.h) - defaults applied in __init__
# --------------------------------------------------------------------------------------------------
_f = C.c_float
_u32 = C.c_uint32
_u16 = C.c_uint16
_u8 = C.c_uint8
_b = C.c_bool


class _Struct(C.Structure):
    _defaults_ = {}

    def __init__(self, **kw):
        super().__init__()
        for k, v in self._defaults_.items():
            self._assign(k, v)
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError("%s has no field %s" % (type(self).__name__, k))
            self._assign(k, v)

    def _assign(self, k, v):
        cur = getattr(self, k)
        if isinstance(cur, C.Array):
            for i, x in enumerate(v):
                cur[i] = x
        else:
            setattr(self, k, v)

    def copy(self):
        o = type(self).__new__(type(self))
        C.memmove(C.addressof(o), C.addressof(self), C.sizeof(self))
        return o


_IDENT = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]


class CommonSettings(_Struct):
    _fields_ = [
        ("viewToClipMatrix", _f * 16), ("viewToClipMatrixPrev", _f * 16), ("worldToViewMatrix", _f * 16),
        ("worldToViewMatrixPrev", _f * 16), ("worldPrevToWorldMatrix", _f * 16),
        ("motionVectorScale", _f * 3), ("cameraJitter", _f * 2), ("cameraJitterPrev", _f * 2),
        ("resourceSize", _u16 * 2), ("resourceSizePrev", _u16 * 2), ("rectSize", _u16 * 2), ("rectSizePrev", _u16 * 2),
        ("viewZScale", _f), ("timeDeltaBetweenFrames", _f), ("denoisingRange", _f), ("disocclusionThreshold", _f),
        ("disocclusionThresholdAlternate", _f), ("cameraAttachedReflectionMaterialID", _f), ("strandMaterialID", _f),
        ("strandThickness", _f), ("splitScreen", _f),
        ("printfAt", _u16 * 2), ("debug", _f), ("rectOrigin", _u32 * 2), ("frameIndex", _u32),
        ("accumulationMode", _u8), ("isMotionVectorInWorldSpace", _b), ("isHistoryConfidenceAvailable", _b),
        ("isDisocclusionThresholdMixAvailable", _b), ("isBaseColorMetalnessAvailable", _b), ("enableValidation", _b),
    ]
    _defaults_ = dict(worldPrevToWorldMatrix=_IDENT, motionVectorScale=[1.0, 1.0, 0.0], viewZScale=1.0, denoisingRange=500000.0,
                      disocclusionThreshold=0.01, disocclusionThresholdAlternate=0.05, cameraAttachedReflectionMaterialID=999.0,
                      strandMaterialID=999.0, strandThickness=80e-6, printfAt=[9999, 9999])


class ReblurHitDistanceParameters(_Struct):
    _fields_ = [("A", _f), ("B", _f), ("C", _f), ("D", _f)]
    _defaults_ = dict(A=3.0, B=0.1, C=20.0, D=-25.0)


class ReblurAntilagSettings(_Struct):
    _fields_ = [("luminanceSigmaScale", _f), ("luminanceSensitivity", _f)]
    _defaults_ = dict(luminanceSigmaScale=4.0, luminanceSensitivity=3.0)


class ResponsiveAccumulationSettings(_Struct):
    _fields_ = [("roughnessThreshold", _f), ("minAccumulatedFrameNum", _u32)]
    _defaults_ = dict(roughnessThreshold=0.0, minAccumulatedFrameNum=3)


class ReblurSettings(_Struct):
    _fields_ = [
        ("hitDistanceParameters", ReblurHitDistanceParameters), ("antilagSettings", ReblurAntilagSettings),
        ("responsiveAccumulationSettings", ResponsiveAccumulationSettings),
        ("maxAccumulatedFrameNum", _u32), ("maxFastAccumulatedFrameNum", _u32), ("maxStabilizedFrameNum", _u32),
        ("historyFixFrameNum", _u32), ("historyFixBasePixelStride", _u32),
        ("diffusePrepassBlurRadius", _f), ("specularPrepassBlurRadius", _f), ("minHitDistanceWeight", _f),
        ("minBlurRadius", _f), ("maxBlurRadius", _f), ("lobeAngleFraction", _f), ("roughnessFraction", _f),
        ("planeDistanceSensitivity", _f), ("specularProbabilityThresholdsForMvModification", _f * 2),
        ("fireflySuppressorMinRelativeScale", _f), ("fastHistoryClampingSigmaScale", _f),
        ("checkerboardMode", _u8), ("hitDistanceReconstructionMode", _u8), ("minMaterialForDiffuse", _u8),
        ("minMaterialForSpecular", _u8), ("enableAntiFirefly", _b), ("usePrepassOnlyForSpecularMotionEstimation", _b),
        ("returnHistoryLengthInsteadOfOcclusion", _b),
    ]
    _defaults_ = dict(maxAccumulatedFrameNum=30, maxFastAccumulatedFrameNum=6, maxStabilizedFrameNum=63, historyFixFrameNum=3,
                      historyFixBasePixelStride=14, diffusePrepassBlurRadius=30.0, specularPrepassBlurRadius=50.0,
                      minHitDistanceWeight=0.1, minBlurRadius=1.0, maxBlurRadius=30.0, lobeAngleFraction=0.15,
                      roughnessFraction=0.15, planeDistanceSensitivity=0.02,
                      specularProbabilityThresholdsForMvModification=[0.5, 0.9], fireflySuppressorMinRelativeScale=2.0,
                      fastHistoryClampingSigmaScale=2.0, minMaterialForDiffuse=4, minMaterialForSpecular=4)

    def __init__(self, **kw):
        super().__init__(**kw)
        for name, cls in (("hitDistanceParameters", ReblurHitDistanceParameters), ("antilagSettings", ReblurAntilagSettings),
                          ("responsiveAccumulationSettings", ResponsiveAccumulationSettings)):
            if name not in kw:
                setattr(self, name, cls())


class RelaxAntilagSettings(_Struct):
    _fields_ = [("accelerationAmount", _f), ("spatialSigmaScale", _f), ("temporalSigmaScale", _f), ("resetAmount", _f)]
    _defaults_ = dict(accelerationAmount=0.3, spatialSigmaScale=4.5, temporalSigmaScale=0.5, resetAmount=0.5)


class RelaxSettings(_Struct):
    _fields_ = [
        ("antilagSettings", RelaxAntilagSettings),
        ("diffuseMaxAccumulatedFrameNum", _u32), ("specularMaxAccumulatedFrameNum", _u32),
        ("diffuseMaxFastAccumulatedFrameNum", _u32), ("specularMaxFastAccumulatedFrameNum", _u32),
        ("historyFixFrameNum", _u32), ("historyFixBasePixelStride", _u32),
        ("spatialVarianceEstimationHistoryThreshold", _u32), ("atrousIterationNum", _u32),
        ("diffusePrepassBlurRadius", _f), ("specularPrepassBlurRadius", _f), ("historyFixEdgeStoppingNormalPower", _f),
        ("fastHistoryClampingSigmaScale", _f), ("diffusePhiLuminance", _f), ("specularPhiLuminance", _f),
        ("diffuseMinLuminanceWeight", _f), ("specularMinLuminanceWeight", _f), ("lobeAngleFraction", _f),
        ("roughnessFraction", _f), ("specularVarianceBoost", _f), ("specularLobeAngleSlack", _f), ("depthThreshold", _f),
        ("minHitDistanceWeight", _f), ("luminanceEdgeStoppingRelaxation", _f), ("normalEdgeStoppingRelaxation", _f),
        ("roughnessEdgeStoppingRelaxation", _f), ("confidenceDrivenRelaxationMultiplier", _f),
        ("confidenceDrivenLuminanceEdgeStoppingRelaxation", _f), ("confidenceDrivenNormalEdgeStoppingRelaxation", _f),
        ("checkerboardMode", _u8), ("hitDistanceReconstructionMode", _u8), ("minMaterialForDiffuse", _u8),
        ("minMaterialForSpecular", _u8), ("enableAntiFirefly", _b), ("enableRoughnessEdgeStopping", _b),
    ]
    _defaults_ = dict(diffuseMaxAccumulatedFrameNum=30, specularMaxAccumulatedFrameNum=30, diffuseMaxFastAccumulatedFrameNum=6,
                      specularMaxFastAccumulatedFrameNum=6, historyFixFrameNum=3, historyFixBasePixelStride=14,
                      spatialVarianceEstimationHistoryThreshold=3, atrousIterationNum=5, diffusePrepassBlurRadius=30.0,
                      specularPrepassBlurRadius=50.0, historyFixEdgeStoppingNormalPower=8.0, fastHistoryClampingSigmaScale=2.0,
                      diffusePhiLuminance=2.0, specularPhiLuminance=1.0, lobeAngleFraction=0.5, roughnessFraction=0.15,
                      specularLobeAngleSlack=0.15, depthThreshold=0.003, minHitDistanceWeight=0.1,
                      luminanceEdgeStoppingRelaxation=0.5, normalEdgeStoppingRelaxation=0.3, roughnessEdgeStoppingRelaxation=1.0,
                      minMaterialForDiffuse=4, minMaterialForSpecular=4, enableRoughnessEdgeStopping=True)

    def __init__(self, **kw):
        super().__init__(**kw)
        if "antilagSettings" not in kw:
            self.antilagSettings = RelaxAntilagSettings()


class SigmaSettings(_Struct):
    _fields_ = [("lightDirection", _f * 3), ("planeDistanceSensitivity", _f), ("maxStabilizedFrameNum", _u32)]
    _defaults_ = dict(planeDistanceSensitivity=0.02, maxStabilizedFrameNum=5)


class ReferenceSettings(_Struct):
    _fields_ = [("maxAccumulatedFrameNum", _u32)]
    _defaults_ = dict(maxAccumulatedFrameNum=1024)


# --------------------------------------------------------------------------------------------------
# C-ABI structs (include/nrdhip.h)
# --------------------------------------------------------------------------------------------------
class DenoiserDesc(C.Structure):
    _fields_ = [("identifier", _u32), ("denoiser", _u32)]


class CreateDesc(C.Structure):
    _fields_ = [("denoisers", C.POINTER(DenoiserDesc)), ("denoisers_num", _u32), ("resource_width", _u16),
                ("resource_height", _u16), ("frame_height", _u16), ("band_own_first", _u16), ("band_own_rows", _u16),
                ("device_plus1", _u16), ("band_row0", C.c_int32), ("flags", _u32)]


class PlaneInfo(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("pitch_bytes", _u32), ("format", _u32), ("width", _u16), ("height", _u16),
                ("bytes_per_texel", _u32), ("name", C.c_char_p)]


class DispatchInfo(C.Structure):
    _fields_ = [("name", C.c_char_p), ("kernel", C.c_char_p), ("identifier", _u32), ("grid_width", _u16),
                ("grid_height", _u16), ("halo_rows", _u16), ("written_num", _u16), ("written", _u32 * 12),
                ("read_num", _u32), ("read", _u32 * 24), ("algorithmic_bytes_per_pixel", _f), ("read_rows", _u16 * 24), ("flags", _u32),
                ("written_prefix", _u32 * 12)]


NO_PLANE = 0xFFFFFFFF      # nrdhip_dispatch_info.written_prefix: the plane travels as it is (NRDHIP_NO_PLANE)
READ_REPROJECTED = 0xFFFF  # nrdhip_dispatch_info.read_rows: previous-frame state read at motion-displaced positions (NRDHIP_READ_REPROJECTED)
DISPATCH_ALL_ROWS = 1      # nrdhip_dispatch_info.flags: pointwise pass that runs on every stored row of a band (NRDHIP_DISPATCH_ALL_ROWS)


class ConfidenceBlurDesc(C.Structure):
    """nrdhip_confidence_blur_desc (Shaders/ConfidenceBlur.cs.hlsl, Source/NRDSample.cpp:3999-4026)"""
    _fields_ = [("ping", C.c_void_p), ("pong", C.c_void_p), ("pitch_bytes", _u32), ("width", _u16), ("height", _u16),
                ("camera_frustum", _f * 4), ("inv_size", _f * 2), ("rect_width", _f), ("unproject", _f), ("ortho_mode", _f),
                ("frame_index", _u32), ("max_accumulated_frame_num", _u32), ("relax", _u32), ("first_pass", _u32), ("passes_num", _u32)]


class UnpackDesc(C.Structure):
    """nrdhip_unpack_desc (NRD-facing part of Shaders/Composition.cs.hlsl:57-64, 74-175)"""
    _fields_ = [("width", _u16), ("height", _u16), ("mode", _u32), ("relax", _u32), ("resolve", _u32),
                ("diff", C.c_void_p), ("diff_pitch", _u32), ("spec", C.c_void_p), ("spec_pitch", _u32),
                ("diff_sh1", C.c_void_p), ("diff_sh1_pitch", _u32), ("spec_sh1", C.c_void_p), ("spec_sh1_pitch", _u32),
                ("normal_roughness", C.c_void_p), ("normal_roughness_pitch", _u32),
                ("shadow", C.c_void_p), ("shadow_pitch", _u32), ("shadow_bytes_per_texel", _u32),
                ("out_diff", C.c_void_p), ("out_diff_pitch", _u32), ("out_spec", C.c_void_p), ("out_spec_pitch", _u32),
                ("out_shadow", C.c_void_p), ("out_shadow_pitch", _u32),
                ("view_to_world", _f * 9), ("camera_frustum", _f * 4), ("inv_rect_size", _f * 2)]


class TaaDesc(C.Structure):
    """nrdhip_taa_desc (Shaders/Taa.cs.hlsl)"""
    _fields_ = [("mv", C.c_void_p), ("mv_pitch", _u32), ("composed", C.c_void_p), ("composed_pitch", _u32),
                ("history", C.c_void_p), ("history_pitch", _u32), ("result", C.c_void_p), ("result_pitch", _u32),
                ("rect_width", _u16), ("rect_height", _u16), ("rect_width_prev", _u16), ("rect_height_prev", _u16),
                ("render_width", _u16), ("render_height", _u16), ("tonemap", _u32), ("hdr_scale", _f), ("taa", _f)]


def _plane_fields(*names):
    out = []
    for n in names:
        out += [(n, C.c_void_p), (n + "_pitch", _u32)]
    return out


class FrontendPackDesc(C.Structure):
    """nrdhip_frontend_pack_desc (producer side, Shaders/TraceOpaque.cs.hlsl:421, :657, :738-757, :800-801 over include/nrd_frontend.h)"""
    _fields_ = [("width", _u16), ("height", _u16), ("mode", _u32), ("relax", _u32), ("sanitize", _u32), ("hit_distance_parameters", _f * 4),
                ("tan_of_light_angular_radius", _f)] + _plane_fields(
        "normal", "material_id", "viewz", "diff", "spec", "diff_direction", "spec_direction", "shadow", "out_normal_roughness", "out_diff", "out_spec",
        "out_diff_sh1", "out_spec_sh1", "out_penumbra", "out_translucency")


class ComposeDesc(C.Structure):
    """nrdhip_compose_desc (Shaders/Composition.cs.hlsl:92-107 re-jitter, :183-188 material re-modulation)"""
    _fields_ = [("width", _u16), ("height", _u16), ("sh", _u32), ("relax", _u32), ("hair_material_id", _u32)] + _plane_fields(
        "diff", "spec", "diff_sh0", "diff_sh1", "spec_sh0", "spec_sh1", "normal_roughness", "viewz", "base_color_metalness", "out_diff", "out_spec") + [
        ("view_to_world", _f * 9), ("camera_frustum", _f * 4), ("inv_rect_size", _f * 2)]


class HistoryState(C.Structure):  # nrdhip_history_state
    _fields_ = [("frame_counter", _u32), ("frames_since_reset", _u32), ("history_valid", _u32), ("reserved", _u32)]


UNPACK_NORMAL, UNPACK_OCCLUSION, UNPACK_SH, PACK_DIRECTIONAL_OCCLUSION = 0, 1, 2, 3
FLAG_EXTERNAL_POOLS = 1
FLAG_GRAPH = 2  # nrdhip_denoise replays one HIP graph per frame (include/nrdhip.h NRDHIP_FLAG_GRAPH)
FLAG_SEPARATE_PASSES = 4  # one dispatch per pass: no fused REBLUR::PrePassTemporalAccumulation (NRDHIP_FLAG_SEPARATE_PASSES)


class NrdError(RuntimeError):
    def __init__(self, what, code, text=""):
        self.code = Result(code) if code in Result._value2member_map_ else code
        super().__init__("%s failed: %s %s" % (what, self.code, text))


# nrdhip_transport (include/nrdhip.h): caller-supplied halo-row transport of the row tiler
TRANSPORT_BEGIN = C.CFUNCTYPE(C.c_int, C.c_void_p)
TRANSPORT_XFER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
TRANSPORT_END = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)


class Transport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("group_begin", TRANSPORT_BEGIN), ("send", TRANSPORT_XFER), ("recv", TRANSPORT_XFER),
                ("group_end", TRANSPORT_END), ("flags", _u32)]  # flags: NRDHIP_TRANSPORT_STREAM_ORDERED = 1


class Backend:
    """A loaded library exposing the C-ABI of include/nrdhip.h under ``prefix``."""

    def __init__(self, path, prefix, device):
        if not os.path.exists(path):
            raise FileNotFoundError(
                "%s is missing - run `python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)" % path)
        self.path, self.prefix, self.device = path, prefix, device
        self.lib = C.CDLL(path)
        self._fn = {}
        self._sig("create", C.c_int, [C.POINTER(CreateDesc), C.POINTER(C.c_void_p)])
        self._sig("destroy", None, [C.c_void_p])
        self._sig("new_frame", C.c_int, [C.c_void_p])
        self._sig("set_common", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t])
        self._sig("set_denoiser", C.c_int, [C.c_void_p, _u32, C.c_void_p, C.c_size_t])
        self._sig("bind", C.c_int, [C.c_void_p, _u32, C.c_void_p, _u32, _u32, _u16, _u16])
        self._sig("denoise", C.c_int, [C.c_void_p, C.POINTER(_u32), _u32, C.c_void_p])
        self._sig("get_history_state", C.c_int, [C.c_void_p, _u32, C.POINTER(HistoryState)])
        self._sig("set_history_state", C.c_int, [C.c_void_p, _u32, C.POINTER(HistoryState)])
        self._sig("set_history_rows", C.c_int, [C.c_void_p, C.c_int32, _u32])
        if hasattr(self.lib, self.prefix + "graph_stats"):  # product library (the CPU oracle of the tests has no graphs)
            self._sig("graph_stats", C.c_int, [C.c_void_p, C.POINTER(_u32)])
        self._sig("dispatch_count", C.c_int, [C.c_void_p, C.POINTER(_u32), _u32, C.POINTER(_u32)])
        self._sig("dispatch_info_get", C.c_int, [C.c_void_p, C.POINTER(_u32), _u32, _u32, C.POINTER(DispatchInfo)])
        self._sig("denoise_range", C.c_int, [C.c_void_p, C.POINTER(_u32), _u32, _u32, _u32, C.c_void_p])
        self._sig("denoise_rows", C.c_int, [C.c_void_p, C.POINTER(_u32), _u32, _u32, _u32, _u32, _u32, C.c_void_p])
        self._sig("pool_size", C.c_int, [C.c_void_p, _u32, C.POINTER(_u32)])
        self._sig("pool_info", C.c_int, [C.c_void_p, _u32, _u32, C.POINTER(PlaneInfo)])
        self._sig("bind_pool", C.c_int, [C.c_void_p, _u32, _u32, C.c_void_p, _u32])
        self._sig("get_memory_mb", C.c_int, [C.c_void_p, C.POINTER(_f)])
        self._sig("sizeof", _u32, [_u32])
        self._sig("last_error", C.c_char_p, [C.c_void_p])
        self._sig("confidence_blur", C.c_int, [C.POINTER(ConfidenceBlurDesc), C.c_void_p])
        self._sig("backend_unpack", C.c_int, [C.POINTER(UnpackDesc), C.c_void_p])
        self._sig("taa", C.c_int, [C.POINTER(TaaDesc), C.c_void_p])
        # introspection + row tiler: part of the product library (and of its host-emulated build); the CPU oracle of the tests
        # implements the per-instance entry points only
        self.has_frontend = hasattr(self.lib, self.prefix + "frontend_pack")
        if self.has_frontend:
            self._sig("frontend_pack", C.c_int, [C.POINTER(FrontendPackDesc), C.c_void_p])
            self._sig("compose", C.c_int, [C.POINTER(ComposeDesc), C.c_void_p])
        self.has_tiler = hasattr(self.lib, self.prefix + "tiler_create")
        if self.has_tiler:
            self._sig("denoiser_kind", C.c_int, [C.c_void_p, _u32, C.POINTER(_u32)])
            self._sig("get_band", C.c_int, [C.c_void_p, C.POINTER(C.c_int32)])
            self._sig("slot_info", C.c_int, [C.c_void_p, _u32, C.POINTER(PlaneInfo)])
            self._sig("required_halo", C.c_int, [C.c_void_p, C.POINTER(_u32), _u32, _u32, C.POINTER(_u32)])
            self._sig("tiler_create", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Transport), C.POINTER(C.c_void_p)])
            self._sig("tiler_destroy", None, [C.c_void_p])
            self._sig("tiler_rccl_unique_id", C.c_int, [C.c_void_p])
            self._sig("tiler_rccl_init", C.c_int, [C.c_void_p, C.c_void_p])
            if hasattr(self.lib, self.prefix + "tiler_rccl_loopback"):
                self._sig("tiler_rccl_loopback", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p])
            self._sig("tiler_exchange_inputs", C.c_int, [C.c_void_p, C.POINTER(_u32), _u32, C.c_void_p])
            self._sig("tiler_denoise", C.c_int, [C.c_void_p, C.POINTER(_u32), _u32, C.c_void_p])
            self._sig("tiler_finish", C.c_int, [C.c_void_p, C.c_void_p])
            self._sig("tiler_halo", C.c_int, [C.c_void_p, C.POINTER(_u32)])
            self._sig("tiler_stats", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)])
            self._sig("tiler_last_error", C.c_char_p, [C.c_void_p])

    def _sig(self, name, res, args):
        f = getattr(self.lib, self.prefix + name)
        f.restype, f.argtypes = res, args
        self._fn[name] = f

    def __getattr__(self, name):
        fn = self.__dict__.get("_fn", {})
        if name in fn:
            return fn[name]
        raise AttributeError(name)

    def check_abi(self):
        want = [CommonSettings, ReblurSettings, RelaxSettings, SigmaSettings, ReferenceSettings, CreateDesc, PlaneInfo, DispatchInfo,
                ConfidenceBlurDesc, UnpackDesc, TaaDesc] + ([FrontendPackDesc, ComposeDesc] if self.has_frontend else [])
        for i, cls in enumerate(want):
            got = self.sizeof(i)
            if got != C.sizeof(cls):
                raise RuntimeError("ABI mismatch: sizeof(%s) python=%d library=%d" % (cls.__name__, C.sizeof(cls), got))
        if self.has_tiler and self.sizeof(13) != C.sizeof(Transport):  # (copied by value in nrdhip_tiler_create: a stale layout would hand it garbage flags)
            raise RuntimeError("ABI mismatch: sizeof(nrdhip_transport) python=%d library=%d" % (C.sizeof(Transport), self.sizeof(13)))
        return True

    @property
    def is_device(self):
        return self.device != "cpu"


def _ptr_pitch(buf):
    """(pointer, row pitch in bytes, width, height) of a 2-D/3-D numpy array or torch tensor (row-major, contiguous rows)."""
    if hasattr(buf, "data_ptr"):  # torch
        assert buf.stride(-1) == 1 or buf.dim() == 2
        return buf.data_ptr(), buf.stride(0) * buf.element_size(), buf.shape[1], buf.shape[0]
    return buf.ctypes.data, buf.strides[0], buf.shape[1], buf.shape[0]


class Integration:
    """Twin of nrd::Integration (Source/NRDSample.cpp:625, :982, :3878-3879, :4080-4152, :744) over the C-ABI."""

    def __init__(self, backend):
        self.backend = backend
        self.handle = None
        self.pools = {0: [], 1: []}
        self._bound = {}

    # nrd::Integration::Recreate(IntegrationCreationDesc, InstanceCreationDesc, device) -> Result
    def recreate(self, denoisers, resource_width, resource_height, frame_height=0, band_row0=0, band_own_first=0, band_own_rows=0, graph=False,
                 separate_passes=False):
        """``denoisers``: list of (identifier, Denoiser). Returns Result (SUCCESS or the failure code), never raises for
        library-side failures - the sample tests ``!= SUCCESS`` (Source/NRDSample.cpp:982-983). ``separate_passes``: one dispatch per pass of
        the pass graph (no fused REBLUR::PrePassTemporalAccumulation) - for callers that want the PrePass result as a plane."""
        self.destroy()
        arr = (DenoiserDesc * len(denoisers))(*[DenoiserDesc(int(i), int(d)) for i, d in denoisers])
        dev = self.backend.device
        device_plus1 = int(dev.split(":")[1]) + 1 if isinstance(dev, str) and dev.startswith("cuda:") else 0  # kernels + checks on the device torch allocates on
        desc = CreateDesc(arr, len(denoisers), resource_width, resource_height, frame_height, band_own_first, band_own_rows, device_plus1,
                          band_row0, FLAG_EXTERNAL_POOLS | (FLAG_GRAPH if graph else 0) | (FLAG_SEPARATE_PASSES if separate_passes else 0))
        h = C.c_void_p()
        r = self.backend.create(C.byref(desc), C.byref(h))
        if r != 0:
            return Result(r)
        self.handle = h
        self.denoisers = list(denoisers)
        self._allocate_pools()
        return Result.SUCCESS

    def _alloc(self, nbytes):
        if self.backend.is_device:
            import torch
            return torch.zeros(nbytes, dtype=torch.uint8, device=self.backend.device)
        import numpy as np
        return np.zeros(nbytes, dtype=np.uint8)

    def _allocate_pools(self):
        for pool in (0, 1):
            n = _u32()
            self.backend.pool_size(self.handle, pool, C.byref(n))
            self.pools[pool] = []
            for i in range(n.value):
                info = PlaneInfo()
                self.backend.pool_info(self.handle, pool, i, C.byref(info))
                pitch = info.width * info.bytes_per_texel
                buf = self._alloc(pitch * info.height).reshape(info.height, pitch)
                ptr = buf.data_ptr() if hasattr(buf, "data_ptr") else buf.ctypes.data
                self._check(self.backend.bind_pool(self.handle, pool, i, ptr, pitch), "bind_pool")
                self.pools[pool].append(dict(name=info.name.decode(), buf=buf, format=Format(info.format), width=info.width,
                                             height=info.height, bpt=info.bytes_per_texel))

    def pool_plane(self, name):
        for pool in (0, 1):
            for p in self.pools[pool]:
                if p["name"] == name:
                    return p
        raise KeyError(name)

    def _check(self, r, what):
        if r != 0:
            raise NrdError(what, r, (self.backend.last_error(self.handle) or b"").decode())

    def new_frame(self):
        self._check(self.backend.new_frame(self.handle), "NewFrame")

    def set_history_rows(self, first_local_row, rows):
        """row tiling: the local rows on which last frame's permanent planes are current (include/nrdhip.h nrdhip_set_history_rows); rows = 0: all"""
        self._check(self.backend.set_history_rows(self.handle, int(first_local_row), int(rows)), "set_history_rows")
        self.history_rows = (int(first_local_row), int(rows))

    def set_common_settings(self, cs):
        self._check(self.backend.set_common(self.handle, C.byref(cs), C.sizeof(cs)), "SetCommonSettings")

    def set_denoiser_settings(self, identifier, settings):
        self._check(self.backend.set_denoiser(self.handle, int(identifier), C.byref(settings), C.sizeof(settings)), "SetDenoiserSettings")

    # nrd::ResourceSnapshot::SetResource
    def set_resource(self, slot, buf, fmt, width=None, height=None):
        """Bind a caller-owned plane. ``buf``: 2-D byte view (rows x pitch bytes) or any array whose rows are contiguous."""
        if hasattr(buf, "data_ptr"):
            ptr, pitch = buf.data_ptr(), buf.stride(0) * buf.element_size()
        else:
            ptr, pitch = buf.ctypes.data, buf.strides[0]
        h = buf.shape[0] if height is None else height
        w = (pitch // FORMAT_BYTES[Format(fmt)]) if width is None else width
        self._bound[int(slot)] = buf  # keep alive
        self._check(self.backend.bind(self.handle, int(slot), ptr, pitch, int(fmt), w, h), "SetResource")

    def _ids(self, identifiers):
        return (_u32 * len(identifiers))(*[int(i) for i in identifiers]), len(identifiers)

    def _stream(self):
        if self.backend.is_device:
            import torch
            return torch.cuda.current_stream().cuda_stream
        return None

    # nrd::Integration::Denoise(identifiers, n, commandBuffer, resourceSnapshot)
    def denoise(self, identifiers):
        ids, n = self._ids(identifiers)
        self._check(self.backend.denoise(self.handle, ids, n, self._stream()), "Denoise")

    # checkpoint / resume (include/nrdhip.h nrdhip_history_state): permanent planes + per-denoiser counters
    def save_history(self, path):
        """write every permanent plane and the history counters of every denoiser to an .npz - between frames"""
        import numpy as np

        out = {}
        for i, p in enumerate(self.pools[0]):
            buf = p["buf"]
            out["plane%d" % i] = buf.cpu().numpy() if hasattr(buf, "cpu") else np.asarray(buf)
            out["name%d" % i] = np.array(p["name"])
        for ident, _ in self.denoisers:
            st = HistoryState()
            self._check(self.backend.get_history_state(self.handle, int(ident), C.byref(st)), "get_history_state")
            out["state%d" % int(ident)] = np.array([st.frame_counter, st.frames_since_reset, st.history_valid], dtype=np.uint32)
        np.savez(path, **out)

    def load_history(self, path):
        """restore what save_history wrote into an instance created with the same denoisers and size - before the next frame"""
        import numpy as np

        z = np.load(path)
        for i, p in enumerate(self.pools[0]):
            if str(z["name%d" % i]) != p["name"] or z["plane%d" % i].shape != tuple(p["buf"].shape):
                raise ValueError("history file does not match this instance: plane %d (%s)" % (i, p["name"]))
            src = z["plane%d" % i]
            if hasattr(p["buf"], "copy_"):
                import torch

                p["buf"].copy_(torch.from_numpy(src))
            else:
                p["buf"][...] = src
        for ident, _ in self.denoisers:
            a = z["state%d" % int(ident)]
            st = HistoryState(int(a[0]), int(a[1]), int(a[2]), 0)
            self._check(self.backend.set_history_state(self.handle, int(ident), C.byref(st)), "set_history_state")

    def graph_stats(self):
        """NRDHIP_FLAG_GRAPH bookkeeping: dict(replayed, instantiated, direct)"""
        out = (_u32 * 3)()
        self._check(self.backend.graph_stats(self.handle, out), "graph_stats")
        return dict(replayed=out[0], instantiated=out[1], direct=out[2])

    def dispatches(self, identifiers):
        ids, n = self._ids(identifiers)
        cnt = _u32()
        self._check(self.backend.dispatch_count(self.handle, ids, n, C.byref(cnt)), "dispatch_count")
        out = []
        for i in range(cnt.value):
            di = DispatchInfo()
            self._check(self.backend.dispatch_info_get(self.handle, ids, n, i, C.byref(di)), "dispatch_info")
            out.append(dict(name=di.name.decode(), kernel=di.kernel.decode(), identifier=di.identifier,
                            grid=(di.grid_width, di.grid_height), halo_rows=di.halo_rows,
                            written=[di.written[k] for k in range(di.written_num)], read=[di.read[k] for k in range(di.read_num)],
                            read_rows=[di.read_rows[k] for k in range(di.read_num)], all_rows=bool(di.flags & DISPATCH_ALL_ROWS),
                            # tap-texel planes {guide texel | signal}: written plane -> the (guide) plane its texels start with
                            written_prefix={di.written[k]: di.written_prefix[k] for k in range(di.written_num) if di.written_prefix[k] != NO_PLANE},
                            bytes_per_pixel=di.algorithmic_bytes_per_pixel))
        return out

    def denoise_range(self, identifiers, first, count):
        ids, n = self._ids(identifiers)
        self._check(self.backend.denoise_range(self.handle, ids, n, first, count, self._stream()), "Denoise(range)")

    PART_FIRST, PART_LAST = 1, 2

    def denoise_rows(self, identifiers, index, row_first, row_count, part=3):
        """one dispatch restricted to local rows [row_first, row_first + row_count) (nrdhip_denoise_rows): row-tiling hosts
        launch boundary strips first so that their halo exchange overlaps the interior"""
        ids, n = self._ids(identifiers)
        self._check(self.backend.denoise_rows(self.handle, ids, n, index, row_first, row_count, part, self._stream()), "Denoise(rows)")

    def memory_usage_mb(self):
        out = (_f * 3)()
        self._check(self.backend.get_memory_mb(self.handle, out), "GetMemoryUsage")
        return dict(total=out[0], persistent=out[1], aliasable=out[2])

    def destroy(self):
        if self.handle is not None:
            self.backend.destroy(self.handle)
            self.handle = None
        self.pools = {0: [], 1: []}
        self._bound = {}

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
