"""Headless twin of the NRD part of Sample::RenderFrame (Source/NRDSample.cpp:3835-3879, :4068-4154, :4213-4227):
NewFrame -> SetCommonSettings -> {SetDenoiserSettings -> Denoise}* with every resource slot bound the way
Sample::Denoise does (:447-501). Works on numpy arrays (CPU oracle, tests only) or torch tensors (HIP backend)."""
import numpy as np

from . import api

RT = api.ResourceType
F = api.Format

# slot -> (key in the synthetic frame dict, format)
INPUT_SLOTS = {
    RT.IN_MV: ("mv", F.RGBA16_SFLOAT),
    RT.IN_NORMAL_ROUGHNESS: ("normal_roughness", F.R10_G10_B10_A2_UNORM),
    RT.IN_VIEWZ: ("viewz", F.R32_SFLOAT),
    RT.IN_DIFF_RADIANCE_HITDIST: ("diff", F.RGBA16_SFLOAT),
    RT.IN_SPEC_RADIANCE_HITDIST: ("spec", F.RGBA16_SFLOAT),
    RT.IN_DIFF_CONFIDENCE: ("confidence", F.RGBA16_SFLOAT),
    RT.IN_SPEC_CONFIDENCE: ("confidence", F.RGBA16_SFLOAT),
    RT.IN_PENUMBRA: ("penumbra", F.R16_SFLOAT),
    RT.IN_TRANSLUCENCY: ("translucency", F.RGBA8_UNORM),
    RT.IN_SIGNAL: ("signal", F.RGBA16_SFLOAT),
    RT.IN_DIFF_HITDIST: ("diff_hitdist", F.R16_UNORM),  # OCCLUSION variants (Source/NRDSample.cpp:488-501)
    RT.IN_SPEC_HITDIST: ("spec_hitdist", F.R16_UNORM),
    # SH variants: SH0 shares the radiance texture, SH1 is a second one (Source/NRDSample.cpp:464-476)
    RT.IN_DIFF_SH0: ("diff", F.RGBA16_SFLOAT),
    RT.IN_DIFF_SH1: ("diff_sh1", F.RGBA16_SFLOAT),
    RT.IN_SPEC_SH0: ("spec", F.RGBA16_SFLOAT),
    RT.IN_SPEC_SH1: ("spec_sh1", F.RGBA16_SFLOAT),
    RT.IN_DIFF_DIRECTION_HITDIST: ("diff_dirocc", F.RGBA16_SFLOAT),  # DIRECTIONAL_OCCLUSION (Source/NRDSample.cpp:488-491)
    RT.IN_DISOCCLUSION_THRESHOLD_MIX: ("disocclusion_mix", F.R8_UNORM),  # optional; the sample never binds it
}
OUTPUT_SLOTS = {
    RT.OUT_DIFF_RADIANCE_HITDIST: ("out_diff", F.RGBA16_SFLOAT, 8),
    RT.OUT_SPEC_RADIANCE_HITDIST: ("out_spec", F.RGBA16_SFLOAT, 8),
    RT.OUT_SHADOW_TRANSLUCENCY: ("out_shadow", F.RGBA8_UNORM, 4),
    RT.OUT_VALIDATION: ("out_validation", F.RGBA8_UNORM, 4),
    RT.OUT_DIFF_HITDIST: ("out_diff_hitdist", F.R16_UNORM, 2),
    RT.OUT_SPEC_HITDIST: ("out_spec_hitdist", F.R16_UNORM, 2),
    RT.OUT_DIFF_SH0: ("out_diff", F.RGBA16_SFLOAT, 8),
    RT.OUT_DIFF_SH1: ("out_diff_sh1", F.RGBA16_SFLOAT, 8),
    RT.OUT_SPEC_SH0: ("out_spec", F.RGBA16_SFLOAT, 8),
    RT.OUT_SPEC_SH1: ("out_spec_sh1", F.RGBA16_SFLOAT, 8),
    RT.OUT_DIFF_DIRECTION_HITDIST: ("out_diff_dirocc", F.RGBA16_SFLOAT, 8),
}


def as_bytes2d(a):
    """view an [H, W(, C)] array as [H, row bytes] uint8 without copying"""
    a = np.ascontiguousarray(a)
    return a.view(np.uint8).reshape(a.shape[0], -1)


class Harness:
    def __init__(self, backend, denoisers, width, height, **band):
        """denoisers: list of api.Denoiser; identifier = enum value (the sample's NRD_ID macro, NRDSample.cpp:226)."""
        self.backend, self.w, self.h = backend, width, height
        self.nrd = api.Integration(backend)
        r = self.nrd.recreate([(int(d), d) for d in denoisers], width, height, **band)
        if r != api.Result.SUCCESS:
            raise api.NrdError("Recreate", int(r))
        self.denoisers = list(denoisers)
        self.outputs = {}
        for slot, (key, fmt, bpt) in OUTPUT_SLOTS.items():
            if key not in self.outputs:  # SH0 outputs share the plain radiance output planes, like the sample's textures
                self.outputs[key] = self._zeros(height, width * bpt)
        self.resident = {}
        # plane key -> api.Format for slots bound in another allowed format of the same texel size (e.g. the sample's RGBA16_SNORM
        # DIRECTIONAL_OCCLUSION planes, Source/NRDSample.cpp:2937: {"diff_dirocc": F.RGBA16_SNORM, "out_diff_dirocc": F.RGBA16_SNORM})
        self.format_override = {}

    def _zeros(self, rows, rowbytes):
        if self.backend.is_device:
            import torch
            return torch.zeros((rows, rowbytes), dtype=torch.uint8, device=self.backend.device)
        return np.zeros((rows, rowbytes), dtype=np.uint8)

    def upload(self, frame):
        """numpy frame dict -> byte planes resident where the backend computes"""
        out = {}
        for key in sorted(set(k for k, _ in INPUT_SLOTS.values())):
            if key not in frame:
                continue
            if hasattr(frame[key], "data_ptr"):  # already a device tensor (frames generated on the GPU)
                import torch
                t = frame[key].contiguous()
                out[key] = t.view(torch.uint8).reshape(t.shape[0], -1)
                continue
            b = as_bytes2d(frame[key])
            if self.backend.is_device:
                import torch
                b = torch.from_numpy(b.copy()).to(self.backend.device)
            else:
                b = b.copy()
            out[key] = b
        return out

    def bind(self, planes):
        for slot, (key, fmt) in INPUT_SLOTS.items():
            if key in planes:
                buf = planes[key]
                fmt = self.format_override.get(key, fmt)
                bpt = api.FORMAT_BYTES[fmt]
                self.nrd.set_resource(slot, buf, fmt, width=buf.shape[1] // bpt, height=buf.shape[0])
        for slot, (key, fmt, bpt) in OUTPUT_SLOTS.items():
            self.nrd.set_resource(slot, self.outputs[key], self.format_override.get(key, fmt), width=self.w, height=self.h)
        # REFERENCE runs in place: the sample binds the same texture to IN_SIGNAL and OUT_SIGNAL (NRDSample.cpp:484-485)
        if "signal" in planes:
            self.nrd.set_resource(RT.OUT_SIGNAL, planes["signal"], F.RGBA16_SFLOAT, width=self.w, height=self.h)

    def frame(self, common, planes, settings=None, order=None):
        """one frame: settings = {Denoiser: settings struct}; order = list of lists of denoisers per Denoise() call"""
        settings = settings or {}
        self.nrd.new_frame()
        self.nrd.set_common_settings(common)
        self.bind(planes)
        calls = order or [[d] for d in self.denoisers]
        for call in calls:
            for d in call:
                if d in settings:
                    self.nrd.set_denoiser_settings(int(d), settings[d])
            self.nrd.denoise([int(d) for d in call])

    def fetch(self, buf):
        if hasattr(buf, "cpu"):
            return buf.cpu().numpy()
        return buf

    def output(self, key, dtype=np.float16, channels=4):
        raw = self.fetch(self.outputs[key])
        return raw.view(dtype).reshape(self.h, self.w, channels) if channels > 1 else raw.view(dtype).reshape(self.h, self.w)

    def pool(self, name):
        p = self.nrd.pool_plane(name)
        return self.fetch(p["buf"])


CHECKERBOARD_KEYS = ("diff", "spec", "diff_sh1", "spec_sh1", "diff_hitdist", "spec_hitdist", "diff_dirocc")


def to_checkerboard(frame, frame_index, white=True):
    """Dense synthetic frame -> the half-width checkerboarded inputs the sample produces in RESOLUTION_HALF tracing
    (Shaders/TraceOpaque.cs.hlsl:482-508): pixel (x, y) lands at texel (x >> 1, y); with CheckerboardMode::WHITE the diffuse
    signal is kept on the squares where Sequence::CheckerBoard(pixelPos, frameIndex) = ((x ^ y) ^ frameIndex) & 1 is 1, the
    specular signal on the 0 squares (BLACK swaps them). Works on numpy arrays and torch tensors."""
    out = dict(frame)
    for key in CHECKERBOARD_KEYS:
        if key not in frame:
            continue
        a = frame[key]
        h, w = a.shape[0], a.shape[1]
        spec = key.startswith("spec")
        phase = (1 if white else 0) ^ (1 if spec else 0)
        half = a[:, 0::2].clone() if hasattr(a, "clone") else a[:, 0::2].copy()  # (w + 1) // 2 texels per row
        for y in range(2):  # row parity decides which pixel of the pair (2k, 2k+1) carries the signal
            # pixel x carries it iff ((x ^ y) ^ frame) & 1 == phase  ->  x parity = phase ^ y ^ frame
            xpar = (phase ^ y ^ frame_index) & 1
            src = a[y::2, xpar::2]
            half[y::2, : src.shape[1]] = src
        out[key] = half
    return out


def pingpong(n, f):
    """camera path position of step f: 0,1,..,n-1,n-2,..,1,0,1,.. (motion vectors stay consistent in both directions)"""
    if n == 1:
        return 0
    period = 2 * (n - 1)
    k = f % period
    return k if k < n else period - k
