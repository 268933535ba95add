"""NRDHIP_FLAG_GRAPH (include/nrdhip.h): nrdhip_denoise replays one HIP graph per frame - the dispatch list is stream-captured every
frame and the instance's executable graph is patched with the new kernel arguments. Results must be bit-identical to pass-by-pass
launches over a sequence that changes the graph's shape (CLEAR_AND_RESTART frames carry the pool clears) and its arguments (every
frame: matrices, frame index, DRS rect); on a stream that cannot be captured the instance falls back to direct launches."""
import numpy as np
import pytest

import util


def run_pair(pkg, api, backend, dens, frames, w=320, h=192, stream=None, rects=None, split_calls=False, sync_every_frame=True):
    import contextlib

    scene = pkg.synth.Scene(w, h, dolly=0.02, denoiser="RELAX" if dens[0].startswith("RELAX") else "REBLUR")
    dd = [api.Denoiser[x] for x in dens]
    st = util.default_settings(api, scene, dd, minMaterialForDiffuse=0, minMaterialForSpecular=1)
    ha = pkg.harness.Harness(backend, dd, w, h)
    hb = pkg.harness.Harness(backend, dd, w, h, graph=True)
    ctx = contextlib.nullcontext()
    if stream is not None:
        import torch

        torch.cuda.synchronize()
        ctx = torch.cuda.stream(stream)
    with ctx:
        for f in range(frames):
            fr = scene.frame(f)
            cs = scene.common_settings(api, fr, f, reset=(f == 0 or f == 4))
            if rects and f in rects:  # dynamic resolution: another rect = other grid sizes in the same graph shape
                cs.rectSize[0], cs.rectSize[1] = rects[f]
                cs.rectSizePrev[0], cs.rectSizePrev[1] = rects.get(f - 1, (w, h))
            elif rects and (f - 1) in rects:
                cs.rectSizePrev[0], cs.rectSizePrev[1] = rects[f - 1]
            pa, pb = ha.upload(fr), hb.upload(fr)
            order = [[d] for d in dd] if split_calls else [[d for d in dd]]  # the sample: one Denoise call per denoiser (NRDSample.cpp:4082, :4126, :4224)
            ha.frame(cs, pa, st, order=order)
            hb.frame(cs, pb, st, order=order)
            if not sync_every_frame and f + 1 < frames:
                continue  # back-to-back submission: nothing waits for the device between frames
            if stream is not None:
                stream.synchronize()
            for key in ha.outputs:
                assert np.array_equal(ha.fetch(ha.outputs[key]), hb.fetch(hb.outputs[key])), (f, key)
    for pool in (0, 1):
        for x, y in zip(ha.nrd.pools[pool], hb.nrd.pools[pool]):
            assert np.array_equal(ha.fetch(x["buf"]), hb.fetch(y["buf"])), x["name"]
    return hb.nrd.graph_stats()


@pytest.mark.gpu
@pytest.mark.parametrize("dens", [["REBLUR_DIFFUSE_SPECULAR", "SIGMA_SHADOW"], ["RELAX_DIFFUSE_SPECULAR"]])
def test_graph_replay_bit_identical(pkg, api, hip, dens):
    import torch

    side = torch.cuda.Stream()
    stats = run_pair(pkg, api, hip, dens, frames=14, stream=side, rects={6: (288, 160)})
    # 14 frames through the graph. Every identifier list keeps a ring of 3 executable graphs (one is only patched after the launch that
    # used it last has finished): instantiated for frames 0 (with the restart's clears), 1, 2 (the other two ring entries), 3 (ring entry
    # 0 had the clears), 4 (clears again), 7 (the entry frame 4 left with clears) - every other frame only patches kernel arguments
    assert stats["replayed"] == 14 and stats["direct"] == 0, stats
    assert stats["instantiated"] <= 6, stats


@pytest.mark.gpu
def test_graph_replay_back_to_back_without_sync(pkg, api, hip):
    """frames submitted back to back with NO synchronisation in between - frame N's kernels may still be queued when frame
    N + 1 patches an executable graph. The ring of executable graphs + the event behind each launch must keep every frame on its own
    constants: final outputs and every pool plane equal the pass-by-pass instance after 12 frames (any frame run with a neighbour's
    matrices / frame index / ping-pong pointers would leave its mark in the histories)"""
    import torch

    side = torch.cuda.Stream()
    stats = run_pair(pkg, api, hip, ["REBLUR_DIFFUSE_SPECULAR", "SIGMA_SHADOW"], frames=12, w=960, h=544, stream=side, sync_every_frame=False)
    assert stats["replayed"] == 12 and stats["direct"] == 0, stats


@pytest.mark.gpu
def test_graph_replay_one_graph_per_identifier_list(pkg, api, hip):
    """the sample's call pattern - one Denoise per denoiser, three a frame: every identifier list keeps its own executable graph, so a
    steady frame patches arguments only (an instance with ONE graph would re-instantiate it at every call)"""
    import torch

    side = torch.cuda.Stream()
    stats = run_pair(pkg, api, hip, ["REBLUR_DIFFUSE_SPECULAR", "SIGMA_SHADOW", "REFERENCE"], frames=14, stream=side, split_calls=True)
    assert stats["replayed"] == 14 * 3 and stats["direct"] == 0, stats
    assert stats["instantiated"] <= 3 * 6, stats  # per list: frames 0-4 and 7 (see test_graph_replay_bit_identical), the other 8 patch arguments


@pytest.mark.gpu
def test_graph_flag_on_the_default_stream_launches_directly(pkg, api, hip):
    stats = run_pair(pkg, api, hip, ["REBLUR_DIFFUSE"], frames=3)
    assert stats["replayed"] == 0 and stats["direct"] == 3, stats


def test_graph_flag_in_the_emulated_build_launches_directly(pkg, api, emulated):
    """the host-emulated build has no graphs: capture reports 'unsupported' and the same host code path launches pass by pass"""
    stats = run_pair(pkg, api, emulated, ["REBLUR_DIFFUSE"], frames=2, w=96, h=64)
    assert stats == dict(replayed=0, instantiated=0, direct=2)
