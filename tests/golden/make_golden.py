"""Generates tests/golden/*.npz with the CPU oracle (the reference holds no golden vectors for this path, SURVEY.md 8c:
"parity unpinned"; these fixtures pin the build's own oracle against drift and travel to the GPU box).
Run from the repo root:  python tests/golden/make_golden.py --why "what formula moved"   (the default flavour -> tests/golden/*.npz; a fixture
                         whose CONTENT changes is only overwritten with --why, and gets a line in tests/golden/CHANGELOG.md with its distance
                         from the file it replaces - tests/golden/changelog.py; tests/test_golden.py checks every committed fixture has one)
                         python tests/golden/make_golden.py --frozen   (liboracle_frozen.so -> tests/golden/frozen/*.npz; these are
                         round 3's files, byte for byte: the frozen flavour's arithmetic has not moved since)
Inputs are the synthetic scene of nrd-sample_amd/synth.py (64x48, 4 frames, moving camera); outputs are every OUT_* plane
after every frame."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import changelog  # noqa: E402

CASES = {
    "reblur_ds_sigma_reference": ["REBLUR_DIFFUSE_SPECULAR", "SIGMA_SHADOW_TRANSLUCENCY", "REFERENCE"],
    "reblur_diffuse": ["REBLUR_DIFFUSE"],
    "reblur_specular_sigma_shadow": ["REBLUR_SPECULAR", "SIGMA_SHADOW"],
    "relax_ds": ["RELAX_DIFFUSE_SPECULAR"],
}
W, H, FRAMES = 64, 48, 4
INPUT_KEYS = ["viewz", "mv", "normal_roughness", "diff", "spec", "penumbra", "translucency", "confidence", "signal",
              "world_to_view", "world_to_view_prev", "view_to_clip"]


def settings_for(api, scene, dens):
    D = api.Denoiser
    s = {}
    for d in dens:
        if d.name.startswith("REBLUR"):
            s[d] = api.ReblurSettings(minMaterialForDiffuse=0, minMaterialForSpecular=1)  # the sample's values (NRDSample.cpp:566-567)
        elif d.name.startswith("RELAX"):
            s[d] = api.RelaxSettings(minMaterialForDiffuse=0, minMaterialForSpecular=1)
        elif d.name.startswith("SIGMA"):
            s[d] = api.SigmaSettings(lightDirection=list(scene.sun))
        else:
            s[d] = api.ReferenceSettings()
    return s


def main():
    frozen = "--frozen" in sys.argv
    why = sys.argv[sys.argv.index("--why") + 1] if "--why" in sys.argv else None
    pkg = graft.load_package()
    graft.build_oracle()
    api, synth, harness = pkg.api, pkg.synth, pkg.harness
    orc = graft.oracle_backend("frozen" if frozen else None)
    scene = synth.Scene(W, H, dolly=0.03)
    frames = [scene.frame(f) for f in range(FRAMES)]
    out_dir = os.path.dirname(os.path.abspath(__file__))
    if frozen:
        out_dir = os.path.join(out_dir, "frozen")
        os.makedirs(out_dir, exist_ok=True)
    else:
        np.savez_compressed(os.path.join(out_dir, "inputs_64x48.npz"),
                            **{"f%d_%s" % (f, k): frames[f][k] for f in range(FRAMES) for k in INPUT_KEYS})
    for name, dn in CASES.items():
        dens = [api.Denoiser[x] for x in dn]
        hz = harness.Harness(orc, dens, W, H)
        st = settings_for(api, scene, dens)
        blob = {}
        for f in range(FRAMES):
            cs = scene.common_settings(api, frames[f], f, reset=(f == 0))
            planes = hz.upload(frames[f])
            hz.frame(cs, planes, st)
            for k, v in hz.outputs.items():
                blob["f%d_%s" % (f, k)] = hz.fetch(v).copy()
            blob["f%d_signal" % f] = hz.fetch(planes["signal"]).copy()
        path = os.path.join(out_dir, name + ".npz")
        if not frozen:  # no silent regeneration of the default fixtures: what moved, and how far
            tmp = path + ".new.npz"
            np.savez_compressed(tmp, **blob)
            new = np.load(tmp)
            new_hash = changelog.content_hash(new)
            old = np.load(path) if os.path.exists(path) else None
            old_hash = changelog.content_hash(old) if old is not None else None
            if old_hash == new_hash:
                os.remove(tmp)
                print(name, "unchanged (%s)" % new_hash)
                continue
            if not why:
                os.remove(tmp)
                raise SystemExit("%s would change (%s -> %s): re-run with --why \"what formula moved\" (tests/golden/CHANGELOG.md gets the distance)" % (name, old_hash, new_hash))
            changelog.append(name, new_hash, old_hash, why, changelog.distance(new, old, FRAMES) if old is not None else {})
            os.replace(tmp, path)
            print(name, "written (%s), CHANGELOG.md updated" % new_hash)
            continue
        np.savez_compressed(path, **blob)
        print(name, "written")


if __name__ == "__main__":
    main()
