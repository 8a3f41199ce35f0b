#!/usr/bin/env python3
"""Offline weight converter: the reference's `superpoint.pt` -> `.spfw`.

The reference loads its weights with `torch::load(model_, common::model_path)`
(orb_slam2/src/cv/sp_extractor.cpp:355) into an `SPFrontend` whose sub-modules are registered as
conv1a ... convDb (:46-62): i.e. a TorchScript-style archive whose parameters are named
`<layer>.weight` (OIHW f32) / `<layer>.bias`.  This tool reads such an archive — or a plain
`state_dict` checkpoint with the same names, which is how the original SuperPoint weights are
distributed — and writes the flat blob libspfe loads (`spfe_config.weights_path`).

Dev-time only: PyTorch is needed HERE, never by libspfe or at run time.

    python tools/convert_weights.py superpoint.pt superpoint.spfw
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def read_named(path):
    """-> {name: numpy array} from a TorchScript archive or a state_dict checkpoint."""
    import torch

    try:
        mod = torch.jit.load(path, map_location="cpu")
        named = dict(mod.named_parameters())
        named.update(dict(mod.named_buffers()))
    except RuntimeError:
        obj = torch.load(path, map_location="cpu", weights_only=True)
        named = obj.get("state_dict", obj) if isinstance(obj, dict) else obj.state_dict()
    out = {}
    for k, v in named.items():
        k = k[len("module."):] if k.startswith("module.") else k   # DataParallel checkpoints
        out[k] = v.detach().to(torch.float32).cpu().numpy()
    return out


def convert(src, dst):
    from sp_orb_slam_amd import weights

    named = read_named(src)
    missing = [n + s for n, _, _, _ in weights.LAYERS for s in (".weight", ".bias") if n + s not in named]
    if missing:
        raise SystemExit("%s: missing tensors %s (found: %s)" % (src, missing[:4], sorted(named)[:6]))
    blob = weights.from_named_tensors(named)
    weights.save(dst, blob)
    return blob


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    b = convert(sys.argv[1], sys.argv[2])
    print("wrote %s: %d parameters" % (sys.argv[2], b.size))
