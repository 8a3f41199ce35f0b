"""CPU: tools/convert_weights.py on archives made here with the
reference's module / parameter names (sp_extractor.cpp:46-62): a TorchScript archive (what
torch::load reads, :355) and a plain state_dict checkpoint."""
import importlib.util
import os

import numpy as np
import pytest

from sp_orb_slam_amd import weights

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("convert_weights", os.path.join(ROOT, "tools", "convert_weights.py"))
cw = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cw)


class _Front(torch.nn.Module):
    def __init__(self):
        super().__init__()
        for name, ci, co, k in weights.LAYERS:
            setattr(self, name, torch.nn.Conv2d(ci, co, k, stride=1, padding=k // 2))

    def forward(self, x):
        return self.conv1a(x)


def _expected(m):
    return weights.from_named_tensors({k: v.detach().numpy() for k, v in m.state_dict().items()})


def test_torchscript_archive_and_state_dict(tmp_path):
    torch.manual_seed(3)
    m = _Front()
    want = _expected(m)
    p1 = str(tmp_path / "superpoint.pt")
    torch.jit.script(m).save(p1)
    out1 = str(tmp_path / "a.spfw")
    assert np.array_equal(cw.convert(p1, out1), want)
    assert np.array_equal(weights.load(out1), want)
    p2 = str(tmp_path / "superpoint_v1.pth")
    torch.save({"module." + k: v for k, v in m.state_dict().items()}, p2)
    assert np.array_equal(cw.convert(p2, str(tmp_path / "b.spfw")), want)
    # layout: OIHW flattened, layer after layer, weight then bias (what pack_layer reads)
    sl = weights.layer_slices()
    assert np.array_equal(want[sl["convPb"][0]].reshape(65, 256, 1, 1), m.convPb.weight.detach().numpy())
    assert np.array_equal(want[sl["conv1a"][2]], m.conv1a.bias.detach().numpy())


def test_missing_tensor_is_reported(tmp_path):
    m = _Front()
    sd = m.state_dict()
    del sd["convDb.bias"]
    p = str(tmp_path / "broken.pth")
    torch.save(sd, p)
    with pytest.raises(SystemExit, match="convDb.bias"):
        cw.convert(p, str(tmp_path / "c.spfw"))


# ---- archives written by the C++ front end (torch::save of a torch::nn::Module — what torch::load at sp_extractor.cpp:355 reads)
_L = [("conv1a", 1, 64, 3), ("conv1b", 64, 64, 3), ("conv2a", 64, 64, 3), ("conv2b", 64, 64, 3), ("conv3a", 64, 128, 3),
      ("conv3b", 128, 128, 3), ("conv4a", 128, 128, 3), ("conv4b", 128, 128, 3), ("convPa", 128, 256, 3), ("convPb", 256, 65, 1),
      ("convDa", 128, 256, 3), ("convDb", 256, 256, 1)]


def _cpp_expected(div):
    """The values tools/cpp_archive/make_cpp_archive.cpp writes (integer hash -> float), recomputed in numpy."""
    out = {}
    for l, (name, ci, co, k) in enumerate(_L):
        ci, co = (1 if l == 0 else max(1, ci // div)), max(1, co // div)
        i = np.arange(co * ci * k * k, dtype=np.uint64)
        h = ((i * np.uint64(2654435761) + np.uint64(l * 0x01000193)) & np.uint64(0xFFFFFFFF)) >> np.uint64(8) & np.uint64(0xFFFF)
        out[name + ".weight"] = (h.astype(np.float32) / np.float32(65536.0) - np.float32(0.5)).reshape(co, ci, k, k)
        j = np.arange(co, dtype=np.uint64)
        hb = ((j * np.uint64(40503) + np.uint64(l * 13 + 7)) & np.uint64(0xFFFFFFFF)) >> np.uint64(4) & np.uint64(0xFFF)
        out[name + ".bias"] = hb.astype(np.float32) / np.float32(4096.0) - np.float32(0.5)
    return out


def test_cpp_frontend_archive_fixture_is_read_by_name():
    """tests/golden/cpp_frontend_archive_div8.pt: written by a C++ program of this repo (tools/cpp_archive/) with
    torch::save on a module whose children carry the reference's register_module names (sp_extractor.cpp:46-62), channel
    counts / 8.  The converter's reader finds every tensor under `<layer>.weight` / `<layer>.bias` with OIHW shapes and the
    exact values."""
    named = cw.read_named(os.path.join(ROOT, "tests", "golden", "cpp_frontend_archive_div8.pt"))
    want = _cpp_expected(8)
    assert set(want) <= set(named)
    for k, v in want.items():
        assert named[k].shape == v.shape and named[k].dtype == np.float32, k
        assert np.array_equal(named[k], v), k


def test_cpp_frontend_archive_full_size_round_trip(tmp_path):
    """Full channel counts: compile the writer against the pip wheel's libtorch (dev container; skipped where g++ or the
    libtorch headers are missing), torch::save, convert, load the .spfw: every parameter in the blob's layout."""
    import shutil
    import subprocess
    exe = os.path.join(ROOT, "tools", "cpp_archive", "bin", "make_cpp_archive")
    if not os.path.exists(exe):
        inc = os.path.join(os.path.dirname(torch.__file__), "include", "torch", "csrc", "api", "include", "torch", "torch.h")
        if shutil.which("g++") is None or not os.path.exists(inc):
            pytest.skip("no g++ / libtorch headers")
        r = subprocess.run(["bash", os.path.join(ROOT, "tools", "cpp_archive", "build.sh")], capture_output=True, text=True)
        if r.returncode != 0:
            pytest.skip("libtorch writer does not build here: " + r.stderr[-300:])
    src = str(tmp_path / "superpoint_cpp.pt")
    subprocess.run([exe, src, "1"], check=True)
    dst = str(tmp_path / "superpoint_cpp.spfw")
    blob = cw.convert(src, dst)
    want = weights.from_named_tensors(_cpp_expected(1))
    assert blob.size == weights.NUM_PARAMS and np.array_equal(blob, want)
    assert np.array_equal(weights.load(dst), want)
