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
