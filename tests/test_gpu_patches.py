"""GPU: spfe_match_patches / spfe_match_patches_record_device vs oracle_match_patches (exact) —
tracker_dust.cpp:113-172."""
import numpy as np
import pytest

from oracle import oracle
from sp_orb_slam_amd import synth, weights
from sp_orb_slam_amd.extractor import SPExtractor

pytestmark = pytest.mark.gpu


def _unit(rng, n):
    d = rng.standard_normal((n, 256)).astype(np.float32)
    return d / np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)


@pytest.fixture(scope="module")
def frame():
    H, W = 240, 320
    ext = SPExtractor(600, H, W, weights.synthetic(7, "dense"), with_heat=False)
    ext(synth.make_image(77, H, W), None)
    fr = ext.last
    yield ext, fr
    ext.close()


def _map_points(rng, fr, m, noise, contention):
    """Map points that re-observe keypoints of the frame: descriptor = keypoint's + noise, position = the
    keypoint's cell + a sub-cell offset so its own cell is one of the four looked at; `contention` of them
    are duplicates of earlier ones (fighting for the same keypoint)."""
    k = rng.integers(0, fr.K, m)
    k[rng.random(m) < contention] = k[0]
    desc = fr.descriptors[k] + np.float32(noise) * _unit(rng, m)
    cx = fr.kp_xy[k, 0].astype(int) // 8
    cy = fr.kp_xy[k, 1].astype(int) // 8
    uv = np.stack([cx - rng.integers(0, 2, m) + rng.random(m) * 0.999, cy - rng.integers(0, 2, m) + rng.random(m) * 0.999], 1)
    return desc.astype(np.float32), uv.astype(np.float32)


@pytest.mark.parametrize("m,noise,contention", [(1, 0.1, 0.0), (150, 0.2, 0.1), (600, 0.4, 0.3), (2000, 0.6, 0.5),
                                                (4096, 0.3, 0.9)])
def test_patches_match_oracle(frame, m, noise, contention):
    ext, fr = frame
    rng = np.random.default_rng(m)
    desc, uv = _map_points(rng, fr, m, noise, contention)
    uv[rng.random(m) < 0.03] = [-3.0, 500.0]                       # some projections fall outside
    got = ext.match_patches(desc, uv, fr.occ_grid, fr.descriptors)
    ref = oracle.match_patches(desc, uv, fr.occ_grid, fr.descriptors)
    assert np.array_equal(got, ref)
    matched = got[got >= 0]
    assert len(np.unique(matched)) == len(matched)                 # a keypoint is given away once
    if contention == 0.0:
        assert (got >= 0).all()


def test_patches_empty_and_thresholds(frame):
    ext, fr = frame
    rng = np.random.default_rng(5)
    assert len(ext.match_patches(np.zeros((0, 256), np.float32), np.zeros((0, 2), np.float32), fr.occ_grid,
                                 fr.descriptors)) == 0
    desc, uv = _map_points(rng, fr, 300, 0.5, 0.2)
    for md in (0.05, 0.3, 0.75, 5.0):
        assert np.array_equal(ext.match_patches(desc, uv, fr.occ_grid, fr.descriptors, md),
                              oracle.match_patches(desc, uv, fr.occ_grid, fr.descriptors, md))


@pytest.mark.parametrize("desc_bf16", [False, True])
def test_patches_against_resident_record(desc_bf16):
    """(desc_bf16: a record made with SPFE_FLAG_DESC_BF16 — its rows are widened on load; the oracle gets the widened rows)"""
    import torch

    H, W, nf = 240, 320, 600
    ext = SPExtractor(nf, H, W, weights.synthetic(7, "dense"), with_heat=False, desc_bf16=desc_bf16)
    img = torch.from_numpy(synth.make_image(78, H, W)[None]).cuda()
    rec = torch.zeros(ext.record_bytes(), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    ext.extract_batch_device(img.data_ptr(), 1, rec.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    fr = ext.view_record(rec.cpu().numpy())
    rng = np.random.default_rng(6)
    desc, uv = _map_points(rng, fr, 500, 0.3, 0.3)
    d_desc, d_uv = torch.from_numpy(desc).cuda(), torch.from_numpy(uv).cuda()
    d_out = torch.full((500,), -7, dtype=torch.int32, device="cuda")
    ext.match_patches_record_device(d_desc.data_ptr(), d_uv.data_ptr(), 500, rec.data_ptr(), d_out.data_ptr(), 0.75,
                                    s.cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), oracle.match_patches(desc, uv, fr.occ_grid, fr.descriptors))
    ext.close()
