"""GPU: SPFE_FLAG_DESC_BF16 — records and results carry the descriptors as bf16 (round-to-nearest-even of the f32
descriptor of sp_extractor.cpp:512-513), everything else unchanged; the record shrinks accordingly."""
import numpy as np
import pytest

from sp_orb_slam_amd import parallel, synth, weights
from sp_orb_slam_amd.extractor import SPExtractor

pytestmark = pytest.mark.gpu
f32 = np.float32


def _rne_bf16(x):
    u = np.ascontiguousarray(x, f32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint16)


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("H,W,B,nf", [(240, 376, 3, 300), (480, 752, 2, 1000)])
def test_bf16_descriptors_are_the_rounded_f32_ones_and_nothing_else_changes(precision, H, W, B, nf):
    blob = weights.synthetic(7, "dense")
    imgs = [synth.make_image(20 + i, H, W) for i in range(B)]
    ref = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False, precision=precision)
    a = ref.extract_batch(imgs)
    full_bytes = ref.record_bytes()
    ref.close()
    ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False, precision=precision, desc_bf16=True)
    assert ext.record_bytes() < full_bytes and full_bytes - ext.record_bytes() >= (nf + 1) * 256 * 2 - 256
    sync = ext.extract_batch(imgs)
    pipe = ext.collect_batch(ext.submit_batch(imgs))
    one = ext(imgs[0], None)
    ext.close()
    for got in (sync, pipe):
        for x, y in zip(a, got):
            assert x.K == y.K and np.array_equal(x.kp_xy, y.kp_xy) and y.status == 0
            assert y.descriptors_bf16 is not None and y.descriptors_bf16.shape == (x.K, 256)
            assert np.array_equal(y.descriptors_bf16, _rne_bf16(x.descriptors))
            assert np.array_equal(y.descriptors.view(np.uint32), y.descriptors_bf16.astype(np.uint32) << 16)
            for name in ("response", "cov2", "cov2_inv", "dense_dust", "semi_dust"):
                assert np.array_equal(getattr(x, name).view(np.uint32), getattr(y, name).view(np.uint32)), name
            assert np.array_equal(x.occ_grid, y.occ_grid)
    assert np.array_equal(one[1].view(np.uint32), sync[0].descriptors.view(np.uint32))   # operator(): the widened rows


def test_device_records_with_bf16_descriptors_decode_and_match():
    import torch
    H, W, nf, B = 240, 376, 200, 2
    blob = weights.synthetic(7, "dense")
    imgs = np.stack([synth.make_image(33 + i, H, W) for i in range(B)])
    ref = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False)
    a = ref.extract_batch(list(imgs))
    ref.close()
    ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False, desc_bf16=True)
    d_img = torch.from_numpy(imgs).cuda()
    d_rec = torch.zeros(B * ext.record_bytes(), dtype=torch.uint8, device="cuda")
    ext.extract_batch_device(d_img.data_ptr(), B, d_rec.data_ptr())
    torch.cuda.synchronize()
    host = d_rec.cpu().numpy()
    lay = parallel.RecordLayout(H, W, nf, desc_bf16=True)
    assert lay.bytes == ext.record_bytes()
    for i in range(B):
        rec = host[i * lay.bytes:(i + 1) * lay.bytes]
        d = lay.unpack(rec)                       # the numpy codec
        v = ext.view_record(rec)                  # the library's view
        assert d["K"] == a[i].K == v.K
        want = (_rne_bf16(a[i].descriptors).astype(np.uint32) << 16)
        assert np.array_equal(d["desc"].view(np.uint32), want)
        assert np.array_equal(v.descriptors.view(np.uint32), want)
        assert np.array_equal(lay.pack(dict(d, descriptors=a[i].descriptors, response=d["response"]))[lay.off_desc:lay.off_occ],
                              rec[lay.off_desc:lay.off_occ])
    # the record-reading matcher widens the rows on load: the oracle's brute-force matcher on the widened descriptors
    from oracle import oracle
    d_out = torch.zeros(ext.match_out_bytes(), dtype=torch.uint8, device="cuda")
    rb = ext.record_bytes()
    for cross in (True, False):
        ext.match_records_device(d_rec.data_ptr() + rb, d_rec.data_ptr(), 1, d_out.data_ptr(), cross)
        torch.cuda.synchronize()
        q, t = ext.view_record(host[rb:2 * rb]), ext.view_record(host[:rb])
        idx, dist = ext.decode_match_out(d_out.cpu().numpy())
        ridx, rdist = oracle.match_bruteforce(q.descriptors, t.descriptors, cross)
        assert np.array_equal(idx[:q.K], ridx)
        assert np.array_equal(dist[:q.K].view(np.uint32), rdist.view(np.uint32))
    ext.close()
