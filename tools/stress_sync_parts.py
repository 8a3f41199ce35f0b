#!/usr/bin/env python3
"""Stress (GPU box): the synchronous host calls in every form the drop-in uses them — spfe_extract_batch, the call in parts
(spfe_extract_begin / _maps / _rows / _finish), with and without the caller's own map buffers (spfe_set_map_buffers, swapped
and dropped between calls), the lazy heat_inv form, SPFE_EARLY_HEAT_COPY on and off — interleaved at random on ONE handle, over
random sizes, batch sizes, feature counts, precisions and detectors, against a plain handle's spfe_extract_batch.  Records,
maps, early maps and early rows must be the same bits.  usage: python tools/stress_sync_parts.py [n_cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sp_orb_slam_amd import synth, weights  # noqa: E402
from sp_orb_slam_amd.extractor import SPExtractor  # noqa: E402

FIELDS = ("kp_xy", "response", "descriptors", "cov2", "cov2_inv", "occ_grid", "dense_dust", "semi_dust")


def bits(x):
    return np.ascontiguousarray(x).view(np.uint8)


def same(a, b, where, lazy):
    assert a.status == 0 and a.K == b.K, (where, a.status, a.K, b.K)
    for f in FIELDS:
        assert np.array_equal(bits(getattr(a, f)), bits(getattr(b, f))), (where, f)
    assert np.array_equal(bits(a.heat), bits(b.heat)), (where, "heat")
    if lazy:
        assert a.heat_inv is None, (where, "heat_inv of the lazy form")
    else:
        assert np.array_equal(bits(a.heat_inv), bits(b.heat_inv)), (where, "heat_inv")


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    sizes = [(120, 160), (240, 376), (480, 752), (480, 640), (136, 200), (360, 640)]
    for case in range(n_cases):
        H, W = sizes[int(rng.integers(len(sizes)))]
        B = int(rng.integers(1, 4)); nf = int(rng.choice([50, 300, 1000])); prec = str(rng.choice(["f32", "bf16"]))
        det = str(rng.choice(["dense", "sparse"])); early = str(rng.choice(["1", "1", "0"])); lazy = bool(rng.integers(2))
        ncalls = int(rng.integers(6, 11))
        calls = [(int(rng.integers(1, B + 1)), str(rng.choice(["batch", "parts", "parts", "single"])), int(rng.integers(4))) for _ in range(ncalls)]
        seeds = [int(rng.integers(1 << 16)) for _ in range(5)]
        print("case %d: %s %dx%d B %d nf %d %s early_copy %s lazy %s calls %s" % (case, prec, W, H, B, nf, det, early, lazy, calls), flush=True)
        blob = weights.synthetic(7, det)
        imgs = [synth.make_image(sd, H, W) for sd in seeds]
        os.environ.pop("SPFE_EARLY_HEAT_COPY", None)
        ref_ext = SPExtractor(nf, H, W, blob, max_batch=B, precision=prec)
        os.environ["SPFE_EARLY_HEAT_COPY"] = early
        ext = SPExtractor(nf, H, W, blob, max_batch=B, precision=prec, lazy_heat_inv=lazy)
        backing = [np.full(B * H * W + 5, np.nan, np.float32) for _ in range(4)]
        bufs = [b[5:].reshape(B, H, W) for b in backing]
        cur = (None, None)
        for k, (n, form, aim) in enumerate(calls):
            pick = [imgs[int(rng.integers(len(imgs)))] for _ in range(n)]
            if form == "single":
                pick = pick[:1]
            want = ref_ext.extract_batch(pick)
            # where this call's maps go: the library's buffers / one pair of the caller's / the other pair / heat only
            cur = [(None, None), (bufs[0], bufs[1]), (bufs[2], bufs[3]), (bufs[0], None)][aim]
            ext.set_map_buffers(*cur)
            if form == "parts":
                ext.extract_begin(pick)
                hm, hi = ext.extract_maps()
                rows = [ext.extract_rows(i) for i in range(len(pick))]
                if hm is not None:
                    assert early == "1"
                    for i in range(len(pick)):
                        assert np.array_equal(bits(hm[i]), bits(want[i].heat)), (case, k, i, "early heat")
                        if hi is not None:
                            assert np.array_equal(bits(hi[i]), bits(want[i].heat_inv)), (case, k, i, "early heat_inv")
                    assert (hi is None) == lazy
                    if cur[0] is not None:
                        assert hm.ctypes.data == cur[0].ctypes.data
                for i, r in enumerate(rows):
                    if r is not None:
                        assert early == "1" and r.shape == (want[i].K, 256), (case, k, i, "early rows")
                        assert np.array_equal(bits(r), bits(want[i].descriptors)), (case, k, i, "early rows")
                got = ext.extract_finish()
            elif form == "single":
                ext(pick[0], None)
                got = [ext.last]
            else:
                got = ext.extract_batch(pick)
            for i in range(len(pick)):
                same(got[i], want[i], (case, k, form, i), lazy)
                if cur[0] is not None:
                    assert np.array_equal(bits(cur[0][i]), bits(want[i].heat)), (case, k, i, "caller's heat buffer")
                if cur[1] is not None and not lazy:
                    assert np.array_equal(bits(cur[1][i]), bits(want[i].heat_inv)), (case, k, i, "caller's heat_inv buffer")
            if lazy and rng.integers(2):
                j = int(rng.integers(len(pick)))
                assert np.array_equal(bits(ext.fetch_heat_inv(j)), bits(want[j].heat_inv)), (case, k, j, "fetched heat_inv")
        ext.close()
        ref_ext.close()
        print("case %d ok" % case, flush=True)
    os.environ.pop("SPFE_EARLY_HEAT_COPY", None)
    print("stress_sync_parts: %d cases, all records and maps bit-identical" % n_cases)


if __name__ == "__main__":
    main()
