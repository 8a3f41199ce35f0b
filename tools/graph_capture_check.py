#!/usr/bin/env python3
"""A synchronous device-resident call inside a hipGraph.  Captures one spfe_extract_batch_device call on a torch stream into a
torch.cuda.CUDAGraph (the library's side streams fork from and join the capturing stream by events), replays it on new frame
contents, compares every field of the records with direct calls (exit code 1 on a difference, 2 when the capture fails) and
times replay against the direct call.  usage: tools/graph_capture_check.py f32|bf16 <frames per call> [H W]
Measured (round 4): replay is bit-identical; it is NOT faster than the direct call on the default paths (752x480 f32 single
frame 0.73 against 0.70 ms: the launch stream already runs its kernels back to back), and 20 % faster where the call crosses
streams by events (each hop ~13 us as stream operations, next to nothing as graph edges)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sp_orb_slam_amd import synth, weights
from sp_orb_slam_amd.extractor import SPExtractor


FIELDS = ("kp_xy", "response", "descriptors", "cov2", "cov2_inv", "occ_grid", "dense_dust", "semi_dust")


def wrap_mode(prec, B, H, W, nf):
    """usage: graph_capture_check.py wrap f32|bf16 <frames per call> [H W]   (run with SPFE_COV_CAPS=,,,,,5)
    ADVICE r5: generation-tagged claim / done maps and a captured call.  A captured call freezes its generation code G and
    every replay leaves G-tagged entries behind; after the codes wrap, a direct call's code is ABOVE G and its atomicMin would
    lose to them (no overlap seen, no keypoint dirty, cov2 silently wrong where regions collide).  With the first code at 5 the
    cycle is 5 4 3 2 1 5 ...: direct 5, 4, 3 - capture at 2 - direct 1 - direct 5 (the wrap) - replay - direct 4 (above G,
    behind a replay) - replay - direct 3, 2 (= G), 1, 5 - replay - direct 4.  Every record against a fresh default handle's."""
    blob = weights.synthetic(7, "dense")
    frames = [torch.from_numpy(synth.make_batch(700 + 10 * k, B, H, W)).cuda() for k in range(2)]
    env = os.environ.pop("SPFE_COV_CAPS", None)
    ref_ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False, precision=prec)
    rb = ref_ext.record_bytes()
    d_rec = torch.zeros(B * rb, dtype=torch.uint8, device="cuda")
    ref = []
    for f in frames:
        ref_ext.extract_batch_device(f.data_ptr(), B, d_rec.data_ptr())
        torch.cuda.synchronize()
        ref.append(d_rec.cpu().numpy().copy())
    ref_ext.close()
    if env is not None:
        os.environ["SPFE_COV_CAPS"] = env
    ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False, precision=prec)
    d_img = frames[0].clone()
    s = torch.cuda.Stream()
    bad = []

    def check(tag, k):
        torch.cuda.synchronize()
        got = d_rec.cpu().numpy()
        for i in range(B):
            a, b = ext.view_record(got[i * rb:(i + 1) * rb]), ext.view_record(ref[k][i * rb:(i + 1) * rb])
            if not (a.K == b.K and a.K > 0 and all(np.array_equal(getattr(a, n), getattr(b, n)) for n in FIELDS)):
                bad.append((tag, i, [n for n in FIELDS if a.K != b.K or not np.array_equal(getattr(a, n), getattr(b, n))]))

    def direct(tag, k):
        d_img.copy_(frames[k])
        d_rec.zero_()
        with torch.cuda.stream(s):
            ext.extract_batch_device(d_img.data_ptr(), B, d_rec.data_ptr(), s.cuda_stream)
        check(tag, k)

    def replay(tag, k):
        d_img.copy_(frames[k])
        d_rec.zero_()
        torch.cuda.synchronize()
        g.replay()
        check(tag, k)

    for n in range(3):
        direct("direct %d" % n, n & 1)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            ext.extract_batch_device(d_img.data_ptr(), B, d_rec.data_ptr(), s.cuda_stream)
    except Exception as e:
        print(prec, B, "capture FAILED:", type(e).__name__, str(e).splitlines()[0][:200], flush=True)
        os._exit(2)
    direct("direct behind the capture", 1)
    direct("direct at the wrap", 0)
    replay("replay 1", 0)
    direct("direct above G behind a replay", 0)
    replay("replay 2", 1)
    for n in range(4):
        direct("direct %d behind replay 2" % n, (n + 1) & 1)
    replay("replay 3", 1)
    direct("direct behind replay 3", 1)
    print(prec, "B", B, "wrap: every call bit-identical to a fresh handle's:", not bad, bad[:6], flush=True)
    ext.close()
    sys.exit(1 if bad else 0)


def main():
    if sys.argv[1] == "wrap":
        a = sys.argv[2:]
        H, W = (int(a[2]), int(a[3])) if len(a) > 3 else (240, 376)
        return wrap_mode(a[0], int(a[1]), H, W, 400)
    H, W, nf = (int(sys.argv[3]), int(sys.argv[4]), 1000) if len(sys.argv) > 4 else (480, 752, 1000)
    blob = weights.synthetic(7, "dense")
    for prec in (sys.argv[1],):
        for B in (int(sys.argv[2]),):
            ext = SPExtractor(nf, H, W, blob, max_batch=B, with_heat=False, precision=prec)
            frames = [torch.from_numpy(synth.make_batch(300 + 10 * k, B, H, W)).cuda() for k in range(3)]
            d_img = frames[0].clone()
            d_rec = torch.zeros(B * ext.record_bytes(), dtype=torch.uint8, device="cuda")
            s = torch.cuda.Stream()
            direct = []
            with torch.cuda.stream(s):
                for f in frames:
                    d_img.copy_(f)
                    ext.extract_batch_device(d_img.data_ptr(), B, d_rec.data_ptr(), s.cuda_stream)
                    s.synchronize()
                    direct.append(d_rec.cpu().numpy().copy())
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g, stream=s):
                    ext.extract_batch_device(d_img.data_ptr(), B, d_rec.data_ptr(), s.cuda_stream)
            except Exception as e:
                print(prec, B, "capture FAILED:", type(e).__name__, str(e).splitlines()[0][:200], flush=True)
                os._exit(2)
            ok = True
            for k, f in enumerate(frames):
                d_img.copy_(f)
                d_rec.zero_()
                torch.cuda.synchronize()
                g.replay()
                torch.cuda.synchronize()
                got = d_rec.cpu().numpy()
                rb = ext.record_bytes()
                for i in range(B):
                    a, b = ext.view_record(got[i * rb:(i + 1) * rb]), ext.view_record(direct[k][i * rb:(i + 1) * rb])
                    same = (a.K == b.K and a.K > 0 and all(np.array_equal(getattr(a, n), getattr(b, n)) for n in
                            ("kp_xy", "response", "descriptors", "cov2", "cov2_inv", "occ_grid", "dense_dust", "semi_dust")))
                    ok = ok and same
            # timing
            def p50(fn, n=300):
                ts = []
                for _ in range(n):
                    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
                return sorted(ts)[n // 2] * 1e3
            with torch.cuda.stream(s):
                t_direct = p50(lambda: ext.extract_batch_device(d_img.data_ptr(), B, d_rec.data_ptr(), s.cuda_stream))
            t_graph = p50(lambda: g.replay())
            print(prec, "B", B, "graph replay bit-identical to direct calls:", ok, " p50 direct %.4f ms, graph %.4f ms" % (t_direct, t_graph), flush=True)
            ext.close()
            if not ok:
                sys.exit(1)


if __name__ == "__main__":
    main()
