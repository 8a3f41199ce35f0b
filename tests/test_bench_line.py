"""CPU: the line bench.py prints.  The driver keeps the last 8 KB of the line it records; in round 5 a 14 KB line lost BASELINE's
own configs[3] leg to that window.  The default line is therefore compact (<= 7500 characters) and ENDS
with a `configs` object carrying BASELINE.json's configurations; this test builds it from canned leg objects of a real run
(tests/golden/bench_line_r05.json: the full line of round 5's committed default run) plus the legs added since, and from an
inflated version of it."""
import copy
import json
import os

import bench

HERE = os.path.dirname(os.path.abspath(__file__))


def _canned():
    with open(os.path.join(HERE, "golden", "bench_line_r05.json")) as f:
        d = json.loads(f.read())
    d["dropin_operator_call_lazy_ms"] = {"p50": 0.93, "p99": 1.01, "calls": 300, "K": 1001, "heat_maps": True, "heat_inv_after_call": False,
                                         "what": "x" * 300}
    fc = copy.deepcopy(d["frontend_chain"])
    fc["final_pose_max_abs_diff_vs_f32"] = 1.5e-3
    d["frontend_chain_bf16"] = fc
    return d


def test_default_line_is_compact_and_ends_with_configs():
    d = _canned()
    assert len(json.dumps(d)) > 13000                       # the full objects: what round 5 printed
    line = bench.build_line(d)
    js = json.dumps(line)
    assert len(js) <= bench.LINE_LIMIT <= 7500, len(js)
    assert list(line)[-1] == "configs"
    # the contract's keys survive, with the roofline and cpu_baseline objects
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] == d["value"] and line["config"]["frames_per_gpu"] == 8
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert "what" not in js and "rule" not in json.dumps(line.get("parity_frame0_detail", {}))
    # BASELINE's configurations, in the driver's tail whatever else the line holds
    tail = js[-8000:]
    c = line["configs"]
    c1 = [k for k in c if k.startswith("configs[1]")][0]
    assert c[c1]["latency_ms_p50"] == d["latency_batch1_ms"]["p50"] and c[c1]["latency_ms_p99"] == d["latency_batch1_ms"]["p99"]
    assert c[c1]["dropin_operator_call_ms_p50"] == d["dropin_operator_call_ms"]["p50"]
    c3 = [k for k in c if "configs[3]" in k][0]
    assert "bf16_1280x720_b8" in c3 and c3 in tail
    leg = d["bf16_1280x720_b8"]
    assert c[c3]["value"] == leg["value"] and c[c3]["ms_per_step"] == leg["ms_per_step"]
    assert c[c3]["kernel_ms"] == leg["roofline"]["kernel_ms"] and c[c3]["frac"] == leg["roofline"]["frac"]
    assert c[c3]["frac_of_sustained"] == leg["roofline"]["frac_of_sustained"]
    assert c[c3]["whole_path_frac_of_peak"] == leg["whole_path_frac_of_peak"] and c[c3]["sclk_avg"] == leg["sclk_mhz"]["avg"]
    assert str(leg["value"]) in tail and str(d["latency_batch1_ms"]["p50"]) in tail
    # ... and in a 2,000-character tail (what BENCH_r05.json kept of the line): configs[1], configs[3] and the headline close the object
    short = js[-2000:]
    assert list(c)[-1] == c1 and list(c)[-2] == c3 and list(c)[-3].startswith("headline")
    assert c1 in short and c3 in short and list(c)[-3] in short
    assert '"latency_ms_p50": %s' % json.dumps(d["latency_batch1_ms"]["p50"]) in short and '"value": %s' % json.dumps(leg["value"]) in short
    assert '"value": %s' % json.dumps(d["value"]) in short
    head = [k for k in c if k.startswith("headline")][0]
    assert c[head]["value"] == d["value"] and c[head]["frac"] == d["roofline"]["frac"]
    for name in ("f32_640x480_b8", "f32_1280x720_b8", "bf16_752x480_b8"):
        assert c[name]["value"] == d[name]["value"]
    c4 = [k for k in c if k.startswith("configs[4]")][0]
    assert c[c4]["frontend_chain_bf16"]["final_pose_max_abs_diff_vs_f32"] == 1.5e-3
    assert c[c4]["frontend_chain"]["consistent_with_camera_motion"] == d["frontend_chain"]["associations_consistent_with_camera_motion"]


def test_an_inflated_line_is_trimmed_not_truncated():
    """More / longer exploratory legs than today: the least important objects give way (named in `line_trimmed`), never `configs`."""
    d = _canned()
    for i in range(6):
        d["extra_leg_%d" % i] = copy.deepcopy(d["dust_alignment"])
    d["dust_alignment"]["more"] = {"k%d" % i: i * 1.234567 for i in range(60)}
    line = bench.build_line(d)
    js = json.dumps(line)
    assert list(line)[-1] == "configs" and list(line)[-2] == "line_trimmed" and line["line_trimmed"]
    assert len(js) <= bench.LINE_LIMIT
    assert line["configs"] == bench.build_line(_canned())["configs"]


def test_verbose_line_keeps_everything_and_still_ends_with_configs():
    d = _canned()
    line = bench.build_line(d, verbose=True)
    assert list(line)[-1] == "configs" and "what" in line["bf16_1280x720_b8"]
    assert all(line[k] == v for k, v in d.items())
