"""bench.py's contract line: ONE stdout line, valid JSON, <= 1 800 characters, carrying every key the driver
reads (round 5's 22.8 KB line came back as `parsed: null`).  The reference's protocol is one
img/s figure (tools/analysis_tools/benchmark.py:99-130); everything beyond the headline goes to the detail file."""
import io
import json
import os
import sys

import pytest

import bench

NEED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "rccl_ranks", "per_rank_ms_per_step",
        "detail_file")
ROOF = ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_us", "traffic", "alg_bytes_per_launch")
CPU = ("value", "unit", "cores", "cpu_model", "kind")


def stub(blow_up=1):
    """A record shaped like the real one, with every free-text field and every detail table `blow_up` times larger."""
    long = "x" * (400 * blow_up)
    ops = {f"op_{i}": {"us_per_call": 1.0 * i, "alg_bytes": 10 ** 8, "note": long} for i in range(40 * blow_up)}
    return {
        "n_gpus": 8, "steps": 20, "warmup": 5, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "metric": "img/s, R3Det R50-FPN 1024x1024 inference (r3det_r50_fpn_1x v1)", "value": 1234.56, "unit": "img/s",
        "ms_per_step": 25.287, "per_rank_ms_per_step": [25.287] * 8, "backend": "nccl", "rccl_ranks": 8,
        "config": {"workload": "BASELINE configs[2] " + long, "workload_short": "BASELINE configs[2]: short",
                   "batch_per_gpu": 4, "global_batch": 32, "nms_type": "v1", "parallelism": long},
        "roofline": {"bound": "hbm", "kernel": "fr_forward_nhwc_wide<true> = the FeatureRefineModule tail " + long,
                     "measured_on": long, "achieved": 5188.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.6485,
                     "avg_launch_us": 51.99, "launches_timed": 20, "traffic": 309742387, "traffic_source": long,
                     "alg_bytes_per_launch": 269746176, "by_field": {"a": {"note": long}}},
        "cpu_baseline": {"value": 1.543, "unit": "img/s", "cores": 256,
                         "cpu_model": "AMD EPYC 9575F 64-Core Processor " + long, "kind": "port", "sample": long,
                         "compare_with": long, "per_op": ops},
        "hot_path": {"what": long, "sync_free": {"what": long}}, "ops": ops, "by_pool": {"what": long},
        "train": {"workload": long}, "rretinanet": {"workload": long}, "kept_per_image": [2000] * 4,
    }


@pytest.mark.parametrize("blow_up", [1, 8])
def test_headline_is_one_short_json_line_with_every_key(blow_up):
    text = bench.headline(stub(blow_up))
    assert "\n" not in text
    assert len(text) < 1800, len(text)
    got = json.loads(text)
    for k in NEED:
        assert k in got, k
    for k in ROOF:
        assert k in got["roofline"], k
    for k in CPU:
        assert k in got["cpu_baseline"], k
    assert got["roofline"]["kernel"] == "fr_forward_nhwc_wide<true>"        # the symbol only
    assert got["config"]["workload"] == "BASELINE configs[2]: short"
    assert set(got["config"]) >= {"workload", "batch_per_gpu", "nms_type"}
    assert got["value"] == 1234.56 and got["ms_per_step"] == 25.287 and got["n_gpus"] == 8
    assert got["roofline"]["frac"] == 0.6485 and got["cpu_baseline"]["kind"] == "port"
    for k in ("ops", "hot_path", "by_pool", "train", "rretinanet"):       # detail only
        assert k not in got


def test_headline_without_roofline_or_cpu_baseline_rows():
    """--mode train / rretinanet and ranks > 1 carry no roofline / cpu_baseline: the line is still valid."""
    rec = stub()
    del rec["roofline"], rec["cpu_baseline"]
    got = json.loads(bench.headline(rec))
    assert "roofline" not in got and "cpu_baseline" not in got and got["metric"].startswith("img/s")


def test_emit_prints_the_headline_last_and_writes_the_detail(tmp_path, monkeypatch, capsys):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    rec = stub(2)
    bench.emit(rec)
    cap = capsys.readouterr()
    lines = cap.out.splitlines()
    assert len(lines) == 1 and len(lines[0]) < 1800
    assert json.loads(lines[0])["detail_file"] == bench.DETAIL_FILE
    assert len(cap.out) + len(cap.err) < 2000          # headline + stderr fit the driver's 2 000-character tail
    for d in (tmp_path, tmp_path / "gpurun_out"):
        full = json.load(open(d / bench.DETAIL_FILE))
        assert full["ops"] == rec["ops"] and full["hot_path"] == rec["hot_path"] and "phases_s" in full


def test_the_committed_round_5_record_fits():
    """The 22.8 KB record the driver could not parse, through the line builder."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_bench.json")
    rec = json.load(open(path))
    assert len(json.dumps(rec)) > 20000
    text = bench.headline(rec)
    assert len(text) < 1800
    got = json.loads(text)
    assert got["value"] == rec["value"] and got["roofline"]["frac"] == rec["roofline"]["frac"]
    assert got["cpu_baseline"]["value"] == rec["cpu_baseline"]["value"]


def test_headline_sheds_text_instead_of_failing():
    """A record whose short fields are themselves oversized still yields one valid line under the limit, numbers intact."""
    rec = stub()
    rec["config"]["workload_short"] = "w" * 3000
    rec["metric"] = "m" * 3000
    rec["cpu_baseline"]["cpu_model"] = "c" * 500
    text = bench.headline(rec)
    assert len(text) <= 1800 and "\n" not in text
    got = json.loads(text)
    assert got["value"] == rec["value"] and got["roofline"]["frac"] == rec["roofline"]["frac"]
    assert got["cpu_baseline"]["value"] == rec["cpu_baseline"]["value"] and got["ms_per_step"] == rec["ms_per_step"]


def test_a_failing_side_section_does_not_cost_the_line(tmp_path, monkeypatch, capsys):
    """op rates, the bounded train / rretinanet entries and the CPU baseline run inside `section`: an exception there is
    recorded in the detail record and the headline still goes out with what was measured."""
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "_ERRORS", {})
    rec = stub()
    del rec["cpu_baseline"]
    with bench.section("cpu baseline"):
        raise RuntimeError("oracle library missing")
    with bench.section("fine"):
        rec["rretinanet"] = {"img_s": 1.0}
    bench.emit(rec)
    cap = capsys.readouterr()
    got = json.loads(cap.out.splitlines()[-1])
    assert got["value"] == rec["value"] and "roofline" in got and "cpu_baseline" not in got
    full = json.load(open(tmp_path / bench.DETAIL_FILE))
    assert list(full["errors"]) == ["cpu baseline"] and "oracle library missing" in full["errors"]["cpu baseline"]
    with pytest.raises(KeyboardInterrupt):          # only Exceptions are swallowed
        with bench.section("interrupted"):
            raise KeyboardInterrupt
