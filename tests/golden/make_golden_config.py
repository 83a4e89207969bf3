"""The ``model = dict(...)`` of the reference's shipped configs as DATA (tests/golden/configs.json): the keyword
trees that mmdet's ``build_detector`` receives for

  configs/r3det/r3det_r50_fpn_1x_dota_v1.py:6-104
  configs/r3det/r3det_tiny_r50_fpn_1x_dota_v1.py
  configs/rretinanet/rretinanet_obb_r50_fpn_1x_dota_v1.py:5-62

read by executing the config files where they lie (plain Python assignments; ``_base_`` lists are not followed:
datasets / schedules / runtime are out of scope).  tests/test_config_build.py builds this package's detectors from
them and compares the module tree with the default constructors.

    python tests/golden/make_golden_config.py
"""
import json
import os

REF = os.environ.get("R3DET_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
CONFIGS = ["configs/r3det/r3det_r50_fpn_1x_dota_v1.py", "configs/r3det/r3det_tiny_r50_fpn_1x_dota_v1.py",
           "configs/rretinanet/rretinanet_obb_r50_fpn_1x_dota_v1.py", "configs/rretinanet/rretinanet_obb_r50_fpn_1x_dota_v2.py",
           "configs/rretinanet/rretinanet_obb_r50_fpn_1x_dota_v3.py", "configs/rretinanet/rretinanet_hbb_r50_fpn_1x_dota_v1.py"]


def merge(base, over):
    """mmcv.Config's _base_ rule for dicts: keys of ``over`` replace / recursively refine those of ``base``
    (lists are replaced)."""
    out = dict(base)
    for k, v in over.items():
        out[k] = merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
    return out


def load_model(rel):
    ns = {}
    with open(os.path.join(REF, rel)) as f:
        exec(compile(f.read(), rel, "exec"), ns)  # noqa: S102  (the reference's own config, build container only)
    model = ns.get("model", {})
    for b in ns.get("_base_", []):
        if "/_base_/" not in b:  # a sibling config: its model dict is the base (datasets / schedules are not followed)
            model = merge(load_model(os.path.normpath(os.path.join(os.path.dirname(rel), b))), model)
    return model


def main():
    out = {}
    for rel in CONFIGS:
        ns = {"model": load_model(rel)}
        out[rel] = ns["model"]
        print(rel, ns["model"]["type"], sorted(ns["model"].keys()))
    with open(os.path.join(OUT, "configs.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
