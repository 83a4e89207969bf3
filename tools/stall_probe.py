"""Where does the one ~50-85 ms step of bench.py's hot-path loop come from?
Per step of the same loop: wall time, GC events (gc.callbacks, with duration), device allocations of the
caching allocator (hipMalloc calls), and -- under `--after-model` -- the same with the detector built,
run and deleted first, as bench.py does.

    python tools/stall_probe.py [--after-model] [--steps 60]
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--after-model", action="store_true")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--no-gc", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if args.after_model:
        torch.backends.cudnn.benchmark = True
        model, img = bench.build_model(dev, 100)
        for _ in range(8):
            bench.model_step(model, img)
        torch.cuda.synchronize()
        del model, img
        gc.collect()
        torch.cuda.empty_cache()
    events = []
    t_gc = [0.0]

    def cb(phase, info):
        if phase == "start":
            t_gc[0] = time.perf_counter()
        else:
            events.append(("gc", info["generation"], round((time.perf_counter() - t_gc[0]) * 1e3, 3)))
    gc.callbacks.append(cb)
    if args.no_gc:
        gc.disable()
    wl = bench.build_hot_workload(dev, 7)
    rows = []
    for i in range(args.steps):
        s0 = torch.cuda.memory_stats(dev)
        n_ev = len(events)
        torch.cuda.synchronize()
        t = time.perf_counter()
        bench.hot_path_step(wl)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) * 1e3
        s1 = torch.cuda.memory_stats(dev)
        rows.append(dict(step=i, ms=round(dt, 3),
                         device_allocs=s1["num_device_alloc"] - s0["num_device_alloc"],
                         device_frees=s1["num_device_free"] - s0["num_device_free"],
                         reserved_mb=round(s1["reserved_bytes.all.current"] / 2 ** 20, 1),
                         gc=events[n_ev:]))
    slow = [r for r in rows if r["ms"] > 5 * sorted(x["ms"] for x in rows)[len(rows) // 2]]
    print(json.dumps(dict(median_ms=sorted(x["ms"] for x in rows)[len(rows) // 2],
                          mean_ms=round(sum(x["ms"] for x in rows) / len(rows), 3), slow=slow,
                          first=rows[:4], gc_events=[r["gc"] for r in rows if r["gc"]][:10])))


if __name__ == "__main__":
    main()
