"""Reference point for the HBM roofline: device copy / elementwise add at the FR level-0 size."""
import torch
for n in (4, 8, 16):
    f = torch.randn(n, 256, 128, 128, device="cuda")
    o = torch.empty_like(f)
    for name, fn in (("copy_", lambda: o.copy_(f)), ("add 1", lambda: torch.add(f, 1.0, out=o))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                fn()
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 100)
        ts.sort()
        b = 2 * f.numel() * 4
        print(f"N={n} {name:6s} {ts[3]:8.1f} us  {b / ts[3] / 1e3:8.1f} GB/s")
