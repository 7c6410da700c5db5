"""Does the 256 MB Infinity Cache (MALL) retain freshly written data? Write a buffer front to back, then read its
first vs its last 128 MB and compare the read rates."""
import torch

dev = "cuda"
for total_mb in (192, 384, 768, 1536):
    n = total_mb * (1 << 20) // 2
    x = torch.empty(n, dtype=torch.bfloat16, device=dev)
    src = torch.randn(1 << 20, device=dev).to(torch.bfloat16)
    part = 128 * (1 << 20) // 2
    out = torch.empty(part, dtype=torch.bfloat16, device=dev)
    res = {}
    for which in ("first", "last", "first", "last"):
        x.view(-1, 1 << 20).copy_(src)  # streaming write, front to back (rows are written in order)
        torch.cuda.synchronize()
        sl = x[:part] if which == "first" else x[-part:]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out.copy_(sl)
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(which, []).append(e0.elapsed_time(e1))
    print(f"buffer {total_mb} MB: copy of first 128 MB {min(res['first']):.3f} ms, of last 128 MB {min(res['last']):.3f} ms "
          f"({2 * 128 / 1024 / min(res['first']) * 1e3:.0f} vs {2 * 128 / 1024 / min(res['last']) * 1e3:.0f} GB/s r+w)", flush=True)
