"""Host-buffer boundary of the 100 MP frame: upload of the fp32 frame, render, download of the uint8 result (pinned and pageable)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H, W = 8192, 12288
for ch, name in ((3, "HWC3"), (4, "HWC4")):
    for pinned in (True, False):
        host = torch.empty((H, W, ch), dtype=torch.float32, pin_memory=pinned)
        host.uniform_(0.0, 1.0) if not pinned else host.zero_()
        out_host = torch.empty((H, W, 3), dtype=torch.uint8, pin_memory=pinned)
        dev = torch.empty((H, W, ch), dtype=torch.float32, device="cuda")
        out_dev = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
        best_up = best_dn = 1e9
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter(); dev.copy_(host, non_blocking=True); torch.cuda.synchronize()
            best_up = min(best_up, time.perf_counter() - t0)
            t0 = time.perf_counter(); out_host.copy_(out_dev, non_blocking=True); torch.cuda.synchronize()
            best_dn = min(best_dn, time.perf_counter() - t0)
        gb_up, gb_dn = host.numel() * 4 / 1e9, out_host.numel() / 1e9
        print(f"{name} {'pinned  ' if pinned else 'pageable'}: upload {gb_up:.2f} GB in {best_up * 1e3:6.1f} ms ({gb_up / best_up:5.1f} GB/s)   "
              f"uint8 download {gb_dn:.2f} GB in {best_dn * 1e3:5.1f} ms ({gb_dn / best_dn:5.1f} GB/s)")
        del host, out_host, dev, out_dev
