"""dot / AXPY / orthogonalize / norm2 on device vectors: us per call and GB/s against their byte models
(dot 16 B/elt, axpy 24, orthogonalize 16 + 24, norm2 8), back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navierstokes_amd import mpk
for n in (1_000_000, 5_000_000, 40_000_000):
    a = torch.rand(n, dtype=torch.float64, device="cuda"); b = torch.rand(n, dtype=torch.float64, device="cuda")
    c = torch.empty(n, dtype=torch.float64, device="cuda")
    ops = {"dot": (lambda: mpk.dot(a, b), 16), "axpy": (lambda: mpk.axpy(1e-9, a, b), 24),
           "orthogonalize": (lambda: mpk.orthogonalize(n, a, b, c, 1e-8), 40), "norm2": (lambda: mpk.norm2(a), 8)}
    for name, (fn, bpe) in ops.items():
        for _ in range(10): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): fn()
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        print(f"BLAS1 n={n:>9} {name:14s} {us:8.1f} us  {bpe * n / us / 1e3:8.1f} GB/s  {bpe * n / us / 1e3 / 80:5.1f} % of 8 TB/s", flush=True)
