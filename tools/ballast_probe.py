"""Does device memory allocated (and kept) BEFORE the matrix move the create-time placement draws (profiles/NOTES.md §4.12)?
usage: python tools/ballast_probe.py <ballast GB>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navierstokes_amd import mpk, synth
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
n = 5_000_000
p, c, v = synth.rows("s15", n)
torch.cuda.init()
ballast = [torch.empty(int(512 << 20), dtype=torch.uint8, device="cuda") for _ in range(int(gb * 2))]  # 512 MB pieces, kept
A = mpk.csrmatrix(n, p, c, v); _ = A.handle
pi = A.placement_info()
(x, y), us = A.alloc_vectors(2, draws=8)
print(f"BALLAST {gb:4.1f} GB kept in front: value-array draws {pi['values']}  column stream {pi['column_stream']}  vector candidates {us}", flush=True)
