"""Probe: can this torch/RCCL build capture all_to_all_single (+ side-stream wait) in a HIP graph?
World size 1 only (the dev box has one GPU); run under torch.distributed.run."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
n = 4096
src = torch.arange(n, dtype=torch.float64, device="cuda")
dst = torch.zeros(n, dtype=torch.float64, device="cuda")
acc = torch.zeros(n, dtype=torch.float64, device="cuda")
def step():
    w = dist.all_to_all_single(dst, src, [n], [n], async_op=True)
    acc.add_(1.0)          # "interior" work overlapping the exchange
    w.wait()
    acc.add_(dst)          # "boundary" work after the exchange
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print("eager us/step", (time.perf_counter() - t0) / 200 * 1e6)
acc.zero_()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    ok = bool((acc == src + 1.0).all())
    t0 = time.perf_counter()
    for _ in range(200): g.replay()
    torch.cuda.synchronize()
    print("graph capture OK, result ok =", ok, "graph us/step", (time.perf_counter() - t0) / 200 * 1e6)
except Exception as e:
    print("graph capture FAILED:", type(e).__name__, str(e)[:300])
dist.destroy_process_group()
