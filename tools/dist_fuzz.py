"""Dev tool: the multi-process push step (tests/dist_gpu_worker.py: powers, 50 unsynchronised repetitions, 36 skewed steps over three x vectors, a value
update, dot, orthogonalize — every rank's slice bitwise) over RANDOM partition shapes: world size, matrix family, rows, band width, all ranks on cuda:0.
   python tools/dist_fuzz.py <first seed> <last seed>      (one torchrun per seed, sequential; prints one line per seed and a summary)"""
import os, random, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
first, last = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(first, last + 1):
    rnd = random.Random(seed)
    world = rnd.choice([2, 2, 3, 3, 4])
    kind = rnd.choice(["s15", "svar", "sfe", "sfe", "s15_up"])
    w = rnd.choice([40, 300, 1500, 2000, 6000])
    n = rnd.randrange(world * max(4 * w, 6000), 260_000)
    if kind == "sfe":
        n -= n % 4
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MI355_DIST_FORCE_SELFCHECK="1", MI355_TEST_EXCHANGE="push",
               MI355_PUSH_SPIN_LOG2="23")
    if rnd.random() < 0.3:
        env["MI355_PUSH_EXT_SPLIT"] = "0"   # the one-launch form of the blocked step although the ranks share the card (small matrices: it fits)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + seed % 200), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), kind, str(n), str(w)]
    t0 = time.time()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
        ok = r.returncode == 0 and any(ln.startswith("DIST_GPU_RESULT ok=1") for ln in r.stdout.splitlines())
        tail = "" if ok else (r.stdout[-600:] + r.stderr[-1500:])
    except subprocess.TimeoutExpired:
        ok, tail = False, "TIMEOUT"
    print(f"seed {seed}: world={world} kind={kind} n={n} w={w} split={env.get('MI355_PUSH_EXT_SPLIT', 'auto')}: {'ok' if ok else 'FAILED'} ({time.time() - t0:.1f} s)", flush=True)
    if not ok:
        bad.append(seed)
        print(tail, flush=True)
        if "TIMEOUT" in tail:
            break  # a hung run: no further GPU work behind it
print(f"DIST_FUZZ seeds {first}-{last}: {len(bad)} failed {bad}")
sys.exit(1 if bad else 0)
