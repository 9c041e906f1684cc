"""Dev tool (round 4): the product with the dot in its epilogue against product + separate dot, sliced-stream kernel, per variant.
   python tools/dot_ab.py [n]"""
import os, sys, subprocess, json
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
if len(sys.argv) > 2:  # child: one variant, one epilogue setting
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from navierstokes_amd import mpk, synth
    p, c, v = synth.rows("s15", n)
    A = mpk.csrmatrix(n, p, c, v).set_kernel("sstream")
    x = torch.from_numpy(synth.x_sin(0, n)).cuda()
    b = torch.from_numpy(np.cos(0.002 * np.arange(n))).cuda()
    y = torch.empty(n, dtype=torch.float64, device="cuda")

    def timed(fn, reps=100):
        for _ in range(10):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    out = dict(kernel=A.kernel_name(), epilogue=A.dot_in_epilogue(), plain_us=round(timed(lambda: mpk.SpMV_CSR(y, x, A)), 2),
               with_dot_us=round(timed(lambda: mpk.SpMV_CSR_dot(y, x, A, b)), 2))
    print(json.dumps(out))
    sys.exit(0)
for form in "0123":
    for epi in "10":
        env = dict(os.environ, MI355_SSTREAM="1", MI355_SSTREAM_FORM=form, MI355_SPMV_DOT_EPILOGUE=epi)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(n), "child"], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(f"form {form} epilogue={epi}:", line[-1] if line else r.stderr[-300:], flush=True)
