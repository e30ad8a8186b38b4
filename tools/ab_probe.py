"""A/B of environment switches on the stand-alone extraction time: ab_probe.py TILE 'A=1 B=2' 'A=0' ...  (each variant = a set of env
assignments, '-' = none). Variants run in fresh processes, interleaved, three times; prints the median ms per extraction."""
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import ctypes as C, importlib, sys, time, torch
sys.path.insert(0, %r)
pkg = importlib.import_module("cubesat-apds_amd")
pl = importlib.import_module("cubesat-apds_amd.pipeline")
L = pkg._lib.lib(); check = pkg._lib.check
dev = torch.device("cuda:0")
T = int(sys.argv[1]); cap = (1 << 18) - 1
kps = torch.empty((cap, 7), dtype=torch.float32, device=dev); desc = torch.empty((cap, 64), dtype=torch.uint8, device=dev)
frames = [torch.from_numpy(pkg.synth.make_tile(T, T, frame_index=i)).to(dev) for i in range(2)]
n = C.c_int(0); st = torch.cuda.Stream(dev)
with torch.cuda.stream(st):
    def go(reps):
        for rep in range(reps):
            f = frames[rep %% 2]
            check(L.apds_dev_akaze_extract(f.data_ptr(), T, T, f.shape[2], f.stride(0), cap, kps.data_ptr(), desc.data_ptr(), cap, C.byref(n), pl.torch_stream()))
        torch.cuda.synchronize()
    go(5)
    best = 1e9
    for k in range(5):
        t0 = time.perf_counter(); go(20); best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
print(best)
''' % ROOT

tile = sys.argv[1]
variants = sys.argv[2:]
res = {v: [] for v in variants}
for rnd in range(3):
    for v in variants:
        env = dict(os.environ)
        if v != "-":
            for kv in v.split():
                k, val = kv.split("=")
                env[k] = val
        out = subprocess.run([sys.executable, "-c", CODE, tile], env=env, capture_output=True, text=True)
        try:
            res[v].append(float(out.stdout.strip().splitlines()[-1]))
        except Exception:
            print("FAILED", v, out.stderr[-400:], flush=True)
for v in variants:
    if res[v]:
        print(f"tile {tile}  {v:40s} median {statistics.median(res[v]):.4f} ms  (runs {' '.join('%.4f' % x for x in res[v])})", flush=True)
