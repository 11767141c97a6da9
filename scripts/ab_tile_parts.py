"""Where the time of the tile product goes (scripts/exp_tile.hip): tiles alone, group rows alone, both; and what a
pure stream reaches on this device with several kernel shapes."""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import maxwell_amd as mx

so = os.path.join(ROOT, "scripts", "bin", "libexp_tile.so")
if "--build" in sys.argv or not os.path.exists(so):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                    "-Wno-unused-value", "-Wno-unused-result", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "maxwell_amd/csrc/device"),
                    os.path.join(ROOT, "scripts/exp_tile.hip"), "-o", so, "-L" + os.path.join(ROOT, "maxwell_amd/lib"), "-lmxg",
                    "-Wl,-Bsymbolic,-rpath," + os.path.join(ROOT, "maxwell_amd/lib")], check=True)
    if "--build" in sys.argv:
        sys.exit(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = mx.Ctx(0)
exp = C.CDLL(so)
ms = C.c_float()
res = {}
alg_r, alg_w = 2_078_543_388, 164_871_048
for variant, wgs in ((3, 8), (3, 4), (3, 16), (4, 0), (5, 0)):
    rc = exp.exp_stream_time(ctx.h, C.c_longlong(alg_r), C.c_longlong(alg_w), variant, wgs, 20, C.byref(ms))
    res["stream_v%d_w%d" % (variant, wgs)] = [rc, round(ms.value, 4), round((alg_r + alg_w) / ms.value / 1e6, 0)]
print(json.dumps(res), flush=True)
sim = mx.Sim.pillbox(n)
A = sim.to_device(ctx, "vecLapl", shift=0.05 * (2 * np.pi) ** 2)
os.environ["MXG_PAT"] = "1"
A.compress()
X = mx.MultiVec(ctx, A.ncols, 1); Y = mx.MultiVec(ctx, A.nrows, 1); X.random(1)
out = {"form": A.product_form()}
for rnd in range(2):
    for name, parts, order in (("tiles", 1, 0), ("groups", 4, 0), ("all_first", 15, 0), ("all_last", 15, 1), ("all_spread", 15, 2)):
        rc = exp.exp_tile_time(ctx.h, A.h, X.h, Y.h, parts, order, 50, C.byref(ms))
        out.setdefault(name, []).append(round(ms.value, 5) if rc == 0 else "rc %d" % rc)
rc = exp.exp_tile_two_streams(ctx.h, A.h, X.h, Y.h, 50, C.byref(ms))
out["two_streams"] = round(ms.value, 5) if rc == 0 else "rc %d" % rc
X8 = mx.MultiVec(ctx, A.ncols, 8); Y8 = mx.MultiVec(ctx, A.nrows, 8); X8.random(1)
for name, parts, order in (("k8_first", 15, 0), ("k8_spread", 15, 2)):
    rc = exp.exp_tile_time(ctx.h, A.h, X8.h, Y8.h, parts, order, 10, C.byref(ms))
    out[name] = round(ms.value, 5) if rc == 0 else "rc %d" % rc
print(json.dumps(out))
