"""Same solve through two builds of the product library (AVS_LIB_PATH): iteration count and solution bits must be equal when only memory
hints differ."""
import sys, os, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch, hashlib, numpy as np
    from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, scenes
    n = int(sys.argv[2]); dev = torch.device("cuda:0")
    sc = scenes.fat_beam(n, 4, device=dev)
    pp = DevicePrepass(sc.res, sc.dx, sc.levels); pi = pp.run(sc.liquid, sc.solid)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels); pp.apply(s); s.set_scene_fields(sc); pp.close()
    s.assemble()
    out = []
    for tol in (1e-3, 1e-8):
        info = s.solve(tol, 8000)
        x = np.asarray(s.solution())
        out.append({"tol": tol, "iterations": int(info.iterations), "resident": bool(info.resident), "sha": hashlib.sha1(x.tobytes()).hexdigest()[:16], "brick": int(s.matrix_format().brick_tiles)})
    print("RESULT " + json.dumps(out))
    sys.exit(0)
for n in (512,):
    for lib in ("libavs_hip.so", "libavs_hip.so", "libavs_hip_ref.so"):
        env = dict(os.environ, AVS_LIB_PATH=os.path.join(ROOT, "adaptiveviscositysolver_amd", lib), AVS_CG_RESIDENT="0")
        p = subprocess.run([sys.executable, __file__, "child", str(n)], env=env, capture_output=True, text=True)
        print(n, lib, [l for l in p.stdout.splitlines() if l.startswith("RESULT")] or p.stderr[-400:])
