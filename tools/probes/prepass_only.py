"""The device pre-pass alone, N frames (for rocprofv3 --kernel-trace --stats): tools/probes/prepass_only.py N LEVELS SCENE FRAMES"""
import sys
import torch
sys.path.insert(0, ".")
from adaptiveviscositysolver_amd import DevicePrepass, scenes  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 5
scene = sys.argv[3] if len(sys.argv) > 3 else "sheet"
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 6
dev = torch.device("cuda:0")
sc = {"beam": lambda: scenes.fat_beam(n, levels, device=dev), "sheet": lambda: scenes.thin_sheet(n, levels, thickness_cells=32, device=dev)}[scene]()
pp = DevicePrepass(sc.res, sc.dx, sc.levels)
for f in range(frames):
    info = pp.run(sc.liquid, sc.solid)
    print(f, round(info.weights_ms, 3), round(info.octree_ms, 3), round(info.classify_ms, 3), round(info.number_ms, 3), flush=True)
