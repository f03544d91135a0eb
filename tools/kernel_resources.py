"""Register / LDS / scratch use of the kernels in the shipped library (from the code objects' metadata notes): `python tools/kernel_resources.py [regex]`."""
from __future__ import annotations

import re
import subprocess
import sys
import tempfile

from isa_check import LIB, LLVM, code_objects


def resources(lib: str = LIB):
    out = {}
    for img in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            txt = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", f.name], text=True)
        cur = {}

        def flush():
            if "name" in cur:
                name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip() or cur["name"]
                out[name] = dict(cur)

        for line in txt.splitlines():
            m = re.match(r"\s*[-]?\s*\.(\w+):\s*(.*)", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip()
            if k == "agpr_count":   # first key of a kernel's record (keys are sorted)
                flush()
                cur = {"agpr": v}
            elif k in ("group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "name"):
                cur[k] = v
        flush()
    return out


if __name__ == "__main__":
    pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
    for name, r in sorted(resources(sys.argv[2] if len(sys.argv) > 2 else LIB).items()):
        if pat.search(name):
            print(f"{name[:110]:110s} vgpr {r.get('vgpr_count'):>4} agpr {r.get('agpr'):>3} sgpr {r.get('sgpr_count'):>4} "
                  f"lds {r.get('group_segment_fixed_size'):>6} scratch {r.get('private_segment_fixed_size'):>5} "
                  f"vspill {r.get('vgpr_spill_count'):>3} sspill {r.get('sgpr_spill_count'):>3}")
