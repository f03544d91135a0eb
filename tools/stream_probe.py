import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveviscositysolver_amd import capi
L = capi.load_probe()
for mode, name in ((0, "read-only 16B/lane"), (1, "read-only non-temporal"), (2, "copy (read+write)")):
    for gb in (1.5, 4.0):
        g = C.c_double()
        capi.check(L.avs_bench_stream(mode, int(gb * 1e9), 20, 0, C.byref(g)))
        print(f"{name:26s} {gb:4.1f} GB : {g.value:8.1f} GB/s")
