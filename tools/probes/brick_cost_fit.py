"""Fit the cost model of the brick kernel's planned walk (BrickCost, csrc/avs_internal.hpp) to per-tile phase stamps.

  AVS_BRICK_DEBUG=80 AVS_BRICK_STAMP_FILE=stamps.bin python tools/probes/spmv_time.py 512      (probe library, on the GPU)
  python tools/probes/brick_cost_fit.py stamps.bin [more.bin ...]

A tile's time = the distance between its first stamp and the next tile's first stamp in the same workgroup (10-ns wall clock).  Least
squares over tile + row + run + word + E-tile + quad terms; prints the coefficients in the order AVS_BRICK_COST takes them, the residual
spread, and what list scheduling on the fitted costs would make of the measured tile times (makespan / mean) next to the strided walk."""
import sys
import numpy as np


def load(path):
    raw = np.fromfile(path, dtype=np.int64)
    hdr = raw[:2].view(np.int32)
    wgs, tiles = int(hdr[0]), int(hdr[1])
    st = raw[2:2 + wgs * tiles * 8].reshape(wgs, tiles, 8)
    return st


def samples(st):
    t0 = st[:, :, 0]
    ok = (t0[:, :-1] > 0) & (t0[:, 1:] > 0)
    dur = (t0[:, 1:] - t0[:, :-1])[ok] / 100.0              # microseconds
    f6, f7, kind = st[:, :-1, 6][ok], st[:, :-1, 7][ok], st[:, :-1, 5][ok]
    nprow, nruns, npq, nrows = f6 & 0xffff, (f6 >> 16) & 0xffff, (f6 >> 32) & 0xffff, (f6 >> 48) & 0xffff
    nsw = f7 & 0xffffffff
    X = np.stack([np.ones_like(dur), nprow, nruns, nsw, (kind == 1).astype(float), npq], 1).astype(float)
    return X, dur, kind


if __name__ == "__main__":
    Xs, ys, ks = zip(*[samples(load(p)) for p in sys.argv[1:]])
    X, y, kind = np.concatenate(Xs), np.concatenate(ys), np.concatenate(ks)
    keep = y < np.percentile(y, 99.5)                        # (a preempted workgroup is not a tile cost)
    coef, *_ = np.linalg.lstsq(X[keep], y[keep], rcond=None)
    pred = X @ coef
    print("samples", len(y), "mean tile us", y.mean(), "G", y[kind != 1].mean(), "E", y[kind == 1].mean() if (kind == 1).any() else None)
    print("AVS_BRICK_COST=" + ",".join(f"{c:.5g}" for c in coef), "  (tile,row,run,word,etile,quad)")
    print("residual: rms %.3f us, relative to the mean tile %.3f" % (np.sqrt(np.mean((y - pred)[keep] ** 2)), np.sqrt(np.mean((y - pred)[keep] ** 2)) / y.mean()))
    for p in sys.argv[1:]:
        st = load(p)
        t0 = st[:, :, 0]
        busy = np.array([(r[r > 0].max() - r[r > 0].min()) / 100.0 for r in t0 if (r > 0).sum() > 1])
        print(p, "workgroup busy time us: min %.1f mean %.1f max %.1f  (max / mean %.3f)" % (busy.min(), busy.mean(), busy.max(), busy.max() / busy.mean()))
