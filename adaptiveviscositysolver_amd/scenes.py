"""Synthetic scenes: analytic liquid SDFs, solid walls, velocity fields.

Pure input synthesis for tests and bench.py: every function returns dense fp32 arrays in
the layouts of include/avs.h (x fastest; array shape (nz, ny, nx)).  Written on torch so the
same code produces inputs on the host (tests) and directly in HBM (bench at 512^3).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch


@dataclass
class Scene:
    res: tuple            # (nx, ny, nz), powers of two
    dx: float
    dt: float
    levels: int
    liquid: torch.Tensor  # centre lattice, fp32, shape (nz, ny, nx)
    solid: torch.Tensor | None = None          # None = constant -1 (no solid)
    viscosity: torch.Tensor | float = 1.0      # centre lattice or constant
    density: torch.Tensor | float = 1000.0
    velocity: list = field(default_factory=list)        # 3 face grids
    solid_velocity: list | None = None                  # None = 0
    use_enhanced_gradients: bool = True
    name: str = ""


def _axes(res, device, dtype=torch.float32):
    nx, ny, nz = res
    x = torch.arange(nx, device=device, dtype=dtype)
    y = torch.arange(ny, device=device, dtype=dtype)
    z = torch.arange(nz, device=device, dtype=dtype)
    return x, y, z


def box_sdf(res, dx, center, half, device="cpu"):
    """Exact signed distance to an axis-aligned box, sampled at cell centres."""
    x, y, z = _axes(res, device)
    px = ((x + 0.5) * dx - center[0]).abs() - half[0]
    py = ((y + 0.5) * dx - center[1]).abs() - half[1]
    pz = ((z + 0.5) * dx - center[2]).abs() - half[2]
    qx = px[None, None, :]
    qy = py[None, :, None]
    qz = pz[:, None, None]
    outside = torch.sqrt(qx.clamp(min=0) ** 2 + qy.clamp(min=0) ** 2 + qz.clamp(min=0) ** 2)
    inside = torch.maximum(torch.maximum(qx, qy), qz).clamp(max=0)
    return (outside + inside).to(torch.float32).contiguous()


def wall_sdf(res, dx, x_wall, device="cpu"):
    """Solid occupying x < x_wall (positive inside the solid, cpp:1157)."""
    x, y, z = _axes(res, device)
    s = (x_wall - (x + 0.5) * dx)[None, None, :].expand(res[2], res[1], res[0])
    return s.to(torch.float32).contiguous()


def smooth_velocity(res, dx, gravity_dt=0.0, device="cpu"):
    """Deterministic smooth face velocity: u = (sin 2pi y cos 2pi z, ...)."""
    nx, ny, nz = res
    two_pi = 2.0 * math.pi
    out = []
    for axis in range(3):
        r = [nx, ny, nz]
        r[axis] += 1
        ax = []
        for a in range(3):
            i = torch.arange(r[a], device=device, dtype=torch.float64)
            ax.append((i if a == axis else i + 0.5) * dx)
        X = ax[0][None, None, :]
        Y = ax[1][None, :, None]
        Z = ax[2][:, None, None]
        if axis == 0:
            v = torch.sin(two_pi * Y) * torch.cos(two_pi * Z) + 0 * X
        elif axis == 1:
            v = torch.sin(two_pi * Z) * torch.cos(two_pi * X) + 0 * Y - gravity_dt
        else:
            v = torch.sin(two_pi * X) * torch.cos(two_pi * Y) + 0 * Z
        out.append(v.to(torch.float32).contiguous())
    return out


def constant_velocity(res, c, device="cpu"):
    nx, ny, nz = res
    out = []
    for axis in range(3):
        r = [nx, ny, nz]
        r[axis] += 1
        out.append(torch.full((r[2], r[1], r[0]), float(c[axis]), dtype=torch.float32, device=device))
    return out


def fat_beam(n, levels, *, variable_viscosity=False, wall=False, device="cpu", dt=1.0 / 60.0,
             viscosity=10000.0, density=1000.0, res=None):
    """The "fat beam" of SURVEY 8(d): half-extents (0.45, 0.225, 0.225) centred in the unit cube."""
    res = res or (n, n, n)
    dx = 1.0 / n
    size = (res[0] * dx, res[1] * dx, res[2] * dx)
    center = (0.5 * size[0], 0.5 * size[1], 0.5 * size[2])
    half = (0.45 * size[0], 0.225 * min(size[1], 1.0), 0.225 * min(size[2], 1.0))
    liquid = box_sdf(res, dx, center, half, device)
    solid = wall_sdf(res, dx, center[0] - half[0] + 3.2 * dx, device) if wall else None
    visc = viscosity
    if variable_viscosity:
        x = (torch.arange(res[0], device=device, dtype=torch.float64) + 0.5) * dx
        visc = (200.0 * (1.0 + 9.0 * x))[None, None, :].expand(res[2], res[1], res[0]).to(torch.float32).contiguous()
    vel = smooth_velocity(res, dx, gravity_dt=9.80665 * dt, device=device)
    return Scene(res=res, dx=dx, dt=dt, levels=levels, liquid=liquid, solid=solid, viscosity=visc,
                 density=density, velocity=vel, name=f"fat_beam_{n}_L{levels}")


def thin_sheet(n, levels, thickness_cells=16, device="cpu", dt=1.0 / 120.0, viscosity=200.0):
    res = (n, n, n)
    dx = 1.0 / n
    liquid = box_sdf(res, dx, (0.5, 0.5, 0.5), (0.45, 0.45, 0.5 * thickness_cells * dx), device)
    vel = smooth_velocity(res, dx, gravity_dt=9.80665 * dt, device=device)
    return Scene(res=res, dx=dx, dt=dt, levels=levels, liquid=liquid, viscosity=viscosity,
                 density=1000.0, velocity=vel, name=f"thin_sheet_{n}_L{levels}")


def sphere(n, levels, radius=0.3, device="cpu", dt=1.0 / 60.0, viscosity=100.0):
    res = (n, n, n)
    dx = 1.0 / n
    x, y, z = _axes(res, device)
    X = ((x + 0.5) * dx - 0.5)[None, None, :]
    Y = ((y + 0.5) * dx - 0.5)[None, :, None]
    Z = ((z + 0.5) * dx - 0.5)[:, None, None]
    liquid = (torch.sqrt(X * X + Y * Y + Z * Z) - radius).to(torch.float32).contiguous()
    vel = smooth_velocity(res, dx, device=device)
    return Scene(res=res, dx=dx, dt=dt, levels=levels, liquid=liquid, viscosity=viscosity,
                 density=1000.0, velocity=vel, name=f"sphere_{n}_L{levels}")


def to_device(scene, device):
    """Same scene with every tensor moved to `device` (bit-identical inputs on host and in HBM: parity
    tests generate once on the host instead of trusting sin / sqrt to agree across devices)."""
    import copy
    mv = lambda t: t.to(device) if isinstance(t, torch.Tensor) else t
    out = copy.copy(scene)
    out.liquid = mv(scene.liquid)
    out.solid = mv(scene.solid)
    out.viscosity = mv(scene.viscosity)
    out.density = mv(scene.density)
    out.velocity = [mv(v) for v in scene.velocity]
    out.solid_velocity = None if scene.solid_velocity is None else [mv(v) for v in scene.solid_velocity]
    return out
