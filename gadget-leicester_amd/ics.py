"""Seeded synthetic initial conditions for the BASELINE.json configs.

Periodic unit box, code units G = 1.  Gas particles come first (indices [0, ngas)), as the
reference requires for SphP[] alignment (allvars.h:1384); DM is type 1.
"""
import numpy as np


def _lattice(ng, offset):
    g = (np.arange(ng) + offset) / ng
    x, y, z = np.meshgrid(g, g, g, indexing="ij")
    return np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1)


def _plane_wave_displacement(q, rng, nwaves, amp):
    """Zel'dovich-like displacement: sum of seeded plane waves with integer wave vectors."""
    d = np.zeros_like(q)
    for _ in range(nwaves):
        k = rng.integers(1, 4, size=3) * rng.choice([-1, 1], size=3)
        phase = rng.uniform(0, 2 * np.pi)
        direction = k / np.linalg.norm(k)
        d += np.sin(2 * np.pi * (q @ k) + phase)[:, None] * direction[None, :]
    return amp * d / np.sqrt(nwaves / 2.0)


def make_ics(ng, gas=True, seed=12345, clustered=True, omega_b_frac=0.16, jitter=0.05,
             rms_disp=1.0, u_gas=1.0e-2, vel_scale=0.1):
    """Returns dict(pos, vel, mass, type, id, u, ngas, boxsize, spacing).

    ng         particles per dimension per species (BASELINE c1: 32 DM only; c2: 64 + 64)
    clustered  False: lattice + uniform random displacement <= 0.2 spacing (config c1 recipe)
               True : plane-wave displacement of rms `rms_disp` spacings + `jitter` random
    """
    rng = np.random.default_rng(seed)
    spacing = 1.0 / ng
    species = []
    if gas:
        species.append((0, 0.5))   # gas lattice offset by half a cell
    species.append((1, 0.0))
    # one common displacement field (both species trace the same flow)
    waves_rng_state = rng.integers(0, 2**31)
    pos_l, vel_l, mass_l, type_l = [], [], [], []
    for ptype, off in species:
        q = _lattice(ng, off)
        if clustered:
            wr = np.random.default_rng(waves_rng_state)
            d = _plane_wave_displacement(q, wr, 3, rms_disp * spacing)
            d += rng.uniform(-jitter, jitter, size=q.shape) * spacing
        else:
            d = rng.uniform(-0.2, 0.2, size=q.shape) * spacing
        p = np.mod(q + d, 1.0)
        v = vel_scale * d / spacing
        total = 1.0  # total mass in the box
        if gas:
            frac = omega_b_frac if ptype == 0 else 1.0 - omega_b_frac
        else:
            frac = 1.0
        m = np.full(len(q), total * frac / len(q))
        pos_l.append(p)
        vel_l.append(v)
        mass_l.append(m)
        type_l.append(np.full(len(q), ptype, np.int32))
    pos = np.concatenate(pos_l)
    # keep strictly inside [0,1) after rounding
    pos[pos >= 1.0] = 0.0
    out = dict(pos=np.ascontiguousarray(pos), vel=np.ascontiguousarray(np.concatenate(vel_l)),
               mass=np.concatenate(mass_l), type=np.concatenate(type_l),
               ngas=(ng ** 3 if gas else 0), boxsize=1.0, spacing=spacing)
    n = len(out["pos"])
    out["id"] = np.arange(1, n + 1, dtype=np.uint32)
    out["u"] = np.full(out["ngas"], u_gas)
    return out


def make_plummer(n, seed=7, a=0.05, center=(0.5, 0.5, 0.5), gas_fraction=0.0):
    """Strongly clustered variant (stresses walk divergence); non-periodic use."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(0.001, 0.999, n)
    r = np.minimum(a / np.sqrt(x ** (-2.0 / 3.0) - 1.0), 0.45)
    costh = rng.uniform(-1, 1, n)
    phi = rng.uniform(0, 2 * np.pi, n)
    sinth = np.sqrt(1 - costh ** 2)
    pos = np.stack([r * sinth * np.cos(phi), r * sinth * np.sin(phi), r * costh], axis=1)
    pos += np.asarray(center)[None, :]
    ngas = int(n * gas_fraction)
    ptype = np.ones(n, np.int32)
    ptype[:ngas] = 0
    return dict(pos=np.ascontiguousarray(pos), vel=rng.normal(0, 0.1, (n, 3)),
                mass=np.full(n, 1.0 / n), type=ptype, ngas=ngas, boxsize=1.0,
                spacing=1.0 / round(n ** (1 / 3)), id=np.arange(1, n + 1, dtype=np.uint32),
                u=np.full(ngas, 1e-2))
