#!/usr/bin/env python3
"""
SURVEY §8 f6 fixture: a window of the reference's OWN scene files (tests/commit/field/velo_001000.npz / dens_001000.npz -- a real
256 x 128 smoke-plume state written by PhiFlow's `field.write`, legacy key set) re-saved in the same format, plus the oracle's result
of one fluid step from that state. /root/reference does not exist on the GPU box, so the window travels as a committed fixture:

    python tests/golden/make_scene_golden.py      # needs /root/reference; writes scene_velo_crop.npz, scene_dens_crop.npz, scene_step.npz

The window keeps the file format byte-for-byte in its metadata entries (dim_names, dim_types, field_type, extrapolation: copied from
the source files); only `data` is cropped and `lower` / `upper` are moved to the window.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import phi_oracle as O   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/tests/commit/field"
Y0, X0, N = 100, 32, 64            # window: cells [Y0, Y0+N) x [X0, X0+N) of the 256 x 128 grid


def main():
    velo = np.load(os.path.join(REF, "velo_001000.npz"), allow_pickle=True)
    dens = np.load(os.path.join(REF, "dens_001000.npz"), allow_pickle=True)
    assert list(velo['dim_names']) == ['y', 'x', 'vector'] and velo['data'].shape == (257, 129, 2)
    h = 200.0 / 256
    lower, upper = np.asarray([Y0 * h, X0 * h]), np.asarray([(Y0 + N) * h, (X0 + N) * h])
    vcrop = np.ascontiguousarray(velo['data'][Y0:Y0 + N + 1, X0:X0 + N + 1, :])
    dcrop = np.ascontiguousarray(dens['data'][Y0:Y0 + N, X0:X0 + N])
    np.savez_compressed(os.path.join(HERE, "scene_velo_crop.npz"), dim_names=velo['dim_names'], dim_types=velo['dim_types'],
                        field_type=velo['field_type'], lower=lower, upper=upper, extrapolation=velo['extrapolation'], data=vcrop)
    np.savez_compressed(os.path.join(HERE, "scene_dens_crop.npz"), dim_names=dens['dim_names'], dim_types=dens['dim_types'],
                        dim_item_names=dens['dim_item_names'], field_type=dens['field_type'], lower=lower.astype(np.float32),
                        upper=upper.astype(np.float32), bounds_item_names=dens['bounds_item_names'], extrapolation=dens['extrapolation'],
                        data=dcrop)
    # one step from that state with the oracle: open (BOUNDARY) velocity, smoke: zero-gradient along x, constant 0 along y
    dom = O.Domain((N, N), tuple(lower), tuple(upper), ((O.OPEN, O.OPEN),) * 2)
    v = [vcrop[:, :-1, 0][None].copy(), vcrop[:-1, :, 1][None].copy()]                   # BOUNDARY stores N + 1 faces along the normal
    s_codes = ((O.CLOSED, O.CLOSED), (O.OPEN, O.OPEN))                                  # dims (y, x)
    s = dcrop[None].copy()
    s = O.semi_lagrangian_centered(s, v, 1.0, dom, s_codes)
    buoy = O.centered_to_staggered(s, dom, s_codes, None, (0.1, 0.0))                     # buoyancy along y (the first dim of the files)
    v = O.semi_lagrangian_staggered(v, v, 1.0, dom)
    v = [a + b for a, b in zip(v, buoy)]
    v, p, info, _ = O.make_incompressible(v, dom, rtol=1e-5, atol=0.0)
    np.savez_compressed(os.path.join(HERE, "scene_step.npz"), smoke=s[0], vy=v[0][0], vx=v[1][0], p=p[0], iterations=info.iterations)
    print("written; CG iterations", info.iterations)


if __name__ == "__main__":
    main()
