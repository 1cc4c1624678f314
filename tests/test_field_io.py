""" PhiFlow .npz field format (reference phi/field/_field_io.py) """
import os

import numpy as np
import pytest

from phiflow_amd import field_io
from phiflow_amd.flow import BOUNDARY, PERIODIC, ZERO, Box, CenteredGrid, StaggeredGrid, combine_sides

REF_FIXTURES = "/root/reference/tests/commit/field"


@pytest.mark.parametrize("ext", [ZERO, BOUNDARY, PERIODIC, combine_sides(x=BOUNDARY, y=(ZERO, BOUNDARY))])
def test_roundtrip(emu_backend, tmp_path, ext):
    rng = np.random.default_rng(0)
    shapes = StaggeredGrid(0, ext, x=6, y=5, backend=emu_backend).component_shapes
    v = StaggeredGrid([rng.standard_normal(s).astype(np.float32) for s in shapes], ext, Box(x=3, y=(1, 2)), x=6, y=5, backend=emu_backend)
    s = CenteredGrid(rng.standard_normal((6, 5)).astype(np.float32), ext, Box(x=3, y=(1, 2)), x=6, y=5, backend=emu_backend)
    field_io.write(v, str(tmp_path / "v.npz"))
    field_io.write(s, str(tmp_path / "s.npz"))
    stored = np.load(tmp_path / "v.npz", allow_pickle=True)
    assert stored['data'].shape == (7, 6, 2) and str(stored['field_type']) == 'StaggeredGrid'       # padded staggered tensor
    v2 = field_io.read(str(tmp_path / "v.npz"), emu_backend)
    s2 = field_io.read(str(tmp_path / "s.npz"), emu_backend)
    assert v2.is_staggered and v2.resolution == v.resolution and v2.boundary == v.boundary and v2.bounds.lower == v.bounds.lower
    for a, b in zip(v.numpy(), v2.numpy()):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(s.numpy(), s2.numpy())


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_FIXTURES, "velo_001000.npz")), reason="reference fixtures not present")
def test_read_reference_fixtures(emu_backend):
    """ tests/commit/field/test__scene.py:158-165 (legacy centred + staggered grids written by PhiFlow itself) """
    density = field_io.read(os.path.join(REF_FIXTURES, "dens_001000.npz"), emu_backend)
    assert density.is_grid and density.is_centered and density.resolution == {'y': 256, 'x': 128}
    assert density.bounds.upper == (200.0, 100.0)
    velocity = field_io.read(os.path.join(REF_FIXTURES, "velo_001000.npz"), emu_backend)
    assert velocity.is_grid and velocity.is_staggered and velocity.resolution == {'y': 256, 'x': 128}
    assert [tuple(t.shape[1:]) for t in velocity.values] == [(257, 128), (256, 129)]          # BOUNDARY: N+1 faces
    raw = np.load(os.path.join(REF_FIXTURES, "velo_001000.npz"), allow_pickle=True)['data']
    np.testing.assert_array_equal(velocity.numpy()[0], raw[:, :-1, 0])
    np.testing.assert_array_equal(velocity.numpy()[1], raw[:-1, :, 1])
