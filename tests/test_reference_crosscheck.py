"""
Optional cross-check of the ORACLE against the real reference: runs only where `phiml` and `phi` are
importable (they are not in the build container, nor on the GPU box: the test is skipped there). It executes the reference's own
`advect.semi_lagrangian` / `fluid.make_incompressible` (phi/physics/advect.py:156, phi/physics/fluid.py:94) on the NumPy backend on the
parity set-ups of tests/commit/physics/test_fluid.py:19-53 and compares with oracle/phi_oracle.py on the same arrays -- the only way the
`[PHIML-RECALL]` parts of the oracle (face alignment of PERIODIC grids, CG recipe, tolerances) ever get pinned.
"""
import numpy as np
import pytest

phiml = pytest.importorskip("phiml")
if str(getattr(phiml, "__version__", "")).endswith("test-double"):
    pytest.skip("tests/fake_phiml is a test double, not PhiML", allow_module_level=True)
pytest.importorskip("phi")

from oracle import phi_oracle as O   # noqa: E402


def _reference_case(ext_name):
    from phi.flow import StaggeredGrid, CenteredGrid, Box, extrapolation, Noise, math
    ext = {'closed': 0, 'open': extrapolation.BOUNDARY, 'periodic': extrapolation.PERIODIC}[ext_name]
    math.seed(0)
    v = StaggeredGrid(Noise(), ext, x=16, y=20, bounds=Box(x=100, y=100)) * 0.2
    smoke = CenteredGrid(Noise(), extrapolation.BOUNDARY, x=16, y=20, bounds=Box(x=100, y=100))
    return v, smoke


def _arrays(v):
    return [v.vector[d].values.numpy('x,y')[None].astype(np.float32) for d in ('x', 'y')]


@pytest.mark.parametrize("ext_name,code", [('closed', O.CLOSED), ('open', O.OPEN), ('periodic', O.PERIODIC)])
def test_oracle_matches_the_reference_numpy_path(ext_name, code):
    from phi.flow import Solve, advect, fluid
    v, _ = _reference_case(ext_name)
    dom = O.Domain((16, 20), (0, 0), (100, 100), ((code, code),) * 2)
    vin = _arrays(v)
    assert [a.shape[1:] for a in vin] == [dom.comp_shape(0), dom.comp_shape(1)]          # Appendix C1: stored-face layout
    adv_ref = _arrays(advect.semi_lagrangian(v, v, 1.0))
    adv_orc = O.semi_lagrangian_staggered(vin, vin, 1.0, dom)
    for a, b in zip(adv_ref, adv_orc):
        np.testing.assert_allclose(a, b, atol=1e-5)                                      # the reference's cross-backend criterion
    v2, p = fluid.make_incompressible(v, (), Solve('CG', 1e-5, 0))
    vo, po, info, _ = O.make_incompressible(vin, dom, rtol=1e-5, atol=0.0)
    pr = p.values.numpy('x,y')[None]
    if not dom.flexible():
        pr, po = pr - pr.mean(), po - po.mean()
    assert np.linalg.norm(pr - po) / np.linalg.norm(po) <= 1e-3
    for a, b in zip(_arrays(v2), vo):
        np.testing.assert_allclose(a, b, atol=1e-4)
