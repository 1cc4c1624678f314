"""
CPU ORACLE (TEST INFRASTRUCTURE ONLY -- never imported by the product package `phiflow_amd`).

NumPy restatement of PhiFlow's incompressible-fluid time step on a StaggeredGrid:
``advect.semi_lagrangian`` + ``fluid.make_incompressible`` (reference: /root/reference, PhiFlow 3.4.0).

Why a restatement: the arithmetic of this path lives in the third-party dependency ``phiml`` (pinned
``phiml>=1.14.0``, reference setup.py:41), which is neither vendored (empty ``PhiML/`` submodule, .gitmodules:1-3)
nor installable here (no network). The functions below restate the PhiFlow call sites and PhiML's published
algorithms (multilinear ``grid_sample``, ``pad`` per extrapolation, Shewchuk-style ``cg``).

PARITY PIN STATUS
  * pinned against every known-answer / property test the reference holds for the path
    (tests/test_oracle_reference_pins.py): tests/commit/physics/test_advect.py:41-45 (exact vector),
    test_advect.py:12-18 (identity advection), tests/commit/physics/test_fluid.py:19-53 (div <= 5e-5 for closed /
    open / periodic / mixed, batched), tests/commit/field/test__grid.py:25-36 (stored-face counts),
    tests/commit/field/test__field_math.py:85-88, tests/commit/physics/test_diffuse.py:68-80 (stencil values).
  * the reference stores NO golden output for make_incompressible / CG and cannot be executed here, so absolute
    pressure values are "parity unpinned" by the reference itself; they are guarded instead by independent
    cross-checks (discrete-FFT Poisson solve, SciPy sparse direct solve of the assembled operator).
  * restated from the reference's call sites WITHOUT any reference test to pin them ("parity unpinned"): `cg_adaptive`, union /
    embedded obstacles, sampling between different grids, the rk4 back-trace, moving / rotating obstacles, MacCormack's limiter.

Conventions (reference citations are relative to /root/reference):
  * arrays are (batch, *spatial) C-contiguous, spatial order x,y,(z) => last spatial axis is the fast one
    (phi/field/_field.py:160-180).
  * boundary codes per axis side describe the *velocity* extrapolation:
      PERIODIC          -> extrapolation.PERIODIC
      CLOSED (+ values) -> ConstantExtrapolation (ZERO = all values 0); the wall face is NOT stored
      OPEN              -> extrapolation.BOUNDARY / ZERO_GRADIENT; the outer face IS stored
    ``valid_outer_faces``: PERIODIC -> (True, False), CLOSED -> (False, False), OPEN -> (True, True)
    (docs/Scene_Format_Specification.md:27, phi/field/_field_math.py:562-575).
"""
from dataclasses import dataclass, field as _dc_field
from typing import List, Optional, Sequence, Tuple

import numpy as np

PERIODIC, CLOSED, OPEN = 0, 1, 2


# --------------------------------------------------------------------------------------------------------------------
# Domain description
# --------------------------------------------------------------------------------------------------------------------
@dataclass
class Domain:
    """Uniform grid + velocity boundary description (UniformGrid + Extrapolation; phi/geom/_grid.py:41-122)."""
    res: Tuple[int, ...]                       # cells per axis (x, y[, z])
    lower: Tuple[float, ...]
    upper: Tuple[float, ...]
    bc: Tuple[Tuple[int, int], ...]            # per axis (lower side code, upper side code)
    bc_val: Optional[np.ndarray] = None        # [axis][side][component] constant velocity on CLOSED sides

    def __post_init__(self):
        self.res = tuple(int(r) for r in self.res)
        self.lower = tuple(float(v) for v in self.lower)
        self.upper = tuple(float(v) for v in self.upper)
        self.bc = tuple((int(lo), int(hi)) for lo, hi in self.bc)
        D = len(self.res)
        if self.bc_val is None:
            self.bc_val = np.zeros((D, 2, D))
        self.bc_val = np.asarray(self.bc_val, dtype=np.float64).reshape(D, 2, D)
        for lo, hi in self.bc:
            assert (lo == PERIODIC) == (hi == PERIODIC), "periodic must be set on both sides of an axis"

    @property
    def rank(self):
        return len(self.res)

    @property
    def dx(self):
        """ dx = size / resolution  (phi/geom/_grid.py:120-122) """
        return tuple((u - l) / r for l, u, r in zip(self.lower, self.upper, self.res))

    def valid_faces(self, d):
        lo, hi = self.bc[d]
        return lo != CLOSED, hi == OPEN

    def comp_shape(self, d):
        """ stored shape of velocity component d (phi/geom/_grid.py:204-209; tests/commit/field/test__grid.py:25-36) """
        lo, hi = self.valid_faces(d)
        s = list(self.res)
        s[d] += int(lo) + int(hi) - 1
        return tuple(s)

    def face_offset(self, d):
        """ physical face number of stored index 0 along d: 0 if the lower face is stored else 1 """
        return 0 if self.valid_faces(d)[0] else 1

    def flexible(self):
        """ extrapolation.is_flexible: True when any side is OPEN (BOUNDARY), cf. SURVEY Appendix C5 / fluid.py:145 """
        return any(c == OPEN for pair in self.bc for c in pair)


def pressure_bc(dom: Domain):
    """ fluid._pressure_extrapolation (phi/physics/fluid.py:264-274): PERIODIC->PERIODIC, OPEN->ZERO (Dirichlet
    ghost), CLOSED->BOUNDARY (Neumann). Returned with the same codes, read as: PERIODIC wrap / CLOSED zero-gradient /
    OPEN zero ghost. """
    return dom.bc


# --------------------------------------------------------------------------------------------------------------------
# Padding == boundary conditions (phiml.math.pad semantics; docs/Fields.md:114-121)
# --------------------------------------------------------------------------------------------------------------------
def _pad_axis(a: np.ndarray, axis: int, lo: int, hi: int, code_lo: int, code_hi: int, c_lo: float, c_hi: float):
    """Pad spatial `axis` (array axis = axis+1 because of the batch dim) by (lo, hi) samples."""
    ax = axis + 1
    n = a.shape[ax]
    parts = []
    if lo > 0:
        if code_lo == PERIODIC:
            idx = np.arange(-lo, 0) % n
            parts.append(np.take(a, idx, axis=ax))
        elif code_lo == OPEN:
            parts.append(np.repeat(np.take(a, [0], axis=ax), lo, axis=ax))
        else:
            shp = list(a.shape); shp[ax] = lo
            parts.append(np.full(shp, c_lo, dtype=a.dtype))
    parts.append(a)
    if hi > 0:
        if code_hi == PERIODIC:
            idx = np.arange(n, n + hi) % n
            parts.append(np.take(a, idx, axis=ax))
        elif code_hi == OPEN:
            parts.append(np.repeat(np.take(a, [n - 1], axis=ax), hi, axis=ax))
        else:
            shp = list(a.shape); shp[ax] = hi
            parts.append(np.full(shp, c_hi, dtype=a.dtype))
    out = np.concatenate(parts, axis=ax) if len(parts) > 1 else a
    if lo < 0:
        out = np.take(out, np.arange(-lo, out.shape[ax]), axis=ax)
    if hi < 0:
        out = np.take(out, np.arange(0, out.shape[ax] + hi), axis=ax)
    return out


def pad_component(a: np.ndarray, comp: int, widths: Sequence[Tuple[int, int]], dom: Domain):
    """ math.pad(values, widths, extrapolation[{'vector': comp}]) for a velocity component.
    Mixed per-side extrapolations pad axis after axis in spatial order, later axes see the already padded array. """
    for axis, (lo, hi) in enumerate(widths):
        if lo == 0 and hi == 0:
            continue
        a = _pad_axis(a, axis, lo, hi, dom.bc[axis][0], dom.bc[axis][1],
                      dom.bc_val[axis][0][comp], dom.bc_val[axis][1][comp])
    return a


def pad_scalar(a: np.ndarray, widths, codes, consts=None):
    """ pad a centred scalar field; codes per axis side use PERIODIC (wrap) / OPEN (edge copy) / CLOSED (constant). """
    for axis, (lo, hi) in enumerate(widths):
        if lo == 0 and hi == 0:
            continue
        c = (0.0, 0.0) if consts is None else consts[axis]
        a = _pad_axis(a, axis, lo, hi, codes[axis][0], codes[axis][1], c[0], c[1])
    return a


# --------------------------------------------------------------------------------------------------------------------
# Multilinear sampling (phiml.math.grid_sample)
# --------------------------------------------------------------------------------------------------------------------
def _tap(a: np.ndarray, idx: List[np.ndarray], codes, consts):
    """Value of `a` (batch, *spatial) at integer indices idx[axis] (each (batch, *pts)), out-of-range indices resolved
    by the extrapolation. Sequential-pad semantics: the LAST axis that is outside a constant side wins."""
    D = a.ndim - 1
    B = a.shape[0]
    const_mask = np.zeros(idx[0].shape, dtype=bool)
    const_val = np.zeros(idx[0].shape, dtype=a.dtype)
    clipped = []
    for axis in range(D):
        n = a.shape[axis + 1]
        i = idx[axis]
        lo_code, hi_code = codes[axis]
        below, above = i < 0, i >= n
        if lo_code == PERIODIC:
            j = np.mod(i, n)
        else:
            j = np.clip(i, 0, n - 1)
            if lo_code == CLOSED:
                const_val = np.where(below, np.asarray(consts[axis][0], dtype=a.dtype), const_val)
                const_mask = const_mask | below
            if hi_code == CLOSED:
                const_val = np.where(above, np.asarray(consts[axis][1], dtype=a.dtype), const_val)
                const_mask = const_mask | above
        clipped.append(j)
    bidx = np.arange(B).reshape((B,) + (1,) * (idx[0].ndim - 1))
    vals = a[(np.broadcast_to(bidx, idx[0].shape),) + tuple(clipped)]
    return np.where(const_mask, const_val, vals)


def grid_sample(a: np.ndarray, coords: List[np.ndarray], codes, consts):
    """ multilinear interpolation of `a` at fractional index coordinates coords[axis] (0 = first sample).
    neighbors * prod(where(binary, frac, 1-frac)) summed over the 2^D taps. """
    D = a.ndim - 1
    fl = [np.floor(c) for c in coords]
    fr = [(c - f).astype(a.dtype) for c, f in zip(coords, fl)]
    i0 = [f.astype(np.int64) for f in fl]
    out = np.zeros(coords[0].shape, dtype=a.dtype)
    for corner in range(1 << D):
        w = np.ones(coords[0].shape, dtype=a.dtype)
        idx = []
        for axis in range(D):
            bit = (corner >> axis) & 1
            idx.append(i0[axis] + bit)
            w = w * (fr[axis] if bit else (1 - fr[axis]))
        out = out + _tap(a, idx, codes, consts) * w
    return out


def _comp_codes(dom: Domain, comp: int):
    codes = dom.bc
    consts = [(dom.bc_val[axis][0][comp], dom.bc_val[axis][1][comp]) for axis in range(dom.rank)]
    return codes, consts


# --------------------------------------------------------------------------------------------------------------------
# Staggered helpers
# --------------------------------------------------------------------------------------------------------------------
def bake_component(v_d: np.ndarray, d: int, dom: Domain):
    """ field.bake_extrapolation for one component: pad the normal axis so that all N_d+1 faces are present
    (phi/field/_field_math.py:20-39). """
    lo, hi = dom.valid_faces(d)
    widths = [(0, 0)] * dom.rank
    widths[d] = (0 if lo else 1, 0 if hi else 1)
    return pad_component(v_d, d, widths, dom)


def component_at_faces(v: List[np.ndarray], c: int, d: int, dom: Domain):
    """ value of component c at the stored faces of component d (c != d): `_shift_resample` + `sample_subgrid`
    (phi/field/_resample.py:279-287, 341-364) == mean of the 4 surrounding c-faces, missing ones from padding.
    sample_subgrid lerps axis after axis with weights (0.5, 0.5): upper*0.5 + lower*0.5. """
    D = dom.rank
    a = v[c]
    off_c, off_d = dom.face_offset(c), dom.face_offset(d)
    n_d = dom.comp_shape(d)
    # along axis c: need c-faces with physical numbers i and i+1 for i in [0, res_c)  -> stored idx i-off_c, i+1-off_c
    # along axis d: need cells m-1 and m for physical face m in [off_d, off_d + n_d[d])
    widths = [(0, 0)] * D
    have_c = a.shape[c + 1]
    lo_c = off_c                                   # faces below stored index 0 that are needed: physical face 0
    hi_c = dom.res[c] + 1 - off_c - have_c         # faces above
    widths[c] = (lo_c, hi_c)
    first_cell = off_d - 1                         # lowest cell index needed along d
    last_cell = off_d + n_d[d] - 1                 # highest cell index needed along d
    widths[d] = (max(0, -first_cell), max(0, last_cell - (dom.res[d] - 1)))
    p = pad_component(a, c, widths, dom)
    # now p has along c: res_c + 1 faces (physical 0..res_c); along d: cells first_cell' .. covering [first_cell, last_cell]
    start_d = first_cell + widths[d][0]
    sl_lo = [slice(None)] * (D + 1)
    sl_hi = [slice(None)] * (D + 1)
    # lerp order follows spatial dims order (x, y, z)
    res = p
    for axis in sorted((c, d)):
        sl_lo = [slice(None)] * (D + 1)
        sl_hi = [slice(None)] * (D + 1)
        if axis == c:
            sl_lo[axis + 1] = slice(0, dom.res[c])
            sl_hi[axis + 1] = slice(1, dom.res[c] + 1)
        else:
            sl_lo[axis + 1] = slice(start_d, start_d + n_d[d])
            sl_hi[axis + 1] = slice(start_d + 1, start_d + n_d[d] + 1)
        res = res[tuple(sl_hi)] * res.dtype.type(0.5) + res[tuple(sl_lo)] * res.dtype.type(0.5)
    assert res.shape[1:] == n_d, (res.shape, n_d)
    return res


def face_positions(d: int, dom: Domain, dtype):
    """ world coordinates of the stored faces of component d (Field.points with BC-determined faces sliced off,
    phi/field/_field.py:146-154; UniformGrid.center of the staggered sub-grid, phi/geom/_grid.py:59-63,204-209). """
    D = dom.rank
    n = dom.comp_shape(d)
    dx = dom.dx
    off = dom.face_offset(d)
    axes = []
    for a in range(D):
        if a == d:
            # sub-grid bounds: lower + (off - 0.5)*dx ; centres at lower + (off - 0.5 + i + 0.5)*dx
            lo = dom.lower[a] + (off - 0.5) * dx[a]
        else:
            lo = dom.lower[a]
        size = n[a] * dx[a]
        local = np.linspace(0.5 / n[a], 1 - 0.5 / n[a], n[a]).astype(dtype)
        axes.append((local * dtype(size) + dtype(lo)).astype(dtype))
    return np.meshgrid(*axes, indexing='ij')


def _index_coords(points: List[np.ndarray], comp: int, dom: Domain, dtype):
    """ local = bounds_c.global_to_local(points) * resolution_c - 0.5 on the component's own sub-grid
    (phi/field/_resample.py:257-258, phi/geom/_box.py:134-152). """
    D = dom.rank
    n = dom.comp_shape(comp)
    dx = dom.dx
    off = dom.face_offset(comp)
    out = []
    for a in range(D):
        lo = dom.lower[a] + ((off - 0.5) * dx[a] if a == comp else 0.0)
        size = n[a] * dx[a]
        out.append((points[a] - dtype(lo)) / dtype(size) * dtype(n[a]) - dtype(0.5))
    return out


# --------------------------------------------------------------------------------------------------------------------
# sampling between DIFFERENT grids (phi/field/_resample.py:66-72,145-161,241-259: sample -> grid_sample at explicit points;
# phi/physics/advect.py:193 "velocity need not be sampled at same locations as field"; Batched_Smoke.ipynb)
# --------------------------------------------------------------------------------------------------------------------
def _centered_index_coords(points: List[np.ndarray], dom: Domain, dtype):
    return [(points[a] - dtype(dom.lower[a])) / dtype(dom.res[a] * dom.dx[a]) * dtype(dom.res[a]) - dtype(0.5) for a in range(dom.rank)]


def sample_staggered_at(v: List[np.ndarray], dom: Domain, points: List[np.ndarray]):
    """ sample(velocity, points): every component interpolated on its own staggered sub-grid with its own padding rule """
    dtype = v[0].dtype.type
    out = []
    for c in range(dom.rank):
        codes, consts = _comp_codes(dom, c)
        out.append(grid_sample(v[c], _index_coords(points, c, dom, dtype), codes, consts))
    return out


def sample_centered_at(s: np.ndarray, dom: Domain, s_codes, s_consts, points: List[np.ndarray]):
    consts = s_consts if s_consts is not None else [(0.0, 0.0)] * dom.rank
    return grid_sample(s, _centered_index_coords(points, dom, s.dtype.type), s_codes, consts)


def integrate_points(pts: List[np.ndarray], vel: List[np.ndarray], dom_v: Domain, dt: float, integrator: str = 'euler'):
    """ advect.euler / advect.rk4 (phi/physics/advect.py:20-36): end points of `pts` moved by the velocity for the time dt (negative
    for the back-trace); every velocity evaluation is sample(velocity, points) on the staggered sub-grids """
    dtype = pts[0].dtype.type
    v0 = sample_staggered_at(vel, dom_v, pts)
    if integrator == 'euler':
        return [p + u * dtype(dt) for p, u in zip(pts, v0)]
    assert integrator == 'rk4', integrator
    v_half = sample_staggered_at(vel, dom_v, [p + dtype(0.5 * dt) * u for p, u in zip(pts, v0)])
    v_half2 = sample_staggered_at(vel, dom_v, [p + dtype(0.5 * dt) * u for p, u in zip(pts, v_half)])
    v_full = sample_staggered_at(vel, dom_v, [p + dtype(dt) * u for p, u in zip(pts, v_half2)])
    v_rk4 = [dtype(1 / 6.) * (a + dtype(2) * (b + c) + d) for a, b, c, d in zip(v0, v_half, v_half2, v_full)]
    return [p + dtype(dt) * u for p, u in zip(pts, v_rk4)]


def semi_lagrangian_centered_general(s: np.ndarray, dom_s: Domain, velocity: List[np.ndarray], dom_v: Domain, dt: float, s_codes, s_consts=None,
                                     correction_strength: Optional[float] = None, integrator: str = 'euler'):
    """ advect.semi_lagrangian (correction_strength None) / advect.mac_cormack of a centred scalar by a velocity on another grid """
    dtype = s.dtype.type
    B = max(s.shape[0], velocity[0].shape[0])
    pts = [np.broadcast_to(p[None], (B,) + p.shape) for p in cell_positions(dom_s, dtype)]
    vel = [np.broadcast_to(c, (B,) + c.shape[1:]) for c in velocity]
    src = np.broadcast_to(s, (B,) + s.shape[1:])
    back = integrate_points(pts, vel, dom_v, -dt, integrator)
    consts = s_consts if s_consts is not None else [(0.0, 0.0)] * dom_s.rank
    fwd = sample_centered_at(src, dom_s, s_codes, consts, back)
    if correction_strength is None:
        return fwd
    ahead = integrate_points(pts, vel, dom_v, dt, integrator)
    bwd = sample_centered_at(fwd, dom_s, s_codes, consts, ahead)
    new = fwd + dtype(correction_strength * 0.5) * (src - bwd)
    lo, hi = closest_limits(src, _centered_index_coords(back, dom_s, dtype), s_codes, consts)
    return np.minimum(np.maximum(new, lo), hi)


def semi_lagrangian_staggered_general(field: List[np.ndarray], dom_f: Domain, velocity: List[np.ndarray], dom_v: Domain, dt: float,
                                      integrator: str = 'euler'):
    """ advect.semi_lagrangian of a staggered field by a velocity on another grid and / or with the rk4 integrator: per component,
    the stored faces are traced back and the component is interpolated on its own sub-grid """
    dtype = field[0].dtype.type
    B = max(field[0].shape[0], velocity[0].shape[0])
    vel = [np.broadcast_to(c, (B,) + c.shape[1:]) for c in velocity]
    out = []
    for d in range(dom_f.rank):
        pts = [np.broadcast_to(p[None], (B,) + p.shape) for p in face_positions(d, dom_f, dtype)]
        back = integrate_points(pts, vel, dom_v, -dt, integrator)
        codes, consts = _comp_codes(dom_f, d)
        out.append(grid_sample(np.broadcast_to(field[d], (B,) + field[d].shape[1:]), _index_coords(back, d, dom_f, dtype), codes, consts))
    return out


def resample_centered_general(s: np.ndarray, dom_s: Domain, s_codes, s_consts, dom_t: Domain, staggered: bool = False,
                              vector: Optional[Sequence[float]] = None):
    """ resample(s [* vector], to=target) with the target on another grid: at its cell centres, or per component at its stored faces """
    dtype = s.dtype.type
    if not staggered:
        pts = [np.broadcast_to(p[None], (s.shape[0],) + p.shape) for p in cell_positions(dom_t, dtype)]
        return sample_centered_at(s, dom_s, s_codes, s_consts, pts)
    vector = [1.0] * dom_t.rank if vector is None else vector
    out = []
    for d in range(dom_t.rank):
        pts = [np.broadcast_to(p[None], (s.shape[0],) + p.shape) for p in face_positions(d, dom_t, dtype)]
        out.append(sample_centered_at(s, dom_s, s_codes, s_consts, pts) * dtype(vector[d]))
    return out


# --------------------------------------------------------------------------------------------------------------------
# a1: semi-Lagrangian advection (phi/physics/advect.py:156-179 with euler :20-24)
# --------------------------------------------------------------------------------------------------------------------
def semi_lagrangian_staggered(field: List[np.ndarray], velocity: List[np.ndarray], dt: float, dom: Domain,
                              field_dom: Optional[Domain] = None):
    """ advect the staggered `field` by the staggered `velocity` (self-advection when both are the same).
    `field_dom` carries the field's own boundary values if they differ from the velocity's (same face layout). """
    field_dom = field_dom or dom
    D = dom.rank
    dtype = field[0].dtype.type
    out = []
    for d in range(D):
        # euler(): v0 = sample(velocity, field.geometry, at='face') ; lookup = points + v0 * (-dt)
        pts = face_positions(d, field_dom, dtype)
        u = [velocity[d] if c == d else component_at_faces(velocity, c, d, dom) for c in range(D)]
        lookup = [pts[a][None] + u[a] * dtype(-dt) for a in range(D)]
        # reduce_sample(): component d of the field sampled at its own lookup points
        coords = _index_coords(lookup, d, field_dom, dtype)
        codes, consts = _comp_codes(field_dom, d)
        out.append(grid_sample(field[d], coords, codes, consts))
    return out


def staggered_at_centers(v: List[np.ndarray], dom: Domain):
    """ staggered velocity sampled at cell centres: mean of the two faces of each cell along the component axis
    (sample_staggered_grid -> _shift_resample; phi/field/_resample.py:279-287). """
    D = dom.rank
    out = []
    for c in range(D):
        b = bake_component(v[c], c, dom)
        lo = [slice(None)] * (D + 1); hi = [slice(None)] * (D + 1)
        lo[c + 1] = slice(0, dom.res[c]); hi[c + 1] = slice(1, dom.res[c] + 1)
        out.append(b[tuple(hi)] * b.dtype.type(0.5) + b[tuple(lo)] * b.dtype.type(0.5))
    return out


def cell_positions(dom: Domain, dtype):
    axes = []
    for a in range(dom.rank):
        n = dom.res[a]
        local = np.linspace(0.5 / n, 1 - 0.5 / n, n).astype(dtype)
        axes.append((local * dtype(dom.upper[a] - dom.lower[a]) + dtype(dom.lower[a])).astype(dtype))
    return np.meshgrid(*axes, indexing='ij')


def semi_lagrangian_centered(s: np.ndarray, velocity: List[np.ndarray], dt: float, dom: Domain,
                             s_codes, s_consts=None):
    """ advect a centred scalar (e.g. smoke) by a staggered velocity. s_codes: per axis side PERIODIC / OPEN
    (zero-gradient) / CLOSED (constant s_consts). """
    D = dom.rank
    dtype = s.dtype.type
    pts = cell_positions(dom, dtype)
    u = staggered_at_centers(velocity, dom)
    coords = []
    for a in range(D):
        look = pts[a][None] + u[a] * dtype(-dt)
        coords.append((look - dtype(dom.lower[a])) / dtype(dom.upper[a] - dom.lower[a]) * dtype(dom.res[a]) - dtype(0.5))
    consts = s_consts if s_consts is not None else [(0.0, 0.0)] * D
    return grid_sample(s, coords, s_codes, consts)


# --------------------------------------------------------------------------------------------------------------------
# f2: MacCormack advection (phi/physics/advect.py:182-215) and centred -> staggered resampling (buoyancy)
# --------------------------------------------------------------------------------------------------------------------
def closest_limits(a: np.ndarray, coords: List[np.ndarray], codes, consts):
    """ min / max over the 2^D grid values surrounding the fractional index coordinates: `Field.closest_values`
    (phi/field/_field.py:409-429 -> phiml closest_grid_values: the floor / floor+1 taps, outside taps from the
    extrapolation) reduced with math.min / math.max over the `closest_<dim>` dims (advect.py:210-212). """
    D = a.ndim - 1
    i0 = [np.floor(c).astype(np.int64) for c in coords]
    lo = hi = None
    for corner in range(1 << D):
        idx = [i0[axis] + ((corner >> axis) & 1) for axis in range(D)]
        t = _tap(a, idx, codes, consts)
        lo = t if lo is None else np.minimum(lo, t)
        hi = t if hi is None else np.maximum(hi, t)
    return lo, hi


def _centered_lookup_coords(velocity: List[np.ndarray], dt: float, dom: Domain, dtype):
    """ index coordinates of `cell centre + dt * v0`, v0 = staggered velocity at the centres (euler, advect.py:20-24) """
    pts = cell_positions(dom, dtype)
    u = staggered_at_centers(velocity, dom)
    coords = []
    for a in range(dom.rank):
        look = pts[a][None] + u[a] * dtype(dt)
        coords.append((look - dtype(dom.lower[a])) / dtype(dom.upper[a] - dom.lower[a]) * dtype(dom.res[a]) - dtype(0.5))
    return coords


def mac_cormack_centered(s: np.ndarray, velocity: List[np.ndarray], dt: float, dom: Domain, s_codes, s_consts=None,
                         correction_strength: float = 1.0):
    """ advect.mac_cormack for a centred scalar (advect.py:203-215): forward + backward semi-Lagrangian pass, error
    correction `fwd + strength * 0.5 * (field - bwd)`, clamped to the closest grid values of the backward lookup. """
    D = dom.rank
    dtype = s.dtype.type
    consts = s_consts if s_consts is not None else [(0.0, 0.0)] * D
    c_bwd = _centered_lookup_coords(velocity, -dt, dom, dtype)
    c_fwd = _centered_lookup_coords(velocity, dt, dom, dtype)
    fwd_adv = grid_sample(s, c_bwd, s_codes, consts)
    bwd_adv = grid_sample(fwd_adv, c_fwd, s_codes, consts)
    new = fwd_adv + dtype(correction_strength * 0.5) * (s - bwd_adv)
    lo, hi = closest_limits(s, c_bwd, s_codes, consts)
    return np.clip(new, lo, hi)


def mac_cormack_staggered(field: List[np.ndarray], velocity: List[np.ndarray], dt: float, dom: Domain,
                          correction_strength: float = 1.0):
    """ advect.mac_cormack for a StaggeredGrid advected by a StaggeredGrid with the same boundary (advect.py:203-215).
    Note on the limiter: `Field.closest_values` (phi/field/_field.py:427-429) returns from its "CenteredGrid" branch for
    every field, i.e. it converts the lookup points with the CELL grid's box and resolution (`box.global_to_local(points)
    * resolution - 0.5`) also for staggered components. Along its own axis d a face with physical number m therefore gets
    the index coordinate m - 0.5 instead of its stored index m - off_d: the clamp window of component d is shifted by
    half a cell. This restatement follows the reference literally (the staggered branch below that return is dead code). """
    D = dom.rank
    dtype = field[0].dtype.type
    out = []
    for d in range(D):
        pts = face_positions(d, dom, dtype)
        u = [velocity[d] if c == d else component_at_faces(velocity, c, d, dom) for c in range(D)]
        p_bwd = [pts[a][None] + u[a] * dtype(-dt) for a in range(D)]
        p_fwd = [pts[a][None] + u[a] * dtype(dt) for a in range(D)]
        codes, consts = _comp_codes(dom, d)
        fwd_adv = grid_sample(field[d], _index_coords(p_bwd, d, dom, dtype), codes, consts)
        bwd_adv = grid_sample(fwd_adv, _index_coords(p_fwd, d, dom, dtype), codes, consts)
        new = fwd_adv + dtype(correction_strength * 0.5) * (field[d] - bwd_adv)
        cell_frame = [(p_bwd[a] - dtype(dom.lower[a])) / dtype(dom.upper[a] - dom.lower[a]) * dtype(dom.res[a]) - dtype(0.5)
                      for a in range(D)]
        lo, hi = closest_limits(field[d], cell_frame, codes, consts)
        out.append(np.clip(new, lo, hi))
    return out


def centered_to_staggered(s: np.ndarray, dom: Domain, s_codes, s_consts=None, vector: Optional[Sequence[float]] = None):
    """ `resample(s * vector, to=velocity)` / `s * vector @ velocity` (Smoke_Plume.ipynb cell 5, test_fluid.py:26):
    sample_grid_at_faces (phi/field/_resample.py:272-276) -> per component the mean of the two cells adjacent to each
    stored face, cells outside the domain from the scalar's extrapolation; then times the constant vector component. """
    D = dom.rank
    dtype = s.dtype.type
    vector = [1.0] * D if vector is None else vector
    out = []
    for d in range(D):
        widths = [(0, 0)] * D
        widths[d] = (1, 1)
        p = pad_scalar(s, widths, s_codes, s_consts)
        lo = [slice(None)] * (D + 1); hi = [slice(None)] * (D + 1)
        off = dom.face_offset(d)
        n = dom.comp_shape(d)[d]
        lo[d + 1] = slice(off, off + n); hi[d + 1] = slice(off + 1, off + 1 + n)
        p = p * dtype(vector[d])     # `s * vector` is formed first, then resampled
        out.append(p[tuple(lo)] * dtype(0.5) + p[tuple(hi)] * dtype(0.5))
    return out


# --------------------------------------------------------------------------------------------------------------------
# a2: divergence (phi/field/_field_math.py:589,617-626)
# --------------------------------------------------------------------------------------------------------------------
def divergence(v: List[np.ndarray], dom: Domain):
    D = dom.rank
    comps = []
    for d in range(D):
        b = bake_component(v[d], d, dom)
        lo = [slice(None)] * (D + 1); hi = [slice(None)] * (D + 1)
        lo[d + 1] = slice(0, dom.res[d]); hi[d + 1] = slice(1, dom.res[d] + 1)
        comps.append((b[tuple(hi)] - b[tuple(lo)]) / b.dtype.type(dom.dx[d]))
    out = comps[0]
    for c in comps[1:]:
        out = out + c
    return out


# --------------------------------------------------------------------------------------------------------------------
# a4/a6: pressure gradient at faces and masked Laplace (phi/physics/fluid.py:165-202,158-161;
#        phi/field/_field_math.py:234-236,535-581)
# --------------------------------------------------------------------------------------------------------------------
def _pressure_pad_codes(dom: Domain):
    """ codes for pad_scalar of the pressure: PERIODIC wrap; velocity CLOSED -> pressure BOUNDARY (edge copy == OPEN code
    of pad_scalar); velocity OPEN -> pressure ZERO (constant 0 == CLOSED code of pad_scalar). """
    conv = {PERIODIC: PERIODIC, CLOSED: OPEN, OPEN: CLOSED}
    return tuple((conv[lo], conv[hi]) for lo, hi in dom.bc)


def pressure_gradient(p: np.ndarray, dom: Domain):
    """ spatial_gradient(p, v.extrapolation, at='face'): one array per component with the velocity's stored-face shape. """
    D = dom.rank
    codes = _pressure_pad_codes(dom)
    out = []
    for d in range(D):
        lo_valid, hi_valid = dom.valid_faces(d)
        if lo_valid and hi_valid:
            wl, wu = (1, 0), (0, 1)
        elif lo_valid and not hi_valid:
            wl, wu = (1, -1), (0, 0)
        elif not lo_valid and hi_valid:
            wl, wu = (0, 0), (-1, 1)
        else:
            wl, wu = (0, -1), (-1, 0)
        W = [(0, 0)] * D
        W[d] = wl
        lower = pad_scalar(p, W, codes)
        W[d] = wu
        upper = pad_scalar(p, W, codes)
        out.append((upper - lower) / p.dtype.type(dom.dx[d]))
    return out


def masked_laplace(p: np.ndarray, dom: Domain, hard_bcs: Optional[List[np.ndarray]] = None,
                   active: Optional[np.ndarray] = None):
    """ fluid.masked_laplace (order 2, staggered): div(hard_bcs * grad p), identity on inactive cells. The gradient
    field carries `remove_constant_offset(v_boundary)` so wall faces contribute zero flux. """
    grad = pressure_gradient(p, dom)
    if hard_bcs is not None:
        grad = [g * h.astype(g.dtype) for g, h in zip(grad, hard_bcs)]
    zero_dom = Domain(dom.res, dom.lower, dom.upper, dom.bc, np.zeros_like(dom.bc_val))
    div = divergence(grad, zero_dom)
    if active is not None:
        return np.where(active > 0, div, p)
    return div


# --------------------------------------------------------------------------------------------------------------------
# a5: conjugate gradients (phiml cg)
# --------------------------------------------------------------------------------------------------------------------
@dataclass
class SolveInfo:
    iterations: np.ndarray
    residual_sq: np.ndarray
    rhs_sq: np.ndarray
    converged: np.ndarray
    diverged: np.ndarray


def _bsum(a):
    return a.reshape(a.shape[0], -1).sum(axis=1)


def _bshape(s, a):
    return s.reshape((-1,) + (1,) * (a.ndim - 1))


def cg(apply_A, y: np.ndarray, x0: np.ndarray, rtol: float, atol: float, max_iter: int, refresh: int = 50):
    """ Shewchuk CG batched over the leading axis, all reductions per batch row.
        tol^2 = max(rtol^2 * sum(y^2), atol^2); stops per row when rsq <= tol^2 / diverged / max_iter. """
    dtype = y.dtype.type
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        x = x0.astype(y.dtype).copy()
        r = y - apply_A(x)
        d = r.copy()
        q = apply_A(d)
        B = y.shape[0]
        rhs_sq = _bsum(y * y)
        tol_sq = np.maximum(dtype(rtol) ** 2 * rhs_sq, dtype(atol) ** 2)
        rsq = _bsum(r * r)
        rsq0 = rsq.copy()
        iterations = np.zeros(B, dtype=np.int32)
        diverged = ~np.all(np.isfinite(x.reshape(B, -1)), axis=1)
        converged = rsq <= tol_sq
        cont = ~converged & ~diverged & (iterations < max_iter)
        it_counter = 0
        while np.any(cont):
            it_counter += 1
            iterations += cont.astype(np.int32)
            dq = _bsum(d * q)
            alpha = np.where(dq != 0, rsq / np.where(dq != 0, dq, 1), 0).astype(y.dtype)
            alpha = alpha * cont.astype(y.dtype)
            x = x + _bshape(alpha, x) * d
            if refresh and it_counter % refresh == 0:
                r = y - apply_A(x)
            else:
                r = r - _bshape(alpha, r) * q
            rsq_old = rsq
            rsq = _bsum(r * r)
            beta = np.where(rsq_old != 0, rsq / np.where(rsq_old != 0, rsq_old, 1), 0).astype(y.dtype)
            d = r + _bshape(beta, d) * d
            q = apply_A(d)
            diverged = (rsq / rsq0 > 100) & (iterations >= 8)
            converged = rsq <= tol_sq
            cont = cont & ~converged & ~diverged & (iterations < max_iter)
    return x, SolveInfo(iterations, rsq, rhs_sq, converged, diverged)


def cg_adaptive(apply_A, y: np.ndarray, x0: np.ndarray, rtol: float, atol: float, max_iter: int, refresh: int = 20):
    """ PhiML's 'CG-adaptive' (`Solve('CG-adaptive', ...)`, /root/reference examples/grids/Fluid_Logo.ipynb):
        the same loop as `cg` with   alpha = sum(d r) / sum(d q)   and   d = r - (sum(r q) / sum(d q)) d,
        true-residual refresh every `refresh` iterations (0 = never). Parity unpinned: phiml is not available here. """
    dtype = y.dtype.type
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        x = x0.astype(y.dtype).copy()
        r = y - apply_A(x)
        d = r.copy()
        q = apply_A(d)
        B = y.shape[0]
        rhs_sq = _bsum(y * y)
        tol_sq = np.maximum(dtype(rtol) ** 2 * rhs_sq, dtype(atol) ** 2)
        rsq = _bsum(r * r)
        rsq0 = rsq.copy()
        iterations = np.zeros(B, dtype=np.int32)
        diverged = ~np.all(np.isfinite(x.reshape(B, -1)), axis=1)
        converged = rsq <= tol_sq
        cont = ~converged & ~diverged & (iterations < max_iter)
        it_counter = 0
        while np.any(cont):
            it_counter += 1
            iterations += cont.astype(np.int32)
            dq = _bsum(d * q)
            dq_safe = np.where(dq != 0, dq, 1)
            alpha = np.where(dq != 0, _bsum(d * r) / dq_safe, 0).astype(y.dtype)
            alpha = alpha * cont.astype(y.dtype)
            x = x + _bshape(alpha, x) * d
            if refresh and it_counter % refresh == 0:
                r = y - apply_A(x)
            else:
                r = r - _bshape(alpha, r) * q
            rsq = _bsum(r * r)
            beta = np.where(dq != 0, -_bsum(r * q) / dq_safe, 0).astype(y.dtype)
            d = r + _bshape(beta, d) * d
            q = apply_A(d)
            diverged = (rsq / rsq0 > 100) & (iterations >= 8)
            converged = rsq <= tol_sq
            cont = cont & ~converged & ~diverged & (iterations < max_iter)
    return x, SolveInfo(iterations, rsq, rhs_sq, converged, diverged)


def laplace_csr(dom: Domain, dtype=np.float32):
    """ the obstacle-free pressure operator of `masked_laplace` ASSEMBLED as a SciPy CSR matrix (N x N, 5 / 7 entries per row) -- what
    `math.solve_linear` actually iterates on in the reference: PhiML traces `masked_laplace` into a sparse matrix per call
    (`jit_compile_linear`, phi/physics/fluid.py:165) and its NumPy backend multiplies with scipy.sparse (SURVEY §8 a4/a5
    [PHIML-RECALL]). Used for the CPU timing of the sparse-matrix CG variant and as an independent check of the stencil. """
    import scipy.sparse as sp
    res = dom.res
    N = int(np.prod(res))
    idx = np.arange(N).reshape(res)
    rows, cols, vals = [], [], []
    diag = np.zeros(res, dtype=np.float64)
    pbc = pressure_bc(dom)
    for a in range(dom.rank):
        w = 1.0 / (dom.dx[a] * dom.dx[a])
        for side, shift in ((0, -1), (1, 1)):
            nb = np.roll(idx, -shift, axis=a)                    # index of the neighbour in direction `shift`
            inside = np.ones(res, dtype=bool)
            edge = [slice(None)] * dom.rank
            edge[a] = 0 if shift < 0 else res[a] - 1
            code = pbc[a][side]
            if code != PERIODIC:
                inside[tuple(edge)] = False
                if code == OPEN:                                 # open velocity boundary = pressure ZERO ghost: the diagonal keeps -w
                    d = np.zeros(res)
                    d[tuple(edge)] = -w
                    diag += d
            rows.append(idx[inside]); cols.append(nb[inside]); vals.append(np.full(int(inside.sum()), w))
            diag[inside] -= w
    rows.append(idx.ravel()); cols.append(idx.ravel()); vals.append(diag.ravel())
    A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(N, N))
    A.sum_duplicates()
    return A.astype(dtype)


# --------------------------------------------------------------------------------------------------------------------
# a7: obstacle masks (phi/physics/fluid.py:130-137,212-240,277-288; phi/geom/_box.py:174-185,217-236;
#     phi/geom/_geom.py:278-308)
# --------------------------------------------------------------------------------------------------------------------
@dataclass
class BoxObstacle:
    lower: Tuple[float, ...]
    upper: Tuple[float, ...]
    velocity: Optional[Tuple[float, ...]] = None           # Obstacle.velocity (fluid.py:27-37)
    angular_velocity: Optional[object] = None              # scalar (2-D) or vector (3-D)
    rotation: Optional[np.ndarray] = None                  # (D, D) matrix box frame -> world (Box.rotated, _box.py:127-152)

    @property
    def center(self):
        return tuple((l + u) / 2 for l, u in zip(self.lower, self.upper))

    def _local(self, pts):
        """ global_to_local(scale=False, origin='center'): R^T (x - center) """
        r = [p - c for p, c in zip(pts, self.center)]
        if self.rotation is None:
            return r
        R = np.asarray(self.rotation, dtype=float)
        return [sum(R[c][a] * r[c] for c in range(len(r))) for a in range(len(r))]

    def lies_inside(self, pts):
        half = [(u - l) / 2 for l, u in zip(self.lower, self.upper)]
        ok = np.ones(pts[0].shape, dtype=bool)
        for a, p in enumerate(self._local(pts)):
            ok &= np.abs(p) <= half[a]
        return ok

    def sdf(self, pts):
        half = [(u - l) / 2 for l, u in zip(self.lower, self.upper)]
        dist = None
        for a, p in enumerate(self._local(pts)):
            da = np.abs(p) - half[a]
            dist = da if dist is None else np.maximum(dist, da)
        return dist


@dataclass
class SphereObstacle:
    center: Tuple[float, ...]
    radius: float
    velocity: Optional[Tuple[float, ...]] = None
    angular_velocity: Optional[object] = None

    def lies_inside(self, pts):
        d2 = sum((p - c) ** 2 for p, c in zip(pts, self.center))
        return d2 <= self.radius ** 2

    def sdf(self, pts):
        d2 = sum((p - c) ** 2 for p, c in zip(pts, self.center))
        return np.sqrt(d2) - self.radius


@dataclass
class EmbeddedObstacle:
    """ Obstacle(embed(geometry, dims)) / geom.infinite_cylinder (phi/geom/_embed.py:38-45,62-66,139-158): the inner geometry is
    evaluated on the coordinates of `axes` only (indices into the domain's axes); it is infinitely long along the others """
    inner: object
    axes: Tuple[int, ...]
    velocity: Optional[Tuple[float, ...]] = None
    angular_velocity: Optional[object] = None

    @property
    def center(self):
        raise NotImplementedError("embedded geometries do not rotate")

    def lies_inside(self, pts):
        return self.inner.lies_inside([pts[a] for a in self.axes])

    def sdf(self, pts):
        return self.inner.sdf([pts[a] for a in self.axes])


@dataclass
class UnionObstacle:
    """ Obstacle(union(geometries)) (phi/geom/_geom_ops.py:96-102, 297-319; stacked boxes: phi/geom/_box.py:175,235): inside = any
    member, signed distance = min over the members; one linear velocity for the whole body, no rotation """
    members: Tuple[object, ...]
    velocity: Optional[Tuple[float, ...]] = None
    angular_velocity: Optional[object] = None

    @property
    def center(self):
        return self.members[0].center      # only used with an angular velocity, which unions do not have here

    def lies_inside(self, pts):
        out = self.members[0].lies_inside(pts)
        for m in self.members[1:]:
            out = out | m.lies_inside(pts)
        return out

    def sdf(self, pts):
        out = self.members[0].sdf(pts)
        for m in self.members[1:]:
            out = np.minimum(out, m.sdf(pts))
        return out


def obstacle_masks(obstacles, dom: Domain, dtype=np.float32):
    """ returns (active[cells] in {0,1}, hard_bcs[d][faces] in {0,1}, soft[d][faces] in [0,1]).
      accessible = ~union(obstacles) sampled hard at cell centres; outside-domain accessibility from
      _accessible_extrapolation (periodic wrap / OPEN 1 / CLOSED 0); hard_bcs = stagger(accessible, minimum);
      soft face mask m = clip(1 - sdf/r, 0, 1) with r = |half size of a face cell| (balance=1), union over obstacles by max. """
    D = dom.rank
    dt = np.float64
    cpts = cell_positions(dom, dt)
    inside = np.zeros(dom.res, dtype=bool)
    for ob in obstacles:
        inside |= ob.lies_inside(cpts)
    active = (~inside).astype(dtype)[None]
    acc_codes = tuple((PERIODIC if lo == PERIODIC else CLOSED, PERIODIC if hi == PERIODIC else CLOSED) for lo, hi in dom.bc)
    acc_consts = [tuple(1.0 if c == OPEN else 0.0 for c in pair) for pair in dom.bc]
    hard, soft = [], []
    radius = float(np.sqrt(sum((0.5 * h) ** 2 for h in dom.dx)))
    for d in range(D):
        lo_valid, hi_valid = dom.valid_faces(d)
        if lo_valid and hi_valid:
            wl, wu = (1, 0), (0, 1)
        elif lo_valid and not hi_valid:
            wl, wu = (1, -1), (0, 0)
        elif not lo_valid and hi_valid:
            wl, wu = (0, 0), (-1, 1)
        else:
            wl, wu = (0, -1), (-1, 0)
        W = [(0, 0)] * D
        W[d] = wl
        lower = pad_scalar(active, W, acc_codes, acc_consts)
        W[d] = wu
        upper = pad_scalar(active, W, acc_codes, acc_consts)
        hard.append(np.minimum(lower, upper))
        # factor that apply_boundary_conditions multiplies stationary velocities with: the obstacles are applied one after
        # the other (fluid.py:225-239), i.e. prod_i (1 - mask_i); returned as soft = 1 - prod for the callers' `1 - soft`
        fpts = face_positions(d, dom, dt)
        keep = np.ones(dom.comp_shape(d), dtype=dtype)
        for ob in obstacles:
            frac = np.clip(1.0 - ob.sdf(fpts) / radius, 0, 1).astype(dtype)
            keep = keep * (1 - frac)
        soft.append((1 - keep)[None])
    return active, hard, soft


def apply_boundary_conditions(v: List[np.ndarray], obstacles, dom: Domain):
    """ fluid.apply_boundary_conditions (fluid.py:212-240): per obstacle, in order,
        mask = resample(geometry, velocity, soft=True, balance=1) = clip(1 - sdf(face) / bounding_radius(face cell), 0, 1)
        v = safe_mul(1 - mask, v) [+ safe_mul(mask, angular_velocity x (x - center) + velocity) for moving obstacles]
    (AngularVelocity without falloff: cross(strength, distances), phi/field/_angular_velocity.py:40-46). """
    D = dom.rank
    dtype = v[0].dtype.type
    radius = float(np.sqrt(sum((0.5 * h) ** 2 for h in dom.dx)))
    out = []
    for d in range(D):
        fpts = face_positions(d, dom, np.float64)
        val = v[d].copy()
        for ob in obstacles:
            m = np.clip(1.0 - ob.sdf(fpts) / radius, 0, 1).astype(dtype)[None]
            keep = 1 - m
            val = np.where(keep == 0, dtype(0), keep * val)
            lin = ob.velocity if ob.velocity is not None else (0.0,) * D
            ang = ob.angular_velocity if ob.angular_velocity is not None else 0.0
            moving = any(float(c) != 0 for c in lin) or np.any(np.asarray(ang, dtype=float) != 0)
            if moving:
                if not np.any(np.asarray(ang, dtype=float) != 0):
                    u = np.zeros(fpts[0].shape)
                elif D == 2:
                    r = [fpts[a] - ob.center[a] for a in range(D)]
                    w = float(np.asarray(ang, dtype=float).reshape(-1)[0])
                    u = (-w * r[1], w * r[0])[d]
                else:
                    r = [fpts[a] - ob.center[a] for a in range(D)]
                    w = np.broadcast_to(np.asarray(ang, dtype=float), (3,)) if np.ndim(ang) == 0 else np.asarray(ang, dtype=float)
                    u = (w[1] * r[2] - w[2] * r[1], w[2] * r[0] - w[0] * r[2], w[0] * r[1] - w[1] * r[0])[d]
                u = (u + float(lin[d])).astype(dtype)[None]
                val = val + np.where(m == 0, dtype(0), m * u)
        out.append(val)
    return out


# --------------------------------------------------------------------------------------------------------------------
# make_incompressible (phi/physics/fluid.py:94-162)
# --------------------------------------------------------------------------------------------------------------------
def gradient_subtract(v: List[np.ndarray], p: np.ndarray, dom: Domain, hard_bcs=None):
    grad = pressure_gradient(p, dom)
    if hard_bcs is not None:
        grad = [g * h.astype(g.dtype) for g, h in zip(grad, hard_bcs)]
    return [vd - g for vd, g in zip(v, grad)]


def balance_divergence(div: np.ndarray, active: Optional[np.ndarray]):
    """ fluid._balance_divergence (fluid.py:205-209) with field.mean over non-batch dims (per batch entry). """
    axes = tuple(range(1, div.ndim))
    if active is not None:
        return div - active * (div.mean(axis=axes, keepdims=True, dtype=div.dtype) / active.mean(axis=axes, keepdims=True, dtype=div.dtype))
    return div - div.mean(axis=axes, keepdims=True, dtype=div.dtype)


def make_incompressible(v: List[np.ndarray], dom: Domain, obstacles=(), x0: Optional[np.ndarray] = None,
                        rtol: float = 1e-5, atol: float = 0.0, max_iter: int = 1000, refresh: Optional[int] = None,
                        balance: Optional[bool] = None, method: str = 'CG', active_user: Optional[np.ndarray] = None):
    """ returns (velocity, pressure, SolveInfo, div_rhs). `method`: 'CG' (refresh 50) or 'CG-adaptive' (refresh 20).
    `active_user` = the `active` argument of fluid.make_incompressible (fluid.py:97): cells where the pressure is solved; with it the
    divergence is never balanced (fluid.py:145: `and all_active`) and non-finite divergence values become 0 (fluid.py:143-144). """
    dtype = v[0].dtype
    hard = active = None
    all_active = active_user is None                                       # fluid.py:124
    if obstacles:
        active, hard, soft = obstacle_masks(obstacles, dom, dtype.type)
        if active_user is not None:
            active = np.asarray(active_user, dtype) * active               # fluid.py:136 "no pressure inside obstacles"
        v = apply_boundary_conditions(v, obstacles, dom)
    elif active_user is not None:
        active = np.broadcast_to(np.asarray(active_user, dtype), (v[0].shape[0],) + tuple(dom.res)).copy()
    div = divergence(v, dom)
    if active is not None:
        with np.errstate(invalid='ignore'):
            div = div * active                                             # fluid.py:139-140 (a plain product: NaN * 0 = NaN)
    if not all_active:
        div = np.where(np.isfinite(div), div, dtype.type(0))               # fluid.py:143-144
    if balance is None:
        balance = (not dom.flexible()) and all_active                      # fluid.py:145
    rhs = balance_divergence(div, active) if balance else div
    if x0 is None:
        x0 = np.zeros_like(div)
    A = lambda p: masked_laplace(p, dom, hard, active)
    if method == 'CG-adaptive':
        p, info = cg_adaptive(A, rhs, x0, rtol, atol, max_iter, 20 if refresh is None else refresh)
    else:
        p, info = cg(A, rhs, x0, rtol, atol, max_iter, 50 if refresh is None else refresh)
    v_new = gradient_subtract(v, p, dom, hard)
    return v_new, p, info, rhs


# --------------------------------------------------------------------------------------------------------------------
# f1: explicit diffusion (phi/physics/diffuse.py:13-60; field.laplace order 2, _field_math.py:119-145)
# --------------------------------------------------------------------------------------------------------------------
def laplace_component(a: np.ndarray, comp: int, dom: Domain):
    """ 5/7-point Laplacian of a velocity component padded with the velocity's own extrapolation. """
    D = dom.rank
    p = pad_component(a, comp, [(1, 1)] * D, dom)
    out = np.zeros_like(a)
    core = tuple([slice(None)] + [slice(1, -1)] * D)
    for axis in range(D):
        lo = list(core); hi = list(core)
        lo[axis + 1] = slice(0, -2); hi[axis + 1] = slice(2, None)
        out = out + (p[tuple(lo)] + p[tuple(hi)] - a.dtype.type(2) * p[core]) / a.dtype.type(dom.dx[axis] ** 2)
    return out


def diffuse_explicit_centered(s: np.ndarray, diffusivity: float, dt: float, dom: Domain, s_codes, s_consts=None, substeps: int = 1):
    """ diffuse.explicit of a CenteredGrid (phi/physics/diffuse.py:13-60): s += (k dt / substeps) * laplace(s), the scalar's own
    extrapolation pads the stencil (field.laplace order 2, phi/field/_field_math.py:119-145) """
    D = dom.rank
    amount = s.dtype.type(diffusivity * dt / substeps)
    for _ in range(substeps):
        lap = np.zeros_like(s)
        for axis in range(D):
            widths = [(0, 0)] * D
            widths[axis] = (1, 1)
            p = pad_scalar(s, widths, s_codes, s_consts)
            lo = [slice(None)] * (D + 1); mid = [slice(None)] * (D + 1); hi = [slice(None)] * (D + 1)
            lo[axis + 1] = slice(0, -2); mid[axis + 1] = slice(1, -1); hi[axis + 1] = slice(2, None)
            lap = lap + (p[tuple(lo)] + p[tuple(hi)] - 2 * p[tuple(mid)]) / s.dtype.type(dom.dx[axis] ** 2)
        s = s + amount * lap
    return s


def diffuse_explicit(v: List[np.ndarray], diffusivity: float, dt: float, dom: Domain, substeps: int = 1):
    for _ in range(substeps):
        v = [vd + vd.dtype.type(diffusivity * dt / substeps) * laplace_component(vd, d, dom) for d, vd in enumerate(v)]
    return v


def diffuse_implicit_centered(s: np.ndarray, diffusivity: float, dt: float, dom: Domain, s_codes, s_consts=None, rtol: float = 1e-5,
                              atol: float = 0.0, max_iter: int = 1000):
    """ diffuse.implicit of a CenteredGrid (phi/physics/diffuse.py:63-92): solve_linear(sharpen, y=s, Solve('CG', x0=s)) with
    sharpen(x) = explicit(x, diffusivity, -dt). `sharpen` is affine when a constant extrapolation is not zero; solve_linear then solves
    the linear part against y - sharpen(0) (jit_compile_linear separates matrix and bias). Returns (u, SolveInfo). """
    sharpen = lambda x: diffuse_explicit_centered(x, diffusivity, -dt, dom, s_codes, s_consts)
    bias = sharpen(np.zeros_like(s))
    return cg(lambda x: sharpen(x) - bias, s - bias, s, rtol, atol, max_iter)


def diffuse_implicit(v: List[np.ndarray], diffusivity: float, dt: float, dom: Domain, rtol: float = 1e-5, atol: float = 0.0, max_iter: int = 1000):
    """ diffuse.implicit of a StaggeredGrid: the Laplacian acts on every component separately (laplace_component), so the solve
    decouples per component -- solved one after the other here with the tolerance relative to the COMPONENT's right-hand side
    (PhiML reduces over the whole staggered tensor; with a tolerance both stop within the same distance of the same solution).
    Returns ([u_d], [SolveInfo_d]). """
    out, infos = [], []
    for d, vd in enumerate(v):
        dtype = vd.dtype.type
        sharpen = lambda x, d=d, dtype=dtype: x - dtype(diffusivity * dt) * laplace_component(x, d, dom)
        bias = sharpen(np.zeros_like(vd))
        u, info = cg(lambda x: sharpen(x) - bias, vd - bias, vd, rtol, atol, max_iter)
        out.append(u)
        infos.append(info)
    return out, infos
