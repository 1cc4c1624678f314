"""
`field.write` / `field.read`: PhiFlow's `.npz` field format (reference: phi/field/_field_io.py:45-119,
docs/Scene_Format_Specification.md:15-28). Files written here can be read by PhiFlow and vice versa:

    keys: dim_names, dim_types, dim_item_names, field_type, lower, upper, bounds_item_names, extrapolation, data

A StaggeredGrid is stored as its padded `staggered_tensor()`: all components padded to `resolution + 1` along every spatial
dim and stacked along a trailing `vector` dim (phi/field/_field.py:586-604); reading slices the valid faces back out
(`unstack_staggered_tensor`, phi/field/_grid.py:179-187).
"""
import pickle
import zipfile
from typing import Optional

import numpy as np

from .backend import HipBackend, default_backend
from .extrapolation import (BOUNDARY, PERIODIC, ConstantExtrapolation, Extrapolation, _Boundary, _Mixed, _Periodic)
from .field import CenteredGrid, Field, StaggeredGrid
from .geom import Box


def extrapolation_to_dict(ext: Extrapolation, dims) -> dict:
    """ phiml `Extrapolation.to_dict()` """
    if isinstance(ext, _Periodic):
        return {'type': 'periodic'}
    if isinstance(ext, _Boundary):
        return {'type': 'boundary'}
    if isinstance(ext, ConstantExtrapolation):
        v = ext.value
        if isinstance(v, dict):
            v = [float(v.get(d, 0.0)) for d in dims]
        return {'type': 'constant', 'value': np.asarray(v)}
    if isinstance(ext, _Mixed):
        return {'type': 'mixed', 'dims': {d: (extrapolation_to_dict(lo, dims), extrapolation_to_dict(up, dims)) for d, (lo, up) in ext.ext.items()}}
    raise ValueError(ext)


def extrapolation_from_dict(d: dict, dims) -> Extrapolation:
    """ phiml `extrapolation.from_dict()` for the supported kinds """
    t = d['type']
    if t == 'periodic':
        return PERIODIC
    if t in ('boundary', 'zero-gradient'):
        return BOUNDARY
    if t == 'constant':
        v = np.asarray(d['value'])
        return ConstantExtrapolation(float(v) if v.ndim == 0 else [float(x) for x in v])
    if t == 'mixed':
        return _Mixed({dim: (extrapolation_from_dict(lo, dims), extrapolation_from_dict(up, dims)) for dim, (lo, up) in d['dims'].items()})
    raise NotImplementedError(f"extrapolation type {t!r} is not supported by the HIP backend")


class _MetadataUnpickler(pickle.Unpickler):
    """ The format stores three small metadata entries (`extrapolation`, `dim_item_names`, `bounds_item_names`) as pickled object
    arrays. `np.load(allow_pickle=True)` would run ANY pickle a shared scene file carries; this unpickler only rebuilds numpy arrays /
    scalars / dtypes and plain containers and refuses every other global. """
    _ALLOWED = {("numpy", "ndarray"), ("numpy", "dtype"), ("builtins", "dict"), ("builtins", "tuple"), ("builtins", "list"),
                ("builtins", "str"), ("builtins", "int"), ("builtins", "float"), ("builtins", "bool"), ("builtins", "complex"),
                ("builtins", "slice"), ("builtins", "NoneType")}
    _ALLOWED_NAMES = {"_reconstruct", "scalar", "_frombuffer"}      # numpy.core.multiarray / numpy._core.multiarray helpers

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED or (module.startswith("numpy") and "multiarray" in module and name in self._ALLOWED_NAMES) \
                or (module.startswith("numpy") and module.endswith("numeric") and name == "_frombuffer"):
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"field file metadata may only contain numpy arrays and plain containers, found {module}.{name}")


def _load_npz(file: str) -> dict:
    """ every entry of the .npz: numeric / string arrays through numpy (allow_pickle=False), object arrays through `_MetadataUnpickler` """
    out = {}
    with zipfile.ZipFile(file) as zf:
        for member in zf.namelist():
            key = member[:-4] if member.endswith(".npy") else member
            with zf.open(member) as fp:
                version = np.lib.format.read_magic(fp)
                shape, fortran, dtype = np.lib.format.read_array_header_1_0(fp) if version == (1, 0) else np.lib.format.read_array_header_2_0(fp)
                if dtype.hasobject:
                    out[key] = _MetadataUnpickler(fp).load()
                    continue
            with zf.open(member) as fp:
                out[key] = np.lib.format.read_array(fp, allow_pickle=False)
    return out


def write(field: Field, file: str):
    """ Writes a (non-batched or batched) grid field to `file` in PhiFlow's .npz format. """
    from .field import require_plain
    require_plain(field, 'write')
    dims = list(field.dims)
    D = len(dims)
    names = (['batch'] if field.batched else []) + dims
    types = (['batch'] if field.batched else []) + ['spatial'] * D
    if field.is_staggered:
        comps = field.numpy()
        comps = comps if field.batched else [c[None] for c in comps]
        B = comps[0].shape[0]
        full = np.zeros((B,) + tuple(field.resolution[d] + 1 for d in dims) + (D,), dtype=comps[0].dtype)
        # staggered_tensor(): pad the normal axis to N+1 faces with the boundary value, all other axes by one extra (edge) layer
        for i, d in enumerate(dims):
            c = field.values[i].detach().cpu().numpy()
            padded = _pad_component_for_storage(field, i, c)
            full[..., i] = padded
        data = full if field.batched else full[0]
        names, types = names + ['vector'], types + ['channel']
        item_names = [None] * (len(names) - 1) + [tuple(dims)]
        ftype = 'StaggeredGrid'
    else:
        data = field.numpy()
        item_names = [None] * len(names)
        ftype = 'CenteredGrid'
    np.savez_compressed(file, dim_names=names, dim_types=types, dim_item_names=np.asarray(item_names, dtype=object), field_type=ftype,
                        lower=np.asarray(field.bounds.lower), upper=np.asarray(field.bounds.upper), bounds_item_names=tuple(dims),
                        extrapolation=extrapolation_to_dict(field.boundary, dims), data=data)


def _pad_component_for_storage(field: Field, i: int, c: np.ndarray) -> np.ndarray:
    """ math.pad(component, {dim_i: (not lo_valid, not up_valid), others: (0, 1)}, extrapolation[vector=i]) """
    codes, vals = field._codes, field._bc_val
    for a, dim in enumerate(field.dims):
        lo_valid, up_valid = field.boundary.valid_outer_faces(dim)
        w = (int(not lo_valid), int(not up_valid)) if a == i else (0, 1)
        ax = a + 1
        parts = []
        for side, width in enumerate(w):
            if width == 0:
                continue
            code = codes[a][side]
            n = c.shape[ax]
            if code == 0:      # periodic
                sl = np.take(c, [n - 1] if side == 0 else [0], axis=ax)
            elif code == 2:    # boundary
                sl = np.take(c, [0] if side == 0 else [n - 1], axis=ax)
            else:
                sl = np.full_like(np.take(c, [0], axis=ax), vals[a][side][i])
            parts.append((side, sl))
        c = np.concatenate([p for s, p in parts if s == 0] + [c] + [p for s, p in parts if s == 1], axis=ax)
    return c


def read(file: str, backend: Optional[HipBackend] = None) -> Field:
    """ Loads a CenteredGrid / StaggeredGrid written by `write()` or by PhiFlow's `field.write` (incl. the legacy files
    tests/commit/field/dens_001000.npz / velo_001000.npz of the reference). Spatial dims keep the file's order. """
    stored = _load_npz(file)
    for key in ('field_type', 'data', 'dim_names', 'dim_types', 'lower', 'upper', 'extrapolation'):
        if key not in stored:
            raise ValueError(f"{file}: not a PhiFlow field file (missing '{key}')")
    ftype = str(stored['field_type'])
    if ftype not in ('CenteredGrid', 'StaggeredGrid'):
        raise NotImplementedError(f"{ftype} not implemented")
    data = stored['data']
    if not (isinstance(data, np.ndarray) and data.dtype.kind == 'f' and len(stored['dim_names']) == len(stored['dim_types']) == data.ndim):
        raise ValueError(f"{file}: 'data' must be a float array with one axis per entry of dim_names / dim_types")
    names = [str(n) for n in stored['dim_names']]
    types = [str(t) for t in stored['dim_types']]
    spatial = [n for n, t in zip(names, types) if t == 'spatial']
    batch = [n for n, t in zip(names, types) if t == 'batch']
    assert len(batch) <= 1, "only one batch dimension is supported"
    order = [names.index(b) for b in batch] + [names.index(s) for s in spatial] + [i for i, t in enumerate(types) if t == 'channel']
    data = np.transpose(data, order)
    bounds_names = stored.get('bounds_item_names')
    if bounds_names is None or getattr(bounds_names, 'shape', None) == () or bounds_names is None:
        bounds_names = spatial
    bounds_names = [str(b) for b in np.atleast_1d(bounds_names)] if not isinstance(bounds_names, list) else bounds_names
    lower, upper = np.atleast_1d(stored['lower']), np.atleast_1d(stored['upper'])
    lower = np.broadcast_to(lower, (len(spatial),)); upper = np.broadcast_to(upper, (len(spatial),))
    box = Box(**{d: (float(lower[bounds_names.index(d)]), float(upper[bounds_names.index(d)])) for d in spatial})
    ext = extrapolation_from_dict(stored['extrapolation'][()], spatial)
    backend = backend or default_backend()
    if ftype == 'CenteredGrid':
        res = {d: data.shape[len(batch) + i] for i, d in enumerate(spatial)}
        return CenteredGrid(data, ext, box, backend=backend, **res)
    res = {d: data.shape[len(batch) + i] - 1 for i, d in enumerate(spatial)}
    comps = []
    for i, d in enumerate(spatial):
        lo_valid, up_valid = ext.valid_outer_faces(d)
        sl = [slice(None)] * len(batch) + [slice(0, -1)] * len(spatial) + [i]
        sl[len(batch) + i] = slice(int(not lo_valid), -int(not up_valid) or None)
        comps.append(np.ascontiguousarray(data[tuple(sl)]))
    return StaggeredGrid(comps, ext, box, backend=backend, **res)
