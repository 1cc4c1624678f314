""" TEST DOUBLE: named-dimension tensors over torch natives -- only what phiflow_amd's plug-in glue calls. """
import numpy as np
import torch

BATCH, SPATIAL, CHANNEL, DUAL, INSTANCE = 'batch', 'spatial', 'channel', 'dual', 'instance'


class Shape:
    def __init__(self, names=(), sizes=(), types=(), item_names=None):
        self.names, self.sizes, self.types = tuple(names), tuple(sizes), tuple(types)
        self.item_names_ = dict(item_names or {})

    def __bool__(self): return len(self.names) > 0
    def __len__(self): return len(self.names)
    def __iter__(self): return iter(Shape([n], [s], [t], {n: self.item_names_[n]} if n in self.item_names_ else None) for n, s, t in zip(self.names, self.sizes, self.types))
    def __contains__(self, name): return (name.names[0] if isinstance(name, Shape) else name) in self.names
    def __and__(self, other):
        names, sizes, types, items = list(self.names), list(self.sizes), list(self.types), dict(self.item_names_)
        for n, s, t in zip(other.names, other.sizes, other.types):
            if n not in names:
                names.append(n); sizes.append(s); types.append(t)
                if n in other.item_names_: items[n] = other.item_names_[n]
        order = {BATCH: 0, INSTANCE: 1, SPATIAL: 2, CHANNEL: 3, DUAL: 3}
        idx = sorted(range(len(names)), key=lambda i: order[types[i]])
        return Shape([names[i] for i in idx], [sizes[i] for i in idx], [types[i] for i in idx], items)
    def get_size(self, name): return self.sizes[self.names.index(name)]
    def only(self, kind): 
        i = [k for k, t in enumerate(self.types) if t == kind]
        return Shape([self.names[k] for k in i], [self.sizes[k] for k in i], [self.types[k] for k in i], self.item_names_)
    def without(self, other):
        drop = set(other.names if isinstance(other, Shape) else ([other] if isinstance(other, str) else other))
        i = [k for k, n in enumerate(self.names) if n not in drop]
        return Shape([self.names[k] for k in i], [self.sizes[k] for k in i], [self.types[k] for k in i], self.item_names_)
    @property
    def volume(self): return int(np.prod(self.sizes)) if self.sizes else 1
    @property
    def batch(self): return self.only(BATCH)
    @property
    def spatial(self): return self.only(SPATIAL)
    @property
    def item_names(self): return self.item_names_.get(self.names[0]) if len(self.names) == 1 else None
    def __repr__(self): return "(" + ", ".join(f"{n}{'ᵇˢᶜᵈⁱ'[[BATCH, SPATIAL, CHANNEL, DUAL, INSTANCE].index(t)]}={s}" for n, s, t in zip(self.names, self.sizes, self.types)) + ")"


def _make(kind, *args, **dims):
    if args and isinstance(args[0], Tensor): return args[0].shape.only(kind)
    if args and hasattr(args[0], 'shape') and isinstance(getattr(args[0], 'shape'), Shape): return args[0].shape.only(kind)
    names, sizes, items = [], [], {}
    for a in args:
        for n in (a.split(',') if isinstance(a, str) else a): names.append(n.strip()); sizes.append(None)
    for n, s in dims.items():
        name = '~' + n if kind == DUAL and not n.startswith('~') else n
        if isinstance(s, (tuple, list)) or (isinstance(s, str) and ',' in s):
            it = tuple(s.split(',')) if isinstance(s, str) else tuple(s)
            items[name] = it; s = len(it)
        names.append(name); sizes.append(s)
    return Shape(names, sizes, [kind] * len(names), items)


def batch(*a, **d): return _make(BATCH, *a, **d)
def spatial(*a, **d): return _make(SPATIAL, *a, **d)
def channel(*a, **d): return _make(CHANNEL, *a, **d)
def dual(*a, **d): return _make(DUAL, *a, **d)
def instance(*a, **d): return _make(INSTANCE, *a, **d)


class _DimIndexer:
    def __init__(self, t, name): self.t, self.name = t, name
    def __getitem__(self, item):
        return self.t[{self.name: item}]


class Tensor:
    def __init__(self, native, shape: Shape):
        assert tuple(native.shape) == tuple(shape.sizes), (tuple(native.shape), shape)
        self._native, self.shape = native, shape

    def native(self, order):
        order = [o.strip() for o in order.split(',')] if isinstance(order, str) else list(order)
        assert sorted(order) == sorted(self.shape.names), (order, self.shape.names)
        return self._native.permute([self.shape.names.index(n) for n in order]) if len(order) > 1 else self._native

    def numpy(self, order): return self.native(order).detach().cpu().numpy()
    def __float__(self): return float(self._native.reshape(-1)[0]) if self.shape.volume == 1 else (_ for _ in ()).throw(ValueError("not a scalar"))

    def __getattr__(self, name):
        if name.startswith('_') or name in ('shape',): raise AttributeError(name)
        shp = object.__getattribute__(self, 'shape')
        if name in shp.names: return _DimIndexer(self, name)
        if '~' + name in shp.names: return _DimIndexer(self, '~' + name)
        raise AttributeError(name)

    def __getitem__(self, item):
        t = self
        for name, idx in item.items():
            name = name if name in t.shape.names else ('~' + name if '~' + name in t.shape.names else name)
            ax = t.shape.names.index(name)
            if isinstance(idx, str): idx = t.shape.item_names_[name].index(idx)
            native = t._native.select(ax, idx)
            t = Tensor(native, t.shape.without(name))
        return t


def tensor(native, shape: Shape = None, *more):
    if isinstance(native, Tensor): return native
    if isinstance(shape, Shape) and more: 
        for m in more: shape = shape & m
    if not isinstance(native, torch.Tensor): native = torch.as_tensor(np.asarray(native))
    if shape is None: shape = Shape()
    sizes = [native.shape[i] if s is None else s for i, s in enumerate(shape.sizes)]
    return Tensor(native, Shape(shape.names, sizes, shape.types, shape.item_names_))


wrap = tensor


def expand(t: Tensor, dims: Shape):
    native, shape = t._native, t.shape
    for n, s, ty in zip(dims.names, dims.sizes, dims.types):
        if n not in shape.names:
            native = native.unsqueeze(0).expand(s, *native.shape)
            shape = Shape((n,) + shape.names, (s,) + shape.sizes, (ty,) + shape.types, shape.item_names_)
    return Tensor(native, shape)


def pack_dims(t: Tensor, dims, packed: Shape):
    names = list(dims.names if isinstance(dims, Shape) else dims)
    keep = [n for n in t.shape.names if n not in names]
    native = t.native(names + keep)
    size = int(np.prod([t.shape.get_size(n) for n in names])) if names else 1
    native = native.reshape((size,) + tuple(native.shape[len(names):]))
    rest = t.shape.without(names)
    return Tensor(native, Shape((packed.names[0],) + rest.names, (size,) + rest.sizes, (packed.types[0],) + rest.types, rest.item_names_))


def unpack_dim(t: Tensor, dim: str, unpacked: Shape):
    ax = t.shape.names.index(dim)
    native = t._native.reshape(tuple(t._native.shape[:ax]) + tuple(unpacked.sizes) + tuple(t._native.shape[ax + 1:]))
    names = t.shape.names[:ax] + unpacked.names + t.shape.names[ax + 1:]
    sizes = t.shape.sizes[:ax] + unpacked.sizes + t.shape.sizes[ax + 1:]
    types = t.shape.types[:ax] + unpacked.types + t.shape.types[ax + 1:]
    return Tensor(native, Shape(names, sizes, types, t.shape.item_names_))


def stack(values, dim: Shape, **_):
    if isinstance(values, dict):
        keys, values = tuple(values.keys()), list(values.values())
    else:
        keys = dim.item_names_.get(dim.names[0])
    natives = [v._native for v in values]
    sizes = [tuple(n.shape) for n in natives]
    if len(set(sizes)) > 1:        # staggered components: different sizes per component -> object holding a list (non-uniform tensor)
        return _Stacked(values, dim.names[0], keys)
    native = torch.stack(natives, dim=-1)
    s0 = values[0].shape
    return Tensor(native, Shape(s0.names + (dim.names[0],), s0.sizes + (len(values),), s0.types + (dim.types[0],), {**s0.item_names_, dim.names[0]: keys}))


class _Stacked(Tensor):
    """ non-uniform stack along one dim (the components of a StaggeredGrid have different shapes) """
    def __init__(self, values, name, keys):
        self._values, self._name, self._keys = list(values), name, tuple(keys)
        s0 = values[0].shape
        sizes = tuple(sz if all(v.shape.sizes[i] == sz for v in values) else None for i, sz in enumerate(s0.sizes))
        self.shape = Shape(s0.names + (name,), sizes + (len(values),), s0.types + (DUAL if name.startswith('~') else CHANNEL,), {name: self._keys})
        self._native = None

    def __getitem__(self, item):
        (name, idx), = item.items()
        name = name if name == self._name else '~' + name
        assert name == self._name
        return self._values[self._keys.index(idx) if isinstance(idx, str) else idx]

    def __getattr__(self, name):
        if name in ('vector',) and self.__dict__.get('_name', '').lstrip('~') == 'vector': return _DimIndexer(self, self._name)
        raise AttributeError(name)


def unstack(value, dim):
    if isinstance(value, Tensor):
        name = dim.names[0] if isinstance(dim, Shape) else dim
        return [value[{name: i}] for i in range(value.shape.get_size(name))]
    return value.unstack(dim)


class Solve:
    def __init__(self, method='auto', rel_tol=None, abs_tol=None, x0=None, max_iterations=1000, suppress=(), preprocess_y=None,
                 preprocess_y_args=(), rank_deficiency=None, gradient_solve=None):
        self.method, self.rel_tol, self.abs_tol, self.x0, self.max_iterations = method, rel_tol, abs_tol, x0, max_iterations
        self.suppress, self.preprocess_y, self.preprocess_y_args, self.rank_deficiency = tuple(suppress), preprocess_y, preprocess_y_args, rank_deficiency

    def with_preprocessing(self, pre_y, *args):
        """ [PHIML-RECALL] Solve.with_preprocessing (call site phi/physics/fluid.py:146) """
        return copy_with(self, preprocess_y=pre_y, preprocess_y_args=args)


def copy_with(obj, **updates):
    """ [PHIML-RECALL] phiml.math.copy_with (call sites phi/physics/fluid.py:148,151) """
    import copy
    new = copy.copy(obj)
    for k, v in updates.items():
        setattr(new, k, v)
    return new


def solve_linear(f, y, solve: Solve, *f_args, **f_kwargs):
    """ TEST DOUBLE of the MATRIX branch of phiml.math.solve_linear [PHIML-RECALL, SURVEY Appendix B]: `f` is the operator PhiML would
    have traced into a sparse matrix (here: passed as a SciPy / torch sparse matrix, cells in C order), `y` a native (batch, *resolution).
    What PhiFlow's call site relies on (phi/physics/fluid.py:145-156): `solve.preprocess_y` is applied to y first, `solve.x0` is the start
    vector, and `Solve(rank_deficiency=1)` reaches the backend as `matrix_offset` -- a constant added to every matrix entry, i.e. the
    backend iterates on A + offset * 1 1^T (the value used here is arbitrary but of the sign that keeps the negative semi-definite
    Laplacian definite; the HIP backend must not depend on it). Returns the backend's SolveResult. """
    from ..backend import default_backend
    if solve.preprocess_y is not None:
        y = solve.preprocess_y(y, *solve.preprocess_y_args)
    be = default_backend()
    N = f.shape[0]
    yn = torch.as_tensor(y).reshape(-1, N)
    x0 = torch.zeros_like(yn) if solve.x0 is None else torch.as_tensor(solve.x0).reshape(-1, N)
    offset = None if not solve.rank_deficiency else -1.0 / N
    return be.linear_solve(solve.method, f, yn, x0, solve.rel_tol, solve.abs_tol, solve.max_iterations, None, offset)


class ConvergenceException(RuntimeError): pass
class NotConverged(ConvergenceException): pass
class Diverged(ConvergenceException): pass


from . import extrapolation   # noqa: F401,E402  (at the bottom: its singletons build tensors)
