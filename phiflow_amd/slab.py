"""
Slab-decomposed pressure solve: ONE simulation split along x over the ranks of a `torch.distributed` group (SURVEY §8 f4 --
the reference has no domain decomposition; this is what lets a single 1024^3 solve use the 8 GPUs of a node).

Every rank owns `n_x / world` planes of x, rhs, r, d. Per CG iteration the ranks exchange
  * the two boundary planes of d_new after the MATVEC phase and of r after the UPDATE phase with their x-neighbours
    (point-to-point over xGMI with the "nccl" = RCCL backend; 256 KB per plane at 256^2 fp32), and
  * two scalars per batch entry (d.Ad and |r|^2) by all-reduce,
and run the same marching kernels as the single-GPU solver on their slab (`phihip_slab_*`, NB_HALO planes). The control
block (alpha, beta, convergence flags) stays on the device; the host only polls every `check_every` iterations.

Communication is enqueue-only: the boundary planes are packed by a device copy into send buffers that live as long as the solver (no
allocation inside the loop; for batch 1 the plane is contiguous and sent in place), the point-to-point exchange of a phase and that
phase's scalar all-reduce are issued TOGETHER (`async_op`) and waited for together -- two communication latencies per iteration instead
of four -- and with the "nccl" (= RCCL) backend a `wait()` is a stream dependency, not a host block. Not done: splitting the marching
kernels into interior / boundary planes so that the exchange hides behind the interior.
"""
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from . import _capi


def slab_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """ planes [begin, end) of `rank` (contiguous blocks, remainder to the first ranks) """
    base, extra = divmod(int(n), int(world))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def post_p2p(specs, group):
    """ Posts a batch of point-to-point messages, `specs` = [("send" | "recv", tensor, global peer rank), ...] in matching order, and returns
    `wait()`. With the "nccl" (= RCCL) backend the device tensors travel as they are and `wait()` is a stream dependency. The "gloo" backend
    has no device-tensor send / recv: device tensors are staged through host copies there (send: copied out before posting; recv: copied
    back inside `wait()`). That path exists for tests -- two ranks sharing ONE GPU, which RCCL refuses ("duplicate GPU") -- and for
    hosts without RCCL; it blocks the host per message, so it is not a performance path. """
    if not specs:
        return lambda: None
    stage = dist.get_backend(group) == "gloo" and any(t.is_cuda for _, t, _ in specs)
    ops, back = [], []
    for kind, t, peer in specs:
        buf = t
        if stage and t.is_cuda:
            buf = t.detach().to("cpu") if kind == "send" else torch.empty(t.shape, dtype=t.dtype, device="cpu")
            if kind == "recv":
                back.append((t, buf))
        ops.append(dist.P2POp(dist.isend if kind == "send" else dist.irecv, buf, peer, group))
    pending = list(dist.batch_isend_irecv(ops))

    def wait():
        for req in pending:
            req.wait()
        for dev, host in back:
            dev.copy_(host)
    return wait


class SlabSolver:
    """ CG on the 7-point pressure operator for a 3-D grid decomposed into x-slabs. `res`, `lower`, `upper`, `bc` describe the
    GLOBAL grid (bc = velocity boundary codes per axis side like `phihip_grid.bc`). """

    def __init__(self, backend, res, lower, upper, bc, dtype=torch.float32, batch: int = 1, group=None):
        assert len(res) == 3, "slab decomposition is implemented for 3-D grids"
        self.be = backend
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.res, self.bc, self.dtype, self.batch = tuple(res), [tuple(p) for p in bc], dtype, batch
        self.begin, self.end = slab_range(res[0], self.rank, self.world)
        assert self.end > self.begin, "more ranks than planes"
        dx = (upper[0] - lower[0]) / res[0]
        periodic = self.bc[0][0] == _capi.BC_PERIODIC
        self.lo_rank = self.rank - 1 if self.rank > 0 else (self.world - 1 if periodic and self.world > 1 else None)
        self.hi_rank = self.rank + 1 if self.rank < self.world - 1 else (0 if periodic and self.world > 1 else None)
        self.halo = (self.lo_rank is not None, self.hi_rank is not None)
        local_bc = [list(p) for p in self.bc]
        if self.world > 1 and periodic:
            pass      # both sides are halos; the periodic code keeps the grid descriptor valid
        code = _capi.PHIHIP_F64 if dtype == torch.float64 else _capi.PHIHIP_F32
        self.grid = _capi.make_grid(3, code, batch, (self.end - self.begin, res[1], res[2]),
                                    (lower[0] + self.begin * dx, lower[1], lower[2]), (lower[0] + self.end * dx, upper[1], upper[2]), local_bc)
        shape = (batch, self.end - self.begin, res[1], res[2])
        plane = (batch, res[1], res[2])
        z = lambda s: backend.zeros(s, dtype)
        self.r, self.d = z(shape), [z(shape), z(shape)]
        self.halos = {name: [z(plane), z(plane)] for name in ("x", "r", "d0", "d1")}
        self.send = [z(plane), z(plane)]      # packed boundary planes (batch > 1: the plane of a slab is strided over the batch)
        self.sums2 = backend.zeros((2 * batch,), torch.float64)
        self.sum1 = backend.zeros((batch,), torch.float64)

    # --- communication ---
    def _plane(self, t: torch.Tensor, side: int) -> torch.Tensor:
        """ boundary plane of the slab as a contiguous buffer: in place for batch 1, packed into the persistent send buffer otherwise """
        view = t[:, 0 if side == 0 else -1]
        if view.is_contiguous():
            return view
        self.send[side].copy_(view)
        return self.send[side]

    def _exchange(self, t: torch.Tensor, halo: List[torch.Tensor], reduce: Optional[torch.Tensor] = None):
        """ boundary planes of `t` -> the neighbours' halo buffers, their boundary planes -> `halo`; `reduce`: per-entry sums of the same
        phase, all-reduced (SUM) concurrently. Everything is issued before anything is waited for. """
        if self.world == 1:
            return
        ops = []
        if self.lo_rank is not None:
            ops += [("send", self._plane(t, 0), self._global(self.lo_rank)), ("recv", halo[0], self._global(self.lo_rank))]
        if self.hi_rank is not None:
            ops += [("send", self._plane(t, 1), self._global(self.hi_rank)), ("recv", halo[1], self._global(self.hi_rank))]
        if self.world == 2 and self.lo_rank == self.hi_rank and self.lo_rank is not None:
            # two ranks on a periodic axis: both messages go to the same peer; order them so that lo matches the peer's hi
            ops = ops if self.rank == 0 else [ops[2], ops[3], ops[0], ops[1]]
        wait_planes = post_p2p(ops, self.group)
        red = dist.all_reduce(reduce, op=dist.ReduceOp.SUM, group=self.group, async_op=True) if reduce is not None else None
        wait_planes()
        if red is not None:
            red.wait()

    def _global(self, r):
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def _allreduce(self, t: torch.Tensor):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    @staticmethod
    def _p(t):
        return t.data_ptr() if t is not None else 0

    def _hp(self, name):
        h = self.halos[name]
        return (h[0].data_ptr() if self.halo[0] else 0, h[1].data_ptr() if self.halo[1] else 0)

    # --- the solve ---
    def solve(self, rhs: torch.Tensor, x: torch.Tensor, rel_tol=1e-5, abs_tol=0.0, max_iterations=1000, refresh_every=50, check_every=10,
              flags: Optional[torch.Tensor] = None):
        """ rhs, x: this rank's slabs (batch, planes, y, z); x holds x0 on entry and the solution on exit. Returns the list of
        `SolveInfo` (identical on every rank). """
        ctx, g, halo, s = self.be.ctx, self.grid, self.halo, self.be.stream()
        csolve = _capi.Solve(float(rel_tol), float(abs_tol), int(max_iterations), int(refresh_every), int(check_every), 0)
        fl = flags.data_ptr() if flags is not None else 0
        for d in self.d:
            d.zero_()
        for h in self.halos.values():
            h[0].zero_(); h[1].zero_()
        self._exchange(x, self.halos["x"])
        ctx.slab_residual(g, halo, fl, x.data_ptr(), self._hp("x"), rhs.data_ptr(), self.r.data_ptr(), self.sums2.data_ptr(), False, s)
        self._exchange(self.r, self.halos["r"], reduce=self.sums2)
        first, sums_in = True, self.sums2
        for k in range(1, int(max_iterations) + 1):
            d_old, d_new = self.d[(k - 1) & 1], self.d[k & 1]
            ho, hn = ("d0", "d1") if (k & 1) else ("d1", "d0")
            ctx.slab_matvec(g, halo, fl, first, sums_in.data_ptr(), self.r.data_ptr(), self._hp("r"), d_old.data_ptr(), self._hp(ho),
                            d_new.data_ptr(), self.sum1.data_ptr(), csolve, s)
            first = False
            self._exchange(d_new, self.halos[hn], reduce=self.sum1)          # planes of d_new + the d.Ad sums: one round of communication
            if refresh_every > 0 and k % refresh_every == 0:
                ctx.slab_update(g, halo, fl, self.sum1.data_ptr(), d_new.data_ptr(), (0, 0), x.data_ptr(), 0, 0, csolve, True, s)
                self._exchange(x, self.halos["x"])
                ctx.slab_residual(g, halo, fl, x.data_ptr(), self._hp("x"), rhs.data_ptr(), self.r.data_ptr(), self.sums2.data_ptr(), True, s)
                self._exchange(self.r, self.halos["r"], reduce=self.sums2)   # [0..batch) = global |r|^2
                sums_in = self.sums2
            else:
                rr = self.sums2[: self.batch]
                ctx.slab_update(g, halo, fl, self.sum1.data_ptr(), d_new.data_ptr(), self._hp(hn), x.data_ptr(), self.r.data_ptr(), rr.data_ptr(),
                                csolve, False, s)
                self._exchange(self.r, self.halos["r"], reduce=rr)           # planes of r + the |r|^2 sums: the second round
                sums_in = self.sums2
            if check_every > 0 and k % check_every == 0 and k < max_iterations:
                infos = ctx.slab_state(g, False, sums_in.data_ptr(), csolve, True, s)
                if not any(i.reserved for i in infos):
                    break
        return ctx.slab_state(g, first, sums_in.data_ptr(), csolve, False, s)


class SlabFluid:
    """ The whole fluid step -- advect.semi_lagrangian(v, v, dt), divergence, pressure solve, gradient subtraction (phi/physics/advect.py:
    156-179, phi/physics/fluid.py:94-162) -- for ONE 3-D simulation decomposed into x-slabs (no reference counterpart).

    Every rank stores the samples it OWNS: the cells [begin, end) of centred fields and of the y / z velocity components, and the x faces
    [face_begin, face_end) (face f = lower face of cell f; the last rank also owns the outer face of an OPEN upper side). Advection,
    divergence and gradient run the ordinary single-GPU kernels on the slab EXTENDED by `ghost` planes of the neighbours' samples on each cut
    side (one packed message per neighbour and operation); the cut sides of the extended grid are declared OPEN, which only shapes results
    inside the ghost zone -- those are discarded. The reach of the operators bounds what is exact: divergence and gradient need one plane,
    the advection |u| dt / dx <= ghost - 1 cells along x (back-trace + multilinear taps + the 4-point means of the other components).
    The pressure solve is `SlabSolver` on the owned cells. Obstacles (r3): `set_obstacles` -- masks rasterised per rank, ghost cells from
    the owner, flags packed on the extended grid; `apply_boundary_conditions` runs on the extended velocity after the advection. """

    def __init__(self, backend, res, lower, upper, bc, dtype=torch.float32, batch: int = 1, bc_val=None, ghost: int = 2, group=None,
                 obstacles=None, overlap: bool = False):
        """ obstacles: list of obstacle descriptions as `_capi.make_obstacles` takes them (GLOBAL coordinates; Box / Sphere, linear and angular
        velocity) -- phi/physics/fluid.py:130-137,212-240 on slabs: see `set_obstacles`. """
        assert len(res) == 3 and ghost >= 1
        self.be, self.group, self.dtype, self.batch, self.ghost = backend, group, dtype, int(batch), int(ghost)
        self.overlap = bool(overlap)      # advect(): ghost exchange in flight while the whole slab is advected, cut-side windows redone after it
        self.res, self.bc = tuple(int(r) for r in res), [tuple(int(c) for c in p) for p in bc]
        self.solver = SlabSolver(backend, res, lower, upper, bc, dtype, batch, group)
        s = self.solver
        self.world, self.rank, self.begin, self.end = s.world, s.rank, s.begin, s.end
        self.lo_rank, self.hi_rank = s.lo_rank, s.hi_rank
        n0 = self.res[0]
        assert self.world == 1 or min(slab_range(n0, r, self.world)[1] - slab_range(n0, r, self.world)[0] for r in range(self.world)) >= ghost + 1, \
            "every slab needs at least ghost + 1 planes"
        self.gl = ghost if self.lo_rank is not None else 0
        self.gr = ghost if self.hi_rank is not None else 0
        # owned x faces (global face numbers): face 0 is stored unless the lower side is CLOSED, face n0 only on an OPEN upper side
        self.face_begin = self.begin if (self.begin > 0 or self.bc[0][0] != _capi.BC_CLOSED) else 1
        self.face_end = self.end + (1 if (self.end == n0 and self.bc[0][1] == _capi.BC_OPEN) else 0)
        dx = (upper[0] - lower[0]) / n0
        local_bc = [list(p) for p in self.bc]
        if self.lo_rank is not None:
            local_bc[0][0] = _capi.BC_OPEN
        if self.hi_rank is not None:
            local_bc[0][1] = _capi.BC_OPEN
        if self.world > 1 and self.bc[0][0] == _capi.BC_PERIODIC:
            local_bc[0] = [_capi.BC_OPEN, _capi.BC_OPEN]
        code = _capi.PHIHIP_F64 if dtype == torch.float64 else _capi.PHIHIP_F32
        self.ext_cells = (self.end + self.gr) - (self.begin - self.gl)
        self.grid = _capi.make_grid(3, code, batch, (self.ext_cells, self.res[1], self.res[2]),
                                    (lower[0] + (self.begin - self.gl) * dx, lower[1], lower[2]),
                                    (lower[0] + (self.end + self.gr) * dx, upper[1], upper[2]), local_bc, bc_val)
        self.ext_shape = [(batch,) + tuple(backend.ctx.component_shape(self.grid, c)) for c in range(3)]
        off_lo = 1 if local_bc[0][0] == _capi.BC_CLOSED else 0
        # extended array of component c: planes [own0[c], own0[c] + own_n[c]) are this rank's own samples, lo_n / hi_n ghost planes around them
        g0 = self.begin - self.gl + off_lo                    # global number of the first stored x face of the extended grid
        self.own0 = [self.face_begin - g0, self.gl, self.gl]
        self.own_n = [self.face_end - self.face_begin, self.end - self.begin, self.end - self.begin]
        self.lo_n = list(self.own0)
        self.hi_n = [self.ext_shape[c][1] - self.own0[c] - self.own_n[c] for c in range(3)]
        assert self.lo_n[0] == self.gl and self.hi_n[1] == self.gr and self.hi_n[2] == self.gr
        assert self.hi_n[0] == (self.gr + 1 if self.hi_rank is not None else 0), (self.hi_n, self.gr)
        self.own_shape = [(batch, self.own_n[c]) + self.ext_shape[c][2:] for c in range(3)]
        self.cell_shape = (batch, self.end - self.begin, self.res[1], self.res[2])
        self._own_lower0 = lower[0] + self.begin * dx
        self._own_upper0 = lower[0] + self.end * dx
        self._bounds = (tuple(lower), tuple(upper))
        self._bc_val = bc_val
        self.obstacles, self.n_obstacles, self.flags_ext, self.flags_own, self.active_count = None, 0, None, None, None
        if obstacles:
            self.set_obstacles(obstacles)

    # --- obstacles (fluid.py:130-137: accessible = ~union(geometries), hard_bcs = stagger(accessible, min), active = accessible) ---
    def set_obstacles(self, items):
        """ Rasterises the obstacles on this rank's OWN cells (`phihip_obstacle_accessible` on the slab's box), obtains the neighbours' mask
        for the ghost cells by the same packed exchange as every other cell field -- a ghost cell of a periodic axis lies outside the
        domain box, its geometry test would be wrong, its owner's is not --, and packs the stencil flags on the extended grid
        (`phihip_build_cellflags`): bits of own cells see the true accessibility of cells across the cut. The solver takes the own-cell
        crop, divergence / gradient the extended array. Call again when obstacles move. """
        be, ctx = self.be, self.be.ctx
        self.obstacles, self.n_obstacles = _capi.make_obstacles(items), len(items)
        lower, upper = self._bounds
        code = _capi.PHIHIP_F64 if self.dtype == torch.float64 else _capi.PHIHIP_F32
        own_grid = _capi.make_grid(3, code, 1, (self.end - self.begin, self.res[1], self.res[2]), (self._own_lower0, lower[1], lower[2]),
                                   (self._own_upper0, upper[1], upper[2]), self.bc, self._bc_val)
        acc_own = be.empty((self.end - self.begin, self.res[1], self.res[2]), torch.uint8)
        ctx.obstacle_accessible(own_grid, self.obstacles, self.n_obstacles, acc_own.data_ptr(), be.stream())
        acc_f = acc_own.to(self.dtype).unsqueeze(0).expand(self.batch, *acc_own.shape).contiguous()
        acc_ext = (self._extend_cells(acc_f)[0] > 0.5).to(torch.uint8).contiguous()
        ext_grid1 = _capi.make_grid(3, code, 1, (self.ext_cells, self.res[1], self.res[2]), tuple(self.grid.lower[d] for d in range(3)),
                                    tuple(self.grid.upper[d] for d in range(3)), [tuple(self.grid.bc[d][s] for s in range(2)) for d in range(3)])
        self.flags_ext = be.empty((self.ext_cells, self.res[1], self.res[2]), torch.uint8)
        ctx.build_cellflags(ext_grid1, acc_ext.data_ptr(), 0, 1, self.flags_ext.data_ptr(), be.stream())
        self.flags_own = self.flags_ext[self.gl: self.gl + (self.end - self.begin)].contiguous()
        count = acc_own.sum(dtype=torch.float64).reshape(1)
        if self.world > 1:
            dist.all_reduce(count, op=dist.ReduceOp.SUM, group=self.group)
        self.active_count = count            # active cells of the WHOLE domain (for _balance_divergence)

    # --- ghost exchange: one packed message per neighbour ---
    def _pack(self, parts: List[torch.Tensor]) -> torch.Tensor:
        return torch.cat([p.reshape(-1) for p in parts]) if parts else self.be.zeros((0,), self.dtype)

    def _extend(self, own: List[torch.Tensor], lo_n: List[int], hi_n: List[int], shapes) -> List[torch.Tensor]:
        """ own[k]: (batch, planes_k, ...) -> arrays with lo_n[k] / hi_n[k] planes of the neighbours' adjacent samples around them. The
        neighbour sends its top lo_n[k] planes to the rank above and its bottom hi_n[k] planes to the rank below (the counts are the same
        on every rank with a neighbour on that side). """
        ext, finish = self._extend_begin(own, lo_n, hi_n, shapes)
        finish()
        return ext

    def _extend_begin(self, own, lo_n, hi_n, shapes, zero_ghosts: bool = False):
        """ `_extend` in two halves: posts the messages and returns (ext, finish) -- `ext` holds the own samples (ghost planes: unset, or zero
        with `zero_ghosts`), `finish()` waits for the neighbours' planes and stores them. With the "nccl" (= RCCL) backend the wait is a stream
        dependency, so kernels enqueued between the two halves run while the planes cross xGMI. """
        ext = [self.be.empty(shapes[k], self.dtype) for k in range(len(own))]
        for k, t in enumerate(own):
            ext[k][:, lo_n[k]: lo_n[k] + t.shape[1]] = t
            if zero_ghosts:
                ext[k][:, : lo_n[k]] = 0
                ext[k][:, lo_n[k] + t.shape[1]:] = 0
        if self.world == 1:
            return ext, (lambda: None)
        # what goes UP is what the rank above lacks below its own samples (ghost planes of every field), what goes DOWN what the rank below
        # lacks above them (ghost cell planes, ghost + 1 x faces: the upper face of its last ghost cell included)
        up_counts = [self.ghost] * len(own)
        down_counts = [self.ghost + 1, self.ghost, self.ghost] if len(own) == 3 else [self.ghost] * len(own)
        send_up = self._pack([t[:, t.shape[1] - up_counts[k]:] for k, t in enumerate(own)]) if self.hi_rank is not None else None
        send_down = self._pack([t[:, : down_counts[k]] for k, t in enumerate(own)]) if self.lo_rank is not None else None
        plane = [self.batch * int(shapes[k][2]) * int(shapes[k][3]) for k in range(len(own))]
        recv_lo = self.be.empty((sum(lo_n[k] * plane[k] for k in range(len(own))),), self.dtype) if self.lo_rank is not None else None
        recv_hi = self.be.empty((sum(hi_n[k] * plane[k] for k in range(len(own))),), self.dtype) if self.hi_rank is not None else None
        ops = []
        gr = self.solver._global
        if self.lo_rank is not None:
            ops += [("send", send_down, gr(self.lo_rank)), ("recv", recv_lo, gr(self.lo_rank))]
        if self.hi_rank is not None:
            ops += [("send", send_up, gr(self.hi_rank)), ("recv", recv_hi, gr(self.hi_rank))]
        if self.world == 2 and self.lo_rank == self.hi_rank and self.lo_rank is not None and self.rank == 1:
            ops = [ops[2], ops[3], ops[0], ops[1]]        # two ranks on a periodic axis: match my "up" with the peer's "down"
        wait_planes = post_p2p(ops, self.group)

        def finish():
            wait_planes()
            for side, buf, counts in ((0, recv_lo, lo_n), (1, recv_hi, hi_n)):
                if buf is None:
                    continue
                pos = 0
                for k in range(len(own)):
                    n = counts[k] * plane[k]
                    if n:
                        block = buf[pos: pos + n].reshape((self.batch, counts[k]) + tuple(shapes[k][2:]))
                        if side == 0:
                            ext[k][:, : counts[k]] = block
                        else:
                            ext[k][:, shapes[k][1] - counts[k]:] = block
                    pos += n
            _keep = (send_up, send_down)      # the send buffers live until the requests have completed
        return ext, finish

    def _overlap_windows(self):
        """ Windows of the extended slab that `advect(overlap)` recomputes once the ghost planes have arrived: per cut side (window grid,
        [plane range of the extended array per component], [planes to take from the window per component]). The whole-slab pass that ran
        on empty ghosts is exact except within reach = ghost + 1 planes of a ghost zone; a window holds the ghosts, those planes and another
        reach of own planes behind them. None: the slab is too thin for two disjoint windows (the plain order is used). """
        if hasattr(self, "_windows"):
            return self._windows
        reach = self.ghost + 1
        own_cells = self.end - self.begin
        self._windows = None
        if own_cells < 4 * reach + 2:
            return None
        code = _capi.PHIHIP_F64 if self.dtype == torch.float64 else _capi.PHIHIP_F32
        dx = (self.grid.upper[0] - self.grid.lower[0]) / self.ext_cells
        bc = [[self.grid.bc[d][s] for s in range(2)] for d in range(3)]
        faces = self.ext_shape[0][1]
        wins = []
        if self.lo_rank is not None:
            wc = self.gl + 2 * reach                   # cells [0, wc) of the extended grid; its lower side is the cut (OPEN): stored face 0 = face 0
            wbc = [list(p) for p in bc]; wbc[0] = [bc[0][0], _capi.BC_OPEN]
            lo0 = self.grid.lower[0]
            g = _capi.make_grid(3, code, self.batch, (wc, self.res[1], self.res[2]), (lo0, self.grid.lower[1], self.grid.lower[2]),
                                (lo0 + wc * dx, self.grid.upper[1], self.grid.upper[2]), wbc, self._bc_val)
            src = [(0, wc + 1), (0, wc), (0, wc)]
            take = [(self.own0[c], self.own0[c] + reach) for c in range(3)]
            wins.append((g, src, take))
        if self.hi_rank is not None:
            wc = self.gr + 2 * reach                   # cells [ext_cells - wc, ext_cells); both sides OPEN: wc + 1 faces, the last stored ones
            wbc = [list(p) for p in bc]; wbc[0] = [_capi.BC_OPEN, bc[0][1]]
            hi0 = self.grid.upper[0]
            g = _capi.make_grid(3, code, self.batch, (wc, self.res[1], self.res[2]), (hi0 - wc * dx, self.grid.lower[1], self.grid.lower[2]),
                                (hi0, self.grid.upper[1], self.grid.upper[2]), wbc, self._bc_val)
            src = [(faces - (wc + 1), faces), (self.ext_cells - wc, self.ext_cells), (self.ext_cells - wc, self.ext_cells)]
            take = [(self.own0[c] + self.own_n[c] - reach, self.own0[c] + self.own_n[c]) for c in range(3)]
            wins.append((g, src, take))
        self._windows = wins
        return wins

    def _extend_velocity(self, v: List[torch.Tensor]) -> List[torch.Tensor]:
        return self._extend(v, self.lo_n, self.hi_n, self.ext_shape)

    def _extend_cells(self, p: torch.Tensor) -> torch.Tensor:
        shape = (self.batch, self.ext_cells, self.res[1], self.res[2])
        return self._extend([p], [self.gl], [self.gr], [shape])[0]

    def _own_velocity(self, ext: List[torch.Tensor]) -> List[torch.Tensor]:
        return [ext[c][:, self.own0[c]: self.own0[c] + self.own_n[c]].contiguous() for c in range(3)]

    # --- the operators ---
    def advect(self, v: List[torch.Tensor], dt: float, check_cfl: bool = True) -> List[torch.Tensor]:
        """ semi-Lagrangian self-advection of the owned velocity samples. Exact while |u_x| dt / dx <= ghost - 1: a back-trace that leaves the
        ghost zone would be clamped by the (fake) OPEN cut side and the result would depend on the number of ranks. `check_cfl` (default)
        verifies the bound on this rank's extended x component -- its own lookups and the 4-point means they use only see those samples, so
        the check needs no collective -- and raises `ValueError` instead of returning rank-dependent values; it costs one small reduction
        and a host read per call (the step's pressure solve synchronises anyway). Construct with a larger `ghost` for faster flows. """
        P = lambda ts: [t.data_ptr() for t in ts]
        windows = self._overlap_windows() if (self.overlap and self.world > 1) else None
        if windows is None:
            ext = self._extend_velocity(v)
        else:
            # the exchange is in flight while the WHOLE extended slab is advected with empty (zero) ghost planes: exact for every sample whose
            # reach -- back-trace, taps, 4-point means: ghost + 1 planes -- stays inside the own samples
            ext, finish = self._extend_begin(v, self.lo_n, self.hi_n, self.ext_shape, zero_ghosts=True)
            out = [torch.empty_like(t) for t in ext]
            self.be.ctx.advect_staggered(self.grid, P(ext), P(ext), P(out), float(dt), self.be.stream())
            finish()
        if check_cfl and self.world > 1:
            dx0 = (self.grid.upper[0] - self.grid.lower[0]) / self.ext_cells
            cfl = float(ext[0].abs().max()) * abs(float(dt)) / dx0
            if not cfl <= self.ghost - 1:
                raise ValueError(f"SlabFluid.advect: |u_x| dt / dx = {cfl:.3f} exceeds ghost - 1 = {self.ghost - 1} on rank {self.rank}: back-traces "
                                 f"would leave the exchanged ghost planes. Use SlabFluid(..., ghost={int(cfl) + 2}) or a smaller dt.")
        if windows is None:
            out = [torch.empty_like(t) for t in ext]
            self.be.ctx.advect_staggered(self.grid, P(ext), P(ext), P(out), float(dt), self.be.stream())
        else:
            # ... and the few planes next to each cut are redone on a window of the completed arrays: the lookups are formed relative to
            # the sample (advect_common.hpp lookup_pairs_rel), so a window reproduces the bits of the whole-slab launch
            for wgrid, src, take in windows:
                win = [ext[c][:, src[c][0]: src[c][1]].contiguous() for c in range(3)]
                wout = [torch.empty_like(t) for t in win]
                self.be.ctx.advect_staggered(wgrid, P(win), P(win), P(wout), float(dt), self.be.stream())
                for c in range(3):
                    a, b = take[c]
                    out[c][:, a: b] = wout[c][:, a - src[c][0]: b - src[c][0]]
        if self.n_obstacles:     # fluid.apply_boundary_conditions (fluid.py:212-240): pointwise in physical coordinates, own samples are exact
            self.be.ctx.apply_obstacles(self.grid, self.obstacles, self.n_obstacles, P(out), self.be.stream())
        return self._own_velocity(out)

    def divergence(self, v: List[torch.Tensor], balance: bool = False) -> torch.Tensor:
        ext = self._extend_velocity(v)
        div = self.be.empty((self.batch, self.ext_cells, self.res[1], self.res[2]), self.dtype)
        fl = self.flags_ext.data_ptr() if self.flags_ext is not None else 0          # div * active (fluid.py:139-140)
        self.be.ctx.divergence(self.grid, [t.data_ptr() for t in ext], fl, 1, False, div.data_ptr(), self.be.stream())
        own = div[:, self.gl: self.gl + (self.end - self.begin)].contiguous()
        if balance:    # fluid._balance_divergence (fluid.py:205-209) over the WHOLE domain: div -= active * sum(div) / sum(active)
            total = own.sum(dim=(1, 2, 3), dtype=torch.float64)
            if self.world > 1:
                dist.all_reduce(total, op=dist.ReduceOp.SUM, group=self.group)
            if self.flags_own is None:
                own -= (total / (self.res[0] * self.res[1] * self.res[2])).to(self.dtype)[:, None, None, None]
            else:
                active = ((self.flags_own & 64) != 0).to(self.dtype)
                own -= active[None] * (total / self.active_count).to(self.dtype)[:, None, None, None]
        return own

    def grad_subtract(self, v: List[torch.Tensor], p: torch.Tensor) -> List[torch.Tensor]:
        ext_v, ext_p = self._extend_velocity(v), self._extend_cells(p)
        fl = self.flags_ext.data_ptr() if self.flags_ext is not None else 0          # grad p * hard_bcs (fluid.py:158-160)
        self.be.ctx.grad_subtract(self.grid, fl, 1, ext_p.data_ptr(), [t.data_ptr() for t in ext_v], self.be.stream())
        return self._own_velocity(ext_v)

    def step(self, v: List[torch.Tensor], p: torch.Tensor, dt: float, rel_tol=1e-5, abs_tol=0.0, max_iterations=1000, refresh_every=50,
             check_every=10, check_cfl: bool = True):
        """ one operator-split time step; `p` holds the pressure guess on entry and the pressure on exit. Returns (v, infos).
        The advection is exact while |u_x| dt / dx <= ghost - 1 (checked per rank, `ValueError` otherwise; see `advect`). """
        v = self.advect(v, dt, check_cfl)
        singular = all(c != _capi.BC_OPEN for pair in self.bc for c in pair)
        div = self.divergence(v, balance=singular)
        infos = self.solver.solve(div, p, rel_tol, abs_tol, max_iterations, refresh_every, check_every, flags=self.flags_own)
        return self.grad_subtract(v, p), infos
