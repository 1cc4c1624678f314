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
            ops += [dist.P2POp(dist.isend, self._plane(t, 0), self._global(self.lo_rank), self.group), dist.P2POp(dist.irecv, halo[0], self._global(self.lo_rank), self.group)]
        if self.hi_rank is not None:
            ops += [dist.P2POp(dist.isend, self._plane(t, 1), self._global(self.hi_rank), self.group), dist.P2POp(dist.irecv, halo[1], self._global(self.hi_rank), self.group)]
        if self.world == 2 and self.lo_rank == self.hi_rank and self.lo_rank is not None:
            # two ranks on a periodic axis: both messages go to the same peer; order them so that lo matches the peer's hi
            ops = ops if self.rank == 0 else [ops[2], ops[3], ops[0], ops[1]]
        pending = list(dist.batch_isend_irecv(ops)) if ops else []
        if reduce is not None:
            pending.append(dist.all_reduce(reduce, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for req in pending:
            req.wait()

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
