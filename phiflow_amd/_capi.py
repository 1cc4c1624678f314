"""
ctypes binding of ``libphihip.so`` (C ABI declared in ``include/phihip.h``).

This is the only place the package touches native code. There is **no CPU fallback**: if the shared library is
missing `load_default_library()` raises `PhiHipLibraryError`, and creating a context without a HIP device fails with
``PHIHIP_ERR_NO_DEVICE``.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_size_t, c_uint8, c_void_p
from typing import Optional, Sequence

PHIHIP_F32, PHIHIP_F64 = 0, 1
BC_PERIODIC, BC_CLOSED, BC_OPEN = 0, 1, 2

K_NAMES = ("advect", "divergence", "cg_residual", "cg_matvec_dot", "cg_update", "cg_scalar", "grad_subtract", "other", "cg_update_r")
K_COUNT = len(K_NAMES)

STATUS_NAMES = {0: "PHIHIP_OK", -1: "PHIHIP_ERR_BAD_ARG", -2: "PHIHIP_ERR_HIP", -3: "PHIHIP_ERR_UNSUPPORTED",
                -4: "PHIHIP_ERR_NO_DEVICE", -5: "PHIHIP_ERR_ALLOC"}

EXPORTED_SYMBOLS = (
    "phihip_version", "phihip_build_id", "phihip_last_error", "phihip_ctx_create", "phihip_ctx_destroy", "phihip_workspace_bytes",
    "phihip_component_shape", "phihip_advect_staggered", "phihip_advect_centered", "phihip_build_cellflags",
    "phihip_divergence", "phihip_laplace_apply", "phihip_cg_solve", "phihip_solve_residuals", "phihip_solve_relative_residual", "phihip_grad_subtract",
    "phihip_make_incompressible", "phihip_diffuse_explicit", "phihip_profile_enable", "phihip_profile_read",
    "phihip_set_tuning", "phihip_mac_cormack_staggered", "phihip_mac_cormack_centered", "phihip_centered_to_staggered",
    "phihip_set_tuning_kernel", "phihip_query_plan", "phihip_obstacle_accessible", "phihip_apply_obstacles",
    "phihip_advect_staggered_backward", "phihip_advect_centered_backward", "phihip_centered_to_staggered_backward",
    "phihip_make_incompressible_backward", "phihip_mac_cormack_staggered_backward", "phihip_mac_cormack_centered_backward",
    "phihip_diffuse_explicit_backward", "phihip_diffuse_explicit_centered", "phihip_diffuse_implicit", "phihip_diffuse_implicit_centered", "phihip_cg_solve_shifted",
    "phihip_slab_residual", "phihip_slab_matvec", "phihip_slab_update", "phihip_slab_state", "phihip_set_small_grid_solver",
    "phihip_grid_sample", "phihip_grid_sample_backward", "phihip_set_deferred_x_update", "phihip_set_advect_halo", "phihip_advect_fallback_stats", "phihip_set_advect_chunk", "phihip_set_advect_windows_2d", "phihip_query_advect_chunk", "phihip_set_autotune", "phihip_allreduce_residual", "phihip_set_single_reduction_cg", "phihip_set_resident_cg", "phihip_set_advect_dma", "phihip_workspace_placement",
)


class PhiHipError(RuntimeError):
    """ A libphihip call returned a negative status. """

    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class PhiHipLibraryError(ImportError):
    """ libphihip.so could not be loaded (not built, or ROCm runtime missing). """


class Grid(ctypes.Structure):
    """ mirrors ``phihip_grid`` """
    _fields_ = [("rank", c_int32), ("dtype", c_int32), ("batch", c_int32), ("res", c_int32 * 3),
                ("lower", c_double * 3), ("upper", c_double * 3), ("bc", (c_int32 * 2) * 3),
                ("bc_val", ((c_double * 3) * 2) * 3)]


class ObstacleStruct(ctypes.Structure):
    """ phihip_obstacle """
    _fields_ = [("kind", c_int32), ("group", c_int32), ("embed_mask", c_int32), ("reserved", c_int32), ("center", c_double * 3), ("half_size", c_double * 3),
                ("velocity", c_double * 3), ("angular_velocity", c_double * 3), ("rotation", c_double * 9)]


OBSTACLE_BOX, OBSTACLE_SPHERE = 0, 1
DIV_BALANCE, DIV_FINITE_GUARD = 1, 4      # bits of the `balance` argument (include/phihip.h PHIHIP_DIV_*)


def make_obstacles(items) -> "ctypes.Array":
    """ items: sequence of dicts(kind, center, half_size, velocity, angular_velocity[, rotation, group]) -> ctypes array of phihip_obstacle """
    arr = (ObstacleStruct * max(1, len(items)))()
    for k, it in enumerate(items):
        arr[k].kind = int(it["kind"])
        arr[k].group = int(it.get("group", 0))
        arr[k].embed_mask = int(it.get("embed_mask", 0))
        for d, val in enumerate(it["center"]):
            arr[k].center[d] = float(val)
        for d, val in enumerate(it["half_size"]):
            arr[k].half_size[d] = float(val)
        for d, val in enumerate(it.get("velocity", ())):
            arr[k].velocity[d] = float(val)
        for d, val in enumerate(it.get("angular_velocity", ())):
            arr[k].angular_velocity[d] = float(val)
        rot = it.get("rotation")
        if rot is not None:   # (rank x rank) matrix, box frame -> world
            for a, row in enumerate(rot):
                for c, val in enumerate(row):
                    arr[k].rotation[3 * a + c] = float(val)
    return arr


class Solve(ctypes.Structure):
    """ mirrors ``phihip_solve`` """
    _fields_ = [("rel_tol", c_double), ("abs_tol", c_double), ("max_iterations", c_int32), ("refresh_every", c_int32),
                ("check_every", c_int32), ("method", c_int32)]


class SolveInfo(ctypes.Structure):
    """ mirrors ``phihip_solve_info`` """
    _fields_ = [("residual_sq", c_double), ("rhs_sq", c_double), ("iterations", c_int32), ("converged", c_int32),
                ("diverged", c_int32), ("reserved", c_int32)]


_Ptr3 = c_void_p * 3


def ptr3(ptrs: Optional[Sequence[int]]):
    """ array of three device pointers (unused trailing entries NULL) or None """
    if ptrs is None:
        return None
    arr = _Ptr3()
    for i in range(3):
        arr[i] = ptrs[i] if i < len(ptrs) and ptrs[i] else None
    return arr


def make_grid(rank: int, dtype: int, batch: int, res, lower, upper, bc, bc_val=None) -> Grid:
    g = Grid()
    g.rank, g.dtype, g.batch = int(rank), int(dtype), int(batch)
    for d in range(rank):
        g.res[d] = int(res[d])
        g.lower[d] = float(lower[d])
        g.upper[d] = float(upper[d])
        for s in range(2):
            g.bc[d][s] = int(bc[d][s])
            for c in range(rank):
                g.bc_val[d][s][c] = float(bc_val[d][s][c]) if bc_val is not None else 0.0
    return g


class Library:
    """ typed function table of one loaded libphihip """

    def __init__(self, path: str, strict: bool = True):
        """ strict=False (benchmark A/B of an older build only): symbols missing from the library are skipped """
        self.path = os.path.abspath(path)
        # PyTorch-ROCm ships its own libamdhip64: whichever HIP runtime is loaded first owns the devices, a second copy sees "No HIP
        # GPUs". When torch is installed (it provides the device tensors for the Python layer), let it load its runtime first so that
        # libphihip binds to the same one. C / C++ callers of the ABI are not affected.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        try:
            self.dll = ctypes.CDLL(self.path)
        except OSError as exc:
            raise PhiHipLibraryError(f"cannot load {self.path}: {exc}. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                     f"(hipcc --offload-arch=gfx950); there is no CPU fallback.") from exc
        missing = [s for s in EXPORTED_SYMBOLS if not hasattr(self.dll, s)]
        if missing and not strict:
            class _Skip:
                argtypes = None
                restype = None
            for name in missing:
                setattr(self.dll, name, _Skip())
            missing = []
        if missing:
            raise PhiHipLibraryError(f"{self.path} lacks symbols declared in include/phihip.h: {missing}")
        d = self.dll
        d.phihip_version.restype = c_int
        d.phihip_build_id.restype = c_char_p
        d.phihip_last_error.restype = c_char_p
        d.phihip_ctx_create.argtypes = [c_int, POINTER(c_void_p)]
        d.phihip_ctx_destroy.argtypes = [c_void_p]
        d.phihip_workspace_bytes.argtypes = [c_void_p, POINTER(c_size_t)]
        d.phihip_component_shape.argtypes = [POINTER(Grid), c_int, POINTER(c_int32 * 3)]
        d.phihip_advect_staggered.argtypes = [c_void_p, POINTER(Grid), POINTER(_Ptr3), POINTER(_Ptr3), POINTER(_Ptr3), c_double, c_void_p]
        d.phihip_advect_centered.argtypes = [c_void_p, POINTER(Grid), c_void_p, POINTER((c_int32 * 2) * 3), POINTER((c_double * 2) * 3),
                                             POINTER(_Ptr3), c_void_p, c_double, c_void_p]
        if hasattr(d, "phihip_grid_sample"):
            d.phihip_grid_sample.argtypes = [c_void_p, POINTER(Grid), c_void_p, c_int, POINTER(_Ptr3), ctypes.c_int64, c_void_p, c_void_p, c_void_p,
                                             c_void_p]
            d.phihip_grid_sample_backward.argtypes = [c_void_p, POINTER(Grid), c_void_p, c_int, POINTER(_Ptr3), ctypes.c_int64, c_void_p, c_void_p,
                                                      POINTER(_Ptr3), c_void_p]
        d.phihip_mac_cormack_staggered.argtypes = [c_void_p, POINTER(Grid), POINTER(_Ptr3), POINTER(_Ptr3), POINTER(_Ptr3), c_double, c_double,
                                                   c_void_p]
        d.phihip_mac_cormack_centered.argtypes = [c_void_p, POINTER(Grid), c_void_p, POINTER((c_int32 * 2) * 3), POINTER((c_double * 2) * 3),
                                                  POINTER(_Ptr3), c_void_p, c_double, c_double, c_void_p]
        d.phihip_centered_to_staggered.argtypes = [c_void_p, POINTER(Grid), c_void_p, POINTER((c_int32 * 2) * 3), POINTER((c_double * 2) * 3),
                                                   POINTER(c_double * 3), c_int, POINTER(_Ptr3), c_void_p]
        d.phihip_obstacle_accessible.argtypes = [c_void_p, POINTER(Grid), POINTER(ObstacleStruct), c_int, c_void_p, c_void_p]
        d.phihip_apply_obstacles.argtypes = [c_void_p, POINTER(Grid), POINTER(ObstacleStruct), c_int, POINTER(_Ptr3), c_void_p]
        d.phihip_advect_staggered_backward.argtypes = [c_void_p, POINTER(Grid), POINTER(_Ptr3), POINTER(_Ptr3), POINTER(_Ptr3), c_double,
                                                       POINTER(_Ptr3), POINTER(_Ptr3), c_void_p]
        d.phihip_advect_centered_backward.argtypes = [c_void_p, POINTER(Grid), c_void_p, POINTER((c_int32 * 2) * 3), POINTER((c_double * 2) * 3),
                                                      POINTER(_Ptr3), c_void_p, c_double, c_void_p, POINTER(_Ptr3), c_void_p]
        d.phihip_centered_to_staggered_backward.argtypes = [c_void_p, POINTER(Grid), POINTER((c_int32 * 2) * 3), POINTER(c_double * 3),
                                                            POINTER(_Ptr3), c_void_p, c_void_p]
        d.phihip_mac_cormack_staggered_backward.argtypes = [c_void_p, POINTER(Grid), POINTER(_Ptr3), POINTER(_Ptr3), POINTER(_Ptr3), c_double,
                                                            c_double, POINTER(_Ptr3), POINTER(_Ptr3), c_void_p]
        d.phihip_mac_cormack_centered_backward.argtypes = [c_void_p, POINTER(Grid), c_void_p, POINTER((c_int32 * 2) * 3),
                                                           POINTER((c_double * 2) * 3), POINTER(_Ptr3), c_void_p, c_double, c_double, c_void_p,
                                                           POINTER(_Ptr3), c_void_p]
        d.phihip_diffuse_explicit_backward.argtypes = [c_void_p, POINTER(Grid), POINTER(_Ptr3), POINTER(_Ptr3), c_double, c_void_p]
        d.phihip_diffuse_explicit_centered.argtypes = [c_void_p, POINTER(Grid), c_void_p, POINTER((c_int32 * 2) * 3), POINTER((c_double * 2) * 3),
                                                       c_void_p, c_double, c_int, c_void_p]
        d.phihip_slab_residual.argtypes = [c_void_p, POINTER(Grid), c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_int, c_void_p]
        d.phihip_slab_matvec.argtypes = [c_void_p, POINTER(Grid), c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, POINTER(Solve), c_void_p]
        d.phihip_slab_update.argtypes = [c_void_p, POINTER(Grid), c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_int, POINTER(Solve), c_void_p]
        d.phihip_slab_state.argtypes = [c_void_p, POINTER(Grid), c_int, c_void_p, POINTER(Solve), POINTER(SolveInfo), c_int, c_void_p]
        d.phihip_make_incompressible_backward.argtypes = [c_void_p, POINTER(Grid), c_void_p, c_int, c_int, POINTER(_Ptr3), c_void_p,
                                                          POINTER(Solve), POINTER(SolveInfo), c_void_p]
        d.phihip_build_cellflags.argtypes = [c_void_p, POINTER(Grid), c_void_p, c_void_p, c_int, c_void_p, c_void_p]
        d.phihip_divergence.argtypes = [c_void_p, POINTER(Grid), POINTER(_Ptr3), c_void_p, c_int, c_int, c_void_p, c_void_p]
        d.phihip_laplace_apply.argtypes = [c_void_p, POINTER(Grid), c_void_p, c_int, c_void_p, c_void_p, c_void_p]
        d.phihip_cg_solve.argtypes = [c_void_p, POINTER(Grid), c_void_p, c_int, c_void_p, c_void_p, POINTER(Solve), POINTER(SolveInfo), c_void_p]
        d.phihip_solve_residuals.argtypes = [c_void_p, c_int, c_void_p, c_void_p]
        d.phihip_solve_relative_residual.argtypes = [c_void_p, c_int, c_void_p, c_void_p]
        d.phihip_grad_subtract.argtypes = [c_void_p, POINTER(Grid), c_void_p, c_int, c_void_p, POINTER(_Ptr3), c_void_p]
        d.phihip_make_incompressible.argtypes = [c_void_p, POINTER(Grid), POINTER(_Ptr3), POINTER(_Ptr3), c_void_p, c_int, c_int, c_void_p,
                                                 c_void_p, POINTER(Solve), POINTER(SolveInfo), c_void_p]
        d.phihip_diffuse_explicit.argtypes = [c_void_p, POINTER(Grid), POINTER(_Ptr3), POINTER(_Ptr3), c_double, c_void_p]
        d.phihip_cg_solve_shifted.argtypes = [c_void_p, POINTER(Grid), c_double, c_double, c_void_p, c_void_p, POINTER(Solve), POINTER(SolveInfo), c_void_p]
        d.phihip_diffuse_implicit.argtypes = [c_void_p, POINTER(Grid), POINTER(_Ptr3), POINTER(_Ptr3), c_double, POINTER(Solve), POINTER(SolveInfo), c_void_p]
        d.phihip_diffuse_implicit_centered.argtypes = [c_void_p, POINTER(Grid), c_void_p, POINTER((c_int32 * 2) * 3), POINTER((c_double * 2) * 3), c_void_p,
                                                       c_double, POINTER(Solve), POINTER(SolveInfo), c_void_p]
        d.phihip_profile_enable.argtypes = [c_void_p, c_int]
        d.phihip_profile_read.argtypes = [c_void_p, POINTER(c_int32 * K_COUNT), POINTER(c_double * K_COUNT), c_int]
        d.phihip_set_tuning.argtypes = [c_void_p, c_int, c_int, c_int]
        d.phihip_set_tuning_kernel.argtypes = [c_void_p, c_int, c_int, c_int, c_int]
        d.phihip_set_small_grid_solver.argtypes = [c_void_p, c_int]
        if hasattr(d, "phihip_set_deferred_x_update"):
            d.phihip_set_deferred_x_update.argtypes = [c_void_p, c_int]
        d.phihip_set_advect_halo.argtypes = [c_void_p, c_int]
        d.phihip_set_advect_chunk.argtypes = [c_void_p, c_int]
        d.phihip_set_advect_windows_2d.argtypes = [c_void_p, c_int]
        d.phihip_query_advect_chunk.argtypes = [c_void_p, POINTER(c_int32)]
        d.phihip_workspace_placement.argtypes = [c_void_p, c_int, POINTER(c_int32), POINTER(c_double)]
        d.phihip_set_autotune.argtypes = [c_void_p, c_int]
        if hasattr(d, 'phihip_set_advect_dma'):
            d.phihip_set_advect_dma.argtypes = [c_void_p, c_int, POINTER(c_int32)]
        d.phihip_set_single_reduction_cg.argtypes = [c_void_p, c_int, ctypes.c_longlong]
        d.phihip_set_resident_cg.argtypes = [c_void_p, c_int, ctypes.c_longlong]
        d.phihip_allreduce_residual.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]
        d.phihip_advect_fallback_stats.argtypes = [c_void_p, POINTER(c_int32 * 2), c_void_p]
        d.phihip_query_plan.argtypes = [c_void_p, POINTER(Grid), c_int, c_int, POINTER(c_int32 * 6)]
        for name in EXPORTED_SYMBOLS:
            if name not in ("phihip_version", "phihip_last_error", "phihip_build_id"):
                getattr(d, name).restype = c_int

    def check(self, status: int):
        if status != 0:
            raise PhiHipError(status, (self.dll.phihip_last_error() or b"").decode(errors="replace"))

    def version(self) -> int:
        return self.dll.phihip_version()

    def build_id(self) -> str:
        """ "<git commit>[+dirty] src:<hash of the sources the library was compiled from>" (phihip_build_id) """
        fn = getattr(self.dll, "phihip_build_id", None)
        raw = fn() if fn is not None and fn.restype is c_char_p else None
        return (raw or b"unknown src:unknown").decode(errors="replace")

    def built_from_tree(self) -> bool:
        """ the library was compiled from the kernel sources that sit next to it now (stale-.so check: the .so is not in git) """
        return self.build_id().rsplit("src:", 1)[-1] == source_hash()


def source_hash() -> str:
    """ sha1 (16 hex digits) over csrc/*.hip, csrc/*.hpp in name order and include/phihip.h -- the Makefile embeds the same value """
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(here, "csrc")
    files = sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".hpp")))
    h = hashlib.sha1()
    for f in [os.path.join(csrc, f) for f in files] + [os.path.join(here, "..", "include", "phihip.h")]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


class Context:
    """ owns one ``phihip_ctx`` (device + workspace). Not thread-safe; one per (process, device). """

    def __init__(self, lib: Library, device: int = 0):
        self.lib = lib
        self.handle = c_void_p()
        lib.check(lib.dll.phihip_ctx_create(int(device), ctypes.byref(self.handle)))

    def close(self):
        if self.handle:
            self.lib.dll.phihip_ctx_destroy(self.handle)
            self.handle = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- thin typed wrappers (pointers are plain ints) ----
    def component_shape(self, grid: Grid, comp: int):
        out = (c_int32 * 3)()
        self.lib.check(self.lib.dll.phihip_component_shape(ctypes.byref(grid), comp, ctypes.byref(out)))
        return tuple(out[d] for d in range(grid.rank))

    def advect_staggered(self, grid, field, velocity, out, dt, stream=0):
        self.lib.check(self.lib.dll.phihip_advect_staggered(self.handle, ctypes.byref(grid), ctypes.byref(ptr3(field)),
                                                            ctypes.byref(ptr3(velocity)), ctypes.byref(ptr3(out)), float(dt), stream or None))

    @staticmethod
    def _scalar_bc(grid, s_bc, s_val):
        bc = ((c_int32 * 2) * 3)()
        val = ((c_double * 2) * 3)()
        for d in range(grid.rank):
            for side in range(2):
                bc[d][side] = int(s_bc[d][side])
                val[d][side] = float(s_val[d][side]) if s_val is not None else 0.0
        return bc, val

    def advect_centered(self, grid, s, s_bc, s_val, velocity, out, dt, stream=0):
        bc, val = self._scalar_bc(grid, s_bc, s_val)
        self.lib.check(self.lib.dll.phihip_advect_centered(self.handle, ctypes.byref(grid), s, ctypes.byref(bc), ctypes.byref(val),
                                                           ctypes.byref(ptr3(velocity)), out, float(dt), stream or None))

    def grid_sample(self, grid, values, values_batch, coords, points, out, out_min=0, out_max=0, stream=0):
        """ math.grid_sample: `grid` describes the VALUES array (res = its shape, bc / bc_val[..][0] = its extrapolation) """
        self.lib.check(self.lib.dll.phihip_grid_sample(self.handle, ctypes.byref(grid), values, int(values_batch), ctypes.byref(ptr3(coords)),
                                                       int(points), out or None, out_min or None, out_max or None, stream or None))

    def grid_sample_backward(self, grid, values, values_batch, coords, points, grad_out, grad_values, grad_coords, stream=0):
        gc = ptr3(grad_coords)
        self.lib.check(self.lib.dll.phihip_grid_sample_backward(self.handle, ctypes.byref(grid), values, int(values_batch), ctypes.byref(ptr3(coords)),
                                                                int(points), grad_out, grad_values or None, ctypes.byref(gc) if gc is not None else None,
                                                                stream or None))

    def mac_cormack_staggered(self, grid, field, velocity, out, dt, correction_strength=1.0, stream=0):
        self.lib.check(self.lib.dll.phihip_mac_cormack_staggered(self.handle, ctypes.byref(grid), ctypes.byref(ptr3(field)),
                                                                 ctypes.byref(ptr3(velocity)), ctypes.byref(ptr3(out)), float(dt),
                                                                 float(correction_strength), stream or None))

    def mac_cormack_centered(self, grid, s, s_bc, s_val, velocity, out, dt, correction_strength=1.0, stream=0):
        bc, val = self._scalar_bc(grid, s_bc, s_val)
        self.lib.check(self.lib.dll.phihip_mac_cormack_centered(self.handle, ctypes.byref(grid), s, ctypes.byref(bc), ctypes.byref(val),
                                                                ctypes.byref(ptr3(velocity)), out, float(dt), float(correction_strength),
                                                                stream or None))

    def centered_to_staggered(self, grid, s, s_bc, s_val, vector, accumulate, out, stream=0):
        bc, val = self._scalar_bc(grid, s_bc, s_val)
        vec = (c_double * 3)(*([float(x) for x in vector] + [0.0] * (3 - len(vector))))
        self.lib.check(self.lib.dll.phihip_centered_to_staggered(self.handle, ctypes.byref(grid), s, ctypes.byref(bc), ctypes.byref(val),
                                                                 ctypes.byref(vec), int(bool(accumulate)), ctypes.byref(ptr3(out)),
                                                                 stream or None))

    def advect_staggered_backward(self, grid, field, velocity, grad_out, dt, grad_field, grad_velocity, stream=0):
        gf, gv = ptr3(grad_field), ptr3(grad_velocity)
        self.lib.check(self.lib.dll.phihip_advect_staggered_backward(
            self.handle, ctypes.byref(grid), ctypes.byref(ptr3(field)), ctypes.byref(ptr3(velocity)), ctypes.byref(ptr3(grad_out)), float(dt),
            ctypes.byref(gf) if gf is not None else None, ctypes.byref(gv) if gv is not None else None, stream or None))

    def advect_centered_backward(self, grid, s, s_bc, s_val, velocity, grad_out, dt, grad_s, grad_velocity, stream=0):
        bc, val = self._scalar_bc(grid, s_bc, s_val)
        gv = ptr3(grad_velocity)
        self.lib.check(self.lib.dll.phihip_advect_centered_backward(
            self.handle, ctypes.byref(grid), s, ctypes.byref(bc), ctypes.byref(val), ctypes.byref(ptr3(velocity)), grad_out, float(dt),
            grad_s or None, ctypes.byref(gv) if gv is not None else None, stream or None))

    def mac_cormack_staggered_backward(self, grid, field, velocity, grad_out, dt, strength, grad_field, grad_velocity, stream=0):
        gv = ptr3(grad_velocity)
        self.lib.check(self.lib.dll.phihip_mac_cormack_staggered_backward(
            self.handle, ctypes.byref(grid), ctypes.byref(ptr3(field)), ctypes.byref(ptr3(velocity)), ctypes.byref(ptr3(grad_out)), float(dt),
            float(strength), ctypes.byref(ptr3(grad_field)), ctypes.byref(gv) if gv is not None else None, stream or None))

    def mac_cormack_centered_backward(self, grid, s, s_bc, s_val, velocity, grad_out, dt, strength, grad_s, grad_velocity, stream=0):
        bc, val = self._scalar_bc(grid, s_bc, s_val)
        gv = ptr3(grad_velocity)
        self.lib.check(self.lib.dll.phihip_mac_cormack_centered_backward(
            self.handle, ctypes.byref(grid), s, ctypes.byref(bc), ctypes.byref(val), ctypes.byref(ptr3(velocity)), grad_out, float(dt),
            float(strength), grad_s, ctypes.byref(gv) if gv is not None else None, stream or None))

    def diffuse_explicit_backward(self, grid, grad_out, grad_in, diffusivity_dt, stream=0):
        self.lib.check(self.lib.dll.phihip_diffuse_explicit_backward(self.handle, ctypes.byref(grid), ctypes.byref(ptr3(grad_out)),
                                                                     ctypes.byref(ptr3(grad_in)), float(diffusivity_dt), stream or None))

    def diffuse_explicit_centered(self, grid, s, s_bc, s_val, out, diffusivity_dt, adjoint=False, stream=0):
        bc, val = self._scalar_bc(grid, s_bc, s_val)
        self.lib.check(self.lib.dll.phihip_diffuse_explicit_centered(self.handle, ctypes.byref(grid), s, ctypes.byref(bc), ctypes.byref(val), out,
                                                                     float(diffusivity_dt), int(bool(adjoint)), stream or None))

    def centered_to_staggered_backward(self, grid, s_bc, vector, grad_out, grad_s, stream=0):
        bc, _ = self._scalar_bc(grid, s_bc, None)
        vec = (c_double * 3)(*([float(x) for x in vector] + [0.0] * (3 - len(vector))))
        self.lib.check(self.lib.dll.phihip_centered_to_staggered_backward(self.handle, ctypes.byref(grid), ctypes.byref(bc), ctypes.byref(vec),
                                                                          ctypes.byref(ptr3(grad_out)), grad_s, stream or None))

    def make_incompressible_backward(self, grid, flags, mask_batch, balance, grad_velocity, grad_pressure, solve: Solve, want_info=True,
                                     stream=0):
        info = (SolveInfo * grid.batch)() if want_info else None
        self.lib.check(self.lib.dll.phihip_make_incompressible_backward(
            self.handle, ctypes.byref(grid), flags or None, int(mask_batch), int(balance) & DIV_BALANCE, ctypes.byref(ptr3(grad_velocity)),
            grad_pressure or None, ctypes.byref(solve), info, stream or None))
        return list(info) if want_info else None

    # ---- f4: slab-decomposed CG phases (see include/phihip.h) ----
    def slab_residual(self, grid, halo, flags, x, x_halo, rhs, r, sums, keep_going=False, stream=0):
        self.lib.check(self.lib.dll.phihip_slab_residual(self.handle, ctypes.byref(grid), int(halo[0]), int(halo[1]), flags or None, x,
                                                         x_halo[0] or None, x_halo[1] or None, rhs, r, sums, int(bool(keep_going)), stream or None))

    def slab_matvec(self, grid, halo, flags, first, sums_in, r, r_halo, d_old, d_halo, d_new, sum_out, solve: Solve, stream=0):
        self.lib.check(self.lib.dll.phihip_slab_matvec(self.handle, ctypes.byref(grid), int(halo[0]), int(halo[1]), flags or None, int(bool(first)),
                                                       sums_in, r, r_halo[0] or None, r_halo[1] or None, d_old, d_halo[0] or None,
                                                       d_halo[1] or None, d_new, sum_out, ctypes.byref(solve), stream or None))

    def slab_update(self, grid, halo, flags, sum_in, d, d_halo, x, r, sum_out, solve: Solve, x_only=False, stream=0):
        self.lib.check(self.lib.dll.phihip_slab_update(self.handle, ctypes.byref(grid), int(halo[0]), int(halo[1]), flags or None, sum_in, d,
                                                       d_halo[0] or None, d_halo[1] or None, x, r or None, sum_out or None, int(bool(x_only)),
                                                       ctypes.byref(solve), stream or None))

    def slab_state(self, grid, first, sums_in, solve: Solve, peek=False, stream=0):
        info = (SolveInfo * grid.batch)()
        self.lib.check(self.lib.dll.phihip_slab_state(self.handle, ctypes.byref(grid), int(bool(first)), sums_in, ctypes.byref(solve), info,
                                                      int(bool(peek)), stream or None))
        return list(info)

    def obstacle_accessible(self, grid, obstacles, count, accessible, stream=0):
        self.lib.check(self.lib.dll.phihip_obstacle_accessible(self.handle, ctypes.byref(grid), obstacles, int(count), accessible, stream or None))

    def apply_obstacles(self, grid, obstacles, count, velocity, stream=0):
        self.lib.check(self.lib.dll.phihip_apply_obstacles(self.handle, ctypes.byref(grid), obstacles, int(count), ctypes.byref(ptr3(velocity)),
                                                           stream or None))

    def build_cellflags(self, grid, accessible, active, mask_batch, flags, stream=0):
        self.lib.check(self.lib.dll.phihip_build_cellflags(self.handle, ctypes.byref(grid), accessible or None, active or None,
                                                           int(mask_batch), flags, stream or None))

    def divergence(self, grid, velocity, flags, mask_batch, balance, div, stream=0):
        self.lib.check(self.lib.dll.phihip_divergence(self.handle, ctypes.byref(grid), ctypes.byref(ptr3(velocity)), flags or None,
                                                      int(mask_batch), int(balance), div, stream or None))

    def laplace_apply(self, grid, flags, mask_batch, p, out, stream=0):
        self.lib.check(self.lib.dll.phihip_laplace_apply(self.handle, ctypes.byref(grid), flags or None, int(mask_batch), p, out,
                                                         stream or None))

    def cg_solve(self, grid, flags, mask_batch, rhs, x, solve: Solve, want_info=True, stream=0):
        info = (SolveInfo * grid.batch)() if want_info else None
        self.lib.check(self.lib.dll.phihip_cg_solve(self.handle, ctypes.byref(grid), flags or None, int(mask_batch), rhs, x,
                                                    ctypes.byref(solve), info, stream or None))
        return list(info) if want_info else None

    def cg_solve_shifted(self, grid, identity, scale, rhs, x, solve: Solve, want_info=True, stream=0):
        """ CG on (identity * I + scale * L) x = rhs with the pressure operator L of `grid` (no obstacle flags) """
        info = (SolveInfo * grid.batch)() if want_info else None
        self.lib.check(self.lib.dll.phihip_cg_solve_shifted(self.handle, ctypes.byref(grid), float(identity), float(scale), rhs, x,
                                                            ctypes.byref(solve), info, stream or None))
        return list(info) if want_info else None

    def solve_residuals(self, batch, out_device, stream=0):
        """ device-side (||r||^2, ||rhs||^2) per batch entry of the last solve, no host sync """
        self.lib.check(self.lib.dll.phihip_solve_residuals(self.handle, int(batch), out_device, stream or None))

    def solve_relative_residual(self, batch, out_device, stream=0):
        """ device-side max over the batch entries of ||r|| / ||rhs|| of the last solve (one double), no host sync """
        self.lib.check(self.lib.dll.phihip_solve_relative_residual(self.handle, int(batch), out_device, stream or None))

    def grad_subtract(self, grid, flags, mask_batch, p, velocity, stream=0):
        self.lib.check(self.lib.dll.phihip_grad_subtract(self.handle, ctypes.byref(grid), flags or None, int(mask_batch), p,
                                                         ctypes.byref(ptr3(velocity)), stream or None))

    def make_incompressible(self, grid, velocity, soft_mask, flags, mask_batch, balance, pressure, div_out, solve: Solve,
                            want_info=True, stream=0):
        info = (SolveInfo * grid.batch)() if want_info else None
        sm = ptr3(soft_mask)
        self.lib.check(self.lib.dll.phihip_make_incompressible(
            self.handle, ctypes.byref(grid), ctypes.byref(ptr3(velocity)), ctypes.byref(sm) if sm is not None else None, flags or None,
            int(mask_batch), int(balance), pressure, div_out or None, ctypes.byref(solve), info, stream or None))
        return list(info) if want_info else None

    def diffuse_explicit(self, grid, velocity, out, diffusivity_dt, stream=0):
        self.lib.check(self.lib.dll.phihip_diffuse_explicit(self.handle, ctypes.byref(grid), ctypes.byref(ptr3(velocity)),
                                                            ctypes.byref(ptr3(out)), float(diffusivity_dt), stream or None))

    def diffuse_implicit(self, grid, velocity, out, diffusivity_dt, solve: Solve, stream=0, want_info=True):
        """ diffuse.implicit of a staggered field: (I - k dt L) out = velocity per component, CG from x0 = velocity; returns rank x batch SolveInfo
        (want_info=False: info = NULL, no host read-back -- the capture-safe form) """
        info = (SolveInfo * (grid.rank * grid.batch))() if want_info else None
        self.lib.check(self.lib.dll.phihip_diffuse_implicit(self.handle, ctypes.byref(grid), ctypes.byref(ptr3(velocity)), ctypes.byref(ptr3(out)),
                                                            float(diffusivity_dt), ctypes.byref(solve), info, stream or None))
        return list(info) if want_info else None

    def diffuse_implicit_centered(self, grid, s, s_bc, s_val, out, diffusivity_dt, solve: Solve, stream=0, want_info=True):
        bc, val = self._scalar_bc(grid, s_bc, s_val)
        info = (SolveInfo * grid.batch)() if want_info else None
        self.lib.check(self.lib.dll.phihip_diffuse_implicit_centered(self.handle, ctypes.byref(grid), s, ctypes.byref(bc), ctypes.byref(val), out,
                                                                     float(diffusivity_dt), ctypes.byref(solve), info, stream or None))
        return list(info) if want_info else None

    def profile_enable(self, enable: bool):
        self.lib.check(self.lib.dll.phihip_profile_enable(self.handle, int(bool(enable))))

    def profile_read(self, reset=True):
        launches = (c_int32 * K_COUNT)()
        ms = (c_double * K_COUNT)()
        self.lib.check(self.lib.dll.phihip_profile_read(self.handle, ctypes.byref(launches), ctypes.byref(ms), int(bool(reset))))
        return {K_NAMES[k]: (launches[k], ms[k]) for k in range(K_COUNT)}

    def set_tuning(self, rows_per_thread=0, threads_per_row=0, chunk_planes=0):
        self.lib.check(self.lib.dll.phihip_set_tuning(self.handle, int(rows_per_thread), int(threads_per_row), int(chunk_planes)))

    def set_tuning_kernel(self, family, rows_per_thread=0, threads_per_row=0, chunk_planes=0):
        """ family: 0 = apply / residual, 1 = MATVEC, 2 = UPDATE """
        self.lib.check(self.lib.dll.phihip_set_tuning_kernel(self.handle, int(family), int(rows_per_thread), int(threads_per_row),
                                                             int(chunk_planes)))

    def set_small_grid_solver(self, enable):
        """ False / True, or an int > 1 = explicit cell limit of the single-kernel solver """
        self.lib.check(self.lib.dll.phihip_set_small_grid_solver(self.handle, int(enable)))

    def set_deferred_x_update(self, enable: bool):
        if hasattr(self.lib.dll, "phihip_set_deferred_x_update"):
            self.lib.check(self.lib.dll.phihip_set_deferred_x_update(self.handle, int(bool(enable))))

    def workspace_placement(self, candidates: int = -1) -> dict:
        """ Candidate allocations of the CG workspace the first solve on a freshly grown workspace chooses from (include/phihip.h phihip_workspace_placement):
        candidates 2 ... 16, 0 / 1 = keep the first allocation, -1 = unchanged. Returns the record of the most recent choice. """
        n = c_int32(0)
        us = (c_double * 2)(0.0, 0.0)
        self.lib.check(self.lib.dll.phihip_workspace_placement(self.handle, int(candidates), ctypes.byref(n), us))
        return {"candidates": int(n.value), "us_first": float(us[0]), "us_kept": float(us[1])}

    def query_advect_chunk(self) -> int:
        """ planes per workgroup of the most recent tiled self-advection (0: none yet / 2-D) """
        out = c_int32(0)
        self.lib.check(self.lib.dll.phihip_query_advect_chunk(self.handle, ctypes.byref(out)))
        return int(out.value)

    def set_advect_dma(self, enable: int = -1) -> bool:
        """ LDS-DMA fill of the tiled self-advection on regular grids (include/phihip.h): enable 1 / 0, -1 = unchanged. Returns whether the most
        recent tiled self-advection of this context took the LDS-DMA kernel. """
        out = c_int32(0)
        self.lib.check(self.lib.dll.phihip_set_advect_dma(self.handle, int(enable), ctypes.byref(out)))
        return bool(out.value)

    def set_advect_windows_2d(self, enable: bool):
        """ LDS-windowed MacCormack / centred advection passes on 2-D grids as well (default: 3-D only; the gather kernels are faster in 2-D) """
        self.lib.check(self.lib.dll.phihip_set_advect_windows_2d(self.handle, int(bool(enable))))

    def set_advect_halo(self, halo: int):
        """ -1 (default): adaptive reach of the LDS-staged advection kernels; 0: gather kernels; 1 / 2: fixed reach in cells (include/phihip.h) """
        self.lib.check(self.lib.dll.phihip_set_advect_halo(self.handle, int(halo)))

    def allreduce_residual(self, comm, values_device, count, op=2, stream=0):
        """ in-place RCCL all-reduce of `count` device doubles over the caller's ncclComm_t (op 0 = sum, 2 = max); asynchronous """
        self.lib.check(self.lib.dll.phihip_allreduce_residual(self.handle, comm, values_device, int(count), int(op), stream or None))

    def set_single_reduction_cg(self, mode: int, max_cells: int = 0):
        """ 0: two launches per CG iteration always; 1: one fused launch per iteration for latency-bound solves (cells x batch <= max_cells,
        0 = built-in threshold); 2: always the fused form """
        self.lib.check(self.lib.dll.phihip_set_single_reduction_cg(self.handle, int(mode), int(max_cells)))

    def set_resident_cg(self, mode: int, max_cells: int = 0):
        """ resident solver for 2-D fp32 grids (the whole 'CG' solve in ONE launch, cg_resident.hip): 0 never, 1 when cells x batch <= max_cells
        (0 = keep the limit), 2 whenever applicable """
        self.lib.check(self.lib.dll.phihip_set_resident_cg(self.handle, int(mode), int(max_cells)))

    def set_autotune(self, enable: bool):
        """ first-call timing of the CG launch-plan candidates (default on; off = the analytic plan, reproducible launch geometry) """
        self.lib.check(self.lib.dll.phihip_set_autotune(self.handle, int(bool(enable))))

    def set_advect_chunk(self, planes: int):
        self.lib.check(self.lib.dll.phihip_set_advect_chunk(self.handle, int(planes)))

    def advect_fallback_stats(self, stream=0):
        """ (workgroups redone by the gather path, workgroups launched) of the most recent tiled self-advection; synchronises """
        out = (c_int32 * 2)()
        self.lib.check(self.lib.dll.phihip_advect_fallback_stats(self.handle, ctypes.byref(out), stream or None))
        return out[0], out[1]

    def query_plan(self, grid, has_flags=False, family=1) -> dict:
        out = (c_int32 * 6)()
        self.lib.check(self.lib.dll.phihip_query_plan(self.handle, ctypes.byref(grid), int(bool(has_flags)), int(family), ctypes.byref(out)))
        return dict(rows=out[0], tpr=out[1], chunk=out[2], nblk=out[3], occupancy=out[4], vec=out[5])

    def workspace_bytes(self) -> int:
        out = c_size_t()
        self.lib.check(self.lib.dll.phihip_workspace_bytes(self.handle, ctypes.byref(out)))
        return out.value


DEFAULT_LIBRARY_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libphihip.so")
_default_library: Optional[Library] = None


def load_default_library() -> Library:
    """ loads phiflow_amd/lib/libphihip.so (built by ``__graft_entry__.build()``); raises `PhiHipLibraryError` if absent. """
    global _default_library
    if _default_library is None:
        override = os.environ.get("PHIHIP_LIBRARY")          # A/B measurements of another BUILD of libphihip (tools/): never a fallback
        if override:
            _default_library = Library(override, strict=False)
            return _default_library
        if not os.path.exists(DEFAULT_LIBRARY_PATH):
            raise PhiHipLibraryError(f"{DEFAULT_LIBRARY_PATH} does not exist. Build the HIP extension first "
                                     f"(`make -C phiflow_amd/csrc` or `__graft_entry__.build()`); phiflow_amd has no CPU fallback.")
        _default_library = Library(DEFAULT_LIBRARY_PATH)
    return _default_library
