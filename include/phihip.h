/*
 * phihip.h -- C ABI of libphihip.so: the MI355X (gfx950) backend for PhiFlow's incompressible-fluid time step.
 *
 * Drop-in boundary. PhiFlow (reference: /root/reference, v3.4.0) is pure Python and has no FFI of its own; the
 * interface it calls on this path is PhiML's Python backend API. Each entry point below replaces one PhiFlow-level
 * operation; a maintainer binds it with ctypes. Citations are `file:line` in /root/reference.
 *
 * Conventions
 *   - All field buffers are DEVICE pointers owned by the caller (e.g. torch-ROCm tensor.data_ptr()); the library owns
 *     only its context / workspace. Nothing is allocated inside the CG loop. Buffers must be aligned to their element size;
 *     16-byte alignment (what hipMalloc / torch allocations have) enables the 16-byte vector paths of the stencil kernels --
 *     pointers that are not 16-byte aligned (offset views) are accepted and take the scalar path.
 *   - Arrays are dense C-contiguous (batch, x, y[, z]); the LAST spatial axis is the fast one
 *     (phi/field/_field.py:160-180). `batch` independent simulations share one grid description (PhiML batch dims).
 *   - Velocity = StaggeredGrid (phi/field/_grid.py:89-176): one array per component d with
 *     shape res + (lo+up-1)*e_d where (lo,up) = valid_outer_faces(d):
 *         PHIHIP_BC_PERIODIC -> (1,0)   N_d   faces (lower face of each cell)
 *         PHIHIP_BC_CLOSED   -> (0,0)   N_d-1 faces (wall faces carry the constant boundary value, not stored)
 *         PHIHIP_BC_OPEN     -> (1,1)   N_d+1 faces
 *     (tests/commit/field/test__grid.py:25-36). Use phihip_component_shape(). A component must keep at least one face: an axis
 *     with ONE cell between two CLOSED sides is rejected (PHIHIP_ERR_BAD_ARG).
 *   - Pressure / divergence / scalars = CenteredGrid, shape res.
 *   - Every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream) except
 *     phihip_cg_solve / phihip_make_incompressible when `info != NULL`, which synchronise the stream to report.
 *   - Return value: 0 on success, negative phihip_status otherwise; phihip_last_error() gives a thread-local message.
 *     Numerical non-convergence is NOT an error: it is reported in phihip_solve_info and the Python layer raises
 *     NotConverged / Diverged like phiml.math.solve_linear does (tests/commit/physics/test_diffuse.py:60-66).
 *   - There is no CPU fallback: without a HIP device phihip_ctx_create fails with PHIHIP_ERR_NO_DEVICE.
 */
#ifndef PHIHIP_H
#define PHIHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHIHIP_VERSION 102 /* 0.1.2 */

typedef enum phihip_status {
    PHIHIP_OK = 0,
    PHIHIP_ERR_BAD_ARG = -1,
    PHIHIP_ERR_HIP = -2,
    PHIHIP_ERR_UNSUPPORTED = -3,
    PHIHIP_ERR_NO_DEVICE = -4,
    PHIHIP_ERR_ALLOC = -5
} phihip_status;

typedef enum phihip_dtype { PHIHIP_F32 = 0, PHIHIP_F64 = 1 } phihip_dtype;

/* velocity extrapolation per axis side (phiml.math.extrapolation): PERIODIC / ConstantExtrapolation / BOUNDARY */
typedef enum phihip_bc { PHIHIP_BC_PERIODIC = 0, PHIHIP_BC_CLOSED = 1, PHIHIP_BC_OPEN = 2 } phihip_bc;

/* UniformGrid + velocity Extrapolation (phi/geom/_grid.py:41-122). Unused trailing axis entries (rank 2) are ignored. */
typedef struct phihip_grid {
    int32_t rank;             /* 2 or 3 */
    int32_t dtype;            /* phihip_dtype */
    int32_t batch;            /* number of independent simulations (1 ... 65535) */
    int32_t res[3];           /* cells per axis x, y[, z] */
    double lower[3];          /* Box bounds */
    double upper[3];
    int32_t bc[3][2];         /* [axis][0 = lower side, 1 = upper side] phihip_bc; periodic must be set on both sides */
    double bc_val[3][2][3];   /* [axis][side][component] constant velocity on CLOSED sides (ZERO -> 0) */
} phihip_grid;

typedef enum phihip_method {
    PHIHIP_METHOD_CG = 0,           /* Solve('CG', ...): alpha = r.r / d.Ad, beta = r'.r' / r.r */
    PHIHIP_METHOD_CG_ADAPTIVE = 1   /* Solve('CG-adaptive', ...) (examples/grids/Fluid_Logo.ipynb): alpha = d.r / d.Ad, d' = r' - (r'.Ad / d.Ad) d */
} phihip_method;

/* phiml.math.Solve subset used by fluid.make_incompressible (phi/physics/fluid.py:96,145-156) */
typedef struct phihip_solve {
    double rel_tol;           /* stop when ||r||^2 <= max(rel_tol^2 ||rhs||^2, abs_tol^2) (per batch entry) */
    double abs_tol;
    int32_t max_iterations;
    int32_t refresh_every;    /* recompute r = y - A x every n-th iteration (PhiML: 50); 0 = never */
    int32_t check_every;      /* > 0: tolerance mode -- the host watches the continue flags the kernels publish into mapped host memory
                               * and throttles its run-ahead with an event every n iterations; 0 = run max_iterations launches */
    int32_t method;           /* phihip_method: 0 = 'CG' (also what 'auto' maps to), 1 = 'CG-adaptive' */
} phihip_solve;

/* per batch entry result of the linear solve (phiml SolveInfo: iterations, residual, converged, diverged) */
typedef struct phihip_solve_info {
    double residual_sq;
    double rhs_sq;
    int32_t iterations;
    int32_t converged;
    int32_t diverged;
    int32_t reserved;
} phihip_solve_info;

typedef struct phihip_ctx phihip_ctx;

/* ---- lifecycle ---------------------------------------------------------------------------------------------- */
int phihip_version(void);
/* "<git commit the library was built at>[+dirty] src:<16 hex digits>": the second part is the sha1 over the library's sources (csrc .hip and
 * .hpp files in name order, then include/phihip.h), which the Python layer recomputes from the tree to detect a stale .so
 * (phiflow_amd._capi.source_hash); bench.py and smoke() print it so that numbers measured on a GPU box are tied to a source state. */
const char* phihip_build_id(void);
const char* phihip_last_error(void);
/* replaces: backend selection `with backend:` / set_default_device('GPU') (phi/torch/flow.py:31-32, demos/Top_Opt/Top_Opt3D.py:190) */
int phihip_ctx_create(int device, phihip_ctx** out);
int phihip_ctx_destroy(phihip_ctx* ctx);
/* bytes of device workspace the context currently holds (grows on demand, never inside the CG loop) */
int phihip_workspace_bytes(const phihip_ctx* ctx, size_t* bytes);
/* stored shape of velocity component `comp` (phi/geom/_grid.py:204-209) */
int phihip_component_shape(const phihip_grid* grid, int comp, int32_t shape[3]);

/* ---- a1: advect.semi_lagrangian with euler back-trace (phi/physics/advect.py:156-179, :20-24) ------------------ */
/* staggered `field` advected by staggered `velocity` (pass the same pointers for self-advection); out must not alias */
int phihip_advect_staggered(phihip_ctx* ctx, const phihip_grid* grid, const void* const field[3],
                            const void* const velocity[3], void* const out[3], double dt, void* stream);
/* centred scalar (e.g. smoke) advected by the staggered velocity; s_bc/s_val = the scalar's own extrapolation
 * (PERIODIC wrap / OPEN zero-gradient / CLOSED constant s_val) */
int phihip_advect_centered(phihip_ctx* ctx, const phihip_grid* grid, const void* s, const int32_t s_bc[3][2],
                           const double s_val[3][2], const void* const velocity[3], void* out, double dt, void* stream);

/* ---- math.grid_sample(grid, coordinates, extrapolation) (call site phi/field/_resample.py:257-259; the gather every advection
 *      above is built from, exposed for samples at arbitrary points: grids of another resolution, particle positions) ------------
 * values: [values_batch][res...] with values_batch = 1 (shared) or grid.batch; `grid.res` = shape of `values`, `grid.bc` / `s_val`-style
 * rule per side taken from grid.bc and grid.bc_val[axis][side][0]: PERIODIC wrap, OPEN = clamp (zero-gradient), CLOSED = constant.
 * coords[d]: [batch][points] FRACTIONAL INDICES into `values` along axis d (PhiML's convention: coordinate i is sample i).
 * out / out_min / out_max: [batch][points]; each may be NULL (out_min and out_max only together): the multilinear interpolation
 * and the min / max over its 2^D taps (Field.closest_values, phi/field/_field.py:409-429). Bounds of `grid` are ignored. */
int phihip_grid_sample(phihip_ctx* ctx, const phihip_grid* grid, const void* values, int values_batch, const void* const coords[3],
                       int64_t points, void* out, void* out_min, void* out_max, void* stream);
/* VJP of phihip_grid_sample's `out`: grad_values [values_batch][res...] and grad_coords[d] [batch][points] are ACCUMULATED (+=);
 * either may be NULL. */
int phihip_grid_sample_backward(phihip_ctx* ctx, const phihip_grid* grid, const void* values, int values_batch,
                                const void* const coords[3], int64_t points, const void* grad_out, void* grad_values,
                                void* const grad_coords[3], void* stream);

/* ---- f2: advect.mac_cormack (phi/physics/advect.py:182-215) -------------------------------------------------------- */
/* forward + backward semi-Lagrangian pass, `fwd + correction_strength * 0.5 * (field - bwd)`, clamped to the min / max of
 * the grid values around the backward lookup (Field.closest_values, phi/field/_field.py:409-429). out must not alias. */
int phihip_mac_cormack_staggered(phihip_ctx* ctx, const phihip_grid* grid, const void* const field[3],
                                 const void* const velocity[3], void* const out[3], double dt, double correction_strength,
                                 void* stream);
int phihip_mac_cormack_centered(phihip_ctx* ctx, const phihip_grid* grid, const void* s, const int32_t s_bc[3][2],
                                const double s_val[3][2], const void* const velocity[3], void* out, double dt,
                                double correction_strength, void* stream);

/* ---- f2: resample(s * vector, to=velocity) / `s * vector @ velocity` (phi/field/_resample.py:156-157,272-276;
 *          buoyancy in Smoke_Plume.ipynb cell 5, tests/commit/physics/test_fluid.py:26) -------------------------------- */
/* out_d = mean of the two cells adjacent to each stored d-face of (s * vector[d]); accumulate != 0 adds to out instead */
int phihip_centered_to_staggered(phihip_ctx* ctx, const phihip_grid* grid, const void* s, const int32_t s_bc[3][2],
                                 const double s_val[3][2], const double vector[3], int accumulate, void* const out[3],
                                 void* stream);

/* ---- a7: obstacle masks (phi/physics/fluid.py:130-137) ---------------------------------------------------------- */
/* Packs per-cell stencil flags (1 byte / cell): bit 2*axis+side = the face on that side is open for flux
 * (hard_bcs = min(accessible_L, accessible_R), outside cells: periodic wrap / OPEN 1 / CLOSED 0), bit 6 = active.
 * accessible / active: uint8 per cell (1 = fluid) or NULL (= all ones). mask_batch = 1 (shared) or grid->batch. */
int phihip_build_cellflags(phihip_ctx* ctx, const phihip_grid* grid, const uint8_t* accessible, const uint8_t* active,
                           int mask_batch, uint8_t* flags, void* stream);

/* ---- f3: obstacles rasterised on the device (phi/physics/fluid.py:130-137, 212-240; phi/geom/_box.py:174-185,217-236;
 *          phi/geom/_sphere.py:107-120; phi/field/_angular_velocity.py:10-47) ------------------------------------------- */
typedef enum phihip_obstacle_kind { PHIHIP_OBSTACLE_BOX = 0, PHIHIP_OBSTACLE_SPHERE = 1 } phihip_obstacle_kind;
/* Obstacle(geometry, velocity, angular_velocity) with geometry = Box / Cuboid (center, half_size) or Sphere (center, radius) */
typedef struct phihip_obstacle {
    int32_t kind;                 /* phihip_obstacle_kind */
    int32_t group;                /* 0: an obstacle of its own. > 0: CONSECUTIVE entries with the same value form ONE obstacle whose
                                   * geometry is their union (phi.geom.union, phi/geom/_geom_ops.py:96-102, _box.py:235): inside = any,
                                   * signed distance = min, i.e. soft mask = max over the members. At most 16 members; they share
                                   * `velocity`, and `angular_velocity` must be 0. */
    int32_t embed_mask;           /* bit d set: the geometry is infinitely long along axis d (x = bit 0): that coordinate is ignored
                                   * (phi.geom.embed / infinite_cylinder, phi/geom/_embed.py:62-66,139-158). Such obstacles neither
                                   * rotate nor carry a rotation matrix. */
    int32_t reserved;
    double center[3];             /* x, y[, z] */
    double half_size[3];          /* box: half extents; sphere: half_size[0] = radius */
    double velocity[3];           /* linear velocity of the obstacle */
    double angular_velocity[3];   /* rank 2: [0] = scalar rotation speed; rank 3: rotation vector */
    double rotation[9];           /* box orientation: row-major matrix R (box frame -> world), local = R^T (x - center)
                                   * (Box.rotated, phi/geom/_box.py:127-152); all zero = identity */
} phihip_obstacle;
/* accessible[cell] = 1 unless the cell centre lies inside any obstacle (`~union(geometries)` sampled at the centres,
 * fluid.py:133); feed it to phihip_build_cellflags. `obstacles` is a HOST array; accessible is a device uint8 array [cells]. */
int phihip_obstacle_accessible(phihip_ctx* ctx, const phihip_grid* grid, const phihip_obstacle* obstacles, int count,
                               uint8_t* accessible, void* stream);
/* fluid.apply_boundary_conditions (fluid.py:212-240), in place on the staggered velocity: for every obstacle in order
 *   m = clip(1 - sdf(x_face) / bounding_radius(face cell), 0, 1)           (resample(geometry, velocity, soft=True, balance=1))
 *   v = safe_mul(1 - m, v) + safe_mul(m, (angular_velocity x (x_face - center) + velocity) . e_d)   (second term: moving only) */
int phihip_apply_obstacles(phihip_ctx* ctx, const phihip_grid* grid, const phihip_obstacle* obstacles, int count,
                           void* const velocity[3], void* stream);

/* ---- a2/a3: field.divergence (phi/field/_field_math.py:589,617-626) and fluid._balance_divergence (fluid.py:205-209) */
/* div = divergence(v) [* active]; flags may be NULL. `balance` is a bit set: PHIHIP_DIV_BALANCE additionally subtracts the
 * (active-weighted) mean -- fluid.py:145-148: non-flexible boundaries and no user-supplied `active`; PHIHIP_DIV_FINITE_GUARD
 * replaces non-finite divergence values by 0 on EVERY cell -- `field.where(field.is_finite(div), div, 0)`, fluid.py:143-144:
 * the user supplied `active`, "the velocity may take NaN values where it does not contribute to the pressure". Without the
 * guard only INACTIVE cells are zeroed (div * active with a select instead of the reference's multiplication: where the
 * reference would hand NaN * 0 = NaN to its solver, this library solves with 0 there). The same bits apply to the `balance`
 * argument of phihip_make_incompressible. */
#define PHIHIP_DIV_BALANCE 1
#define PHIHIP_DIV_FINITE_GUARD 4
int phihip_divergence(phihip_ctx* ctx, const phihip_grid* grid, const void* const velocity[3], const uint8_t* flags,
                      int mask_batch, int balance, void* div, void* stream);

/* ---- a4: fluid.masked_laplace (phi/physics/fluid.py:165-202), matrix-free 5/7-point operator ------------------- */
int phihip_laplace_apply(phihip_ctx* ctx, const phihip_grid* grid, const uint8_t* flags, int mask_batch,
                         const void* p, void* out, void* stream);

/* ---- a5: math.solve_linear(masked_laplace, rhs, Solve('CG', ...)) (phi/physics/fluid.py:156) ------------------- */
/* x holds x0 on entry and the solution on exit. info: array of grid->batch entries or NULL (no host sync). */
int phihip_cg_solve(phihip_ctx* ctx, const phihip_grid* grid, const uint8_t* flags, int mask_batch, const void* rhs,
                    void* x, const phihip_solve* solve, phihip_solve_info* info, void* stream);
/* The same CG on  (identity * I + scale * L) x = rhs,  L = the operator of phihip_cg_solve on this grid without obstacle flags
 * (neighbour rule from the codes: PERIODIC wraps, CLOSED = no flux / zero-gradient, OPEN = zero ghost). identity = 1, scale = -k dt is
 * the system of implicit diffusion; this entry is what the PhiML plug-in uses when `solve_linear` hands it such a matrix. The system
 * must be definite (e.g. identity > 0, scale < 0); no rank-deficiency handling. */
int phihip_cg_solve_shifted(phihip_ctx* ctx, const phihip_grid* grid, double identity, double scale, const void* rhs, void* x,
                            const phihip_solve* solve, phihip_solve_info* info, void* stream);

/* Device-side result of the most recent solve on this context: out[2*b] = ||r||^2, out[2*b+1] = ||rhs||^2 per batch entry,
 * written asynchronously on `stream` into DEVICE memory (no host sync) -- the operand of the one all-reduce per step that a
 * batch-sharded multi-GPU run performs. */
int phihip_solve_residuals(phihip_ctx* ctx, int batch, double* out_device, void* stream);

/* The same result reduced on the device to ONE double: out_device[0] = max over the batch entries of ||r|| / ||rhs|| (0 where rhs = 0) --
 * directly the operand of the step's all-reduce(MAX), without the four elementwise launches a host library would need for it. */
int phihip_solve_relative_residual(phihip_ctx* ctx, int batch, double* out_device, void* stream);

/* The ONE collective of a batch-sharded multi-GPU step, for C / C++ callers that hold an RCCL communicator themselves (the
 * Python layer goes through torch.distributed, whose `nccl` backend IS RCCL): all-reduces `count` device doubles in place over `comm`
 * (an `ncclComm_t` passed as void*), op: 0 = sum, 2 = max (RCCL's ncclRedOp_t values) -- e.g. the 2 * batch values written by
 * phihip_solve_residuals, or max over ranks of ||r|| / ||rhs||. Enqueued on `stream`; no host synchronisation. librccl is NOT a link
 * dependency of libphihip: `ncclAllReduce` is resolved at the first call from the RCCL already loaded into the process (the one that
 * created `comm`), else from librccl.so.1; PHIHIP_ERR_UNSUPPORTED if neither exists. */
int phihip_allreduce_residual(phihip_ctx* ctx, void* comm, double* values_device, int count, int op, void* stream);

/* ---- a6: v -= hard_bcs * spatial_gradient(p, at=face) (phi/physics/fluid.py:158-161) ---------------------------- */
/* in-place on velocity */
int phihip_grad_subtract(phihip_ctx* ctx, const phihip_grid* grid, const uint8_t* flags, int mask_batch, const void* p,
                         void* const velocity[3], void* stream);

/* ---- fluid.make_incompressible (phi/physics/fluid.py:94-162), fused a2..a6 ------------------------------------- */
/* velocity is projected in place; pressure holds x0 on entry and p on exit; div_out (optional) receives the rhs.
 * soft_mask (optional): per-component face factors (1 - obstacle mask) applied first (apply_boundary_conditions). */
int phihip_make_incompressible(phihip_ctx* ctx, const phihip_grid* grid, void* const velocity[3],
                               const void* const soft_mask[3], const uint8_t* flags, int mask_batch, int balance,
                               void* pressure, void* div_out, const phihip_solve* solve, phihip_solve_info* info,
                               void* stream);

/* ---- f1: diffuse.explicit order 2 on the staggered velocity (phi/physics/diffuse.py:13-60) ---------------------- */
int phihip_diffuse_explicit(phihip_ctx* ctx, const phihip_grid* grid, const void* const velocity[3],
                            void* const out[3], double diffusivity_dt, void* stream);

/* diffuse.implicit (phi/physics/diffuse.py:63-92; Heat_Flow.ipynb, Burgers.ipynb): out = (I - k dt L)^-1 field by CG from x0 = field
 * (solve_linear(sharpen, y=field, solve) with sharpen(x) = explicit(x, k, -dt)). The solver is the CG of the pressure path on the field's
 * own lattice -- every staggered component separately -- with the field's own extrapolation: PERIODIC wraps, OPEN = zero-gradient,
 * CLOSED = the constant bc_val / s_val, whose contribution is the affine part of `sharpen` and moves to the right-hand side.
 * solve: method CG or CG-adaptive, tolerances relative to the right-hand side like phihip_cg_solve. info (optional):
 * [rank][batch] for the staggered form (component-major), [batch] for the centred form. out must not alias the input. */
int phihip_diffuse_implicit(phihip_ctx* ctx, const phihip_grid* grid, const void* const velocity[3], void* const out[3],
                            double diffusivity_dt, const phihip_solve* solve, phihip_solve_info* info, void* stream);
int phihip_diffuse_implicit_centered(phihip_ctx* ctx, const phihip_grid* grid, const void* s, const int32_t s_bc[3][2],
                                     const double s_val[3][2], void* out, double diffusivity_dt, const phihip_solve* solve,
                                     phihip_solve_info* info, void* stream);

/* ---- f5: backward passes (vector-Jacobian products) -- PhiFlow is differentiable through its backends' autodiff
 *          (tests/commit/physics/test_fluid.py:55-73, tests/commit/test_colab_fluids_tutorial.py:11-34) ------------------- */
/* Gradients are ACCUMULATED (+=) into grad_* buffers (zero them first); NULL skips that gradient. */
int phihip_advect_staggered_backward(phihip_ctx* ctx, const phihip_grid* grid, const void* const field[3],
                                     const void* const velocity[3], const void* const grad_out[3], double dt,
                                     void* const grad_field[3], void* const grad_velocity[3], void* stream);
int phihip_advect_centered_backward(phihip_ctx* ctx, const phihip_grid* grid, const void* s, const int32_t s_bc[3][2],
                                    const double s_val[3][2], const void* const velocity[3], const void* grad_out, double dt,
                                    void* grad_s, void* const grad_velocity[3], void* stream);
/* MacCormack: recomputes the semi-Lagrangian intermediate, then correction-pass adjoint + semi-Lagrangian adjoint; a clamped
 * sample passes its gradient to the extremal tap. grad_field / grad_s are required, grad_velocity may be NULL. */
int phihip_mac_cormack_staggered_backward(phihip_ctx* ctx, const phihip_grid* grid, const void* const field[3],
                                          const void* const velocity[3], const void* const grad_out[3], double dt,
                                          double correction_strength, void* const grad_field[3], void* const grad_velocity[3],
                                          void* stream);
int phihip_mac_cormack_centered_backward(phihip_ctx* ctx, const phihip_grid* grid, const void* s, const int32_t s_bc[3][2],
                                         const double s_val[3][2], const void* const velocity[3], const void* grad_out,
                                         double dt, double correction_strength, void* grad_s, void* const grad_velocity[3],
                                         void* stream);
/* adjoint of phihip_diffuse_explicit: grad_in += (I + k dt L)^T grad_out */
int phihip_diffuse_explicit_backward(phihip_ctx* ctx, const phihip_grid* grid, const void* const grad_out[3],
                                     void* const grad_in[3], double diffusivity_dt, void* stream);
/* diffuse.explicit of a CenteredGrid with its own extrapolation (phi/physics/diffuse.py:13-60); adjoint != 0: `out` is the
 * ACCUMULATED input gradient and `s` the output gradient */
int phihip_diffuse_explicit_centered(phihip_ctx* ctx, const phihip_grid* grid, const void* s, const int32_t s_bc[3][2],
                                     const double s_val[3][2], void* out, double diffusivity_dt, int adjoint, void* stream);
int phihip_centered_to_staggered_backward(phihip_ctx* ctx, const phihip_grid* grid, const int32_t s_bc[3][2],
                                          const double vector[3], const void* const grad_out[3], void* grad_s, void* stream);
/* adjoint of phihip_make_incompressible (implicit-function gradient of the linear solve like phiml's solve_linear backward):
 * grad_velocity holds dL/dv_out on entry and dL/dv_in (before soft masks) on exit; grad_pressure = dL/dp or NULL. */
int phihip_make_incompressible_backward(phihip_ctx* ctx, const phihip_grid* grid, const uint8_t* flags, int mask_batch,
                                        int balance, void* const grad_velocity[3], const void* grad_pressure,
                                        const phihip_solve* solve, phihip_solve_info* info, void* stream);

/* ---- f4: ONE simulation decomposed into slabs along x over several ranks (no reference counterpart) -------
 * Each rank calls these phases on its slab (`grid` = the LOCAL slab: res[0] = local planes, bounds = the slab's box); the host
 * layer (phiflow_amd/slab.py) exchanges the boundary planes with the neighbour ranks into `*_lo` / `*_hi` (one plane
 * [batch][y][z] each, used where halo_lo / halo_hi != 0 -- otherwise the grid's own boundary rule applies on that side) and
 * all-reduces the per-batch sums between the phases (`sums` are DEVICE doubles).
 *   residual: r = rhs - A x ; sums[0..batch) = local sum r^2, sums[batch..2 batch) = local sum rhs^2
 *   matvec  : d_new = r + beta d_old, beta from the GLOBAL sums_in (first: sums_in = {sum r^2, sum rhs^2} of the residual phase,
 *             afterwards the global sum r^2 of the last update); sum_out[batch] = local d_new . A d_new
 *   update  : x += alpha d ; r -= alpha A d, alpha from the GLOBAL sum_in = d . A d ; sum_out[batch] = local sum r^2
 *             (x_only: the true-residual refresh iteration, follow with residual(keep_going = 1))
 *   state   : folds the last global sum r^2 into the control block and reports it (synchronises); peek != 0 leaves the chain
 *             untouched; info[b].reserved = 1 while entry b would keep iterating */
int phihip_slab_residual(phihip_ctx* ctx, const phihip_grid* grid, int halo_lo, int halo_hi, const uint8_t* flags, const void* x,
                         const void* x_lo, const void* x_hi, const void* rhs, void* r, double* sums, int keep_going, void* stream);
int phihip_slab_matvec(phihip_ctx* ctx, const phihip_grid* grid, int halo_lo, int halo_hi, const uint8_t* flags, int first,
                       const double* sums_in, const void* r, const void* r_lo, const void* r_hi, const void* d_old,
                       const void* d_lo, const void* d_hi, void* d_new, double* sum_out, const phihip_solve* solve, void* stream);
int phihip_slab_update(phihip_ctx* ctx, const phihip_grid* grid, int halo_lo, int halo_hi, const uint8_t* flags,
                       const double* sum_in, const void* d, const void* d_lo, const void* d_hi, void* x, void* r, double* sum_out,
                       int x_only, const phihip_solve* solve, void* stream);
int phihip_slab_state(phihip_ctx* ctx, const phihip_grid* grid, int first, const double* sums_in, const phihip_solve* solve,
                      phihip_solve_info* info, int peek, void* stream);

/* ---- measurement ------------------------------------------------------------------------------------------------ */
/* Kernel families timed with hipEvent pairs on the solve stream while profiling is enabled. */
typedef enum phihip_kernel_id {
    PHIHIP_K_ADVECT = 0, PHIHIP_K_DIVERGENCE = 1, PHIHIP_K_CG_RESIDUAL = 2, PHIHIP_K_CG_MATVEC_DOT = 3,
    PHIHIP_K_CG_UPDATE = 4, PHIHIP_K_CG_SCALAR = 5, PHIHIP_K_GRAD_SUBTRACT = 6, PHIHIP_K_OTHER = 7,
    PHIHIP_K_CG_UPDATE_R = 8,   /* the r-only form of the update (odd iterations of the deferred x update): a kernel of its own */
    PHIHIP_K_COUNT = 9
} phihip_kernel_id;
int phihip_profile_enable(phihip_ctx* ctx, int enable);
/* synchronises, then returns launches and summed milliseconds per family since the last reset */
int phihip_profile_read(phihip_ctx* ctx, int32_t launches[PHIHIP_K_COUNT], double total_ms[PHIHIP_K_COUNT], int reset);
/* tile configuration of the CG marching kernels: rows per thread (1,2,4) and threads per row (16,32,64); threads_per_row = 128 selects the ROW
 * tile (r4: whole rows of 65 ... 128 16-byte vectors, the lane count per row taken from the grid, no halo columns -- available for such rows
 * only, otherwise the full-width tile (1, 64) is used); 0 = auto */
int phihip_set_tuning(phihip_ctx* ctx, int rows_per_thread, int threads_per_row, int chunk_planes);
/* the same for one kernel family only: 0 = operator apply / residual, 1 = MATVEC (d = r + beta d; d.Ad), 2 = UPDATE (x, r) */
int phihip_set_tuning_kernel(phihip_ctx* ctx, int family, int rows_per_thread, int threads_per_row, int chunk_planes);

/* Grids of at most 8192 cells per batch entry (16384 for fp32 batches of >= 8 entries) are solved by ONE kernel (one workgroup per batch entry, the
 * search direction in LDS, no kernel boundary or host polling inside the loop; cg_small.hip). enable = 0 forces the two-launch marching
 * kernels for every size -- it also switches the automatic choice of the single-reduction form off -- (A/B measurements, tests of the
 * marching path on small grids); enable > 1 sets the cell limit explicitly
 * (capped at 16384 fp32 / 8192 fp64). Default: enabled. */
int phihip_set_small_grid_solver(phihip_ctx* ctx, int enable);
/* 'CG' with the marching kernels updates the solution every OTHER iteration only: x does not enter the recurrence, and the step that
 * was skipped is recovered in the next update from operands that kernel reads anyway (d_k = (d_{k+1} - r_{k+1}) / beta_{k+1}), so an
 * iteration moves 7 instead of 8 words per cell. Same iterates r, d, alpha, beta; x equal up to rounding. enable = 0 updates x in
 * every iteration (A/B measurements, tests). Default: enabled. */
int phihip_set_deferred_x_update(phihip_ctx* ctx, int enable);
/* Reach of the LDS-staged advection kernels (advect_tile.hip: self-advection of the staggered velocity, one launch for all components;
 * advect_win.hip: correction pass of mac_cormack(v, v), semi_lagrangian / mac_cormack of a centred scalar). Lookups displaced by less than
 * `halo` cells are served from LDS; (tile, plane) units with a larger displacement are recomputed by a fix-up launch with the gather code --
 * same result on either path.
 *   -1 (default)  adaptive per kind of pass: the fix-up launch publishes how many units fell back, the next pass of that kind picks
 *                 1 cell (fastest below CFL 1), 2 cells (+10-25 % time, immune below CFL 2) or the gather kernels (flat cost) from that
 *                 fraction, and probes the cheaper form every 64 calls. The choice depends on the data only, not on timing.
 *    0            the one-launch-per-component gather kernels for every call
 *    1 / 2        fixed reach (the staggered MacCormack correction has reach 1 only)
 *    3            experimental: reach 1 with 16-row tiles for the self-advection (3-D) */
int phihip_set_advect_halo(phihip_ctx* ctx, int halo);
/* The other advection passes -- the correction pass of mac_cormack(v, v), semi_lagrangian / mac_cormack of a centred scalar -- are served from
 * LDS windows too (advect_win.hip; halo 1, gather fix-up for larger displacements) whenever halo != 0, on 3-D grids. On 2-D grids the gather
 * kernels are faster (one plane per workgroup: nothing to overlap the fill with) and stay the default; enable != 0 switches the windows on
 * there as well (parity tests, A/B measurements). */
int phihip_set_advect_windows_2d(phihip_ctx* ctx, int enable);
/* r5: the tiled self-advection fills its LDS ring by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass) on REGULAR grids --
 * 3-D, reach 1, no CLOSED side, periodic fast axis with rows of whole 16-byte vectors, 16-byte aligned arrays; every other grid keeps the
 * register-staged kernel (per-element padding). enable: 1 (default) / 0 (also: environment PHIHIP_ADVECT_DMA=0) / -1 = leave as it is;
 * *last_was_dma (may be NULL) = 1 when the most recent tiled self-advection of this context took the LDS-DMA kernel. Same samples, same
 * arithmetic, same bits either way (asserted by the parity tests). */
int phihip_set_advect_dma(phihip_ctx* ctx, int enable, int32_t* last_was_dma);
/* r6: placement of the CG workspace. What an iteration of Solve('CG') (phi/physics/fluid.py:156-161) costs on vectors beyond the caches depends on WHICH
 * allocations hold r, d0, d1 -- the relative position of the streams in the physical address space: 512^3 fp32 0.666 ... 0.731 ms per iteration between six
 * workspaces alive at once, every one stable to 0.1 % (profiles/r06_ws_placement_probe.jsonl). The first solve on a freshly grown workspace therefore allocates
 * `candidates` triples, times the iteration loop with the tuned launch plans on each and keeps the fastest (behind the first-call autotune and under its
 * conditions: autotune on, stream not capturing; never more than half of the free device memory; the others are freed before the call returns). Results are
 * bit-identical whichever is kept. Vectors of <= 72 MB (256^3 fp32: the Infinity Cache regime) cost the same wherever they live and are not placed;
 * candidates held at once stay within 32 GiB. candidates: 2 ... 16, 0 / 1 = keep the first allocation, < 0 = leave as it is (default 12; environment
 * PHIHIP_WS_CANDIDATES at context creation). *last_candidates (may be NULL) = triples the most recent choice had (0: none yet), last_us (may be NULL) =
 * {microseconds per iteration on the first allocation, on the one kept}. */
int phihip_workspace_placement(phihip_ctx* ctx, int candidates, int32_t* last_candidates, double last_us[2]);
/* planes of the slow axis one workgroup of the tiled self-advection marches over (3-D); 0 = planned from the kernel's occupancy */
int phihip_set_advect_chunk(phihip_ctx* ctx, int planes);
/* planes per workgroup the most recent tiled self-advection of this context ran with (3-D; 0 = none yet / 2-D): what the first-call
 * measurement settled on -- tools/path_workload.py pins it (phihip_set_advect_chunk) in the profiled runs so that no candidate launch shares
 * the kernel's name */
int phihip_query_advect_chunk(phihip_ctx* ctx, int32_t* planes);
/* Diagnostics of the most recent LDS-staged advection launch on this context (synchronises `stream`): out[0] = (tile, plane) units that met
 * a lookup outside their LDS window and were redone by the gather path, out[1] = units of the launch (tiles x planes x batch entries).
 * {0, 0} if none has run. (Until r3 the unit was a workgroup's whole chunk of planes.) */
int phihip_advect_fallback_stats(phihip_ctx* ctx, int32_t out[2], void* stream);
/* Solve('CG') on grids whose iteration is bound by the two kernel boundaries rather than by memory traffic (batched 2-D, small 3-D) runs the
 * SINGLE-REDUCTION form of CG (Chronopoulos & Gear): one launch per iteration that carries w = A r and s = A p as vectors -- the same
 * iterates as the two-launch form (alpha, beta from five sums of the previous launch: gamma = r.r, delta = (A r).r, mu = r.s, nu = (A r).p,
 * sigma = p.s; p'.A p' = delta + beta (mu + nu) + beta^2 sigma is an identity of the stored vectors, so the attainable accuracy is that of
 * the two-launch form -- the textbook closure delta - beta gamma / alpha of rounds 1-2 stalled 1-2 digits early in fp32,
 * tools/cg1_accuracy.py), 10 instead of 7 words per cell, half the launches: 1.1-1.3x faster per iteration up to ~1 M cells x batch
 * (512^2: 10.7 -> 8.2 us). mode 0: never; 1 (default): when cells x batch <= max_cells (0 = built-in threshold, 1.2 M fp32 / 0.6 M fp64);
 * 2: always. 'CG-adaptive', slab-decomposed solves and grids of the single-workgroup solver are not affected. */
int phihip_set_single_reduction_cg(phihip_ctx* ctx, int mode, long long max_cells);
/* Resident solver for 2-D fp32 grids (cg_resident.hip; r4): the WHOLE 'CG' solve of a batch is ONE launch. A batch entry is owned by
 * ceil(n_y / 16) workgroups of 1024 threads that keep r, A r, A p, p and x in registers for the whole solve and meet at one barrier per
 * iteration (boundary rows + five partial sums through L2); the control logic (tolerances, divergence test, true-residual refresh --
 * phiml's cg loop) runs on the device, the launch ends when its entries have converged, the host never polls. Same
 * recurrences as the single-reduction form above. Applicable to rank-2 fp32 grids (r6: with or without cell flags -- a thread keeps the four flag bytes of each of its vectors in one register), rows of whole 16-byte vectors up to
 * 512 cells, batch x workgroups <= compute units; everything else keeps the launch-per-iteration kernels.
 * mode 0: never; 1 (DEFAULT since r6): batches of >= 2 entries, and single entries with rows of <= 256 cells (128^2 ... 256^2: 7.1 -> 6.2-6.6 us per iteration), with cells x batch <= max_cells (0 = keep the current limit, initially 16 Mi; r6: a batch of more entries than one launch holds runs as sub-batches one behind the other where one launch holds >= 2 Mi cells: 16 x 512^2 24.5 -> 19.7 us, 64 x 512^2 87.9 -> 79.2); 2: whenever
 * applicable. Measured on the MI355X (us per iteration, launch forms -> resident): 8 x 512^2 14.3 -> 10.1, 4 x 512^2 11.4 -> 8.5, 2 x 512^2 9.4 -> 8.1,
 * 16 x 256^2 11.6 -> 8.4, 4 x 384^2 10.6 -> 8.1; ONE 512^2 entry 7.8 -> 8.3 (hence >= 2 entries in mode 1).
 * r6: (i) the solve number of the exchange's tags lives on the DEVICE and is bumped by a one-workgroup kernel in front of every launch, so a captured solve is
 * replayable (until r5 the solver was refused under capture); (ii) resident solves of one process are chained by an event across streams and contexts, so two of them
 * never each hold half of the chip; (iii) whether the grid fits is asked of the occupancy calculator before every launch -- that made it safe as the default. The launch
 * can be made COOPERATIVE (environment PHIHIP_RESIDENT_COOP=1 at context creation: the runtime then guarantees co-residency and refuses a launch that does not fit); it is
 * not the default because a cooperative launch synchronises with every queue of the device: +0.04 ms per solve in a fresh process, +0.5 ms in a process that owns side
 * streams (profiles/r06_resident_coop_cost.txt). Every wait is
 * still bounded (~1 s: e.g. a foreign kernel that occupies CUs for that long): the solve then fails with PHIHIP_ERR_HIP instead of hanging -- the failing
 * call itself when it asked for `info` (the only case in which the library synchronises with the launch), otherwise the NEXT resident solve of the context.
 * "Applicable" asks the occupancy calculator (workgroups per CU x CUs >= batch x workgroups per entry, at most 64 per entry). PHIHIP_RESIDENT_CG=0|1|2 in the
 * environment sets the mode at context creation. */
int phihip_set_resident_cg(phihip_ctx* ctx, int mode, long long max_cells);
/* The first CG solve on a (grid, dtype, batch) times the tile / chunk candidates of its three marching kernels on the context's workspace
 * (a few dozen launches, once) and caches the fastest per kernel family; phihip_query_plan reports the result. enable = 0 (or
 * PHIHIP_AUTOTUNE=0 in the environment when the context is created) keeps the analytic launch plan: same launch geometry, hence the same
 * summation order of the dot products, in every process. Explicit phihip_set_tuning[_kernel] settings always win. Default: enabled. */
int phihip_set_autotune(phihip_ctx* ctx, int enable);
/* launch plan the library would use for this grid and kernel family: out = {rows per thread, threads per row, planes per
 * workgroup, workgroups per batch entry, resident workgroups per CU of that kernel, vector width} */
int phihip_query_plan(phihip_ctx* ctx, const phihip_grid* grid, int has_flags, int family, int32_t out[6]);

#ifdef __cplusplus
}
#endif
#endif /* PHIHIP_H */
