#!/usr/bin/env python3
"""
Launches EVERY kernel of the hot path a few times so that one rocprofv3 run sees them all, and writes a
manifest that says how many bytes each kernel moves by construction. tools/kernel_roofline.py joins the manifest with the rocprofv3
kernel trace (durations) and PMC passes (FETCH_SIZE / WRITE_SIZE) into profiles/rNN_kernel_roofline.json.

One GROUP per process, so that a kernel name maps to one problem size:
    f32_256   256^3 fp32 periodic Taylor-Green: the benchmark step's kernels + every f-row kernel (diffusion, centred advection, MacCormack,
              buoyancy resample, obstacle kernels, grid_sample, the adjoints)
    f32_512   512^3 fp32 periodic pressure solve (BASELINE configs[2]): the CG kernels on an HBM-resident working set
    f64_384   384^3 fp64 closed cavity + box obstacle (BASELINE configs[4]): advection, divergence, flagged CG kernels, gradient subtraction

    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d OUT/stats -o k -- python tools/path_workload.py --group f32_256 --manifest OUT/manifest.json
    cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT/FETCH_SIZE -o pmc -- python tools/path_workload.py --group f32_256
    cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d OUT/WRITE_SIZE -o pmc -- python tools/path_workload.py --group f32_256

`--lib tests/hipemu/libphihip_emu.so --device cpu --size 12` is a dry run on the CPU emulation (checks the call sequence, no timings).
"""
import argparse
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phiflow_amd import _capi as C   # noqa: E402

MANIFEST = []


def note(kernel, label, bytes_per_launch, words, basis, launches_per_call=1):
    """ kernel: substring (regex) of the demangled kernel name; bytes the kernel moves per launch by construction """
    MANIFEST.append(dict(kernel=kernel, label=label, bytes_per_launch=int(bytes_per_launch), words_per_cell=words, basis=basis,
                         launches_per_call=launches_per_call))


def P(ts):
    return [t.data_ptr() for t in ts]


def calibration_copy(dev):
    if dev.type != "cuda":
        return
    a = torch.randn(128 * 1024 * 1024, device=dev)       # 512 MiB: known bytes for the FETCH_SIZE / WRITE_SIZE units (tools/pmc_summary.py)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()


def sync(dev):
    if dev.type == "cuda":
        torch.cuda.synchronize()


def group_f32_256(ctx, dev, n, reps):
    L = 2 * math.pi
    w, N = 4, n ** 3
    grid = C.make_grid(3, C.PHIHIP_F32, 1, (n, n, n), (0, 0, 0), (L, L, L), ((0, 0),) * 3)
    h = L / n
    idx = torch.arange(n, device=dev, dtype=torch.float64)
    face, cent = idx * h, (idx + 0.5) * h
    u = (torch.cos(face)[:, None] * torch.sin(cent)[None, :])[:, :, None].expand(n, n, n)
    v_ = (-torch.sin(cent)[:, None] * torch.cos(face)[None, :])[:, :, None].expand(n, n, n)
    vel = [t.to(torch.float32).unsqueeze(0).contiguous() for t in (u, v_, 0.3 * u.transpose(0, 2))]
    out = [torch.empty_like(t) for t in vel]
    tmp = [torch.empty_like(t) for t in vel]
    s = torch.rand(1, n, n, n, device=dev)
    s2 = torch.empty_like(s)
    p = torch.zeros(1, n, n, n, device=dev)
    div = torch.empty_like(p)
    dt = 0.5 * h
    s_bc = ((C.BC_PERIODIC, C.BC_PERIODIC),) * 3      # the scalar shares the periodic box
    solve = C.Solve(0.0, 0.0, 12, 0, 0, 0)
    for _ in range(reps):
        # a1: self-advection, LDS-tiled (one launch) and the gather kernels (one launch per component)
        ctx.set_advect_halo(1)
        ctx.advect_staggered(grid, P(vel), P(vel), P(out), dt)
        if hasattr(ctx, "set_advect_dma"):      # r5: the same pass through the register-staged kernel (what non-regular grids run)
            ctx.set_advect_dma(0)
            ctx.advect_staggered(grid, P(vel), P(vel), P(out), dt)
            ctx.set_advect_dma(1)
        ctx.set_advect_halo(0)
        ctx.advect_staggered(grid, P(vel), P(vel), P(out), dt)
        ctx.set_advect_halo(1)
        # f2: centred scalar (smoke), MacCormack (centred / staggered): LDS windows (advect_win.hip), then the gather kernels (halo 0)
        for halo in (1, 0):
            ctx.set_advect_halo(halo)
            ctx.advect_centered(grid, s.data_ptr(), s_bc, None, P(vel), s2.data_ptr(), dt)
            ctx.mac_cormack_centered(grid, s.data_ptr(), s_bc, None, P(vel), s2.data_ptr(), dt, 1.0)
            ctx.mac_cormack_staggered(grid, P(vel), P(vel), P(tmp), dt, 1.0)
        ctx.set_advect_halo(1)
        ctx.centered_to_staggered(grid, s.data_ptr(), s_bc, None, (0.0, 0.0, 0.1), True, P(tmp))
        # f1: explicit diffusion, staggered and centred
        ctx.diffuse_explicit(grid, P(vel), P(tmp), 0.1 * dt)
        ctx.diffuse_explicit_centered(grid, s.data_ptr(), s_bc, None, s2.data_ptr(), 0.1 * dt)
        # a2-a6: projection = divergence (+ balance sums), CG, gradient subtraction
        p.zero_()
        ctx.make_incompressible(grid, P(out), None, 0, 1, True, p.data_ptr(), div.data_ptr(), solve, want_info=False)
        ctx.laplace_apply(grid, 0, 1, p.data_ptr(), div.data_ptr())
    sync(dev)
    note(r"advect_self_dma_kernel<float", "a1 semi-Lagrangian self-advection, LDS ring filled by LDS-DMA (r5; the benchmark step)", 6 * w * N, 6, "read 3 + write 3 components")
    note(r"advect_self_tile_kernel<float, 3", "a1 semi-Lagrangian self-advection, register-staged LDS tiles (grids the LDS-DMA kernel cannot take; here: PHIHIP_ADVECT_DMA=0 pass)", 6 * w * N, 6, "read 3 + write 3 components")
    note(r"advect_staggered_kernel<float, 3, \d, 0>", "a1 gather kernel, one component per launch", 4 * w * N, 4, "read 3 components (taps) + write 1")
    note(r"advect_win_kernel<float, 1, 3", "f2 semi-Lagrangian advection of a centred scalar, LDS windows (r4)", 5 * w * N, 5, "read scalar + 3 components, write scalar")
    note(r"advect_win_kernel<float, 2, 3", "f2 MacCormack correction pass, centred scalar, LDS windows (r4)", 6 * w * N, 6, "read scalar, forward result, 3 components; write 1")
    note(r"advect_win_kernel<float, 0, 3", "f2 MacCormack correction pass of the staggered velocity, ALL components, LDS windows (r4)", 9 * w * N, 9,
         "read 3 velocity + 3 forward-pass components; write 3")
    note(r"advect_win_fixup_kernel<float", "fix-up launch behind every LDS-window pass (workgroups whose lookups left the window; none here)", 0, 0, "reads one flag per workgroup")
    note(r"advect_self_fixup_kernel<float", "fix-up launch behind the tiled self-advection (none flagged here)", 0, 0, "reads one flag per workgroup")
    note(r"advect_centered_kernel<float, 3, 0>", "f2 semi-Lagrangian advection of a centred scalar, gather kernel (halo 0)", 5 * w * N, 5, "read scalar + 3 components, write scalar")
    note(r"advect_centered_kernel<float, 3, 1>", "f2 MacCormack correction pass, centred scalar, gather kernel (halo 0)", 6 * w * N, 6, "read scalar, forward result, 3 components; write 1")
    note(r"advect_staggered_kernel<float, 3, \d, 1>", "f2 MacCormack correction pass, one staggered component, gather kernel (halo 0)", 6 * w * N, 6, "read field, forward result, 3 components; write 1")
    note(r"centered_to_staggered_vec_kernel<float, 3", "f2 buoyancy: v += resample(s * (0, 0, 0.1)), one launch (r4: 16-byte vectors; one component has a non-zero factor)", 3 * w * N, 3,
         "read scalar, read + write the component")
    note(r"centered_to_staggered_kernel<float>", "f2 buoyancy resample, scalar kernel (rows that are not whole vectors)", 3 * w * N, 3, "read scalar, read + write the component")
    note(r"divergence_vec_kernel<float, 3", "a2 divergence + balance sums (r4: 16-byte vectors)", 4 * w * N, 4, "read 3 components, write div")
    note(r"divergence_kernel<float, 3>", "a2 divergence + balance sums, scalar kernel", 4 * w * N, 4, "read 3 components, write div")
    note(r"march_kernel<float, 4, \d, \d+, 8, false", "a3+a5 initial residual with the balance shift folded in", 4 * w * N, 4, "read x, y; write y, r")
    note(r"march_kernel<float, 4, \d, \d+, 2, false", "a5 CG MATVEC d = r + beta d, d.Ad", 3 * w * N, 3, "read r, d; write d")
    note(r"march_kernel<float, 4, \d, \d+, 6, false", "a5 CG UPDATE_R r -= alpha A d", 3 * w * N, 3, "read r, d; write r")
    note(r"march_kernel<float, 4, \d, \d+, 7, false", "a5 CG UPDATE_X2 x += two steps, r -= alpha A d", 5 * w * N, 5, "read x, r, d; write x, r")
    note(r"march_kernel<float, 4, \d, \d+, 0, false", "a4 masked_laplace apply AND (r4) f1 diffuse.explicit: one MODE_APPLY pass per component / scalar with the operator I + k dt L", 2 * w * N, 2, "read p, write A p")
    note(r"march_apply_multi_kernel<float", "f1 diffuse.explicit of the staggered velocity, r6: ALL components in one launch (I + k dt L per lattice)", 6 * w * N, 6, "read + write 3 components")
    note(r"grad_subtract_vec_kernel<float, 3", "a6 gradient subtraction, all components", 7 * w * N, 7, "read p, read + write 3 components")
    note(r"mask_faces_kernel<float", "f5 projection adjoint: hard_bcs mask of the face gradients", 2 * w * N, 2, "read + write one component")

    # f3: obstacles on the device
    obs = C.make_obstacles([dict(kind=C.OBSTACLE_BOX, center=(L / 2, L / 2, L / 2), half_size=(L / 8, L / 8, L / 8), velocity=(0.1, 0, 0),
                                 angular_velocity=(0, 0, 0.2)),
                            dict(kind=C.OBSTACLE_SPHERE, center=(L / 4, L / 4, L / 4), half_size=(L / 10, 0, 0))])
    acc = torch.empty(n, n, n, dtype=torch.uint8, device=dev)
    flags = torch.empty(n, n, n, dtype=torch.uint8, device=dev)
    for _ in range(reps):
        ctx.obstacle_accessible(grid, obs, 2, acc.data_ptr())
        ctx.build_cellflags(grid, acc.data_ptr(), 0, 1, flags.data_ptr())
        ctx.apply_obstacles(grid, obs, 2, P(tmp))
    sync(dev)
    note(r"obstacle_accessible_kernel", "f3 obstacle rasterisation (2 obstacles; r5: memset + the patches of the obstacles' index bounding box only: the bytes by construction are an upper bound)", 1 * N, 0.25, "write 1 byte per cell")
    note(r"cellflags_vec_kernel", "a7 packed stencil flags, byte-parallel kernel (r5: 16 cells per thread)", 2 * N, 0.5, "read + write 1 byte per cell")
    note(r"cellflags_kernel", "a7 packed stencil flags, one byte per thread (rows that are not whole 4-byte vectors)", 2 * N, 0.5, "read + write 1 byte per cell")
    note(r"apply_obstacles_kernel<float>", "f3 apply_boundary_conditions, r5: ALL components in one launch over the patches of the obstacles' bounding box (2 obstacles; samples farther than one cell radius from every obstacle are neither read nor written: the bytes by construction are an upper bound)", 3 * 2 * w * N, 6, "read + write three components")

    # math.grid_sample + adjoint (fields on different grids, rk4)
    npts = N
    coords = [(torch.rand(1, npts, device=dev) * (n - 1)).contiguous() for _ in range(3)]
    so = torch.empty(1, npts, device=dev)
    gval = torch.zeros(1, n, n, n, device=dev)
    gco = [torch.zeros(1, npts, device=dev) for _ in range(3)]
    for _ in range(max(1, reps // 2)):
        ctx.grid_sample(grid, s.data_ptr(), 1, P(coords), npts, so.data_ptr())
        ctx.grid_sample_backward(grid, s.data_ptr(), 1, P(coords), npts, so.data_ptr(), gval.data_ptr(), P(gco))
    sync(dev)
    note(r"grid_sample_kernel<float, 3", "math.grid_sample at N random points (uncoalesced gathers by construction)", 5 * w * N, 5, "read 3 coordinates + the field, write 1")
    note(r"grid_sample_bwd_kernel<float, 3", "grid_sample adjoint (atomic scatter to 8 taps + d/d coords)", 10 * w * N, 10,
         "read 3 coords, grad_out, field; read-modify-write grad_values; write 3 coord gradients")

    # f5: adjoints of the step's operators
    g_out = [torch.randn_like(t) for t in vel]
    gf = [torch.zeros_like(t) for t in vel]
    gv = [torch.zeros_like(t) for t in vel]
    gs = torch.zeros_like(s)
    for _ in range(max(1, reps // 2)):
        ctx.advect_staggered_backward(grid, P(vel), P(vel), P(g_out), dt, P(gf), P(gv))
        ctx.advect_centered_backward(grid, s.data_ptr(), s_bc, None, P(vel), so.view(1, n, n, n).data_ptr(), dt, gs.data_ptr(), P(gv))
        ctx.mac_cormack_centered_backward(grid, s.data_ptr(), s_bc, None, P(vel), so.view(1, n, n, n).data_ptr(), dt, 1.0, gs.data_ptr(), P(gv))
        ctx.mac_cormack_staggered_backward(grid, P(vel), P(vel), P(g_out), dt, 1.0, P(gf), P(gv))
        ctx.diffuse_explicit_backward(grid, P(g_out), P(gf), 0.1 * dt)
        ctx.centered_to_staggered_backward(grid, s_bc, (0.0, 0.0, 0.1), P(g_out), gs.data_ptr())
        ctx.make_incompressible_backward(grid, 0, 1, True, P(g_out), 0, solve, want_info=False)
    sync(dev)
    note(r"advect_bwd_trace_kernel<float, 3, \d, true>", "f5 advection adjoint pass A, one staggered component per launch: back-trace, store x*, g, g d(out)/d(x*)", 12 * w * N, 12,
         "read grad_out, field, 3 velocity components; write 3 coordinates, g, 3 du (7 scratch words)")
    note(r"advect_bwd_trace_all_kernel<float, 3>", "f5 advection adjoint pass A, r6: ALL staggered components in one launch", 36 * w * N, 36,
         "per component: read grad_out, field, 3 velocity components; write 3 coordinates, g, 3 du")
    note(r"advect_bwd_field_gather_all_kernel<float, 3>", "f5 advection adjoint pass B, r6: ALL staggered components in one launch", 18 * w * N, 18,
         "per component: read 3 coordinates + g (one-cell halo, staged in LDS), read + write grad_field")
    note(r"advect_bwd_velocity_gather_all_kernel<float, 3, true>", "f5 advection adjoint pass C (staggered), r6: ALL velocity components in one launch", 15 * w * N, 15,
         "per component: read du of the 3 source components, read + write grad_velocity")
    note(r"mac_cormack_bwd_all_kernel<float, 3>", "f5 adjoint of the MacCormack correction pass, r6: ALL staggered components in one launch", 45 * w * N, 45,
         "per component: read grad_out, field, forward result, 3 velocity components; rmw grad_field, grad_fwd; write the sample record + 3 du")
    note(r"advect_bwd_trace_kernel<float, 3, 2, false>", "f5 advection adjoint pass A, centred scalar", 12 * w * N, 12,
         "read grad_out, scalar, 3 velocity components; write 3 coordinates, g, 3 du")
    note(r"advect_bwd_field_gather_kernel<float, 3>", "f5 advection adjoint pass B: field gradient as a gather of hat weights over the 27 neighbouring samples", 6 * w * N, 6,
         "read 3 coordinates + g (with a one-cell halo, staged in LDS), read + write grad_field")
    note(r"advect_bwd_velocity_gather_kernel<float, 3, \d, true>", "f5 advection adjoint pass C (staggered): transposed 4-point means, one velocity component per launch", 5 * w * N, 5,
         "read du of the 3 source components, read + write grad_velocity")
    note(r"advect_bwd_velocity_gather_kernel<float, 3, \d, false>", "f5 advection adjoint pass C (centred samples): transposed cell-centre means", 3 * w * N, 3,
         "read du, read + write grad_velocity")
    note(r"mac_cormack_bwd_kernel<float, 3", "f5 adjoint of the MacCormack correction pass (centred scalar / one staggered component per launch); lookup scatter via passes B / C", 15 * w * N, 15,
         "read grad_out, field, forward result, 3 velocity components; rmw grad_field, grad_fwd, 3 grad_velocity components")
    note(r"diffuse_kernel<float, true>", "f5 adjoint of explicit diffusion as a gather (no atomics), one component per launch", 3 * w * N, 3, "read grad_out, read + write grad_in")
    note(r"c2s_bwd_kernel<float", "f5 adjoint of the buoyancy resample", 3 * w * N, 3, "read grad_out component, rmw grad_s")
    return grid


def group_f32_512(ctx, dev, n, reps):
    L = 2 * math.pi
    w, N = 4, n ** 3
    grid = C.make_grid(3, C.PHIHIP_F32, 1, (n, n, n), (0, 0, 0), (L, L, L), ((0, 0),) * 3)
    rhs = torch.randn(1, n, n, n, generator=torch.Generator(device=dev).manual_seed(0), device=dev)
    rhs -= rhs.mean()
    x = torch.zeros_like(rhs)
    ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 8 * reps, 0, 0, 0), want_info=False)
    sync(dev)
    note(r"march_kernel<float, 4, \d, \d+, 1, false", "a5 initial residual r = y - A x", 3 * w * N, 3, "read x, y; write r")
    note(r"march_kernel<float, 4, \d, \d+, 2, false", "a5 CG MATVEC d = r + beta d, d.Ad", 3 * w * N, 3, "read r, d; write d")
    note(r"march_kernel<float, 4, \d, \d+, 6, false", "a5 CG UPDATE_R r -= alpha A d", 3 * w * N, 3, "read r, d; write r")
    note(r"march_kernel<float, 4, \d, \d+, 7, false", "a5 CG UPDATE_X2 x += two steps, r -= alpha A d", 5 * w * N, 5, "read x, r, d; write x, r")
    return grid


def group_f64_384(ctx, dev, n, reps):
    w, N = 8, n ** 3
    bcv = np.zeros((3, 2, 3))
    bcv[2, 1, 0] = 1.0                                        # lid: z+ wall moves along x (Lid_Driven_Cavity.ipynb cell 5 by analogy)
    grid = C.make_grid(3, C.PHIHIP_F64, 1, (n, n, n), (0, 0, 0), (1, 1, 1), ((1, 1),) * 3, bcv)
    c = (np.arange(n) + 0.5) / n
    inside = np.abs(c - 0.5) <= 0.125
    acc = ~(inside[:, None, None] & inside[None, :, None] & inside[None, None, :])
    acc_t = torch.from_numpy(acc.astype(np.uint8)).to(dev)
    flags = torch.empty(n, n, n, dtype=torch.uint8, device=dev)
    ctx.build_cellflags(grid, acc_t.data_ptr(), 0, 1, flags.data_ptr())
    shapes = [ctx.component_shape(grid, d) for d in range(3)]
    g = torch.Generator(device="cpu").manual_seed(0)
    v = [(torch.randn(1, *s, generator=g, dtype=torch.float64) * 0.01).to(dev) for s in shapes]
    v2 = [torch.empty_like(t) for t in v]
    p = torch.zeros(1, n, n, n, dtype=torch.float64, device=dev)
    solve = C.Solve(0.0, 0.0, 12, 0, 0, 0)
    dt = 0.5 / n
    for _ in range(reps):
        ctx.advect_staggered(grid, P(v), P(v), P(v2), dt)
        p.zero_()
        ctx.make_incompressible(grid, P(v2), None, flags.data_ptr(), 1, True, p.data_ptr(), 0, solve, want_info=False)
    sync(dev)
    Nf = sum(int(np.prod(s)) for s in shapes)
    note(r"advect_self_dma_kernel<double", "a1 self-advection, closed box, LDS ring filled by LDS-DMA (r5, GEN instantiation: wall constants from a table, patch elements)", 2 * w * Nf, 6, "read 3 + write 3 components")
    note(r"advect_self_tile_kernel<double, 3", "a1 self-advection, closed box, register-staged LDS tiles (r4; r5 only where the LDS-DMA kernel cannot take the grid)", 2 * w * Nf, 6, "read 3 + write 3 components")
    note(r"divergence_vec_kernel<double, 3", "a2 divergence * active + balance sums (r4: vector kernel)", w * (Nf + N) + N, 4, "read 3 components + flags, write div")
    note(r"divergence_kernel<double, 3>", "a2 divergence * active + balance sums, scalar kernel", w * (Nf + N) + N, 4, "read 3 components + flags, write div")
    note(r"march_kernel<double, 2, \d, \d+, 8, true", "a3+a5 initial residual with balance shift, flags", 4 * w * N + N, 4, "read x, y, flags; write y, r")
    note(r"march_kernel<double, 2, \d, \d+, 2, true", "a5 CG MATVEC with cell flags", 3 * w * N + N, 3, "read r, d, flags; write d")
    note(r"march_kernel<double, 2, \d, \d+, 6, true", "a5 CG UPDATE_R with cell flags", 3 * w * N + N, 3, "read r, d, flags; write r")
    note(r"march_kernel<double, 2, \d, \d+, 7, true", "a5 CG UPDATE_X2 with cell flags", 5 * w * N + N, 5, "read x, r, d, flags; write x, r")
    note(r"grad_subtract_vec_kernel<double, 3", "a6 gradient subtraction with hard_bcs flags, all components", w * (N + 2 * Nf) + N, 7, "read p, flags; read + write 3 components")
    note(r"grad_subtract_kernel<double, 3>", "a6 scalar gradient kernel (only when the vector path is not taken)", w * (N + 2 * Nf // 3) + N, 3, "read p, flags; read + write 1 component")
    return grid


def cg_case(ctx, dev, group, n):
    """ the pressure solve of a group: grid, flags pointer (0: none), right-hand side, solution vector (+ the flag tensor, kept alive by the caller) """
    L = 2 * math.pi
    if group == "f64_384":
        grid = C.make_grid(3, C.PHIHIP_F64, 1, (n, n, n), (0, 0, 0), (1, 1, 1), ((1, 1),) * 3)
        c = (np.arange(n) + 0.5) / n
        inside = np.abs(c - 0.5) <= 0.125
        acc = torch.from_numpy((~(inside[:, None, None] & inside[None, :, None] & inside[None, None, :])).astype(np.uint8)).to(dev)
        flags = torch.empty(n, n, n, dtype=torch.uint8, device=dev)
        ctx.build_cellflags(grid, acc.data_ptr(), 0, 1, flags.data_ptr())
        dtype, fptr = torch.float64, flags.data_ptr()
    else:
        grid = C.make_grid(3, C.PHIHIP_F32, 1, (n, n, n), (0, 0, 0), (L, L, L), ((0, 0),) * 3)
        dtype, fptr, flags = torch.float32, 0, None
    rhs = torch.randn(1, n, n, n, generator=torch.Generator(device=dev).manual_seed(0), device=dev, dtype=dtype)
    if flags is not None:
        rhs *= (flags & 64 != 0)
    rhs -= rhs.mean()
    x = torch.zeros_like(rhs)
    return grid, fptr, rhs, x, flags, dtype


def time_iteration(ctx, dev, grid, fptr, rhs, x, iters=100):
    x.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ctx.cg_solve(grid, fptr, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, iters, 0, 0, 0), want_info=False)
    e1.record()
    sync(dev)
    return e0.elapsed_time(e1) / iters


def tune_and_time(ctx, dev, group, n):
    """ `--write-plans` (run WITHOUT a profiler): the first-call autotune runs, its winners are read back (phihip_query_plan) together with the
    chunk the tiled advection settled on, and the untraced wall time of one CG iteration is measured -- the traced runs pin exactly these
    plans (`--plans`), so that no autotune candidate shares a kernel name with the kernel that is being profiled, and
    tools/kernel_roofline.py checks the traced per-iteration sum against the wall time of an iteration (the traced process's own, see main). """
    L = 2 * math.pi
    grid, fptr, rhs, x, flags, dtype = cg_case(ctx, dev, group, n)
    iters = 100
    ctx.cg_solve(grid, fptr, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 10, 0, 0, 0), want_info=False)       # tunes
    sync(dev)
    forced = json.loads(os.environ.get("PHIHIP_FORCE_PLANS", "{}") or "{}")      # r6: {"<family>": [rows, tpr, chunk]} -- a tile the autotune did not pick, measured
    for fam, (rows, tpr, chunk) in forced.items():                               # with the same tooling (the wide row tiles of configs[4])
        ctx.set_tuning_kernel(int(fam), int(rows), int(tpr), int(chunk))
    if forced:
        ctx.cg_solve(grid, fptr, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 10, 0, 0, 0), want_info=False)
        sync(dev)
    ms = time_iteration(ctx, dev, grid, fptr, rhs, x, iters)
    plans = {str(f): ctx.query_plan(grid, flags is not None, f) for f in (0, 1, 2, 3)}
    out = {"group": group, "size": n, "plans": plans, "untraced_ms_per_cg_iteration": ms, "iterations_timed": iters}
    if hasattr(ctx, "workspace_placement"):
        out["workspace_placement"] = ctx.workspace_placement()
    if group in ("f32_256", "f64_384") and hasattr(ctx, "query_advect_chunk"):      # (r5: the fp64 group too -- its traced average used to include the chunk candidates)
        shapes = [ctx.component_shape(grid, d) for d in range(3)]
        v = [torch.randn(1, *sh, device=dev, dtype=dtype) * 0.01 for sh in shapes]
        o = [torch.empty_like(t) for t in v]
        ctx.advect_staggered(grid, [t.data_ptr() for t in v], [t.data_ptr() for t in v], [t.data_ptr() for t in o], 0.5 * L / n)
        sync(dev)
        out["advect_chunk"] = ctx.query_advect_chunk()
    return out


def pin_plans(ctx, plans):
    ctx.set_autotune(False)
    for fam, q in plans.get("plans", {}).items():
        ctx.set_tuning_kernel(int(fam), int(q["rows"]), int(q["tpr"]), int(q["chunk"]))
    if plans.get("advect_chunk"):
        ctx.set_advect_chunk(int(plans["advect_chunk"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write-plans", default="", help="tune, time one CG iteration untraced and write the launch plans here (run without a profiler)")
    ap.add_argument("--plans", default="", help="pin the launch plans of this file (written by --write-plans) and switch the autotune off")
    ap.add_argument("--group", default="f32_256", choices=["f32_256", "f32_512", "f64_384"])
    ap.add_argument("--size", type=int, default=0, help="cells per axis (default: the group's size)")
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--manifest", default="", help="write the kernel -> moved bytes table here")
    ap.add_argument("--lib", default="")
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args()
    dev = torch.device(args.device)
    lib = C.Library(args.lib) if args.lib else C.load_default_library()
    ctx = C.Context(lib, 0)
    n = args.size or {"f32_256": 256, "f32_512": 512, "f64_384": 384}[args.group]
    if args.write_plans:
        rec = tune_and_time(ctx, dev, args.group, n)
        rec["build_id"] = lib.build_id()
        with open(args.write_plans, "w") as f:
            json.dump(rec, f, indent=1)
        return
    plans = None
    if args.plans:
        with open(args.plans) as f:
            plans = json.load(f)
        assert plans["build_id"] == lib.build_id(), (plans["build_id"], lib.build_id())
        pin_plans(ctx, plans)
    calibration_copy(dev)
    {"f32_256": group_f32_256, "f32_512": group_f32_512, "f64_384": group_f64_384}[args.group](ctx, dev, n, args.reps)
    same_process = None
    if plans is not None:
        # r6 (last session): what an iteration costs depends on the allocations that hold the workspace (cg.hip place_workspace: +-3 ... 10 % between processes at
        # identical plans) -- the wall time the traced kernel durations are checked against is therefore measured HERE, in the traced process, on the workspace the
        # traced launches used (100 more iterations of the same kernels with the same plans: they join the traced averages)
        grid, fptr, rhs, x, flags, _ = cg_case(ctx, dev, args.group, n)
        ctx.cg_solve(grid, fptr, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 4, 0, 0, 0), want_info=False)
        sync(dev)
        same_process = {"ms_per_cg_iteration": time_iteration(ctx, dev, grid, fptr, rhs, x, 100),
                        "workspace_placement": ctx.workspace_placement() if hasattr(ctx, "workspace_placement") else None}
    if args.manifest:
        with open(args.manifest, "w") as f:
            json.dump(dict(group=args.group, size=n, build_id=lib.build_id(), source_matches_tree=lib.built_from_tree(), pinned_plans=plans, same_process=same_process,
                           kernels=MANIFEST), f, indent=1)


if __name__ == "__main__":
    main()
