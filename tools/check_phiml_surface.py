#!/usr/bin/env python3
"""
Prints every mismatch between the PhiML / PhiFlow surface that `phiflow_amd/linear.py` (HipPhimlBackend) and
`phiflow_amd/phiml_plugin.py` rely on and the `phiml` / `phi` packages importable HERE.

Why: `phiml` (pinned `>=1.14.0` in /root/reference/setup.py:41) is neither vendored in the reference nor installable in the build
image, so the Backend signatures, the `SolveResult` fields and the `Solve` attributes used by the Level-B boundary are recollections
([PHIML-RECALL]). On a machine that has PhiML this script is the first thing to run:

    python tools/check_phiml_surface.py            # real packages; exit code 1 if anything differs, 0 if clean, 2 if phiml is absent
    python tools/check_phiml_surface.py --fake     # the in-tree test double (tests/fake_phiml): checks that this tool and the double agree

Reference call sites that define the expectations: phi/physics/fluid.py:145-156,165 (Solve.with_preprocessing, copy_with(rank_deficiency=1),
math.solve_linear, jit_compile_linear), phi/__init__.py:41-63 (BACKENDS, init_backend), phi/torch/flow.py:15-35, phi/field/_resample.py:259
(math.grid_sample), tests/commit/physics/test_fluid.py:21-22,67 (`with backend:`, `backend.supports(Backend.jacobian)`).
"""
import argparse
import inspect
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# name -> parameter names in order (after self); a trailing '?' marks parameters that may be absent or defaulted
BACKEND_METHODS = {
    "linear_solve": ["method", "lin", "y", "x0", "rtol", "atol", "max_iter", "pre", "matrix_offset"],
    "conjugate_gradient": ["lin", "y", "x0", "rtol", "atol", "max_iter", "pre", "matrix_offset"],
    "grid_sample": ["grid", "coordinates", "extrapolation"],
}
SOLVE_RESULT_FIELDS = ["method", "x", "residual", "iterations", "function_evaluations", "converged", "diverged", "message"]
SOLVE_ATTRS = ["method", "rel_tol", "abs_tol", "x0", "max_iterations", "suppress", "preprocess_y", "preprocess_y_args", "rank_deficiency"]
MATH_FUNCTIONS = ["tensor", "wrap", "expand", "pack_dims", "unpack_dim", "stack", "unstack", "batch", "spatial", "channel", "dual", "instance",
                  "copy_with", "solve_linear", "Solve", "NotConverged", "Diverged", "ConvergenceException"]
EXTRAPOLATIONS = ["PERIODIC", "BOUNDARY", "ZERO", "ZERO_GRADIENT", "ConstantExtrapolation"]
PHI_NAMES = {"phi.field": ["Field"], "phi.geom": ["Box", "Sphere", "UniformGrid"],
             "phi.physics.fluid": ["make_incompressible", "Obstacle", "_get_obstacles_for", "_pressure_extrapolation"],
             "phi.physics.advect": ["semi_lagrangian", "mac_cormack", "advect", "euler"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fake", action="store_true", help="check tests/fake_phiml instead of installed packages")
    args = ap.parse_args()
    if args.fake:
        sys.path.insert(0, os.path.join(ROOT, "tests", "fake_phiml"))
    try:
        import phiml
        from phiml import backend as pb, math
        from phiml.math import extrapolation as ext
    except Exception as err:
        print(f"phiml is not importable here ({type(err).__name__}: {err}); nothing checked. Level B stays EXPERIMENTAL (INTEGRATION.md §3).")
        return 2
    print(f"checking phiml {getattr(phiml, '__version__', '?')} at {os.path.dirname(phiml.__file__)}")
    problems = []

    def note(msg):
        problems.append(msg)
        print("MISMATCH:", msg)

    # ---- phiml.backend ----
    if not isinstance(getattr(pb, "BACKENDS", None), list):
        note("phiml.backend.BACKENDS is not a list (make_phiml_backend appends the 'hip' backend to it)")
    Backend = getattr(pb, "Backend", None)
    if Backend is None:
        note("phiml.backend.Backend missing")
    else:
        for name, want in BACKEND_METHODS.items():
            fn = getattr(Backend, name, None)
            if fn is None:
                note(f"Backend.{name} missing")
                continue
            have = [p for p in inspect.signature(fn).parameters if p != "self"]
            if have != want:
                note(f"Backend.{name}{tuple(have)} != expected {tuple(want)}")
        for name in ("name", "__enter__", "__exit__"):
            if not hasattr(Backend, name):
                note(f"Backend.{name} missing (`with backend:` protocol, tests/commit/physics/test_fluid.py:21-22)")
    sr = getattr(pb, "SolveResult", None)
    if sr is None:
        note("phiml.backend.SolveResult missing (HipPhimlBackend.linear_solve returns the bare solution then)")
    else:
        fields = list(getattr(sr, "_fields", [])) or [p for p in inspect.signature(sr).parameters]
        if fields != SOLVE_RESULT_FIELDS:
            note(f"SolveResult fields {fields} != expected {SOLVE_RESULT_FIELDS}")
    try:
        torch_be = None
        try:
            from phiml.backend.torch import TORCH as torch_be          # noqa: N811
        except Exception:
            torch_be = getattr(getattr(pb, "torch", None), "TORCH", None)
        if torch_be is None:
            note("no torch backend singleton at phiml.backend.torch.TORCH (HipPhimlBackend would subclass the abstract Backend: no tensor ops)")
        elif Backend is not None and not isinstance(torch_be, Backend):
            note("phiml.backend.torch.TORCH is not a Backend instance")
    except Exception as err:
        note(f"probing the torch backend failed: {err}")

    # ---- phiml.math ----
    for name in MATH_FUNCTIONS:
        if not hasattr(math, name):
            note(f"phiml.math.{name} missing")
    Solve = getattr(math, "Solve", None)
    if Solve is not None:
        try:
            s = Solve("CG", 1e-5, 0)
            for a in SOLVE_ATTRS:
                if not hasattr(s, a):
                    note(f"Solve.{a} missing")
            if not hasattr(s, "with_preprocessing"):
                note("Solve.with_preprocessing missing (phi/physics/fluid.py:146)")
            if hasattr(math, "copy_with"):
                s2 = math.copy_with(s, rank_deficiency=1)
                if getattr(s2, "rank_deficiency", None) != 1:
                    note("copy_with(solve, rank_deficiency=1) does not set the attribute (fluid.py:148)")
        except Exception as err:
            note(f"Solve('CG', 1e-5, 0) failed: {err}")
    for name in EXTRAPOLATIONS:
        if not hasattr(ext, name):
            note(f"phiml.math.extrapolation.{name} missing")

    # ---- phi ----
    import importlib
    for mod, names in PHI_NAMES.items():
        try:
            m = importlib.import_module(mod)
        except Exception as err:
            note(f"{mod} not importable: {err}")
            continue
        for name in names:
            if not hasattr(m, name):
                note(f"{mod}.{name} missing")

    # ---- the backend itself ----
    try:
        sys.path.insert(0, ROOT)
        from phiflow_amd import linear
        cls_src = inspect.getsource(linear.make_phiml_backend)
        if "matrix_offset" not in cls_src:
            note("HipPhimlBackend.linear_solve lost its matrix_offset parameter")
    except Exception as err:
        note(f"phiflow_amd.linear not importable: {err}")

    print(f"{len(problems)} mismatch(es)")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
