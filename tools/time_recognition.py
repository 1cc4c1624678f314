#!/usr/bin/env python3
"""
Cost of the PhiML-level boundary per solve, measured on the device: the assembled 7-point matrix of masked_laplace with a
box obstacle is built as a torch sparse CSR tensor ON the GPU (what a torch-based PhiML backend hands to `linear_solve`), then
    fingerprint          matrix_fingerprint: the cache key, paid by EVERY solve
    recognise (device)   recognise_laplace_stencil_torch: paid when the matrix changed (new grid, or a MOVING obstacle: every step)
    recognise (host)     the round-3 path: device -> host copy + SciPy / NumPy pass (only up to --host-max cells: it takes seconds)
are timed, one JSON line per size.   python tools/time_recognition.py --sizes 64,128,256 [--host-max 2200000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phiflow_amd import _capi as C      # noqa: E402
from phiflow_amd import linear as L     # noqa: E402


def assemble_on_device(n, dev, box):
    """ masked_laplace of a closed n^3 box with a cuboid obstacle [box[0], box[1])^3 of inactive cells: torch sparse CSR, float32, on `dev` """
    N = n ** 3
    idx = torch.arange(N, device=dev, dtype=torch.int64)
    c = [(idx // (n * n)), (idx // n) % n, idx % n]
    acc = torch.ones(N, dtype=torch.bool, device=dev)
    inside = torch.ones(N, dtype=torch.bool, device=dev)
    for a in range(3):
        inside &= (c[a] >= box[0]) & (c[a] < box[1])
    acc &= ~inside
    rows, cols, vals = [], [], []
    diag = torch.zeros(N, dtype=torch.float32, device=dev)
    strides = [n * n, n, 1]
    for a in range(3):
        for shift in (-1, 1):
            ok = (c[a] + shift >= 0) & (c[a] + shift < n)
            nb = (idx + shift * strides[a]).clamp(0, N - 1)
            couple = ok & acc & acc[nb]
            rows.append(idx[couple]); cols.append(nb[couple]); vals.append(torch.ones(int(couple.sum()), device=dev))
            diag -= couple.to(torch.float32)
    diag = torch.where(acc, diag, torch.ones_like(diag))
    rows.append(idx); cols.append(idx); vals.append(diag)
    m = torch.sparse_coo_tensor(torch.stack([torch.cat(rows), torch.cat(cols)]), torch.cat(vals), (N, N)).coalesce()
    return m.to_sparse_csr()


def timed(fn, dev, reps=3):
    fn()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="64,128,256")
    ap.add_argument("--host-max", type=int, default=2200000, help="largest cell count for the host (SciPy) recognition")
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args()
    dev = torch.device(a.device)
    for n in [int(x) for x in a.sizes.split(",")]:
        A = assemble_on_device(n, dev, (n // 3, n // 2))
        rec = {"size": n, "cells": n ** 3, "nnz": int(A.values().numel()), "device": str(dev)}
        rec["fingerprint_ms"], _ = timed(lambda: L.matrix_fingerprint(A), dev)

        def on_device():
            row, col, val = L._torch_entries(A)
            res = L.infer_resolution_torch(row, col, n ** 3)
            return L.recognise_laplace_stencil_torch(row, col, val, res)
        rec["recognise_device_ms"], d = timed(on_device, dev, reps=2)
        rec["bc"] = [list(p) for p in d["bc"]]
        rec["inactive_cells"] = int(((d["flags"] & 64) == 0).sum()) if d["flags"] is not None else 0
        if n ** 3 <= a.host_max:
            def on_host():
                import scipy.sparse as sp
                host = lambda t: t.detach().cpu().numpy()
                M = sp.csr_matrix((host(A.values()), host(A.col_indices()), host(A.crow_indices())), shape=tuple(A.shape))
                return L.recognise_laplace_stencil(M, L.infer_resolution(M))
            rec["recognise_host_ms"], dh = timed(on_host, dev, reps=1)
            assert np.array_equal(dh["flags"], d["flags"].cpu().numpy()) and list(dh["bc"]) == list(d["bc"])
        print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in rec.items()}), flush=True)
        del A
        if dev.type == "cuda":
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
