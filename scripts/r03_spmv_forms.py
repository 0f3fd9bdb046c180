"""Tooling: where does the generic (variable-width, row-sorted) form of kkt_spmv_rows lose time?
Times the KKT operator alone (kkt_time_apply, HIP events on the library's stream) on heat-control
systems that differ in ONE thing -- the spatial matrices -- and on the Stokes outer system:

  p1     P1 triangles, 257^2 dofs, width 7 everywhere (the templated fixed-width form)
  q2     Q2 quadrilaterals, 257^2 dofs, rows of 9 / 15 / 25 entries (generic form, one structure)
  p2     P2 triangles (one velocity component of the Stokes space), rows of <= 19 / <= 9 entries
  stokes the Stokes-control outer operator (two structures per velocity row: two launches)

    python scripts/r03_spmv_forms.py [--n_t 64] [--reps 20]
"""
import argparse
import ctypes as C
import json
import os
import sys
import types

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np

from control_amd import _lib, problems
from control_amd.blocks import instationary_blocks
from control_amd.fem import rectangle_p2p1

ap = argparse.ArgumentParser()
ap.add_argument("--n_t", type=int, default=64)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--cases", default="p1,q2,p2,stokes")
ap.add_argument("--options", default="", help="k=v,k=v passed to the library")
a = ap.parse_args()
opts = dict(kv.split("=") for kv in a.options.split(",") if kv) or None


def timed(g, label, extra=None):
    lib, h = g._lib, g.handle
    info = g.info()
    n = info["n_local"]
    d_x, d_y = C.c_void_p(), C.c_void_p()
    g._ck(lib.kkt_vec_alloc(h, C.byref(d_x)))
    g._ck(lib.kkt_vec_alloc(h, C.byref(d_y)))
    g._ck(lib.kkt_vec_upload(h, d_x, _lib.f64(problems.rng_vector(n))[1]))
    ms = C.c_float()
    g._ck(lib.kkt_time_apply(h, d_x, d_y, 5, C.byref(ms)))
    g._ck(lib.kkt_time_apply(h, d_x, d_y, a.reps, C.byref(ms)))
    t = ms.value / a.reps
    b = info["bytes_streamed"]
    out = {"case": label, "unknowns": int(n), "bytes": int(b), "launch_ms": round(t, 4),
           "GBs": round(b / t / 1e6, 1), "frac": round(b / t / 1e6 / 8000.0, 3),
           "stored_slots_over_nnz": info.get("sell_fill")}
    out.update(extra or {})
    print(json.dumps(out), flush=True)


for case in a.cases.split(","):
    if case in ("p1", "q2"):
        p = problems.heat_problem(space=case, n=256 if case == "p1" else 128, n_t=a.n_t,
                                  beta=1e-4, share=False)
        timed(problems.gpu_system(p, options=opts), case)
    elif case == "p2":
        th = rectangle_p2p1(128, 128, 2.0, 2.0)
        n2 = th.n_v // 2
        sd = types.SimpleNamespace(M=th.M_v[:n2, :n2].tocsr(), K=th.K_v[:n2, :n2].tocsr(),
                                   n_dofs=n2, coords=th.coords_v,
                                   boundary=th.boundary_v[th.boundary_v < n2])
        tau = 2.0 / (a.n_t - 1.0)
        b00, b01, b10, b11, m = instationary_blocks(sd.M, sd.K, tau, 1e-4, a.n_t, False, share=False)
        p = dict(sd=sd, tau=tau, beta=1e-4, n_t=a.n_t, CN=False, m=m, blocks=(b00, b01, b10, b11),
                 nodes=sd.boundary)
        lens = np.diff(sd.K.indptr)
        timed(problems.gpu_system(p, options=opts), case,
              {"row_lengths": {int(k): int(v) for k, v in zip(*np.unique(lens, return_counts=True))}})
    elif case == "stokes":
        p = problems.stokes_problem(n=128, n_t=a.n_t // 2, share=False)
        outer, _ = problems.stokes_gpu(p, options=opts)
        timed(outer, case)
