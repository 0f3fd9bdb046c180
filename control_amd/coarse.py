"""Coarse spaces for the two-grid form of the block-Schur preconditioner's sub-solves.

The reference's sub-solves ``(tau K + c M)^-1`` are BoomerAMG cycles (``control/control.py:
2277-2288``); north_star replaces them by Chebyshev SpMV sweeps.  On their own these need
``~ 1.6 sqrt(kappa)`` dependent sweeps per solve (80 on 256^2 P1); with a Galerkin correction on
a small coarse space in front of them the sweeps only have to cover the upper part of the
spectrum and 8 do.  The coarse space is given to the library as a prolongation matrix
``P`` (``kkt_pc_desc.p_*``): this module builds the multilinear one from the dof coordinates --
the same coordinates the sweep programs use as their tiling hint.  Host-side set-up code.
"""
from __future__ import annotations

import itertools

import numpy as np
import scipy.sparse as sp

__all__ = ["multilinear_coarse_space", "cells_for"]


def cells_for(coords, target_nodes=300):
    """Cells per axis of a coarse tensor grid over the bounding box with about ``target_nodes``
    nodes, cells as close to cubes as the box allows."""
    X = np.asarray(coords, dtype=np.float64)
    ext = X.max(axis=0) - X.min(axis=0)
    dim = X.shape[1]
    live = ext > 0
    if not live.any():
        return np.ones(dim, dtype=np.int64)
    vol = np.prod(ext[live])
    h = (vol / float(target_nodes)) ** (1.0 / live.sum())
    cells = np.ones(dim, dtype=np.int64)
    cells[live] = np.maximum(1, np.rint(ext[live] / h - 1.0).astype(np.int64))
    return cells


def multilinear_coarse_space(coords, bc_nodes=(), cells=None, target_nodes=300):
    """``P`` (n x n_c CSR, float64): row r interpolates from the corners of the coarse cell that
    holds ``coords[r]`` with multilinear weights; rows of Dirichlet dofs (``bc_nodes``) are empty;
    coarse functions no free row touches are dropped.  Rows that share their coordinates are the
    components of a vector-valued space: component k (k-th occurrence of the point) gets its own
    copy of the coarse functions.  ``cells``: cells per axis (default: ``cells_for``)."""
    X = np.ascontiguousarray(coords, dtype=np.float64)
    n, dim = X.shape
    free = np.ones(n, dtype=bool)
    free[np.asarray(bc_nodes, dtype=np.int64)] = False
    cells = cells_for(X, target_nodes) if cells is None else np.broadcast_to(
        np.asarray(cells, dtype=np.int64), (dim,)).copy()
    lo = X.min(axis=0)
    ext = np.where(X.max(axis=0) - lo > 0, X.max(axis=0) - lo, 1.0)
    # component of every row: its rank among the rows with the same coordinates
    _, inv = np.unique(X, axis=0, return_inverse=True)
    inv = np.ravel(inv)
    order = np.argsort(inv, kind="stable")
    comp = np.zeros(n, dtype=np.int64)
    same = np.concatenate([[False], inv[order][1:] == inv[order][:-1]])
    run = np.zeros(n, dtype=np.int64)
    for i in range(1, n):
        run[i] = run[i - 1] + 1 if same[i] else 0
    comp[order] = run
    ncomp = int(comp.max()) + 1
    t = (X - lo) / ext * cells                       # position in cell units
    i0 = np.minimum(np.floor(t).astype(np.int64), cells - 1)
    f = t - i0
    nodes_per_axis = cells + 1
    n_grid = int(np.prod(nodes_per_axis))
    rows, cols, vals = [], [], []
    idx = np.flatnonzero(free)
    for corner in itertools.product((0, 1), repeat=dim):
        c = np.asarray(corner)
        w = np.prod(np.where(c == 1, f[idx], 1.0 - f[idx]), axis=1)
        node = np.zeros(idx.size, dtype=np.int64)
        for k in range(dim):
            node = node * nodes_per_axis[k] + (i0[idx, k] + c[k])
        keep = w > 0.0
        rows.append(idx[keep])
        cols.append(comp[idx][keep] * n_grid + node[keep])
        vals.append(w[keep])
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    used, cols = np.unique(cols, return_inverse=True)
    P = sp.csr_matrix((vals, (rows, np.ravel(cols))), shape=(n, used.size))
    P.sum_duplicates()
    P.sort_indices()
    return P
