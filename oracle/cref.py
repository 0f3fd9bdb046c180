"""ctypes front-end of the C/OpenMP restatement (oracle/csrc/kkt_ref.c).  TEST
INFRASTRUCTURE: imported only by tests/ and by bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

from . import kkt_oracle as ko

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libkktref.so")
i32p, f64p, u8p = C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_uint8)


class Csr(C.Structure):
    _fields_ = [("nrows", C.c_int32), ("indptr", i32p), ("indices", i32p), ("vals", f64p)]


class RefSys(C.Structure):
    _fields_ = [("m", C.c_int32), ("nx", C.c_int32), ("nb", C.c_int32 * 4),
                ("bi", i32p * 4), ("bj", i32p * 4), ("blk", C.POINTER(Csr) * 4),
                ("mask", u8p)]


class RefPc(C.Structure):
    _fields_ = [("n_t", C.c_int32), ("nx", C.c_int32), ("tau", C.c_double),
                ("beta", C.c_double), ("epsilon", C.c_double),
                ("M", C.POINTER(Csr)), ("Mt", C.POINTER(Csr)), ("mdinv", f64p),
                ("D10", C.POINTER(Csr)), ("S10", C.POINTER(Csr)), ("S01", C.POINTER(Csr)),
                ("F", C.POINTER(Csr)), ("Fdinv", C.POINTER(f64p)),
                ("G", C.POINTER(Csr)), ("Gdinv", C.POINTER(f64p)), ("mask", u8p),
                ("mass_its", C.c_int32), ("schur_its", C.c_int32),
                ("mass_emin", C.c_double), ("mass_emax", C.c_double),
                ("schur_emin", C.c_double), ("schur_emax", C.c_double),
                ("coarse_cycles", C.c_int32), ("nc", C.c_int32),
                ("P", C.POINTER(Csr)), ("PT", C.POINTER(Csr)),
                ("FEinv", C.POINTER(f64p)), ("GEinv", C.POINTER(f64p))]


def usable_cores(cap=16):
    """Cores this process may really use: scheduler affinity capped by the cgroup CPU quota (a
    one-GPU box grants a share of a much larger host; OpenMP's default of one thread per visible
    core then oversubscribes the quota and the spin-waiting team crawls)."""
    n = len(os.sched_getaffinity(0))
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt[0] != "max":
            n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
    except (OSError, ValueError, IndexError):
        pass
    return max(1, min(n, cap))


def load(build=True):
    if build and not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", HERE])
    lib = C.CDLL(LIB)
    if "OMP_NUM_THREADS" not in os.environ:
        try:
            C.CDLL("libgomp.so.1").omp_set_num_threads(usable_cores())
        except OSError:
            pass
    lib.ref_kkt_apply.argtypes = [C.POINTER(RefSys), f64p, f64p]
    lib.ref_pc_apply_BE.argtypes = [C.POINTER(RefPc), f64p, f64p]
    lib.ref_gmres_BE.argtypes = [C.POINTER(RefSys), C.POINTER(RefPc), f64p, f64p, C.c_int,
                                 C.c_double, C.c_double, C.c_double, C.c_int, f64p, C.c_int,
                                 C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.ref_gmres_BE.restype = C.c_int
    return lib


class CRef:
    """BE heat-control system + block-Schur preconditioner in the C restatement."""

    def __init__(self, blocks, m, nx, nodes, M, n_t, tau, beta, mass, schur, epsilon=1.0e-3,
                 coarse=None):
        """``coarse``: ``(P, cycles)`` -- the two-grid form of the Schur sub-solves."""
        self.lib = load()
        self._keep = []
        self.m, self.nx = m, nx
        mask = np.zeros(nx, dtype=np.uint8)
        mask[nodes] = 1
        self._mask = mask
        cache = {}

        def csr(A):
            if id(A) in cache:
                return cache[id(A)]
            A = sp.csr_matrix(A)
            A.sort_indices()
            ip = np.ascontiguousarray(A.indptr, dtype=np.int32)
            ix = np.ascontiguousarray(A.indices, dtype=np.int32)
            va = np.ascontiguousarray(A.data, dtype=np.float64)
            self._keep += [ip, ix, va, A]
            c = Csr(A.shape[0], ip.ctypes.data_as(i32p), ix.ctypes.data_as(i32p),
                    va.ctypes.data_as(f64p))
            cache[id(A)] = c
            return c

        def arr(items):
            a = (Csr * len(items))(*items)
            self._keep.append(a)
            return a
        self.sys = RefSys()
        self.sys.m, self.sys.nx = m, nx
        self.sys.mask = mask.ctypes.data_as(u8p)
        for q, blk in enumerate(blocks):
            ent = [(i, j, A) for (i, j), A in blk.items() if A is not None]
            bi = np.array([e[0] for e in ent], dtype=np.int32)
            bj = np.array([e[1] for e in ent], dtype=np.int32)
            self._keep += [bi, bj]
            self.sys.nb[q] = len(ent)
            self.sys.bi[q] = bi.ctypes.data_as(i32p)
            self.sys.bj[q] = bj.ctypes.data_as(i32p)
            self.sys.blk[q] = arr([csr(e[2]) for e in ent])
        b00, b01, b10, b11 = blocks
        n = n_t
        shift = tau / beta**0.5

        def coef(i):
            return 0.0 if i == 0 else ((epsilon**0.5) * shift if i == n - 1 else shift)
        scache = {}

        def schur_mat(blk, c):
            key = (id(blk), c)
            if key not in scache:
                At = ko.assemble_with_bcs(blk if c == 0.0 else blk + c * M, nodes)
                dinv = np.ascontiguousarray(1.0 / At.diagonal())
                einv = None
                if coarse is not None:
                    einv = np.ascontiguousarray(ko.coarse_inverse(At, ko.CoarseSpace(*coarse)))
                self._keep += [At, dinv, einv]
                scache[key] = (csr(At), dinv, einv)
            return scache[key]
        Mt = ko.assemble_with_bcs(M, nodes)
        mdinv = np.ascontiguousarray(1.0 / Mt.diagonal())
        self._keep += [Mt, mdinv]
        F = [schur_mat(b10[(i, i)], coef(i)) for i in range(n)]
        G = [schur_mat(b01[(i, i)], coef(i)) for i in range(n)]
        pc = RefPc()
        pc.n_t, pc.nx, pc.tau, pc.beta, pc.epsilon = n, nx, tau, beta, epsilon
        one = lambda A: C.pointer(csr(A))   # noqa: E731
        pc.M, pc.Mt = one(M), one(Mt)
        pc.mdinv = mdinv.ctypes.data_as(f64p)
        pc.D10 = arr([csr(b10[(i, i)]) for i in range(n)])
        pc.S10 = arr([csr(b10[(i, i - 1)]) if i >= 1 else csr(M) for i in range(n)])
        pc.S01 = arr([csr(b01[(i, i + 1)]) if i + 1 < n else csr(M) for i in range(n)])
        pc.F = arr([f[0] for f in F])
        pc.G = arr([g[0] for g in G])
        fd = (f64p * n)(*[f[1].ctypes.data_as(f64p) for f in F])
        gd = (f64p * n)(*[g[1].ctypes.data_as(f64p) for g in G])
        self._keep += [fd, gd]
        pc.Fdinv, pc.Gdinv = fd, gd
        pc.mask = mask.ctypes.data_as(u8p)
        pc.mass_its, pc.mass_emin, pc.mass_emax = mass
        pc.schur_its, pc.schur_emin, pc.schur_emax = schur[:3]     # (real intervals: BE heat control)
        if coarse is not None:
            Pm = sp.csr_matrix(coarse[0])
            Pm.sort_indices()
            PT = sp.csr_matrix(Pm.T)
            PT.sort_indices()
            pc.coarse_cycles, pc.nc = int(coarse[1]), Pm.shape[1]
            pc.P, pc.PT = one(Pm), one(PT)
            fe = (f64p * n)(*[f[2].ctypes.data_as(f64p) for f in F])
            ge = (f64p * n)(*[g[2].ctypes.data_as(f64p) for g in G])
            self._keep += [fe, ge]
            pc.FEinv, pc.GEinv = fe, ge
        self.pc = pc

    def mult(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        self.lib.ref_kkt_apply(C.byref(self.sys), x.ctypes.data_as(f64p), y.ctypes.data_as(f64p))
        return y

    def pc_apply(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64)
        u = np.empty_like(b)
        self.lib.ref_pc_apply_BE(C.byref(self.pc), b.ctypes.data_as(f64p), u.ctypes.data_as(f64p))
        return u

    def gmres(self, b, x0, restart=10, rtol=1e-6, atol=0.0, divtol=1e4, max_it=50):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.ascontiguousarray(x0, dtype=np.float64).copy()
        hist = np.zeros(max_it + 8)
        nh, reason = C.c_int(), C.c_int()
        its = self.lib.ref_gmres_BE(C.byref(self.sys), C.byref(self.pc), b.ctypes.data_as(f64p),
                                    x.ctypes.data_as(f64p), restart, rtol, atol, divtol, max_it,
                                    hist.ctypes.data_as(f64p), len(hist), C.byref(nh),
                                    C.byref(reason))
        return x, its, reason.value, hist[:min(nh.value, len(hist))]
