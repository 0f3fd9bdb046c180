"""Launcher-side plumbing for time-sharded systems: one process per GPU.

The data path (halo hand-offs, Krylov all-reduces) lives in ``libkkt.so``
(``csrc/comm.cpp``): RCCL over xGMI in production, or caller-provided host callbacks.
This module only bootstraps a transport for a ``MultiBlockSystem``:

* ``RcclComm``    -- production.  Rank 0 draws an ``ncclUniqueId`` and hands it to the
                    other ranks of the node through a file under ``/tmp`` keyed by the
                    launcher's pid and ``MASTER_PORT`` (all ranks of one node share both).
* ``CallbackComm``-- host-staged transport over any pair of Python functions
                    (``multiprocessing`` pipes in the tests, ``torch.distributed``/gloo).

PyTorch is imported only by the gloo rehearsal transport.
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

from . import _lib

__all__ = ["RcclComm", "RcclOrGloo", "CallbackComm", "PipeTransport", "GlooTransport", "make_comm",
           "shard_range"]


def shard_range(m, rank, world):
    lo, hi = C.c_int(), C.c_int()
    rc = _lib.load().kkt_shard_range(m, rank, world, C.byref(lo), C.byref(hi))
    if rc != 0:
        raise ValueError("bad shard arguments")
    return lo.value, hi.value


_uid_serial = [0]      # communicators created by this process (one rendezvous file each)


def _watchdog(seconds, what):
    """Bounded wait for a collective start-up that cannot be cancelled (ncclCommInitRank, the
    first all-reduce): if it has not finished after `seconds`, say so and end this process
    with a non-zero code instead of hanging the launcher."""
    import sys
    import threading

    def fire():
        print(f"[kkt] {what} did not finish within {seconds:.0f} s: giving up", file=sys.stderr,
              flush=True)
        os._exit(3)
    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


class _CommBase:
    def __init__(self, rank, world):
        self.rank, self.world = int(rank), int(world)
        self._n_attached = 0

    def attach(self, system):
        raise NotImplementedError


class RcclComm(_CommBase):
    """RCCL transport; ``attach`` creates one communicator per system."""

    def __init__(self, rank, world, rendezvous_dir=None, timeout=300.0):
        super().__init__(rank, world)
        if rendezvous_dir is None:
            # launcher pid + its start time (a recycled pid must not find stale files)
            try:
                with open(f"/proc/{os.getppid()}/stat") as f:
                    start = f.read().rsplit(")", 1)[1].split()[19]
            except (OSError, IndexError):
                start = "0"
            key = f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}_{start}"
            rendezvous_dir = os.path.join("/tmp", f"kkt_rdv_{key}")
        self._dir = rendezvous_dir
        self._timeout = timeout
        os.makedirs(self._dir, exist_ok=True)

    def _exchange_id(self, lib):
        # (numbered per process, not per object: a second transport object of the same launch
        # must not find the first one's file)
        path = os.path.join(self._dir, f"uid_{_uid_serial[0]}")
        _uid_serial[0] += 1
        if self.rank == 0:
            buf = C.create_string_buffer(128)
            rc = lib.kkt_comm_unique_id(buf)
            if rc != 0:
                raise _lib.KktError(rc, lib.kkt_last_error(None).decode())
            tmp = path + f".tmp{os.getpid()}"
            with open(tmp, "wb") as f:
                f.write(buf.raw)
            os.replace(tmp, path)          # atomic: readers see all 128 bytes or nothing
            return buf.raw
        t0 = time.time()
        while True:
            try:
                with open(path, "rb") as f:
                    raw = f.read()
                if len(raw) == 128:
                    return raw
            except FileNotFoundError:
                pass
            if time.time() - t0 > self._timeout:
                raise TimeoutError(f"no RCCL unique id at {path}")
            time.sleep(0.01)

    def attach(self, system):
        lib = system._lib
        raw = self._exchange_id(lib)
        self._n_attached += 1
        buf = C.create_string_buffer(raw, 128)
        system._ck(lib.kkt_comm_init_rccl(system.handle, buf))


class CallbackComm(_CommBase):
    """Host-staged transport.  ``allreduce(buf: ndarray, op)`` reduces in place over ranks
    (op 0 = sum, 1 = max); ``sendrecv(send, dst, n_recv, src) -> ndarray | None``."""

    def __init__(self, rank, world, allreduce, sendrecv):
        super().__init__(rank, world)
        self._ar, self._sr = allreduce, sendrecv
        self._keep = []

    def attach(self, system):
        ar, sr = self._ar, self._sr

        def c_allreduce(user, buf, n, op):
            try:
                a = np.ctypeslib.as_array(buf, shape=(n,))
                ar(a, op)
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1

        def c_sendrecv(user, send, n_send, dst, recv, n_recv, src):
            try:
                s = np.ctypeslib.as_array(send, shape=(n_send,)) if n_send > 0 else None
                r = sr(s, dst, int(n_recv), src)
                if n_recv > 0:
                    np.ctypeslib.as_array(recv, shape=(n_recv,))[:] = r
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1

        f1, f2 = _lib.ALLREDUCE_FN(c_allreduce), _lib.SENDRECV_FN(c_sendrecv)
        self._keep += [f1, f2]
        system._ck(system._lib.kkt_comm_init_callbacks(system.handle, f1, f2, None))
        self._n_attached += 1


class PipeTransport:
    """allreduce / sendrecv over ``multiprocessing`` connections (tests, few ranks).

    ``conns[r]`` is a duplex connection to rank r (None for the own rank).  Sums are
    formed in rank order on every rank, so all ranks hold bitwise-identical results.
    """

    def __init__(self, rank, world, conns):
        self.rank, self.world, self.conns = rank, world, conns

    def allreduce(self, buf, op):
        mine = np.array(buf, copy=True)
        for r in range(self.world):
            if r != self.rank:
                self.conns[r].send(mine)
        parts = [mine if r == self.rank else self.conns[r].recv() for r in range(self.world)]
        out = parts[0].copy()
        for q in parts[1:]:
            out = out + q if op == 0 else np.maximum(out, q)
        buf[:] = out

    def sendrecv(self, send, dst, n_recv, src):
        if dst is not None and dst >= 0 and send is not None:
            self.conns[dst].send(np.array(send, copy=True))
        if src is not None and src >= 0 and n_recv > 0:
            return self.conns[src].recv()
        return None


class GlooTransport:
    """allreduce / sendrecv over ``torch.distributed`` with the gloo backend (host-staged).
    A rehearsal transport: it lets ``bench.py --gpus N`` run with all ranks on ONE GPU
    (``KKT_TRANSPORT=gloo``), which RCCL refuses; the production transport is RCCL."""

    def __init__(self, rank, world):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group("gloo", rank=rank, world_size=world)
        self._torch, self._dist = torch, dist
        self.rank, self.world = rank, world

    def allreduce(self, buf, op):
        t = self._torch.from_numpy(np.ascontiguousarray(buf))
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM if op == 0
                              else self._dist.ReduceOp.MAX)
        buf[:] = t.numpy()

    def sendrecv(self, send, dst, n_recv, src):
        reqs, out = [], None
        if dst is not None and dst >= 0 and send is not None:
            reqs.append(self._dist.isend(self._torch.from_numpy(np.array(send, copy=True)), dst))
        if src is not None and src >= 0 and n_recv > 0:
            t = self._torch.empty(n_recv, dtype=self._torch.float64)
            reqs.append(self._dist.irecv(t, src))
            out = t
        for r in reqs:
            r.wait()
        return None if out is None else out.numpy()


class RcclOrGloo(_CommBase):
    """RCCL, checked when it is attached (bounded wait); the ranks agree on the outcome over a
    gloo group of the launcher's rendezvous.  If any rank could not bring RCCL up: with
    ``allow_fallback`` every rank falls back to the host-staged gloo transport and says so
    (``name``) -- the rehearsal with all ranks on one GPU, which RCCL refuses; without it every
    rank raises, so that a multi-GPU run never silently measures the host-staged transport."""

    def __init__(self, rank, world, allow_fallback=False, timeout=180.0):
        super().__init__(rank, world)
        self._rccl = RcclComm(rank, world, timeout=timeout)
        self._gloo = None
        self._allow_fallback = allow_fallback
        self._timeout = timeout
        self.name = "rccl"

    def _agree(self, ok):
        import datetime
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group("gloo", rank=self.rank, world_size=self.world,
                                    timeout=datetime.timedelta(seconds=self._timeout))
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    def attach(self, system):
        if self._gloo is not None:
            self._gloo.attach(system)
            return
        why = ""
        # ncclCommInitRank waits for every rank: if RCCL came up on some ranks only, it (or the
        # first all-reduce) would wait for ever -- bounded by a watchdog that ends the process
        dog = _watchdog(self._timeout, f"rank {self.rank}: RCCL start-up (ncclCommInitRank / first all-reduce)")
        try:
            self._rccl.attach(system)
            v = C.c_double(float(self.rank))
            system._ck(system._lib.kkt_comm_max(system.handle, C.byref(v)))
            ok = v.value == float(self.world - 1)
            if not ok:
                why = f"max over ranks gave {v.value}"
        except Exception as e:                      # noqa: BLE001 -- any failure means fallback
            ok, why = False, f"{type(e).__name__}: {e}"
        finally:
            dog.cancel()
        if self._agree(ok):
            return
        import sys
        if not self._allow_fallback:
            print(f"[kkt] rank {self.rank}: RCCL transport not usable "
                  f"({why or 'another rank failed'}); no fallback on distinct GPUs", file=sys.stderr,
                  flush=True)
            raise RuntimeError("RCCL did not start on every rank (the host-staged gloo transport is "
                               "only used when all ranks share one GPU: KKT_DEVICE, or "
                               "KKT_TRANSPORT=gloo)")
        print(f"[kkt] rank {self.rank}: RCCL transport not usable ({why or 'another rank failed'}); "
              "falling back to the host-staged gloo transport", file=sys.stderr, flush=True)
        tr = GlooTransport(self.rank, self.world)
        self._gloo = CallbackComm(self.rank, self.world, tr.allreduce, tr.sendrecv)
        self._gloo.attach(system)
        self.name = "gloo (host-staged; RCCL did not start)"


def make_comm(rank, world, local_rank=0):
    """Transport for ``bench.py``: RCCL over xGMI.  When every rank is pinned to one GPU
    (``KKT_DEVICE``, the one-GPU rehearsal) RCCL refuses and the ranks agree to fall back to the
    host-staged gloo transport; ``KKT_TRANSPORT=gloo`` selects that transport outright.  On
    distinct GPUs there is no fallback: a run whose RCCL does not start fails."""
    if os.environ.get("KKT_TRANSPORT", "rccl") == "gloo":
        tr = GlooTransport(rank, world)
        c = CallbackComm(rank, world, tr.allreduce, tr.sendrecv)
        c.name = "gloo"
        return c
    return RcclOrGloo(rank, world, allow_fallback="KKT_DEVICE" in os.environ)
