// Transport for time-sharded systems: tiny all-reduces for the Krylov inner
// products, nearest-neighbour vector hand-offs for the time coupling.  The reference has
// no such layer: its only parallelism is PETSc's spatial decomposition over MPI
// (preconditioner.py:706); time sharding is new functionality.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/kkt.h"

namespace kkt {

struct System;

class Comm {
   public:
    virtual ~Comm() {}
    // in-place sum over ranks of n doubles in device memory, ordered on `s`
    virtual void allreduce_sum(double *d_buf, int n, hipStream_t s) = 0;
    // send to `dst` (or -1) and receive from `src` (or -1), device buffers, ordered on `s`
    virtual void sendrecv(const double *d_send, int64_t n_send, int dst, double *d_recv,
                          int64_t n_recv, int src, hipStream_t s) = 0;
    virtual void barrier(hipStream_t s) = 0;
    virtual double max_host(double v, hipStream_t s) = 0;
};

Comm *make_callback_comm(kkt_allreduce_fn ar, kkt_sendrecv_fn sr, void *user);
Comm *make_rccl_comm(int rank, int world, const void *unique_id_128);
void rccl_unique_id(void *out128);

// x halos for the operator: x0 block hi-1 -> rank+1 (its halo_x0_lo), x1 block lo -> rank-1
void comm_exchange_x_halos(System &S, const double *d_x);
// two-family shards (outer incompressible system): per (variable, family) the level below and /
// or above, as the stencil needs them (flags agreed over the ranks at the first call)
void comm_exchange_x_halos2(System &S, const double *d_x);
// raw-row halos for the CN transform: rho0 block lo -> rank-1 (halo_r0_hi),
// rho1 block hi-1 -> rank+1 (halo_r1_lo)
void comm_exchange_row_halos(System &S, const double *d_y);

}  // namespace kkt
