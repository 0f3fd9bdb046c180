#include "comm.hpp"

#include <dlfcn.h>
#include <string.h>

#include <string>
#include <vector>

#include "system.hpp"

namespace kkt {

// ------------------------------------------------------------ host-staged callbacks

namespace {

class CallbackComm : public Comm {
    kkt_allreduce_fn ar_;
    kkt_sendrecv_fn sr_;
    void *user_;
    std::vector<double> hs_, hr_;

   public:
    CallbackComm(kkt_allreduce_fn ar, kkt_sendrecv_fn sr, void *user)
        : ar_(ar), sr_(sr), user_(user) {}

    void allreduce_sum(double *d_buf, int n, hipStream_t s) override {
        hs_.resize(n);
        HIPCHK(hipMemcpyAsync(hs_.data(), d_buf, n * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (ar_(user_, hs_.data(), n, KKT_OP_SUM) != 0)
            fail(KKT_ERR_COMM, "allreduce callback failed");
        HIPCHK(hipMemcpyAsync(d_buf, hs_.data(), n * sizeof(double), hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
    }

    void sendrecv(const double *d_send, int64_t n_send, int dst, double *d_recv, int64_t n_recv,
                  int src, hipStream_t s) override {
        if (dst < 0) n_send = 0;
        if (src < 0) n_recv = 0;
        hs_.resize(n_send);
        hr_.resize(n_recv);
        if (n_send) {
            HIPCHK(hipMemcpyAsync(hs_.data(), d_send, n_send * sizeof(double),
                                  hipMemcpyDeviceToHost, s));
        }
        HIPCHK(hipStreamSynchronize(s));
        if (sr_(user_, hs_.data(), n_send, dst, hr_.data(), n_recv, src) != 0)
            fail(KKT_ERR_COMM, "sendrecv callback failed");
        if (n_recv) {
            HIPCHK(hipMemcpyAsync(d_recv, hr_.data(), n_recv * sizeof(double),
                                  hipMemcpyHostToDevice, s));
            HIPCHK(hipStreamSynchronize(s));
        }
    }

    void barrier(hipStream_t s) override {
        HIPCHK(hipStreamSynchronize(s));
        double z = 0.0;
        if (ar_(user_, &z, 1, KKT_OP_SUM) != 0) fail(KKT_ERR_COMM, "barrier callback failed");
    }

    double max_host(double v, hipStream_t s) override {
        HIPCHK(hipStreamSynchronize(s));
        if (ar_(user_, &v, 1, KKT_OP_MAX) != 0) fail(KKT_ERR_COMM, "max callback failed");
        return v;
    }
};

// ------------------------------------------------------------------- RCCL over xGMI

typedef void *ncclComm_p;
struct ncclUniqueIdT {
    char internal[128];
};
typedef int (*fn_GetUniqueId)(ncclUniqueIdT *);
typedef int (*fn_CommInitRank)(ncclComm_p *, int, ncclUniqueIdT, int);
typedef int (*fn_CommDestroy)(ncclComm_p);
typedef int (*fn_AllReduce)(const void *, void *, size_t, int, int, ncclComm_p, hipStream_t);
typedef int (*fn_Send)(const void *, size_t, int, int, ncclComm_p, hipStream_t);
typedef int (*fn_Recv)(void *, size_t, int, int, ncclComm_p, hipStream_t);
typedef int (*fn_Group)();
typedef const char *(*fn_ErrStr)(int);

constexpr int NCCL_FLOAT64 = 8;   // rccl.h:467
constexpr int NCCL_SUM = 0;       // rccl.h:448
constexpr int NCCL_MAX = 2;       // rccl.h:450

struct RcclApi {
    void *lib = nullptr;
    fn_GetUniqueId GetUniqueId = nullptr;
    fn_CommInitRank CommInitRank = nullptr;
    fn_CommDestroy CommDestroy = nullptr;
    fn_AllReduce AllReduce = nullptr;
    fn_Send Send = nullptr;
    fn_Recv Recv = nullptr;
    fn_Group GroupStart = nullptr, GroupEnd = nullptr;
    fn_ErrStr ErrStr = nullptr;
};

RcclApi &rccl() {
    static RcclApi api;
    if (api.lib) return api;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1",
                           "/opt/rocm/lib/librccl.so"};
    for (const char *n : names) {
        api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (api.lib) break;
    }
    if (!api.lib) fail(KKT_ERR_COMM, std::string("cannot load librccl: ") + dlerror());
    auto sym = [&](const char *n) {
        void *p = dlsym(api.lib, n);
        if (!p) fail(KKT_ERR_COMM, std::string("librccl lacks symbol ") + n);
        return p;
    };
    api.GetUniqueId = (fn_GetUniqueId)sym("ncclGetUniqueId");
    api.CommInitRank = (fn_CommInitRank)sym("ncclCommInitRank");
    api.CommDestroy = (fn_CommDestroy)sym("ncclCommDestroy");
    api.AllReduce = (fn_AllReduce)sym("ncclAllReduce");
    api.Send = (fn_Send)sym("ncclSend");
    api.Recv = (fn_Recv)sym("ncclRecv");
    api.GroupStart = (fn_Group)sym("ncclGroupStart");
    api.GroupEnd = (fn_Group)sym("ncclGroupEnd");
    api.ErrStr = (fn_ErrStr)sym("ncclGetErrorString");
    return api;
}

void nccl_check(int r, const char *what) {
    if (r != 0) fail(KKT_ERR_COMM, std::string("RCCL: ") + rccl().ErrStr(r) + " in " + what);
}

class RcclComm : public Comm {
    ncclComm_p comm_ = nullptr;
    double *d_scalar_ = nullptr;

   public:
    RcclComm(int rank, int world, const void *uid) {
        ncclUniqueIdT id;
        memcpy(&id, uid, sizeof id);
        nccl_check(rccl().CommInitRank(&comm_, world, id, rank), "ncclCommInitRank");
        d_scalar_ = dev_alloc<double>(1);
    }
    ~RcclComm() override {
        if (comm_) rccl().CommDestroy(comm_);
        if (d_scalar_) (void)hipFree(d_scalar_);
    }
    void allreduce_sum(double *d_buf, int n, hipStream_t s) override {
        nccl_check(rccl().AllReduce(d_buf, d_buf, (size_t)n, NCCL_FLOAT64, NCCL_SUM, comm_, s),
                   "ncclAllReduce");
    }
    void sendrecv(const double *d_send, int64_t n_send, int dst, double *d_recv, int64_t n_recv,
                  int src, hipStream_t s) override {
        nccl_check(rccl().GroupStart(), "ncclGroupStart");
        if (dst >= 0 && n_send)
            nccl_check(rccl().Send(d_send, (size_t)n_send, NCCL_FLOAT64, dst, comm_, s), "ncclSend");
        if (src >= 0 && n_recv)
            nccl_check(rccl().Recv(d_recv, (size_t)n_recv, NCCL_FLOAT64, src, comm_, s), "ncclRecv");
        nccl_check(rccl().GroupEnd(), "ncclGroupEnd");
    }
    void barrier(hipStream_t s) override {
        HIPCHK(hipMemsetAsync(d_scalar_, 0, sizeof(double), s));
        allreduce_sum(d_scalar_, 1, s);
        HIPCHK(hipStreamSynchronize(s));
    }
    double max_host(double v, hipStream_t s) override {
        HIPCHK(hipMemcpyAsync(d_scalar_, &v, sizeof v, hipMemcpyHostToDevice, s));
        nccl_check(rccl().AllReduce(d_scalar_, d_scalar_, 1, NCCL_FLOAT64, NCCL_MAX, comm_, s),
                   "ncclAllReduce(max)");
        double out = 0.0;
        HIPCHK(hipMemcpyAsync(&out, d_scalar_, sizeof out, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return out;
    }
};

}  // namespace

Comm *make_callback_comm(kkt_allreduce_fn ar, kkt_sendrecv_fn sr, void *user) {
    return new CallbackComm(ar, sr, user);
}
Comm *make_rccl_comm(int rank, int world, const void *uid) { return new RcclComm(rank, world, uid); }
void rccl_unique_id(void *out128) {
    ncclUniqueIdT id;
    nccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(out128, &id, sizeof id);
}

// ------------------------------------------------------------------- halo exchanges

void comm_exchange_x_halos(System &S, const double *d_x) {
    if (!S.comm) fail(KKT_ERR_STATE, "time-sharded system without a transport");
    const int up = S.rank + 1 < S.world ? S.rank + 1 : -1;
    const int dn = S.rank > 0 ? S.rank - 1 : -1;
    // forward: my last x0 block -> rank+1's halo_x0_lo
    S.comm->sendrecv(d_x + S.local_offset(0, S.n0_loc - 1), S.nx0, up, S.d_halo_x0_lo, S.nx0, dn,
                     S.stream);
    // backward: my first x1 block -> rank-1's halo_x1_hi
    S.comm->sendrecv(d_x + S.local_offset(1, 0), S.nx1, dn, S.d_halo_x1_hi, S.nx1, up, S.stream);
}

void comm_exchange_x_halos2(System &S, const double *d_x) {
    if (!S.comm) fail(KKT_ERR_STATE, "time-sharded system without a transport");
    const int up = S.rank + 1 < S.world ? S.rank + 1 : -1;
    const int dn = S.rank > 0 ? S.rank - 1 : -1;
    const int nl = S.hi - S.lo;
    if (!S.halo2_agreed) {
        // a rank sends what its neighbour's stencil reads: agree on the flags once
        for (int v = 0; v < 2; ++v)
            for (int f = 0; f < 2; ++f)
                for (int side = 0; side < 2; ++side) {
                    const double m = S.comm->max_host(S.halo2_used[v][f][side] ? 1.0 : 0.0, S.stream);
                    S.halo2_used[v][f][side] = m > 0.5;
                    if (S.halo2_used[v][f][side] && !S.d_halo2[v][f][side]) {
                        const int64_t nxh = v == 0 ? S.nx0 : S.nx1;
                        S.d_halo2[v][f][side] = dev_alloc<double>(nxh);
                        HIPCHK(hipMemset(S.d_halo2[v][f][side], 0, nxh * 8));
                    }
                }
        S.halo2_agreed = true;
    }
    for (int v = 0; v < 2; ++v) {
        const int64_t nxh = v == 0 ? S.nx0 : S.nx1;
        for (int f = 0; f < S.families; ++f) {
            if (S.halo2_used[v][f][0])   // level lo-1 of my upper neighbour is my level hi-1
                S.comm->sendrecv(d_x + S.local_offset(v, f * nl + nl - 1), nxh, up,
                                 S.d_halo2[v][f][0], nxh, dn, S.stream);
            if (S.halo2_used[v][f][1])   // level hi of my lower neighbour is my level lo
                S.comm->sendrecv(d_x + S.local_offset(v, f * nl), nxh, dn, S.d_halo2[v][f][1], nxh,
                                 up, S.stream);
        }
    }
}

void comm_exchange_row_halos(System &S, const double *d_y) {
    if (!S.comm) fail(KKT_ERR_STATE, "time-sharded system without a transport");
    const int up = S.rank + 1 < S.world ? S.rank + 1 : -1;
    const int dn = S.rank > 0 ? S.rank - 1 : -1;
    // one exchange per group of blocks with its own transform (a variable, or one family of a
    // variable): T_1 reads the first block of the rank above, T_2 the last block of the rank below
    for (const TimeGroup &g : S.time_groups) {
        const double *yb = d_y + (g.first_local_block < S.n0_loc
                                      ? (int64_t)g.first_local_block * S.nx0
                                      : (int64_t)S.n0_loc * S.nx0 +
                                            (int64_t)(g.first_local_block - S.n0_loc) * S.nx1);
        if (g.kind == 1)
            S.comm->sendrecv(yb, g.nx, dn, g.d_halo, g.nx, up, S.stream);
        else
            S.comm->sendrecv(yb + (int64_t)(g.n - 1) * g.nx, g.nx, up, g.d_halo, g.nx, dn, S.stream);
    }
}

}  // namespace kkt
