// extern "C" boundary of libkkt (include/kkt.h).  No C++ exception leaves this file.
#include <cmath>
#include <cstring>
#include <string>

#include "comm.hpp"
#include "pc.hpp"
#include "system.hpp"

using namespace kkt;

struct kkt_system {
    System S;
};

static std::string g_create_error;

#define KKT_TRY(h, ...)                                   \
    if (!(h)) return KKT_ERR_ARG;                         \
    System &S = (h)->S;                                   \
    try {                                                 \
        if (hipSetDevice(S.device) != hipSuccess)         \
            fail(KKT_ERR_HIP, "hipSetDevice failed");     \
        __VA_ARGS__;                                      \
        return KKT_OK;                                    \
    } catch (const Error &e) {                            \
        S.err = e.msg;                                    \
        return e.code;                                    \
    } catch (const std::exception &e) {                   \
        S.err = e.what();                                 \
        return KKT_ERR_STATE;                             \
    }

extern "C" {

int kkt_create(kkt_handle *out, int device_id) {
    if (!out) return KKT_ERR_ARG;
    *out = nullptr;
    try {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
            fail(KKT_ERR_HIP, "no HIP device available: libkkt has no CPU path");
        if (device_id < 0 || device_id >= ndev) fail(KKT_ERR_ARG, "device id out of range");
        HIPCHK(hipSetDevice(device_id));
        kkt_system *h = new kkt_system();
        h->S.device = device_id;
        HIPCHK(hipStreamCreateWithFlags(&h->S.stream, hipStreamNonBlocking));
        *out = h;
        return KKT_OK;
    } catch (const Error &e) {
        g_create_error = e.msg;
        return e.code;
    } catch (const std::exception &e) {
        g_create_error = e.what();
        return KKT_ERR_STATE;
    }
}

int kkt_destroy(kkt_handle h) {
    if (!h) return KKT_OK;
    delete h;
    return KKT_OK;
}

const char *kkt_last_error(kkt_handle h) { return h ? h->S.err.c_str() : g_create_error.c_str(); }

int kkt_set_option(kkt_handle h, const char *key, const char *value) {
    KKT_TRY(h, {
        static const char *known[] = {"sell_r", "sell_sort", "no_graph", "persistent", "prog_mode",
                                      "prog_waves", "prog_steps", "tile_depth", "tile_waves",
                                      "lanes", "lane_chunks", "kernarg_ops", "shared_rows",
                                      "verbose", "stamps", "tile_poll_delay", "tile_unfused",
                                      "debug_drop_handoff", "stage_timers", "sell_sigma", "ragged_switch", "ragged_xcd", "apply_xcd", "pc_xcd",
                                      "interleave"};
        if (!key || !value) fail(KKT_ERR_ARG, "null option");
        bool ok = false;
        for (const char *k : known) ok = ok || std::strcmp(k, key) == 0;
        if (!ok) fail(KKT_ERR_ARG, std::string("unknown option: ") + key);
        S.options[key] = value;
    });
}

int kkt_set_tile_coordinates(kkt_handle h, int dim, int64_t n, const double *coords) {
    KKT_TRY(h, {
        if (dim < 1 || dim > 3 || n < 1 || !coords) fail(KKT_ERR_ARG, "bad tile coordinates");
        for (size_t i = 0; i < (size_t)n * dim; ++i)      // (they are sorted: no NaN)
            if (!std::isfinite(coords[i])) fail(KKT_ERR_ARG, "tile coordinates must be finite");
        S.tile_coords.assign(coords, coords + (size_t)n * dim);
        S.tile_dim = dim;
    });
}

int kkt_set_layout(kkt_handle h, int n00, int n11, int64_t nx0, int64_t nx1, int CN, int s00,
                   int s11) {
    KKT_TRY(h, S.set_layout(n00, n11, nx0, nx1, CN, s00, s11));
}

int kkt_shard_range(int m, int rank, int world, int *lo, int *hi) {
    if (m < 1 || world < 1 || rank < 0 || rank >= world || !lo || !hi) return KKT_ERR_ARG;
    // contiguous, as even as possible, earlier ranks take the remainder
    const int q = m / world, r = m % world;
    *lo = rank * q + (rank < r ? rank : r);
    *hi = *lo + q + (rank < r ? 1 : 0);
    return KKT_OK;
}

int kkt_set_shard(kkt_handle h, int rank, int world) { KKT_TRY(h, S.set_shard(rank, world)); }
int kkt_set_shard_families(kkt_handle h, int rank, int world, int families) {
    KKT_TRY(h, S.set_shard(rank, world, families));
}

int kkt_add_block(kkt_handle h, int q, int i, int j, int64_t nrows, int64_t ncols,
                  const int32_t *indptr, const int32_t *indices, const double *values,
                  int64_t share_id) {
    KKT_TRY(h, S.add_block(q, i, j, nrows, ncols, indptr, indices, values, share_id));
}

int kkt_update_block_values(kkt_handle h, int q, int i, int j, const double *values) {
    KKT_TRY(h, S.update_block_values(q, i, j, values));
}

int kkt_set_bc(kkt_handle h, int k, int64_t n, const int32_t *idx, double alpha) {
    KKT_TRY(h, S.set_bc(k, n, idx, alpha));
}

int kkt_set_const_nullspace(kkt_handle h, int k, double alpha) {
    KKT_TRY(h, S.set_const_ns(k, alpha));
}

int kkt_finalize(kkt_handle h) { KKT_TRY(h, S.finalize()); }

int kkt_set_pc_schur(kkt_handle h, const kkt_pc_desc *desc) {
    KKT_TRY(h, {
        if (!desc) fail(KKT_ERR_ARG, "null descriptor");
        S.pc.reset();
        S.pc_cb = nullptr;
        S.pc.reset(new SchurPC(S, *desc));
        S.pc_stale = false;
    });
}

int kkt_set_pc_stokes(kkt_handle h, kkt_handle inner, kkt_handle commutator,
                      const kkt_pc_stokes_desc *desc) {
    KKT_TRY(h, {
        if (!desc || !inner || !commutator) fail(KKT_ERR_ARG, "null argument");
        S.pc.reset();
        S.pc_cb = nullptr;
        S.pc.reset(new StokesPC(S, inner->S, commutator->S, *desc));
    });
}

int kkt_set_pc_callback(kkt_handle h, kkt_pc_callback fn, void *user) {
    KKT_TRY(h, {
        if (!fn) fail(KKT_ERR_ARG, "null callback");
        S.pc.reset();
        S.pc_cb = fn;
        S.pc_cb_user = user;
    });
}

int kkt_set_pc_identity(kkt_handle h) {
    KKT_TRY(h, {
        S.pc.reset();
        S.pc_cb = nullptr;
    });
}

int kkt_set_krylov(kkt_handle h, int type, int pc_side, int restart, double rtol, double atol,
                   double divtol, int max_it) {
    KKT_TRY(h, {
        if (type != KKT_KSP_GMRES && type != KKT_KSP_FGMRES && type != KKT_KSP_MINRES)
            fail(KKT_ERR_ARG, "linear_solver must be gmres, fgmres or minres");
        if (restart < 1 || max_it < 0 || rtol < 0 || atol < 0) fail(KKT_ERR_ARG, "bad KSP options");
        S.ksp.type = type;
        S.ksp.pc_side = pc_side;
        S.ksp.restart = restart;
        S.ksp.rtol = rtol;
        S.ksp.atol = atol;
        S.ksp.divtol = divtol > 0 ? divtol : 1.0e4;
        S.ksp.max_it = max_it;
    });
}

// ---- host-array variants: stage through temporary device vectors
namespace {
struct TmpVec {
    System &S;
    double *p;
    explicit TmpVec(System &S_) : S(S_), p(S_.new_vec()) {}
    ~TmpVec() { (void)hipFree(p); }
};
void up(System &S, double *d, const double *h) {
    HIPCHK(hipMemcpyAsync(d, h, S.n_local * 8, hipMemcpyHostToDevice, S.stream));
    HIPCHK(hipStreamSynchronize(S.stream));
}
void down(System &S, const double *d, double *h) {
    HIPCHK(hipMemcpyAsync(h, d, S.n_local * 8, hipMemcpyDeviceToHost, S.stream));
    HIPCHK(hipStreamSynchronize(S.stream));
}
}  // namespace

int kkt_apply(kkt_handle h, const double *x, double *y) {
    KKT_TRY(h, {
        if (!x || !y) fail(KKT_ERR_ARG, "null vector");
        if (!S.finalized) fail(KKT_ERR_STATE, "system not finalized");
        TmpVec dx(S), dy(S);
        up(S, dx.p, x);
        S.apply(dx.p, dy.p);
        down(S, dy.p, y);
    });
}

int kkt_pc_apply(kkt_handle h, const double *x, double *y) {
    KKT_TRY(h, {
        if (!x || !y) fail(KKT_ERR_ARG, "null vector");
        if (!S.finalized) fail(KKT_ERR_STATE, "system not finalized");
        TmpVec dx(S), dy(S);
        up(S, dx.p, x);
        S.pc_apply(dx.p, dy.p);
        if (S.pc) {
            // a sweep program that timed out (on any rank of a time shard: the decision is
            // collective) is replaced by plain launches and the application redone (same
            // arithmetic)
            std::string why;
            if (S.pc_timed_out_agreed(&why)) {
                if (!S.pc_fallback_plain(why)) fail(KKT_ERR_HIP, why);
                S.pc_apply(dx.p, dy.p);
                if (S.pc_timed_out_agreed(&why)) fail(KKT_ERR_HIP, why);
            }
        }
        down(S, dy.p, y);
        if (S.pc_cb_failed) {
            S.pc_cb_failed = false;
            fail(KKT_ERR_CALLBACK, "Error encountered in preconditioner callback");
        }
    });
}

int kkt_solve(kkt_handle h, const double *b, double *u, int *its, int *reason, double *rnorm,
              double *hist, int hist_cap, int *hist_len) {
    KKT_TRY(h, {
        if (!b || !u) fail(KKT_ERR_ARG, "null vector");
        if (!S.finalized) fail(KKT_ERR_STATE, "system not finalized");
        TmpVec db(S), du(S);
        up(S, db.p, b);
        up(S, du.p, u);
        S.solve(db.p, du.p, its, reason, rnorm, hist, hist_cap, hist_len);
        down(S, du.p, u);
    });
}

// ---- device-resident variants
int64_t kkt_local_size(kkt_handle h) { return h ? h->S.n_local : -1; }

int kkt_vec_alloc(kkt_handle h, double **d_vec) {
    KKT_TRY(h, {
        if (!d_vec) fail(KKT_ERR_ARG, "null out pointer");
        if (!S.finalized) fail(KKT_ERR_STATE, "system not finalized");
        *d_vec = S.new_vec();
        S.sync();
    });
}
int kkt_vec_free(kkt_handle h, double *d_vec) {
    KKT_TRY(h, {
        if (d_vec) HIPCHK(hipFree(d_vec));
    });
}
int kkt_vec_upload(kkt_handle h, double *d_vec, const double *host) {
    KKT_TRY(h, up(S, d_vec, host));
}
int kkt_vec_download(kkt_handle h, const double *d_vec, double *host) {
    KKT_TRY(h, down(S, d_vec, host));
}
int kkt_apply_device(kkt_handle h, const double *d_x, double *d_y) {
    KKT_TRY(h, S.apply(d_x, d_y));
}
int kkt_pc_apply_device(kkt_handle h, const double *d_x, double *d_y) {
    KKT_TRY(h, S.pc_apply(d_x, d_y));
}
int kkt_solve_device(kkt_handle h, const double *d_b, double *d_u, int *its, int *reason,
                     double *rnorm, double *hist, int hist_cap, int *hist_len) {
    KKT_TRY(h, S.solve(d_b, d_u, its, reason, rnorm, hist, hist_cap, hist_len));
}
int kkt_sync(kkt_handle h) { KKT_TRY(h, S.sync()); }

static void time_loop(System &S, bool pc, const double *d_x, double *d_y, int reps, float *ms) {
    if (reps < 1 || !ms) fail(KKT_ERR_ARG, "bad timing arguments");
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, S.stream));
    for (int r = 0; r < reps; ++r) {
        if (pc)
            S.pc_apply(d_x, d_y);
        else
            S.apply(d_x, d_y);
    }
    HIPCHK(hipEventRecord(e1, S.stream));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipEventElapsedTime(ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
}
int kkt_time_apply(kkt_handle h, const double *d_x, double *d_y, int reps, float *ms) {
    KKT_TRY(h, time_loop(S, false, d_x, d_y, reps, ms));
}
int kkt_time_pc_apply(kkt_handle h, const double *d_x, double *d_y, int reps, float *ms) {
    KKT_TRY(h, time_loop(S, true, d_x, d_y, reps, ms));
}

int kkt_time_pc_sweeps(kkt_handle h, const double *d_x, double *d_y, float *ms, int *launches,
                       int64_t *phases) {
    KKT_TRY(h, {
        if (!d_x || !d_y || !ms || !launches || !phases) fail(KKT_ERR_ARG, "null argument");
        S.pc_apply_timed(d_x, d_y, ms, launches, phases);
    });
}

int kkt_get_stage_times(kkt_handle h, kkt_stage_times *out) {
    KKT_TRY(h, {
        if (!out) fail(KKT_ERR_ARG, "null argument");
        *out = S.stage_times;
    });
}

int kkt_time_pc_stages(kkt_handle h, const double *d_x, double *d_y, kkt_pc_stage_times *out) {
    KKT_TRY(h, {
        if (!d_x || !d_y || !out) fail(KKT_ERR_ARG, "null argument");
        S.pc_apply_timed_stages(d_x, d_y, out);
    });
}

int kkt_debug_set_steplock(kkt_handle h, const kkt_steplock *lock) {
    KKT_TRY(h, {
        if (!lock) {
            S.steplock = kkt_steplock{};
        } else {
            if (lock->n_steps < 1 || lock->restart < 1 || !lock->V || !lock->h || !lock->v_next)
                fail(KKT_ERR_ARG, "incomplete step-lock description");
            S.steplock = *lock;
        }
    });
}

int kkt_get_info(kkt_handle h, kkt_info *info) {
    KKT_TRY(h, {
        if (!info) fail(KKT_ERR_ARG, "null info");
        *info = S.info;
    });
}

// diagnostic builds only (make EXTRA=-DKKT_STAMPS): per-workgroup cycle sums of the
// persistent row program; not part of include/kkt.h
int kkt_debug_prog_stats(kkt_handle h, unsigned long long *out, int n) {
    KKT_TRY(h, {
        if (!S.pc) fail(KKT_ERR_STATE, "no built-in preconditioner");
        S.pc->debug_read(out, n);
    });
}

// ---- multi-GPU transport
int kkt_comm_unique_id(void *id_out_128) {
    if (!id_out_128) return KKT_ERR_ARG;
    try {
        rccl_unique_id(id_out_128);
        return KKT_OK;
    } catch (const Error &e) {
        g_create_error = e.msg;
        return e.code;
    }
}
int kkt_comm_init_rccl(kkt_handle h, const void *uid) {
    KKT_TRY(h, {
        if (!uid) fail(KKT_ERR_ARG, "null unique id");
        S.comm.reset(make_rccl_comm(S.rank, S.world, uid));
    });
}
int kkt_comm_init_callbacks(kkt_handle h, kkt_allreduce_fn ar, kkt_sendrecv_fn sr, void *user) {
    KKT_TRY(h, {
        if (!ar || !sr) fail(KKT_ERR_ARG, "null transport callbacks");
        S.comm.reset(make_callback_comm(ar, sr, user));
    });
}
int kkt_comm_barrier(kkt_handle h) {
    KKT_TRY(h, {
        if (S.comm)
            S.comm->barrier(S.stream);
        else
            S.sync();
    });
}
int kkt_comm_max(kkt_handle h, double *v) {
    KKT_TRY(h, {
        if (!v) fail(KKT_ERR_ARG, "null value");
        if (S.comm) *v = S.comm->max_host(*v, S.stream);
    });
}

}  // extern "C"
