// Built-in block Schur-complement preconditioner on the GPU: the `pc_linear` closures of
// Control.Stationary.construct_pc (reference control/control.py:351-450) and
// Control.Instationary.construct_pc (control.py:1943-2440).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "comm.hpp"
#include "system.hpp"
#include "tiles.hpp"

namespace kkt {

// [emin, emax] of the Jacobi-scaled matrix D^-1 A by `steps` Lanczos steps on the device
// (spectrum.cpp); Ritz values: emin is an upper estimate of the smallest eigenvalue, emax a
// lower estimate of the largest.
struct Spectrum {
    double emin, emax;
    int steps;
};
Spectrum jacobi_spectrum(System &S, int pattern, const double *vals, const double *dinv,
                         const uint8_t *rowmask, int max_steps,
                         bool zero_mean = false);
// spectral radius of D^-1 S for the values `skew_vals` of a skew-symmetric matrix on `pattern`
// (power iteration on -(D^-1 S)^2; an estimate from below)
double jacobi_skew_radius(System &S, int pattern, const double *skew_vals, const double *dinv,
                          const uint8_t *rowmask, int max_steps, int *steps_out);

struct PcStep {
    enum Kind { ROWS, TIME, COPY, COMM, PROG, TILE, EV_RECORD, EV_WAIT, COARSE, ROWS_IL } kind;
    IlOp *d_il = nullptr;           // ROWS_IL: one IlOp per group of four time levels
    int il_groups = 0, il_slices = 0, il_w = 0;
    const double *einv = nullptr;   // COARSE: y = (x or 0) + P E^-1 P^T cr
    const double *cr = nullptr;     // COARSE: the residual that is restricted
    int lane = 0;                   // 0: the system's stream; 1: the side stream
    int ev = -1;                    // EV_RECORD / EV_WAIT: event index
    RowLaunch rows;                 // ROWS
    double *y = nullptr;            // TIME / COPY
    const double *x = nullptr;
    int tkind = 0, n = 0;
    int64_t nx = 0;                 // TIME: block length; COPY / COMM: element count
    const double *lo_halo = nullptr, *hi_halo = nullptr;   // TIME on a time shard
    int dst = -1, src = -1;         // COMM: send x to dst, receive y from src
    int nphases = 0;                // PROG: rows.d_ops holds nphases single-block RowOps
    bool granule = false;           // PROG: data-flow form (tagged granules)
    PhaseLite *d_lite = nullptr;    // PROG: compact per-phase records (data-flow form)
    int gmode = 0;                  // PROG: 0 counters, 1 data-flow fixed width, 2 data-flow any width
    TileLevel *d_levels = nullptr;  // TILE: nlevels time levels of `its` steps each
    int nlevels = 0, its = 0;
    uint32_t epoch0 = 0;            // TILE: first hand-off tag of this launch minus one
    bool fused = true;              // TILE: kernel variant with the level update
    bool clear = false;             // TILE: zero the granule buffers first
    bool coarse = false;            // TILE: two-grid levels
    const double **d_einv = nullptr;   // TILE, coarse: per level (P^T A P)^-1
    uint32_t cepoch0 = 0;           // TILE, coarse: first coarse-exchange tag of this launch minus one
};

// A device-resident pc_fn: reads the nullspace-corrected right-hand side from in(), leaves
// pc_fn(b) in out() (both n_local doubles, fixed buffers).
class PcBase {
   public:
    virtual ~PcBase() {}
    virtual void run() = 0;
    virtual double *in() = 0;
    virtual double *out() = 0;
    virtual void values_changed() {}
    virtual void check() {}
    // A persistent sweep program gave up waiting for a neighbour (bounded spins): true once,
    // with the diagnostic record in *why.  fallback_plain() then rebuilds the preconditioner as
    // plain launches (same arithmetic), so the caller can redo the work instead of failing.
    virtual bool timed_out(std::string *) { return false; }
    virtual bool fallback_plain() { return false; }
    // device word that is non-zero after a time-out (null: this preconditioner has no programs)
    virtual const unsigned *err_word() const { return nullptr; }
    // one application replayed step by step with events (kkt_time_pc_stages)
    virtual void time_stages(kkt_pc_stage_times *out);
    virtual void debug_read(unsigned long long *, int) {}
    // measurement: time the persistent programs of the next run() with events
    virtual void time_programs(float *ms, int *launches, int64_t *phases) {
        *ms = 0.f;
        *launches = 0;
        *phases = 0;
    }
};

class SchurPC : public PcBase {
   public:
    SchurPC(System &S, const kkt_pc_desc &d);
    ~SchurPC() override;
    // u = pc_fn(b) on the fixed internal vectors in_ / out_ (bc-corrected by the caller)
    void run() override;
    double *in() override { return in_; }
    double *out() override { return out_; }
    void values_changed() override;
    void check() override;   // throws if a persistent row program reported a time-out
    bool timed_out(std::string *why) override;
    bool fallback_plain() override;
    const unsigned *err_word() const override { return d_err_; }
    void time_stages(kkt_pc_stage_times *out) override;
    void debug_read(unsigned long long *out, int n) override;   // diagnostic builds (KKT_STAMPS)
    void time_programs(float *ms, int *launches, int64_t *phases) override;
    int bc_set() const { return bc_set_; }
    // degree and interval the sub-solves of a typical time level run with (given or derived)
    int schur_its() const { return schur_its_; }
    int coarse_cycles() const { return coarse_cycles_; }
    double typical_emin() const { return typical_emin_; }
    double typical_emax() const { return typical_emax_; }
    int64_t n_launches() const { return (int64_t)steps_.size(); }

   private:
    System &S_;
    kkt_pc_desc d_;
    std::vector<int32_t> m_indptr_, m_indices_, bc_idx_;
    std::vector<double> m_values_;
    int n_ = 1;                // blocks per variable (global)
    int lo_ = 0, hi_ = 1;      // owned block range
    int64_t nx_ = 0;
    int bc_set_ = -1;
    const uint8_t *mask_ = nullptr;
    int m_pat_ = -1;
    double *m_vals_ = nullptr;           // bc-assembled M (cols masked)
    double *m_dinv_ = nullptr;
    int32_t *zero_off_ = nullptr;
    double *in_ = nullptr, *out_ = nullptr;
    double *B_ = nullptr, *T_ = nullptr;             // n_ blocks each
    double *P_[3] = {nullptr, nullptr, nullptr};     // Chebyshev rotation, n_ blocks each
    std::vector<void *> owned_;                      // device allocations to free
    struct Mat {
        double *vals;
        double *dinv;
        double emin = 0.0, emax = 0.0;   // Chebyshev interval of this matrix (given or estimated)
        double eimag = 0.0;              // > 0: imaginary semi-axis of the spectrum's ellipse
        double *einv = nullptr;          // two-grid form: (P^T A P)^-1, nc x nc row-major (shared
                                         // by matrices with equal values: freed through einv_owned_)
    };
    int schur_its_ = 0;                  // degree of the sub-solves (given or derived)
    double typical_emin_ = 0.0, typical_emax_ = 0.0;
    int resolve_its(const Mat &typical);
    int64_t spectrum_steps_ = 0;         // Lanczos steps spent on estimates (reported when verbose)
    // two-grid form of the sub-solves (kkt_pc_desc.coarse_*)
    int coarse_cycles_ = 0;
    std::vector<int32_t> p_indptr_, p_indices_;
    std::vector<double> p_values_;
    std::vector<int32_t> pt_indptr_, pt_indices_;      // P^T (host copies: tile plan)
    std::vector<double> pt_values_;
    CoarseDev coarse_;
    std::vector<double *> einv_owned_;
    double *R_ = nullptr;                // residual of a cycle (one block)
    void build_coarse();
    double *coarse_inverse(const double *vals);       // (P^T A P)^-1 on the device
    void emit_coarse(const double *r, const double *x_in, double *x_out, const double *einv);
    std::map<std::pair<const double *, uint64_t>, Mat> mats_;
    double *h_u0_ = nullptr, *h_u1_ = nullptr, *h_t_ = nullptr;   // one-block halos
    std::vector<PcStep> steps_;
    struct Segment {
        size_t first = 0, last = 0;
        bool comm = false;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
    };
    std::vector<Segment> segments_;
    bool use_graph_ = true;
    // Side lane: work that does not depend on the sweep in flight (the mass solves and the
    // right-hand-side products of later time levels) runs on a second stream while the
    // latency-bound sweep program occupies a quarter of the wave slots.
    bool use_lanes_ = false;
    hipStream_t side_ = nullptr;
    std::vector<hipEvent_t> events_;
    int cur_lane_ = 0;
    int n_events_ = 0;
    int emit_record(int lane);          // returns the event index
    void emit_wait(int lane, int ev);
    // persistent row programs (kernels.hip, pc_row_program)
    bool use_programs_ = true;
    int prog_wpw_ = 0, prog_nwg_ = 0;
    int32_t *d_dep_ = nullptr;
    unsigned *d_flags_ = nullptr, *d_err_ = nullptr;
    bool prog_granule_ = false;
    bool prog_lowreg_ = false;   // counter form at three waves per SIMD (> 2 048 slices)
    int prog_mode_ = 0;   // 0 counter form, 1 pc_row_program_g, 2 pc_row_program_gw
    unsigned long long *d_g0_ = nullptr, *d_g1_ = nullptr;
    size_t granule_words_ = 0;
    void fuse_programs();
    bool setup_row_programs();
    bool legacy_tried_ = false, legacy_ok_ = false;
    // tile form (tile_kernels.hip): levels of the sweeps as the emit functions recorded them
    struct SweepLevel {
        size_t first = 0, last = 0;     // steps_[first, last) are the level's single-block steps
        TileLevel lev{};
        std::vector<TileCoef> coef;     // steps 2 .. its
        int its = 0;
        bool eligible = false;
        const double *b_after = nullptr;   // the right-hand side after the update (B_i)
        bool coarse = false;               // two-grid level: coef holds sweeps 1 .. its
        const double *einv = nullptr;      // its (P^T A P)^-1
    };
    std::vector<SweepLevel> sweep_levels_;
    TilePlan tile_plan_;
    bool tile_tried_ = false, tile_ok_ = false;
    size_t tile_lds_checked_ = 0;       // dynamic LDS bytes residency was checked for
    // coarse corrections inside the tile program (kernels.hpp, TileCoarseDev)
    TileCoarseDev h_tile_coarse_{};     // host copy (device pointers inside)
    TileCoarseDev *d_tile_coarse_ = nullptr;
    bool tile_coarse_ok_ = false;
    uint32_t tile_cepoch_cursor_ = 0;
    bool build_tile_coarse();
    unsigned long long *d_tg_[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<void *> tile_owned_;    // coefficient tables of the current program
    uint32_t tile_epoch_cursor_ = 0;    // hand-off tags handed out to the launches of one application
    bool tile_cleared_ = false;
    bool prepare_tiles();
    bool fuse_tile_run(size_t k, size_t e, std::vector<PcStep> &out);

    struct Term {
        const double *vals;
        const double *x;
    };
    struct Lin {   // y = mask(ca * sum_t A_t x_t + cy * yin + cz * z)
        std::vector<Term> terms;
        double *y;
        double ca = 1.0, cy = 0.0, cz = 0.0;
        const double *yin = nullptr, *z = nullptr;
        // optional fused first Chebyshev step of the solve that follows: y2 = c3 * dinv * y
        double *y2 = nullptr;
        const double *dinv = nullptr;
        double c3 = 0.0;
    };
    struct Cheb {
        const double *vals;   // nullptr: no matrix term (first step)
        const double *dinv, *b, *pk, *pkm1;
        double *y;
        double c1, c2, c3, post1 = 1.0, post2 = 1.0;
    };
    void build();
    void clear_program();
    const double *block_vals(int q, int i, int j) const;
    Mat schur_matrix(const double *base_vals, double c);
    void emit_lin(const std::vector<Lin> &ops);
    void emit_cheb(const std::vector<Cheb> &ops);
    double *il_P_[3] = {nullptr, nullptr, nullptr};
    size_t il_cap_ = 0;
    void emit_time(double *y, const double *x, int kind, int n, const double *lo_halo = nullptr,
                   const double *hi_halo = nullptr);
    void emit_comm(const double *send, int dst, double *recv, int src);
    // its-step Jacobi-Chebyshev solves of `count` independent systems in lock step
    struct Solve {
        const double *vals, *dinv, *b;
        double *out;
        double post1 = 1.0, post2 = 1.0;
    };
    // batched solves on one matrix with the iterates of four levels interleaved (kernels.hpp IlOp)
    bool emit_solves_interleaved(const std::vector<Solve> &sv, int its, double emin, double emax);
    void emit_solves(const std::vector<Solve> &sv, int its, double emin, double emax,
                     double *const P[3], int64_t pstride, bool first_done = false,
                     std::vector<TileCoef> *coef_out = nullptr, double eimag = 0.0,
                     const Mat *mat = nullptr);
    // the sweep step "b -= A u_prev (masked), then solve": the update and the first
    // Chebyshev step share one launch
    void emit_update_and_solve(Lin upd, const Solve &sv, int its, double emin, double emax,
                               double eimag = 0.0, const Mat *mat = nullptr);
    void emit_coarse_solve(const Lin *upd, const Solve &sv, const Mat &F);
    // position of the transposed entry of every stored entry of the preconditioner's sparsity
    // structure (device array, built on first use; null when the structure is not symmetric)
    const int32_t *transpose_positions();
    int32_t *d_tpos_ = nullptr;
    bool tpos_tried_ = false;
    void push_rows(std::vector<RowOp> &r);
    void build_stationary();
    void build_BE();
    void build_CN();
    void replay(size_t first, size_t last);
};

// Preconditioner of the incompressible (Stokes / Navier-Stokes) control systems:
// the pc_fn closures of Stationary.incompressible_linear_solve (reference
// control/control.py:986-1085) and Instationary.incompressible_linear_solve, BE branch
// (control.py:4515-4687).  u_0 = `inner_its` GMRES iterations on the velocity KKT system
// (a second handle with its own block-Schur preconditioner); then per pressure block
// h = s2 (sB B u_0 - b_1), m = K_p^-1 h (Jacobi-Chebyshev in place of one BoomerAMG cycle),
// g = C m with C the pressure-space commutator block system (a third handle), and
// u_1 = M_p^-1 g (20 Jacobi-Chebyshev steps, control.py:957-971).
class StokesPC : public PcBase {
   public:
    StokesPC(System &outer, System &inner, System &commutator, const kkt_pc_stokes_desc &d);
    ~StokesPC() override;
    void run() override;
    double *in() override { return in_; }
    double *out() override { return out_; }
    void check() override;
    bool timed_out(std::string *why) override { return inner_.pc && inner_.pc->timed_out(why); }
    bool fallback_plain() override { return inner_.pc && inner_.pc->fallback_plain(); }
    const unsigned *err_word() const override { return inner_.pc ? inner_.pc->err_word() : nullptr; }

   private:
    System &S_, &inner_, &comm_;
    int n_ = 1;   // pressure blocks per variable
    bool cn_ = false;
    int64_t nv_ = 0, np_ = 0;
    double sB_ = 1.0, s2_ = 1.0;
    int kp_its_ = 0, mp_its_ = 0;
    double kp_emin_ = 0, kp_emax_ = 0, mp_emin_ = 0, mp_emax_ = 0;
    struct DevMat {
        int pat = -1;
        double *vals = nullptr, *dinv = nullptr;
    } B_, Kp_, Mp_;
    double *in_ = nullptr, *out_ = nullptr, *h_ = nullptr, *m_ = nullptr, *g_ = nullptr;
    double *P_[3] = {nullptr, nullptr, nullptr};
    double *halo_a_ = nullptr, *halo_b_ = nullptr;   // CN on time shards: neighbour blocks of the T scans
    std::vector<void *> owned_;
    // a step of a pressure-space chain: a batched row launch, or -- two-grid K_p solve -- the
    // Galerkin correction x_out = x_in + P E^-1 P^T r of every block
    struct ChainStep {
        RowLaunch L;
        bool coarse = false;
        const double *r = nullptr, *x_in = nullptr;
        double *x_out = nullptr;
    };
    std::vector<RowLaunch> lin_;
    std::vector<ChainStep> kp_steps_, mp_steps_;
    // the two Chebyshev chains (hundreds of small launches on fixed buffers) replayed as graphs
    struct Chain {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        bool failed = false;
    } kp_chain_, mp_chain_;
    void run_chain(Chain &c, const std::vector<ChainStep> &steps);
    // two-grid K_p solve
    CoarseDev kp_coarse_;
    double *kp_einv_ = nullptr, *kp_r_ = nullptr, *kp_x0_ = nullptr;
    int kp_cycles_ = 0;
    void build_kp_coarse(const kkt_pc_stokes_desc &d);
    void emit_kp_two_grid(int its, double emin, double emax, const double *b, double *out);
    DevMat upload(int64_t nrows, int64_t ncols, const int32_t *ip, const int32_t *ix,
                  const double *v, bool want_dinv);
    void emit_cheb(std::vector<ChainStep> &dst, const DevMat &A, int its, double emin, double emax,
                   const double *b, double *out);
};

}  // namespace kkt
