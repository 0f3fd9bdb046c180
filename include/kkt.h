/*
 * libkkt -- MI355X-native all-at-once KKT solver: C-ABI drop-in boundary.
 *
 * The reference (sleveque/control) has no native code and no FFI; its boundary for this
 * path is the Python class preconditioner/preconditioner.py:216 `MultiBlockSystem` and
 * its method `.solve()` (preconditioner.py:337-786).  Every entry point below cites the
 * reference lines whose work it takes over.  The Python mirror of that class
 * (control_amd/multiblock.py) binds these symbols with ctypes; INTEGRATION.md shows the
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ types, no exceptions cross the boundary
 *   - every function returns KKT_OK (0) or a negative KKT_ERR_* code; the message is
 *     available from kkt_last_error()
 *   - host arrays are caller-owned; the library copies during the call and never keeps
 *     a host pointer after returning (pc callback excepted, see kkt_set_pc_callback)
 *   - a handle is bound to one GPU and is not thread-safe
 *   - vectors are fp64; the flat KKT vector is the n_blocks_00 blocks of variable 0
 *     followed by the n_blocks_11 blocks of variable 1, block k at a contiguous offset
 *     (preconditioner.py:286-287)
 */
#ifndef KKT_H
#define KKT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kkt_system *kkt_handle;

enum {
    KKT_OK = 0,
    KKT_ERR_ARG = -1,       /* bad argument / call order */
    KKT_ERR_HIP = -2,       /* HIP runtime error */
    KKT_ERR_STATE = -3,     /* object not in the state the call needs */
    KKT_ERR_CALLBACK = -4,  /* a host preconditioner callback reported failure */
    KKT_ERR_COMM = -5       /* multi-GPU transport error */
};

/* quadrants of the 2x2 block system (preconditioner.py:303) */
enum { KKT_Q00 = 0, KKT_Q01 = 1, KKT_Q10 = 2, KKT_Q11 = 3 };

/* Krylov methods (`solver_parameters["linear_solver"]`, preconditioner.py:733) */
enum { KKT_KSP_GMRES = 0, KKT_KSP_FGMRES = 1, KKT_KSP_MINRES = 2 };
/* `solver_parameters["pc_side"]` (preconditioner.py:735-736); DEFAULT = the method's own */
enum { KKT_PC_SIDE_DEFAULT = -1, KKT_PC_LEFT = 0, KKT_PC_RIGHT = 1 };

/* PETSc KSPConvergedReason values the reference tests against (preconditioner.py:769) */
enum {
    KKT_CONVERGED_RTOL = 2,
    KKT_CONVERGED_ATOL = 3,
    KKT_CONVERGED_HAPPY_BREAKDOWN = 5,
    KKT_DIVERGED_ITS = -3,
    KKT_DIVERGED_DTOL = -4,
    KKT_DIVERGED_BREAKDOWN = -5,
    KKT_DIVERGED_INDEFINITE_PC = -8,   /* minres: r.Br < 0 */
    KKT_DIVERGED_NANORINF = -9,
    KKT_DIVERGED_INDEFINITE_MAT = -10  /* minres: Lanczos r.Br < 0 after the first step */
};

/* ------------------------------------------------------------------ life cycle */

/* One system on GPU `device_id`.  Replaces the MultiBlockSystem object
 * (preconditioner.py:216-335). */
int kkt_create(kkt_handle *out, int device_id);
int kkt_destroy(kkt_handle h);
/* Message of the last failing call on `h` (or of the last failing kkt_create if NULL). */
const char *kkt_last_error(kkt_handle h);

/* Execution options: which kernel form runs, not what is computed (every form of a step
 * performs the same arithmetic in the same order; tests toggle them to assert that).  Set a
 * key before the call that reads it: storage keys before kkt_set_layout / kkt_add_block,
 * preconditioner keys before kkt_set_pc_schur.  Unknown keys are rejected.
 *   "sell_r"      "1" | "2"      rows per lane of the SELL-64R storage (default 2)
 *   "sell_sort"   "0" | "1"      row-sorted storage for ragged structures (default 1)
 *   "ragged_switch" "0"          operator apply on ragged structures (P2 / Stokes blocks) with the
 *                                slot loop instead of the width-switched kernel
 *   "ragged_xcd"  "0"            ... in dispatch order instead of the XCD-aware workgroup order
 *   "apply_xcd"   "1"            XCD-aware order for the fixed-width operator launches too
 *                                (measured slower; off)
 *   "pc_xcd"      "0"            batched preconditioner steps in dispatch order instead of the
 *                                XCD-aware workgroup order
 *   "interleave"  "0"            batched mass solves with one vector per time level instead of
 *                                the iterates of four levels interleaved
 *   "no_graph"    "1"            replay the preconditioner as plain launches, no hipGraph
 *   "persistent"  "0"            time sweeps as one launch per step (no sweep programs)
 *   "prog_mode"   "auto" | "tile" | "dataflow" | "flags" | "w"   sweep-program form (auto, the
 *                                default: tile where it fits, else dataflow, else flags;
 *                                "dataflow": the row programs only, data-flow form preferred)
 *   "prog_waves"  "1".."8"       waves per workgroup of the dataflow / flags forms
 *   "prog_steps"  "0"            dataflow form without compact STEP records
 *   "tile_depth"  "1".."16"      SpMV steps per hand-off of the tile form (default: modelled)
 *   "tile_waves"  "1".."8"       waves per workgroup of the tile form (default 8)
 *   "stage_timers" "1"           HIP events around the stages of every Krylov iteration
 *                                (kkt_get_stage_times)
 *   "lanes", "lane_chunks", "kernarg_ops", "shared_rows", "verbose"   diagnostics
 * The library never reads the process environment: a key that was never set has its default.
 * (The Python mirror forwards KKT_<KEY> variables of developer scripts as explicit calls.) */
int kkt_set_option(kkt_handle h, const char *key, const char *value);

/* Optional hint for the tile form of the preconditioner's sweep programs: coordinates of the
 * dofs of the spatial block the sweeps run on (variable 0: n = N_x rows, `dim` = 1..3 doubles
 * each, row-major; the components of a vector-valued space carry their node's coordinates --
 * in Firedrake: the interpolated SpatialCoordinate of the space).  With them the rows are cut
 * into tiles by coordinate bisection (boxes) instead of bisection of the sparsity graph, whose
 * cuts are slanted in meshes with diagonal edges: smaller rings per hand-off (256^2 P1: 630
 * instead of 784 rows at depth 7; 64^3 P1: 764 instead of 1 030).  Speed only: results do not
 * depend on the partition.  Before kkt_set_pc_schur. */
int kkt_set_tile_coordinates(kkt_handle h, int dim, int64_t n, const double *coords);

/* ------------------------------------------------------------------ definition */

/* Block counts and spatial sizes (preconditioner.py:217-222, 276-302).
 * sub_n_blocks_* = -1 means None.  Must precede every other definition call. */
int kkt_set_layout(kkt_handle h, int n_blocks_00, int n_blocks_11,
                   int64_t nx0, int64_t nx1, int CN,
                   int sub_n_blocks_00_0, int sub_n_blocks_11_0);

/* Time sharding (new functionality): this handle owns block rows
 * [lo, hi) of BOTH variables, where [lo, hi) is rank's slice of n_blocks_00 ==
 * n_blocks_11 rows split evenly over `world` ranks.  Vectors passed to every later call
 * are the local shard [x0_lo..x0_hi-1, x1_lo..x1_hi-1].  Only blocks of owned rows may be
 * added.  Omit for a single GPU. */
int kkt_set_shard(kkt_handle h, int rank, int world);
/* The same for systems whose flat blocks are `families` runs of time levels per variable
 * (2: the outer incompressible system -- velocity blocks (v, zeta), pressure blocks (mu, p),
 * control.py:3654-3673): the rank owns levels [lo, hi) of every family, local vectors are
 * [family 0 levels lo..hi-1, family 1 levels lo..hi-1] per variable.  With Crank-Nicolson the
 * sub-block split of the time transforms (sub_n_blocks_*_0, preconditioner.py:471-525) must be
 * the family boundary. */
int kkt_set_shard_families(kkt_handle h, int rank, int world, int families);
/* Row range owned by `rank` of `world` for `m` block rows (pure host arithmetic). */
int kkt_shard_range(int m, int rank, int world, int *lo, int *hi);

/* One assembled block A_ij of a quadrant as CSR (replaces `assemble(block_ij)`,
 * preconditioner.py:305-328; the adapter feeds `petscmat.getValuesCSR()`).
 * Column indices must be sorted within a row.  Blocks with the same share_id >= 0
 * share one copy of the values on the device (time-invariant operators, "mode S");
 * share_id < 0 gives the block its own copy ("mode G", what the reference stores).
 * Blocks with identical sparsity structure always share the index arrays. */
int kkt_add_block(kkt_handle h, int quadrant, int i, int j,
                  int64_t nrows, int64_t ncols,
                  const int32_t *indptr, const int32_t *indices,
                  const double *values, int64_t share_id);
/* New values on the stored structure of a block (re-linearisation in a Picard loop,
 * control.py:3377-3590); valid after kkt_finalize.  A built-in preconditioner whose matrices
 * are sums with block values is marked stale and rebuilt once, on the device, at its next
 * application -- not per updated block. */
int kkt_update_block_values(kkt_handle h, int quadrant, int i, int j,
                            const double *values);

/* DirichletBCNullspace(bcs, alpha) on flat block k (k < n_blocks_00: variable 0,
 * otherwise variable 1): preconditioner.py:158-197. */
int kkt_set_bc(kkt_handle h, int block_k, int64_t n_idx, const int32_t *idx,
               double alpha);
/* ConstantNullspace(alpha) on flat block k: preconditioner.py:133-155. */
int kkt_set_const_nullspace(kkt_handle h, int block_k, double alpha);

/* Freeze the definition and build the device data (HBM layout: DESIGN.md). */
int kkt_finalize(kkt_handle h);

/* ------------------------------------------------------------- preconditioner */

enum { KKT_PC_STATIONARY = 0, KKT_PC_INSTATIONARY_BE = 1, KKT_PC_INSTATIONARY_CN = 2 };

/* Built-in block Schur-complement preconditioner: the closures built by
 * Control.Stationary.construct_pc (control.py:351-450) and
 * Control.Instationary.construct_pc (control.py:1943-2440; CN 1995-2189, BE 2191-2438).
 * It reads block_10 / block_01 from the blocks already added.  Mass solves are
 * `mass_its` Jacobi-Chebyshev steps on [mass_emin, mass_emax] (control.py:1967-1982;
 * mass_its == 0: one Jacobi application, control.py:1984-1991).  The hypre sub-solves of
 * the reference are replaced by `schur_its` Jacobi-Chebyshev steps on
 * [schur_emin, schur_emax] (BASELINE.json north_star).  A hand-set interval much wider than the
 * spectrum is harmless for symmetric blocks and harmful for blocks with convection: outside the
 * interval's ellipse the Chebyshev polynomial grows with the imaginary part, and a wide interval
 * has a weak normalisation -- give the matrix's own bounds (or let the library estimate them). */
typedef struct kkt_pc_desc {
    int kind;            /* KKT_PC_* */
    int n_t;             /* time levels (ignored for STATIONARY) */
    double tau;          /* time step (ignored for STATIONARY) */
    double beta;         /* regularisation parameter */
    double epsilon;      /* BE final-time scaling, control.py:2836 (1e-3) */
    int64_t nx;          /* spatial dofs of the mass matrix */
    const int32_t *m_indptr;   /* mass matrix `self._M_v` as CSR */
    const int32_t *m_indices;
    const double *m_values;
    int64_t n_bc;        /* homogeneous Dirichlet dofs (bcs_v = bcs_zeta) */
    const int32_t *bc_idx;
    int mass_its;
    double mass_emin, mass_emax;
    int schur_its;       /* -1: 1.6 sqrt(emax / emin) of a typical time level's matrix */
    double schur_emin, schur_emax;   /* schur_emin <= 0: per matrix, from a Lanczos estimate of its
                                        Jacobi-scaled spectrum on the device (spectrum.cpp) */
    double schur_eimag;  /* > 0: the Jacobi-scaled spectrum of the sub-solve matrices lies in the
                            ellipse with real semi-axis (schur_emax - schur_emin) / 2 and imaginary
                            semi-axis schur_eimag around their mid-point (forward operators with a
                            convection term, control.py:1887-1896: the blocks are not symmetric);
                            the sweeps keep their three-term form with the coefficients of that
                            ellipse (Manteuffel 1977).  0: real interval.  With schur_emin <= 0 it
                            is estimated per matrix too (symmetric part for the interval, spectral
                            radius of the skew part for the semi-axis). */
    /* Two-grid form of the Schur sub-solves (coarse_cycles == 0: off).  A sub-solve is
     * `coarse_cycles` times [x += P (P^T A P)^-1 P^T (b - A x); `schur_its` Jacobi-Chebyshev
     * sweeps on [schur_emin, schur_emax] from x] -- the reference calls BoomerAMG here
     * (control.py:2277-2288); the sweeps keep the shape north_star prescribes and only have to
     * cover the part of the spectrum the coarse space does not see (8 sweeps on [emax / 30, emax]
     * instead of 80 on the whole spectrum of 256^2 P1).  P: nx x n_coarse CSR, rows of Dirichlet
     * dofs empty (control_amd.coarse.multilinear_coarse_space builds it from dof coordinates).
     * The Galerkin matrices are formed and inverted at set-up, one per distinct sub-solve matrix.
     * With schur_emin <= 0: emax from the matrix, emin = emax / 30; schur_its = -1: 8. */
    int coarse_cycles;
    int64_t n_coarse;
    const int32_t *p_indptr, *p_indices;
    const double *p_values;
} kkt_pc_desc;

int kkt_set_pc_schur(kkt_handle h, const kkt_pc_desc *desc);

/* Preconditioner of the incompressible control systems: the pc_fn closures of
 * Stationary.incompressible_linear_solve (control.py:986-1085) and of
 * Instationary.incompressible_linear_solve (BE control.py:4515-4687, CN control.py:4318-4513).  `h` is the outer system
 * (variable 0 = velocity blocks v then zeta, variable 1 = pressure blocks mu then p); `inner`
 * is the velocity KKT system with its own preconditioner and KSP options already set
 * (the reference runs 5 GMRES iterations, control.py:1005-1010); `commutator` is the
 * pressure-space block system block_**_int_p (control.py:976-984, 3818-3820).  Both handles
 * must stay alive while `h` uses them.  B = -(div v, q) is the unscaled divergence block. */
typedef struct kkt_pc_stokes_desc {
    int n_p_blocks;          /* pressure blocks per variable (1 stationary, n_t BE, n_t-1 CN) */
    int cn;                  /* 1: Crank-Nicolson branch -- T_2 / T_1 on tau B u_0 before b_1 is
                                subtracted and their inverses after the scaling (control.py:4407-4428) */
    int64_t nv, np;          /* dofs of one velocity / pressure block */
    double b_scale;          /* tau (instationary, control.py:4577) or 1 */
    double post_scale;       /* 1 / tau^2 (control.py:4596-4601) or 1 */
    const int32_t *b_indptr, *b_indices;      /* B: np x nv */
    const double *b_values;
    const int32_t *kp_indptr, *kp_indices;    /* K_p: np x np */
    const double *kp_values;
    const int32_t *mp_indptr, *mp_indices;    /* M_p: np x np */
    const double *mp_values;
    int kp_its;              /* Jacobi-Chebyshev steps replacing the BoomerAMG cycle on K_p;
                                -1: the degree of the inner system's sub-solves (inner sub-solves in
                                two-grid form: 5 sqrt(kappa) of K_p's own non-zero spectrum, <= 600) */
    double kp_emin, kp_emax; /* kp_emin <= 0: lower bound of the inner sub-solves, upper bound
                                estimated from K_p (two-grid inner sub-solves: both ends estimated
                                from K_p, constants deflated) */
    int mp_its;              /* control.py:957-971: 20 (0: one Jacobi application, :973-979) */
    double mp_emin, mp_emax;
    /* Two-grid form of the K_p solve (0 cycles: the plain polynomial above): kp_coarse_cycles x
     * [Galerkin correction on the coarse space P_p (np x kp_n_coarse, CSR; its columns must sum to
     * the constant vector, which K_p annihilates: the Galerkin matrix is inverted with the
     * constants deflated, E + (trace E / n_c^2) 1 1^T); kp_its sweeps on [kp_emin, kp_emax] from
     * the corrected iterate].  Needs kp_its >= 1 and explicit bounds. */
    int kp_coarse_cycles;
    int64_t kp_n_coarse;
    const int32_t *kp_p_indptr, *kp_p_indices;
    const double *kp_p_values;
} kkt_pc_stokes_desc;
int kkt_set_pc_stokes(kkt_handle h, kkt_handle inner, kkt_handle commutator,
                      const kkt_pc_stokes_desc *desc);

/* Arbitrary user `pc_fn(u_0, u_1, b_0, b_1)` (preconditioner.py:337-345, 623-627) on host
 * arrays: slow path kept for API parity (`P=` of every *_solve, control.py:3257-3258).
 * The library downloads b, calls `fn`, uploads u.  Non-zero return -> the solve fails
 * with KKT_ERR_CALLBACK ("Error encountered in PETSc solve", preconditioner.py:771-772). */
typedef int (*kkt_pc_callback)(void *user, const double *b_0, const double *b_1,
                               double *u_0, double *u_1);
int kkt_set_pc_callback(kkt_handle h, kkt_pc_callback fn, void *user);
/* Default pc_fn: u = b (preconditioner.py:342-345). */
int kkt_set_pc_identity(kkt_handle h);

/* ---------------------------------------------------------------------- solve */

/* KSP options (preconditioner.py:732-748): type, side, GMRES restart, tolerances.
 * divtol <= 0 selects PETSc's default 1e4. */
int kkt_set_krylov(kkt_handle h, int type, int pc_side, int restart,
                   double rtol, double atol, double divtol, int max_it);

/* y = A x: MultiBlockSystemMatrix.mult (preconditioner.py:375-543), host arrays. */
int kkt_apply(kkt_handle h, const double *x, double *y);
/* y = P^-1 x: Preconditioner.apply (preconditioner.py:562-656), host arrays. */
int kkt_pc_apply(kkt_handle h, const double *x, double *y);
/* The whole of MultiBlockSystem.solve after the matrices exist (preconditioner.py:
 * 658-766): corrected initial guess and right-hand side, Krylov loop, corrected
 * solution.  `u` holds the initial guess on entry and the solution on return.
 * hist (may be NULL) receives the monitored residual norms, iteration 0 first. */
int kkt_solve(kkt_handle h, const double *b, double *u,
              int *its, int *reason, double *rnorm,
              double *hist, int hist_cap, int *hist_len);

/* ---------------------------------------------------- device-resident variants */

/* Vectors that stay in HBM between calls (benchmarks; callers with resident data). */
int64_t kkt_local_size(kkt_handle h);             /* doubles in one local KKT vector */
int kkt_vec_alloc(kkt_handle h, double **d_vec);
int kkt_vec_free(kkt_handle h, double *d_vec);
int kkt_vec_upload(kkt_handle h, double *d_vec, const double *host);
int kkt_vec_download(kkt_handle h, const double *d_vec, double *host);
int kkt_apply_device(kkt_handle h, const double *d_x, double *d_y);
int kkt_pc_apply_device(kkt_handle h, const double *d_x, double *d_y);
int kkt_solve_device(kkt_handle h, const double *d_b, double *d_u,
                     int *its, int *reason, double *rnorm,
                     double *hist, int hist_cap, int *hist_len);
int kkt_sync(kkt_handle h);

/* `reps` back-to-back kkt_apply_device / kkt_pc_apply_device launches timed with HIP
 * events on the library's own stream; *ms = total elapsed milliseconds. */
int kkt_time_apply(kkt_handle h, const double *d_x, double *d_y, int reps, float *ms);
int kkt_time_pc_apply(kkt_handle h, const double *d_x, double *d_y, int reps, float *ms);
/* One preconditioner application with HIP events around every persistent sweep program of the
 * built-in preconditioner: *ms = their summed duration, *launches / *phases = how many programs
 * and dependent phases ran (0 when the preconditioner has no such programs).  Measurement only. */
int kkt_time_pc_sweeps(kkt_handle h, const double *d_x, double *d_y, float *ms, int *launches,
                       int64_t *phases);

/* Per-stage GPU time of the last kkt_solve* with gmres / fgmres (option "stage_timers" = "1"):
 * HIP events on the library's stream between the stages of every iteration, summed over the
 * solve.  operator = kkt_apply incl. its halo exchange; pc = Preconditioner.apply; orth =
 * classical Gram-Schmidt (dots, updates, norm) without its all-reduces; allreduce = the
 * all-reduces of the inner products (time-sharded handles; includes the wait for the slowest
 * rank); other = residual set-up, normalisation, solution update, host round trips. */
typedef struct kkt_stage_times {
    double operator_ms, pc_ms, orth_ms, allreduce_ms, other_ms, total_ms;
    int64_t iterations, operator_applies, pc_applies;
} kkt_stage_times;
int kkt_get_stage_times(kkt_handle h, kkt_stage_times *out);
/* One application of the built-in block-Schur preconditioner replayed step by step with HIP
 * events: sweeps = the persistent sweep programs (or, without them, the single-block steps of
 * the time sweeps), batched = the steps over all time levels at once (mass solves, products,
 * time transforms), comm = the hand-offs between ranks (time-sharded handles: includes the
 * wait for the neighbour's pipeline stage).  Works on time-sharded handles (collective: every
 * rank calls it).  Measurement only. */
typedef struct kkt_pc_stage_times {
    double sweeps_ms, batched_ms, comm_ms, total_ms;
    int64_t sweep_launches, sweep_phases, batched_launches, comm_steps;
} kkt_pc_stage_times;
int kkt_time_pc_stages(kkt_handle h, const double *d_x, double *d_y, kkt_pc_stage_times *out);

/* Step-locked parity hook (tests): while set, kkt_solve / kkt_solve_device with gmres or
 * fgmres replace their Krylov basis v_0 .. v_it by the caller's vectors before inner step `it`
 * of global step s (s < n_steps), and record what the step produced from them: the classical
 * Gram-Schmidt coefficients h_0 .. h_it and ||w|| after the projection (h + s * (restart + 2)),
 * and the normalised new basis vector (v_next + s * n_local).  Two implementations of GMRES
 * separate exponentially along a trajectory; single steps from identical inputs do not
 * (preconditioner.py:732-759 is third-party PETSc code: this pins the restatement step by step
 * against the CPU oracle).  V holds n_steps * (restart + 1) * n_local doubles; all arrays are
 * host memory owned by the caller and must stay valid through the NEXT solve, which consumes
 * the hook: it is cleared when that solve returns (or by passing NULL). */
typedef struct kkt_steplock {
    int n_steps;
    int restart;
    const double *V;
    double *h;
    double *v_next;
} kkt_steplock;
int kkt_debug_set_steplock(kkt_handle h, const kkt_steplock *lock);

/* Byte accounting of the stored operator ("algorithmic bytes"). */
typedef struct kkt_info {
    int64_t n_local;            /* local KKT vector length */
    int64_t n_blocks_stored;    /* (i,j) blocks held by this handle */
    int64_t n_value_arrays;     /* distinct value arrays (== blocks in mode G) */
    int64_t n_patterns;         /* distinct sparsity structures */
    int64_t nnz_blocks;         /* sum of nnz over stored blocks */
    int64_t rows_blocks;        /* sum of rows over stored blocks */
    int64_t bytes_algorithmic;  /* sum_unique[12 nnz + 4(rows+1)] + 16 N */
    int64_t bytes_device_values;/* padded value bytes actually resident */
    int64_t bytes_device_index; /* padded index bytes actually resident */
    int64_t bytes_streamed;     /* bytes one kkt_apply must move with what is stored once read
                                   once: sum over value arrays 8 nnz + sum over sparsity
                                   structures [4 nnz + 4 (rows + 1)] + 16 N */
    double last_solve_ms;       /* wall time of the last kkt_solve* Krylov loop */
    int64_t last_pc_applies;    /* preconditioner applications in the last solve */
    int64_t last_op_applies;    /* operator applications in the last solve */
    int64_t program_fallbacks;  /* times a persistent sweep program timed out waiting for a
                                   neighbour workgroup and the preconditioner was rebuilt as
                                   plain launches (kkt_last_error holds the diagnostic record) */
    /* how the time sweeps of the built-in preconditioner run (0 everywhere: none built yet) */
    int64_t sweep_form;         /* 0 plain launches, 1 counter row program, 2 data-flow row
                                   program, 3 tile program */
    int64_t sweep_tiles, sweep_threads, sweep_depth, sweep_row_slots;   /* tile program plan */
    int64_t sweep_its;          /* Chebyshev degree of the sub-solves (given or derived) */
    int64_t apply_launches;     /* kernel launches of one kkt_apply (block rows only) */
    int64_t apply_switched;     /* ... of which run the width-switched kernel for ragged
                                   structures (P2 / Stokes blocks; option "ragged_switch") */
} kkt_info;
int kkt_get_info(kkt_handle h, kkt_info *info);

/* ------------------------------------------------------------------ multi-GPU */

/* Transport for time-sharded handles.  RCCL: `unique_id` is the 128-byte ncclUniqueId
 * from kkt_comm_unique_id() on rank 0, distributed by the launcher. */
int kkt_comm_unique_id(void *id_out_128);
int kkt_comm_init_rccl(kkt_handle h, const void *unique_id_128);
/* Host-staged transport through caller functions (tests; any launcher without RCCL).
 * allreduce: in-place reduction over ranks of n doubles, op = KKT_OP_SUM or KKT_OP_MAX;
 * sendrecv: send `n_send` doubles to `dst` (or -1: nothing) and receive `n_recv` from
 * `src` (or -1).  Return 0 on success. */
enum { KKT_OP_SUM = 0, KKT_OP_MAX = 1 };
typedef int (*kkt_allreduce_fn)(void *user, double *buf, int n, int op);
typedef int (*kkt_sendrecv_fn)(void *user, const double *send, int64_t n_send, int dst,
                               double *recv, int64_t n_recv, int src);
int kkt_comm_init_callbacks(kkt_handle h, kkt_allreduce_fn ar, kkt_sendrecv_fn sr,
                            void *user);
/* Barrier and max over ranks of a scalar (timing bracket of bench.py). */
int kkt_comm_barrier(kkt_handle h);
int kkt_comm_max(kkt_handle h, double *value_inout);

#ifdef __cplusplus
}
#endif
#endif /* KKT_H */
