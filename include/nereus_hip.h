/*
 * nereus_hip.h — C ABI of libnereus_hip.so: the MI355X (gfx950) SPH fluid step behind the host
 * classes Nereus::SPH / Nereus::IISPH.
 *
 * This is the boundary a host program binds (C, C++, ctypes, cgo, JNI ...): plain pointers and sizes,
 * no C++ or torch types.  It REPLACES the reference's CUDA launcher layer, the `extern "C"` block of
 * sph/sph.cuh:19-230 (defined in sph/sph_cuda.cu), with a context API (nrs_*): device-resident particle
 * state, one call per update().  It is what nereus_amd/host/ (our Nereus::SPH / Nereus::IISPH) and bench.py
 * use; INTEGRATION.md maps every reference entry point to its replacement.
 *
 * The library reads NO environment variables: every switch that changes which kernels run is an NRS_FLAG_* bit of nrs_config.flags.
 * Conventions: every nrs_* call returns 0 on success, a negative NRS_E_* code otherwise, and
 * nrs_last_error() gives the message (no exceptions cross the ABI).  The caller owns host buffers,
 * the library owns device buffers.  A context is bound to one HIP device and one stream; it is not
 * thread-safe.  Particle arrays are AoS xyzw of SReal exactly as the reference's host arrays
 * (sph/sph.cpp:341-368): pos4/vel4 = 4*N SReal, scalars = N SReal; SReal = float or double by
 * nrs_config.precision (the reference's DOUBLE_PRECISION switch, common/common.h:23-43).
 */
#ifndef NEREUS_HIP_H
#define NEREUS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* SphSimParams (common/sph_kernel.cuh:13-59), SReal=float: 132 bytes, align 4. */
typedef struct nrs_params_f32 {
    uint32_t gridSize[3];
    uint32_t numCells;
    float worldOrigin[3];
    float cellSize[3];
    uint32_t numBodies;
    uint32_t maxParticlesPerCell;
    float gasStiffness, viscosity, surfaceTension, restDensity, particleMass, interactionRadius, timestep,
        particleRadius;
    float gravity[3];
    float soundSpeed;
    float beta;
    float kpoly, kpoly_grad, kpress_grad, kvisc_grad, kvisc_denum, ksurf1, ksurf2, bpol;
} nrs_params_f32;

/* SphSimParams, SReal=double: 240 bytes, align 8. */
typedef struct nrs_params_f64 {
    uint32_t gridSize[3];
    uint32_t numCells;
    double worldOrigin[3];
    double cellSize[3];
    uint32_t numBodies;
    uint32_t maxParticlesPerCell;
    double gasStiffness, viscosity, surfaceTension, restDensity, particleMass, interactionRadius, timestep,
        particleRadius;
    double gravity[3];
    double soundSpeed;
    double beta;
    double kpoly, kpoly_grad, kpress_grad, kvisc_grad, kvisc_denum, ksurf1, ksurf2, bpol;
} nrs_params_f64;

typedef struct nrs_ctx nrs_ctx;

enum { NRS_SOLVER_SESPH = 0, NRS_SOLVER_IISPH = 1 };     /* Nereus::SPH (sph/sph.h:23) / Nereus::IISPH (iisph.h:8) */
enum { NRS_KERNELS_MONAGHAN = 0, NRS_KERNELS_MULLER = 1 }; /* KERNEL_SET, common/common.h:14-15 */

/* nrs_config.flags */
enum {
    NRS_FLAG_REFERENCE_ORDER = 1u << 0, /* gather kernels walk the 27 cells and sum in exactly the reference's
                                           order (one thread per slot, per-cell partial sums): slower, used for
                                           bit-level comparison with the oracle */
    NRS_FLAG_NO_FUSION = 1u << 2,       /* keep forces, integrate and hash as three launches (default: a full step
                                           on the production kernels fuses them into the force kernel) */
    NRS_FLAG_NO_SHARED_LISTS = 1u << 3, /* the force kernel re-scans the neighbourhood instead of consuming the hit lists
                                           the density kernel of the same step found (saves HIT_CAP*4 B/particle of HBM) */
    NRS_FLAG_FULL_SORT = 1u << 4,       /* sort all (hash, index) pairs from scratch every step, as the reference does
                                           (sph_cuda.cu:310-313); default: only the particles that changed cell are
                                           sorted and merged into the still-sorted rest (same result, element for element) */
    NRS_FLAG_FAST_ARITH = 1u << 5,      /* tolerance mode (fp32, Muller kernels, SESPH) of the FORCE walk: reciprocals instead of
                                           IEEE divisions, v_rsq instead of correctly rounded square roots, float powers, fused
                                           multiply-adds (the density stays exact).  Hash / index / cell tables are unaffected
                                           (calcGridPos keeps its true division); densities, forces and the integrated state agree
                                           with the default reference-order IEEE arithmetic to ~1e-6 relative per step.  The
                                           reference itself is built with --use_fast_math (CMakeLists.txt:85).  Ignored (exact
                                           arithmetic) for fp64, Monaghan kernels, IISPH and NRS_FLAG_REFERENCE_ORDER. */
    NRS_FLAG_IISPH_SELF_BY_SLOT = 1u << 6, /* IISPH: computePressure / computePressureForce exclude the particle ITSELF from their
                                           neighbour sums.  The reference excludes the slot whose number equals the CUDA thread
                                           id instead (sph_kernel_impl.cuh:1412,1568), which makes its result depend
                                           on the ORDER of the input arrays: the same particles permuted differ by centimetres
                                           after four steps.  Default off = the reference's behaviour; a
                                           multi-GPU slab run can only be compared with a single-domain run with this flag on. */
    NRS_FLAG_NO_WALL_WORKGROUPS = 1u << 7, /* gather launches with ONE kind of workgroup: particles next to a wall are evaluated in place
                                           by the general (boundary-aware) code instead of by the wall workgroups that walk the step's
                                           wall list.  Same results bit for bit; slower on scenes with walls */
    NRS_FLAG_STAGED_SCAN = 1u << 8,     /* SESPH fp32 Muller: the density launch is the LDS-STAGED scan (k_density_staged: one wavefront
                                           per 64 consecutive sorted slots, row hulls by ballot + readlane, one z-plane of candidates
                                           staged in LDS) instead of the quantised global-memory scan.  Same lists, same sums, bit for
                                           bit; measured slower, kept as the north-star's literal kernel shape */
    NRS_FLAG_IISPH_INPLACE_P = 1u << 1, /* reserved: the reference's racy in-place Jacobi is NOT
                                           offered; P_l is always double-buffered */
};

/* error codes */
enum {
    NRS_OK = 0,
    NRS_E_INVALID = -1,  /* bad argument */
    NRS_E_HIP = -2,      /* a HIP runtime call failed */
    NRS_E_CAPACITY = -3, /* more particles than nrs_config.capacity */
    NRS_E_STATE = -4,    /* call not valid in the context's current state */
    NRS_E_NODEVICE = -5, /* no usable HIP device */
    NRS_E_NOTREADY = -6, /* non-blocking query: the result is not there yet */
};

typedef struct nrs_config {
    uint32_t struct_size; /* = sizeof(nrs_config) */
    int32_t device;       /* HIP device ordinal; -1 = the calling thread's current device */
    int32_t solver;       /* NRS_SOLVER_* */
    int32_t precision;    /* 32 or 64: sizeof(SReal)*8 (DOUBLE_PRECISION) */
    int32_t kernel_set;   /* NRS_KERNELS_* (KERNEL_SET) */
    int32_t surface_tension; /* USE_SURFACE_TENSION (CMakeLists.txt:28) */
    uint32_t flags;       /* NRS_FLAG_* */
    uint32_t reserved;
    uint64_t capacity;    /* max fluid particles; run-time replacement of MAX_PARTICLE_NUMBER (sph/sph.h:19) */
    void *stream;         /* hipStream_t to launch on, or NULL to let the context create its own */
} nrs_config;

/* Pipeline stages, for nrs_step_partial / nrs_stage_ms.  Order is the order of SPH::update()
 * (sph/sph.cpp:233-284) and IISPH::update() (sph/iisph/iisph.cpp:172-216). */
enum {
    NRS_STAGE_HASH = 1,      /* calcHash                     sph_cuda.cu:230 */
    NRS_STAGE_SORT = 2,      /* sortParticles                sph_cuda.cu:58 */
    NRS_STAGE_REORDER = 3,   /* reorderDataAndFindCellStart  sph_cuda.cu:295 */
    NRS_STAGE_DENSITY = 4,   /* computeDensityPressure kernel sph_kernel_impl.cuh:365 */
    NRS_STAGE_FORCES = 5,    /* computeForces kernel          sph_kernel_impl.cuh:609 */
    NRS_STAGE_INTEGRATE = 6, /* integrateSystem              sph_cuda.cu:211 */
    NRS_STAGE_I_DENSITY = 10,      /* computeIisphDensity        sph_kernel_impl.cuh:770 */
    NRS_STAGE_I_DISPLACEMENT = 11, /* computeDisplacementFactor  :851 */
    NRS_STAGE_I_ADVECTION = 12,    /* computeAdvectionFactor     :1114 */
    NRS_STAGE_I_SOLVE = 13,        /* computeSumDijPj + computePressure loop, sph_cuda.cu:736-823 */
    NRS_STAGE_I_PFORCE = 14,       /* computePressureForce       :1497 */
    NRS_STAGE_I_INTEGRATE = 15,    /* iisph_integrate            :1625 */
    NRS_STAGE_COUNT = 16
};

/* Arrays readable through nrs_get_array / nrs_device_ptr.  "sorted" = in grid-hash order of the
 * current step, as the reference's m_dSorted* arrays (sph/sph.h:105-111). */
enum {
    NRS_ARR_POS = 0,        /* SVec4[N]  current particle positions (what the next update() starts from) */
    NRS_ARR_VEL = 1,        /* SVec4[N] */
    NRS_ARR_PRESSURE = 2,   /* SReal[N]  IISPH warm-start pressure (m_pressure, iisph.cpp:216) */
    NRS_ARR_HASH = 3,       /* uint32[N] sorted cell hashes (m_dGridParticleHash) */
    NRS_ARR_INDEX = 4,      /* uint32[N] sorted particle indices (m_dGridParticleIndex) */
    NRS_ARR_CELL_START = 5, /* uint32[numCells], 0xffffffff = empty */
    NRS_ARR_CELL_END = 6,   /* uint32[numCells], defined only where CELL_START != 0xffffffff */
    NRS_ARR_SORTED_POS = 7, /* SVec4[N] */
    NRS_ARR_SORTED_VEL = 8, /* SVec4[N] */
    NRS_ARR_DENS = 9,       /* SReal[N] */
    NRS_ARR_PRES = 10,      /* SReal[N] */
    NRS_ARR_FORCES = 11,    /* SVec4[N] (w=0) */
    NRS_ARR_B_HASH = 12,    /* boundary: uint32[Nb] */
    NRS_ARR_B_INDEX = 13,
    NRS_ARR_B_CELL_START = 14,
    NRS_ARR_B_CELL_END = 15,
    NRS_ARR_B_SORTED = 16,  /* SVec4[Nb]: xyz = sorted boundary position, w = its Vbi */
    NRS_ARR_DENS_ADV = 20,  /* IISPH (sph/iisph/iisph.h:27-41) */
    NRS_ARR_DENS_CORR = 21,
    NRS_ARR_P_L = 22,
    NRS_ARR_AII = 23,
    NRS_ARR_VEL_ADV = 24,
    NRS_ARR_FORCES_ADV = 25,
    NRS_ARR_FORCES_P = 26,
    NRS_ARR_DII_FLUID = 27,
    NRS_ARR_DII_BOUNDARY = 28,
    NRS_ARR_SUM_DIJ = 29,
};

const char *nrs_last_error(void);
/* library/ABI version: (major<<16)|minor */
uint32_t nrs_version(void);
/* number of HIP devices visible (0 if none / runtime unusable) */
int nrs_device_count(void);

/* Replaces SPH::SPH()/_initialize() device allocation (sph/sph.cpp:132-188, iisph.cpp:123-159).
 * `params` points to nrs_params_f32 or nrs_params_f64 according to cfg->precision. */
int nrs_create(const nrs_config *cfg, const void *params, nrs_ctx **out);
int nrs_destroy(nrs_ctx *ctx);

/* setParameters (sph/sph.cuh:40, sph_cuda.cu:183-187).  Changing gridSize/numCells re-allocates the
 * cell tables as SPH::_initializeGrid does (sph.cpp:193-202). */
int nrs_set_params(nrs_ctx *ctx, const void *params);
int nrs_get_params(nrs_ctx *ctx, void *params);

/* H2D of particle state: replaces the cudaMemcpy H2D at the top of update() (sph.cpp:233-234,
 * iisph.cpp:172-174).  Writes particles [first, first+count) and sets N = max(N, first+count).
 * vel4/pres may be NULL (zeros).  nrs_set_num_particles truncates/extends N within capacity. */
int nrs_upload_particles(nrs_ctx *ctx, const void *pos4, const void *vel4, const void *pres, uint64_t first,
                         uint64_t count);
int nrs_set_num_particles(nrs_ctx *ctx, uint64_t n);
uint64_t nrs_num_particles(nrs_ctx *ctx);

/* SPH::updateGpuBoundaries (sph/sph.cpp:391-432): upload boundary particles bi4 (SVec4[nb]) and volumes
 * vbi (SReal[nb]); if update_grid != 0 apply SPH::updateGrid (sph.cpp:313-337: origin = AABBmin-0.1,
 * gridSize = nextPow2(ceil((extent+0.1)/h))) and re-allocate the cell tables; then hash, sort and
 * build the boundary cell ranges.  nb = 0 clears the boundaries.  The (possibly new) grid is visible
 * through nrs_get_params. */
int nrs_set_boundaries(nrs_ctx *ctx, const void *bi4, const void *vbi, uint64_t nb, int update_grid);

/* update() x nsteps with state resident on the device (no per-step PCIe traffic).  Asynchronous for the caller: a call with
 * nsteps >= 2 hands the steps to a thread owned by the context and returns at once (before the device has finished, usually before
 * it has started); nsteps <= 1 enqueues the step on the calling thread and returns without waiting for the device.  Every other
 * nrs_* call on the context first waits until the queued steps have been enqueued, and nrs_synchronize / nrs_download /
 * nrs_get_array additionally wait for the device.  A call the context's state does not allow (mid-update after nrs_step_partial, a
 * host-driven IISPH step in progress) is refused at once with NRS_E_STATE; an error that a QUEUED step runs into is returned by the
 * next call on the context, and the steps queued behind it are dropped.
 * (Why a thread: the step loop reads one number back per step — the mover count that sizes the coherent re-sort's library calls, or
 * the IISPH solver's exit test, sph_cuda.cu:736-741 — so whoever enqueues it runs about one step ahead of the device, not nsteps
 * ahead.  NRS_FLAG_FULL_SORT SESPH contexts have no such read-back.) */
int nrs_step(nrs_ctx *ctx, int nsteps);
/* Test hook: run ONE update() but stop after `stop_stage` (NRS_STAGE_*), leaving the intermediate
 * arrays readable through nrs_get_array.  After a partial step the particle state is mid-update:
 * re-upload before stepping again. */
int nrs_step_partial(nrs_ctx *ctx, int stop_stage);
int nrs_synchronize(nrs_ctx *ctx);

/* D2H: replaces the cudaMemcpy D2H at the end of update() (sph.cpp:283-284, iisph.cpp:214-216).
 * Any pointer may be NULL.  Order = grid-hash order of the last step, as in the reference. */
int nrs_download(nrs_ctx *ctx, void *pos4, void *vel4, void *pres);
/* Viewer hand-off that does not stall the simulation (consumer: the render loop's getHostPos(), main.cpp:587-588;
 * the reference copies pos+vel back synchronously at the end of every update(), sph.cpp:283-284).
 * nrs_snapshot_begin copies the current positions (and velocities if with_vel != 0) device-to-device into a staging
 * buffer on the context's stream, then to page-locked host memory owned by the library on a separate copy stream:
 * steps enqueued afterwards overlap with the PCIe transfer.  Two snapshots can be in flight (a third call first waits
 * for the oldest).  nrs_snapshot_wait hands out the OLDEST pending snapshot: with block == 0 it returns
 * NRS_E_NOTREADY (not an error, nrs_last_error is not set) while the transfer is running.  The pointers stay valid
 * until two further nrs_snapshot_begin calls; *step = number of nrs_step steps the state had seen.  vel4 is NULL for
 * a snapshot taken without velocities. */
int nrs_snapshot_begin(nrs_ctx *ctx, int with_vel);
int nrs_snapshot_wait(nrs_ctx *ctx, int block, const void **pos4, const void **vel4, uint64_t *n, uint64_t *step);

/* Copy one of NRS_ARR_* to host memory (dst_bytes must be >= the array's size; returns it in *out_bytes
 * when dst == NULL). */
int nrs_get_array(nrs_ctx *ctx, int which, void *dst, uint64_t dst_bytes, uint64_t *out_bytes);
/* Device address of one of NRS_ARR_* (valid until the next call that re-allocates or swaps buffers,
 * i.e. read it after each step).  For zero-copy consumers (renderer VBO upload, halo packing). */
int nrs_device_ptr(nrs_ctx *ctx, int which, void **dptr, uint64_t *bytes);

/* IISPH: solver iterations of the last step (the `l` of sph_cuda.cu:736). */
int nrs_last_iterations(nrs_ctx *ctx, uint32_t *iters);
/* Cap on IISPH solver iterations per step (0 = none, as the reference). */
int nrs_set_max_iterations(nrs_ctx *ctx, uint32_t max_iters);

/* Per-stage device time, measured with HIP events recorded on the context's stream around the stage's
 * launches.  stage_mask: bit s set = time NRS_STAGE_s (0 = off, 0xffffffff = every stage).  nrs_set_profiling also
 * resets the accumulated times.  nrs_stage_ms returns the time of that stage summed over all steps since the last
 * nrs_set_profiling call and how many launches of the stage that covers; it synchronizes the stream (nrs_step itself
 * does not: event pairs are resolved lazily). */
int nrs_set_profiling(nrs_ctx *ctx, uint32_t stage_mask);
int nrs_stage_ms(nrs_ctx *ctx, int stage, float *ms, uint32_t *launches);

/* ---- multi-GPU slab decomposition (new; the reference is single-GPU — SURVEY §8e) ---------------------------
 * One context per rank/GPU.  A rank owns the particles whose GLOBAL grid cell-x index (floor((x-origin.x)/cell.x),
 * same SphSimParams on every rank) lies in [cell_lo, cell_hi).  Per step, before nrs_step:
 *   nrs_slab_pack   partitions the current particles (stable, deterministic): owned ones are compacted, leavers and
 *                   the halo_cells-wide border layers are written to the two caller-owned DEVICE message buffers;
 *   (caller)        exchanges the buffers with the left/right neighbour — RCCL send/recv over xGMI, e.g.
 *                   torch.distributed.batch_isend_irecv on the tensors that own the buffers;
 *   nrs_slab_unpack appends the received migrants (they become owned) and halo copies (read-only, pos.w = 2).
 * nrs_step then evaluates density on owned particles plus one cell beyond each cut and forces on owned particles
 * only.  Message buffer: nrs_slab_message_bytes(capacity, precision) bytes =
 *   [u32 nMigrants, u32 nHalo, u32 0, u32 0 | vec4 pos[capacity] | vec4 vel[capacity]].
 * Pass NULL for the neighbour that does not exist (ends of the chain).  IISPH contexts: halo_cells >= 8 and the step is driven
 * through nrs_iisph_predict / _iterate / _finish (below). */
int nrs_slab_configure(nrs_ctx *ctx, int32_t cell_lo, int32_t cell_hi, int32_t halo_cells);
/* counts (optional) receives {stay, migrate-left, halo-left, migrate-right, halo-right, ghost}.
 * With counts == NULL nrs_slab_pack does not wait for the device: the two messages are complete in stream order when it returns (the
 * caller enqueues its sends on the same stream right behind it), and the stream populations are read back together with the headers of
 * the received messages inside nrs_slab_unpack — ONE host synchronisation per exchange.  A message-capacity overflow is then reported
 * by nrs_slab_unpack (or by whichever call on the context comes first) with NRS_E_CAPACITY.  nrs_slab_last_counts returns the counts
 * of the last pack afterwards.  Passing counts makes nrs_slab_pack wait for them itself. */
int nrs_slab_pack(nrs_ctx *ctx, void *send_left, void *send_right, uint64_t capacity, uint32_t counts[6]);
int nrs_slab_unpack(nrs_ctx *ctx, const void *recv_left, const void *recv_right, uint64_t capacity);
int nrs_slab_last_counts(nrs_ctx *ctx, uint32_t counts[6]);
/* owned particles (the first nrs_num_owned() entries of NRS_ARR_POS/VEL right after nrs_slab_pack/unpack) */
uint64_t nrs_num_owned(nrs_ctx *ctx);
uint64_t nrs_slab_message_bytes(uint64_t capacity, int precision);
/* Owned particles (pos.w == 1) per global cell-x column first_cell .. first_cell+ncells-1, to host memory: the input
 * of a count-balanced re-cut.  New cuts are applied by calling nrs_slab_configure again; particles that now belong
 * to a neighbour leave with the next nrs_slab_pack (a cut may therefore move by less than a slab width at a time). */
int nrs_slab_histogram(nrs_ctx *ctx, int32_t first_cell, uint32_t ncells, uint32_t *counts);

/* IISPH step in three calls, for runs in which the solver loop's exit test needs a value the library cannot form alone — a slab
 * run: pressureSolve (sph_cuda.cu:736-741) loops `while ((rho_avg - 1000) > 1 || l < 2)` with rho_avg the average predicted
 * density over ALL particles, i.e. over all ranks.
 *   nrs_iisph_predict   hash / sort / reorder + predictAdvection (density, displacement factors, advection factors)
 *   nrs_iisph_iterate   ONE relaxed-Jacobi iteration; returns the sum of the corrected densities over the particles this rank
 *                       owns and their number (all particles of a single-domain context): the caller adds these up over the
 *                       ranks (one scalar all-reduce), forms rho_avg = (SReal)sum / count and decides
 *   nrs_iisph_finish    pressure force + integration
 * nrs_step on an IISPH context is the same three phases with the loop inside.  Slab runs (nrs_slab_configure on an IISPH context)
 * need a halo of at least 2 * iterations + 4 cells — every iteration consumes two cells of halo validity, the predict stages three,
 * the pressure force one — i.e. >= 8; nrs_iisph_iterate fails with NRS_E_STATE when the loop runs longer than the halo supports.
 * The warm-start pressure of every particle travels in vel.w of the slab messages (iisph_integrate zeroes vel.w anyway). */
int nrs_iisph_predict(nrs_ctx *ctx);
int nrs_iisph_iterate(nrs_ctx *ctx, double *sum_density_corr, uint64_t *count);
int nrs_iisph_finish(nrs_ctx *ctx);

/* Coherent re-sort statistics since nrs_create: steps whose (hash, index) pairs were produced by sorting only the
 * particles that changed cell and merging them into the rest, and how many of those fell back to the full radix sort
 * because more than half of the particles had moved (the measured break-even).  (Steps after an upload
 * or a grid change and partial steps always use the full sort and are not counted.) */
int nrs_resort_stats(nrs_ctx *ctx, uint64_t *steps, uint64_t *fallbacks);

/* Diagnostics of the last completed step, as doubles (they synchronize the stream):
 *   NRS_STAT_MOVERS         particles that changed grid cell (the mover count of the last coherent re-sort; -1 if the
 *                           last step sorted from scratch without counting)
 *   NRS_STAT_HIT_OVERFLOW   particles whose neighbour hit list overflowed (they take the reference-order cell walk)
 *   NRS_STAT_HIT_MEAN/_MAX  neighbours (fluid + boundary hits) per particle kept in the hit lists
 *   NRS_STAT_UNSTAGED       particles whose wavefront could not stage its neighbour rows in LDS (grid-edge cells, or
 *                           hulls longer than the pool) and scanned them from global memory instead
 * The HIT_* / UNSTAGED values need the shared hit lists of the production kernels (NRS_E_STATE otherwise). */
enum { NRS_STAT_MOVERS = 0, NRS_STAT_HIT_OVERFLOW = 1, NRS_STAT_HIT_MEAN = 2, NRS_STAT_HIT_MAX = 3, NRS_STAT_UNSTAGED = 4 };
int nrs_get_stat(nrs_ctx *ctx, int which, double *out);

/* Akinci boundary volumes on the device (no context needed): vbi[i] = 1 / sum_k W_poly6(|x_i - x_k|, h) over the boundary
 * particles k within h of i, i included — what the reference takes from its un-vendored submodule
 * (sample_spheres::boundary_forces::getVbi, main.cpp:546; own implementation, parity unpinned).  bi4: nb xyzw particles of
 * SReal (precision 32/64), vbi: nb SReal, both HOST buffers; device < 0 = current device.  Uses the same hash / radix sort /
 * cell-range / 27-cell gather as the solver, on a private grid. */
int nrs_boundary_volumes(int device, int precision, const void *bi4, uint64_t nb, double h, void *vbi);

/* Test hook: the DEVICE smoothing kernels / vector helpers (common/kernels_impl.cuh:85-203, helper_math.h semantics) on n
 * caller-supplied separations r3 (and second vectors s3, may be NULL), HOST buffers of 3n SReal; which = 0 Wdefault, 1
 * Wdefault_grad, 2 Wpressure_grad, 3 Wviscosity_grad, 4 Wmonaghan, 5 Wmonaghan_grad, 8 dot, 9 length, 10 vec*float, 11 float*vec,
 * 12 vec/float, 13 identity, 14 +, 15 - (the numbering of oracle/ref_kernels_driver.cpp; scalars land in out[3i]).  Lets the
 * tests compare the product's arithmetic bit for bit with the reference's own header compiled unmodified. */
int nrs_eval_smoothing(int precision, int which, uint64_t n, const void *r3, const void *s3, double h, double c0, double c1, void *out);

/* maxDensity / maxVelocity (sph/sph.cuh, sph_cuda.cu:32-53): diagnostics over the sorted arrays. */
int nrs_max_density(nrs_ctx *ctx, double *out);
int nrs_max_velocity(nrs_ctx *ctx, double *out_speed);

#ifdef __cplusplus
}
#endif
#endif /* NEREUS_HIP_H */
