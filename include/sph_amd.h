/*
 * sph_amd -- C ABI of the MI355X SPH particle-pair path.
 *
 * The reference (jammy4536/SPH) has no plugin or FFI seam: its hot path is three free
 * functions over C++ types plus file-scope globals, called from main() only.  Each entry
 * point below names the reference function or lines it stands in for.  All entry points
 * return SPH_OK (0) or an SPH_E* code; none throws or exits.  A context is driven by one
 * host thread at a time.
 *
 * Precision contract (reference struct Particle, WCXSPH3D.cpp:81-92): positions, velocities,
 * XSPH velocities and accelerations are double; rho, p, Rrho and m are float, and every
 * place where the reference rounds through float is rounded through float here.
 */
#ifndef SPH_AMD_H
#define SPH_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    SPH_OK = 0,
    SPH_EINVAL = 1,   /* bad argument or parameter */
    SPH_EDEVICE = 2,  /* a HIP call failed; see sph_last_error */
    SPH_ENOMEM = 3,
    SPH_ESTATE = 4,   /* call made in the wrong order (e.g. step before upload) */
    SPH_ECOMM = 5     /* RCCL failure in a multi-GPU context */
};

/* Which reference program's pair physics. */
enum {
    SPH_VARIANT_WCSPH = 0,  /* WCSPH.cpp: one global mass, no XSPH, x advanced with v */
    SPH_VARIANT_WCXSPH = 1  /* WCXSPH.cpp / WCXSPH3D.cpp: per-particle mass, XSPH */
};

/* `particles` (state n) or `particlesh` (state n+1) of the reference's main(). */
enum { SPH_STATE_N = 0, SPH_STATE_NP1 = 1 };

/* Constants of one run: the reference's compile-time block (WCXSPH3D.cpp:39-79,97,142-143;
 * WCXSPH.cpp:36-74,92; WCSPH.cpp:42-78,96) passed by value.  The caller computes them in
 * the reference's operation order (sph_case_params does). */
typedef struct sph_params {
    int32_t  dim;              /* 2 or 3 */
    int32_t  variant;          /* SPH_VARIANT_* */
    uint64_t npts;             /* all particles, wall particles first */
    uint64_t bound_parts;      /* original indices below this are wall particles */
    double   H, HSQ;           /* support radius and its square */
    double   correc;           /* kernel normalisation: 7/(4 pi H^2) or 21/(16 pi H^3) */
    double   search_radius_sq; /* the reference's `search_radius` (a squared length) */
    double   B, gam, rho0, Cs; /* Tait equation of state */
    double   alpha, eps;       /* artificial viscosity, XSPH factor */
    double   g;                /* added to f[g_axis] (reference: -9.81) */
    int32_t  g_axis;
    int32_t  subits;           /* Newmark-beta sweeps per step (reference: 8) */
    double   beta;             /* Newmark beta as the reference holds it: (double)0.25f */
    double   mass;             /* the single mass of SPH_VARIANT_WCSPH */
    double   dt0;              /* initial time step (reference: 0.0002) */
    /* Implementation knobs; 0 selects the default. */
    double   skin_frac;        /* neighbour-candidate skin as a fraction of the search radius;
                                  0: chosen from the particles' travel per step (starts at 0.05) */
    int32_t  max_neighbours;   /* initial per-particle candidate capacity (grows on demand) */
    int32_t  device;           /* HIP device ordinal; -1 = current device */
    /* Slab decomposition along x, one context per GPU (no counterpart in the reference, which
     * is one process on one core).  nranks <= 1: a single domain, the fields below are unused.
     * Otherwise npts is the number of particles this rank owns at first, npts_global the total
     * (the convergence diagnostic divides by it, WCXSPH3D.cpp:288), and [slab_lo, slab_hi) the
     * x-interval this rank owns.  capacity > npts is the room per array: of the excess, half
     * lets the owned set grow by migration, a quarter each holds the left and right halo (and
     * bounds how many particles may cross one plane between two list rebuilds).  migrate != 0:
     * at the list rebuild that opens a step, particles outside [slab_lo, slab_hi) are handed to
     * the neighbour on that side; the local order of upload/download is then renumbered (see
     * sph_local_count). */
    int32_t  rank, nranks;
    uint64_t npts_global;
    uint64_t capacity;
    double   slab_lo, slab_hi;
    int32_t  migrate;
    int32_t  reserved_;
    /* Implementation knob.  Step lists: once per step, on the positions the step's second sweep
     * starts from, the candidates within search radius * (1 + margin) are copied out of every
     * candidate row, and the sweeps walk only those; a step whose particles outrun the margin
     * is redone on the candidate lists.  0: used when the candidate skin is wide enough for them to pay, margin chosen from
     * the travel seen between a step's sweeps; > 0: always, with this margin; < 0: never. */
    double   inner_skin_frac;
} sph_params;

/* Caller-owned host arrays in ORIGINAL particle order, one array per component
 * (x[0] holds all x coordinates).  NULL entries are skipped.  Layout of the reference's
 * struct Particle, turned into structure-of-arrays. */
typedef struct sph_fields {
    double* x[3];
    double* v[3];
    double* V[3];
    double* f[3];
    float*  rho;
    float*  p;
    float*  Rrho;
    float*  m;
} sph_fields;

/* The reference's file-scope run state (WCXSPH3D.cpp:70-71,76-79). */
typedef struct sph_scalars {
    double dt;
    float  t;
    double maxmu, maxf, errsum, logbase;
    /* diagnostics of this implementation */
    uint64_t list_rebuilds;   /* neighbour-candidate rebuilds so far */
    uint64_t step_redos;      /* steps repeated because a particle outran the skin */
    uint64_t hit_pairs;       /* i != j pairs inside the search radius in the last sweep */
    uint64_t host_syncs;      /* times the host has waited for the device on the stepping path so
                                 far, list rebuilds not counted (one per step on one GPU; one per
                                 step over a device-buffer transport) */
} sph_scalars;

typedef struct sph_ctx sph_ctx;

/* Replaces the construction of both State vectors and the kd-tree in main()
 * (WCXSPH3D.cpp:444-454).  Allocates device memory for params->npts particles. */
int sph_create(const sph_params* params, sph_ctx** out);
void sph_destroy(sph_ctx* ctx);
const char* sph_last_error(const sph_ctx* ctx);

/* Host -> device copy of one State (what InitSPH and the particlesh copy loop fill:
 * WCXSPH3D.cpp:448-451).  Components beyond dim are ignored. */
int sph_upload(sph_ctx* ctx, int32_t which_state, const sph_fields* src);
/* Device -> host copy of one State in original order (what write_frame_data reads:
 * WCXSPH3D.cpp:408-437). */
int sph_download(sph_ctx* ctx, int32_t which_state, const sph_fields* dst);

/* Forces(State&, tree) on one state (WCXSPH3D.cpp:139-205; WCXSPH.cpp:134-201;
 * WCSPH.cpp:137-200), including the fixed-radius search it performs per particle. */
int sph_forces(sph_ctx* ctx, int32_t which_state);
/* The loop after the t=0 Forces call that zeroes the wall particles' f
 * (WCXSPH3D.cpp:457-458). */
int sph_zero_boundary_forces(sph_ctx* ctx, int32_t which_state);

/* nsteps x Newmark_Beta(particles, particlesh, tree) (WCXSPH3D.cpp:251-303;
 * WCXSPH.cpp:245-298; WCSPH.cpp:245-296).  `out` may be NULL. */
int sph_step(sph_ctx* ctx, int32_t nsteps, sph_scalars* out);
/* sph_step for callers that cannot allocate device memory (tight serving loops, harnesses that
 * forbid hipMalloc).  No counterpart in the reference.  Same kernels and same results as
 * sph_step, but the pair kernel's per-workgroup partial results and their fold live in a fixed
 * array inside the library's code object (about 2.6 MB per GPU, held until the process ends)
 * instead of the context's buffers, and the call makes no hipMalloc/hipFree.  It is not a call
 * for code under HIP graph capture: like sph_step it reads the step's scalars back and waits
 * for the stream once per step (and a few times more when the lists are rebuilt).  If a
 * buffer would have to grow (the candidate lists overflow their rows) it returns SPH_ENOMEM
 * with state n untouched and the step in hand abandoned: the same step is then made through
 * sph_step, which starts it again from its beginning.  Single-domain contexts of at most
 * SPH_NOALLOC_MAX_PARTICLES particles; larger ones and slab contexts get SPH_EINVAL.  All calls
 * on one GPU share that GPU's array: calls that overlap on one GPU (two host threads, two
 * contexts) race. */
#define SPH_NOALLOC_MAX_PARTICLES 4194304u
int sph_step_noalloc(sph_ctx* ctx, int32_t nsteps, sph_scalars* out);
/* Diagnostic: the first `nsweeps` sweeps (Forces + wall reset + update) of ONE
 * Newmark_Beta call, without the time-step update and without the pn = pnp1 copy, and with
 * f, V and Rrho of state n+1 stored after the last of them. */
int sph_sweeps(sph_ctx* ctx, int32_t nsweeps, sph_scalars* out);

int sph_get_scalars(sph_ctx* ctx, sph_scalars* out);
int sph_set_dt(sph_ctx* ctx, double dt);

/* DensityReinit(particlesh, tree) (WCXSPH3D.cpp:208-249; WCXSPH.cpp:204-243; WCSPH.cpp:203-243):
 * rho of state n+1 in the reference's order, a particle seeing the new rho of its neighbours
 * with a smaller index.  On a slab-decomposed context the call is collective (every rank makes
 * it) and the order is that of the global indices.  A single-domain context of more than 12 288
 * particles is walked in one launch that needs the whole GPU (all its workgroups resident); if
 * the GPU does not grant that the call restores rho and walks one launch per dependency level
 * instead -- same values (environment: SPH_REINIT_WALK=levels, SPH_REINIT_LIMIT_S). */
int sph_density_reinit(sph_ctx* ctx);

/* The fixed-radius neighbour sets that Forces() would see for one state, as CSR over
 * original indices, columns ascending, the particle itself included (the result of
 * radiusSearch at WCXSPH3D.cpp:157 for every particle).  row_offsets has npts+1 entries.
 * With cols == NULL only row_offsets is filled. */
int sph_neighbours(sph_ctx* ctx, int32_t which_state, uint64_t* row_offsets, uint32_t* cols);

/* ---- timing of the pair kernel, for benchmarks ------------------------------------- */
/* When enabled, every launch of the fused pair/update kernel is bracketed by HIP events
 * on the context's stream. sph_profile_read returns the summed duration and launch count
 * since the last read, waiting for the stream first. */
int sph_profile_enable(sph_ctx* ctx, int32_t on);
int sph_profile_read(sph_ctx* ctx, double* pair_ms_total, uint64_t* pair_launches,
                     double* rebuild_ms_total, uint64_t* rebuilds);
/* More detail (call it BEFORE sph_profile_read, which clears the records): [0] pair launches that
 * walked the candidate lists, [1] pair launches that walked the step lists, [2] draws of the step
 * lists from the candidates.  Both arrays have three entries. */
int sph_profile_read_kinds(sph_ctx* ctx, double* ms3, uint64_t* launches3);
/* Slab runs, what the exchanges cost (call it BEFORE sph_profile_read, which clears the records;
 * no counterpart in the reference, which is one process): [0] the halo exchange of a sweep, HIP
 * events on the communication stream round the transport's send/receive, which runs beside the
 * interior pair launch; [1] the step's agreement on its scalars (the all-reduce that takes the
 * place of the globals maxmu, maxf, errsum: WCXSPH3D.cpp:198-200, 280-298), events on the compute
 * stream; [2] the halo set-up of a list rebuild and [3] the migration of particles that left the
 * slab, both host time (they wait for the device).  Sums in ms and counts. */
int sph_profile_read_comm(sph_ctx* ctx, double* ms4, uint64_t* count4);

/* How the last sph_density_reinit went through the particles (no counterpart in the reference,
 * whose DensityReinit is one loop in index order, WCXSPH3D.cpp:205-252; every way below gives that
 * loop's values): [0] one of SPH_REINIT_*, [1] launches of the walk's kernels, [2] slab runs: the
 * rounds of exchanges with the neighbouring ranks, [3] bit 0: the ranks walked one after the other
 * because they share a GPU; bit 1: the call needed no list rebuild (the stepping path's lists stood, and
 * the walk's lists were made on their order).  The array has four entries. */
#define SPH_REINIT_NONE 0
#define SPH_REINIT_ONE_WORKGROUP 1      /* small cases: one workgroup, a barrier per dependency level */
#define SPH_REINIT_LEVELS 2             /* one launch per dependency level */
#define SPH_REINIT_ONE_LAUNCH 3         /* one launch, a queue per workgroup */
#define SPH_REINIT_ONE_LAUNCH_ROUNDS 4  /* slab runs: that launch once per round of exchanges */
int sph_reinit_info(sph_ctx* ctx, uint64_t* out4);

/* ---- slab decomposition ----------------------------------------------------------------
 * No counterpart in the reference.  A rank owns the particles whose x lay in its interval at
 * upload.  Before every Forces call each rank needs, from each slab neighbour, the records of
 * the particles within the candidate radius of the shared plane (the halo); per step the
 * ranks agree on max(mu), max|f|, sum|dx|^2 and on whether the step must be redone.  The
 * library does the selecting, packing and scheduling; moving bytes between ranks is the
 * transport's job. */
typedef struct sph_transport {
    void* user;
    /* 1: buffers handed to the callbacks are host memory (the library stages them); 0: device
     * memory, and `stream` (a hipStream_t) orders the transfer. */
    int32_t host_buffers;
    /* sizeof(sph_transport) as the caller was compiled: zero the struct, set this, then the rest.
     * sph_set_transport refuses any other value, so a caller built against a header with fewer
     * callbacks (or one that left the struct uninitialised) is told so instead of having a
     * callback it never set called. */
    uint32_t struct_size;
    /* Send `nsendL` bytes to the left neighbour (rank-1) and `nsendR` to the right (rank+1);
     * receive exactly `nrecvL` / `nrecvR` bytes from them.  A side without a neighbour has
     * byte counts 0.  Returns 0 on success. */
    int (*sendrecv)(void* user, const void* sendL, uint64_t nsendL, void* recvL, uint64_t nrecvL,
                    const void* sendR, uint64_t nsendR, void* recvR, uint64_t nrecvR, void* stream);
    /* In-place all-reduce of n doubles over all ranks; op 0 = max, 1 = sum. */
    int (*allreduce)(void* user, double* vals, int32_t n, int32_t op, void* stream);
    /* Optional (may be NULL: two allreduce calls are made instead).  The step's agreement in one
     * exchange: maxvals[0..nmax) folded with max and sumvals[0..nsum) with a sum, in place. */
    int (*allreduce_max_sum)(void* user, double* maxvals, int32_t nmax, double* sumvals, int32_t nsum, void* stream);
} sph_transport;

/* Installs the transport (copied).  Required before the first Forces/step when nranks > 1.
 * SPH_EINVAL when struct_size is not sizeof(sph_transport) or a required callback is NULL. */
int sph_set_transport(sph_ctx* ctx, const sph_transport* t);
/* Global original indices of this rank's particles, in upload order (wall particles are the
 * ones below bound_parts of the GLOBAL numbering).  Required when nranks > 1. */
int sph_upload_global_ids(sph_ctx* ctx, const uint32_t* gids);
/* Particles this rank owns now, and how many it has received / handed over so far.  After a
 * migration the arrays of sph_upload/sph_download have this many entries, in an order that
 * sph_download_global_ids reports (entry k belongs to the particle with that global index). */
int sph_local_count(sph_ctx* ctx, uint64_t* npts, uint64_t* migrated_in, uint64_t* migrated_out);
int sph_download_global_ids(sph_ctx* ctx, uint32_t* gids);
/* Size of the halo since the last list rebuild: particles this rank sends to its left [0] and
 * right [1] neighbour at every sweep (64 bytes each), and receives from them. */
int sph_halo_counts(sph_ctx* ctx, uint64_t* sent2, uint64_t* received2);
/* Moves this rank's x-interval (all ranks call it with planes that still tile the line).  With
 * migrate on, the particles now outside the interval are handed over at the next step's list
 * rebuild; a particle travels one slab per rebuild, so move a plane by less than a slab, and by
 * no more particles than the per-side room allows (see capacity). */
int sph_set_slab_interval(sph_ctx* ctx, double slab_lo, double slab_hi);
/* A transport over RCCL (ncclSend/ncclRecv/ncclAllReduce on the context's device).
 * `unique_id` is the 128-byte ncclUniqueId, identical on all ranks (rank 0 obtains it with
 * sph_rccl_unique_id and the launcher broadcasts it). */
int sph_rccl_unique_id(void* unique_id_128);
/* Diagnostic: runs the RCCL transport's own calls on one GPU -- a one-rank communicator, the
 * grouped send/receive with itself as both neighbours, the grouped max+sum all-reduce -- and
 * checks what comes back.  `msg` receives "ok" or what failed. */
int sph_rccl_selftest(int32_t device, char* msg, int32_t msg_len);
int sph_use_rccl(sph_ctx* ctx, const void* unique_id_128);

/* ---- the reference's case set-up, host side ----------------------------------------- */
/* The "Simulation Definition" block (WCXSPH3D.cpp:39-48; WCXSPH.cpp:36-45; WCSPH.cpp:42-51). */
enum { SPH_PROGRAM_WCSPH = 0, SPH_PROGRAM_WCXSPH = 1, SPH_PROGRAM_WCXSPH3D = 2 };
enum { SPH_SIMTYPE_BOX = 0, SPH_SIMTYPE_DROPLET = 1, SPH_SIMTYPE_JET = 2 };
typedef struct sph_case {
    int32_t program;     /* SPH_PROGRAM_* */
    int32_t npart[3];    /* xyPART */
    double  pstep;       /* Pstep */
    double  box[3];      /* Box */
    double  start[3];    /* Start */
    double  dt0;         /* initial dt */
    /* Settings-file overrides of constants the reference fixes at compile time; 0 keeps the
     * reference's value (rho0 1000, Cs 50, alpha 0.025, gravity 9.81). */
    double  rho0, Cs, alpha, gravity;
    /* Not in the reference (InitSPH only makes the box and one block): a second body of fluid
     * on the same lattice spacing and with the block's particle mass, listed after the block.
     * simtype 0: none; 1 (droplet): the lattice points within drop_radius of drop_centre;
     * 2 (jet): those within drop_radius of the axis through drop_centre along the last
     * coordinate and within drop_height/2 of it along that axis.  They start with velocity
     * drop_v. */
    int32_t simtype;
    int32_t reserved_;
    double  drop_radius;
    double  drop_centre[3];
    double  drop_height;
    double  drop_v[3];
} sph_case;

/* The constants as shipped in the given reference program. */
int sph_case_default(int32_t program, sph_case* out);
/* Derived constants in the reference's operation order, plus particle counts. */
int sph_case_params(const sph_case* c, sph_params* out, double* simmass, double* boundmass);
/* InitSPH (WCXSPH3D.cpp:306-379; WCXSPH.cpp:301-350; WCSPH.cpp:297-350): fills x, v, V, f,
 * rho, p, Rrho, m of `dst` (arrays of params.npts entries; NULL entries skipped). */
int sph_case_init(const sph_case* c, const sph_fields* dst);

/* The particles of a case whose x lies in [xlo, xhi), in the order of InitSPH, with their
 * global indices.  Call with dst == NULL to get the count only. */
int sph_case_init_range(const sph_case* c, double xlo, double xhi, const sph_fields* dst, uint32_t* gids,
                        uint64_t* count);
/* Histogram of particle x-coordinates over [0, box_x] in nbins equal bins, fluid and wall
 * particles counted separately: for choosing slab planes that balance the load. */
int sph_case_histogram_x(const sph_case* c, int32_t nbins, uint64_t* fluid_counts, uint64_t* wall_counts);

#ifdef __cplusplus
}
#endif
#endif
