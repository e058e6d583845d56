/* Context and C ABI of the MI355X SPH pair path (see include/sph_amd.h). */
#include "../../include/sph_amd.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unistd.h>
#include <vector>

#include "sph_kernels.h"

namespace {

constexpr int NREC = 3;

struct EventPair {
    hipEvent_t a, b;
    int kind; /* pair launches: 0 walked the candidate lists, 1 the step lists */
};

}  // namespace

struct sph_ctx {
    sph_params prm;
    DevParams dp;
    int dim = 0;
    bool xsph = false;
    uint32_t n = 0;        /* particles this context owns */
    uint32_t n_halo = 0;   /* neighbours' particles kept behind the owned ones (slab runs) */
    uint32_t cap = 0;      /* room in every per-particle array */
    uint32_t npairblocks = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    /* state n, cell order: x,v,V,f per component; rho,Rrho,p,m */
    double* nd[12] = {};
    double* nd_alt[12] = {};
    float* nfl[4] = {};
    float* nfl_alt[4] = {};
    /* state n+1: pair-loop records (three buffers: the step's start state is kept while the
     * sweeps alternate between the other two) and the outputs of Forces */
    Rec* rec[NREC] = {};
    Rec* rec_alt = nullptr;
    int b0 = 0; /* record buffer that holds state n+1 */
    double* pd[6] = {}; /* f[3], V[3] */
    double* pd_alt[6] = {};
    float* pRrho = nullptr;
    float* pRrho_alt = nullptr;
    float* pp = nullptr; /* pressure of state n+1 (the records carry p/rho^2 instead) */
    float* pp_alt = nullptr;
    uint32_t* id = nullptr; /* position in the caller's arrays; SPH_HALO_ID for halo particles */
    uint32_t* id_alt = nullptr;
    uint32_t* gid = nullptr; /* global original index (slab runs); the wall test reads it */
    uint32_t* gid_alt = nullptr;

    /* sort, cells, candidate lists */
    uint32_t *keys = nullptr, *keys_alt = nullptr, *perm = nullptr, *perm_alt = nullptr;
    uint32_t* hist = nullptr;
    size_t hist_cap = 0;
    uint32_t* bsum = nullptr;
    size_t bsum_cap = 0;
    uint32_t* cbeg = nullptr; /* cell table: first sorted position with key >= c, ncells + 1 entries */
    uint32_t* hbeg = nullptr; /* the same for the halo particles */
    size_t cell_cap = 0;
    uint32_t* nbr = nullptr;
    uint32_t cap_k = 0;
    uint32_t* cnt = nullptr;
    /* the step's working lists (a subset of the candidate lists, drawn at the step's start) */
    uint32_t* nbr_in = nullptr;
    uint32_t nbr_in_cap_k = 0;
    uint32_t* cnt_in = nullptr;
    /* Step lists: the step's second sweep draws them from the candidates with radius
     * R*(1+inner_frac), the later sweeps walk them, and so does the next step's first sweep when
     * they still cover it (step_lists_valid).  inner_mode 0: used when the candidate skin is wide
     * enough to pay, inner_frac following the travel seen between a step's later sweeps;
     * 1: always, with the caller's inner_frac; -1: never. */
    int inner_mode = 0;
    double inner_frac = 0.02, r_in2 = 0, limit_in = 0, outer_budget = 0;
    bool two_level = false;        /* decided at each rebuild */
    bool step_lists_valid = false; /* nbr_in serves the positions state n+1 holds now */
    uint32_t build_trips = BUILD_TRIPS; /* records per particle in the list build (SPH_BUILD_TRIPS; 0: the ring kernel alone) */
    uint64_t build_handed_on = 0;       /* particles whose rows the ring kernel wrote because their records did not fit */
    bool np1_out_in_n = false;     /* f, V, Rrho of state n+1 exist only as state n's (see settle_np1_outputs) */
    double in_travel = 0;          /* running estimate of the travel between a step's second and last sweep */
    double two_level_min_skin = 0.08, skin_k1 = 2.2, skin_k2 = 12.0, skin_max = 0.5;
    bool skin_k2_auto = true; /* skin_k2 follows the memory left for the list rows (end of sph_create) */
    const uint32_t* use_nbr = nullptr; /* what the pair kernel walks in the sweep under way */
    const uint32_t* use_cnt = nullptr;
    uint32_t* need = nullptr;
    GridParams grid;
    QLattice lattice; /* of the 8-byte position entries the list build and the draw test, set at every rebuild */
    bool check_lists = false; /* SPH_CHECK_LISTS: verify the rows after every list build */
    bool list_valid = false;
    bool lists_for_step = false; /* built by the stepping path (wall-wall pairs may be left out) */
    double skin_frac = 0.05, r_list2 = 0, disp_limit = 0;
    uint32_t brick_log2 = 2;
    uint32_t run_fine = 4; /* fine cells per cell along the run axis (SPH_RUN_FINE) */
    double disp_host = 0;
    /* The skin follows the motion unless the caller fixed it: step_travel is a running upper
     * estimate of how far the fastest particle gets in one step (the same number on every rank
     * of a slab run: it derives from the agreed travel bound). */
    bool skin_auto = true;
    double step_travel = 0;

    DevScalars* S = nullptr;
    DevScalars* S_host = nullptr; /* pinned */
    double* part = nullptr;  /* per-workgroup partials of the pair kernel */
    double* part2 = nullptr; /* their first fold */
    /* where the launches of the call under way leave their partials: the two buffers above, or
     * this device's copy of the file-scope scratch array (sph_step_noalloc) */
    double* part_use = nullptr;
    double* part2_use = nullptr;
    double* scratch = nullptr; /* device address of this device's scratch array */
    bool no_alloc = false;     /* the call under way may not allocate or free device memory */
    bool trace = false;        /* SPH_TRACE=1: a line per list rebuild on stderr */
    uint64_t draws = 0;
    double* bbox_part = nullptr;
    double* bbox_host = nullptr; /* pinned */
    uint32_t* need_host = nullptr; /* pinned, 8 words */
    void* tmp = nullptr; /* cap * 8 bytes staging, original order */

    /* DensityReinit: dependency counters, frontier queue, control block (allocated on first use) */
    uint32_t *ri_pend = nullptr, *ri_queue = nullptr;
    ReinitCtl* ri_ctl = nullptr;
    uint32_t* ri_flow = nullptr; /* control words of the one-launch walk */
    uint32_t ri_owners = 0;      /* its workgroups (all resident at once); 0 = walk by levels */
    const uint32_t* ri_nbr = nullptr; /* the lists the walk reads: the candidate lists of a rebuild of its own, or rows */
    const uint32_t* ri_cnt = nullptr; /* built for it on the standing order (reinit_lists_on_standing_order) */
    bool ri_fresh_lists = false;      /* SPH_REINIT_LISTS=fresh: always rebuild */
    bool probe_standing = false;      /* SPH_NEIGHBOURS_LISTS=standing: sph_neighbours reads lists made the same way */
    uint64_t ri_quick_lists = 0;      /* calls served without a rebuild */
    size_t ri_gran_off = 0;      /* where the one-launch walk's published words began at the last call */
    bool ri_take_turns = false;  /* slab run, ranks sharing a GPU: the one-launch walks run one rank at a time */
    bool lost = false;           /* a launch of this context never came back: every later call fails at once, and
                                    sph_destroy frees nothing on the device (it would wait for that launch) */
    uint32_t ri_gen = 0;         /* its calls so far: the tag of the densities a call publishes */
    double ri_limit_s = 2.0;     /* a wave of it gives up after waiting this long for one thing (a whole walk takes < 1 s) */
    uint64_t reinit_launches = 0; /* of the last density re-initialisation: launches of its walk, */
    uint64_t reinit_rounds = 0;   /* exchange rounds (slab runs) */
    uint32_t reinit_walked = 0;   /* and how it walked (SPH_REINIT_*) */

    uint64_t rebuilds = 0, redos = 0;
    uint64_t host_syncs = 0; /* host waits for the device on the stepping path outside list rebuilds */
    bool uploaded_n = false, uploaded_np1 = false;

    /* slab decomposition */
    bool dist = false;
    int rank = 0, nranks = 1;
    bool has_side[2] = {false, false}; /* a neighbour on the left / right */
    uint32_t halo_cap = 0;             /* per side: halo particles, and particles migrating per rebuild */
    uint32_t own_cap = 0;              /* most particles this rank may come to own */
    bool migrate = false;
    uint32_t* drop = nullptr;          /* 1: handed to a neighbour in the rebuild under way */
    Mig *migsend = nullptr, *migrecv = nullptr; /* [2][halo_cap] */
    size_t stage_half = 0;             /* bytes per side in the pinned staging buffers */
    uint64_t migrated_in = 0, migrated_out = 0;
    uint32_t nsend[2] = {0, 0}, nrecv[2] = {0, 0};
    uint32_t* send_idx = nullptr;  /* [2][halo_cap] cell-order positions of the particles sent */
    uint32_t* halo_pos = nullptr;  /* [2*halo_cap] where the k-th received particle lives */
    Rec *sendbuf = nullptr, *recvbuf = nullptr; /* [2][halo_cap] */
    uint32_t *sendgid = nullptr, *recvgid = nullptr;
    char *h_send = nullptr, *h_recv = nullptr; /* pinned staging for host transports */
    uint32_t *flags = nullptr, *fpos = nullptr;
    hipStream_t comm = nullptr;
    hipEvent_t ev_packed = nullptr, ev_recvd = nullptr;
    sph_transport tr;
    std::atomic<bool> have_tr{false}; /* sph_use_rccl may run on a second host thread while the caller gives up waiting */
    double* red_dev = nullptr;
    uint32_t iL_end = 0, iR_begin = 0; /* owned ranges [0,iL_end) and [iR_begin,n) touch a halo */
    void* rccl = nullptr;              /* state of the built-in RCCL transport */
    void (*rccl_free)(void*) = nullptr;

    /* pair-kernel timing */
    bool profile = false;
    std::vector<EventPair> ev_pool;
    size_t ev_used = 0;
    std::vector<EventPair> rb_pool;
    size_t rb_used = 0;
    std::vector<EventPair> dr_pool; /* draws of the step lists */
    /* slab runs: a sweep's halo exchange (on the communication stream) and a step's agreement on
     * its scalars (on the compute stream), and the host's time in a rebuild's halo set-up and
     * migration (both wait for the device) */
    std::vector<EventPair> hx_pool, ar_pool;
    size_t hx_used = 0, ar_used = 0;
    double halo_setup_ms = 0, migrate_ms = 0;
    uint64_t halo_setups = 0, migrations = 0;
    size_t dr_used = 0;
};

namespace {

constexpr int BBOX_BLOCKS = 1024;

/* Scratch of sph_step_noalloc: the pair kernel's per-workgroup partials (PART_FIELDS per
 * workgroup) and their first fold, for cases of up to SPH_NOALLOC_MAX_PARTICLES particles.
 * One copy per GPU, held until the process ends.  In host code the array's name is only a
 * handle for hipGetSymbolAddress: the address kept per device below is what the launches get. */
constexpr uint32_t NOALLOC_PART_SLOTS = (SPH_NOALLOC_MAX_PARTICLES + PAIR_PARTICLES - 1) / PAIR_PARTICLES + 8 + 24;
constexpr size_t NOALLOC_SCRATCH_DOUBLES = (size_t)PART_FIELDS * NOALLOC_PART_SLOTS + (size_t)PART_FIELDS * PART_GROUPS;
static __device__ __align__(256) double g_noalloc_scratch[NOALLOC_SCRATCH_DOUBLES];
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
double* g_scratch_ptr[MAX_DEVICES] = {};

/* Device address of the current device's copy of the scratch array (looked up once per device). */
hipError_t scratch_of_current_device(int device, double** out) {
    if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        double* p = nullptr;
        const hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(g_noalloc_scratch));
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

#define HIPCHK(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            char b_[512];                                                                     \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),    \
                     __FILE__, __LINE__);                                                     \
            c->err = b_;                                                                      \
            return SPH_EDEVICE;                                                               \
        }                                                                                     \
    } while (0)

#define CHK(call)                     \
    do {                              \
        int r_ = (call);              \
        if (r_ != SPH_OK) return r_;  \
    } while (0)

inline uint32_t nblk(uint32_t n) { return std::max(1u, (n + SPH_BLOCK - 1) / SPH_BLOCK); }
/* workgroups of the list kernels (64 particles each) and of the pair kernel (PAIR_SPAN each) over
 * `count` particles, a multiple of 8 */
inline uint32_t pairblocks(uint32_t count) { return ((count + PAIR_PARTICLES - 1) / PAIR_PARTICLES + 7u) & ~7u; }
inline uint32_t pairgrid(uint32_t count) { return count ? ((count + PAIR_SPAN - 1) / PAIR_SPAN + 7u) & ~7u : 0u; }

int fail(sph_ctx* c, int code, const char* msg) {
    c->err = msg;
    return code;
}

template <typename T>
int dalloc(sph_ctx* c, T** p, size_t count) {
    if (c->no_alloc) return fail(c, SPH_ENOMEM, "this call may not allocate device memory and a buffer has to grow: make the same call through sph_step once");
    HIPCHK(hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T)));
    return SPH_OK;
}
/* frees *p unless the call under way may not (the dalloc that follows then reports it) */
template <typename T>
int dfree(sph_ctx* c, T** p) {
    if (c->no_alloc || !*p) return SPH_OK;
    HIPCHK(hipFree(*p));
    *p = nullptr;
    return SPH_OK;
}

/* index helpers for the n-state component arrays */
inline int NX(int d) { return d; }
inline int NV(int d) { return 3 + d; }
inline int NVV(int d) { return 6 + d; }
inline int NF(int d) { return 9 + d; }
enum { F_RHO = 0, F_RRHO = 1, F_P = 2, F_M = 3 };

inline uint32_t n_local(const sph_ctx* c) { return c->n + c->n_halo; }
/* the array the wall test reads: global ids in a slab run, upload positions otherwise */
inline const uint32_t* wall_ids(const sph_ctx* c) { return c->dist ? c->gid : c->id; }

int exclusive_scan(sph_ctx* c, const uint32_t* in, uint32_t* out, uint32_t n) {
    const uint32_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nb > c->bsum_cap) {
        CHK(dfree(c, &c->bsum));
        CHK(dalloc(c, &c->bsum, (size_t)nb * 2));
        c->bsum_cap = (size_t)nb * 2;
    }
    hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(SPH_BLOCK), 0, c->stream, in, n, c->bsum);
    hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(SPH_BLOCK), 0, c->stream, c->bsum, nb);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(SPH_BLOCK), 0, c->stream, in, out, n, c->bsum);
    HIPCHK(hipGetLastError());
    return SPH_OK;
}

/* Sorts m (key, position) pairs; on return `keys` holds the sorted keys and `perm` the old
 * position of each sorted entry (the four pointers are swapped as the passes ping-pong). */
int radix_sort(sph_ctx* c, uint32_t*& keys, uint32_t*& keys_alt, uint32_t*& perm, uint32_t*& perm_alt, uint32_t m,
               uint64_t ncells) {
    if (m == 0) return SPH_OK;
    const uint32_t nb = (m + RS_TILE - 1) / RS_TILE;
    const size_t hsz = (size_t)256 * nb;
    if (hsz > c->hist_cap) {
        CHK(dfree(c, &c->hist));
        CHK(dalloc(c, &c->hist, hsz));
        c->hist_cap = hsz;
    }
    int bits = 1;
    while (bits < 32 && (1ull << bits) < (unsigned long long)ncells) ++bits;
    const int passes = (bits + 7) / 8;
    const uint32_t* vin = nullptr;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        hipLaunchKernelGGL(k_rs_hist, dim3(nb), dim3(SPH_BLOCK), 0, c->stream, keys, m, shift, c->hist, nb);
        CHK(exclusive_scan(c, c->hist, c->hist, (uint32_t)hsz));
        hipLaunchKernelGGL(k_rs_scatter, dim3(nb), dim3(SPH_BLOCK), 0, c->stream, keys, vin, keys_alt, perm_alt, m, shift,
                           c->hist, nb);
        std::swap(keys, keys_alt);
        std::swap(perm, perm_alt);
        vin = perm;
    }
    HIPCHK(hipGetLastError());
    return SPH_OK;
}

void launch_hash(sph_ctx* c, const Rec* r, uint32_t m, const uint32_t* drop, uint32_t* keys) {
    if (m == 0) return;
    if (c->dim == 3) hipLaunchKernelGGL(k_hash<3>, dim3(nblk(m)), dim3(SPH_BLOCK), 0, c->stream, r, m, c->grid, drop, keys);
    else hipLaunchKernelGGL(k_hash<2>, dim3(nblk(m)), dim3(SPH_BLOCK), 0, c->stream, r, m, c->grid, drop, keys);
}

/* Where the 8-byte lattice entries live: in the spare record array (64 bytes per particle, free
 * between a reorder and the next one), a quarter of the way in -- the list build reads up to
 * fifteen entries past the last particle, which still lie inside that array. */
inline uint2* qpos_of(sph_ctx* c) { return (uint2*)((char*)c->rec_alt + (size_t)c->cap * 16); }

int ensure_step_lists(sph_ctx* c, bool* ok);

/* One run of the list build over the cell table as it stands: the rows of every owned particle
 * into `rows`/`counts` and, when `tin` is given, the entries within the second radius into
 * `tin`/`tcnt` as well.  range2: the squared distance the cell ranges are chosen for; rl2q, rin2q:
 * the two radii squared in lattice spacings.  *need_out: the longest row asked for (more than the
 * rows hold: the caller grows them and builds again).  The lattice entries must be in place. */
int run_list_build(sph_ctx* c, const Rec* r, double range2, uint32_t rl2q, uint32_t* rows, uint32_t* counts, int prune_walls,
                   uint32_t rin2q, uint32_t* tin, uint32_t* tcnt, uint32_t* need_out, int stale_cells = 0) {
    const uint32_t* hb = c->n_halo ? c->hbeg : nullptr;
    HIPCHK(hipMemsetAsync(c->need, 0, 2 * sizeof(uint32_t), c->stream));
    /* the particles whose records do not fit the build's LDS are named here for the ring kernel */
    uint32_t* const todo = (uint32_t*)c->tmp;
    /* the ring kernel over the `count` particles of `list` (nullptr: all of them) */
    auto ring = [&](const uint32_t* list, uint32_t count) {
        const dim3 grid(list ? (count + PAIR_PARTICLES - 1) / PAIR_PARTICLES : pairblocks(c->n));
        if (c->dim == 3)
            hipLaunchKernelGGL(k_build_list_ring<3>, grid, dim3(SPH_BLOCK), 0, c->stream, r, qpos_of(c), c->n, c->grid, c->cbeg, range2,
                               rl2q, rows, c->cap_k, c->cap_k, counts, c->need, prune_walls, hb, rin2q, tin, tcnt, list, count, stale_cells);
        else
            hipLaunchKernelGGL(k_build_list_ring<2>, grid, dim3(SPH_BLOCK), 0, c->stream, r, qpos_of(c), c->n, c->grid, c->cbeg, range2,
                               rl2q, rows, c->cap_k, c->cap_k, counts, c->need, prune_walls, hb, rin2q, tin, tcnt, list, count, stale_cells);
    };
    if (c->build_trips == 0) {
        ring(nullptr, 0u);
    } else if (c->dim == 3) {
        hipLaunchKernelGGL(k_build_list<3>, dim3(pairblocks(c->n)), dim3(SPH_BLOCK), 0, c->stream, r, qpos_of(c), c->n, c->grid, c->cbeg,
                           range2, rl2q, rows, c->cap_k, c->cap_k, counts, c->need, prune_walls, hb, rin2q, tin, tcnt, c->build_trips,
                           todo, stale_cells);
    } else {
        hipLaunchKernelGGL(k_build_list<2>, dim3(pairblocks(c->n)), dim3(SPH_BLOCK), 0, c->stream, r, qpos_of(c), c->n, c->grid, c->cbeg,
                           range2, rl2q, rows, c->cap_k, c->cap_k, counts, c->need, prune_walls, hb, rin2q, tin, tcnt, c->build_trips,
                           todo, stale_cells);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->need_host, c->need, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->need_host[1] > 0) {
        const uint32_t handed_on = c->need_host[1];
        if (handed_on > c->n) return fail(c, SPH_ESTATE, "list build: more particles handed on than there are");
        c->build_handed_on += handed_on;
        ring(todo, handed_on);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(c->need_host, c->need, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    *need_out = *c->need_host;
    return SPH_OK;
}

/* development check (SPH_CHECK_LISTS): every word of every row names a particle slot */
int check_rows(sph_ctx* c, const uint32_t* rows, const uint32_t* counts, const uint32_t* tin, const uint32_t* tcnt) {
    if (!c->check_lists) return SPH_OK;
    uint32_t* bad = (uint32_t*)c->tmp;
    HIPCHK(hipMemsetAsync(bad, 0, sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(k_check_lists, dim3(nblk(c->n)), dim3(SPH_BLOCK), 0, c->stream, rows, c->cap_k, c->cap_k, counts, c->n, c->cap, bad);
    if (tin)
        hipLaunchKernelGGL(k_check_lists, dim3(nblk(c->n)), dim3(SPH_BLOCK), 0, c->stream, tin, c->cap_k, c->cap_k, tcnt, c->n, c->cap, bad);
    uint32_t h = 0;
    HIPCHK(hipMemcpyAsync(&h, bad, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (h) return fail(c, SPH_ESTATE, "list check: a row names a slot beyond the arrays or holds too many entries");
    return SPH_OK;
}

/* with_step_lists: the build writes the step lists too (radius r_in2), on the same positions */
int build_lists(sph_ctx* c, const Rec* r, uint32_t wall_below, bool with_step_lists) {
    /* lattice entries and wall marks of the owned and the neighbours' particles (in the spare
     * record array, which nothing else uses between a reorder and the next one) */
    const uint32_t m = c->n + c->n_halo;
    const QLattice& L = c->lattice;
    hipLaunchKernelGGL(k_fill_qpos, dim3(nblk(m)), dim3(SPH_BLOCK), 0, c->stream, r, m, L, c->dim, wall_ids(c), c->dp.bound_parts,
                       qpos_of(c));
    /* radii in lattice spacings, widened by the two spacings the rounding can cost */
    const double rlq = std::sqrt(c->r_list2) * L.invq + 2.0, rinq = std::sqrt(c->r_in2) * L.invq + 2.0;
    const uint32_t rl2q = (uint32_t)std::ceil(rlq * rlq), rin2q = (uint32_t)std::ceil(rinq * rinq);
    if (c->cap_k & 3u) return fail(c, SPH_ESTATE, "list rows must hold a multiple of four entries");
    for (int attempt = 0; attempt < 4; ++attempt) {
        if (with_step_lists) { /* as many rows again, of the same length */
            bool ok = true;
            CHK(ensure_step_lists(c, &ok));
        }
        uint32_t* const tin = with_step_lists ? c->nbr_in : nullptr;
        uint32_t* const tcnt = with_step_lists ? c->cnt_in : nullptr;
        uint32_t need = 0;
        CHK(run_list_build(c, r, c->r_list2, rl2q, c->nbr, c->cnt, wall_below ? 1 : 0, rin2q, tin, tcnt, &need));
        if (need <= c->cap_k) return check_rows(c, c->nbr, c->cnt, tin, tcnt);
        /* some particle has more candidates than the lists hold: grow and build again */
        /* rows of whole 128-byte lines: a stride that is not a multiple of 64 bytes makes every
         * 16-entry group of a row straddle two lines (measured: a third more HBM reads in the pair
         * launch and in the draw, 6 % and 18 % more time) */
        const uint32_t newcap = (need + need / 8 + 31u) & ~31u;
        CHK(dfree(c, &c->nbr));
        CHK(dalloc(c, &c->nbr, (size_t)newcap * c->own_cap));
        HIPCHK(hipMemsetAsync(c->nbr, 0, sizeof(uint32_t) * (size_t)newcap * c->own_cap, c->stream));
        c->cap_k = newcap;
    }
    return fail(c, SPH_ESTATE, "candidate lists kept overflowing");
}

/* ---- transport helpers (slab runs) --------------------------------------------------------- */

/* Exchange device buffers with the slab neighbours on the communication stream.  Byte counts
 * of a side without a neighbour are zero. */
int exchange_bytes(sph_ctx* c, const void* sL, size_t nsL, void* rL, size_t nrL, const void* sR, size_t nsR, void* rR,
                   size_t nrR) {
    if (!c->tr.host_buffers) {
        if (c->tr.sendrecv(c->tr.user, sL, nsL, rL, nrL, sR, nsR, rR, nrR, (void*)c->comm) != 0)
            return fail(c, SPH_ECOMM, "transport sendrecv failed");
        return SPH_OK;
    }
    const size_t half = c->stage_half;
    if (nsL) HIPCHK(hipMemcpyAsync(c->h_send, sL, nsL, hipMemcpyDeviceToHost, c->comm));
    if (nsR) HIPCHK(hipMemcpyAsync(c->h_send + half, sR, nsR, hipMemcpyDeviceToHost, c->comm));
    HIPCHK(hipStreamSynchronize(c->comm));
    if (c->tr.sendrecv(c->tr.user, c->h_send, nsL, c->h_recv, nrL, c->h_send + half, nsR, c->h_recv + half, nrR, nullptr) != 0)
        return fail(c, SPH_ECOMM, "transport sendrecv failed");
    if (nrL) HIPCHK(hipMemcpyAsync(rL, c->h_recv, nrL, hipMemcpyHostToDevice, c->comm));
    if (nrR) HIPCHK(hipMemcpyAsync(rR, c->h_recv + half, nrR, hipMemcpyHostToDevice, c->comm));
    return SPH_OK;
}

/* In-place all-reduce of a few host doubles (op 0 = max, 1 = sum). */
int allreduce_host(sph_ctx* c, double* vals, int n, int op) {
    if (!c->dist) return SPH_OK;
    if (c->tr.host_buffers) {
        if (c->tr.allreduce(c->tr.user, vals, n, op, nullptr) != 0) return fail(c, SPH_ECOMM, "transport allreduce failed");
        return SPH_OK;
    }
    HIPCHK(hipMemcpyAsync(c->red_dev, vals, sizeof(double) * n, hipMemcpyHostToDevice, c->comm));
    if (c->tr.allreduce(c->tr.user, c->red_dev, n, op, (void*)c->comm) != 0)
        return fail(c, SPH_ECOMM, "transport allreduce failed");
    HIPCHK(hipMemcpyAsync(vals, c->red_dev, sizeof(double) * n, hipMemcpyDeviceToHost, c->comm));
    HIPCHK(hipStreamSynchronize(c->comm));
    return SPH_OK;
}

/* The step's agreement: red_dev[0..RED_NMAX) folded with max and the RED_NSUM values behind them
 * with a sum, over all ranks, in place on the device and in the order of the compute stream.
 * A device-buffer transport makes no host synchronisation here. */
int allreduce_step_scalars(sph_ctx* c) {
    double* const dmax = c->red_dev;
    double* const dsum = c->red_dev + RED_NMAX;
    if (!c->tr.host_buffers) {
        if (c->tr.allreduce_max_sum) {
            if (c->tr.allreduce_max_sum(c->tr.user, dmax, RED_NMAX, dsum, RED_NSUM, (void*)c->stream) != 0)
                return fail(c, SPH_ECOMM, "transport allreduce failed");
            return SPH_OK;
        }
        if (c->tr.allreduce(c->tr.user, dmax, RED_NMAX, 0, (void*)c->stream) != 0 ||
            c->tr.allreduce(c->tr.user, dsum, RED_NSUM, 1, (void*)c->stream) != 0)
            return fail(c, SPH_ECOMM, "transport allreduce failed");
        return SPH_OK;
    }
    double* const h = (double*)c->h_send; /* pinned */
    HIPCHK(hipMemcpyAsync(h, c->red_dev, sizeof(double) * (RED_NMAX + RED_NSUM), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    ++c->host_syncs;
    if (c->tr.allreduce_max_sum) {
        if (c->tr.allreduce_max_sum(c->tr.user, h, RED_NMAX, h + RED_NMAX, RED_NSUM, nullptr) != 0)
            return fail(c, SPH_ECOMM, "transport allreduce failed");
    } else if (c->tr.allreduce(c->tr.user, h, RED_NMAX, 0, nullptr) != 0 ||
               c->tr.allreduce(c->tr.user, h + RED_NMAX, RED_NSUM, 1, nullptr) != 0) {
        return fail(c, SPH_ECOMM, "transport allreduce failed");
    }
    HIPCHK(hipMemcpyAsync(c->red_dev, h, sizeof(double) * (RED_NMAX + RED_NSUM), hipMemcpyHostToDevice, c->stream));
    return SPH_OK;
}

/* Exchange the counts a rank is about to send to each side; returns what it will receive. */
int exchange_counts(sph_ctx* c, const uint32_t send[2], uint32_t recv[2]) {
    uint64_t s[2] = {send[0], send[1]}, r[2] = {0, 0};
    uint64_t* ds = (uint64_t*)c->red_dev; /* scratch: [0..1] send, [2..3] recv */
    HIPCHK(hipMemcpyAsync(ds, s, sizeof s, hipMemcpyHostToDevice, c->comm));
    HIPCHK(hipMemsetAsync(ds + 2, 0, 16, c->comm));
    CHK(exchange_bytes(c, ds, c->has_side[0] ? 8 : 0, ds + 2, c->has_side[0] ? 8 : 0, ds + 1, c->has_side[1] ? 8 : 0,
                       ds + 3, c->has_side[1] ? 8 : 0));
    HIPCHK(hipMemcpyAsync(r, ds + 2, sizeof r, hipMemcpyDeviceToHost, c->comm));
    HIPCHK(hipStreamSynchronize(c->comm));
    recv[0] = (uint32_t)std::min<uint64_t>(r[0], UINT32_MAX);
    recv[1] = (uint32_t)std::min<uint64_t>(r[1], UINT32_MAX);
    return SPH_OK;
}

/* Every rank reports a local failure; all of them stop if any did (nobody is left waiting). */
int agree_ok(sph_ctx* c, bool local_ok, int code, const char* msg) {
    double bad[1] = {local_ok ? 0.0 : 1.0};
    CHK(allreduce_host(c, bad, 1, 0));
    if (bad[0] != 0.0) return fail(c, code, msg);
    return SPH_OK;
}

/* Selects the owned particles each neighbour needs (x <= thr[0] for the left one, x >= thr[1]
 * for the right one; ascending cell-order positions), sends their records and global ids, and
 * places what arrives behind the owned particles of record buffer `src`, sorted by cell. */
int exchange_halo_at_rebuild(sph_ctx* c, int src, const double thr[2]) {
    const uint32_t n = c->n;
    uint32_t counts[2] = {0, 0};
    for (int side = 0; side < 2; ++side) {
        if (!c->has_side[side]) continue;
        hipLaunchKernelGGL(k_flag_halo, dim3(nblk(n)), dim3(SPH_BLOCK), 0, c->stream, c->rec[src], n, side, thr[side],
                           c->flags);
        CHK(exclusive_scan(c, c->flags, c->fpos, n));
        hipLaunchKernelGGL(k_compact, dim3(nblk(n)), dim3(SPH_BLOCK), 0, c->stream, c->flags, c->fpos, n, c->halo_cap,
                           c->send_idx + (size_t)side * c->halo_cap);
        HIPCHK(hipGetLastError());
        /* count = last scan value + last flag */
        HIPCHK(hipMemcpyAsync(&c->need_host[2 + side * 2], c->fpos + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(&c->need_host[3 + side * 2], c->flags + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        counts[side] = c->need_host[2 + side * 2] + c->need_host[3 + side * 2];
        if (counts[side] > c->halo_cap) counts[side] = UINT32_MAX; /* reported after the ranks have compared notes */
    }
    {
        /* an overflow anywhere stops every rank here, before anyone waits for a message */
        double over[1] = {(counts[0] == UINT32_MAX || counts[1] == UINT32_MAX) ? 1.0 : 0.0};
        CHK(allreduce_host(c, over, 1, 0));
        if (over[0] != 0.0) return fail(c, SPH_ENOMEM, "halo larger than the capacity given at sph_create (on some rank)");
    }
    c->nsend[0] = counts[0];
    c->nsend[1] = counts[1];
    /* pack records and global ids */
    for (int side = 0; side < 2; ++side) {
        const uint32_t m = c->nsend[side];
        if (!m) continue;
        const size_t off = (size_t)side * c->halo_cap;
        hipLaunchKernelGGL(k_gather_rec, dim3(nblk(m)), dim3(SPH_BLOCK), 0, c->stream, c->rec[src], c->send_idx + off, m,
                           c->sendbuf + off);
        hipLaunchKernelGGL(k_gather_u32, dim3(nblk(m)), dim3(SPH_BLOCK), 0, c->stream, c->gid, c->send_idx + off, m,
                           c->sendgid + off);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    {
        uint32_t arrive[2] = {0, 0};
        CHK(exchange_counts(c, c->nsend, arrive));
        CHK(agree_ok(c, arrive[0] <= c->halo_cap && arrive[1] <= c->halo_cap, SPH_ENOMEM,
                     "incoming halo larger than the capacity given at sph_create"));
        c->nrecv[0] = arrive[0];
        c->nrecv[1] = arrive[1];
    }
    const size_t hc = c->halo_cap;
    CHK(exchange_bytes(c, c->sendbuf, (size_t)c->nsend[0] * sizeof(Rec), c->recvbuf, (size_t)c->nrecv[0] * sizeof(Rec),
                       c->sendbuf + hc, (size_t)c->nsend[1] * sizeof(Rec), c->recvbuf + hc,
                       (size_t)c->nrecv[1] * sizeof(Rec)));
    HIPCHK(hipStreamSynchronize(c->comm));
    CHK(exchange_bytes(c, c->sendgid, (size_t)c->nsend[0] * 4, c->recvgid, (size_t)c->nrecv[0] * 4, c->sendgid + hc,
                       (size_t)c->nsend[1] * 4, c->recvgid + hc, (size_t)c->nrecv[1] * 4));
    HIPCHK(hipStreamSynchronize(c->comm));

    /* place behind the owned particles: left block, then right block */
    const uint32_t m = c->nrecv[0] + c->nrecv[1];
    CHK(agree_ok(c, (uint64_t)n + m <= c->cap, SPH_ENOMEM, "owned + halo particles exceed the capacity"));
    Rec* r = c->rec[src];
    if (c->nrecv[0]) {
        HIPCHK(hipMemcpyAsync(r + n, c->recvbuf, (size_t)c->nrecv[0] * sizeof(Rec), hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->gid + n, c->recvgid, (size_t)c->nrecv[0] * 4, hipMemcpyDeviceToDevice, c->stream));
    }
    if (c->nrecv[1]) {
        HIPCHK(hipMemcpyAsync(r + n + c->nrecv[0], c->recvbuf + hc, (size_t)c->nrecv[1] * sizeof(Rec), hipMemcpyDeviceToDevice,
                              c->stream));
        HIPCHK(hipMemcpyAsync(c->gid + n + c->nrecv[0], c->recvgid + hc, (size_t)c->nrecv[1] * 4, hipMemcpyDeviceToDevice,
                              c->stream));
    }
    c->n_halo = m;
    if (m == 0) return SPH_OK;
    hipLaunchKernelGGL(k_fill<uint32_t>, dim3(nblk(m)), dim3(SPH_BLOCK), 0, c->stream, c->id + n, SPH_HALO_ID, m);
    /* sort the halo region by cell */
    launch_hash(c, r + n, m, nullptr, c->keys + n);
    uint32_t *k = c->keys + n, *ka = c->keys_alt + n, *p = c->perm + n, *pa = c->perm_alt + n;
    CHK(radix_sort(c, k, ka, p, pa, m, (uint64_t)c->grid.ncells));
    if (k != c->keys + n) HIPCHK(hipMemcpyAsync(c->keys + n, k, (size_t)m * 4, hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL(k_reorder_halo, dim3(nblk(m)), dim3(SPH_BLOCK), 0, c->stream, p, m, r + n, c->rec_alt + n, c->gid + n,
                       c->gid_alt + n);
    HIPCHK(hipMemcpyAsync(r + n, c->rec_alt + n, (size_t)m * sizeof(Rec), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->gid + n, c->gid_alt + n, (size_t)m * 4, hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL(k_halo_pos, dim3(nblk(m)), dim3(SPH_BLOCK), 0, c->stream, p, m, n, c->halo_pos);
    hipLaunchKernelGGL(k_cell_start, dim3(nblk(c->grid.ncells + 1u)), dim3(SPH_BLOCK), 0, c->stream, c->keys + n, m, n,
                       c->grid.ncells, c->hbeg);
    HIPCHK(hipGetLastError());
    return SPH_OK;
}

MigArgs mig_args(sph_ctx* c, int src) {
    MigArgs a;
    memset(&a, 0, sizeof a);
    for (int q = 0; q < 12; ++q) a.nd[q] = c->nd[q];
    for (int q = 0; q < 6; ++q) a.pd[q] = c->pd[q];
    for (int q = 0; q < 4; ++q) a.nfl[q] = c->nfl[q];
    a.pRrho = c->pRrho;
    a.pp = c->pp;
    a.rec = c->rec[src];
    a.gid = c->gid;
    return a;
}

/* Hands the particles that left [slab_lo, slab_hi) to the neighbour on that side and appends
 * the ones the neighbours hand over behind the owned particles.  The leavers stay in place,
 * marked in c->drop; the sort that follows moves them behind every cell.  On return *m_sort
 * is the number of entries to sort and *n_new the number of particles owned from now on. */
int migrate_particles(sph_ctx* c, int src, uint32_t* m_sort, uint32_t* n_new) {
    const uint32_t n = c->n;
    const size_t hc = c->halo_cap;
    HIPCHK(hipMemsetAsync(c->drop, 0, sizeof(uint32_t) * c->cap, c->stream));
    uint32_t counts[2] = {0, 0};
    bool fits = true;
    for (int side = 0; side < 2; ++side) {
        if (!c->has_side[side]) continue;
        const double plane = side == 0 ? c->prm.slab_lo : c->prm.slab_hi;
        hipLaunchKernelGGL(k_flag_leavers, dim3(nblk(n)), dim3(SPH_BLOCK), 0, c->stream, c->rec[src], n, side, plane, c->flags,
                           c->drop);
        CHK(exclusive_scan(c, c->flags, c->fpos, n));
        hipLaunchKernelGGL(k_compact, dim3(nblk(n)), dim3(SPH_BLOCK), 0, c->stream, c->flags, c->fpos, n, c->halo_cap,
                           c->send_idx + side * hc);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&c->need_host[2], c->fpos + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(&c->need_host[3], c->flags + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        counts[side] = c->need_host[2] + c->need_host[3];
        if (counts[side] > c->halo_cap) fits = false;
    }
    if (counts[0] + counts[1] >= n) fits = false; /* a rank must keep at least one particle */
    CHK(agree_ok(c, fits, SPH_ENOMEM, "more particles leave a slab at once than the capacity given at sph_create allows"));
    MigArgs a = mig_args(c, src);
    for (int side = 0; side < 2; ++side)
        if (counts[side])
            hipLaunchKernelGGL(k_pack_mig, dim3(nblk(counts[side])), dim3(SPH_BLOCK), 0, c->stream, a, c->send_idx + side * hc,
                               counts[side], c->migsend + side * hc);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    uint32_t arrive[2] = {0, 0};
    CHK(exchange_counts(c, counts, arrive));
    const uint64_t na = (uint64_t)arrive[0] + arrive[1];
    const bool room = arrive[0] <= c->halo_cap && arrive[1] <= c->halo_cap &&
                      (uint64_t)n - counts[0] - counts[1] + na <= c->own_cap && (uint64_t)n + na <= c->cap;
    CHK(agree_ok(c, room, SPH_ENOMEM, "a slab received more particles than the capacity given at sph_create allows"));
    CHK(exchange_bytes(c, c->migsend, (size_t)counts[0] * sizeof(Mig), c->migrecv, (size_t)arrive[0] * sizeof(Mig),
                       c->migsend + hc, (size_t)counts[1] * sizeof(Mig), c->migrecv + hc, (size_t)arrive[1] * sizeof(Mig)));
    HIPCHK(hipStreamSynchronize(c->comm));
    if (arrive[0])
        hipLaunchKernelGGL(k_unpack_mig, dim3(nblk(arrive[0])), dim3(SPH_BLOCK), 0, c->stream, a, c->migrecv, arrive[0], n);
    if (arrive[1])
        hipLaunchKernelGGL(k_unpack_mig, dim3(nblk(arrive[1])), dim3(SPH_BLOCK), 0, c->stream, a, c->migrecv + hc, arrive[1],
                           n + arrive[0]);
    HIPCHK(hipGetLastError());
    if (na) /* positions in the caller's arrays mean nothing for newcomers; the order is renumbered below */
        hipLaunchKernelGGL(k_fill<uint32_t>, dim3(nblk((uint32_t)na)), dim3(SPH_BLOCK), 0, c->stream, c->id + n, 0u, (uint32_t)na);
    *m_sort = n + (uint32_t)na;
    *n_new = n - counts[0] - counts[1] + (uint32_t)na;
    c->migrated_out += counts[0] + counts[1];
    c->migrated_in += na;
    return SPH_OK;
}

/* Re-bin on the positions held in record buffer `src`, bring every live array into the new
 * cell order, (slab runs) fetch the neighbours' boundary particles, rebuild cell table and
 * candidate lists, reset the travel bound.  Collective in a slab run. */
int rebuild(sph_ctx* c, int src, bool for_step = false, bool may_migrate = false) {
    uint32_t n = c->n;
    const int dim = c->dim;
    if (c->dist && !c->have_tr) return fail(c, SPH_ESTATE, "no transport installed (sph_set_transport / sph_use_rccl)");
    EventPair* ev = nullptr;
    if (c->profile && !(c->no_alloc && c->rb_used == c->rb_pool.size())) {
        if (c->rb_used == c->rb_pool.size()) {
            EventPair e;
            HIPCHK(hipEventCreate(&e.a));
            HIPCHK(hipEventCreate(&e.b));
            c->rb_pool.push_back(e);
        }
        ev = &c->rb_pool[c->rb_used++];
        HIPCHK(hipEventRecord(ev->a, c->stream));
    }
    c->n_halo = 0; /* the neighbours' particles are fetched afresh below */
    c->list_valid = false; /* until the new lists stand: a rebuild that fails half-way leaves none */
    uint32_t m_sort = n;
    const uint32_t* drop = nullptr;
    bool renumber = false;
    if (c->dist && c->migrate && may_migrate && src == c->b0) {
        uint32_t n_new = n;
        const auto t0 = std::chrono::steady_clock::now();
        CHK(migrate_particles(c, src, &m_sort, &n_new));
        if (c->profile) {
            c->migrate_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            ++c->migrations;
        }
        renumber = m_sort != n || n_new != n;
        drop = c->drop;
        n = n_new; /* kernels below that walk the owned particles before the sort use m_sort and drop */
    }

    /* inside a step, wall-wall pairs can be dropped while the walls are at rest */
    double walls_move = 1.0;
    if (for_step && c->dp.bound_parts > 0) {
        WallRestArgs w;
        memset(&w, 0, sizeof w);
        w.n = m_sort;
        w.wall_below = c->dp.bound_parts;
        w.dim = dim;
        w.xsph = c->xsph;
        w.rec = c->rec[src];
        for (int d = 0; d < 3; ++d) {
            w.nv[d] = c->nd[NV(d)];
            w.nV[d] = c->nd[NVV(d)];
            w.nf[d] = c->nd[NF(d)];
        }
        w.id = wall_ids(c);
        HIPCHK(hipMemsetAsync(c->need, 0, sizeof(uint32_t), c->stream));
        hipLaunchKernelGGL(k_walls_at_rest, dim3(nblk(m_sort)), dim3(SPH_BLOCK), 0, c->stream, w, c->need);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(c->need_host, c->need, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    }
    /* bounding box */
    hipLaunchKernelGGL(k_bbox, dim3(BBOX_BLOCKS), dim3(SPH_BLOCK), 0, c->stream, c->rec[src], m_sort, dim, drop, c->bbox_part);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->bbox_host, c->bbox_part, sizeof(double) * 6 * BBOX_BLOCKS, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (for_step && c->dp.bound_parts > 0) walls_move = *c->need_host ? 1.0 : 0.0;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int b = 0; b < BBOX_BLOCKS; ++b)
        for (int d = 0; d < 3; ++d) {
            lo[d] = std::min(lo[d], c->bbox_host[b * 6 + d]);
            hi[d] = std::max(hi[d], c->bbox_host[b * 6 + 3 + d]);
        }
    c->step_lists_valid = false; /* the order of the particles changes */
    const double Rs = std::sqrt(c->prm.search_radius_sq);
    if (c->inner_mode == 0) {
        /* the step lists' margin: twice the travel seen between a step's second and last sweep
         * (2 % of the radius until there is a measurement) */
        c->inner_frac = c->in_travel > 0 ? std::min(0.08, std::max(0.004, 5.0 * c->in_travel / Rs)) : 0.02;
    }
    if (c->skin_auto && for_step && c->step_travel > 0) {
        /* Cost per step ~ pair work, of which a part grows like (1+s)^dim, + a rebuild every
         * 0.5*s*R/travel steps: least near s = sqrt(k * travel/R / dim), and lists should serve at
         * least two steps.  k = 2.2 (measured) when every sweep walks the candidates; with step
         * lists no sweep does (only the draw reads them, once per step), and a wider skin (k = 12) pays. */
        const double u = c->step_travel / Rs;
        const double s1 = std::min(0.5, std::max(0.03, std::max(std::sqrt(c->skin_k1 * u / dim), 5.0 * u)));
        const double s2 = std::min(c->skin_max, std::max(0.03, std::max(std::sqrt(c->skin_k2 * u / dim), 5.0 * u)));
        const bool two = c->inner_mode == 1 || (c->inner_mode == 0 && s2 >= c->two_level_min_skin);
        c->skin_frac = two ? s2 : s1;
    }
    c->two_level = for_step && c->inner_mode >= 0 && c->inner_frac < c->skin_frac &&
                   (c->inner_mode == 1 || c->skin_frac >= c->two_level_min_skin);
    /* lists that serve one evaluation on these very positions (Forces at t = 0, DensityReinit, the
     * neighbour probe) need no room for travel: the skin that the stepping path has grown in
     * developed flow would only make their rows 2-3 times longer */
    double skin_here = for_step ? c->skin_frac : std::min(c->skin_frac, 0.01);
    if (!for_step)
        if (const char* e = getenv("SPH_ONESHOT_SKIN")) skin_here = std::max(0.0, atof(e)); /* tests: long rows */
    const double r_list = std::sqrt(c->prm.search_radius_sq) * (1.0 + skin_here);
    const double reach = r_list * (1.0 + 1e-6); /* halo selection distance beyond a neighbour's extent */
    double thr_send[2] = {0, 0};
    if (c->dist) {
        /* The ranks agree on: whether wall-wall pairs may be dropped; the y/z extent of the
         * whole set (halo particles lie wherever the neighbours' particles are); and every
         * rank's x-extent.  A rank sends its neighbour what lies within `reach` of that
         * neighbour's extent, so particles that have drifted across a plane are still served. */
        const int R = c->nranks;
        std::vector<double> v(5 + 2 * (size_t)R, -1e300);
        v[0] = walls_move;
        v[1] = dim > 1 ? -lo[1] : 0; v[2] = dim > 1 ? hi[1] : 0;
        v[3] = dim > 2 ? -lo[2] : 0; v[4] = dim > 2 ? hi[2] : 0;
        v[5 + 2 * c->rank] = -lo[0];
        v[6 + 2 * c->rank] = hi[0];
        CHK(allreduce_host(c, v.data(), (int)v.size(), 0));
        walls_move = v[0];
        if (dim > 1) { lo[1] = -v[1]; hi[1] = v[2]; }
        if (dim > 2) { lo[2] = -v[3]; hi[2] = v[4]; }
        /* only adjacent slabs may interact; every rank sees the same table and stops together */
        for (int r = 0; r + 2 < R; ++r)
            if (!(-v[5 + 2 * (r + 2)] - v[6 + 2 * r] > reach))
                return fail(c, SPH_ESTATE, "slabs thinner than the interaction range: non-adjacent ranks would interact");
        if (c->has_side[0]) {
            thr_send[0] = v[6 + 2 * (c->rank - 1)] + reach; /* left neighbour's largest x, plus reach */
            lo[0] = lo[0] - reach;                           /* what the left neighbour sends: x >= this */
        }
        if (c->has_side[1]) {
            thr_send[1] = -v[5 + 2 * (c->rank + 1)] - reach;
            hi[0] = hi[0] + reach;
        }
    }
    for (int d = 0; d < dim; ++d)
        if (!(hi[d] >= lo[d])) { lo[d] = 0; hi[d] = 0; } /* no finite particle */

    /* Cells of edge >= the candidate radius (plus a relative margin that covers the rounding
     * of the cell-coordinate arithmetic).  The last axis is the run axis: cut `zf` times finer,
     * fastest in the sort key (see GridParams); in 3-D the columns are grouped in bricks of
     * 2^brick_log2 x 2^brick_log2 and the column axes padded to whole bricks.  If a stray
     * particle blew the box up, stretch the cells until the table fits its allocation. */
    GridParams g;
    memset(&g, 0, sizeof g);
    for (int d = 0; d < 3; ++d) {
        g.lo[d] = 0;
        g.inv[d] = 1;
        g.n[d] = 1;
        g.nt[d] = 1;
        g.ts[d] = 0;
    }
    const int run = dim - 1;
    const uint32_t tbits = dim == 3 ? c->brick_log2 : 0u;
    double edge = r_list * (1.0 + 1e-9);
    uint64_t ncells = 0;
    for (int attempt = 0; attempt < 200; ++attempt) {
        double tot = 1.0;
        for (int d = 0; d < dim; ++d) {
            const double cells = std::floor((hi[d] - lo[d]) / edge) + 1.0;
            const double tb = (double)(1u << tbits);
            tot *= d == run ? cells * c->run_fine : std::ceil(cells / tb) * tb;
        }
        if (tot + 1.0 <= (double)c->cell_cap && tot < 2147483648.0) {
            ncells = 1;
            for (int d = 0; d < dim; ++d) {
                const double cells = std::floor((hi[d] - lo[d]) / edge) + 1.0;
                const uint32_t tb = 1u << tbits;
                g.lo[d] = lo[d];
                g.inv[d] = 1.0 / edge;
                if (d == run) {
                    g.n[d] = (int)cells;
                    ncells *= (uint64_t)g.n[d] * c->run_fine;
                } else {
                    g.ts[d] = tbits;
                    g.nt[d] = (int)std::ceil(cells / tb);
                    g.n[d] = g.nt[d] * (int)tb;
                    ncells *= (uint64_t)g.n[d];
                }
            }
            break;
        }
        edge *= 1.26;
    }
    if (ncells == 0) return fail(c, SPH_ESTATE, "cell grid exceeds its table");
    g.ncells = (uint32_t)ncells;
    g.zf = c->run_fine;
    g.nrun = (uint32_t)g.n[run] * c->run_fine;
    g.invf = g.inv[run] * (double)c->run_fine;
    c->grid = g;
    /* the lattice of the draw's 8-byte entries, whose 32-bit byte offsets bound the array's size */
    for (int d = 0; d < 3; ++d) c->lattice.org[d] = d < dim ? g.lo[d] : 0.0;
    c->lattice.invq = DRAW_LATTICE_PER_RLIST / r_list;
    if (c->cap >= (1u << 29)) c->two_level = false;

    /* sort the owned particles by cell */
    launch_hash(c, c->rec[src], m_sort, drop, c->keys);
    CHK(radix_sort(c, c->keys, c->keys_alt, c->perm, c->perm_alt, m_sort, (uint64_t)g.ncells + (drop ? 1u : 0u)));
    /* the leavers sorted behind every cell; from here on this rank owns n particles */
    c->n = n;
    c->dp.n = n;

    /* physical reorder of every live array */
    ReorderArgs a;
    memset(&a, 0, sizeof a);
    a.perm = c->perm;
    a.n = n;
    int k = 0;
    int map_nd[12], map_pd[6];
    for (int q = 0; q < 4; ++q)
        for (int d = 0; d < dim; ++d) {
            if (q == 2 && !c->xsph) continue; /* no V in the single-mass variant */
            a.sd[k] = c->nd[q * 3 + d];
            a.dd[k] = c->nd_alt[q * 3 + d];
            map_nd[k] = q * 3 + d;
            ++k;
        }
    const int n_nd = k;
    /* f, V, Rrho and p of state n+1 hold nothing of their own after a committed step (see
     * settle_np1_outputs): they are filled from state n's, in the new order, when asked for */
    const bool outputs_live = !c->np1_out_in_n;
    for (int q = 0; q < 2 && outputs_live; ++q)
        for (int d = 0; d < dim; ++d) {
            if (q == 1 && !c->xsph) continue;
            a.sd[k] = c->pd[q * 3 + d];
            a.dd[k] = c->pd_alt[q * 3 + d];
            map_pd[k - n_nd] = q * 3 + d;
            ++k;
        }
    a.nd = k;
    for (int q = 0; q < 4; ++q) {
        a.sf[q] = c->nfl[q];
        a.df[q] = c->nfl_alt[q];
    }
    a.nf = 4;
    if (outputs_live) {
        a.sf[4] = c->pRrho;
        a.df[4] = c->pRrho_alt;
        a.sf[5] = c->pp;
        a.df[5] = c->pp_alt;
        a.nf = 6;
    }
    a.rs[0] = c->rec[c->b0];
    a.rd[0] = c->rec_alt;
    a.nr = 1;
    bool carried = false;
    if (src != c->b0) {
        /* positions came from a scratch buffer: carry it along through a free record buffer */
        int freeb = 0;
        while (freeb == src || freeb == c->b0) ++freeb;
        a.rs[1] = c->rec[src];
        a.rd[1] = c->rec[freeb];
        a.nr = 2;
        carried = true;
    }
    a.ids = c->id;
    a.idd = c->id_alt;
    if (c->dist) {
        a.gids = c->gid;
        a.gidd = c->gid_alt;
    }
    hipLaunchKernelGGL(k_reorder, dim3(nblk(n)), dim3(SPH_BLOCK), 0, c->stream, a);
    HIPCHK(hipGetLastError());
    for (int q = 0; q < n_nd; ++q) std::swap(c->nd[map_nd[q]], c->nd_alt[map_nd[q]]);
    for (int q = n_nd; q < k; ++q) std::swap(c->pd[map_pd[q - n_nd]], c->pd_alt[map_pd[q - n_nd]]);
    for (int q = 0; q < 4; ++q) std::swap(c->nfl[q], c->nfl_alt[q]);
    if (outputs_live) {
        std::swap(c->pRrho, c->pRrho_alt);
        std::swap(c->pp, c->pp_alt);
    }
    std::swap(c->rec[c->b0], c->rec_alt);
    std::swap(c->id, c->id_alt);
    if (c->dist) std::swap(c->gid, c->gid_alt);
    if (renumber) /* the set changed: the order of download/upload is this rebuild's cell order from now on */
        hipLaunchKernelGGL(k_iota, dim3(nblk(n)), dim3(SPH_BLOCK), 0, c->stream, c->id, n);
    if (carried) {
        int freeb = 0;
        while (freeb == src || freeb == c->b0) ++freeb;
        std::swap(c->rec[src], c->rec[freeb]);
    }

    /* cell table of the owned particles */
    hipLaunchKernelGGL(k_cell_start, dim3(nblk(g.ncells + 1u)), dim3(SPH_BLOCK), 0, c->stream, c->keys, n, 0u, g.ncells, c->cbeg);
    HIPCHK(hipGetLastError());

    if (c->dist) {
        const auto t0 = std::chrono::steady_clock::now();
        CHK(exchange_halo_at_rebuild(c, src, thr_send));
        if (c->profile) {
            c->halo_setup_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            ++c->halo_setups;
        }
        /* Owned particles that can see a halo particle, and all those sent to a neighbour, have
         * x <= thr_send[0] or x >= thr_send[1]: with bricks numbered x-slowest they sit in the
         * first / last few brick columns, contiguous ranges of the cell order. */
        /* keys per column of bricks along x (3-D: a row of bricks, 2-D: one x cell) */
        const uint32_t col = dim == 3 ? (((uint32_t)g.nt[1]) << (g.ts[0] + g.ts[1])) * g.nrun : g.nrun;
        const uint32_t cols = (uint32_t)g.nt[0];
        auto cellx = [&](double x) {
            double cc = std::floor((x - g.lo[0]) * g.inv[0]);
            cc = std::max(cc, 0.0);
            cc = std::min(cc, (double)(g.n[0] - 1));
            return (uint32_t)cc;
        };
        const uint32_t colsL = c->has_side[0] ? std::min(cols, (cellx(thr_send[0]) >> g.ts[0]) + 1u) : 0u;
        const uint32_t colR0 = c->has_side[1] ? (cellx(thr_send[1]) >> g.ts[0]) : cols;
        hipLaunchKernelGGL(k_lower_bounds, dim3(1), dim3(1), 0, c->stream, c->keys, n, colsL * col, colR0 * col, c->need + 2);
        HIPCHK(hipMemcpyAsync(c->need_host + 6, c->need + 2, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        c->iL_end = c->has_side[0] ? c->need_host[6] : 0;
        c->iR_begin = c->has_side[1] ? c->need_host[7] : n;
        if (c->iL_end > c->iR_begin) c->iL_end = c->iR_begin; /* thin slab: the two layers overlap, every particle is in one */
    }

    c->r_list2 = r_list * r_list;
    c->disp_limit = 0.5 * (r_list - std::sqrt(c->prm.search_radius_sq)) * 0.999;
    {
        /* step lists: radius R*(1+inner_frac), never beyond the candidate radius */
        const double R = std::sqrt(c->prm.search_radius_sq);
        const double r_in = std::min(R * (1.0 + c->inner_frac), r_list);
        c->r_in2 = r_in * r_in;
        c->limit_in = 0.5 * (r_in - R) * 0.999;
        /* a pair within r_in now was within r_list when the candidates were listed as long as
         * both have travelled at most this far since */
        c->outer_budget = 0.5 * (r_list - r_in) * 0.999;
    }
    const uint32_t wall_below = (for_step && c->dp.bound_parts > 0 && walls_move == 0.0) ? c->dp.bound_parts : 0u;
    CHK(build_lists(c, c->rec[src], wall_below, c->two_level));
    c->step_lists_valid = c->two_level; /* drawn on these very positions */
    c->lists_for_step = for_step;
    c->use_nbr = c->nbr;
    c->use_cnt = c->cnt;

    /* travel bound restarts */
    const double z[2] = {0.0, c->disp_limit};
    HIPCHK(hipMemcpyAsync(&c->S->disp, z, sizeof z, hipMemcpyHostToDevice, c->stream));
    const double zi[2] = {0.0, c->limit_in};
    HIPCHK(hipMemcpyAsync(&c->S->disp_in, zi, sizeof zi, hipMemcpyHostToDevice, c->stream));
    c->disp_host = 0;
    c->list_valid = true;
    ++c->rebuilds;
    if (c->trace)
        fprintf(stderr, "[sph] rebuild %llu: n %u skin %.4f two_level %d inner %.5f step_travel/R %.5f in_travel/R %.3e rows %u "
                        "handed on %llu\n",
                (unsigned long long)c->rebuilds, c->n, skin_here, (int)c->two_level, c->inner_frac,
                c->step_travel / std::sqrt(c->prm.search_radius_sq), c->in_travel / std::sqrt(c->prm.search_radius_sq), c->cap_k,
                (unsigned long long)c->build_handed_on);
    if (ev) HIPCHK(hipEventRecord(ev->b, c->stream));
    return SPH_OK;
}

OutState out_state(sph_ctx* c, int which) {
    OutState o;
    memset(&o, 0, sizeof o);
    for (int d = 0; d < 3; ++d) {
        o.f[d] = which == SPH_STATE_N ? c->nd[NF(d)] : c->pd[d];
        o.V[d] = which == SPH_STATE_N ? c->nd[NVV(d)] : c->pd[3 + d];
    }
    o.Rrho = which == SPH_STATE_N ? c->nfl[F_RRHO] : c->pRrho;
    return o;
}

NState n_state(sph_ctx* c) {
    NState s;
    for (int d = 0; d < 3; ++d) {
        s.x[d] = c->nd[NX(d)];
        s.v[d] = c->nd[NV(d)];
        s.V[d] = c->nd[NVV(d)];
        s.f[d] = c->nd[NF(d)];
    }
    s.rho = c->nfl[F_RHO];
    s.Rrho = c->nfl[F_RRHO];
    s.p = c->nfl[F_P];
    return s;
}

/* A committed step leaves f, V, Rrho (and in a slab run p) of state n+1 only in state n's arrays
 * (pn = pnp1: the same values, kept once -- see one_step).  Whoever is about to read state n+1's
 * arrays, or to change state n's, has them copied first.  The stepping path itself never reads
 * them (particles that migrate carry them along as they are, and the flag is the same on all ranks). */
int settle_np1_outputs(sph_ctx* c) {
    if (!c->np1_out_in_n) return SPH_OK;
    const size_t n = c->n;
    for (int d = 0; d < c->dim; ++d) {
        HIPCHK(hipMemcpyAsync(c->pd[d], c->nd[NF(d)], n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        if (c->xsph) HIPCHK(hipMemcpyAsync(c->pd[3 + d], c->nd[NVV(d)], n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    HIPCHK(hipMemcpyAsync(c->pRrho, c->nfl[F_RRHO], n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->pp, c->nfl[F_P], n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    c->np1_out_in_n = false;
    return SPH_OK;
}

/* An event pair from a pool that grows on demand (nullptr when not profiling, or when the pool is
 * used up in a call that may not allocate). */
EventPair* take_events(sph_ctx* c, std::vector<EventPair>& pool, size_t& used) {
    if (!c->profile || (c->no_alloc && used == pool.size())) return nullptr;
    if (used == pool.size()) {
        EventPair e;
        if (hipEventCreate(&e.a) != hipSuccess) return nullptr;
        if (hipEventCreate(&e.b) != hipSuccess) { (void)hipEventDestroy(e.a); return nullptr; }
        e.kind = 0;
        pool.push_back(e);
    }
    return &pool[used++];
}

/* Launches the pair kernel over particles [first, first+count); its workgroups leave their
 * partials in slots [part_base, part_base + grid) of each field of c->part. */
template <bool FUSED>
int launch_pair(sph_ctx* c, const Rec* rin, Rec* rout, OutState o, int store_out, int zero_wall, uint32_t first,
                uint32_t count, uint32_t part_base, uint32_t part_n) {
    if (count == 0) return SPH_OK;
    NState ns = n_state(c);
    dim3 g(pairgrid(count)), b(PAIR_BLOCK);
    /* record arrays of 4 GiB and more need 64-bit piece addresses */
    const bool wide = (uint64_t)c->cap * sizeof(Rec) >= (1ull << 32);
#define LP(D, X, W)                                                                                                  \
    hipLaunchKernelGGL((k_pair<D, X, FUSED, W>), g, b, 0, c->stream, c->dp, rin, c->use_nbr, c->cap_k, c->use_cnt, o, \
                       store_out, zero_wall, ns, wall_ids(c), rout, c->pp, c->S, c->part_use, first, count,          \
                       part_base, part_n)
    if (c->dim == 3) { if (wide) LP(3, true, true); else LP(3, true, false); }
    else if (c->xsph) { if (wide) LP(2, true, true); else LP(2, true, false); }
    else { if (wide) LP(2, false, true); else LP(2, false, false); }
#undef LP
    HIPCHK(hipGetLastError());
    return SPH_OK;
}

/* the fused sweep launch */
int launch_sweep(sph_ctx* c, const Rec* rin, Rec* rout, OutState o, int store_out, uint32_t first,
                 uint32_t count, uint32_t part_base, uint32_t part_n) {
    return launch_pair<true>(c, rin, rout, o, store_out, 1, first, count, part_base, part_n);
}

/* Draws the step lists from the candidate lists on the positions held in `r` (owned and halo
 * particles), and starts their travel bound. */
int draw_step_lists(sph_ctx* c, const Rec* r, bool entries_ready = false) {
    const uint32_t m = c->n + c->n_halo;
    EventPair* ev = nullptr;
    if (c->profile && !(c->no_alloc && c->dr_used == c->dr_pool.size())) {
        if (c->dr_used == c->dr_pool.size()) {
            EventPair e;
            HIPCHK(hipEventCreate(&e.a));
            HIPCHK(hipEventCreate(&e.b));
            c->dr_pool.push_back(e);
        }
        ev = &c->dr_pool[c->dr_used++];
        HIPCHK(hipEventRecord(ev->a, c->stream));
    }
    const QLattice& L = c->lattice;
    /* entries_ready: the sweep that wrote the owned particles' records left their lattice entries as
     * well; the neighbours' particles (behind the owned ones) arrived as records only */
    if (!entries_ready)
        hipLaunchKernelGGL(k_fill_qpos, dim3(nblk(m)), dim3(SPH_BLOCK), 0, c->stream, r, m, L, c->dim, nullptr, 0u, qpos_of(c));
    else if (c->n_halo)
        hipLaunchKernelGGL(k_fill_qpos, dim3(nblk(c->n_halo)), dim3(SPH_BLOCK), 0, c->stream, r + c->n, c->n_halo, L, c->dim, nullptr,
                           0u, qpos_of(c) + c->n);
    /* radius in lattice spacings, widened by the two spacings the rounding can cost */
    const double rq = std::sqrt(c->r_in2) * L.invq + 2.0;
    const uint32_t r2q = (uint32_t)std::ceil(rq * rq);
    const dim3 g(pairblocks(c->n)), b(SPH_BLOCK); /* 64 particles per workgroup, a multiple of 8 workgroups */
    if (c->dim == 3)
        hipLaunchKernelGGL(k_draw_lists<3>, g, b, 0, c->stream, qpos_of(c), c->n, c->nbr, c->cap_k, c->cnt, r2q, c->nbr_in,
                           c->cnt_in, c->S);
    else
        hipLaunchKernelGGL(k_draw_lists<2>, g, b, 0, c->stream, qpos_of(c), c->n, c->nbr, c->cap_k, c->cnt, r2q, c->nbr_in,
                           c->cnt_in, c->S);
    HIPCHK(hipGetLastError());
    if (ev) HIPCHK(hipEventRecord(ev->b, c->stream));
    if (c->trace && (c->draws++ % 16) == 0) { /* how the drawn rows fill the pair kernel's waves */
        unsigned long long* d = (unsigned long long*)c->tmp;
        unsigned long long h[8];
        HIPCHK(hipMemsetAsync(d, 0, sizeof h, c->stream));
        hipLaunchKernelGGL(k_rounds_stats, dim3(nblk(c->n)), dim3(SPH_BLOCK), 0, c->stream, c->cnt_in, c->n, d);
        hipLaunchKernelGGL(k_rounds_stats, dim3(nblk(c->n)), dim3(SPH_BLOCK), 0, c->stream, c->cnt, c->n, d + 4);
        HIPCHK(hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        const double waves = (c->n + 15) / 16;
        fprintf(stderr, "[sph] draw: step lists %.2f entries/particle, %.2f rounds/wave run for %.2f needed per particle "
                        "(%.2f with a workgroup's particles dealt to its waves by row length); "
                        "candidates %.2f entries/particle, %.2f rounds/wave\n",
                (double)h[2] / c->n, h[0] / waves, (double)h[1] / c->n, h[3] / waves, (double)h[6] / c->n, h[4] / waves);
    }
    return SPH_OK;
}

int pack_n_into(sph_ctx* c, Rec* dst) {
    PackArgs a;
    a.n = c->n;
    a.dim = c->dim;
    for (int d = 0; d < 3; ++d) {
        a.x[d] = c->nd[NX(d)];
        a.v[d] = c->nd[NV(d)];
    }
    a.rho = c->nfl[F_RHO];
    a.p = c->nfl[F_P];
    a.m = c->nfl[F_M];
    a.rec = dst;
    hipLaunchKernelGGL(k_pack, dim3(nblk(c->n)), dim3(SPH_BLOCK), 0, c->stream, a);
    HIPCHK(hipGetLastError());
    return SPH_OK;
}

int read_scalars(sph_ctx* c) {
    HIPCHK(hipMemcpyAsync(c->S_host, c->S, sizeof(DevScalars), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    ++c->host_syncs;
    return SPH_OK;
}

void fill_scalars(sph_ctx* c, sph_scalars* out) {
    if (!out) return;
    const DevScalars& s = *c->S_host;
    out->dt = s.dt;
    out->t = s.t;
    out->maxmu = s.maxmu;
    out->maxf = s.maxf;
    out->errsum = s.errsum;
    out->logbase = s.logbase;
    out->list_rebuilds = c->rebuilds;
    out->step_redos = c->redos;
    out->hit_pairs = s.hit_pairs;
    out->host_syncs = c->host_syncs;
}

/* One sweep's pair/update launches.  A single domain is one launch.  A slab first does the
 * layers next to its neighbours, hands their new records to the transport on the
 * communication stream, and computes its interior while they travel. */
int sweep_launches(sph_ctx* c, int cur, int nxt, bool last, bool commit, bool entries, uint32_t* part_n_out) {
    const uint32_t n = c->n;
    OutState o = out_state(c, SPH_STATE_NP1);
    if (entries) { /* the draw that follows reads the new positions' lattice entries */
        o.qpos = qpos_of(c);
        o.lat = c->lattice;
    }
    if (!c->dist) {
        /* commit: the launch also writes state n (see k_pair, store_out == 2) */
        *part_n_out = pairgrid(n);
        return launch_sweep(c, c->rec[cur], c->rec[nxt], o, last ? (commit ? 2 : 1) : 0, 0, n, 0, *part_n_out);
    }
    const uint32_t nL = c->iL_end, nR = n - c->iR_begin, nI = c->iR_begin - c->iL_end;
    const uint32_t bL = pairgrid(nL), bR = pairgrid(nR), bI = pairgrid(nI);
    const uint32_t part_n = bL + bR + bI;
    *part_n_out = part_n;
    CHK(launch_sweep(c, c->rec[cur], c->rec[nxt], o, last ? 1 : 0, 0, nL, 0, part_n));
    CHK(launch_sweep(c, c->rec[cur], c->rec[nxt], o, last ? 1 : 0, c->iR_begin, nR, bL, part_n));
    const size_t hc = c->halo_cap;
    for (int side = 0; side < 2; ++side)
        if (c->nsend[side])
            hipLaunchKernelGGL(k_gather_rec, dim3(nblk(c->nsend[side])), dim3(SPH_BLOCK), 0, c->stream, c->rec[nxt],
                               c->send_idx + side * hc, c->nsend[side], c->sendbuf + side * hc);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev_packed, c->stream));
    CHK(launch_sweep(c, c->rec[cur], c->rec[nxt], o, last ? 1 : 0, c->iL_end, nI, bL + bR, part_n));
    /* the exchange runs beside the interior launch */
    HIPCHK(hipStreamWaitEvent(c->comm, c->ev_packed, 0));
    EventPair* const hx = take_events(c, c->hx_pool, c->hx_used);
    if (hx) HIPCHK(hipEventRecord(hx->a, c->comm));
    CHK(exchange_bytes(c, c->sendbuf, (size_t)c->nsend[0] * sizeof(Rec), c->recvbuf, (size_t)c->nrecv[0] * sizeof(Rec),
                       c->sendbuf + hc, (size_t)c->nsend[1] * sizeof(Rec), c->recvbuf + hc,
                       (size_t)c->nrecv[1] * sizeof(Rec)));
    if (hx) HIPCHK(hipEventRecord(hx->b, c->comm));
    HIPCHK(hipEventRecord(c->ev_recvd, c->comm));
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_recvd, 0));
    /* received records go where the halo particles live in this rank's cell order; the
     * receive buffer holds the left block at 0 and the right block at halo_cap */
    if (c->nrecv[0])
        hipLaunchKernelGGL(k_scatter_rec, dim3(nblk(c->nrecv[0])), dim3(SPH_BLOCK), 0, c->stream, c->recvbuf, c->halo_pos, 0u,
                           c->nrecv[0], c->rec[nxt]);
    if (c->nrecv[1])
        hipLaunchKernelGGL(k_scatter_rec, dim3(nblk(c->nrecv[1])), dim3(SPH_BLOCK), 0, c->stream, c->recvbuf + hc,
                           c->halo_pos + c->nrecv[0], 0u, c->nrecv[1], c->rec[nxt]);
    HIPCHK(hipGetLastError());
    return SPH_OK;
}

/* Room for the step lists (as many rows as the candidate lists; sph_create makes it, and it is
 * made again when the candidate rows have grown). */
int ensure_step_lists(sph_ctx* c, bool* ok) {
    *ok = true;
    if (c->nbr_in && c->nbr_in_cap_k == c->cap_k && c->cnt_in) return SPH_OK;
    CHK(dfree(c, &c->nbr_in));
    CHK(dalloc(c, &c->nbr_in, (size_t)c->cap_k * c->own_cap));
    /* the pair kernel fetches whole groups of a row: every entry must index the arrays */
    HIPCHK(hipMemsetAsync(c->nbr_in, 0, sizeof(uint32_t) * (size_t)c->cap_k * c->own_cap, c->stream));
    c->nbr_in_cap_k = c->cap_k;
    if (!c->cnt_in) CHK(dalloc(c, &c->cnt_in, c->cap));
    c->step_lists_valid = false;
    return SPH_OK;
}

/* One Newmark_Beta call, or its first `nsweeps` sweeps when partial. */
int one_step(sph_ctx* c, int nsweeps, bool partial) {
    const int subits = c->prm.subits;
    if (!partial) nsweeps = subits;
    int cur = c->b0;
    bool allow_two = true; /* cleared when this step outran its step lists: the redo walks the candidates only */
    for (int mode = 0;; ++mode) {
        /* mode 0: keep the lists if the travel budget allows; 1: fresh lists at the step's
         * start; 2: fresh lists before every sweep (exact for any motion) */
        bool two = allow_two && mode < 2 && c->two_level && nsweeps >= 2;
        /* what is left of the budget must cover this step: the running estimate of a step's
         * travel (1.25 x) when there is one, half the budget otherwise */
        const double budget0 = two ? c->outer_budget : c->disp_limit;
        const bool enough = c->step_travel > 0 ? c->disp_host + 1.25 * c->step_travel <= budget0
                                               : c->disp_host <= 0.5 * budget0;
        if (mode >= 1 || !c->list_valid || !c->lists_for_step || !enough) {
            CHK(rebuild(c, c->b0, true, true));
            two = allow_two && mode < 2 && c->two_level && nsweeps >= 2;
        }
        if (two) CHK(ensure_step_lists(c, &two));
        const double cand_budget = two ? c->outer_budget : c->disp_limit;
        const double disp_at_start = c->disp_host; /* 0 after a rebuild */
        hipLaunchKernelGGL(k_step_begin, dim3(1), dim3(1), 0, c->stream, c->S);
        cur = c->b0;
        /* Two-level: every sweep walks the step lists.  They are drawn from the candidates on
         * the positions the step's second sweep starts from -- the later sweeps' positions
         * differ from those by dt^2-sized corrections, and the next step's first sweep starts
         * where this step ends -- and here, when the last draw does not cover the positions the
         * step starts from (after a rebuild, or when the last step ended beyond the margin).
         * Rows of the step lists keep the candidates' order and the kernel skips pairs outside
         * the radius either way, so every sum is the same with or without them. */
        if (two && !c->step_lists_valid) CHK(draw_step_lists(c, c->rec[cur]));
        for (int k = 0; k < nsweeps; ++k) {
            if (mode == 2 && k > 0) {
                /* the start state is not needed again: let the sweep's input take its place.  Not
                 * in a call that may not allocate: there a rebuild whose rows overflow ends the
                 * call, and the step is made again from its start through sph_step. */
                if (!c->no_alloc) c->b0 = cur;
                CHK(rebuild(c, cur, true));
            }
            int nxt = 0;
            while (nxt == cur || nxt == c->b0) ++nxt;
            const bool last = k == nsweeps - 1;
            c->use_nbr = two ? c->nbr_in : c->nbr;
            c->use_cnt = two ? c->cnt_in : c->cnt;
            EventPair* ev = nullptr;
            if (c->profile && !(c->no_alloc && c->ev_used == c->ev_pool.size())) {
                if (c->ev_used == c->ev_pool.size()) {
                    EventPair e;
                    HIPCHK(hipEventCreate(&e.a));
                    HIPCHK(hipEventCreate(&e.b));
                    c->ev_pool.push_back(e);
                }
                ev = &c->ev_pool[c->ev_used++];
                ev->kind = two ? 1 : 0;
                HIPCHK(hipEventRecord(ev->a, c->stream));
            }
            uint32_t part_n = 0;
            /* a single domain's last sweep commits the step itself; ranks first agree on whether
             * the step stands (below), and k_commit follows */
            const bool draw_next = two && k == 0 && !last;
            CHK(sweep_launches(c, cur, nxt, last, last && !partial && !c->dist, draw_next, &part_n));
            if (ev) HIPCHK(hipEventRecord(ev->b, c->stream));
            hipLaunchKernelGGL(k_fold_parts, dim3(PART_GROUPS, PART_FIELDS), dim3(SPH_BLOCK), 0, c->stream, c->part_use, part_n, PART_FIELDS,
                               c->part2_use);
            /* what reads the positions just written decides which travel bound is checked: the
             * next sweep through the candidate lists (out to R), the draw through the candidate
             * lists (out to the step lists' radius), or the next sweep through the step lists */
            double cand_limit = -1.0;
            if (mode < 2) {
                if (!two) cand_limit = c->disp_limit;
                else if (k == 0) cand_limit = c->outer_budget;
            }
            hipLaunchKernelGGL(k_sweep_finalize, dim3(1), dim3(1), 0, c->stream, c->dp, c->S, c->part2_use, k,
                               (last && !partial) ? 1 : (last ? 2 : 0), cand_limit, (two && k >= 1) ? 1 : 0, 0);
            if (draw_next) CHK(draw_step_lists(c, c->rec[nxt], true));
            cur = nxt;
        }
        CommitArgs a;
        memset(&a, 0, sizeof a);
        a.n = c->n;
        a.dim = c->dim;
        a.rec = c->rec[cur];
        for (int d = 0; d < 3; ++d) {
            a.nx[d] = c->nd[NX(d)];
            a.nv[d] = c->nd[NV(d)];
        }
        a.nrho = c->nfl[F_RHO];
        if (c->dist) {
            /* the ranks agree on the maxima, the sums and on whether to redo the step -- one
             * exchange, on the device, in stream order; the time step and the commit follow
             * from the agreed values, and the host looks at the result once */
            hipLaunchKernelGGL(k_pack_global, dim3(1), dim3(1), 0, c->stream, c->S, c->red_dev);
            EventPair* const ar = take_events(c, c->ar_pool, c->ar_used);
            if (ar) HIPCHK(hipEventRecord(ar->a, c->stream));
            CHK(allreduce_step_scalars(c));
            if (ar) HIPCHK(hipEventRecord(ar->b, c->stream));
            hipLaunchKernelGGL(k_apply_global, dim3(1), dim3(1), 0, c->stream, c->dp, c->S, c->red_dev, partial ? 0 : 1);
        }
        if (!partial) {
            if (c->dist) hipLaunchKernelGGL(k_commit, dim3(nblk(c->n)), dim3(SPH_BLOCK), 0, c->stream, a, c->S);
            hipLaunchKernelGGL(k_commit_scalars, dim3(1), dim3(1), 0, c->stream, c->S);
        }
        HIPCHK(hipGetLastError());
        CHK(read_scalars(c));
        const double disp_in_seen = c->S_host->disp_in;
        c->disp_host = c->S_host->disp;
        c->step_lists_valid = false;
        if (c->S_host->violation && mode < 2) {
            if (two && !(disp_in_seen <= c->limit_in)) {
                /* the later sweeps moved the particles beyond the step lists' margin */
                c->in_travel = std::max(1.25 * disp_in_seen, 2.0 * c->in_travel);
                allow_two = false;
            } else {
                /* the step outran its candidate lists: it travelled at least the whole budget */
                c->step_travel = std::max(c->step_travel, 1.25 * cand_budget);
            }
            ++c->redos;
            continue;
        }
        /* The last sweep wrote f, V, Rrho (and p): as state n's when it committed the step (a single
         * domain).  A slab run's sweep wrote them as state n+1's and the step stands: they become
         * state n's by changing places with the arrays that held state n's (which k_commit did not
         * touch); either way state n+1's arrays are now the ones to be filled on demand. */
        if (!partial && c->dist) {
            for (int d = 0; d < c->dim; ++d) {
                std::swap(c->pd[d], c->nd[NF(d)]);
                std::swap(c->pd[3 + d], c->nd[NVV(d)]);
            }
            std::swap(c->pRrho, c->nfl[F_RRHO]);
            std::swap(c->pp, c->nfl[F_P]);
        }
        c->np1_out_in_n = !partial;
        if (!partial) {
            /* mode 2 restarts the bound at every sweep; what it saw is a lower bound then */
            const double seen = c->disp_host - disp_at_start;
            if (seen == seen && seen >= 0) c->step_travel = std::max(seen, 0.8 * c->step_travel);
            if (two && disp_in_seen == disp_in_seen) {
                c->in_travel = std::max(disp_in_seen, 0.8 * c->in_travel);
                /* the lists drawn by this step serve the next step's first sweep if they cover
                 * the positions this step ended on */
                c->step_lists_valid = disp_in_seen <= c->limit_in;
            }
        }
        break;
    }
    c->b0 = cur;
    c->use_nbr = c->nbr;
    c->use_cnt = c->cnt;
    return SPH_OK;
}

template <typename T>
int put_component(sph_ctx* c, T* dst, size_t stride, const T* host) {
    const uint32_t m = n_local(c);
    HIPCHK(hipMemcpyAsync(c->tmp, host, sizeof(T) * c->n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_put<T>, dim3(nblk(m)), dim3(SPH_BLOCK), 0, c->stream, dst, stride, (const T*)c->tmp, c->id, m);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return SPH_OK;
}

template <typename T>
int get_component(sph_ctx* c, const T* src, size_t stride, T* host) {
    const uint32_t m = n_local(c);
    hipLaunchKernelGGL(k_get<T>, dim3(nblk(m)), dim3(SPH_BLOCK), 0, c->stream, src, stride, (T*)c->tmp, c->id, m);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(host, c->tmp, sizeof(T) * c->n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SPH_OK;
}

}  // namespace

/* ------------------------------------------------------------------------------------------ */

extern "C" {

const char* sph_last_error(const sph_ctx* c) { return c ? c->err.c_str() : "null context"; }

int sph_create(const sph_params* p, sph_ctx** out) {
    if (!p || !out) return SPH_EINVAL;
    *out = nullptr;
    sph_ctx* c = new sph_ctx();
    *out = c; /* returned even on failure so that the message can be read; destroy it either way */
    c->prm = *p;
    memset(&c->tr, 0, sizeof c->tr);
    if (p->dim != 2 && p->dim != 3) return fail(c, SPH_EINVAL, "dim must be 2 or 3");
    if (p->variant != SPH_VARIANT_WCSPH && p->variant != SPH_VARIANT_WCXSPH)
        return fail(c, SPH_EINVAL, "unknown variant");
    if (p->variant == SPH_VARIANT_WCSPH && p->dim != 2)
        return fail(c, SPH_EINVAL, "the single-mass variant exists in 2-D only");
    if (p->npts == 0 || p->npts >= (1ull << 31)) return fail(c, SPH_EINVAL, "npts out of range");
    c->dist = p->nranks > 1;
    if (!c->dist && p->bound_parts > p->npts) return fail(c, SPH_EINVAL, "bound_parts > npts");
    if (!(p->H > 0) || !(p->search_radius_sq > 0) || !(p->rho0 > 0) || p->subits < 1 || p->subits > 64 ||
        p->g_axis < 0 || p->g_axis >= p->dim)
        return fail(c, SPH_EINVAL, "bad physical constants");
    if (c->dist) {
        if (p->rank < 0 || p->rank >= p->nranks) return fail(c, SPH_EINVAL, "rank out of range");
        if (p->npts_global < p->npts || p->npts_global >= (1ull << 32)) return fail(c, SPH_EINVAL, "npts_global out of range");
        if (p->capacity <= p->npts || p->capacity >= (1ull << 31)) return fail(c, SPH_EINVAL, "capacity must exceed npts (room for the halo)");
        if (!(p->slab_hi > p->slab_lo)) return fail(c, SPH_EINVAL, "empty slab interval");
        c->rank = p->rank;
        c->nranks = p->nranks;
        c->has_side[0] = p->rank > 0;
        c->has_side[1] = p->rank < p->nranks - 1;
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(c, SPH_EDEVICE, "no HIP device: the pair path has no CPU fallback");
    if (p->device >= 0) {
        if (p->device >= ndev) return fail(c, SPH_EINVAL, "device ordinal out of range");
        HIPCHK(hipSetDevice(p->device));
    }
    HIPCHK(hipGetDevice(&c->device));
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));

    c->dim = p->dim;
    c->xsph = p->variant == SPH_VARIANT_WCXSPH;
    c->n = (uint32_t)p->npts;
    c->cap = c->dist ? (uint32_t)p->capacity : c->n;
    /* of the room beyond the first owned set: half for growth by migration, a quarter for each side's halo */
    c->halo_cap = (c->cap - c->n) / 4;
    c->own_cap = c->cap - 2 * c->halo_cap;
    c->migrate = c->dist && p->migrate != 0;
    c->npairblocks = pairblocks(c->own_cap) + 24; /* up to three launches per sweep, each padded to 8 */
    if (p->skin_frac > 0) {
        c->skin_frac = p->skin_frac;
        c->skin_auto = false;
    }
    if (p->inner_skin_frac > 0) {
        c->inner_frac = p->inner_skin_frac;
        c->inner_mode = 1;
    } else if (p->inner_skin_frac < 0) {
        c->inner_mode = -1;
    }
    if (const char* e = getenv("SPH_TRACE")) c->trace = atoi(e) != 0;
    if (const char* e = getenv("SPH_CHECK_LISTS")) c->check_lists = atoi(e) != 0;
    if (const char* e = getenv("SPH_REINIT_LISTS")) c->ri_fresh_lists = !strcmp(e, "fresh");
    if (const char* e = getenv("SPH_NEIGHBOURS_LISTS")) c->probe_standing = !strcmp(e, "standing");
    if (const char* e = getenv("SPH_BUILD_TRIPS")) c->build_trips = (uint32_t)std::min<long>(BUILD_TRIPS, std::max<long>(0, atol(e)));
    if (const char* e = getenv("SPH_TWO_LEVEL_MIN_SKIN")) c->two_level_min_skin = atof(e);
    c->skin_k2_auto = getenv("SPH_SKIN_K2") == nullptr;
    if (const char* e = getenv("SPH_SKIN_K2")) c->skin_k2 = atof(e);
    if (const char* e = getenv("SPH_SKIN_MAX")) c->skin_max = std::min(1.0, std::max(0.1, atof(e)));
    c->brick_log2 = c->dim == 3 ? 2u : 3u;
    if (const char* e = getenv("SPH_BRICK_LOG2")) c->brick_log2 = (uint32_t)std::min(5, std::max(0, atoi(e)));
    if (const char* e = getenv("SPH_RUN_FINE")) c->run_fine = (uint32_t)std::min(16, std::max(1, atoi(e)));

    DevParams& d = c->dp;
    memset(&d, 0, sizeof d);
    d.dim = p->dim;
    d.xsph = c->xsph;
    d.n = c->n;
    d.bound_parts = (uint32_t)p->bound_parts;
    d.H = p->H;
    d.HSQ = p->HSQ;
    d.correc = p->correc;
    d.R2 = p->search_radius_sq;
    d.Bgam = p->B * p->gam;
    d.B = p->B;
    d.gam = p->gam;
    d.rho0 = p->rho0;
    d.Cs = p->Cs;
    d.alpha = p->alpha;
    d.eps = p->eps;
    d.g = p->g;
    d.g_axis = p->g_axis;
    d.subits = p->subits;
    d.beta = p->beta;
    {
        const float bf = (float)p->beta;
        const float o = 1 - 2 * bf; /* float arithmetic, as `(1-2*beta)` with a float beta */
        d.omb = (double)o;
    }
    d.mass = p->mass;
    d.hvisc = 0.01 * p->HSQ;
    d.npts_d = 1.0 * (double)(c->dist ? p->npts_global : p->npts);
    d.sqrtBgam = std::sqrt(d.Bgam);
    d.invH = 1.0 / p->H;
    d.gconst = 5.0 * (1.0 / p->HSQ) * p->correc;

    const size_t n = c->cap;
    for (int q = 0; q < 4; ++q)
        for (int k = 0; k < c->dim; ++k) {
            CHK(dalloc(c, &c->nd[q * 3 + k], n));
            CHK(dalloc(c, &c->nd_alt[q * 3 + k], n));
            HIPCHK(hipMemsetAsync(c->nd[q * 3 + k], 0, n * sizeof(double), c->stream));
        }
    for (int q = 0; q < 4; ++q) {
        /* an even count: the pair kernel reads these as aligned pairs of floats */
        CHK(dalloc(c, &c->nfl[q], n + (n & 1)));
        CHK(dalloc(c, &c->nfl_alt[q], n + (n & 1)));
        HIPCHK(hipMemsetAsync(c->nfl[q], 0, (n + (n & 1)) * sizeof(float), c->stream));
        HIPCHK(hipMemsetAsync(c->nfl_alt[q], 0, (n + (n & 1)) * sizeof(float), c->stream));
    }
    for (int b = 0; b < NREC; ++b) {
        CHK(dalloc(c, &c->rec[b], n));
        HIPCHK(hipMemsetAsync(c->rec[b], 0, n * sizeof(Rec), c->stream));
    }
    CHK(dalloc(c, &c->rec_alt, n));
    HIPCHK(hipMemsetAsync(c->rec_alt, 0, n * sizeof(Rec), c->stream));
    for (int q = 0; q < 2; ++q)
        for (int k = 0; k < c->dim; ++k) {
            CHK(dalloc(c, &c->pd[q * 3 + k], n));
            CHK(dalloc(c, &c->pd_alt[q * 3 + k], n));
            HIPCHK(hipMemsetAsync(c->pd[q * 3 + k], 0, n * sizeof(double), c->stream));
        }
    CHK(dalloc(c, &c->pRrho, n + (n & 1))); /* as long as state n's float arrays: they change places with them */
    CHK(dalloc(c, &c->pRrho_alt, n + (n & 1)));
    HIPCHK(hipMemsetAsync(c->pRrho, 0, n * sizeof(float), c->stream));
    CHK(dalloc(c, &c->pp, n + (n & 1)));
    CHK(dalloc(c, &c->pp_alt, n + (n & 1)));
    HIPCHK(hipMemsetAsync(c->pp, 0, n * sizeof(float), c->stream));
    CHK(dalloc(c, &c->id, n));
    CHK(dalloc(c, &c->id_alt, n));
    hipLaunchKernelGGL(k_iota, dim3(nblk(c->n)), dim3(SPH_BLOCK), 0, c->stream, c->id, c->n);
    CHK(dalloc(c, &c->keys, n));
    CHK(dalloc(c, &c->keys_alt, n));
    CHK(dalloc(c, &c->perm, n));
    CHK(dalloc(c, &c->perm_alt, n));
    c->cell_cap = std::max<size_t>(8 * n, (size_t)1 << 20);
    CHK(dalloc(c, &c->cbeg, c->cell_cap));
    c->cap_k = p->max_neighbours > 0 ? (uint32_t)p->max_neighbours : (c->dim == 3 ? 96u : 48u); /* defaults: whole 64-byte groups */
    c->cap_k = (c->cap_k + 3u) & ~3u;
    CHK(dalloc(c, &c->nbr, (size_t)c->cap_k * c->own_cap));
    /* the pair kernel fetches whole groups of a row: every entry, used or not, must index the arrays */
    HIPCHK(hipMemsetAsync(c->nbr, 0, sizeof(uint32_t) * (size_t)c->cap_k * c->own_cap, c->stream));
    CHK(dalloc(c, &c->cnt, n));
    if (c->inner_mode >= 0) { /* the step lists: as many rows again */
        bool ok = true;
        CHK(ensure_step_lists(c, &ok));
    }
    CHK(dalloc(c, &c->need, 8));
    CHK(dalloc(c, &c->S, 1));
    HIPCHK(hipHostMalloc((void**)&c->S_host, sizeof(DevScalars)));
    HIPCHK(hipHostMalloc((void**)&c->bbox_host, sizeof(double) * 6 * BBOX_BLOCKS));
    HIPCHK(hipHostMalloc((void**)&c->need_host, 8 * sizeof(uint32_t)));
    CHK(dalloc(c, &c->part, (size_t)PART_FIELDS * c->npairblocks));
    CHK(dalloc(c, &c->part2, (size_t)PART_FIELDS * PART_GROUPS));
    c->part_use = c->part;
    c->part2_use = c->part2;
    HIPCHK(scratch_of_current_device(c->device, &c->scratch));
    /* the sort's histogram and the scans' block sums at their largest, so that a list rebuild
     * never allocates */
    c->hist_cap = (size_t)256 * ((n + RS_TILE - 1) / RS_TILE);
    CHK(dalloc(c, &c->hist, c->hist_cap));
    c->bsum_cap = 2 * std::max((c->hist_cap + SCAN_TILE - 1) / SCAN_TILE, (n + SCAN_TILE - 1) / SCAN_TILE);
    CHK(dalloc(c, &c->bsum, c->bsum_cap));
    CHK(dalloc(c, &c->bbox_part, (size_t)6 * BBOX_BLOCKS));
    HIPCHK(hipMalloc(&c->tmp, n * 8));
    if (c->dist) {
        const size_t hc = c->halo_cap;
        CHK(dalloc(c, &c->gid, n));
        CHK(dalloc(c, &c->gid_alt, n));
        CHK(dalloc(c, &c->hbeg, c->cell_cap));
        CHK(dalloc(c, &c->send_idx, 2 * hc));
        CHK(dalloc(c, &c->halo_pos, 2 * hc));
        CHK(dalloc(c, &c->sendbuf, 2 * hc));
        CHK(dalloc(c, &c->recvbuf, 2 * hc));
        CHK(dalloc(c, &c->sendgid, 2 * hc));
        CHK(dalloc(c, &c->recvgid, 2 * hc));
        CHK(dalloc(c, &c->flags, n));
        CHK(dalloc(c, &c->fpos, n));
        CHK(dalloc(c, &c->red_dev, 64 + 2 * (size_t)c->nranks));
        CHK(dalloc(c, &c->drop, n));
        CHK(dalloc(c, &c->migsend, 2 * hc));
        CHK(dalloc(c, &c->migrecv, 2 * hc));
        c->stage_half = hc * sizeof(Mig) + 64;
        HIPCHK(hipHostMalloc((void**)&c->h_send, 2 * c->stage_half));
        HIPCHK(hipHostMalloc((void**)&c->h_recv, 2 * c->stage_half));
        HIPCHK(hipStreamCreateWithFlags(&c->comm, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_recvd, hipEventDisableTiming));
    }

    memset(c->S_host, 0, sizeof(DevScalars));
    c->S_host->dt = p->dt0;
    c->S_host->dt_next = p->dt0;
    HIPCHK(hipMemcpyAsync(c->S, c->S_host, sizeof(DevScalars), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->skin_k2_auto) {
        /* The wider skin (k = 24: a rebuild every seventh step instead of every fifth in the dam
         * break, 3.5 % less time per step) needs rows of up to ~224 entries in both sets of
         * lists, 1.8 KB per particle; a case that fills the GPU keeps the narrower one. */
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        c->skin_k2 = (double)c->own_cap * 1800.0 <= 0.5 * (double)free_b ? 24.0 : 12.0;
    }
    return SPH_OK;
}

void sph_destroy(sph_ctx* c) {
    if (!c) return;
    if (c->lost) { /* anything that waits for the stream would wait for ever */
        delete c;
        return;
    }
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)hipStreamSynchronize(c->comm);
    if (c->rccl && c->rccl_free) c->rccl_free(c->rccl);
    for (int q = 0; q < 12; ++q) {
        if (c->nd[q]) (void)hipFree(c->nd[q]);
        if (c->nd_alt[q]) (void)hipFree(c->nd_alt[q]);
    }
    for (int q = 0; q < 4; ++q) {
        if (c->nfl[q]) (void)hipFree(c->nfl[q]);
        if (c->nfl_alt[q]) (void)hipFree(c->nfl_alt[q]);
    }
    for (int b = 0; b < NREC; ++b)
        if (c->rec[b]) (void)hipFree(c->rec[b]);
    if (c->rec_alt) (void)hipFree(c->rec_alt);
    for (int q = 0; q < 6; ++q) {
        if (c->pd[q]) (void)hipFree(c->pd[q]);
        if (c->pd_alt[q]) (void)hipFree(c->pd_alt[q]);
    }
    void* ptrs[] = {c->pp,   c->pp_alt, c->pRrho, c->pRrho_alt, c->id,       c->id_alt,   c->gid,     c->gid_alt, c->keys,
                    c->keys_alt, c->perm, c->perm_alt, c->hist, c->bsum,     c->cbeg,     c->hbeg,
                    c->nbr,  c->cnt,    c->need,  c->S,         c->part,     c->part2,    c->bbox_part, c->tmp,   c->send_idx,
                    c->halo_pos, c->sendbuf, c->recvbuf, c->sendgid, c->recvgid, c->flags, c->fpos,   c->red_dev, c->drop, c->migsend, c->migrecv,
                    c->ri_pend, c->ri_queue, c->ri_ctl, c->ri_flow, c->nbr_in, c->cnt_in};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (c->S_host) (void)hipHostFree(c->S_host);
    if (c->bbox_host) (void)hipHostFree(c->bbox_host);
    if (c->need_host) (void)hipHostFree(c->need_host);
    if (c->h_send) (void)hipHostFree(c->h_send);
    if (c->h_recv) (void)hipHostFree(c->h_recv);
    for (auto& e : c->ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto& e : c->rb_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto& e : c->dr_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto& e : c->hx_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto& e : c->ar_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (c->ev_packed) (void)hipEventDestroy(c->ev_packed);
    if (c->ev_recvd) (void)hipEventDestroy(c->ev_recvd);
    if (c->comm) (void)hipStreamDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int sph_set_transport(sph_ctx* c, const sph_transport* t) {
    if (!c || !t) return SPH_EINVAL;
    if (t->struct_size != sizeof(sph_transport))
        return fail(c, SPH_EINVAL, "sph_transport.struct_size is not sizeof(sph_transport): zero the struct, set struct_size, "
                                   "then the callbacks (a caller built against another version of the header?)");
    if (!t->sendrecv || !t->allreduce) return fail(c, SPH_EINVAL, "sph_transport: sendrecv and allreduce are required");
    if (!c->dist) return fail(c, SPH_ESTATE, "a transport is only used when nranks > 1");
    c->tr = *t;
    c->have_tr = true;
    return SPH_OK;
}

int sph_upload_global_ids(sph_ctx* c, const uint32_t* gids) {
    if (!c || !gids) return SPH_EINVAL;
    if (!c->dist) return fail(c, SPH_ESTATE, "global ids are only used when nranks > 1");
    HIPCHK(hipSetDevice(c->device));
    c->list_valid = false;
    return put_component<uint32_t>(c, c->gid, 1, gids);
}

int sph_local_count(sph_ctx* c, uint64_t* npts, uint64_t* migrated_in, uint64_t* migrated_out) {
    if (!c || !npts) return SPH_EINVAL;
    *npts = c->n;
    if (migrated_in) *migrated_in = c->migrated_in;
    if (migrated_out) *migrated_out = c->migrated_out;
    return SPH_OK;
}

int sph_halo_counts(sph_ctx* c, uint64_t* sent2, uint64_t* received2) {
    if (!c || !sent2 || !received2) return SPH_EINVAL;
    for (int s = 0; s < 2; ++s) {
        sent2[s] = c->dist ? c->nsend[s] : 0;
        received2[s] = c->dist ? c->nrecv[s] : 0;
    }
    return SPH_OK;
}

int sph_set_slab_interval(sph_ctx* c, double slab_lo, double slab_hi) {
    if (!c) return SPH_EINVAL;
    if (!c->dist) return fail(c, SPH_ESTATE, "slab intervals are only used when nranks > 1");
    if (!(slab_hi > slab_lo)) return fail(c, SPH_EINVAL, "empty slab interval");
    c->prm.slab_lo = slab_lo;
    c->prm.slab_hi = slab_hi;
    c->list_valid = false; /* the next step rebuilds its lists and, with migrate on, hands particles over */
    return SPH_OK;
}

int sph_download_global_ids(sph_ctx* c, uint32_t* gids) {
    if (!c || !gids) return SPH_EINVAL;
    if (!c->dist) return fail(c, SPH_ESTATE, "global ids are only used when nranks > 1");
    HIPCHK(hipSetDevice(c->device));
    return get_component<uint32_t>(c, c->gid, 1, gids);
}

int sph_upload(sph_ctx* c, int32_t which, const sph_fields* s) {
    if (c && c->lost) return fail(c, SPH_ESTATE, "the context is lost (an earlier call reported a launch that never came back)");
    if (!c || !s) return SPH_EINVAL;
    if (which != SPH_STATE_N && which != SPH_STATE_NP1) return fail(c, SPH_EINVAL, "bad state selector");
    HIPCHK(hipSetDevice(c->device));
    CHK(settle_np1_outputs(c));
    const int dim = c->dim;
    c->list_valid = false; /* positions or the walls' state may change */
    if (which == SPH_STATE_N) {
        for (int d = 0; d < dim; ++d) {
            if (s->x[d]) CHK(put_component<double>(c, c->nd[NX(d)], 1, s->x[d]));
            if (s->v[d]) CHK(put_component<double>(c, c->nd[NV(d)], 1, s->v[d]));
            if (s->V[d] && c->xsph) CHK(put_component<double>(c, c->nd[NVV(d)], 1, s->V[d]));
            if (s->f[d]) CHK(put_component<double>(c, c->nd[NF(d)], 1, s->f[d]));
        }
        if (s->rho) CHK(put_component<float>(c, c->nfl[F_RHO], 1, s->rho));
        if (s->p) CHK(put_component<float>(c, c->nfl[F_P], 1, s->p));
        if (s->Rrho) CHK(put_component<float>(c, c->nfl[F_RRHO], 1, s->Rrho));
        if (s->m) CHK(put_component<float>(c, c->nfl[F_M], 1, s->m));
        c->uploaded_n = true;
    } else {
        Rec* r = c->rec[c->b0];
        for (int d = 0; d < dim; ++d) {
            if (s->x[d]) CHK(put_component<double>(c, &r->xv[d][0], 8, s->x[d]));
            if (s->v[d]) CHK(put_component<double>(c, &r->xv[d][1], 8, s->v[d]));
            if (s->f[d]) CHK(put_component<double>(c, c->pd[d], 1, s->f[d]));
            if (s->V[d] && c->xsph) CHK(put_component<double>(c, c->pd[3 + d], 1, s->V[d]));
        }
        if (s->rho) CHK(put_component<float>(c, &r->rho, 16, s->rho));
        if (s->p) CHK(put_component<float>(c, c->pp, 1, s->p));
        if (s->m) CHK(put_component<float>(c, &r->m, 16, s->m));
        if (s->Rrho) CHK(put_component<float>(c, c->pRrho, 1, s->Rrho));
        if (s->rho || s->p) {
            hipLaunchKernelGGL(k_fix_prr, dim3(nblk(c->n)), dim3(SPH_BLOCK), 0, c->stream, r, c->pp, c->n);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(c->stream));
        }
        c->uploaded_np1 = true;
    }
    return SPH_OK;
}

int sph_download(sph_ctx* c, int32_t which, const sph_fields* s) {
    if (c && c->lost) return fail(c, SPH_ESTATE, "the context is lost (an earlier call reported a launch that never came back)");
    if (!c || !s) return SPH_EINVAL;
    if (which != SPH_STATE_N && which != SPH_STATE_NP1) return fail(c, SPH_EINVAL, "bad state selector");
    HIPCHK(hipSetDevice(c->device));
    CHK(settle_np1_outputs(c));
    const int dim = c->dim;
    if (which == SPH_STATE_N) {
        for (int d = 0; d < dim; ++d) {
            if (s->x[d]) CHK(get_component<double>(c, c->nd[NX(d)], 1, s->x[d]));
            if (s->v[d]) CHK(get_component<double>(c, c->nd[NV(d)], 1, s->v[d]));
            if (s->V[d] && c->xsph) CHK(get_component<double>(c, c->nd[NVV(d)], 1, s->V[d]));
            if (s->f[d]) CHK(get_component<double>(c, c->nd[NF(d)], 1, s->f[d]));
        }
        if (s->rho) CHK(get_component<float>(c, c->nfl[F_RHO], 1, s->rho));
        if (s->p) CHK(get_component<float>(c, c->nfl[F_P], 1, s->p));
        if (s->Rrho) CHK(get_component<float>(c, c->nfl[F_RRHO], 1, s->Rrho));
        if (s->m) CHK(get_component<float>(c, c->nfl[F_M], 1, s->m));
    } else {
        const Rec* r = c->rec[c->b0];
        for (int d = 0; d < dim; ++d) {
            if (s->x[d]) CHK(get_component<double>(c, &r->xv[d][0], 8, s->x[d]));
            if (s->v[d]) CHK(get_component<double>(c, &r->xv[d][1], 8, s->v[d]));
            if (s->f[d]) CHK(get_component<double>(c, c->pd[d], 1, s->f[d]));
            if (s->V[d] && c->xsph) CHK(get_component<double>(c, c->pd[3 + d], 1, s->V[d]));
        }
        if (s->rho) CHK(get_component<float>(c, &r->rho, 16, s->rho));
        if (s->p) CHK(get_component<float>(c, c->pp, 1, s->p));
        if (s->m) CHK(get_component<float>(c, &r->m, 16, s->m));
        if (s->Rrho) CHK(get_component<float>(c, c->pRrho, 1, s->Rrho));
    }
    return SPH_OK;
}

int sph_forces(sph_ctx* c, int32_t which) {
    if (c && c->lost) return fail(c, SPH_ESTATE, "the context is lost (an earlier call reported a launch that never came back)");
    if (!c) return SPH_EINVAL;
    if (which != SPH_STATE_N && which != SPH_STATE_NP1) return fail(c, SPH_EINVAL, "bad state selector");
    if (which == SPH_STATE_N ? !c->uploaded_n : !c->uploaded_np1) return fail(c, SPH_ESTATE, "state was never uploaded");
    HIPCHK(hipSetDevice(c->device));
    CHK(settle_np1_outputs(c));
    int src = c->b0;
    if (which == SPH_STATE_N) {
        src = (c->b0 + 1) % NREC;
        CHK(pack_n_into(c, c->rec[src]));
    }
    CHK(rebuild(c, src));
    /* lists built for state n's positions do not serve state n+1 */
    if (which == SPH_STATE_N) c->list_valid = false;
    hipLaunchKernelGGL(k_step_begin, dim3(1), dim3(1), 0, c->stream, c->S);
    const uint32_t part_n = pairgrid(c->n);
    CHK(launch_pair<false>(c, c->rec[src], nullptr, out_state(c, which), 1, 0, 0, c->n, 0, part_n));
    hipLaunchKernelGGL(k_fold_parts, dim3(PART_GROUPS, 1), dim3(SPH_BLOCK), 0, c->stream, c->part_use, part_n, 1, c->part2_use);
    hipLaunchKernelGGL(k_forces_finalize, dim3(1), dim3(1), 0, c->stream, c->S, c->part2_use);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->dist) {
        CHK(read_scalars(c));
        double v[1] = {c->S_host->maxmu};
        CHK(allreduce_host(c, v, 1, 0));
        HIPCHK(hipMemcpyAsync(&c->S->maxmu, v, sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return SPH_OK;
}

int sph_zero_boundary_forces(sph_ctx* c, int32_t which) {
    if (c && c->lost) return fail(c, SPH_ESTATE, "the context is lost (an earlier call reported a launch that never came back)");
    if (!c) return SPH_EINVAL;
    if (which != SPH_STATE_N && which != SPH_STATE_NP1) return fail(c, SPH_EINVAL, "bad state selector");
    HIPCHK(hipSetDevice(c->device));
    CHK(settle_np1_outputs(c));
    OutState o = out_state(c, which);
    hipLaunchKernelGGL(k_zero_wall_f, dim3(nblk(c->n)), dim3(SPH_BLOCK), 0, c->stream, o.f[0], o.f[1], o.f[2], c->dim,
                       wall_ids(c), c->n, c->dp.bound_parts);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return SPH_OK;
}

int sph_step(sph_ctx* c, int32_t nsteps, sph_scalars* out) {
    if (c && c->lost) return fail(c, SPH_ESTATE, "the context is lost (an earlier call reported a launch that never came back)");
    if (!c || nsteps < 0) return SPH_EINVAL;
    if (!c->uploaded_n || !c->uploaded_np1) return fail(c, SPH_ESTATE, "both states must be uploaded before stepping");
    HIPCHK(hipSetDevice(c->device));
    for (int s = 0; s < nsteps; ++s) CHK(one_step(c, 0, false));
    if (nsteps == 0) CHK(read_scalars(c));
    fill_scalars(c, out);
    return SPH_OK;
}

int sph_step_noalloc(sph_ctx* c, int32_t nsteps, sph_scalars* out) {
    if (c && c->lost) return fail(c, SPH_ESTATE, "the context is lost (an earlier call reported a launch that never came back)");
    if (!c || nsteps < 0) return SPH_EINVAL;
    if (!c->uploaded_n || !c->uploaded_np1) return fail(c, SPH_ESTATE, "both states must be uploaded before stepping");
    if (c->dist) return fail(c, SPH_EINVAL, "sph_step_noalloc serves single-domain contexts only");
    if (c->cap > SPH_NOALLOC_MAX_PARTICLES || (size_t)c->npairblocks > NOALLOC_PART_SLOTS)
        return fail(c, SPH_EINVAL, "sph_step_noalloc: more particles than the fixed scratch array serves (SPH_NOALLOC_MAX_PARTICLES)");
    HIPCHK(hipSetDevice(c->device));
    /* the pair kernel's partials and their fold go to this device's scratch array; each sweep
     * overwrites the slots it reads back, so nothing has to be cleared between calls */
    c->part_use = c->scratch;
    c->part2_use = c->scratch + (size_t)PART_FIELDS * NOALLOC_PART_SLOTS;
    c->no_alloc = true;
    int rc = SPH_OK;
    for (int s = 0; s < nsteps && rc == SPH_OK; ++s) rc = one_step(c, 0, false);
    if (rc == SPH_OK && nsteps == 0) rc = read_scalars(c);
    c->no_alloc = false;
    c->part_use = c->part;
    c->part2_use = c->part2;
    if (rc != SPH_OK) return rc;
    fill_scalars(c, out);
    return SPH_OK;
}

int sph_sweeps(sph_ctx* c, int32_t nsweeps, sph_scalars* out) {
    if (c && c->lost) return fail(c, SPH_ESTATE, "the context is lost (an earlier call reported a launch that never came back)");
    if (!c || nsweeps < 1 || nsweeps > c->prm.subits) return SPH_EINVAL;
    if (!c->uploaded_n || !c->uploaded_np1) return fail(c, SPH_ESTATE, "both states must be uploaded before stepping");
    HIPCHK(hipSetDevice(c->device));
    CHK(one_step(c, nsweeps, true));
    fill_scalars(c, out);
    return SPH_OK;
}

int sph_get_scalars(sph_ctx* c, sph_scalars* out) {
    if (c && c->lost) return fail(c, SPH_ESTATE, "the context is lost (an earlier call reported a launch that never came back)");
    if (!c || !out) return SPH_EINVAL;
    HIPCHK(hipSetDevice(c->device));
    CHK(read_scalars(c));
    fill_scalars(c, out);
    return SPH_OK;
}

int sph_set_dt(sph_ctx* c, double dt) {
    if (c && c->lost) return fail(c, SPH_ESTATE, "the context is lost (an earlier call reported a launch that never came back)");
    if (!c || !(dt > 0)) return SPH_EINVAL;
    HIPCHK(hipSetDevice(c->device));
    const double v[2] = {dt, dt};
    HIPCHK(hipMemcpyAsync(&c->S->dt, v, sizeof v, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SPH_OK;
}

namespace {

/* DensityReinit's lists without a list rebuild.  While the stepping path's lists stand, the
 * particles are still in the order and in the cell table of the last rebuild, and none has
 * travelled further than the bound kept for those lists.  A particle's neighbours within r now were
 * therefore within r + travel of where it is now when the cells were filled: the build walks the
 * cell ranges chosen for that distance and tests the candidates on their positions of now (fresh
 * lattice entries), wall-wall pairs included.  The rows go where the step lists live (those are
 * drawn afresh at the start of the next step, which costs a draw instead of the rebuild that a
 * re-sorted order would force on it); the candidate lists, the order and the travel bound of the
 * stepping path are left as they are.  *done stays false when this cannot be had (no standing
 * lists, travel beyond what one cell's width covers, rows too short). */
int reinit_lists_on_standing_order(sph_ctx* c, bool* done) {
    *done = false;
    if (c->ri_fresh_lists || !c->list_valid || !c->lists_for_step) {
        if (c->trace) fprintf(stderr, "[sph] reinit: no lists on the standing order: asked for fresh ones %d, lists stand %d, made for stepping %d\n",
                              (int)c->ri_fresh_lists, (int)c->list_valid, (int)c->lists_for_step);
        return SPH_OK;
    }
    const double R = std::sqrt(c->prm.search_radius_sq);
    double skin = 0.01;
    if (const char* e = getenv("SPH_ONESHOT_SKIN")) skin = std::max(0.0, atof(e));
    const double r = R * (1.0 + skin);
    const double range = r + c->disp_host * (1.0 + 1e-9) + 1e-12 * R;
    /* the neighbouring columns are all the build looks at: the range must not exceed a cell's edge */
    if (!(c->disp_host >= 0.0) || !(range * (1.0 + 1e-6) <= 1.0 / c->grid.inv[0])) {
        if (c->trace) fprintf(stderr, "[sph] reinit: no lists on the standing order: range %.4g, cell edge %.4g\n", range, 1.0 / c->grid.inv[0]);
        return SPH_OK;
    }
    /* the rows go where the step lists live; a context made without step lists has no such rows,
     * and none are made for this */
    if (c->inner_mode < 0 && !c->nbr_in) return SPH_OK;
    bool ok = true;
    CHK(ensure_step_lists(c, &ok));
    const Rec* rec = c->rec[c->b0];
    const uint32_t m = c->n + c->n_halo;
    const QLattice& L = c->lattice;
    hipLaunchKernelGGL(k_fill_qpos, dim3(nblk(m)), dim3(SPH_BLOCK), 0, c->stream, rec, m, L, c->dim, nullptr, 0u, qpos_of(c));
    const double rq = r * L.invq + 2.0;
    const uint32_t r2q = (uint32_t)std::ceil(rq * rq);
    uint32_t need = 0;
    CHK(run_list_build(c, rec, range * range, r2q, c->nbr_in, c->cnt_in, 0, 0u, nullptr, nullptr, &need, 1));
    c->step_lists_valid = false; /* their rows have just been used for something else */
    if (need > c->cap_k) {
        if (c->trace) fprintf(stderr, "[sph] reinit: no lists on the standing order: a row of %u entries, rows hold %u\n", need, c->cap_k);
        return SPH_OK;
    }
    CHK(check_rows(c, c->nbr_in, c->cnt_in, nullptr, nullptr));
    c->ri_nbr = c->nbr_in;
    c->ri_cnt = c->cnt_in;
    ++c->ri_quick_lists;
    *done = true;
    return SPH_OK;
}

/* One launch of the one-launch walk (or one round of it, in a slab run) and the wait for it.  The
 * waves give up by themselves after the context's time limit; should the launch outlive that by
 * far, the host raises the stop flag from a second stream, and if even that does not end it the
 * context is lost.  *stop_word: the stop flag as the launch left it (0, REINIT_STOP_GAVE_UP,
 * REINIT_STOP_DRY). */
int reinit_flow_launch(sph_ctx* c, Rec* r, const uint32_t* ids, uint32_t owners, uint32_t qseg, unsigned long long* gran,
                       int partial, uint32_t* stop_word) {
    const uint64_t limit = (uint64_t)(c->ri_limit_s * 1e8); /* ticks of the 100 MHz clock */
    uint32_t* const h = (uint32_t*)c->need_host; /* pinned */
    if (c->dim == 3)
        hipLaunchKernelGGL(k_reinit_flow<3>, dim3(owners), dim3(REINIT_FLOW_THREADS), 0, c->stream, c->dp, r, c->ri_nbr, c->cap_k,
                           c->ri_cnt, ids, c->ri_pend, c->ri_queue, c->ri_flow, qseg, limit, gran, c->ri_gen, partial);
    else
        hipLaunchKernelGGL(k_reinit_flow<2>, dim3(owners), dim3(REINIT_FLOW_THREADS), 0, c->stream, c->dp, r, c->ri_nbr, c->cap_k,
                           c->ri_cnt, ids, c->ri_pend, c->ri_queue, c->ri_flow, qseg, limit, gran, c->ri_gen, partial);
    HIPCHK(hipGetLastError());
    ++c->reinit_launches;
    HIPCHK(hipMemcpyAsync(h, c->ri_flow, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    const auto t_start = std::chrono::steady_clock::now();
    bool raised = false;
    for (;;) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) HIPCHK(q);
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        if (!raised && waited > 2.0 * c->ri_limit_s + 5.0) {
            hipStream_t side;
            HIPCHK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
            static const uint32_t one = REINIT_STOP_GAVE_UP;
            HIPCHK(hipMemcpyAsync(c->ri_flow, &one, sizeof one, hipMemcpyHostToDevice, side));
            HIPCHK(hipStreamSynchronize(side));
            HIPCHK(hipStreamDestroy(side));
            raised = true;
            fprintf(stderr, "[sph] density re-initialisation: stop flag raised by the host after %.1f s\n", waited);
        }
        if (waited > 4.0 * c->ri_limit_s + 20.0) {
            c->lost = true;
            c->ri_owners = 0;
            return fail(c, SPH_ESTATE, "density re-initialisation does not come back; the context is lost "
                                       "(its stream still holds the launch: end the process with a failure code)");
        }
        if (waited > 0.002) std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    *stop_word = *h;
    return SPH_OK;
}

/* The walk of DensityReinit over lists that are already built; `one_launch` picks the one-launch
 * walk (a single rank's large case).  *stalled: that walk gave up with all its waves gone. */
int reinit_walk(sph_ctx* c, bool one_launch, bool* stalled) {
    *stalled = false;
    const uint32_t n = c->n;
    const uint32_t qcap = c->cap;
    Rec* r = c->rec[c->b0];
    const uint32_t* ids = wall_ids(c); /* the walk's order: original index, global in a slab run */
    HIPCHK(hipMemsetAsync(c->ri_ctl, 0, sizeof(ReinitCtl), c->stream));
    uint32_t level = 0; /* the next level to launch */
#define RI_LEVEL()                                                                                                        \
    do {                                                                                                                  \
        if (c->dim == 3) hipLaunchKernelGGL(k_reinit_level<3>, dim3(grid), dim3(REINIT_LEVEL_THREADS), 0, c->stream, c->dp, r, \
                                            c->ri_nbr, c->cap_k, c->ri_cnt, ids, c->ri_pend, c->ri_queue, c->ri_ctl, qcap, level); \
        else hipLaunchKernelGGL(k_reinit_level<2>, dim3(grid), dim3(REINIT_LEVEL_THREADS), 0, c->stream, c->dp, r, c->ri_nbr,    \
                                c->cap_k, c->ri_cnt, ids, c->ri_pend, c->ri_queue, c->ri_ctl, qcap, level);                     \
        ++level;                                                                                                          \
    } while (0)
    uint32_t* touches = c->dist ? c->flags : nullptr;
    /* a single rank's large case is walked in one launch: every workgroup a queue of its own */
    const uint32_t owners = one_launch ? c->ri_owners : 0u;
    const uint32_t qseg = owners ? (n + owners - 1) / owners + 1u : 0u;
    if (owners) {
        if ((size_t)owners * qseg + 2 + 2 * ((size_t)n + c->n_halo) > (size_t)3 * c->cap + 1024)
            return fail(c, SPH_ESTATE, "density re-initialisation: queues do not fit");
        ++c->ri_gen;
        /* the published words sit behind the queues; when the particle count has changed (a slab
         * run's migration) they lie where queue entries were, which could pass for tags */
        const size_t gran_off = ((size_t)owners * qseg + 1) & ~(size_t)1;
        if (gran_off != c->ri_gran_off) {
            HIPCHK(hipMemsetAsync(c->ri_queue + gran_off, 0, sizeof(uint32_t) * ((size_t)3 * c->cap + 1024 - gran_off), c->stream));
            c->ri_gran_off = gran_off;
        }
        HIPCHK(hipMemsetAsync(c->ri_queue, 0xFF, sizeof(uint32_t) * (size_t)owners * qseg, c->stream));
        HIPCHK(hipMemsetAsync(c->ri_flow, 0, sizeof(uint32_t) * REINIT_FLOW_WORDS(owners), c->stream));
    }
    if (c->dim == 3)
        hipLaunchKernelGGL(k_reinit_count<3>, dim3((n + SPH_BLOCK / 4 - 1) / (SPH_BLOCK / 4)), dim3(SPH_BLOCK), 0, c->stream, c->dp, r, c->ri_nbr, c->cap_k, c->ri_cnt,
                           ids, c->ri_pend, c->ri_queue, c->ri_ctl, touches, owners, qseg, c->ri_flow);
    else
        hipLaunchKernelGGL(k_reinit_count<2>, dim3((n + SPH_BLOCK / 4 - 1) / (SPH_BLOCK / 4)), dim3(SPH_BLOCK), 0, c->stream, c->dp, r, c->ri_nbr, c->cap_k, c->ri_cnt,
                           ids, c->ri_pend, c->ri_queue, c->ri_ctl, touches, owners, qseg, c->ri_flow);
    HIPCHK(hipGetLastError());
    ReinitCtl* h = (ReinitCtl*)c->need_host; /* pinned, 8 words */
    c->reinit_walked = c->dist ? (owners ? SPH_REINIT_ONE_LAUNCH_ROUNDS : SPH_REINIT_LEVELS)
                               : (n <= REINIT_SINGLE_MAX ? SPH_REINIT_ONE_WORKGROUP : owners ? SPH_REINIT_ONE_LAUNCH : SPH_REINIT_LEVELS);
    if (c->dist && owners) {
        /* Slab run, one-launch walk in rounds.  The walk is in global index order, so a particle
         * may wait for one of a neighbouring rank: it is then not released.  A round: every rank
         * walks until its walk has run dry (k_reinit_flow, partial); the ranks tell their
         * neighbours the new rho of the shared particles that are finished; each publishes what it
         * is told, recounts the waiting particles next to those and queues the ones set free.
         * Same values as the walk over the whole domain. */
        unsigned long long* const gran = (unsigned long long*)(c->ri_queue + (((size_t)owners * qseg + 1) & ~(size_t)1));
        const size_t hc = c->halo_cap;
        std::vector<uint32_t> words(REINIT_FLOW_WORDS(owners));
        auto read_counts = [&](uint64_t* pushed, uint64_t* finished) -> int {
            HIPCHK(hipMemcpyAsync(words.data(), c->ri_flow, sizeof(uint32_t) * words.size(), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            *pushed = *finished = 0;
            for (uint32_t o = 0; o < owners; ++o) {
                *pushed += words[REINIT_FLOW_TAIL(o)];
                *finished += words[REINIT_FLOW_DONE(o)];
            }
            return SPH_OK;
        };
        uint64_t done_before = 0;
        for (uint64_t round = 0;; ++round) {
            HIPCHK(hipMemsetAsync(c->ri_flow, 0, sizeof(uint32_t), c->stream)); /* the stop flag */
            uint32_t stop_word = 0;
            if (c->ri_take_turns) {
                /* ranks that share a GPU: two walks that each hold part of the chip would wait for
                 * workgroups the other one keeps out; one rank at a time */
                for (int turn = 0; turn < c->nranks; ++turn) {
                    if (turn == c->rank) CHK(reinit_flow_launch(c, r, ids, owners, qseg, gran, 1, &stop_word));
                    double token[1] = {0.0};
                    CHK(allreduce_host(c, token, 1, 1));
                }
            } else {
                CHK(reinit_flow_launch(c, r, ids, owners, qseg, gran, 1, &stop_word));
            }
            /* a rank whose walk gave up still takes part in the round's exchanges, then all stop */
            uint64_t pushed = 0, done_now = 0;
            CHK(read_counts(&pushed, &done_now));
            uint2* const sb = (uint2*)c->sendbuf;
            uint2* const rb = (uint2*)c->recvbuf;
            for (int side = 0; side < 2; ++side)
                if (c->nsend[side])
                    hipLaunchKernelGGL(k_reinit_pack_flow, dim3(nblk(c->nsend[side])), dim3(SPH_BLOCK), 0, c->stream, gran, c->ri_gen,
                                       c->send_idx + side * hc, c->nsend[side], sb + side * hc);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(c->ev_packed, c->stream));
            HIPCHK(hipStreamWaitEvent(c->comm, c->ev_packed, 0));
            CHK(exchange_bytes(c, sb, (size_t)c->nsend[0] * sizeof(uint2), rb, (size_t)c->nrecv[0] * sizeof(uint2), sb + hc,
                               (size_t)c->nsend[1] * sizeof(uint2), rb + hc, (size_t)c->nrecv[1] * sizeof(uint2)));
            HIPCHK(hipEventRecord(c->ev_recvd, c->comm));
            HIPCHK(hipStreamWaitEvent(c->stream, c->ev_recvd, 0));
            if (c->nrecv[0])
                hipLaunchKernelGGL(k_reinit_unpack_flow, dim3(nblk(c->nrecv[0])), dim3(SPH_BLOCK), 0, c->stream, rb, c->halo_pos,
                                   c->nrecv[0], r, gran, c->ri_gen);
            if (c->nrecv[1])
                hipLaunchKernelGGL(k_reinit_unpack_flow, dim3(nblk(c->nrecv[1])), dim3(SPH_BLOCK), 0, c->stream, rb + hc,
                                   c->halo_pos + c->nrecv[0], c->nrecv[1], r, gran, c->ri_gen);
            if (c->dim == 3)
                hipLaunchKernelGGL(k_reinit_rescan_flow<3>, dim3(nblk(n)), dim3(SPH_BLOCK), 0, c->stream, c->dp, r, c->ri_nbr, c->cap_k,
                                   c->ri_cnt, ids, c->ri_pend, c->ri_queue, c->ri_flow, touches, gran, c->ri_gen, owners, qseg);
            else
                hipLaunchKernelGGL(k_reinit_rescan_flow<2>, dim3(nblk(n)), dim3(SPH_BLOCK), 0, c->stream, c->dp, r, c->ri_nbr, c->cap_k,
                                   c->ri_cnt, ids, c->ri_pend, c->ri_queue, c->ri_flow, touches, gran, c->ri_gen, owners, qseg);
            HIPCHK(hipGetLastError());
            uint64_t pushed_after = 0, done_same = 0;
            CHK(read_counts(&pushed_after, &done_same));
            /* all ranks: particles not finished yet, whether this round moved anything, and whether
             * any rank's walk gave up */
            double v[3] = {(double)(n - done_now), (double)(done_now - done_before) + (double)(pushed_after - done_now),
                           stop_word == REINIT_STOP_GAVE_UP ? 1.0 : 0.0};
            CHK(allreduce_host(c, v, 3, 1));
            c->reinit_rounds = round + 1;
            if (v[2] != 0.0) {
                *stalled = true;
                return fail(c, SPH_ESTATE, "density re-initialisation: a rank's one-launch walk gave up");
            }
            if (v[0] == 0.0) break;
            if (v[1] == 0.0) return fail(c, SPH_ESTATE, "density re-initialisation stalled before every particle was done");
            done_before = done_now;
        }
    } else if (c->dist) {
        /* The walk is in global index order, so a particle may wait for one of a neighbouring
         * rank.  Rounds: every rank walks as far as its own particles allow; the ranks then tell
         * their neighbours the new rho of the shared particles that are finished, each recounts
         * the waiting particles next to those, and the walk goes on from the ones set free.
         * Same values as the walk over the whole domain. */
        const uint32_t nh = c->nrecv[0] + c->nrecv[1];
        const size_t hc = c->halo_cap;
        if (nh) HIPCHK(hipMemsetAsync(c->ri_pend + n, 0, sizeof(uint32_t) * nh, c->stream));
        const int batch = 16, grid = 256;
        uint32_t done_before = 0;
        for (uint64_t round = 0;; ++round) {
            for (;;) { /* the local walk, until its frontier is empty */
                for (int b = 0; b < batch; ++b) RI_LEVEL();
                c->reinit_launches += batch;
                HIPCHK(hipGetLastError());
                HIPCHK(hipMemcpyAsync(h, c->ri_ctl, sizeof(ReinitCtl), hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                if (h->cnt[level % 3u] == 0) break;
            }
            const uint32_t done_now = h->done;
            /* finished shared particles go to the neighbours */
            uint2* const sb = (uint2*)c->sendbuf;
            uint2* const rb = (uint2*)c->recvbuf;
            for (int side = 0; side < 2; ++side)
                if (c->nsend[side])
                    hipLaunchKernelGGL(k_reinit_pack, dim3(nblk(c->nsend[side])), dim3(SPH_BLOCK), 0, c->stream, r, c->ri_pend,
                                       c->send_idx + side * hc, c->nsend[side], sb + side * hc);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(c->ev_packed, c->stream));
            HIPCHK(hipStreamWaitEvent(c->comm, c->ev_packed, 0));
            CHK(exchange_bytes(c, sb, (size_t)c->nsend[0] * sizeof(uint2), rb, (size_t)c->nrecv[0] * sizeof(uint2), sb + hc,
                               (size_t)c->nsend[1] * sizeof(uint2), rb + hc, (size_t)c->nrecv[1] * sizeof(uint2)));
            HIPCHK(hipEventRecord(c->ev_recvd, c->comm));
            HIPCHK(hipStreamWaitEvent(c->stream, c->ev_recvd, 0));
            if (c->nrecv[0])
                hipLaunchKernelGGL(k_reinit_unpack, dim3(nblk(c->nrecv[0])), dim3(SPH_BLOCK), 0, c->stream, rb, c->halo_pos,
                                   c->nrecv[0], r, c->ri_pend);
            if (c->nrecv[1])
                hipLaunchKernelGGL(k_reinit_unpack, dim3(nblk(c->nrecv[1])), dim3(SPH_BLOCK), 0, c->stream, rb + hc,
                                   c->halo_pos + c->nrecv[0], c->nrecv[1], r, c->ri_pend);
            /* those set free by the neighbours' news join the frontier of the next level */
            if (c->dim == 3)
                hipLaunchKernelGGL(k_reinit_rescan<3>, dim3(nblk(n)), dim3(SPH_BLOCK), 0, c->stream, c->dp, r, c->ri_nbr, c->cap_k,
                                   c->ri_cnt, ids, c->ri_pend, c->ri_queue, c->ri_ctl, touches, qcap, level);
            else
                hipLaunchKernelGGL(k_reinit_rescan<2>, dim3(nblk(n)), dim3(SPH_BLOCK), 0, c->stream, c->dp, r, c->ri_nbr, c->cap_k,
                                   c->ri_cnt, ids, c->ri_pend, c->ri_queue, c->ri_ctl, touches, qcap, level);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(h, c->ri_ctl, sizeof(ReinitCtl), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            /* all ranks: particles not finished yet, and whether this round moved anything */
            double v[2] = {(double)(n - done_now), (double)(done_now - done_before) + (double)h->cnt[level % 3u]};
            CHK(allreduce_host(c, v, 2, 1));
            c->reinit_rounds = round + 1;
            if (v[0] == 0.0) break;
            if (v[1] == 0.0) return fail(c, SPH_ESTATE, "density re-initialisation stalled before every particle was done");
            done_before = done_now;
        }
    } else if (n <= REINIT_SINGLE_MAX) { /* a small case is walked by one workgroup, level after level */
        ++c->reinit_launches;
        if (c->dim == 3)
            hipLaunchKernelGGL(k_reinit_single<3>, dim3(1), dim3(REINIT_SINGLE_THREADS), 0, c->stream, c->dp, r, c->ri_nbr, c->cap_k,
                               c->ri_cnt, ids, c->ri_pend, c->ri_queue, c->ri_ctl);
        else
            hipLaunchKernelGGL(k_reinit_single<2>, dim3(1), dim3(REINIT_SINGLE_THREADS), 0, c->stream, c->dp, r, c->ri_nbr, c->cap_k,
                               c->ri_cnt, ids, c->ri_pend, c->ri_queue, c->ri_ctl);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h, c->ri_ctl, sizeof(ReinitCtl), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (h->done != n || h->cnt[0] != n)
            return fail(c, SPH_ESTATE, "density re-initialisation stalled before every particle was done");
    } else if (owners) {
        unsigned long long* const gran = (unsigned long long*)(c->ri_queue + (((size_t)owners * qseg + 1) & ~(size_t)1));
        if (c->trace) fprintf(stderr, "[sph] reinit: one-launch walk, %u workgroups, %u places each, call %u\n", owners, qseg, c->ri_gen);
        uint32_t stop_word = 0;
        CHK(reinit_flow_launch(c, r, ids, owners, qseg, gran, 0, &stop_word));
        if (stop_word != 0) {
            *stalled = true; /* every wave has left: the caller may walk again, by levels */
            return fail(c, SPH_ESTATE, "density re-initialisation stalled before every particle was done");
        }
    } else {
        /* one launch per dependency level; the host looks at the queues every so often */
        const int batch = 256, grid = 256; /* one wavefront per released particle, 4096 per pass of the grid */
        for (uint64_t guard = 0;; ++guard) {
            for (int b = 0; b < batch; ++b) RI_LEVEL();
            c->reinit_launches += batch;
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(h, c->ri_ctl, sizeof(ReinitCtl), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            if (h->cnt[level % 3u] == 0) {
                if (h->done != n) return fail(c, SPH_ESTATE, "density re-initialisation stalled before every particle was done");
                break;
            }
            if (guard > (uint64_t)n / batch + 4) return fail(c, SPH_ESTATE, "density re-initialisation did not finish");
        }
    }
#undef RI_LEVEL
    return SPH_OK;
}

}  // namespace

int sph_density_reinit(sph_ctx* c) {
    if (c && c->lost) return fail(c, SPH_ESTATE, "the context is lost (an earlier call reported a launch that never came back)");
    if (!c) return SPH_EINVAL;
    if (!c->uploaded_np1) return fail(c, SPH_ESTATE, "state was never uploaded");
    HIPCHK(hipSetDevice(c->device));
    CHK(settle_np1_outputs(c)); /* p of state n+1 is read below */
    if (!c->ri_pend) {
        CHK(dalloc(c, &c->ri_pend, c->cap));
        /* three frontier queues used in turn (walk by levels); the one-launch walk keeps its
         * queues (n + a few places) and behind them 8 bytes per particle of published densities */
        CHK(dalloc(c, &c->ri_queue, (size_t)3 * c->cap + 1024));
        HIPCHK(hipMemsetAsync(c->ri_queue, 0, sizeof(uint32_t) * ((size_t)3 * c->cap + 1024), c->stream));
        CHK(dalloc(c, &c->ri_ctl, 1));
        /* the one-launch walk needs all its workgroups on the chip together */
        int per_cu = 0;
        const void* fn = c->dim == 3 ? (const void*)k_reinit_flow<3> : (const void*)k_reinit_flow<2>;
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, REINIT_FLOW_THREADS, 0));
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, c->device));
        c->ri_owners = per_cu > 0 ? (uint32_t)std::min(256, prop.multiProcessorCount) : 0u;
        if (const char* e = getenv("SPH_REINIT_WALK")) {
            if (!strcmp(e, "levels")) c->ri_owners = 0;
        }
        /* fewer workgroups (tests: several ranks' walks on one GPU at the same time) */
        if (const char* e = getenv("SPH_REINIT_OWNERS"))
            if (c->ri_owners) c->ri_owners = (uint32_t)std::min<long>(c->ri_owners, std::max<long>(1, atol(e)));
        if (const char* e = getenv("SPH_REINIT_LIMIT_S")) c->ri_limit_s = std::max(1e-6, atof(e));
        if (c->ri_owners) CHK(dalloc(c, &c->ri_flow, REINIT_FLOW_WORDS(c->ri_owners)));
        if (c->dist) {
            /* Ranks that share a GPU (test set-ups) cannot all have their walk's workgroups on the
             * chip together: they then walk one after the other.  Every rank tells the others which
             * GPU of which machine it runs on. */
            char host[256] = {0};
            gethostname(host, sizeof host - 1);
            uint32_t hh = 2166136261u;
            for (const char* p = host; *p; ++p) hh = (hh ^ (uint8_t)*p) * 16777619u;
            std::vector<double> where(2 * (size_t)c->nranks, 0.0);
            where[2 * c->rank] = (double)hh;
            where[2 * c->rank + 1] = (double)(((uint32_t)prop.pciDomainID << 16) | ((uint32_t)prop.pciBusID << 8) | (uint32_t)prop.pciDeviceID) + 1.0;
            CHK(allreduce_host(c, where.data(), 2 * c->nranks, 1));
            for (int k = 0; k < c->nranks; ++k)
                if (k != c->rank && where[2 * k] == where[2 * c->rank] && where[2 * k + 1] == where[2 * c->rank + 1])
                    c->ri_take_turns = true;
            if (const char* e = getenv("SPH_REINIT_TURNS")) c->ri_take_turns = atoi(e) != 0;
            /* all ranks walk the same way */
            double agree[2] = {c->ri_owners ? 0.0 : 1.0, c->ri_take_turns ? 1.0 : 0.0};
            CHK(allreduce_host(c, agree, 2, 0));
            if (agree[0] != 0.0) c->ri_owners = 0;
            c->ri_take_turns = agree[1] != 0.0;
            if (c->trace) fprintf(stderr, "[sph] reinit: rank %d of %d, %s%s\n", c->rank, c->nranks,
                                  c->ri_owners ? "one-launch walk in rounds" : "walk by levels", c->ri_take_turns ? ", one rank at a time" : "");
        }
    }
    /* complete lists (wall-wall pairs included) on the positions of state n+1: on the standing
     * order when the stepping path's lists stand, else by a rebuild (a slab run fetches its
     * neighbours' particles with it) */
    bool quick = false;
    CHK(reinit_lists_on_standing_order(c, &quick));
    if (c->dist) { /* all ranks walk the same lists */
        double v[1] = {quick ? 0.0 : 1.0};
        CHK(allreduce_host(c, v, 1, 0));
        quick = v[0] == 0.0;
    }
    if (!quick) {
        CHK(rebuild(c, c->b0));
        c->ri_nbr = c->nbr;
        c->ri_cnt = c->cnt;
    }
    const uint32_t n = c->n;
    Rec* r = c->rec[c->b0];
    bool stalled = false;
    c->reinit_launches = c->reinit_rounds = 0;
    if (c->ri_owners && (c->dist || n > REINIT_SINGLE_MAX)) {
        /* The one-launch walk needs every one of its workgroups on the chip at once.  Should it
         * give up (compute units taken by somebody else's work), rho goes back to what it was
         * and this context walks by levels from now on.  In a slab run every rank hears of one
         * rank's giving up in the same round, and all of them change over together. */
        float* const keep = (float*)c->tmp;
        const uint32_t held = n + c->n_halo; /* the neighbours' particles take the ranks' news in place */
        hipLaunchKernelGGL(k_rho_copy, dim3(nblk(held)), dim3(SPH_BLOCK), 0, c->stream, r, keep, held, 0);
        const int rc = reinit_walk(c, true, &stalled);
        if (rc != SPH_OK && !stalled) return rc;
        if (stalled) {
            fprintf(stderr, "[sph] density re-initialisation: the one-launch walk gave up; walking by levels\n");
            hipLaunchKernelGGL(k_rho_copy, dim3(nblk(held)), dim3(SPH_BLOCK), 0, c->stream, r, keep, held, 1);
            c->ri_owners = 0;
            CHK(reinit_walk(c, false, &stalled));
        }
    } else {
        CHK(reinit_walk(c, false, &stalled));
    }
    /* p/rho^2 of the records follows the new rho (Forces forms it from p and rho, WCXSPH3D.cpp:180) */
    hipLaunchKernelGGL(k_fix_prr, dim3(nblk(n)), dim3(SPH_BLOCK), 0, c->stream, r, c->pp, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return SPH_OK;
}

int sph_reinit_info(sph_ctx* c, uint64_t* out4) {
    if (!c || !out4) return SPH_EINVAL;
    out4[0] = c->reinit_walked;
    out4[1] = c->reinit_launches;
    out4[2] = c->reinit_rounds;
    out4[3] = (c->ri_take_turns ? 1u : 0u) | (c->ri_nbr != nullptr && c->ri_nbr == c->nbr_in ? 2u : 0u);
    return SPH_OK;
}

int sph_neighbours(sph_ctx* c, int32_t which, uint64_t* row_offsets, uint32_t* cols) {
    if (c && c->lost) return fail(c, SPH_ESTATE, "the context is lost (an earlier call reported a launch that never came back)");
    if (!c || !row_offsets) return SPH_EINVAL;
    if (which != SPH_STATE_N && which != SPH_STATE_NP1) return fail(c, SPH_EINVAL, "bad state selector");
    if (which == SPH_STATE_N ? !c->uploaded_n : !c->uploaded_np1) return fail(c, SPH_ESTATE, "state was never uploaded");
    HIPCHK(hipSetDevice(c->device));
    int src = c->b0;
    if (which == SPH_STATE_N) {
        src = (c->b0 + 1) % NREC;
        CHK(pack_n_into(c, c->rec[src]));
    }
    /* SPH_NEIGHBOURS_LISTS=standing (tests): state n+1's sets from lists made on the stepping path's
     * standing order, the way DensityReinit makes its own, when that can be had */
    const uint32_t* rows = nullptr;
    const uint32_t* rcnt = nullptr;
    if (which == SPH_STATE_NP1 && c->probe_standing) {
        bool standing = false;
        CHK(reinit_lists_on_standing_order(c, &standing));
        if (c->dist) {
            double v[1] = {standing ? 0.0 : 1.0};
            CHK(allreduce_host(c, v, 1, 0));
            standing = v[0] == 0.0;
        }
        if (standing) {
            rows = c->nbr_in;
            rcnt = c->cnt_in;
        }
    }
    if (!rows) {
        CHK(rebuild(c, src));
        if (which == SPH_STATE_N) c->list_valid = false;
        rows = c->nbr;
        rcnt = c->cnt;
    }
    const uint32_t n = c->n;
    uint32_t* counts = (uint32_t*)c->tmp;
    if (c->dim == 3)
        hipLaunchKernelGGL(k_nbr_count<3>, dim3(nblk(n)), dim3(SPH_BLOCK), 0, c->stream, c->dp, c->rec[src], rows, c->cap_k,
                           rcnt, c->id, counts);
    else
        hipLaunchKernelGGL(k_nbr_count<2>, dim3(nblk(n)), dim3(SPH_BLOCK), 0, c->stream, c->dp, c->rec[src], rows, c->cap_k,
                           rcnt, c->id, counts);
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> hc(n);
    HIPCHK(hipMemcpyAsync(hc.data(), counts, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    uint64_t off = 0;
    for (uint32_t i = 0; i < n; ++i) {
        row_offsets[i] = off;
        off += hc[i];
    }
    row_offsets[n] = off;
    if (!cols) return SPH_OK;
    uint64_t* doffs = nullptr;
    uint32_t* dcols = nullptr;
    HIPCHK(hipMalloc((void**)&doffs, sizeof(uint64_t) * ((size_t)n + 1)));
    if (hipMalloc((void**)&dcols, sizeof(uint32_t) * std::max<uint64_t>(off, 1)) != hipSuccess) {
        (void)hipFree(doffs);
        return fail(c, SPH_ENOMEM, "no device memory for the neighbour table");
    }
    int rc = SPH_OK;
    const uint32_t* colid = c->dist ? c->gid : c->id; /* slab runs report global indices */
    do {
        if (hipMemcpyAsync(doffs, row_offsets, sizeof(uint64_t) * ((size_t)n + 1), hipMemcpyHostToDevice, c->stream) !=
            hipSuccess) { rc = SPH_EDEVICE; break; }
        if (c->dim == 3)
            hipLaunchKernelGGL(k_nbr_fill<3>, dim3(nblk(n)), dim3(SPH_BLOCK), 0, c->stream, c->dp, c->rec[src], rows,
                               c->cap_k, rcnt, c->id, colid, doffs, dcols);
        else
            hipLaunchKernelGGL(k_nbr_fill<2>, dim3(nblk(n)), dim3(SPH_BLOCK), 0, c->stream, c->dp, c->rec[src], rows,
                               c->cap_k, rcnt, c->id, colid, doffs, dcols);
        if (hipGetLastError() != hipSuccess) { rc = SPH_EDEVICE; break; }
        if (hipMemcpyAsync(cols, dcols, sizeof(uint32_t) * off, hipMemcpyDeviceToHost, c->stream) != hipSuccess) {
            rc = SPH_EDEVICE;
            break;
        }
        if (hipStreamSynchronize(c->stream) != hipSuccess) rc = SPH_EDEVICE;
    } while (0);
    (void)hipFree(doffs);
    (void)hipFree(dcols);
    if (rc != SPH_OK) c->err = "neighbour table transfer failed";
    return rc;
}

int sph_profile_enable(sph_ctx* c, int32_t on) {
    if (!c) return SPH_EINVAL;
    c->profile = on != 0;
    return SPH_OK;
}

int sph_profile_read(sph_ctx* c, double* pair_ms, uint64_t* pair_launches, double* rebuild_ms, uint64_t* rebuilds) {
    if (!c) return SPH_EINVAL;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    double tot = 0;
    for (size_t i = 0; i < c->ev_used; ++i) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_pool[i].a, c->ev_pool[i].b));
        tot += ms;
    }
    if (pair_ms) *pair_ms = tot;
    if (pair_launches) *pair_launches = c->ev_used;
    c->ev_used = 0;
    tot = 0;
    for (size_t i = 0; i < c->rb_used; ++i) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c->rb_pool[i].a, c->rb_pool[i].b));
        tot += ms;
    }
    if (rebuild_ms) *rebuild_ms = tot;
    if (rebuilds) *rebuilds = c->rb_used;
    c->rb_used = 0;
    c->dr_used = 0;
    c->hx_used = c->ar_used = 0;
    c->halo_setup_ms = c->migrate_ms = 0;
    c->halo_setups = c->migrations = 0;
    return SPH_OK;
}

int sph_profile_read_comm(sph_ctx* c, double* ms4, uint64_t* count4) {
    if (!c || !ms4 || !count4) return SPH_EINVAL;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->comm) HIPCHK(hipStreamSynchronize(c->comm));
    for (int k = 0; k < 4; ++k) { ms4[k] = 0; count4[k] = 0; }
    for (size_t i = 0; i < c->hx_used; ++i) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c->hx_pool[i].a, c->hx_pool[i].b));
        ms4[0] += ms;
    }
    count4[0] = c->hx_used;
    for (size_t i = 0; i < c->ar_used; ++i) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c->ar_pool[i].a, c->ar_pool[i].b));
        ms4[1] += ms;
    }
    count4[1] = c->ar_used;
    ms4[2] = c->halo_setup_ms;
    count4[2] = c->halo_setups;
    ms4[3] = c->migrate_ms;
    count4[3] = c->migrations;
    return SPH_OK;
}

int sph_profile_read_kinds(sph_ctx* c, double* ms3, uint64_t* launches3) {
    if (!c || !ms3 || !launches3) return SPH_EINVAL;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int k = 0; k < 3; ++k) { ms3[k] = 0; launches3[k] = 0; }
    for (size_t i = 0; i < c->ev_used; ++i) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_pool[i].a, c->ev_pool[i].b));
        const int k = c->ev_pool[i].kind ? 1 : 0;
        ms3[k] += ms;
        ++launches3[k];
    }
    for (size_t i = 0; i < c->dr_used; ++i) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c->dr_pool[i].a, c->dr_pool[i].b));
        ms3[2] += ms;
        ++launches3[2];
    }
    return SPH_OK;
}

/* ---- built-in transport over RCCL -------------------------------------------------------
 * The library is opened at run time, so that a process which already carries an RCCL (one
 * that imported torch, say) shares that copy, and single-GPU users need none. */
struct RcclState {
    void* lib = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    int peerL = -1, peerR = 1; /* ranks of the slab neighbours */
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
};

static void* rccl_open() {
    void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    return lib;
}

static int rccl_sendrecv(void* user, const void* sL, uint64_t nsL, void* rL, uint64_t nrL, const void* sR, uint64_t nsR,
                         void* rR, uint64_t nrR, void* stream) {
    RcclState* r = (RcclState*)user;
    hipStream_t st = (hipStream_t)stream;
    if (!(nsL || nrL || nsR || nrR)) return 0;
    bool ok = r->GroupStart() == ncclSuccess;
    if (nsL) ok = ok && r->Send(sL, nsL, ncclUint8, r->peerL, r->comm, st) == ncclSuccess;
    if (nrL) ok = ok && r->Recv(rL, nrL, ncclUint8, r->peerL, r->comm, st) == ncclSuccess;
    if (nsR) ok = ok && r->Send(sR, nsR, ncclUint8, r->peerR, r->comm, st) == ncclSuccess;
    if (nrR) ok = ok && r->Recv(rR, nrR, ncclUint8, r->peerR, r->comm, st) == ncclSuccess;
    ok = (r->GroupEnd() == ncclSuccess) && ok;
    return ok ? 0 : 1;
}

static int rccl_allreduce(void* user, double* vals, int32_t n, int32_t op, void* stream) {
    RcclState* r = (RcclState*)user;
    return r->AllReduce(vals, vals, (size_t)n, ncclDouble, op ? ncclSum : ncclMax, r->comm, (hipStream_t)stream) == ncclSuccess
               ? 0
               : 1;
}

static int rccl_allreduce_max_sum(void* user, double* maxvals, int32_t nmax, double* sumvals, int32_t nsum, void* stream) {
    RcclState* r = (RcclState*)user;
    hipStream_t st = (hipStream_t)stream;
    bool ok = r->GroupStart() == ncclSuccess;
    ok = ok && r->AllReduce(maxvals, maxvals, (size_t)nmax, ncclDouble, ncclMax, r->comm, st) == ncclSuccess;
    ok = ok && r->AllReduce(sumvals, sumvals, (size_t)nsum, ncclDouble, ncclSum, r->comm, st) == ncclSuccess;
    ok = (r->GroupEnd() == ncclSuccess) && ok;
    return ok ? 0 : 1;
}

static void rccl_release(void* p) {
    RcclState* r = (RcclState*)p;
    if (r->comm && r->CommDestroy) (void)r->CommDestroy(r->comm);
    delete r; /* the library handle stays open: other users in the process may share it */
}

/* Opens the library and looks the calls up; returns nullptr or what is missing. */
static const char* rccl_load(RcclState** out) {
    RcclState* r = new RcclState();
    r->lib = rccl_open();
    if (!r->lib) {
        delete r;
        return "librccl.so.1 could not be opened";
    }
#define SYM(field, name)                                                  \
    r->field = (decltype(r->field))dlsym(r->lib, name);                   \
    if (!r->field) {                                                      \
        delete r;                                                         \
        return "RCCL lacks " name;                                        \
    }
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(Send, "ncclSend")
    SYM(Recv, "ncclRecv")
    SYM(AllReduce, "ncclAllReduce")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
#undef SYM
    *out = r;
    return nullptr;
}

int sph_rccl_selftest(int32_t device, char* msg, int32_t msg_len) {
    auto say = [&](const char* m) {
        if (msg && msg_len > 0) snprintf(msg, (size_t)msg_len, "%s", m);
    };
    say("");
    if (device >= 0 && hipSetDevice(device) != hipSuccess) { say("hipSetDevice failed"); return SPH_EDEVICE; }
    RcclState* r = nullptr;
    if (const char* why = rccl_load(&r)) { say(why); return SPH_ECOMM; }
    ncclUniqueId id;
    auto get = (ncclResult_t(*)(ncclUniqueId*))dlsym(r->lib, "ncclGetUniqueId");
    if (!get || get(&id) != ncclSuccess) { say("ncclGetUniqueId failed"); delete r; return SPH_ECOMM; }
    if (r->CommInitRank(&r->comm, 1, id, 0) != ncclSuccess) { say("ncclCommInitRank (one rank) failed"); delete r; return SPH_ECOMM; }
    r->rank = 0;
    r->nranks = 1;
    r->peerL = r->peerR = 0; /* both neighbours are this rank itself */
    int rc = SPH_OK;
    hipStream_t st = nullptr;
    char* buf = nullptr;
    const size_t n = 1 << 16; /* bytes per message */
    std::vector<char> h(4 * n), back(4 * n);
    double hv[8] = {1.5, -2.0, 3.25, 0.0, 7.0, 10.0, 20.5, -4.0}, hb[8];
    do {
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipMalloc((void**)&buf, 4 * n + 64) != hipSuccess) {
            say("stream or buffer could not be made");
            rc = SPH_EDEVICE;
            break;
        }
        for (size_t k = 0; k < 2 * n; ++k) h[k] = (char)(k * 7 + (k >> 9));
        /* [0,n) goes left, [n,2n) goes right; what comes from the left lands in [2n,3n), from the right in [3n,4n) */
        if (hipMemcpyAsync(buf, h.data(), 2 * n, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemsetAsync(buf + 2 * n, 0, 2 * n, st) != hipSuccess) { say("copy failed"); rc = SPH_EDEVICE; break; }
        if (rccl_sendrecv(r, buf, n, buf + 2 * n, n, buf + n, n, buf + 3 * n, n, (void*)st) != 0) { say("grouped send/recv failed"); rc = SPH_ECOMM; break; }
        double* dv = (double*)(buf + 4 * n);
        if (hipMemcpyAsync(dv, hv, sizeof hv, hipMemcpyHostToDevice, st) != hipSuccess) { say("copy failed"); rc = SPH_EDEVICE; break; }
        if (rccl_allreduce_max_sum(r, dv, RED_NMAX, dv + RED_NMAX, RED_NSUM, (void*)st) != 0) { say("grouped all-reduce failed"); rc = SPH_ECOMM; break; }
        if (rccl_allreduce(r, dv, 2, 0, (void*)st) != 0 || rccl_allreduce(r, dv + 2, 2, 1, (void*)st) != 0) { say("all-reduce failed"); rc = SPH_ECOMM; break; }
        if (hipMemcpyAsync(back.data(), buf, 4 * n, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(hb, dv, sizeof hb, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            say("copy back failed");
            rc = SPH_EDEVICE;
            break;
        }
        /* with itself as both neighbours the first message sent is the first received */
        if (memcmp(back.data() + 2 * n, h.data(), n) != 0 || memcmp(back.data() + 3 * n, h.data() + n, n) != 0) {
            say("send/recv delivered other bytes");
            rc = SPH_ECOMM;
            break;
        }
        if (memcmp(hb, hv, sizeof hv) != 0) { say("a one-rank all-reduce changed the values"); rc = SPH_ECOMM; break; }
        say("ok");
    } while (0);
    if (buf) (void)hipFree(buf);
    if (st) (void)hipStreamDestroy(st);
    rccl_release(r);
    return rc;
}

int sph_rccl_unique_id(void* unique_id_128) {
    if (!unique_id_128) return SPH_EINVAL;
    void* lib = rccl_open();
    if (!lib) return SPH_ECOMM;
    auto get = (ncclResult_t(*)(ncclUniqueId*))dlsym(lib, "ncclGetUniqueId");
    if (!get) return SPH_ECOMM;
    ncclUniqueId id;
    if (get(&id) != ncclSuccess) return SPH_ECOMM;
    memcpy(unique_id_128, &id, sizeof id);
    return SPH_OK;
}

int sph_use_rccl(sph_ctx* c, const void* unique_id_128) {
    if (!c || !unique_id_128) return SPH_EINVAL;
    if (!c->dist) return fail(c, SPH_ESTATE, "a transport is only used when nranks > 1");
    if (c->rccl) return fail(c, SPH_ESTATE, "the RCCL transport is already installed");
    HIPCHK(hipSetDevice(c->device));
    RcclState* r = nullptr;
    if (const char* why = rccl_load(&r)) return fail(c, SPH_ECOMM, why);
    r->rank = c->rank;
    r->nranks = c->nranks;
    r->peerL = c->rank - 1;
    r->peerR = c->rank + 1;
    ncclUniqueId id;
    memcpy(&id, unique_id_128, sizeof id);
    const ncclResult_t ir = r->CommInitRank(&r->comm, c->nranks, id, c->rank);
    if (ir != ncclSuccess) {
        auto es = (const char* (*)(ncclResult_t))dlsym(r->lib, "ncclGetErrorString");
        auto le = (const char* (*)(ncclComm_t))dlsym(r->lib, "ncclGetLastError");
        c->err = std::string("ncclCommInitRank failed: ") + (es ? es(ir) : "?") + " / " + (le ? le(nullptr) : "");
        delete r;
        return SPH_ECOMM;
    }
    if (c->have_tr) { /* the caller gave up waiting and installed another transport meanwhile */
        (void)r->CommDestroy(r->comm);
        delete r;
        return fail(c, SPH_ESTATE, "another transport was installed while the RCCL communicator was being created");
    }
    c->rccl = r;
    c->rccl_free = rccl_release;
    sph_transport t;
    memset(&t, 0, sizeof t);
    t.struct_size = sizeof t;
    t.user = r;
    t.host_buffers = 0;
    t.sendrecv = rccl_sendrecv;
    t.allreduce = rccl_allreduce;
    t.allreduce_max_sum = rccl_allreduce_max_sum;
    c->tr = t;
    c->have_tr = true;
    return SPH_OK;
}


}  /* extern "C" */
