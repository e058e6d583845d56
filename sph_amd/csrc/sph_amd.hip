/* Context and C ABI of the MI355X SPH pair path (see include/sph_amd.h). */
#include "../../include/sph_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sph_kernels.h"

namespace {

constexpr int NREC = 3;

struct EventPair {
    hipEvent_t a, b;
};

}  // namespace

struct sph_ctx {
    sph_params prm;
    DevParams dp;
    int dim = 0;
    bool xsph = false;
    uint32_t n = 0, stride = 0, nblocks = 0, npairblocks = 0;
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
    uint32_t* id = nullptr;
    uint32_t* id_alt = nullptr;

    /* sort, cells, candidate lists */
    uint32_t *keys = nullptr, *keys_alt = nullptr, *perm = nullptr, *perm_alt = nullptr;
    uint32_t* hist = nullptr;
    size_t hist_cap = 0;
    uint32_t* bsum = nullptr;
    size_t bsum_cap = 0;
    uint32_t *cbeg = nullptr, *cend = nullptr;
    size_t cell_cap = 0;
    uint32_t* nbr = nullptr;
    uint32_t cap_k = 0;
    uint32_t* cnt = nullptr;
    uint32_t* need = nullptr;
    GridParams grid;
    bool list_valid = false;
    bool lists_for_step = false; /* built by the stepping path (wall-wall pairs may be left out) */
    double skin_frac = 0.1, r_list2 = 0, disp_limit = 0;
    uint32_t brick_log2 = 2;
    double disp_host = 0;

    DevScalars* S = nullptr;
    DevScalars* S_host = nullptr; /* pinned */
    double* part = nullptr;  /* per-workgroup partials of the pair kernel */
    double* part2 = nullptr; /* their first fold */
    double* bbox_part = nullptr;
    double* bbox_host = nullptr; /* pinned */
    uint32_t* need_host = nullptr;
    void* tmp = nullptr; /* n * 8 bytes staging, original order */

    uint64_t rebuilds = 0, redos = 0;
    bool uploaded_n = false, uploaded_np1 = false;

    /* pair-kernel timing */
    bool profile = false;
    std::vector<EventPair> ev_pool;
    size_t ev_used = 0;
    std::vector<EventPair> rb_pool;
    size_t rb_used = 0;
};

namespace {

constexpr int BBOX_BLOCKS = 1024;

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

inline uint32_t nblk(uint32_t n) { return (n + SPH_BLOCK - 1) / SPH_BLOCK; }

template <typename T>
int dalloc(sph_ctx* c, T** p, size_t count) {
    HIPCHK(hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T)));
    return SPH_OK;
}

int fail(sph_ctx* c, int code, const char* msg) {
    c->err = msg;
    return code;
}

/* index helpers for the n-state component arrays */
inline int NX(int d) { return d; }
inline int NV(int d) { return 3 + d; }
inline int NVV(int d) { return 6 + d; }
inline int NF(int d) { return 9 + d; }
enum { F_RHO = 0, F_RRHO = 1, F_P = 2, F_M = 3 };

int exclusive_scan(sph_ctx* c, const uint32_t* in, uint32_t* out, uint32_t n) {
    const uint32_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nb > c->bsum_cap) {
        if (c->bsum) HIPCHK(hipFree(c->bsum));
        c->bsum = nullptr;
        CHK(dalloc(c, &c->bsum, (size_t)nb * 2));
        c->bsum_cap = (size_t)nb * 2;
    }
    hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(SPH_BLOCK), 0, c->stream, in, n, c->bsum);
    hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(SPH_BLOCK), 0, c->stream, c->bsum, nb);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(SPH_BLOCK), 0, c->stream, in, out, n, c->bsum);
    HIPCHK(hipGetLastError());
    return SPH_OK;
}

/* keys (unsorted, in c->keys) -> c->keys sorted, c->perm = sorted position -> old index */
int radix_sort(sph_ctx* c, uint32_t ncells) {
    const uint32_t n = c->n;
    const uint32_t nb = (n + RS_TILE - 1) / RS_TILE;
    const size_t hsz = (size_t)256 * nb;
    if (hsz > c->hist_cap) {
        if (c->hist) HIPCHK(hipFree(c->hist));
        c->hist = nullptr;
        CHK(dalloc(c, &c->hist, hsz));
        c->hist_cap = hsz;
    }
    int bits = 1;
    while (bits < 32 && (1ull << bits) < (unsigned long long)ncells) ++bits;
    const int passes = (bits + 7) / 8;
    const uint32_t* vin = nullptr;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        hipLaunchKernelGGL(k_rs_hist, dim3(nb), dim3(SPH_BLOCK), 0, c->stream, c->keys, n, shift, c->hist, nb);
        CHK(exclusive_scan(c, c->hist, c->hist, (uint32_t)hsz));
        hipLaunchKernelGGL(k_rs_scatter, dim3(nb), dim3(SPH_BLOCK), 0, c->stream, c->keys, vin, c->keys_alt,
                           c->perm_alt, n, shift, c->hist, nb);
        std::swap(c->keys, c->keys_alt);
        std::swap(c->perm, c->perm_alt);
        vin = c->perm;
    }
    HIPCHK(hipGetLastError());
    return SPH_OK;
}

template <int DIM>
void launch_hash(sph_ctx* c, const Rec* r) {
    hipLaunchKernelGGL(k_hash<DIM>, dim3(c->nblocks), dim3(SPH_BLOCK), 0, c->stream, r, c->n, c->grid, c->keys);
}

int build_lists(sph_ctx* c, const Rec* r, uint32_t wall_below) {
    for (int attempt = 0; attempt < 4; ++attempt) {
        HIPCHK(hipMemsetAsync(c->need, 0, sizeof(uint32_t), c->stream));
        if (c->dim == 3)
            hipLaunchKernelGGL(k_build_list<3>, dim3(c->nblocks), dim3(SPH_BLOCK), 0, c->stream, r, c->n,
                               c->grid, c->cbeg, c->cend, c->r_list2, c->nbr, c->cap_k, c->cap_k, c->cnt, c->need, c->id, wall_below);
        else
            hipLaunchKernelGGL(k_build_list<2>, dim3(c->nblocks), dim3(SPH_BLOCK), 0, c->stream, r, c->n,
                               c->grid, c->cbeg, c->cend, c->r_list2, c->nbr, c->cap_k, c->cap_k, c->cnt, c->need, c->id, wall_below);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(c->need_host, c->need, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        const uint32_t need = *c->need_host;
        if (need <= c->cap_k) return SPH_OK;
        /* some particle has more candidates than the lists hold: grow and build again */
        const uint32_t newcap = (need + need / 8 + 7u) & ~3u;
        HIPCHK(hipFree(c->nbr));
        c->nbr = nullptr;
        CHK(dalloc(c, &c->nbr, (size_t)newcap * c->stride));
        c->cap_k = newcap;
    }
    return fail(c, SPH_ESTATE, "candidate lists kept overflowing");
}

/* Re-bin on the positions held in record buffer `src`, bring every live array into the new
 * cell order, rebuild cell table and candidate lists, reset the travel bound. */
int rebuild(sph_ctx* c, int src, bool for_step = false) {
    const uint32_t n = c->n;
    const int dim = c->dim;
    EventPair* ev = nullptr;
    if (c->profile) {
        if (c->rb_used == c->rb_pool.size()) {
            EventPair e;
            HIPCHK(hipEventCreate(&e.a));
            HIPCHK(hipEventCreate(&e.b));
            c->rb_pool.push_back(e);
        }
        ev = &c->rb_pool[c->rb_used++];
        HIPCHK(hipEventRecord(ev->a, c->stream));
    }
    /* bounding box */
    hipLaunchKernelGGL(k_bbox, dim3(BBOX_BLOCKS), dim3(SPH_BLOCK), 0, c->stream, c->rec[src], n, dim, c->bbox_part);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->bbox_host, c->bbox_part, sizeof(double) * 6 * BBOX_BLOCKS, hipMemcpyDeviceToHost,
                          c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int b = 0; b < BBOX_BLOCKS; ++b)
        for (int d = 0; d < 3; ++d) {
            lo[d] = std::min(lo[d], c->bbox_host[b * 6 + d]);
            hi[d] = std::max(hi[d], c->bbox_host[b * 6 + 3 + d]);
        }
    /* cell edge: candidate radius plus a relative margin that covers the rounding of the
     * cell-coordinate arithmetic */
    const double r_list = std::sqrt(c->prm.search_radius_sq) * (1.0 + c->skin_frac);
    GridParams g;
    memset(&g, 0, sizeof g);
    for (int d = 0; d < 3; ++d) {
        g.lo[d] = 0;
        g.inv[d] = 1;
        g.n[d] = 1;
        g.nt[d] = 1;
        g.ts[d] = 0;
    }
    for (int d = 0; d < dim; ++d)
        if (!(hi[d] >= lo[d])) { lo[d] = 0; hi[d] = 0; } /* no finite particle */
    /* bricks of 2^brick_log2 cells per axis; the grid is padded to whole bricks.  If a stray
     * particle blew the box up, stretch the cells until the table fits its allocation. */
    double edge = r_list * (1.0 + 1e-9);
    uint64_t ncells = 0;
    for (int attempt = 0; attempt < 200; ++attempt) {
        double tot = 1.0;
        for (int d = 0; d < dim; ++d) {
            const double cells = std::floor((hi[d] - lo[d]) / edge) + 1.0;
            const double tb = (double)(1u << c->brick_log2);
            const double nt = std::ceil(cells / tb);
            tot *= nt * tb;
        }
        if (tot <= (double)c->cell_cap && tot < 2147483648.0) {
            ncells = 1;
            for (int d = 0; d < dim; ++d) {
                const double cells = std::floor((hi[d] - lo[d]) / edge) + 1.0;
                const uint32_t tb = 1u << c->brick_log2;
                g.ts[d] = c->brick_log2;
                g.nt[d] = (int)std::ceil(cells / tb);
                g.n[d] = g.nt[d] * (int)tb;
                g.lo[d] = lo[d];
                g.inv[d] = 1.0 / edge;
                ncells *= (uint64_t)g.n[d];
            }
            break;
        }
        edge *= 1.26;
    }
    if (ncells == 0) return fail(c, SPH_ESTATE, "cell grid exceeds its table");
    g.ncells = (uint32_t)ncells;
    c->grid = g;

    if (dim == 3) launch_hash<3>(c, c->rec[src]);
    else launch_hash<2>(c, c->rec[src]);
    CHK(radix_sort(c, g.ncells));

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
    for (int q = 0; q < 2; ++q)
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
    a.sf[4] = c->pRrho;
    a.df[4] = c->pRrho_alt;
    a.sf[5] = c->pp;
    a.df[5] = c->pp_alt;
    a.nf = 6;
    a.rs[0] = c->rec[c->b0];
    a.rd[0] = c->rec_alt;
    a.nr = 1;
    Rec* spare = nullptr;
    if (src != c->b0) {
        /* positions came from a scratch buffer: carry it along through a free record buffer */
        int freeb = 0;
        while (freeb == src || freeb == c->b0) ++freeb;
        a.rs[1] = c->rec[src];
        a.rd[1] = c->rec[freeb];
        a.nr = 2;
        spare = c->rec[freeb];
    }
    a.ids = c->id;
    a.idd = c->id_alt;
    hipLaunchKernelGGL(k_reorder, dim3(c->nblocks), dim3(SPH_BLOCK), 0, c->stream, a);
    HIPCHK(hipGetLastError());
    for (int q = 0; q < n_nd; ++q) std::swap(c->nd[map_nd[q]], c->nd_alt[map_nd[q]]);
    for (int q = n_nd; q < k; ++q) std::swap(c->pd[map_pd[q - n_nd]], c->pd_alt[map_pd[q - n_nd]]);
    for (int q = 0; q < 4; ++q) std::swap(c->nfl[q], c->nfl_alt[q]);
    std::swap(c->pRrho, c->pRrho_alt);
    std::swap(c->pp, c->pp_alt);
    std::swap(c->rec[c->b0], c->rec_alt);
    std::swap(c->id, c->id_alt);
    if (spare) {
        int freeb = 0;
        while (freeb == src || freeb == c->b0) ++freeb;
        std::swap(c->rec[src], c->rec[freeb]);
    }

    /* cell table */
    HIPCHK(hipMemsetAsync(c->cbeg, 0xFF, sizeof(uint32_t) * g.ncells, c->stream));
    HIPCHK(hipMemsetAsync(c->cend, 0, sizeof(uint32_t) * g.ncells, c->stream));
    hipLaunchKernelGGL(k_cell_mark, dim3(c->nblocks), dim3(SPH_BLOCK), 0, c->stream, c->keys, n, c->cbeg, c->cend);
    HIPCHK(hipGetLastError());

    c->r_list2 = r_list * r_list;
    c->disp_limit = 0.5 * (r_list - std::sqrt(c->prm.search_radius_sq)) * 0.999;
    /* inside a step, wall-wall pairs can be dropped while the walls are at rest */
    uint32_t wall_below = 0;
    if (for_step && c->dp.bound_parts > 0) {
        WallRestArgs w;
        memset(&w, 0, sizeof w);
        w.n = n;
        w.wall_below = c->dp.bound_parts;
        w.dim = dim;
        w.xsph = c->xsph;
        w.rec = c->rec[src];
        for (int d = 0; d < 3; ++d) {
            w.nv[d] = c->nd[NV(d)];
            w.nV[d] = c->nd[NVV(d)];
            w.nf[d] = c->nd[NF(d)];
        }
        w.id = c->id;
        HIPCHK(hipMemsetAsync(c->need, 0, sizeof(uint32_t), c->stream));
        hipLaunchKernelGGL(k_walls_at_rest, dim3(c->nblocks), dim3(SPH_BLOCK), 0, c->stream, w, c->need);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(c->need_host, c->need, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (*c->need_host == 0) wall_below = c->dp.bound_parts;
    }
    CHK(build_lists(c, c->rec[src], wall_below));
    c->lists_for_step = for_step;

    /* travel bound restarts */
    const double z[2] = {0.0, c->disp_limit};
    HIPCHK(hipMemcpyAsync(&c->S->disp, z, sizeof z, hipMemcpyHostToDevice, c->stream));
    c->disp_host = 0;
    c->list_valid = true;
    ++c->rebuilds;
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
    return s;
}

template <bool FUSED>
int launch_pair(sph_ctx* c, const Rec* rin, Rec* rout, OutState o, int store_out, int zero_wall) {
    NState ns = n_state(c);
    dim3 g(c->npairblocks), b(SPH_BLOCK);
    static const int dbg = getenv("SPH_DBG") ? atoi(getenv("SPH_DBG")) : 0; /* timing experiments only */
#define LP(D, X)                                                                                               \
    hipLaunchKernelGGL((k_pair<D, X, FUSED>), g, b, 0, c->stream, c->dp, rin, c->nbr, c->cap_k, c->cnt, o,     \
                       store_out, zero_wall, ns, c->id, rout, c->pp, c->S, c->part, dbg)
    if (c->dim == 3) LP(3, true);
    else if (c->xsph) LP(2, true);
    else LP(2, false);
#undef LP
    HIPCHK(hipGetLastError());
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
    hipLaunchKernelGGL(k_pack, dim3(c->nblocks), dim3(SPH_BLOCK), 0, c->stream, a);
    HIPCHK(hipGetLastError());
    return SPH_OK;
}

int read_scalars(sph_ctx* c) {
    HIPCHK(hipMemcpyAsync(c->S_host, c->S, sizeof(DevScalars), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
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
}

/* One Newmark_Beta call, or its first `nsweeps` sweeps when partial. */
int one_step(sph_ctx* c, int nsweeps, bool partial) {
    const int subits = c->prm.subits;
    if (!partial) nsweeps = subits;
    int cur = c->b0;
    for (int mode = 0;; ++mode) {
        /* mode 0: keep the lists if the travel budget allows; 1: fresh lists at the step's
         * start; 2: fresh lists before every sweep (exact for any motion) */
        if (mode >= 1 || !c->list_valid || !c->lists_for_step || !(c->disp_host <= 0.5 * c->disp_limit))
            CHK(rebuild(c, c->b0, true));
        hipLaunchKernelGGL(k_step_begin, dim3(1), dim3(1), 0, c->stream, c->S);
        cur = c->b0;
        for (int k = 0; k < nsweeps; ++k) {
            if (mode == 2 && k > 0) {
                /* the start state is not needed again: let the sweep's input take its place */
                c->b0 = cur;
                CHK(rebuild(c, cur, true));
            }
            int nxt = 0;
            while (nxt == cur || nxt == c->b0) ++nxt;
            const bool last = k == nsweeps - 1;
            EventPair* ev = nullptr;
            if (c->profile) {
                if (c->ev_used == c->ev_pool.size()) {
                    EventPair e;
                    HIPCHK(hipEventCreate(&e.a));
                    HIPCHK(hipEventCreate(&e.b));
                    c->ev_pool.push_back(e);
                }
                ev = &c->ev_pool[c->ev_used++];
                HIPCHK(hipEventRecord(ev->a, c->stream));
            }
            CHK(launch_pair<true>(c, c->rec[cur], c->rec[nxt], out_state(c, SPH_STATE_NP1), last ? 1 : 0, 1));
            if (ev) HIPCHK(hipEventRecord(ev->b, c->stream));
            hipLaunchKernelGGL(k_fold_parts, dim3(PART_GROUPS), dim3(SPH_BLOCK), 0, c->stream, c->part, c->npairblocks,
                               PART_FIELDS, c->part2);
            hipLaunchKernelGGL(k_sweep_finalize, dim3(1), dim3(1), 0, c->stream, c->dp, c->S, c->part2, k,
                               (last && !partial) ? 1 : (last ? 2 : 0), mode < 2 ? 1 : 0);
            cur = nxt;
        }
        if (!partial) {
            CommitArgs a;
            memset(&a, 0, sizeof a);
            a.n = c->n;
            a.dim = c->dim;
            a.xsph = c->xsph;
            a.rec = c->rec[cur];
            for (int d = 0; d < 3; ++d) {
                a.pf[d] = c->pd[d];
                a.pV[d] = c->pd[3 + d];
                a.nx[d] = c->nd[NX(d)];
                a.nv[d] = c->nd[NV(d)];
                a.nV[d] = c->nd[NVV(d)];
                a.nf[d] = c->nd[NF(d)];
            }
            a.pRrho = c->pRrho;
            a.pp = c->pp;
            a.nrho = c->nfl[F_RHO];
            a.np = c->nfl[F_P];
            a.nRrho = c->nfl[F_RRHO];
            hipLaunchKernelGGL(k_commit, dim3(c->nblocks), dim3(SPH_BLOCK), 0, c->stream, a, c->S);
            hipLaunchKernelGGL(k_commit_scalars, dim3(1), dim3(1), 0, c->stream, c->S);
        }
        HIPCHK(hipGetLastError());
        CHK(read_scalars(c));
        c->disp_host = c->S_host->disp;
        if (c->S_host->violation && mode < 2) {
            ++c->redos;
            continue;
        }
        break;
    }
    c->b0 = cur;
    return SPH_OK;
}

template <typename T>
int put_component(sph_ctx* c, T* dst, size_t stride, const T* host) {
    HIPCHK(hipMemcpyAsync(c->tmp, host, sizeof(T) * c->n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_put<T>, dim3(c->nblocks), dim3(SPH_BLOCK), 0, c->stream, dst, stride, (const T*)c->tmp,
                       c->id, c->n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return SPH_OK;
}

template <typename T>
int get_component(sph_ctx* c, const T* src, size_t stride, T* host) {
    hipLaunchKernelGGL(k_get<T>, dim3(c->nblocks), dim3(SPH_BLOCK), 0, c->stream, src, stride, (T*)c->tmp, c->id,
                       c->n);
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
    if (p->dim != 2 && p->dim != 3) return fail(c, SPH_EINVAL, "dim must be 2 or 3");
    if (p->variant != SPH_VARIANT_WCSPH && p->variant != SPH_VARIANT_WCXSPH)
        return fail(c, SPH_EINVAL, "unknown variant");
    if (p->variant == SPH_VARIANT_WCSPH && p->dim != 2)
        return fail(c, SPH_EINVAL, "the single-mass variant exists in 2-D only");
    if (p->npts == 0 || p->npts >= (1ull << 31)) return fail(c, SPH_EINVAL, "npts out of range");
    if (p->bound_parts > p->npts) return fail(c, SPH_EINVAL, "bound_parts > npts");
    if (!(p->H > 0) || !(p->search_radius_sq > 0) || !(p->rho0 > 0) || p->subits < 1 || p->subits > 64 ||
        p->g_axis < 0 || p->g_axis >= p->dim)
        return fail(c, SPH_EINVAL, "bad physical constants");

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
    c->stride = (c->n + 63u) & ~63u;
    c->nblocks = nblk(c->n);
    c->npairblocks = ((c->n + PAIR_PARTICLES - 1) / PAIR_PARTICLES + 7u) & ~7u; /* a multiple of the XCD count */
    if (p->skin_frac > 0) c->skin_frac = p->skin_frac;
    c->brick_log2 = c->dim == 3 ? 2u : 3u;
    if (const char* e = getenv("SPH_BRICK_LOG2")) c->brick_log2 = (uint32_t)std::min(5, std::max(0, atoi(e)));

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
    d.npts_d = 1.0 * (double)p->npts;

    const size_t n = c->n;
    for (int q = 0; q < 4; ++q)
        for (int k = 0; k < c->dim; ++k) {
            CHK(dalloc(c, &c->nd[q * 3 + k], n));
            CHK(dalloc(c, &c->nd_alt[q * 3 + k], n));
            HIPCHK(hipMemsetAsync(c->nd[q * 3 + k], 0, n * sizeof(double), c->stream));
        }
    for (int q = 0; q < 4; ++q) {
        CHK(dalloc(c, &c->nfl[q], n));
        CHK(dalloc(c, &c->nfl_alt[q], n));
        HIPCHK(hipMemsetAsync(c->nfl[q], 0, n * sizeof(float), c->stream));
    }
    for (int b = 0; b < NREC; ++b) {
        CHK(dalloc(c, &c->rec[b], n));
        HIPCHK(hipMemsetAsync(c->rec[b], 0, n * sizeof(Rec), c->stream));
    }
    CHK(dalloc(c, &c->rec_alt, n));
    for (int q = 0; q < 2; ++q)
        for (int k = 0; k < c->dim; ++k) {
            CHK(dalloc(c, &c->pd[q * 3 + k], n));
            CHK(dalloc(c, &c->pd_alt[q * 3 + k], n));
            HIPCHK(hipMemsetAsync(c->pd[q * 3 + k], 0, n * sizeof(double), c->stream));
        }
    CHK(dalloc(c, &c->pRrho, n));
    CHK(dalloc(c, &c->pRrho_alt, n));
    HIPCHK(hipMemsetAsync(c->pRrho, 0, n * sizeof(float), c->stream));
    CHK(dalloc(c, &c->pp, n));
    CHK(dalloc(c, &c->pp_alt, n));
    HIPCHK(hipMemsetAsync(c->pp, 0, n * sizeof(float), c->stream));
    CHK(dalloc(c, &c->id, n));
    CHK(dalloc(c, &c->id_alt, n));
    hipLaunchKernelGGL(k_iota, dim3(c->nblocks), dim3(SPH_BLOCK), 0, c->stream, c->id, c->n);
    CHK(dalloc(c, &c->keys, n));
    CHK(dalloc(c, &c->keys_alt, n));
    CHK(dalloc(c, &c->perm, n));
    CHK(dalloc(c, &c->perm_alt, n));
    c->cell_cap = std::max<size_t>(8 * n, (size_t)1 << 20);
    CHK(dalloc(c, &c->cbeg, c->cell_cap));
    CHK(dalloc(c, &c->cend, c->cell_cap));
    c->cap_k = p->max_neighbours > 0 ? (uint32_t)p->max_neighbours : (c->dim == 3 ? 96u : 40u);
    c->cap_k = (c->cap_k + 3u) & ~3u;
    CHK(dalloc(c, &c->nbr, (size_t)c->cap_k * c->stride));
    CHK(dalloc(c, &c->cnt, n));
    CHK(dalloc(c, &c->need, 1));
    CHK(dalloc(c, &c->S, 1));
    HIPCHK(hipHostMalloc((void**)&c->S_host, sizeof(DevScalars)));
    HIPCHK(hipHostMalloc((void**)&c->bbox_host, sizeof(double) * 6 * BBOX_BLOCKS));
    HIPCHK(hipHostMalloc((void**)&c->need_host, sizeof(uint32_t)));
    CHK(dalloc(c, &c->part, (size_t)PART_FIELDS * c->npairblocks));
    CHK(dalloc(c, &c->part2, (size_t)PART_FIELDS * PART_GROUPS));
    CHK(dalloc(c, &c->bbox_part, (size_t)6 * BBOX_BLOCKS));
    HIPCHK(hipMalloc(&c->tmp, n * 8));

    memset(c->S_host, 0, sizeof(DevScalars));
    c->S_host->dt = p->dt0;
    c->S_host->dt_next = p->dt0;
    HIPCHK(hipMemcpyAsync(c->S, c->S_host, sizeof(DevScalars), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SPH_OK;
}

void sph_destroy(sph_ctx* c) {
    if (!c) return;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
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
    void* ptrs[] = {c->pp, c->pp_alt, c->pRrho, c->pRrho_alt, c->id,   c->id_alt, c->keys, c->keys_alt, c->perm,     c->perm_alt,
                    c->hist,  c->bsum,      c->cbeg, c->cend,   c->nbr,  c->cnt,      c->need,     c->S,
                    c->part, c->part2, c->bbox_part, c->tmp};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (c->S_host) (void)hipHostFree(c->S_host);
    if (c->bbox_host) (void)hipHostFree(c->bbox_host);
    if (c->need_host) (void)hipHostFree(c->need_host);
    for (auto& e : c->ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto& e : c->rb_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int sph_upload(sph_ctx* c, int32_t which, const sph_fields* s) {
    if (!c || !s) return SPH_EINVAL;
    if (which != SPH_STATE_N && which != SPH_STATE_NP1) return fail(c, SPH_EINVAL, "bad state selector");
    HIPCHK(hipSetDevice(c->device));
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
            if (s->x[d]) {
                CHK(put_component<double>(c, &r->xv[d][0], 8, s->x[d]));
                c->list_valid = false;
            }
            if (s->v[d]) CHK(put_component<double>(c, &r->xv[d][1], 8, s->v[d]));
            if (s->f[d]) CHK(put_component<double>(c, c->pd[d], 1, s->f[d]));
            if (s->V[d] && c->xsph) CHK(put_component<double>(c, c->pd[3 + d], 1, s->V[d]));
        }
        if (s->rho) CHK(put_component<float>(c, &r->rho, 16, s->rho));
        if (s->p) CHK(put_component<float>(c, c->pp, 1, s->p));
        if (s->m) CHK(put_component<float>(c, &r->m, 16, s->m));
        if (s->Rrho) CHK(put_component<float>(c, c->pRrho, 1, s->Rrho));
        if (s->rho || s->p) {
            hipLaunchKernelGGL(k_fix_prr, dim3(c->nblocks), dim3(SPH_BLOCK), 0, c->stream, r, c->pp, c->n);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(c->stream));
        }
        c->uploaded_np1 = true;
    }
    return SPH_OK;
}

int sph_download(sph_ctx* c, int32_t which, const sph_fields* s) {
    if (!c || !s) return SPH_EINVAL;
    if (which != SPH_STATE_N && which != SPH_STATE_NP1) return fail(c, SPH_EINVAL, "bad state selector");
    HIPCHK(hipSetDevice(c->device));
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
    if (!c) return SPH_EINVAL;
    if (which != SPH_STATE_N && which != SPH_STATE_NP1) return fail(c, SPH_EINVAL, "bad state selector");
    if (which == SPH_STATE_N ? !c->uploaded_n : !c->uploaded_np1) return fail(c, SPH_ESTATE, "state was never uploaded");
    HIPCHK(hipSetDevice(c->device));
    int src = c->b0;
    if (which == SPH_STATE_N) {
        src = (c->b0 + 1) % NREC;
        CHK(pack_n_into(c, c->rec[src]));
    }
    CHK(rebuild(c, src));
    if (which == SPH_STATE_N) {
        /* rebuild moved the scratch buffer; find it again: it is the one that is neither b0 nor free */
        src = (c->b0 + 1) % NREC;
        /* lists were built for state n's positions; state n+1 may differ */
        c->list_valid = false;
    }
    hipLaunchKernelGGL(k_step_begin, dim3(1), dim3(1), 0, c->stream, c->S);
    CHK(launch_pair<false>(c, c->rec[src], nullptr, out_state(c, which), 1, 0));
    hipLaunchKernelGGL(k_fold_parts, dim3(PART_GROUPS), dim3(SPH_BLOCK), 0, c->stream, c->part, c->npairblocks, 1, c->part2);
    hipLaunchKernelGGL(k_forces_finalize, dim3(1), dim3(1), 0, c->stream, c->S, c->part2);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return SPH_OK;
}

int sph_zero_boundary_forces(sph_ctx* c, int32_t which) {
    if (!c) return SPH_EINVAL;
    if (which != SPH_STATE_N && which != SPH_STATE_NP1) return fail(c, SPH_EINVAL, "bad state selector");
    HIPCHK(hipSetDevice(c->device));
    OutState o = out_state(c, which);
    hipLaunchKernelGGL(k_zero_wall_f, dim3(c->nblocks), dim3(SPH_BLOCK), 0, c->stream, o.f[0], o.f[1], o.f[2], c->dim,
                       c->id, c->n, c->dp.bound_parts);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return SPH_OK;
}

int sph_step(sph_ctx* c, int32_t nsteps, sph_scalars* out) {
    if (!c || nsteps < 0) return SPH_EINVAL;
    if (!c->uploaded_n || !c->uploaded_np1) return fail(c, SPH_ESTATE, "both states must be uploaded before stepping");
    HIPCHK(hipSetDevice(c->device));
    for (int s = 0; s < nsteps; ++s) CHK(one_step(c, 0, false));
    if (nsteps == 0) CHK(read_scalars(c));
    fill_scalars(c, out);
    return SPH_OK;
}

int sph_sweeps(sph_ctx* c, int32_t nsweeps, sph_scalars* out) {
    if (!c || nsweeps < 1 || nsweeps > c->prm.subits) return SPH_EINVAL;
    if (!c->uploaded_n || !c->uploaded_np1) return fail(c, SPH_ESTATE, "both states must be uploaded before stepping");
    HIPCHK(hipSetDevice(c->device));
    CHK(one_step(c, nsweeps, true));
    fill_scalars(c, out);
    return SPH_OK;
}

int sph_get_scalars(sph_ctx* c, sph_scalars* out) {
    if (!c || !out) return SPH_EINVAL;
    HIPCHK(hipSetDevice(c->device));
    CHK(read_scalars(c));
    fill_scalars(c, out);
    return SPH_OK;
}

int sph_set_dt(sph_ctx* c, double dt) {
    if (!c || !(dt > 0)) return SPH_EINVAL;
    HIPCHK(hipSetDevice(c->device));
    const double v[2] = {dt, dt};
    HIPCHK(hipMemcpyAsync(&c->S->dt, v, sizeof v, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SPH_OK;
}

int sph_density_reinit(sph_ctx* c) {
    if (!c) return SPH_EINVAL;
    return fail(c, SPH_ESTATE, "density re-initialisation is not built yet");
}

int sph_neighbours(sph_ctx* c, int32_t which, uint64_t* row_offsets, uint32_t* cols) {
    if (!c || !row_offsets) return SPH_EINVAL;
    if (which != SPH_STATE_N && which != SPH_STATE_NP1) return fail(c, SPH_EINVAL, "bad state selector");
    if (which == SPH_STATE_N ? !c->uploaded_n : !c->uploaded_np1) return fail(c, SPH_ESTATE, "state was never uploaded");
    HIPCHK(hipSetDevice(c->device));
    int src = c->b0;
    if (which == SPH_STATE_N) {
        src = (c->b0 + 1) % NREC;
        CHK(pack_n_into(c, c->rec[src]));
    }
    CHK(rebuild(c, src));
    if (which == SPH_STATE_N) {
        src = (c->b0 + 1) % NREC;
        c->list_valid = false;
    }
    const uint32_t n = c->n;
    uint32_t* counts = (uint32_t*)c->tmp;
    if (c->dim == 3)
        hipLaunchKernelGGL(k_nbr_count<3>, dim3(c->nblocks), dim3(SPH_BLOCK), 0, c->stream, c->dp, c->rec[src], c->nbr,
                           c->cap_k, c->cnt, c->id, counts);
    else
        hipLaunchKernelGGL(k_nbr_count<2>, dim3(c->nblocks), dim3(SPH_BLOCK), 0, c->stream, c->dp, c->rec[src], c->nbr,
                           c->cap_k, c->cnt, c->id, counts);
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
    do {
        if (hipMemcpyAsync(doffs, row_offsets, sizeof(uint64_t) * ((size_t)n + 1), hipMemcpyHostToDevice, c->stream) !=
            hipSuccess) { rc = SPH_EDEVICE; break; }
        if (c->dim == 3)
            hipLaunchKernelGGL(k_nbr_fill<3>, dim3(c->nblocks), dim3(SPH_BLOCK), 0, c->stream, c->dp, c->rec[src],
                               c->nbr, c->cap_k, c->cnt, c->id, doffs, dcols);
        else
            hipLaunchKernelGGL(k_nbr_fill<2>, dim3(c->nblocks), dim3(SPH_BLOCK), 0, c->stream, c->dp, c->rec[src],
                               c->nbr, c->cap_k, c->cnt, c->id, doffs, dcols);
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
    return SPH_OK;
}

}  /* extern "C" */
