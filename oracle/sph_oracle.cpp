/*
 * CPU oracle for the SPH particle-pair path -- TEST INFRASTRUCTURE ONLY (see sph_oracle.h).
 * PARITY UNPINNED: no golden vectors exist in the reference; see the header.
 *
 * Every function cites the reference lines it restates.  Storage precisions follow the
 * reference's Particle struct (WCXSPH3D.cpp:81-92): xi, v, V, f are double, rho, p,
 * Rrho, m are float.  Expressions are written in the reference's operation order and
 * the file is compiled with -std=c++11 -O3 -ffp-contract=off and no -march flag, the
 * floating-point model of the reference's makefile (makefile:3).
 *
 * Third-party behaviour restated here (libraries absent from the reference tree):
 *  - nanoflann (version not pinned by the reference) radiusSearch: all points with
 *    squared L2 distance strictly below the given radius argument, the distance
 *    accumulated one dimension at a time, results sorted ascending by distance.  Ties in
 *    distance are broken here by index (nanoflann's own tie order is unspecified).
 *  - Eigen3 (version not pinned) fixed-size vectors, and FullPivLU for DensityReinit:
 *    complete pivoting on the largest absolute coefficient (first in column-major order),
 *    rank threshold epsilon*size*|max pivot|.
 */
#include "sph_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#ifndef M_PI
#define M_PI (4.0 * atan(1.0))
#endif

namespace {

/* WCXSPH3D.cpp:81-92 (2-D files: two-component vectors; WCSPH has no V and no m, the
 * oracle keeps them and fills m with the single global mass). */
struct Particle {
    double xi[3], v[3], V[3], f[3];
    float rho, p, Rrho, m;
};

typedef std::vector<Particle> State;

}  // namespace

struct orc_sim {
    orc_case cs;
    orc_consts k;
    State particles;   /* state n   (main: `particles`)  */
    State particlesh;  /* state n+1 (main: `particlesh`) */
    /* file-scope globals of the reference (WCXSPH3D.cpp:70-79) */
    double dt;
    float t;
    double maxmu, maxf, errsum, logbase;
    int search;
};

/* ------------------------------------------------------------------------------------------ */
/* constants                                                                                  */

void orc_default_case(int32_t variant, orc_case* c) {
    memset(c, 0, sizeof(*c));
    c->variant = variant;
    c->pstep = 0.1;
    c->dt0 = 0.0002;
    if (variant == ORC_WCSPH) { /* WCSPH.cpp:42-50 */
        c->npart[0] = 25; c->npart[1] = 38; c->npart[2] = 1;
        c->box[0] = 15.0; c->box[1] = 8.0;
        c->start[0] = 0.31; c->start[1] = 0.31;
    } else if (variant == ORC_WCXSPH) { /* WCXSPH.cpp:36-44 */
        c->npart[0] = 40; c->npart[1] = 48; c->npart[2] = 1;
        c->box[0] = 15; c->box[1] = 5;
        c->start[0] = 0.11; c->start[1] = 0.11;
    } else { /* WCXSPH3D.cpp:39-47 */
        c->npart[0] = 5; c->npart[1] = 5; c->npart[2] = 5;
        c->box[0] = 1; c->box[1] = 1; c->box[2] = 1;
        c->start[0] = 0.15; c->start[1] = 0.15; c->start[2] = 0.15;
    }
}

/* Boundary lattice sizes: ceil(Box/step) with step = factor*r0 (WCXSPH3D.cpp:319-328,
 * WCXSPH.cpp:314-319, WCSPH.cpp:312-317). */
static void wall_grid(const orc_case* c, int dim, int* N, double* step) {
    const double r0 = c->pstep;
    const double fac = (c->variant == ORC_WCSPH) ? 0.7 : 0.5;
    for (int d = 0; d < dim; ++d) {
        double s = r0 * fac;
        N[d] = (int)ceil(c->box[d] / s);
        step[d] = c->box[d] / N[d];
    }
}

static uint64_t count_wall(const orc_case* c, int dim) {
    int N[3];
    double st[3];
    wall_grid(c, dim, N, st);
    if (dim == 2) /* perimeter walk, WCXSPH.cpp:321-336 */
        return (uint64_t)(N[1] + 1) + (uint64_t)(N[0] - 1) + (uint64_t)N[1] + (uint64_t)N[0];
    /* WCXSPH3D.cpp:329-360 */
    uint64_t n = 2ull * (uint64_t)(N[1] + 1) * (uint64_t)(N[2] + 1);
    n += (uint64_t)(N[0] - 1) * (uint64_t)(N[1] - 1);
    n += 2ull * (uint64_t)(N[0] - 1) * (uint64_t)(N[2] + 1);
    return n;
}

/* the added body (simtype 1, 2): visit(x) for each lattice point, i slowest */
template <typename Visit>
static void visit_drop(const orc_case* c, int dim, Visit visit) {
    if (c->simtype != 1 && c->simtype != 2) return;
    const int m = (int)floor(c->drop_radius / c->pstep);
    const int mh = c->simtype == 2 ? (int)floor(0.5 * c->drop_height / c->pstep) : m;
    const int last = dim - 1;
    int lim[3] = {m, m, m};
    lim[last] = mh;
    if (dim == 2) lim[2] = 0;
    for (int i = -lim[0]; i <= lim[0]; ++i)
        for (int j = -lim[1]; j <= lim[1]; ++j)
            for (int kk = -lim[2]; kk <= lim[2]; ++kk) {
                const int idx[3] = {i, j, kk};
                double r2 = 0.0;
                for (int d = 0; d < dim; ++d) {
                    if (c->simtype == 2 && d == last) continue;
                    const double dd = idx[d] * c->pstep;
                    r2 += dd * dd;
                }
                if (r2 > c->drop_radius * c->drop_radius) continue;
                double x[3] = {0, 0, 0};
                for (int d = 0; d < dim; ++d) x[d] = c->drop_centre[d] + idx[d] * c->pstep;
                visit(x);
            }
}

int orc_derive(const orc_case* c, orc_consts* k) {
    memset(k, 0, sizeof(*k));
    if (c->variant < ORC_WCSPH || c->variant > ORC_WCXSPH3D) return 1;
    const int dim = (c->variant == ORC_WCXSPH3D) ? 3 : 2;
    k->dim = dim;
    /* WCXSPH3D.cpp:40,48 */
    k->sim_pts = 1;
    double span[3];
    for (int d = 0; d < dim; ++d) {
        if (c->npart[d] <= 0) return 2;
        k->sim_pts *= (uint64_t)c->npart[d];
        const double finish = c->start[d] + c->pstep * c->npart[d] * 1.0;
        span[d] = finish - c->start[d];
    }
    k->rho0 = c->rho0 > 0 ? c->rho0 : 1000.0;
    /* WCXSPH3D.cpp:52-53 / WCXSPH.cpp:49-50 / WCSPH.cpp:55 */
    double vol = span[0] * span[1];
    if (dim == 3) vol = vol * span[2];
    k->simmass = k->rho0 * (vol / (1.0 * k->sim_pts));
    if (c->variant == ORC_WCXSPH3D) k->boundmass = 1.0 * k->rho0 * (vol / (1.0 * k->sim_pts));
    else if (c->variant == ORC_WCXSPH) k->boundmass = 0.9 * k->rho0 * (vol / (1.0 * k->sim_pts));
    else k->boundmass = k->simmass;
    {   /* the added body shares the block's spacing and mass; it only adds points */
        uint64_t extra = 0;
        visit_drop(c, dim, [&](const double*) { ++extra; });
        k->sim_pts += extra;
    }
    k->Cs = c->Cs > 0 ? c->Cs : 50;
    k->gam = 7.0;
    k->B = k->rho0 * pow(k->Cs, 2) / k->gam; /* WCXSPH3D.cpp:56 */
    /* WCXSPH3D.cpp:59-63 / WCXSPH.cpp:56-57,120 */
    if (dim == 3) k->H = 3 * sqrt((3 / M_PI) * (k->simmass / k->rho0));
    else k->H = sqrt((3 / M_PI) * (k->simmass / k->rho0));
    k->HSQ = k->H * k->H;
    if (dim == 3) {
        const double HCB = k->HSQ * k->H;
        k->correc = 21 / (16 * M_PI * HCB);
    } else {
        k->correc = 7 / (4 * M_PI * k->HSQ);
    }
    /* WCXSPH3D.cpp:97 / WCXSPH.cpp:92 / WCSPH.cpp:96 */
    if (c->variant == ORC_WCSPH) k->search_radius = 4 * k->HSQ + 0.001 * k->H;
    else k->search_radius = 4 * k->HSQ;
    k->alpha = c->alpha > 0 ? c->alpha : 0.025; /* WCXSPH3D.cpp:142 */
    k->eps = 0.2;     /* WCXSPH3D.cpp:143 */
    const float beta = 0.25;
    k->beta = beta;
    k->subits = 8;
    k->g_axis = dim - 1; /* f(1) in 2-D, f(2) in 3-D: WCXSPH.cpp:191, WCXSPH3D.cpp:195 */
    k->g = c->gravity > 0 ? -c->gravity : -9.81;
    k->bound_parts = count_wall(c, dim);
    k->npts = k->bound_parts + k->sim_pts;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* InitSPH                                                                                    */

static Particle make_particle(const double* x, int dim, float rho, float Rrho, float m) {
    Particle P;
    memset(&P, 0, sizeof(P));
    for (int d = 0; d < dim; ++d) P.xi[d] = x[d];
    P.rho = rho;
    P.p = 0.0;
    P.Rrho = Rrho;
    P.m = m;
    return P;
}

/* WCXSPH3D.cpp:306-379, WCXSPH.cpp:301-350, WCSPH.cpp:297-350 */
static void init_sph(const orc_case* c, const orc_consts* k, State& ps) {
    const int dim = k->dim;
    const float rho = (float)k->rho0;
    const float Rrho = 0.0;
    const float bm = (float)k->boundmass;
    const float sm = (float)k->simmass;
    int N[3];
    double st[3];
    wall_grid(c, dim, N, st);
    ps.clear();
    ps.reserve(k->npts);
    if (dim == 2) {
        const int Nx = N[0], Ny = N[1];
        const double stepx = st[0], stepy = st[1];
        for (int i = 0; i <= Ny; ++i) {
            double x[2] = {0.0, i * stepy};
            ps.push_back(make_particle(x, 2, rho, Rrho, bm));
        }
        for (int i = 1; i < Nx; ++i) {
            double x[2] = {i * stepx, c->box[1]};
            ps.push_back(make_particle(x, 2, rho, Rrho, bm));
        }
        for (int i = Ny; i > 0; --i) {
            double x[2] = {c->box[0], i * stepy};
            ps.push_back(make_particle(x, 2, rho, Rrho, bm));
        }
        for (int i = Nx; i > 0; --i) {
            double x[2] = {i * stepx, 0.0};
            ps.push_back(make_particle(x, 2, rho, Rrho, bm));
        }
        for (int i = 0; i < c->npart[0]; ++i)
            for (int j = 0; j < c->npart[1]; ++j) {
                double x[2] = {c->start[0] + i * c->pstep, c->start[1] + j * c->pstep};
                ps.push_back(make_particle(x, 2, rho, Rrho, sm));
            }
    } else {
        const int Nx = N[0], Ny = N[1], Nz = N[2];
        const double stepx = st[0], stepy = st[1], stepz = st[2];
        for (int j = 0; j <= Ny; ++j)
            for (int kk = 0; kk <= Nz; ++kk) {
                double x[3] = {0.0, j * stepy, kk * stepz};
                ps.push_back(make_particle(x, 3, rho, Rrho, bm));
                x[0] = c->box[0];
                ps.push_back(make_particle(x, 3, rho, Rrho, bm));
            }
        for (int i = 1; i < Nx; ++i)
            for (int j = 1; j < Ny; ++j) {
                double x[3] = {i * stepx, j * stepy, 0};
                ps.push_back(make_particle(x, 3, rho, Rrho, bm));
            }
        for (int i = 1; i < Nx; ++i)
            for (int kk = 0; kk <= Nz; ++kk) {
                double x[3] = {i * stepx, 0, kk * stepz};
                ps.push_back(make_particle(x, 3, rho, Rrho, bm));
                x[1] = c->box[1];
                ps.push_back(make_particle(x, 3, rho, Rrho, bm));
            }
        for (int i = 0; i < c->npart[0]; ++i)
            for (int j = 0; j < c->npart[1]; ++j)
                for (int kk = 0; kk < c->npart[2]; ++kk) {
                    double x[3] = {c->start[0] + i * c->pstep, c->start[1] + j * c->pstep,
                                   c->start[2] + kk * c->pstep};
                    ps.push_back(make_particle(x, 3, rho, Rrho, sm));
                }
    }
    visit_drop(c, dim, [&](const double* x) {
        Particle P = make_particle(x, dim, rho, Rrho, sm);
        for (int d = 0; d < dim; ++d) P.v[d] = c->drop_v[d];
        ps.push_back(P);
    });
}

/* ------------------------------------------------------------------------------------------ */
/* neighbour search (restated nanoflann radiusSearch semantics)                               */

namespace {

/* Squared distance as nanoflann's L2 adaptor accumulates it for dim <= 3:
 * result += (a[d]-b[d])^2, d ascending, starting from 0. */
inline double dist2(const double* a, const double* b, int dim) {
    double r = 0.0;
    for (int d = 0; d < dim; ++d) {
        const double diff = a[d] - b[d];
        r += diff * diff;
    }
    return r;
}

typedef std::vector<std::pair<size_t, double> > Matches;

struct ByDistThenIndex {
    bool operator()(const std::pair<size_t, double>& a, const std::pair<size_t, double>& b) const {
        if (a.second != b.second) return a.second < b.second;
        return a.first < b.first;
    }
};

/* Uniform-grid index over one State's positions; any cell edge >= sqrt(radius) finds
 * every point the brute-force scan finds, the acceptance test itself is identical. */
struct CellIndex {
    int dim;
    double lo[3], inv;
    int n[3];
    std::vector<uint32_t> start, items;

    void build(const State& ps, int dim_, double radius_sq) {
        dim = dim_;
        const double edge = sqrt(radius_sq) * 1.0000001;
        inv = 1.0 / edge;
        double hi[3];
        for (int d = 0; d < 3; ++d) { lo[d] = 0; hi[d] = 0; n[d] = 1; }
        bool first = true;
        for (size_t i = 0; i < ps.size(); ++i) {
            bool ok = true;
            for (int d = 0; d < dim; ++d) ok = ok && std::isfinite(ps[i].xi[d]);
            if (!ok) continue;
            for (int d = 0; d < dim; ++d) {
                if (first || ps[i].xi[d] < lo[d]) lo[d] = ps[i].xi[d];
                if (first || ps[i].xi[d] > hi[d]) hi[d] = ps[i].xi[d];
            }
            first = false;
        }
        size_t total = 1;
        for (int d = 0; d < dim; ++d) {
            double cells = floor((hi[d] - lo[d]) * inv) + 1;
            if (cells > 4096) cells = 4096; /* far-away strays are clamped into edge cells */
            n[d] = (int)cells;
            total *= (size_t)n[d];
        }
        std::vector<uint32_t> cnt(total + 1, 0);
        std::vector<uint32_t> cell(ps.size());
        for (size_t i = 0; i < ps.size(); ++i) {
            cell[i] = (uint32_t)cell_of(ps[i].xi);
            cnt[cell[i] + 1]++;
        }
        for (size_t c = 0; c < total; ++c) cnt[c + 1] += cnt[c];
        start = cnt;
        items.resize(ps.size());
        std::vector<uint32_t> fill(start.begin(), start.end() - 1);
        for (size_t i = 0; i < ps.size(); ++i) items[fill[cell[i]]++] = (uint32_t)i;
    }
    int coord(double x, int d) const {
        double c = floor((x - lo[d]) * inv);
        if (!(c > 0)) c = 0; /* also catches NaN */
        if (c > n[d] - 1) c = n[d] - 1;
        return (int)c;
    }
    size_t cell_of(const double* x) const {
        size_t c = 0;
        for (int d = dim - 1; d >= 0; --d) c = c * (size_t)n[d] + (size_t)coord(x[d], d);
        return c;
    }
};

void radius_search_brute(const State& ps, int dim, const double* q, double radius, Matches& out) {
    out.clear();
    for (size_t j = 0; j < ps.size(); ++j) {
        const double d2 = dist2(q, ps[j].xi, dim);
        if (d2 < radius) out.push_back(std::make_pair(j, d2));
    }
    std::sort(out.begin(), out.end(), ByDistThenIndex());
}

void radius_search_cells(const State& ps, const CellIndex& ci, const double* q, double radius,
                         Matches& out) {
    out.clear();
    const int dim = ci.dim;
    int c0[3] = {0, 0, 0};
    for (int d = 0; d < dim; ++d) c0[d] = ci.coord(q[d], d);
    const int zlo = dim == 3 ? std::max(c0[2] - 1, 0) : 0;
    const int zhi = dim == 3 ? std::min(c0[2] + 1, ci.n[2] - 1) : 0;
    for (int cz = zlo; cz <= zhi; ++cz)
        for (int cy = std::max(c0[1] - 1, 0); cy <= std::min(c0[1] + 1, ci.n[1] - 1); ++cy)
            for (int cx = std::max(c0[0] - 1, 0); cx <= std::min(c0[0] + 1, ci.n[0] - 1); ++cx) {
                const size_t c = ((size_t)cz * ci.n[1] + cy) * ci.n[0] + cx;
                for (uint32_t s = ci.start[c]; s < ci.start[c + 1]; ++s) {
                    const size_t j = ci.items[s];
                    const double d2 = dist2(q, ps[j].xi, dim);
                    if (d2 < radius) out.push_back(std::make_pair(j, d2));
                }
            }
    std::sort(out.begin(), out.end(), ByDistThenIndex());
}

/* The kd-tree stand-in: rebuilt wherever the reference calls buildIndex(). */
struct Searcher {
    const State* ps;
    int dim, method;
    double radius;
    CellIndex ci;
    void build(const State& s, int dim_, int method_, double radius_) {
        ps = &s; dim = dim_; method = method_; radius = radius_;
        if (method == ORC_SEARCH_CELLS) ci.build(s, dim, radius);
    }
    void search(const double* q, Matches& out) const {
        if (method == ORC_SEARCH_CELLS) radius_search_cells(*ps, ci, q, radius, out);
        else radius_search_brute(*ps, dim, q, radius, out);
    }
};

/* ------------------------------------------------------------------------------------------ */
/* Wendland C2 kernel                                                                         */

/* WCXSPH3D.cpp:121-128 (2-D: WCXSPH.cpp:116-123 with the factor written inline) */
inline double W2Kernel(const orc_consts& k, double dist) {
    const double q = dist / k.H;
    if (q < 2) return (pow(1 - 0.5 * q, 4)) * (2 * q + 1) * k.correc;
    return 0;
}

/* WCXSPH3D.cpp:130-137 */
inline void W2GradK(const orc_consts& k, const double* Rij, double dist, double* out) {
    const double q = dist / k.H;
    if (q < 2) {
        const double p3 = pow(1 - 0.5 * q, 3);
        for (int d = 0; d < k.dim; ++d) out[d] = 5.0 * (Rij[d] / k.HSQ) * p3 * k.correc;
    } else {
        for (int d = 0; d < k.dim; ++d) out[d] = 0.0;
    }
}

inline double norm(const double* a, int dim) {
    double s = a[0] * a[0] + a[1] * a[1];
    if (dim == 3) s = s + a[2] * a[2];
    return sqrt(s);
}

inline double dot(const double* a, const double* b, int dim) {
    double s = a[0] * b[0] + a[1] * b[1];
    if (dim == 3) s = s + a[2] * b[2];
    return s;
}

}  // namespace

/* ------------------------------------------------------------------------------------------ */
/* Forces                                                                                     */

/* WCXSPH3D.cpp:139-205 / WCXSPH.cpp:134-201 / WCSPH.cpp:137-200 */
static void Forces(orc_sim* S, State& part, const Searcher& tree) {
    const orc_consts& k = S->k;
    const int dim = k.dim;
    const bool xsph = S->cs.variant != ORC_WCSPH;
    const double B = k.B, gam = k.gam, H = k.H, HSQ = k.HSQ;
    const double mass = k.simmass;
    S->maxmu = 0;
    const double alpha = k.alpha;
    const double eps = k.eps;
    Matches matches;
    for (size_t ii = 0; ii < part.size(); ++ii) {
        Particle& pi = part[ii];
        for (int d = 0; d < 3; ++d) pi.f[d] = 0.0;
        if (xsph)
            for (int d = 0; d < 3; ++d) pi.V[d] = pi.v[d];
        double contrib[3] = {0.0, 0.0, 0.0};
        double Rrhocontr = 0.0;
        pi.Rrho = 0.0;
        double mumax = 0; /* max over the mu vector, which starts as {0} */
        tree.search(pi.xi, matches);
        for (size_t mi = 0; mi < matches.size(); ++mi) {
            const Particle pj = part[matches[mi].first];
            double Rij[3] = {0, 0, 0}, Vij[3] = {0, 0, 0}, Grad[3] = {0, 0, 0};
            for (int d = 0; d < dim; ++d) {
                Rij[d] = pj.xi[d] - pi.xi[d];
                Vij[d] = pj.v[d] - pi.v[d];
            }
            const double r = norm(Rij, dim);
            W2GradK(k, Rij, r, Grad);
            const double cbar = 0.5 * (sqrt((B * gam) / pi.rho) + sqrt((B * gam) / pj.rho));
            const double vdotr = dot(Vij, Rij, dim);
            const double rhoij = 0.5 * (pi.rho + pj.rho);
            const double muij = H * vdotr / (r * r + 0.01 * HSQ);
            if (mumax < muij) mumax = muij;
            if (xsph) {
                const double Kern = W2Kernel(k, r);
                double pifac = alpha * cbar * muij / rhoij;
                if (vdotr > 0) pifac = 0;
                const double sc = pifac - pi.p / pow(pi.rho, 2) - pj.p / pow(pj.rho, 2);
                for (int d = 0; d < dim; ++d) contrib[d] += pj.m * Grad[d] * sc;
                const double xs = eps * (pj.m / rhoij) * Kern;
                for (int d = 0; d < dim; ++d) pi.V[d] += xs * Vij[d];
                Rrhocontr -= pj.m * (dot(Vij, Grad, dim));
            } else {
                double pifac = -alpha * cbar * muij / rhoij;
                if (vdotr > 0) pifac = 0;
                const double sc = pifac + pi.p / pow(pi.rho, 2) + pj.p / pow(pj.rho, 2);
                for (int d = 0; d < dim; ++d) contrib[d] -= Grad[d] * sc;
                Rrhocontr -= (dot(Vij, Grad, dim));
            }
        }
        if (xsph) {
            pi.Rrho = Rrhocontr;
            for (int d = 0; d < dim; ++d) pi.f[d] = contrib[d];
        } else {
            pi.Rrho = Rrhocontr * mass;
            for (int d = 0; d < dim; ++d) pi.f[d] = contrib[d] * mass;
        }
        pi.f[k.g_axis] += k.g;
        if (mumax > S->maxmu) S->maxmu = mumax;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* DensityReinit                                                                              */

namespace {

/* Eigen FullPivLU restated for an n x n matrix (n = 3 or 4): complete pivoting, in-place
 * LU, rank with the default threshold, then solve A*beta = e0.  Returns false when the
 * matrix is not invertible by that rank test. */
bool fullpiv_solve_e0(const double* Ain, int n, double* beta) {
    double lu[16];
    for (int i = 0; i < n * n; ++i) lu[i] = Ain[i]; /* row-major lu[r*n+c] */
    int rowT[4], colT[4];
    int nonzero = n;
    double maxpivot = 0.0;
    for (int kk = 0; kk < n; ++kk) {
        int br = kk, bc = kk;
        double biggest = -1.0;
        for (int c = kk; c < n; ++c)       /* column-major visit, strict > keeps the first */
            for (int r = kk; r < n; ++r) {
                const double a = fabs(lu[r * n + c]);
                if (a > biggest) { biggest = a; br = r; bc = c; }
            }
        if (biggest == 0.0) {
            nonzero = kk;
            for (int i = kk; i < n; ++i) { rowT[i] = i; colT[i] = i; }
            break;
        }
        if (biggest > maxpivot) maxpivot = biggest;
        rowT[kk] = br;
        colT[kk] = bc;
        if (kk != br)
            for (int c = 0; c < n; ++c) std::swap(lu[kk * n + c], lu[br * n + c]);
        if (kk != bc)
            for (int r = 0; r < n; ++r) std::swap(lu[r * n + kk], lu[r * n + bc]);
        if (kk < n - 1)
            for (int r = kk + 1; r < n; ++r) lu[r * n + kk] /= lu[kk * n + kk];
        if (kk < n - 1)
            for (int r = kk + 1; r < n; ++r)
                for (int c = kk + 1; c < n; ++c) lu[r * n + c] -= lu[r * n + kk] * lu[kk * n + c];
    }
    const double eps = 2.220446049250313e-16;
    const double thr = fabs(maxpivot) * (eps * n);
    int rank = 0;
    for (int i = 0; i < nonzero; ++i)
        if (fabs(lu[i * n + i]) > thr) ++rank;
    if (rank != n) return false;
    /* c = P * e0 */
    int perm[4];
    for (int i = 0; i < n; ++i) perm[i] = i;
    /* P built from the row transpositions: row k of (P*b) after applying swaps in order */
    double c[4];
    for (int i = 0; i < n; ++i) c[i] = (i == 0) ? 1.0 : 0.0;
    for (int kk = 0; kk < n; ++kk) std::swap(c[kk], c[rowT[kk]]);
    /* unit-lower forward substitution */
    for (int r = 0; r < n; ++r)
        for (int cc = 0; cc < r; ++cc) c[r] -= lu[r * n + cc] * c[cc];
    /* upper back substitution */
    for (int r = n - 1; r >= 0; --r) {
        for (int cc = r + 1; cc < n; ++cc) c[r] -= lu[r * n + cc] * c[cc];
        c[r] /= lu[r * n + r];
    }
    /* undo the column transpositions: x = Q * c */
    for (int kk = n - 1; kk >= 0; --kk) std::swap(c[kk], c[colT[kk]]);
    for (int i = 0; i < n; ++i) beta[i] = c[i];
    (void)perm;
    return true;
}

}  // namespace

/* WCXSPH3D.cpp:208-249 / WCXSPH.cpp:204-243 / WCSPH.cpp:203-243.  Gauss-Seidel: rho is
 * overwritten in place and later particles read the updated values. */
static void DensityReinit(orc_sim* S, State& p, const Searcher& tree) {
    const orc_consts& k = S->k;
    const int dim = k.dim;
    const int n = dim + 1;
    const bool single_mass = S->cs.variant == ORC_WCSPH;
    const double mass = k.simmass;
    Matches matches;
    for (size_t ii = 0; ii < p.size(); ++ii) {
        Particle& pi = p[ii];
        double A[16];
        for (int i = 0; i < 16; ++i) A[i] = 0.0;
        tree.search(pi.xi, matches);
        for (size_t mi = 0; mi < matches.size(); ++mi) {
            const Particle pj = p[matches[mi].first];
            double Rij[3] = {0, 0, 0};
            for (int d = 0; d < dim; ++d) Rij[d] = pi.xi[d] - pj.xi[d];
            double e[4];
            e[0] = 1;
            for (int d = 0; d < dim; ++d) e[d + 1] = Rij[d];
            const double W = W2Kernel(k, norm(Rij, dim));
            for (int r = 0; r < n; ++r)
                for (int c = 0; c < n; ++c) {
                    double abar;
                    if (r == 0) abar = e[c];
                    else if (c == 0) abar = e[r];
                    else abar = e[c] * e[r];
                    if (single_mass) A[r * n + c] += W * abar * mass / pj.rho;
                    else A[r * n + c] += W * abar * pj.m / pj.rho;
                }
        }
        double Beta[4] = {0, 0, 0, 0};
        if (!fullpiv_solve_e0(A, n, Beta)) {
            Beta[0] = (1 / A[0]);
            for (int i = 1; i < n; ++i) Beta[i] = (1 / A[0]) * 0.0;
        }
        double rho = 0.0;
        for (size_t mi = 0; mi < matches.size(); ++mi) {
            const Particle& pj = p[matches[mi].first];
            double Rij[3] = {0, 0, 0};
            for (int d = 0; d < dim; ++d) Rij[d] = pi.xi[d] - pj.xi[d];
            double poly = Beta[0] + Beta[1] * Rij[0] + Beta[2] * Rij[1];
            if (dim == 3) poly = poly + Beta[3] * Rij[2];
            if (single_mass) rho += mass * W2Kernel(k, norm(Rij, dim)) * poly;
            else rho += pj.m * W2Kernel(k, norm(Rij, dim)) * poly;
        }
        pi.rho = rho;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Newmark_Beta                                                                               */

/* WCXSPH3D.cpp:251-303 / WCXSPH.cpp:245-298 / WCSPH.cpp:245-296.  The Searcher plays the
 * kd-tree over pnp1: it is rebuilt where the reference calls buildIndex(). */
static void Newmark_Beta(orc_sim* S, State& pn, State& pnp1, Searcher& tree, int max_sweeps) {
    const orc_consts& k = S->k;
    const int dim = k.dim;
    const int variant = S->cs.variant;
    const float beta = 0.25;
    const double B = k.B, rho0 = k.rho0, gam = k.gam, H = k.H, Cs = k.Cs;
    const size_t bound_parts = k.bound_parts;
    double& dt = S->dt;
    std::vector<double> xih(pnp1.size() * 3);
    for (size_t i = 0; i < pnp1.size(); ++i)
        for (int d = 0; d < 3; ++d) xih[i * 3 + d] = pnp1[i].xi[d];

    for (int kk = 0; kk < k.subits; ++kk) {
        if (max_sweeps >= 0 && kk >= max_sweeps) return;
        Forces(S, pnp1, tree);
        for (size_t i = 0; i < bound_parts; ++i) {
            for (int d = 0; d < 3; ++d) pnp1[i].f[d] = 0.0;
            if (variant != ORC_WCXSPH3D)
                for (int d = 0; d < 3; ++d) pnp1[i].v[d] = 0.0;
            if (variant != ORC_WCSPH)
                for (int d = 0; d < 3; ++d) pnp1[i].V[d] = 0.0;
        }
        for (size_t i = 0; i < pnp1.size(); ++i)
            for (int d = 0; d < 3; ++d) xih[i * 3 + d] = pnp1[i].xi[d];

        for (size_t i = 0; i < pn.size(); ++i) {
            const double* adv = (variant == ORC_WCSPH) ? pn[i].v : pn[i].V;
            for (int d = 0; d < dim; ++d)
                pnp1[i].v[d] = pn[i].v[d] + 0.5 * dt * (pn[i].f[d] + pnp1[i].f[d]);
            pnp1[i].rho = pn[i].rho + 0.5 * dt * (pn[i].Rrho + pnp1[i].Rrho);
            for (int d = 0; d < dim; ++d)
                pnp1[i].xi[d] = pn[i].xi[d] + dt * adv[d] + 0.5 * (dt * dt) * (1 - 2 * beta) * pn[i].f[d] +
                                (dt * dt * beta) * pnp1[i].f[d];
            pnp1[i].p = B * (pow(pnp1[i].rho / rho0, gam) - 1);
        }
        tree.build(pnp1, dim, S->search, k.search_radius);

        S->errsum = 0.0;
        for (size_t i = 0; i < pnp1.size(); ++i) {
            double r[3] = {0, 0, 0};
            for (int d = 0; d < dim; ++d) r[d] = pnp1[i].xi[d] - xih[i * 3 + d];
            double sq = r[0] * r[0] + r[1] * r[1];
            if (dim == 3) sq = sq + r[2] * r[2];
            S->errsum += sq;
        }
        if (kk == 0) S->logbase = log10(sqrt(S->errsum / (1.0 * k.npts)));
    }
    if (max_sweeps >= 0 && max_sweeps < k.subits) return;

    double best = -1.0;
    for (size_t i = 0; i < pnp1.size(); ++i) {
        const double nf = norm(pnp1[i].f, dim);
        if (i == 0 || best < nf) best = nf; /* std::max_element keeps the first maximum */
    }
    S->maxf = best;
    const double dtf = sqrt(H / S->maxf);
    const double dtcv = H / (Cs + S->maxmu);
    dt = 0.3 * std::min(dtf, dtcv);

    pn = pnp1;
    S->t += dt;
}

/* ------------------------------------------------------------------------------------------ */
/* C API                                                                                      */

orc_sim* orc_create(const orc_case* c) {
    orc_consts k;
    if (orc_derive(c, &k) != 0) return 0;
    orc_sim* S = new orc_sim();
    S->cs = *c;
    S->k = k;
    init_sph(c, &k, S->particles);
    if (S->particles.size() != k.npts) { delete S; return 0; }
    S->particlesh = S->particles; /* WCXSPH3D.cpp:450-451 */
    S->dt = c->dt0;
    S->t = 0.f;
    S->maxmu = 0; S->maxf = 0; S->errsum = 0.0; S->logbase = 0.0;
    S->search = ORC_SEARCH_CELLS;
    return S;
}

void orc_destroy(orc_sim* s) { delete s; }
void orc_get_consts(const orc_sim* s, orc_consts* out) { *out = s->k; }
void orc_set_search(orc_sim* s, int32_t method) { s->search = method; }

static State& pick(orc_sim* s, int which) { return which == ORC_STATE_N ? s->particles : s->particlesh; }

void orc_get_state(const orc_sim* cs, int32_t which, double* x, double* v, double* V, double* f,
                   float* rho, float* p, float* Rrho, float* m) {
    orc_sim* s = const_cast<orc_sim*>(cs);
    const State& ps = pick(s, which);
    const int dim = s->k.dim;
    for (size_t i = 0; i < ps.size(); ++i) {
        for (int d = 0; d < dim; ++d) {
            if (x) x[i * dim + d] = ps[i].xi[d];
            if (v) v[i * dim + d] = ps[i].v[d];
            if (V) V[i * dim + d] = ps[i].V[d];
            if (f) f[i * dim + d] = ps[i].f[d];
        }
        if (rho) rho[i] = ps[i].rho;
        if (p) p[i] = ps[i].p;
        if (Rrho) Rrho[i] = ps[i].Rrho;
        if (m) m[i] = ps[i].m;
    }
}

void orc_set_state(orc_sim* s, int32_t which, const double* x, const double* v, const double* V,
                   const double* f, const float* rho, const float* p, const float* Rrho,
                   const float* m) {
    State& ps = pick(s, which);
    const int dim = s->k.dim;
    for (size_t i = 0; i < ps.size(); ++i) {
        for (int d = 0; d < dim; ++d) {
            if (x) ps[i].xi[d] = x[i * dim + d];
            if (v) ps[i].v[d] = v[i * dim + d];
            if (V) ps[i].V[d] = V[i * dim + d];
            if (f) ps[i].f[d] = f[i * dim + d];
        }
        if (rho) ps[i].rho = rho[i];
        if (p) ps[i].p = p[i];
        if (Rrho) ps[i].Rrho = Rrho[i];
        if (m) ps[i].m = m[i];
    }
}

void orc_forces(orc_sim* s, int32_t which) {
    State& ps = pick(s, which);
    Searcher tree;
    tree.build(ps, s->k.dim, s->search, s->k.search_radius);
    Forces(s, ps, tree);
}

void orc_zero_boundary_f(orc_sim* s, int32_t which) {
    State& ps = pick(s, which);
    for (size_t i = 0; i < s->k.bound_parts; ++i)
        for (int d = 0; d < 3; ++d) ps[i].f[d] = 0.0;
}

void orc_step(orc_sim* s, int32_t nsteps, int32_t max_sweeps) {
    Searcher tree;
    tree.build(s->particlesh, s->k.dim, s->search, s->k.search_radius);
    for (int i = 0; i < nsteps; ++i) {
        Newmark_Beta(s, s->particles, s->particlesh, tree, max_sweeps);
        if (max_sweeps >= 0 && max_sweeps < s->k.subits) return;
    }
}

void orc_density_reinit(orc_sim* s) {
    Searcher tree;
    tree.build(s->particlesh, s->k.dim, s->search, s->k.search_radius);
    DensityReinit(s, s->particlesh, tree);
}

void orc_get_scalars(const orc_sim* s, double* dt, float* t, double* maxmu, double* maxf,
                     double* errsum, double* logbase) {
    if (dt) *dt = s->dt;
    if (t) *t = s->t;
    if (maxmu) *maxmu = s->maxmu;
    if (maxf) *maxf = s->maxf;
    if (errsum) *errsum = s->errsum;
    if (logbase) *logbase = s->logbase;
}

void orc_set_dt(orc_sim* s, double dt) { s->dt = dt; }

void orc_neighbours(orc_sim* s, int32_t which, int32_t method, uint64_t* row_offsets, uint32_t* cols) {
    State& ps = pick(s, which);
    Searcher tree;
    tree.build(ps, s->k.dim, method, s->k.search_radius);
    Matches matches;
    std::vector<uint32_t> row;
    uint64_t off = 0;
    for (size_t i = 0; i < ps.size(); ++i) {
        row_offsets[i] = off;
        tree.search(ps[i].xi, matches);
        if (cols) {
            row.clear();
            for (size_t mi = 0; mi < matches.size(); ++mi) row.push_back((uint32_t)matches[mi].first);
            std::sort(row.begin(), row.end());
            for (size_t mi = 0; mi < row.size(); ++mi) cols[off + mi] = row[mi];
        }
        off += matches.size();
    }
    row_offsets[ps.size()] = off;
}

double orc_w2kernel(const orc_sim* s, double r) { return W2Kernel(s->k, r); }
void orc_w2gradk(const orc_sim* s, const double* rij, double r, double* grad_out) {
    W2GradK(s->k, rij, r, grad_out);
}
