/* Host-side case set-up: the reference's compile-time constants and its InitSPH lattices,
 * made run-time so that the same code serves the shipped cases and the scaled ones. */
#include "../../include/sph_amd.h"

#include <cmath>
#include <cstring>

#ifndef M_PI
#define M_PI (4.0 * atan(1.0))
#endif

namespace {

struct WallGrid {
    int n[3];
    double step[3];
};

int case_dim(const sph_case* c) { return c->program == SPH_PROGRAM_WCXSPH3D ? 3 : 2; }

/* Wall lattice: spacing 0.7*r0 (WCSPH.cpp:312-317) or 0.5*r0 (WCXSPH.cpp:314-319,
 * WCXSPH3D.cpp:319-328), rounded so that it divides the box. */
WallGrid wall_grid(const sph_case* c) {
    WallGrid w;
    const double frac = c->program == SPH_PROGRAM_WCSPH ? 0.7 : 0.5;
    for (int d = 0; d < 3; ++d) {
        w.n[d] = 0;
        w.step[d] = 0;
    }
    for (int d = 0; d < case_dim(c); ++d) {
        const double want = c->pstep * frac;
        w.n[d] = (int)ceil(c->box[d] / want);
        w.step[d] = c->box[d] / w.n[d];
    }
    return w;
}

/* Visits every particle of InitSPH in the reference's order: walls first, then the fluid
 * block.  emit(x, is_wall). */
/* the added droplet / jet: lattice points around drop_centre, first index slowest */
template <typename Emit>
void visit_added_body(const sph_case* c, Emit emit) {
    if (c->simtype != SPH_SIMTYPE_DROPLET && c->simtype != SPH_SIMTYPE_JET) return;
    const int dim = case_dim(c), last = dim - 1;
    const int m = (int)floor(c->drop_radius / c->pstep);
    int lim[3] = {m, m, m};
    if (c->simtype == SPH_SIMTYPE_JET) lim[last] = (int)floor(0.5 * c->drop_height / c->pstep);
    if (dim == 2) lim[2] = 0;
    for (int i = -lim[0]; i <= lim[0]; ++i)
        for (int j = -lim[1]; j <= lim[1]; ++j)
            for (int k = -lim[2]; k <= lim[2]; ++k) {
                const int idx[3] = {i, j, k};
                double r2 = 0.0;
                for (int d = 0; d < dim; ++d) {
                    if (c->simtype == SPH_SIMTYPE_JET && d == last) continue;
                    const double dd = idx[d] * c->pstep;
                    r2 += dd * dd;
                }
                if (r2 > c->drop_radius * c->drop_radius) continue;
                double x[3] = {0, 0, 0};
                for (int d = 0; d < dim; ++d) x[d] = c->drop_centre[d] + idx[d] * c->pstep;
                emit(x, false);
            }
}

/* emit(x, is_wall, in_added_body) */
template <typename Emit>
void visit_all(const sph_case* c, Emit emit);

template <typename Emit>
void visit_particles(const sph_case* c, Emit emit) {
    const WallGrid w = wall_grid(c);
    double x[3] = {0, 0, 0};
    if (case_dim(c) == 2) {
        /* perimeter walk: left side up, top, right side down, bottom (WCXSPH.cpp:321-336) */
        for (int i = 0; i <= w.n[1]; ++i) { x[0] = 0.0; x[1] = i * w.step[1]; emit(x, true); }
        for (int i = 1; i < w.n[0]; ++i) { x[0] = i * w.step[0]; x[1] = c->box[1]; emit(x, true); }
        for (int i = w.n[1]; i > 0; --i) { x[0] = c->box[0]; x[1] = i * w.step[1]; emit(x, true); }
        for (int i = w.n[0]; i > 0; --i) { x[0] = i * w.step[0]; x[1] = 0.0; emit(x, true); }
        for (int i = 0; i < c->npart[0]; ++i)
            for (int j = 0; j < c->npart[1]; ++j) {
                x[0] = c->start[0] + i * c->pstep;
                x[1] = c->start[1] + j * c->pstep;
                emit(x, false);
            }
        return;
    }
    /* open-topped box: x faces in pairs, floor, y faces in pairs (WCXSPH3D.cpp:329-360) */
    for (int j = 0; j <= w.n[1]; ++j)
        for (int k = 0; k <= w.n[2]; ++k) {
            x[1] = j * w.step[1];
            x[2] = k * w.step[2];
            x[0] = 0.0; emit(x, true);
            x[0] = c->box[0]; emit(x, true);
        }
    for (int i = 1; i < w.n[0]; ++i)
        for (int j = 1; j < w.n[1]; ++j) {
            x[0] = i * w.step[0];
            x[1] = j * w.step[1];
            x[2] = 0;
            emit(x, true);
        }
    for (int i = 1; i < w.n[0]; ++i)
        for (int k = 0; k <= w.n[2]; ++k) {
            x[0] = i * w.step[0];
            x[2] = k * w.step[2];
            x[1] = 0; emit(x, true);
            x[1] = c->box[1]; emit(x, true);
        }
    for (int i = 0; i < c->npart[0]; ++i)
        for (int j = 0; j < c->npart[1]; ++j)
            for (int k = 0; k < c->npart[2]; ++k) {
                x[0] = c->start[0] + i * c->pstep;
                x[1] = c->start[1] + j * c->pstep;
                x[2] = c->start[2] + k * c->pstep;
                emit(x, false);
            }
}

template <typename Emit>
void visit_all(const sph_case* c, Emit emit) {
    visit_particles(c, [&](const double* x, bool wall) { emit(x, wall, false); });
    visit_added_body(c, [&](const double* x, bool) { emit(x, false, true); });
}

uint64_t wall_count(const sph_case* c) {
    const WallGrid w = wall_grid(c);
    if (case_dim(c) == 2) return (uint64_t)(w.n[1] + 1) + (uint64_t)(w.n[0] - 1) + (uint64_t)w.n[1] + (uint64_t)w.n[0];
    return 2ull * (uint64_t)(w.n[1] + 1) * (uint64_t)(w.n[2] + 1) + (uint64_t)(w.n[0] - 1) * (uint64_t)(w.n[1] - 1) +
           2ull * (uint64_t)(w.n[0] - 1) * (uint64_t)(w.n[2] + 1);
}

}  // namespace

extern "C" {

int sph_case_default(int32_t program, sph_case* out) {
    if (!out) return SPH_EINVAL;
    memset(out, 0, sizeof *out);
    out->program = program;
    out->pstep = 0.1;   /* Pstep, all three programs */
    out->dt0 = 0.0002;  /* dt, all three programs */
    switch (program) {
        case SPH_PROGRAM_WCSPH: /* WCSPH.cpp:42,49-50 */
            out->npart[0] = 25; out->npart[1] = 38; out->npart[2] = 1;
            out->box[0] = 15.0; out->box[1] = 8.0;
            out->start[0] = 0.31; out->start[1] = 0.31;
            return SPH_OK;
        case SPH_PROGRAM_WCXSPH: /* WCXSPH.cpp:36,43-44 */
            out->npart[0] = 40; out->npart[1] = 48; out->npart[2] = 1;
            out->box[0] = 15; out->box[1] = 5;
            out->start[0] = 0.11; out->start[1] = 0.11;
            return SPH_OK;
        case SPH_PROGRAM_WCXSPH3D: /* WCXSPH3D.cpp:39,46-47 */
            for (int d = 0; d < 3; ++d) { out->npart[d] = 5; out->box[d] = 1; out->start[d] = 0.15; }
            return SPH_OK;
    }
    return SPH_EINVAL;
}

int sph_case_params(const sph_case* c, sph_params* p, double* simmass_out, double* boundmass_out) {
    if (!c || !p) return SPH_EINVAL;
    if (c->program < SPH_PROGRAM_WCSPH || c->program > SPH_PROGRAM_WCXSPH3D) return SPH_EINVAL;
    const int dim = case_dim(c);
    memset(p, 0, sizeof *p);
    uint64_t simpts = 1;
    double extent[3] = {0, 0, 0};
    for (int d = 0; d < dim; ++d) {
        if (c->npart[d] < 1 || !(c->box[d] > 0) || !(c->pstep > 0)) return SPH_EINVAL;
        simpts *= (uint64_t)c->npart[d];
        const double finish = c->start[d] + c->pstep * c->npart[d] * 1.0; /* Finish, WCXSPH3D.cpp:48 */
        extent[d] = finish - c->start[d];
    }
    const double rho0 = c->rho0 > 0 ? c->rho0 : 1000.0;
    double volume = extent[0] * extent[1];
    if (dim == 3) volume = volume * extent[2];
    /* Simmass / Boundmass (WCXSPH3D.cpp:52-53, WCXSPH.cpp:49-50), mass (WCSPH.cpp:55) */
    const double per_particle = volume / (1.0 * simpts);
    const double simmass = rho0 * per_particle;
    double boundmass = simmass;
    if (c->program == SPH_PROGRAM_WCXSPH) boundmass = 0.9 * rho0 * per_particle;
    if (c->program == SPH_PROGRAM_WCXSPH3D) boundmass = 1.0 * rho0 * per_particle;
    const double Cs = c->Cs > 0 ? c->Cs : 50, gam = 7.0;
    p->dim = dim;
    p->variant = c->program == SPH_PROGRAM_WCSPH ? SPH_VARIANT_WCSPH : SPH_VARIANT_WCXSPH;
    p->bound_parts = wall_count(c);
    uint64_t added = 0;
    visit_added_body(c, [&](const double*, bool) { ++added; });
    p->npts = p->bound_parts + simpts + added;
    /* H (WCXSPH3D.cpp:59 keeps the square root and multiplies by 3; WCXSPH.cpp:56; WCSPH.cpp:61) */
    const double h2d = sqrt((3 / M_PI) * (simmass / rho0));
    p->H = dim == 3 ? 3 * h2d : h2d;
    p->HSQ = p->H * p->H;
    if (dim == 3) {
        const double HCB = p->HSQ * p->H;
        p->correc = 21 / (16 * M_PI * HCB); /* WCXSPH3D.cpp:63 */
    } else {
        p->correc = 7 / (4 * M_PI * p->HSQ); /* written inline at WCXSPH.cpp:120,129 */
    }
    p->search_radius_sq = 4 * p->HSQ; /* WCXSPH3D.cpp:97, WCXSPH.cpp:92 */
    if (c->program == SPH_PROGRAM_WCSPH) p->search_radius_sq = 4 * p->HSQ + 0.001 * p->H; /* WCSPH.cpp:96 */
    p->rho0 = rho0;
    p->Cs = Cs;
    p->gam = gam;
    p->B = rho0 * pow(Cs, 2) / gam; /* WCXSPH3D.cpp:56 */
    p->alpha = c->alpha > 0 ? c->alpha : 0.025; /* WCXSPH3D.cpp:142 */
    p->eps = 0.2;                   /* WCXSPH3D.cpp:143 */
    p->g = c->gravity > 0 ? -c->gravity : -9.81;
    p->g_axis = dim - 1;            /* WCXSPH3D.cpp:195, WCXSPH.cpp:191 */
    p->subits = 8;
    const float beta = 0.25;        /* a float in the reference, WCXSPH3D.cpp:72 */
    p->beta = beta;
    p->mass = simmass;
    p->dt0 = c->dt0;
    p->device = -1;
    if (simmass_out) *simmass_out = simmass;
    if (boundmass_out) *boundmass_out = boundmass;
    return SPH_OK;
}

int sph_case_init(const sph_case* c, const sph_fields* dst) {
    if (!c || !dst) return SPH_EINVAL;
    sph_params p;
    double simmass, boundmass;
    const int rc = sph_case_params(c, &p, &simmass, &boundmass);
    if (rc != SPH_OK) return rc;
    const int dim = p.dim;
    const float rho = (float)p.rho0;
    const float wall_m = (float)boundmass, fluid_m = (float)simmass;
    uint64_t i = 0;
    visit_all(c, [&](const double* x, bool wall, bool added) {
        for (int d = 0; d < dim; ++d) {
            if (dst->x[d]) dst->x[d][i] = x[d];
            if (dst->v[d]) dst->v[d][i] = added ? c->drop_v[d] : 0.0;
            if (dst->V[d]) dst->V[d][i] = 0.0;
            if (dst->f[d]) dst->f[d][i] = 0.0;
        }
        if (dst->rho) dst->rho[i] = rho;
        if (dst->p) dst->p[i] = 0.0f;
        if (dst->Rrho) dst->Rrho[i] = 0.0f;
        if (dst->m) dst->m[i] = wall ? wall_m : fluid_m;
        ++i;
    });
    return i == p.npts ? SPH_OK : SPH_EINVAL;
}

int sph_case_init_range(const sph_case* c, double xlo, double xhi, const sph_fields* dst, uint32_t* gids,
                        uint64_t* count) {
    if (!c || !count) return SPH_EINVAL;
    sph_params p;
    double simmass, boundmass;
    const int rc = sph_case_params(c, &p, &simmass, &boundmass);
    if (rc != SPH_OK) return rc;
    if (p.npts >= (1ull << 32)) return SPH_EINVAL;
    const int dim = p.dim;
    const float rho = (float)p.rho0;
    const float wall_m = (float)boundmass, fluid_m = (float)simmass;
    uint64_t i = 0, g = 0;
    visit_all(c, [&](const double* x, bool wall, bool added) {
        const uint64_t gi = g++;
        if (!(x[0] >= xlo && x[0] < xhi)) return;
        if (dst) {
            for (int d = 0; d < dim; ++d) {
                if (dst->x[d]) dst->x[d][i] = x[d];
                if (dst->v[d]) dst->v[d][i] = added ? c->drop_v[d] : 0.0;
                if (dst->V[d]) dst->V[d][i] = 0.0;
                if (dst->f[d]) dst->f[d][i] = 0.0;
            }
            if (dst->rho) dst->rho[i] = rho;
            if (dst->p) dst->p[i] = 0.0f;
            if (dst->Rrho) dst->Rrho[i] = 0.0f;
            if (dst->m) dst->m[i] = wall ? wall_m : fluid_m;
        }
        if (gids) gids[i] = (uint32_t)gi;
        ++i;
    });
    *count = i;
    return SPH_OK;
}

int sph_case_histogram_x(const sph_case* c, int32_t nbins, uint64_t* fluid_counts, uint64_t* wall_counts) {
    if (!c || nbins < 1 || !fluid_counts || !wall_counts) return SPH_EINVAL;
    sph_params p;
    const int rc = sph_case_params(c, &p, nullptr, nullptr);
    if (rc != SPH_OK) return rc;
    for (int b = 0; b < nbins; ++b) fluid_counts[b] = wall_counts[b] = 0;
    const double scale = nbins / c->box[0];
    visit_all(c, [&](const double* x, bool wall, bool) {
        double b = floor(x[0] * scale);
        if (!(b >= 0)) b = 0;
        if (b > nbins - 1) b = nbins - 1;
        (wall ? wall_counts : fluid_counts)[(int)b] += 1;
    });
    return SPH_OK;
}

}  /* extern "C" */
