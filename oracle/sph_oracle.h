/*
 * CPU oracle for the SPH particle-pair path -- TEST INFRASTRUCTURE ONLY.
 *
 * A dependency-free restatement (no Eigen, no nanoflann) of the algorithm in the
 * reference programs WCSPH.cpp / WCXSPH.cpp / WCXSPH3D.cpp.  It is the checker that
 * the HIP path is compared with; nothing in the shipped library calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures, and it
 * cannot be compiled here (Eigen3 and nanoflann are not vendored and not installed).
 * The oracle is pinned only by known answers that follow from the reference source
 * (particle counts, constants, t=0 forces, first-sweep closed form) and by agreement
 * between its two neighbour back-ends (brute force and cell list).
 */
#ifndef SPH_ORACLE_H
#define SPH_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Which reference program's physics. */
enum { ORC_WCSPH = 0, ORC_WCXSPH = 1, ORC_WCXSPH3D = 2 };

/* Which particle vector: `particles` (state n) or `particlesh` (state n+1). */
enum { ORC_STATE_N = 0, ORC_STATE_NP1 = 1 };

/* Neighbour back-end. */
enum { ORC_SEARCH_BRUTE = 0, ORC_SEARCH_CELLS = 1 };

/* The compile-time "Simulation Definition" block of the reference
 * (WCXSPH3D.cpp:39-48, WCXSPH.cpp:36-45, WCSPH.cpp:42-51), made run-time. */
typedef struct orc_case {
    int32_t variant;      /* ORC_* */
    int32_t npart[3];     /* xyPART (third entry ignored in 2-D) */
    double  pstep;        /* Pstep */
    double  box[3];       /* Box */
    double  start[3];     /* Start */
    double  dt0;          /* initial dt (reference: 0.0002) */
    /* Not in the reference (its InitSPH only makes the box and one block): a second body of
     * fluid on the same lattice spacing, listed after the block.  simtype 0: none; 1: the
     * lattice points within drop_radius of drop_centre (droplet); 2: those within drop_radius
     * of the vertical axis through drop_centre and within drop_height/2 of it along the last
     * axis (jet).  All of them start with velocity drop_v. */
    int32_t simtype;
    int32_t reserved_;
    double  drop_radius;
    double  drop_centre[3];
    double  drop_height;
    double  drop_v[3];
    /* The reference's compile-time constants, changeable the way a settings file would: 0 keeps
     * the reference's value (rho0 1000, Cs 50, alpha 0.025, gravity 9.81 pointing down). */
    double  rho0, Cs, alpha, gravity;
} orc_case;

/* Derived constants (WCXSPH3D.cpp:50-79,97 and the parallel 2-D lines). */
typedef struct orc_consts {
    int32_t  dim;
    uint64_t sim_pts, bound_parts, npts;
    double   simmass, boundmass;
    double   rho0, Cs, gam, B;
    double   H, HSQ, correc;
    double   search_radius;   /* squared length, as in the reference */
    double   alpha, eps;
    double   beta;            /* (double)0.25f */
    int32_t  subits;
    int32_t  g_axis;
    double   g;               /* -9.81 */
} orc_consts;

typedef struct orc_sim orc_sim;

/* Fills the reference's default case for a variant (the constants as shipped). */
void orc_default_case(int32_t variant, orc_case* out);

int  orc_derive(const orc_case* c, orc_consts* out);

/* InitSPH + the copy into particlesh (WCXSPH3D.cpp:448-451). Does NOT run Forces. */
orc_sim* orc_create(const orc_case* c);
void     orc_destroy(orc_sim* s);
void     orc_get_consts(const orc_sim* s, orc_consts* out);
void     orc_set_search(orc_sim* s, int32_t method);

/* Field access, particle-major: x,v,V,f are [npts][dim] doubles; rho,p,Rrho,m are
 * [npts] floats.  NULL pointers are skipped. */
void orc_get_state(const orc_sim* s, int32_t which, double* x, double* v, double* V,
                   double* f, float* rho, float* p, float* Rrho, float* m);
void orc_set_state(orc_sim* s, int32_t which, const double* x, const double* v,
                   const double* V, const double* f, const float* rho, const float* p,
                   const float* Rrho, const float* m);

/* Forces() on one state (WCXSPH3D.cpp:139-205). */
void orc_forces(orc_sim* s, int32_t which);
/* main():457-458 -- zero f of the wall particles. */
void orc_zero_boundary_f(orc_sim* s, int32_t which);
/* nsteps x Newmark_Beta(particles, particlesh) (WCXSPH3D.cpp:251-303).
 * max_sweeps < subits stops the FIRST step after that many sweeps (no dt update, no
 * commit) so single sweeps can be inspected; pass a negative value for whole steps. */
void orc_step(orc_sim* s, int32_t nsteps, int32_t max_sweeps);
/* DensityReinit(particlesh) (WCXSPH3D.cpp:208-249). */
void orc_density_reinit(orc_sim* s);

/* Scalars: dt, t, maxmu, maxf, errsum, logbase. */
void orc_get_scalars(const orc_sim* s, double* dt, float* t, double* maxmu, double* maxf,
                     double* errsum, double* logbase);
void orc_set_dt(orc_sim* s, double dt);

/* Fixed-radius neighbour sets of one state as CSR, columns ascending by index.
 * row_offsets has npts+1 entries.  Call with cols == NULL to get the total in
 * row_offsets[npts] only. */
void orc_neighbours(orc_sim* s, int32_t which, int32_t method, uint64_t* row_offsets,
                    uint32_t* cols);

/* Kernel and gradient factor for unit tests: W2Kernel(r) and the scalar g(r) such that
 * W2GradK(Rij,r) = Rij-componentwise formula; grad_out gets dim components. */
double orc_w2kernel(const orc_sim* s, double r);
void   orc_w2gradk(const orc_sim* s, const double* rij, double r, double* grad_out);

#ifdef __cplusplus
}
#endif
#endif
