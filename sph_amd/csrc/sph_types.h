/* Internal device/host types shared by the kernels and the context. */
#ifndef SPH_TYPES_H
#define SPH_TYPES_H

#include <stdint.h>

/* Pair-loop input of one particle of state n+1: everything a neighbour contributes, in one
 * 64-byte line so that a gather touches a single aligned segment.  Precisions as in the
 * reference's struct Particle (WCXSPH3D.cpp:85-86): x, v double; rho, p, m float. */
struct __attribute__((aligned(16))) Rec {
    double xv[3][2]; /* 16-byte chunk d = { x_d, v_d } */
    float  rho, m;   /* chunk 3 */
    double prr;      /* (double)p / ((double)rho*(double)rho): the reference's
                        pi.p/pow(pi.rho,2) (WCXSPH3D.cpp:180), formed once per particle */
};
static_assert(sizeof(Rec) == 64, "Rec must be one 64-byte line");

/* Run constants as the kernels take them (by value). */
struct DevParams {
    int32_t  dim, xsph;
    uint32_t n, bound_parts;
    double   H, HSQ, correc, R2;
    double   Bgam;      /* B*gam, the numerator of the sound-speed term (WCXSPH3D.cpp:172) */
    double   B, gam, rho0, Cs;
    double   alpha, eps, g;
    int32_t  g_axis, subits;
    double   beta;      /* (double)beta_f */
    double   omb;       /* (double)(1 - 2*beta_f), evaluated in float as the reference does */
    double   mass;
    double   hvisc;     /* 0.01*HSQ */
    double   npts_d;    /* 1.0*npts */
    double   sqrtBgam;  /* sqrt(B*gam) */
    double   invH;      /* 1/H */
    double   gconst;    /* 5*(1/HSQ)*correc: the gradient's constant factor (WCXSPH3D.cpp:135) */
};

/* Uniform grid over the particles' bounding box, cell edge >= candidate radius, and the order
 * the particles are sorted into.  The last axis (z in 3-D, y in 2-D) is the RUN axis: cells are
 * cut `zf` times finer along it and are the fastest-varying part of the sort key, so that the
 * particles of one column of cells (one cell across in the other axes) are contiguous and
 * sorted along the run axis.  A neighbour query then reads ONE contiguous range per
 * neighbouring column -- the part of the column that can be within reach -- instead of three
 * whole cells.  In 3-D the columns themselves are grouped in bricks of 2^ts x 2^ts columns,
 * bricks numbered with x slowest (a slab in x is a contiguous key range). */
struct GridParams {
    double   lo[3];
    double   inv[3];  /* 1 / cell edge (all axes share one edge) */
    int32_t  n[3];    /* cells per axis; the column axes padded to whole bricks */
    int32_t  nt[3];   /* bricks per column axis */
    uint32_t ts[3];   /* log2 of the brick edge in cells, column axes */
    uint32_t ncells;  /* columns * nrun: the sort key's range */
    uint32_t zf;      /* fine cells per cell along the run axis */
    uint32_t nrun;    /* fine cells per column = n[run axis] * zf */
    double   invf;    /* 1 / fine cell edge */
};

/* Device-resident run state: the reference's file-scope globals (WCXSPH3D.cpp:70-79) plus
 * the accumulators of the sweep in flight. */
struct DevScalars {
    double dt;        /* time step in use */
    double dt_next;   /* set by the last sweep, adopted at commit */
    double maxmu, maxf, errsum, logbase;
    double errsum0;   /* errsum of the step's first sweep (the logbase numerator) */
    double disp;      /* upper bound of any particle's travel since the candidate lists were built */
    double disp_limit;
    float  t;
    int32_t violation;            /* a sweep started with disp > disp_limit */
    unsigned long long hit_pairs; /* i != j pairs inside the radius in the last finished sweep */
    double disp_in;   /* travel bound since the step's working lists were drawn; its limit */
    double limit_in;
};

#endif
