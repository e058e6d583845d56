/* Timing experiments on the pair kernel (tools/ablate_pair.py): given to the compiler in front of
 * the library's sources,
 *   make -C sph_amd/csrc -B HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -include ../../tools/pair_hooks.h -DSPH_ABLATE=n" OUT=...
 * SPH_ABLATE & 1: every fetch aimed at the particle's own record (what perfectly coalesced fetches
 * would cost); SPH_ABLATE & 2: the arithmetic of every other round skipped (what evaluating each
 * pair once would save).  Results of such a build are wrong on purpose. */
#ifndef PAIR_HOOKS_H
#define PAIR_HOOKS_H
#if defined(SPH_ABLATE) && (SPH_ABLATE & 1)
#define PAIR_HOOK_FETCH(j, i) j = i;
#endif
#if defined(SPH_ABLATE) && (SPH_ABLATE & 2)
#define PAIR_HOOK_HIT(hit, t) hit = hit && !((t) & 1u);
#endif
#endif
